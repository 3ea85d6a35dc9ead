"""The evaluator's hot loop on the GPU.

``Evaluator.val_func_process_rgbX`` (engine/evaluator.py:501-522) runs the network once on the image pair and, with
``is_flip``, a second time on the horizontally flipped pair, then adds the un-flipped second score.
``Evaluator.scale_process_rgbX`` (engine/evaluator.py:452-499) calls it once per sliding window -- numpy slicing,
normalisation and padding per window, one host-to-device copy and one or two batch-1 forwards each, a device-to-host
copy and a CPU ``cv2.resize`` of the class scores per scale.  Here:

* the plain and the flipped pass are ONE batch (``flip_pair_scores``);
* every window of a scale is cut from ONE normalised, padded device image and the windows go through the network
  as batches (``scale_process_rgbX``: window batch x flip = up to ``2 * max_windows`` images per forward);
* the window scores are accumulated on the device and resized there (``F.interpolate``, bilinear,
  ``align_corners=False`` = cv2's INTER_LINEAR sampling grid); only the final (H, W, classes) array goes to the host.

Drop-in for the reference's ``Evaluator`` (same method names, arguments and return values):

    evaluator.val_func_process_rgbX = types.MethodType(evaluator_ops.val_func_process_rgbX, evaluator)
    evaluator.scale_process_rgbX = types.MethodType(evaluator_ops.scale_process_rgbX, evaluator)
    evaluator.sliding_eval_rgbX = types.MethodType(evaluator_ops.sliding_eval_rgbX, evaluator)

The window grid reproduces the reference's arithmetic literally, including its mixed use of ``crop_size[0]`` /
``stride[0]`` for the column direction (evaluator.py:472-478): a drop-in must score the same pixels.  With the
reference's configurations (``eval_scale_array = [1]``) nothing is resized before the network, so the scores are those
of the reference loop to rounding; at other scales cv2's fixed-point uint8 resize differs from the float resize used
here in the last bit of some input pixels.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F


def flip_pair_scores(model, rgb: torch.Tensor, modal_x: torch.Tensor, is_flip: bool) -> torch.Tensor:
    """score[0] (+ flipped score un-flipped) for a batch-1 pair, computed in one forward when ``is_flip``."""
    if not is_flip:
        return model(rgb, modal_x)[0]
    both = model(torch.cat([rgb, rgb.flip(-1)], dim=0), torch.cat([modal_x, modal_x.flip(-1)], dim=0))
    return both[0] + both[1].flip(-1)


def batch_scores(model, rgb: torch.Tensor, modal_x: torch.Tensor, is_flip: bool) -> torch.Tensor:
    """exp(score (+ un-flipped score of the flipped input)) for a batch of pairs, flips riding in the same forward"""
    n = rgb.shape[0]
    if not is_flip:
        return torch.exp(model(rgb, modal_x))
    both = model(torch.cat([rgb, rgb.flip(-1)], dim=0), torch.cat([modal_x, modal_x.flip(-1)], dim=0))
    return torch.exp(both[:n] + both[n:].flip(-1))


def val_func_process_rgbX(self, input_data, input_modal_x, device=None):
    """Mirror of engine/evaluator.py:501-522 (``self`` = the reference's Evaluator)."""
    input_data = torch.from_numpy(np.ascontiguousarray(input_data[None, :, :, :], dtype=np.float32)).cuda(device)
    input_modal_x = torch.from_numpy(np.ascontiguousarray(input_modal_x[None, :, :, :], dtype=np.float32)).cuda(device)
    with torch.cuda.device(input_data.get_device()):
        self.val_func.eval()
        self.val_func.to(input_data.get_device())
        with torch.no_grad():
            score = flip_pair_scores(self.val_func, input_data, input_modal_x, bool(self.is_flip))
            score = torch.exp(score)
    return score


def _pad_margin(rows: int, cols: int, crop):
    """utils/transforms.py:61-75 (pad_image_to_shape): centred zero padding up to `crop`, margins (top, bottom, left, right)"""
    ph = crop[0] - rows if crop[0] - rows > 0 else 0
    pw = crop[1] - cols if crop[1] - cols > 0 else 0
    return ph // 2, ph // 2 + ph % 2, pw // 2, pw // 2 + pw % 2


def _normalized_planes(img: np.ndarray, mean, std, device) -> torch.Tensor:
    """utils/transforms.py:182-187 (normalize, float64 on the host like the reference) -> (C, H, W) float32 on the device"""
    arr = img.astype(np.float64) / 255.0
    arr = (arr - mean) / std
    if arr.ndim == 2:
        arr = arr[:, :, None]
    return torch.from_numpy(np.ascontiguousarray(arr.transpose(2, 0, 1), dtype=np.float32)).cuda(device)


def window_grid(pad_rows: int, pad_cols: int, crop, stride_rate: float):
    """(s_y, e_y, s_x, e_x) of every window, in the reference's order and with its index arithmetic
    (engine/evaluator.py:464-478: the column direction uses stride[0] / crop_size[0], the row direction stride[1] /
    crop_size[1])."""
    stride = (int(np.ceil(crop[0] * stride_rate)), int(np.ceil(crop[1] * stride_rate)))
    r_grid = int(np.ceil((pad_rows - crop[0]) / stride[0])) + 1
    c_grid = int(np.ceil((pad_cols - crop[1]) / stride[1])) + 1
    wins = []
    for gy in range(r_grid):
        for gx in range(c_grid):
            s_x, s_y = gx * stride[0], gy * stride[1]
            e_x, e_y = min(s_x + crop[0], pad_cols), min(s_y + crop[1], pad_rows)
            s_x, s_y = e_x - crop[0], e_y - crop[1]
            if s_x < 0 or s_y < 0:
                raise ValueError(f"window start ({s_y}, {s_x}) is negative: the reference's slicing would wrap around "
                                 f"(padded image {pad_rows}x{pad_cols}, crop {tuple(crop)})")
            wins.append((s_y, e_y, s_x, e_x))
    return wins


def scale_process_rgbX(self, img, modal_x, ori_shape, crop_size, stride_rate, device=None, max_windows: int = 4):
    """Mirror of engine/evaluator.py:452-499 for one scale; returns the (ori_rows, ori_cols, classes) float32 array."""
    crop = (int(crop_size[0]), int(crop_size[1]))
    new_rows, new_cols = img.shape[0], img.shape[1]
    self.val_func.eval()
    self.val_func.cuda(device)
    dev = next(self.val_func.parameters()).device
    rgb = _normalized_planes(img if img.shape[2] >= 3 else np.concatenate((img, img, img), axis=2), self.norm_mean, self.norm_std, dev)
    mx = _normalized_planes(modal_x, 0, 1, dev) if modal_x.ndim == 2 else _normalized_planes(modal_x, self.norm_mean, self.norm_std, dev)
    flip = bool(self.is_flip)

    def padded(t, rows, cols):      # zero padding AFTER the normalisation, as process_image_rgbX does
        top, bottom, left, right = _pad_margin(rows, cols, crop)
        return F.pad(t, (left, right, top, bottom)), (top, bottom, left, right)

    with torch.no_grad(), torch.cuda.device(dev):
        if new_cols <= crop[1] or new_rows <= crop[0]:
            a, m = padded(rgb, new_rows, new_cols)
            b, _ = padded(mx, new_rows, new_cols)
            score = batch_scores(self.val_func, a[None], b[None], flip)[0]
            score = score[:, m[0]:score.shape[1] - m[1], m[2]:score.shape[2] - m[3]]
        else:
            # pad_image_to_shape of the RAW image with zeros, then normalisation per window: a zero raw pixel normalises
            # to (0 - mean) / std, so the big image is padded with that value (only where it is smaller than the crop)
            top, bottom, left, right = _pad_margin(new_rows, new_cols, crop)
            def pad_raw(t, mean, std):
                if top + bottom + left + right == 0:
                    return t
                fill = torch.as_tensor((0.0 - np.asarray(mean, dtype=np.float64)) / np.asarray(std, dtype=np.float64), dtype=torch.float32,
                                       device=t.device).reshape(-1, 1, 1)
                out = fill.expand(t.shape[0], t.shape[1] + top + bottom, t.shape[2] + left + right).clone()
                out[:, top:top + t.shape[1], left:left + t.shape[2]] = t
                return out
            rgb_p = pad_raw(rgb, self.norm_mean, self.norm_std)
            mx_p = pad_raw(mx, 0, 1) if modal_x.ndim == 2 else pad_raw(mx, self.norm_mean, self.norm_std)
            pad_rows, pad_cols = rgb_p.shape[1], rgb_p.shape[2]
            wins = window_grid(pad_rows, pad_cols, crop, stride_rate)
            data_scale = torch.zeros(self.class_num, pad_rows, pad_cols, device=dev)
            for i in range(0, len(wins), max_windows):
                chunk = wins[i:i + max_windows]
                subs, subx, marg = [], [], None
                for (s_y, e_y, s_x, e_x) in chunk:
                    a, marg = padded(rgb_p[:, s_y:e_y, s_x:e_x], e_y - s_y, e_x - s_x)
                    b, _ = padded(mx_p[:, s_y:e_y, s_x:e_x], e_y - s_y, e_x - s_x)
                    subs.append(a)
                    subx.append(b)
                sc = batch_scores(self.val_func, torch.stack(subs), torch.stack(subx), flip)
                sc = sc[:, :, marg[0]:sc.shape[2] - marg[1], marg[2]:sc.shape[3] - marg[3]]
                for k, (s_y, e_y, s_x, e_x) in enumerate(chunk):
                    data_scale[:, s_y:e_y, s_x:e_x] += sc[k]
            score = data_scale[:, top:data_scale.shape[1] - bottom, left:data_scale.shape[2] - right]
        if tuple(score.shape[1:]) != (int(ori_shape[0]), int(ori_shape[1])):
            score = F.interpolate(score[None], size=(int(ori_shape[0]), int(ori_shape[1])), mode="bilinear", align_corners=False)[0]
        return score.permute(1, 2, 0).contiguous().cpu().numpy()


def _resize_hw(arr: np.ndarray, scale: float, nearest: bool, device) -> np.ndarray:
    """cv2.resize(arr, None, fx=scale, fy=scale, INTER_LINEAR | INTER_NEAREST) on the device (float arithmetic)"""
    if scale == 1:
        return arr
    t = torch.from_numpy(np.ascontiguousarray(arr)).to(device)
    hw = t if t.ndim == 2 else t.permute(2, 0, 1)
    hw = hw[None, None] if t.ndim == 2 else hw[None]
    size = (int(round(arr.shape[0] * scale)), int(round(arr.shape[1] * scale)))
    if nearest:
        out = F.interpolate(hw.float(), size=size, mode="nearest")
    else:
        out = F.interpolate(hw.float(), size=size, mode="bilinear", align_corners=False)
    if arr.dtype == np.uint8:
        out = out.round().clamp_(0, 255)
    out = out[0, 0] if t.ndim == 2 else out[0].permute(1, 2, 0)
    return out.to(t.dtype).cpu().numpy()


def sliding_eval_rgbX(self, img, modal_x, crop_size, stride_rate, device=None):
    """Mirror of engine/evaluator.py:432-450: sum of the per-scale scores, arg-max."""
    crop = (crop_size, crop_size) if isinstance(crop_size, int) else tuple(crop_size)
    ori_rows, ori_cols, _ = img.shape
    processed = np.zeros((ori_rows, ori_cols, self.class_num))
    dev = torch.device("cuda", torch.cuda.current_device() if device is None else device)
    for s in self.multi_scales:
        img_scale = _resize_hw(img, s, False, dev)
        mx_scale = _resize_hw(modal_x, s, modal_x.ndim == 2, dev)
        processed += scale_process_rgbX(self, img_scale, mx_scale, (ori_rows, ori_cols), crop, stride_rate, device)
    return processed.argmax(2)
