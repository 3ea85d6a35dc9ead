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
  ``align_corners=False`` = cv2's INTER_LINEAR sampling grid for float data); only the final (H, W, classes) array goes
  to the host;
* the per-scale resize of the uint8 inputs (``sliding_eval_rgbX``, engine/evaluator.py:438-446) runs on the device in
  cv2's own integer arithmetic (``resize_like_cv2``), so the network sees the bytes the reference's loop would feed it.

Drop-in for the reference's ``Evaluator`` (same method names, arguments and return values):

    evaluator.val_func_process_rgbX = types.MethodType(evaluator_ops.val_func_process_rgbX, evaluator)
    evaluator.scale_process_rgbX = types.MethodType(evaluator_ops.scale_process_rgbX, evaluator)
    evaluator.sliding_eval_rgbX = types.MethodType(evaluator_ops.sliding_eval_rgbX, evaluator)

The window grid reproduces the reference's arithmetic literally, including its mixed use of ``crop_size[0]`` /
``stride[0]`` for the column direction (evaluator.py:472-478): a drop-in must score the same pixels.  With a crop that
is not square that arithmetic produces NEGATIVE window starts once the scaled image is larger than the crop -- NYU /
SUN-RGBD evaluate at ``eval_scale_array = [0.75, 1, 1.25]`` with ``eval_crop_size = [480, 640]`` and ``eval_flip``
(configs/config_nyu.py:114-117; MFNet and PST900 use ``[1]``), and at 1.25 the 600 x 800 image gets row windows
``[-40:600]`` -- which numpy and torch slicing wrap to the LAST 40 rows; the reference therefore scores only those rows
at that scale.  ``window_grid`` resolves the slices the same way, so the sum over the scales is the reference's.
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


def _as_slice(start: int, stop: int, size: int):
    """what ``array[start:stop]`` selects on an axis of `size` (negative indices count from the end, as numpy's and
    torch's slicing do for the reference's img_pad[s_y:e_y, s_x:e_x] and data_scale[:, s_y:e_y, s_x:e_x])"""
    lo, hi, _ = slice(start, stop).indices(size)
    return lo, max(lo, hi)


def window_grid(pad_rows: int, pad_cols: int, crop, stride_rate: float):
    """(s_y, e_y, s_x, e_x) of every window, in the reference's order and with its index arithmetic
    (engine/evaluator.py:464-478: the column direction uses stride[0] / crop_size[0], the row direction stride[1] /
    crop_size[1]); a negative start is resolved as the reference's slicing resolves it (module docstring)."""
    stride = (int(np.ceil(crop[0] * stride_rate)), int(np.ceil(crop[1] * stride_rate)))
    r_grid = int(np.ceil((pad_rows - crop[0]) / stride[0])) + 1
    c_grid = int(np.ceil((pad_cols - crop[1]) / stride[1])) + 1
    wins = []
    for gy in range(r_grid):
        for gx in range(c_grid):
            s_x, s_y = gx * stride[0], gy * stride[1]
            e_x, e_y = min(s_x + crop[0], pad_cols), min(s_y + crop[1], pad_rows)
            s_x, s_y = e_x - crop[0], e_y - crop[1]
            (s_y, e_y), (s_x, e_x) = _as_slice(s_y, e_y, pad_rows), _as_slice(s_x, e_x, pad_cols)
            if e_y == s_y or e_x == s_x:
                raise ValueError(f"empty window (padded image {pad_rows}x{pad_cols}, crop {tuple(crop)}): the reference's "
                                 f"cv2.copyMakeBorder would fail on it too")
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
                subs, subx, margs = [], [], []
                for (s_y, e_y, s_x, e_x) in chunk:
                    a, marg = padded(rgb_p[:, s_y:e_y, s_x:e_x], e_y - s_y, e_x - s_x)
                    b, _ = padded(mx_p[:, s_y:e_y, s_x:e_x], e_y - s_y, e_x - s_x)
                    subs.append(a)
                    subx.append(b)
                    margs.append(marg)
                sc = batch_scores(self.val_func, torch.stack(subs), torch.stack(subx), flip)
                for k, (s_y, e_y, s_x, e_x) in enumerate(chunk):
                    m = margs[k]
                    data_scale[:, s_y:e_y, s_x:e_x] += sc[k][:, m[0]:sc.shape[2] - m[1], m[2]:sc.shape[3] - m[3]]
            score = data_scale[:, top:data_scale.shape[1] - bottom, left:data_scale.shape[2] - right]
        if tuple(score.shape[1:]) != (int(ori_shape[0]), int(ori_shape[1])):
            score = F.interpolate(score[None], size=(int(ori_shape[0]), int(ori_shape[1])), mode="bilinear", align_corners=False)[0]
        return score.permute(1, 2, 0).contiguous().cpu().numpy()


_COEF_ONE = 1 << 11         # cv2's INTER_RESIZE_COEF_SCALE: bilinear weights of 8-bit images are 11-bit fixed point


def _sample_table(dst: int, src: int, inv_scale: float):
    """Host-side tables of cv::resize's linear path (tiny: one entry per destination row / column, built on the host by
    cv2 as well): source index floor((d + 0.5) / inv_scale - 0.5), evaluated in double and held in float32, and the
    float32 weight of the next sample."""
    pos = ((np.arange(dst, dtype=np.float64) + 0.5) * (1.0 / inv_scale) - 0.5).astype(np.float32)
    base = np.floor(pos)
    return base.astype(np.int64), (pos - base).astype(np.float32)


def _fixed(w: np.ndarray) -> np.ndarray:
    return np.clip(np.rint(w.astype(np.float32) * np.float32(_COEF_ONE)), -32768, 32767).astype(np.int32)


def resize_like_cv2(arr: np.ndarray, scale: float, nearest: bool, device) -> np.ndarray:
    """``cv2.resize(arr, None, fx=scale, fy=scale, interpolation=INTER_NEAREST if nearest else INTER_LINEAR)`` computed
    on the device, bit for bit for uint8 (H, W) / (H, W, C) arrays (OpenCV 4.x modules/imgproc/src/resize.cpp, the
    non-IPP path the reference's evaluation runs; oracle/evaluator_oracle.py restates it one pixel at a time):
    destination size round-half-even(src * scale); INTER_NEAREST index min(floor(d / scale), size - 1); INTER_LINEAR of
    8-bit data with 11-bit fixed-point weights, an int32 horizontal pass and the vertical pass
    (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2; a factor of exactly 1/2 is cv2's 2 x 2 area mean.
    Float32 arrays take the same sampling grid in float arithmetic (not bit-exact: cv2 may use IPP there)."""
    sh, sw = arr.shape[0], arr.shape[1]
    dh, dw = int(np.rint(sh * float(scale))), int(np.rint(sw * float(scale)))
    if dh <= 0 or dw <= 0:
        raise ValueError(f"cv2.resize of a {sh}x{sw} image by {scale} has no pixels")
    if (dh, dw) == (sh, sw):
        return arr.copy()
    if arr.dtype not in (np.uint8, np.float32) and not nearest:
        raise TypeError(f"resize_like_cv2: INTER_LINEAR is restated for uint8 and float32 images, got {arr.dtype}")
    src = torch.from_numpy(np.ascontiguousarray(arr if arr.ndim == 3 else arr[:, :, None])).to(device)

    def dev(a, dtype):
        return torch.as_tensor(a, dtype=dtype, device=device)

    inv = float(scale)
    if nearest:
        xs = np.minimum(np.floor(np.arange(dw) * (1.0 / inv)).astype(np.int64), sw - 1)
        ys = np.minimum(np.floor(np.arange(dh) * (1.0 / inv)).astype(np.int64), sh - 1)
        out = src.index_select(0, dev(ys, torch.long)).index_select(1, dev(xs, torch.long))
    elif arr.dtype == np.uint8 and abs(1.0 / inv - 2.0) < np.finfo(np.float64).eps:
        # cv::resize turns INTER_LINEAR into the fast INTER_AREA for an exact 2 x 2 reduction
        px = F.pad(src.permute(2, 0, 1).to(torch.int32), (0, 2 * dw - sw if 2 * dw > sw else 0, 0, 2 * dh - sh if 2 * dh > sh else 0))
        px = px[:, :2 * dh, :2 * dw]
        ones = torch.zeros(1, px.shape[1], px.shape[2], dtype=torch.int32, device=device)
        ones[:, :sh, :sw] = 1
        cells = lambda t: t[:, 0::2, 0::2] + t[:, 0::2, 1::2] + t[:, 1::2, 0::2] + t[:, 1::2, 1::2]
        total, count = cells(px), cells(ones)
        # a cell counts as "full" where cv2's fast loop covers it: both rows inside and dx < src_width / 2
        full = count == 4
        mean = torch.round(total.float() / count.clamp(min=1).float()).to(torch.int32)
        out = torch.where(full, (total + 2) >> 2, torch.where(count > 0, mean, torch.zeros_like(total)))
        out = out.clamp_(0, 255).to(torch.uint8).permute(1, 2, 0)
    else:
        xb, xt = _sample_table(dw, sw, inv)
        yb, yt = _sample_table(dh, sh, inv)
        edge = (xb < 0) | (xb >= sw - 1)                 # columns: index clamped AND the weight pair forced to (1, 0)
        xt = np.where(edge, np.float32(0), xt)
        x0 = np.clip(xb, 0, sw - 1)
        x1 = np.minimum(x0 + 1, sw - 1)
        y0, y1 = np.clip(yb, 0, sh - 1), np.clip(yb + 1, 0, sh - 1)      # rows: indices clipped, weights kept
        if arr.dtype == np.uint8:
            a0, a1 = dev(_fixed(np.float32(1) - xt), torch.int32), dev(_fixed(xt), torch.int32)
            b0, b1 = dev(_fixed(np.float32(1) - yt), torch.int32), dev(_fixed(yt), torch.int32)
            s32 = src.to(torch.int32)
            rows = s32.index_select(1, dev(x0, torch.long)) * a0[None, :, None] + s32.index_select(1, dev(x1, torch.long)) * a1[None, :, None]
            rows = rows >> 4
            out = (((rows.index_select(0, dev(y0, torch.long)) * b0[:, None, None]) >> 16)
                   + ((rows.index_select(0, dev(y1, torch.long)) * b1[:, None, None]) >> 16) + 2) >> 2
            out = out.clamp_(0, 255).to(torch.uint8)
        else:
            a1, b1 = dev(xt, torch.float32), dev(yt, torch.float32)
            rows = src.index_select(1, dev(x0, torch.long)) * (1 - a1)[None, :, None] + src.index_select(1, dev(x1, torch.long)) * a1[None, :, None]
            out = rows.index_select(0, dev(y0, torch.long)) * (1 - b1)[:, None, None] + rows.index_select(0, dev(y1, torch.long)) * b1[:, None, None]
    out = out.contiguous().cpu().numpy()
    return out[:, :, 0] if arr.ndim == 2 else out


def sliding_eval_rgbX(self, img, modal_x, crop_size, stride_rate, device=None):
    """Mirror of engine/evaluator.py:432-450: sum of the per-scale scores, arg-max."""
    crop = (crop_size, crop_size) if isinstance(crop_size, int) else tuple(crop_size)
    ori_rows, ori_cols, _ = img.shape
    processed = np.zeros((ori_rows, ori_cols, self.class_num))
    dev = torch.device("cuda", torch.cuda.current_device() if device is None else device)
    for s in self.multi_scales:
        img_scale = resize_like_cv2(img, s, False, dev)
        mx_scale = resize_like_cv2(modal_x, s, modal_x.ndim == 2, dev)
        processed += scale_process_rgbX(self, img_scale, mx_scale, (ori_rows, ori_cols), crop, stride_rate, device)
    return processed.argmax(2)
