"""TEST INFRASTRUCTURE ONLY: literal restatement of the reference's sliding-window evaluation
(engine/evaluator.py:432-450 sliding_eval_rgbX, :452-499 scale_process_rgbX, :501-522 val_func_process_rgbX, :525-560
process_image_rgbX, utils/transforms.py:61-75 pad_image_to_shape, :182-187 normalize) -- one window at a time, numpy
slicing / normalising / padding per window, one batch-1 forward per window (two with is_flip), scores accumulated window
by window, one pass per entry of multi_scales.

cv2 (opencv-python >= 4.5.0, requirements.txt:4) is a dependency that is neither vendored in the reference nor installed
in this image.  What the path needs of it is restated here from OpenCV's published algorithm
(modules/imgproc/src/resize.cpp of the 4.x line; the function names below are that file's):
  * cv2.resize(u8, None, fx, fy, INTER_LINEAR): destination size cvRound(src * f) (round half to even); sample position
    (d + 0.5) / f - 0.5 evaluated in double and rounded to float; coefficients 1 - t, t rounded to 11-bit fixed point
    (INTER_RESIZE_COEF_BITS); horizontal pass in int32 (HResizeLinear), vertical pass
    (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2 (VResizeLinear<uchar, int, short, ...>); rows clipped
    into the image, columns clamped with the coefficient pair forced to (1, 0) (resize(): the xofs / ialpha loop);
    a factor of exactly 1/2 in both directions is turned into INTER_AREA's 2 x 2 mean, (a + b + c + d + 2) >> 2
    (ResizeAreaFast_Invoker), a factor of 1 is a copy;
  * cv2.resize(., INTER_NEAREST): source index min(floor(d * (1 / f)), size - 1) (resizeNN);
  * cv2.resize(f32, dsize, INTER_LINEAR): the same sampling positions with f = dsize / ssize, float coefficients, the
    horizontal pass then the vertical pass in float32;
  * cv2.copyMakeBorder(BORDER_CONSTANT, 0) is np.pad.
The IPP branch of cv::resize is not taken for 8-bit linear or for nearest (ipp_resize(): "doesn't match OpenCV
exactly" unless useIPP_NotExact()), so the C++ path above is what the reference's evaluation runs.
Pinned by tests/test_evaluator_oracle.py against hand-derived known answers (identity, 2x up / down of a ramp, the
fixed-point rounding cases, the 1/2 special case); parity UNPINNED against a cv2 binary, which this image does not have.
Used by tests/test_model_gpu.py to check sigma_amd/engine/evaluator_ops.py."""
import numpy as np
import torch


def normalize(img, mean, std):
    img = img.astype(np.float64) / 255.0
    img = img - mean
    img = img / std
    return img


def pad_image_to_shape(img, shape):
    ph = shape[0] - img.shape[0] if shape[0] - img.shape[0] > 0 else 0
    pw = shape[1] - img.shape[1] if shape[1] - img.shape[1] > 0 else 0
    margin = (ph // 2, ph // 2 + ph % 2, pw // 2, pw // 2 + pw % 2)
    pads = ((margin[0], margin[1]), (margin[2], margin[3])) + (((0, 0),) if img.ndim == 3 else ())
    return np.pad(img, pads, mode="constant", constant_values=0), margin


def process_image_rgbX(img, modal_x, crop_size, mean, std):
    p_img = normalize(img, mean, std)
    p_x = normalize(modal_x, 0, 1) if modal_x.ndim == 2 else normalize(modal_x, mean, std)
    p_img, margin = pad_image_to_shape(p_img, crop_size)
    p_x, _ = pad_image_to_shape(p_x, crop_size)
    p_img = p_img.transpose(2, 0, 1)
    p_x = p_x[np.newaxis, ...] if modal_x.ndim == 2 else p_x.transpose(2, 0, 1)
    return p_img, p_x, margin


def val_func_process_rgbX(model, input_data, input_modal_x, is_flip, device):
    a = torch.FloatTensor(np.ascontiguousarray(input_data[None], dtype=np.float32)).to(device)
    b = torch.FloatTensor(np.ascontiguousarray(input_modal_x[None], dtype=np.float32)).to(device)
    with torch.no_grad():
        score = model(a, b)[0]
        if is_flip:
            score = score + model(a.flip(-1), b.flip(-1))[0].flip(-1)
        return torch.exp(score)


def scale_process_rgbX(model, img, modal_x, ori_shape, crop_size, stride_rate, class_num, mean, std, is_flip, device):
    new_rows, new_cols, _ = img.shape
    if new_cols <= crop_size[1] or new_rows <= crop_size[0]:
        a, b, margin = process_image_rgbX(img, modal_x, crop_size, mean, std)
        score = val_func_process_rgbX(model, a, b, is_flip, device)
        score = score[:, margin[0]:(score.shape[1] - margin[1]), margin[2]:(score.shape[2] - margin[3])]
    else:
        stride = (int(np.ceil(crop_size[0] * stride_rate)), int(np.ceil(crop_size[1] * stride_rate)))
        img_pad, margin = pad_image_to_shape(img, crop_size)
        x_pad, margin = pad_image_to_shape(modal_x, crop_size)
        pad_rows, pad_cols = img_pad.shape[0], img_pad.shape[1]
        r_grid = int(np.ceil((pad_rows - crop_size[0]) / stride[0])) + 1
        c_grid = int(np.ceil((pad_cols - crop_size[1]) / stride[1])) + 1
        data_scale = torch.zeros(class_num, pad_rows, pad_cols, device=device)
        for gy in range(r_grid):
            for gx in range(c_grid):
                s_x = gx * stride[0]
                s_y = gy * stride[1]
                e_x = min(s_x + crop_size[0], pad_cols)
                e_y = min(s_y + crop_size[1], pad_rows)
                s_x = e_x - crop_size[0]
                s_y = e_y - crop_size[1]
                img_sub = img_pad[s_y:e_y, s_x:e_x, :]
                x_sub = x_pad[s_y:e_y, s_x:e_x] if x_pad.ndim == 2 else x_pad[s_y:e_y, s_x:e_x, :]
                a, b, tm = process_image_rgbX(img_sub, x_sub, crop_size, mean, std)
                t = val_func_process_rgbX(model, a, b, is_flip, device)
                t = t[:, tm[0]:(t.shape[1] - tm[1]), tm[2]:(t.shape[2] - tm[3])]
                data_scale[:, s_y:e_y, s_x:e_x] += t
        score = data_scale[:, margin[0]:(data_scale.shape[1] - margin[1]), margin[2]:(data_scale.shape[2] - margin[3])]
    score = score.permute(1, 2, 0)
    return cv2_resize(np.ascontiguousarray(score.cpu().numpy()), (ori_shape[1], ori_shape[0]))


# ---- cv2.resize restated (see the header) ---------------------------------------------------------------------------
_COEF_BITS = 11                      # INTER_RESIZE_COEF_BITS
_COEF_ONE = 1 << _COEF_BITS          # INTER_RESIZE_COEF_SCALE


def cv_round(v):
    """cvRound / saturate_cast<int>(double): nearest integer, halves to even"""
    return int(np.rint(v))


def _linear_table(dst, src, scale):
    """resize(): per destination index the source index (clamped, as the xofs loop does for columns) and the float32
    weight of the NEXT sample; `scale` = 1 / inv_scale.  Returns (index, t, raw index)."""
    idx = np.zeros(dst, dtype=np.int64)
    raw = np.zeros(dst, dtype=np.int64)
    t = np.zeros(dst, dtype=np.float32)
    for d in range(dst):
        f = np.float32((d + 0.5) * scale - 0.5)
        s = int(np.floor(f))
        f = np.float32(f - np.float32(s))
        raw[d] = s
        if s < 0:
            f, s = np.float32(0), 0
        if s >= src - 1:
            f, s = np.float32(0), src - 1
        idx[d], t[d] = s, f
    return idx, t, raw


def _fix(w):
    """saturate_cast<short>(float * INTER_RESIZE_COEF_SCALE)"""
    return int(np.clip(np.rint(np.float32(w) * np.float32(_COEF_ONE)), -32768, 32767))


def _area_half_u8(src, dh, dw):
    """ResizeAreaFast_Invoker with scale 2 x 2 on uint8: full 2 x 2 cells -> (sum + 2) >> 2, cells cut by the border ->
    saturate_cast<uchar>(float(sum) / count), cells outside -> 0"""
    sh, sw, cn = src.shape
    out = np.zeros((dh, dw, cn), dtype=np.uint8)
    full_w = sw // 2
    for dy in range(dh):
        sy0 = dy * 2
        if sy0 >= sh:
            continue
        row_full = sy0 + 2 <= sh
        for dx in range(dw):
            sx0 = dx * 2
            if row_full and dx < full_w:
                cell = src[sy0:sy0 + 2, sx0:sx0 + 2].astype(np.int64)
                out[dy, dx] = (cell.sum(axis=(0, 1)) + 2) >> 2
            elif sx0 < sw:
                cell = src[sy0:min(sy0 + 2, sh), sx0:min(sx0 + 2, sw)].astype(np.int64)
                cnt = cell.shape[0] * cell.shape[1]
                out[dy, dx] = np.clip(np.rint((cell.sum(axis=(0, 1)).astype(np.float32) / np.float32(cnt))), 0, 255)
    return out


def cv2_resize(img, dsize=None, fx=0.0, fy=0.0, nearest=False):
    """cv2.resize(img, dsize, fx=fx, fy=fy, interpolation=INTER_NEAREST if nearest else INTER_LINEAR) for uint8 (bit
    exact restatement) and float32 (float restatement) arrays of shape (H, W) or (H, W, C); dsize = (width, height)."""
    a = img[:, :, None] if img.ndim == 2 else img
    sh, sw, cn = a.shape
    if dsize is None:
        dw, dh = cv_round(sw * fx), cv_round(sh * fy)
        inv_x, inv_y = float(fx), float(fy)
    else:
        dw, dh = int(dsize[0]), int(dsize[1])
        inv_x, inv_y = dw / sw, dh / sh
    assert dw > 0 and dh > 0
    if (dh, dw) == (sh, sw):
        return img.copy()
    scale_x, scale_y = 1.0 / inv_x, 1.0 / inv_y
    if nearest:
        xs = np.minimum(np.floor(np.arange(dw) * scale_x).astype(np.int64), sw - 1)
        ys = np.minimum(np.floor(np.arange(dh) * scale_y).astype(np.int64), sh - 1)
        out = a[ys][:, xs]
        return out[:, :, 0] if img.ndim == 2 else out
    if a.dtype == np.uint8 and abs(scale_x - 2) < np.finfo(np.float64).eps and abs(scale_y - 2) < np.finfo(np.float64).eps:
        out = _area_half_u8(a, dh, dw)
        return out[:, :, 0] if img.ndim == 2 else out
    xi, xt, _ = _linear_table(dw, sw, scale_x)
    _, _, yraw = _linear_table(dh, sh, scale_y)
    yt = np.zeros(dh, dtype=np.float32)
    for d in range(dh):                                  # the row weights are NOT reset at the border (resize(): yofs loop)
        f = np.float32((d + 0.5) * scale_y - 0.5)
        yt[d] = np.float32(f - np.float32(np.floor(f)))
    y0 = np.clip(yraw, 0, sh - 1)                        # resizeGeneric_Invoker: clip(sy0 + k, 0, ssize.height)
    y1 = np.clip(yraw + 1, 0, sh - 1)
    x1 = np.minimum(xi + 1, sw - 1)
    if a.dtype == np.uint8:
        a0 = np.array([_fix(np.float32(1) - t) for t in xt], dtype=np.int64)
        a1 = np.array([_fix(t) for t in xt], dtype=np.int64)
        b0 = np.array([_fix(np.float32(1) - t) for t in yt], dtype=np.int64)
        b1 = np.array([_fix(t) for t in yt], dtype=np.int64)
        s = a.astype(np.int64)
        rows = s[:, xi] * a0[None, :, None] + s[:, x1] * a1[None, :, None]              # HResizeLinear (int32 in cv2)
        r0, r1 = rows[y0], rows[y1]
        out = (((b0[:, None, None] * (r0 >> 4)) >> 16) + ((b1[:, None, None] * (r1 >> 4)) >> 16) + 2) >> 2
        out = np.clip(out, 0, 255).astype(np.uint8)
    elif a.dtype == np.float32:
        a0, a1 = (np.float32(1) - xt), xt
        b0, b1 = (np.float32(1) - yt), yt
        rows = a[:, xi] * a0[None, :, None] + a[:, x1] * a1[None, :, None]
        out = (rows[y0] * b0[:, None, None] + rows[y1] * b1[:, None, None]).astype(np.float32)
    else:
        raise TypeError(f"cv2_resize restates uint8 and float32 only, got {a.dtype}")
    return out[:, :, 0] if img.ndim == 2 else out


def sliding_eval_rgbX(model, img, modal_x, crop_size, stride_rate, class_num, mean, std, is_flip, multi_scales, device):
    """engine/evaluator.py:432-450"""
    ori_rows, ori_cols, _ = img.shape
    processed_pred = np.zeros((ori_rows, ori_cols, class_num))
    for s in multi_scales:
        img_scale = cv2_resize(img, None, fx=s, fy=s)
        if modal_x.ndim == 2:
            modal_x_scale = cv2_resize(modal_x, None, fx=s, fy=s, nearest=True)
        else:
            modal_x_scale = cv2_resize(modal_x, None, fx=s, fy=s)
        processed_pred += scale_process_rgbX(model, img_scale, modal_x_scale, (ori_rows, ori_cols), crop_size, stride_rate,
                                             class_num, mean, std, is_flip, device)
    return processed_pred.argmax(2), processed_pred
