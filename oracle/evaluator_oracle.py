"""TEST INFRASTRUCTURE ONLY: literal restatement of the reference's sliding-window evaluation for one scale
(engine/evaluator.py:452-499 scale_process_rgbX, :501-522 val_func_process_rgbX, :525-560 process_image_rgbX,
utils/transforms.py:61-75 pad_image_to_shape, :182-187 normalize) -- one window at a time, numpy slicing / normalising /
padding per window, one batch-1 forward per window (two with is_flip), scores accumulated window by window.
cv2 is not installed in this image: copyMakeBorder(BORDER_CONSTANT, 0) is np.pad, and the final cv2.resize of the
scores (INTER_LINEAR) is torch's bilinear resize with the same sampling grid (identity when the size is unchanged).
Used by tests/test_model_gpu.py to check sigma_amd/engine/evaluator_ops.py; parity unpinned against cv2 itself."""
import numpy as np
import torch
import torch.nn.functional as F


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
    if tuple(score.shape[1:]) != (ori_shape[0], ori_shape[1]):
        score = F.interpolate(score[None], size=(ori_shape[0], ori_shape[1]), mode="bilinear", align_corners=False)[0]
    return score.permute(1, 2, 0).cpu().numpy()
