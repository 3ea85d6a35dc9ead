"""oracle/evaluator_oracle.py's restatement of cv2.resize against hand-derived known answers, and the device resize of
sigma_amd/engine/evaluator_ops.py (run on CPU tensors here, on the GPU in tests/test_model_gpu.py) against it, bit for bit."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import evaluator_oracle as EO                                    # noqa: E402
from sigma_amd.engine import evaluator_ops as E                              # noqa: E402


def test_destination_size_rounds_half_to_even():
    # cvRound(ssize * f): 2.5 -> 2, 3.5 -> 4, 7.5 -> 8 (cv2.resize(np.zeros((5, 7, 3), np.uint8), None, fx=.5, fy=.5).shape == (2, 4, 3))
    assert EO.cv2_resize(np.zeros((5, 7, 3), np.uint8), None, fx=0.5, fy=0.5).shape == (2, 4, 3)
    assert EO.cv2_resize(np.zeros((6, 10), np.uint8), None, fx=0.75, fy=0.75).shape == (4, 8)       # 4.5 -> 4, 7.5 -> 8
    assert EO.cv2_resize(np.zeros((480, 640, 3), np.uint8), None, fx=1.25, fy=1.25).shape == (600, 800, 3)
    assert EO.cv2_resize(np.zeros((480, 640, 3), np.uint8), None, fx=0.75, fy=0.75).shape == (360, 480, 3)


def test_linear_u8_known_answers():
    # 2x up-sampling of [0, 100]: sample positions -0.25 (clamped), 0.25, 0.75, 1.25 (clamped); weights 512 / 1536 of 2048;
    # vertical pass on the single (clipped) row: ((512 * 3200) >> 16) + ((1536 * 3200) >> 16) + 2 >> 2 = (25 + 75 + 2) >> 2
    out = EO.cv2_resize(np.array([[0, 100]], np.uint8), None, fx=2, fy=2)
    assert out.tolist() == [[0, 25, 75, 100], [0, 25, 75, 100]]
    # 0.75 of [10, 255, 7, 9] (one row): position 1/6 -> weights 1707 / 341; 10 * 1707 + 255 * 341 = 104025, >> 4 = 6501,
    # ((1707 * 6501) >> 16) + ((341 * 6501) >> 16) + 2 >> 2 = (169 + 33 + 2) >> 2 = 51; middle: position 1.5 -> 1024 / 1024 of
    # (255, 7): 268288 >> 4 = 16768 -> ((1707 * 16768) >> 16) + ((341 * 16768) >> 16) + 2 >> 2 = (436 + 87 + 2) >> 2 = 131;
    # last: position 2.8333 (float32 2.8333333) -> weights 341 / 1707 of (7, 9): 17750 >> 4 = 1109 -> (28 + 5 + 2) >> 2 = 8
    out = EO.cv2_resize(np.array([[10, 255, 7, 9]], np.uint8), None, fx=0.75, fy=0.75)
    assert out.tolist() == [[51, 131, 8]]
    # a column: the ROW weights are not reset at the border, the row indices are clipped
    out = EO.cv2_resize(np.array([[0], [200]], np.uint8), None, fx=2, fy=2)
    assert out[:, 0].tolist() == [0, 50, 150, 200]


def test_half_scale_is_the_area_mean():
    img = np.array([[1, 2, 9, 9], [2, 2, 9, 10]], np.uint8)
    assert EO.cv2_resize(img, None, fx=0.5, fy=0.5).tolist() == [[2, 9]]           # (7 + 2) >> 2 = 2, (37 + 2) >> 2 = 9
    odd = np.array([[10, 20, 31], [10, 20, 32], [1, 2, 3]], np.uint8)               # 3 x 3 -> 2 x 2 (1.5 -> 2): border cells by count
    assert EO.cv2_resize(odd, None, fx=0.5, fy=0.5).tolist() == [[15, 32], [2, 3]]  # (60 + 2) >> 2; round(63 / 2) = 32 (half to even); 3 / 2 -> 2; 3


def test_nearest_known_answers():
    img = np.arange(12, dtype=np.uint8).reshape(3, 4)
    assert EO.cv2_resize(img, None, fx=0.75, fy=0.75, nearest=True).tolist() == [[0, 1, 2], [4, 5, 6]]     # floor(d / .75) = 0, 1, 2
    assert EO.cv2_resize(img, None, fx=1.25, fy=1.25, nearest=True).shape == (4, 5)
    assert EO.cv2_resize(img, None, fx=1.25, fy=1.25, nearest=True)[0].tolist() == [0, 0, 1, 2, 3]        # floor(d * .8)
    assert EO.cv2_resize(img, None, fx=1, fy=1, nearest=True).tolist() == img.tolist()


def test_constants_and_identity_survive():
    rng = np.random.default_rng(0)
    for c in (0, 1, 127, 254, 255):
        for f in (0.75, 1.25, 1.5, 1.75, 0.5):
            out = EO.cv2_resize(np.full((37, 53, 3), c, np.uint8), None, fx=f, fy=f)
            assert out.min() == c and out.max() == c, (c, f)
    img = rng.integers(0, 256, (9, 11, 3), dtype=np.uint8)
    assert np.array_equal(EO.cv2_resize(img, None, fx=1, fy=1), img)
    # the fixed-point result stays within one grey level of exact bilinear interpolation on cv2's sampling grid
    img = rng.integers(0, 256, (48, 64, 3), dtype=np.uint8)
    ref = torch.nn.functional.interpolate(torch.from_numpy(img).permute(2, 0, 1)[None].double(), size=(60, 80), mode="bilinear",
                                          align_corners=False)[0].permute(1, 2, 0).numpy()
    assert np.abs(EO.cv2_resize(img, None, fx=1.25, fy=1.25).astype(np.float64) - ref).max() <= 1.0


SCALES = [0.5, 0.75, 1, 1.25, 1.5, 1.75, 2, 0.6, 1.1]


@pytest.mark.parametrize("shape", [(48, 64, 3), (37, 53, 3), (5, 7, 1), (1, 9, 3), (23, 2, 3), (30, 40)])
def test_device_resize_is_the_oracle_bit_for_bit(shape):
    rng = np.random.default_rng(sum(shape))
    img = rng.integers(0, 256, shape, dtype=np.uint8)
    for f in SCALES:
        if min(round(shape[0] * f), round(shape[1] * f)) < 1:
            continue
        for nearest in (False, True):
            want = EO.cv2_resize(img, None, fx=f, fy=f, nearest=nearest)
            got = E.resize_like_cv2(img, f, nearest, torch.device("cpu"))
            assert got.dtype == np.uint8 and got.shape == want.shape, (f, nearest)
            assert np.array_equal(got, want), (shape, f, nearest, np.argwhere(got != want)[:4])
    depth = rng.integers(0, 60000, shape[:2]).astype(np.uint16)                 # 2-D modal_x of any dtype: nearest only
    assert np.array_equal(E.resize_like_cv2(depth, 1.25, True, torch.device("cpu")), EO.cv2_resize(depth, None, fx=1.25, fy=1.25, nearest=True))
    with pytest.raises(TypeError):
        E.resize_like_cv2(depth, 1.25, False, torch.device("cpu"))
    fl = rng.standard_normal(shape).astype(np.float32)
    want = EO.cv2_resize(fl, None, fx=1.25, fy=1.25)
    assert np.allclose(E.resize_like_cv2(fl, 1.25, False, torch.device("cpu")), want, rtol=1e-6, atol=1e-6)


def test_window_grid_resolves_negative_starts_like_the_reference_slicing():
    """engine/evaluator.py:472-478 with crop (480, 640) on the 600 x 800 image of scale 1.25: s_y = 600 - 640 = -40, and
    img_pad[-40:600] is the last 40 rows"""
    wins = E.window_grid(600, 800, (480, 640), 2 / 3)
    assert wins == [(560, 600, 0, 480), (560, 600, 320, 800), (560, 600, 0, 480), (560, 600, 320, 800)]
    img = np.arange(600 * 800).reshape(600, 800)
    assert np.array_equal(img[-40:600, 0:480], img[560:600, 0:480])
    # square crops (MFNet-style evaluation of larger images): the plain grid
    assert E.window_grid(600, 800, (480, 480), 2 / 3) == [(0, 480, 0, 480), (0, 480, 320, 800), (120, 600, 0, 480), (120, 600, 320, 800)]
