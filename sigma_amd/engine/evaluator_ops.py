"""The evaluator's model call: ``Evaluator.val_func_process_rgbX``
(engine/evaluator.py:501-522) runs the network once on the image pair and, with ``is_flip``, a second time on the
horizontally flipped pair, then adds the un-flipped second score.  Both passes have identical shapes, so here they are
ONE batch-2 forward (the Siamese encoder already runs RGB and X as one batch: four images per launch instead of two
launches of two), which halves the launch count of an evaluation that is launch-bound at batch 1.

Drop-in: ``evaluator.val_func_process_rgbX = types.MethodType(val_func_process_rgbX, evaluator)`` -- same arguments,
same return value (exp of the summed log-scores of image 0, shape (classes, H, W)).
"""
from __future__ import annotations

import numpy as np
import torch


def flip_pair_scores(model, rgb: torch.Tensor, modal_x: torch.Tensor, is_flip: bool) -> torch.Tensor:
    """score[0] (+ flipped score un-flipped) for a batch-1 pair, computed in one forward when ``is_flip``."""
    if not is_flip:
        return model(rgb, modal_x)[0]
    both = model(torch.cat([rgb, rgb.flip(-1)], dim=0), torch.cat([modal_x, modal_x.flip(-1)], dim=0))
    return both[0] + both[1].flip(-1)


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
