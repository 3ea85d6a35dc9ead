#!/usr/bin/env python3
"""Evaluator model call with flip (engine/evaluator.py:501-522): two batch-1 passes vs one batch-2 pass
(sigma_amd/engine/evaluator_ops.py), sigma_small 480x640, eval mode."""
import json
import os
import sys
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sigma_amd.engine.evaluator_ops import flip_pair_scores  # noqa: E402
from tests.model_utils import build_model, fill  # noqa: E402


def main():
    model = build_model("sigma_small", 40, 480, 640).cuda().eval()
    rgb, x, _ = fill.make_inputs(1, 480, 640, 40, seed=3)
    rgb, x = rgb.cuda(), x.cuda()

    def two():
        s = model(rgb, x)[0]
        s = s + model(rgb.flip(-1), x.flip(-1))[0].flip(-1)
        return s

    def one():
        return flip_pair_scores(model, rgb, x, True)

    out = {}
    with torch.no_grad():
        for name, fn in (("two_passes", two), ("one_batch2_pass", one), ("two_passes", two), ("one_batch2_pass", one)):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10):
                fn()
            e.record()
            torch.cuda.synchronize()
            out[name] = round(s.elapsed_time(e) / 10, 2)
        err = float((two() - one()).abs().max() / two().abs().max())
    print(json.dumps(dict(workload="sigma_small 480x640 eval, flip test-time augmentation, batch 1", ms=out, rel_diff=err)))


if __name__ == "__main__":
    main()
