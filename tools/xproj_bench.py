#!/usr/bin/env python3
"""The six stacked GEMMs of the SS2D core (x_proj / dt_proj, forward and backward) on the split-operand kernels
(sigma_amd.gemm.bgemm_*) against the vendor fp32 batched GEMM formulation, per encoder stage of sigma_small at batch 8
(16 images per pass).   python tools/xproj_bench.py [--iters 20]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sigma_amd import gemm  # noqa: E402
from tools.scan_bench import time_call  # noqa: E402

STAGES = [("enc_s0", 16, 192, 38, 6, 19200), ("enc_s1", 16, 384, 44, 12, 4800), ("enc_s2", 16, 768, 56, 24, 1200), ("enc_s3", 16, 1536, 80, 48, 300)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    dev = "cuda"
    from sigma_amd.tuning import enable_tuned_gemms
    enable_tuned_gemms()
    for name, B, d, c, R, L in STAGES:
        g = torch.Generator().manual_seed(0)
        r = lambda *s: torch.randn(*s, generator=g).to(dev)
        xs, Wst, dtw = r(B, 2, d, L), r(2, 2 * c, d) * 0.05, r(4, d, R) * 0.2
        p4, dd, dp4, du = r(B, 4, c, L), r(B, 4, d, L), r(B, 4, c, L), r(B, 4, d, L)
        delta, dxs = torch.empty(B, 4, d, L, device=dev), torch.empty(B, 2, d, L, device=dev)
        WstT, dtwT = Wst.transpose(1, 2).contiguous(), dtw.transpose(1, 2).contiguous()
        dW, dWd = torch.zeros(2, 2 * c, d, device=dev), torch.zeros(4, d, R, device=dev)
        du3 = du.view(2 * B, 2, d, L)
        rec = dict(stage=name, dims=[B, d, c, R, L])
        own = {
            "x_proj": lambda: gemm.bgemm_nn(Wst, xs.view(2 * B, d, L), p4.view(2 * B, 2 * c, L)),
            "x_dgrad+du": lambda: gemm.bgemm_nn(WstT, dp4.view(2 * B, 2 * c, L), dxs.view(2 * B, d, L), residual=du3[:, 0], residual2=du3[:, 1]),
            "x_wgrad": lambda: gemm.bgemm_nt_sum(dp4.view(2 * B, 2 * c, L), xs.view(2 * B, d, L), dW, accumulate=False),
        }
        if R % 4 == 0:
            own.update({
                "dt_proj": lambda: gemm.bgemm_nn(dtw, p4.view(4 * B, c, L)[:, :R], delta.view(4 * B, d, L)),
                "dt_dgrad": lambda: gemm.bgemm_nn(dtwT, dd.view(4 * B, d, L), dp4.view(4 * B, c, L)[:, :R]),
                "dt_wgrad": lambda: gemm.bgemm_nt_sum(dd.view(4 * B, d, L), p4.view(4 * B, c, L)[:, :R], dWd),
            })
        from sigma_amd.ss2d_fused import _pair_sum_add

        def x_dgrad_vendor():
            o = torch.matmul(WstT.unsqueeze(0), dp4.view(B, 2, 2 * c, L))
            _pair_sum_add(du, o, B * 2, d * L)
            return o

        def dt_dgrad_vendor():
            dp4[:, :, :R] = torch.matmul(dtwT.unsqueeze(0), dd)

        vendor = {
            "x_proj": lambda: torch.matmul(Wst.unsqueeze(0), xs),
            "x_dgrad+du": x_dgrad_vendor,
            "x_wgrad": lambda: torch.matmul(dp4.view(B, 2, 2 * c, L), xs.transpose(-1, -2)).sum(0),
            "dt_proj": lambda: torch.matmul(dtw.unsqueeze(0), p4[:, :, :R]),
            "dt_dgrad": dt_dgrad_vendor,
            "dt_wgrad": lambda: torch.matmul(dd, p4[:, :, :R].transpose(-1, -2)).sum(0),
        }
        for k in vendor:
            rec[k + "_vendor_us"] = round(time_call(vendor[k], a.iters) * 1e6, 1)
            if k in own:
                rec[k + "_own_us"] = round(time_call(own[k], a.iters) * 1e6, 1)
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
