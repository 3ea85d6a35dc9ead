#!/usr/bin/env python3
"""Split-operand MFMA GEMMs (csrc/gemm_split.hip) against the vendor fp32 GEMM on the projection shapes of the
sigma_small training step at batch 8 (encoder: 16 images per pass).

    python tools/gemm_bench.py [--iters 20] [--out file.jsonl]

Per shape: fwd y = x W^T (nt), dgrad dx = dy W (nn), wgrad dW = dy^T x (tn); microseconds, fp32-equivalent TFLOP/s
(2 M N K / t), bytes moved once / t as a fraction of 6.3 TB/s, and the error against fp64 of both paths.
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sigma_amd import gemm  # noqa: E402
from tools.scan_bench import time_call  # noqa: E402

# (name, tokens M, in K, out N)
SHAPES = [
    ("enc_s0_in_proj", 16 * 19200, 96, 384), ("enc_s0_out_proj", 16 * 19200, 192, 96),
    ("enc_s1_in_proj", 16 * 4800, 192, 768), ("enc_s1_out_proj", 16 * 4800, 384, 192),
    ("enc_s2_in_proj", 16 * 1200, 384, 1536), ("enc_s2_out_proj", 16 * 1200, 768, 384),
    ("enc_s3_in_proj", 16 * 300, 768, 3072), ("enc_s3_out_proj", 16 * 300, 1536, 768),
    ("merge_s0", 16 * 4800, 384, 192), ("merge_s1", 16 * 1200, 768, 384), ("merge_s2", 16 * 300, 1536, 768),
    ("dec_s2_in_proj", 8 * 1200, 384, 1536), ("dec_s0_in_proj", 8 * 19200, 96, 384), ("dec_final_linear", 8 * 76800, 96, 96),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--out", default="")
    ap.add_argument("--shapes", default="")
    ap.add_argument("--only", default="", help="comma separated column labels (e.g. nt_split3,tn_split3); default all")
    a = ap.parse_args()
    rows = []
    dev = "cuda"
    from sigma_amd.tuning import enable_tuned_gemms, tuned_gemms_active
    enable_tuned_gemms()                      # the fp32 column = the tuned vendor GEMMs of the training step
    for name, M, K, N in SHAPES:
        if a.shapes and name not in a.shapes.split(","):
            continue
        g = torch.Generator().manual_seed(0)
        x = torch.randn(M, K, generator=g).to(dev)
        w = (0.05 * torch.randn(N, K, generator=g)).to(dev)
        dy = torch.randn(M, N, generator=g).to(dev)
        flops = 2.0 * M * N * K
        byts = 4.0 * (M * K + N * K + M * N)
        rec = dict(shape=name, M=M, K=K, N=N)
        ref = None
        if M * N <= 40e6:
            ref = x.double() @ w.double().t()
        for label, fn in (("nt_split3", lambda: gemm.gemm_nt(x, w)), ("nt_split6", lambda: gemm.gemm_nt(x, w, pieces=3)),
                          ("nn_split6", lambda: gemm.gemm_nn(dy, w, pieces=3)), ("tn_split6", lambda: gemm.gemm_tn(dy, x, pieces=3)),
                          ("nt_fp32", lambda: torch.mm(x, w.t())),
                          ("nn_split3", lambda: gemm.gemm_nn(dy, w)), ("nn_fp32", lambda: torch.mm(dy, w)),
                          ("tn_split3", lambda: gemm.gemm_tn(dy, x)), ("tn_fp32", lambda: torch.mm(dy.t(), x))):
            if a.only and label not in a.only.split(","):
                continue
            t = time_call(fn, a.iters)
            rec[label + "_us"] = round(t * 1e6, 1)
            rec[label + "_TF"] = round(flops / t / 1e12, 1)
            rec[label + "_hbm_frac"] = round(byts / t / 6.3e12, 3)
        if ref is not None and not a.only:
            rec["nt_split3_err"] = float((gemm.gemm_nt(x, w).double() - ref).abs().max() / ref.abs().max())
            rec["nt_split6_err"] = float((gemm.gemm_nt(x, w, pieces=3).double() - ref).abs().max() / ref.abs().max())
            rec["nt_fp32_err"] = float((torch.mm(x, w.t()).double() - ref).abs().max() / ref.abs().max())
        rows.append(rec)
        print(json.dumps(rec), flush=True)
        del x, w, dy, ref
        torch.cuda.empty_cache()
    print(json.dumps(dict(tuned_gemms=tuned_gemms_active())), flush=True)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)) or ".", exist_ok=True)
        with open(a.out, "w") as f:
            for r in rows:
                f.write(json.dumps(r) + "\n")


if __name__ == "__main__":
    main()
