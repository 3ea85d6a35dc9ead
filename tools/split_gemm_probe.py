#!/usr/bin/env python3
"""Probe: fp32 GEMM vs split-operand bf16 GEMM (a = a_hi + a_lo, b = b_hi + b_lo in bf16; one K-concatenated GEMM
[a_hi | a_hi | a_lo] x [b_hi ; b_lo ; b_hi] with fp32 accumulate/output) on the linear shapes of sigma_small.

    python tools/split_gemm_probe.py
Prints per shape: time of torch.mm fp32, of the bf16 GEMM on pre-split operands (K' = 3K and the 6-term K' = 6K
variant), of the splitting itself (torch ops), and the max / rms error of each against an fp64 product.
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.scan_bench import time_call  # noqa: E402

SHAPES = [  # (M, K, N)   x (M,K) @ w^T (K,N)
    (19200, 384, 1536), (19200, 768, 384), (19200, 1536, 384), (153600, 96, 384), (153600, 192, 96),
    (38400, 192, 768), (38400, 384, 192), (4800, 768, 3072), (4800, 1536, 768),
]


def split2(t):
    hi = t.to(torch.bfloat16)
    lo = (t - hi.float()).to(torch.bfloat16)
    return hi, lo


def split3(t):
    hi = t.to(torch.bfloat16)
    r = t - hi.float()
    mid = r.to(torch.bfloat16)
    lo = (r - mid.float()).to(torch.bfloat16)
    return hi, mid, lo


def main():
    dev = "cuda"
    torch.manual_seed(0)
    for M, K, N in SHAPES:
        a = torch.randn(M, K, device=dev)
        b = (torch.randn(K, N, device=dev) / K ** 0.5)
        ref = (a.double() @ b.double())
        scale = float(ref.abs().max())
        rec = dict(M=M, K=K, N=N, gflop=round(2e-9 * M * K * N, 2))
        t = time_call(lambda: torch.mm(a, b), 10)
        c = torch.mm(a, b)
        rec["fp32_us"] = round(t * 1e6, 1)
        rec["fp32_tflops"] = round(2e-12 * M * K * N / t, 1)
        rec["fp32_err_max"] = float((c.double() - ref).abs().max()) / scale
        # plain bf16
        ah, al = split2(a)
        bh, bl = split2(b)
        t = time_call(lambda: torch.mm(ah, bh, out_dtype=torch.float32), 10)
        c = torch.mm(ah, bh, out_dtype=torch.float32)
        rec["bf16_us"] = round(t * 1e6, 1)
        rec["bf16_err_max"] = float((c.double() - ref).abs().max()) / scale
        # 3-term split, K' = 3K
        A3 = torch.cat([ah, ah, al], dim=1).contiguous()
        B3 = torch.cat([bh, bl, bh], dim=0).contiguous()
        t = time_call(lambda: torch.mm(A3, B3, out_dtype=torch.float32), 10)
        c = torch.mm(A3, B3, out_dtype=torch.float32)
        rec["split3_us"] = round(t * 1e6, 1)
        rec["split3_err_max"] = float((c.double() - ref).abs().max()) / scale
        rec["split3_err_rms"] = float((c.double() - ref).pow(2).mean().sqrt()) / float(ref.pow(2).mean().sqrt())
        # 6-term split (3-way operands), K' = 6K: hi*hi + hi*mid + mid*hi + hi*lo + mid*mid + lo*hi
        a3 = split3(a)
        b3 = split3(b)
        A6 = torch.cat([a3[0], a3[0], a3[1], a3[0], a3[1], a3[2]], dim=1).contiguous()
        B6 = torch.cat([b3[0], b3[1], b3[0], b3[2], b3[1], b3[0]], dim=0).contiguous()
        t = time_call(lambda: torch.mm(A6, B6, out_dtype=torch.float32), 10)
        c = torch.mm(A6, B6, out_dtype=torch.float32)
        rec["split6_us"] = round(t * 1e6, 1)
        rec["split6_err_max"] = float((c.double() - ref).abs().max()) / scale
        # cost of forming A3 with torch ops (a fused kernel would be one pass: read 4 B, write 6 B per element)
        t = time_call(lambda: torch.cat([a.to(torch.bfloat16), a.to(torch.bfloat16), (a - a.to(torch.bfloat16).float()).to(torch.bfloat16)], dim=1), 10)
        rec["split_torch_ops_us"] = round(t * 1e6, 1)
        rec["split_fused_estimate_us"] = round(M * K * 10 / 4.0e12 * 1e6, 1)
        print(json.dumps(rec), flush=True)
        del a, b, ref, A3, B3, A6, B6


if __name__ == "__main__":
    main()
