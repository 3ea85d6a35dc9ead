#!/usr/bin/env python3
"""Phase clocks of the split-operand GEMM k-step loop (library built with -DSIGMA_GEMM_PROF=1; SIGMA_HIP_LIB selects it).

    SIGMA_HIP_LIB=sigma_amd/lib/libsigma_hip_gprof.so python tools/diag/gemm_prof.py [shape ...]
"""
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sigma_amd import _capi, gemm  # noqa: E402
from tools.gemm_bench import SHAPES  # noqa: E402

NAMES = ["wait_loads", "split_store", "barrier1", "load_issue", "frag_mfma", "barrier2", "epilogue", "item_switch"]


def read():
    out = (ctypes.c_uint64 * 16)()
    assert _capi.load().sigma_scan_debug_read(out) == 0
    return list(out)


def main():
    want = sys.argv[1:] or ["enc_s2_in_proj"]
    dev = "cuda"
    for name, M, K, N in SHAPES:
        if name not in want:
            continue
        g = torch.Generator().manual_seed(0)
        x = torch.randn(M, K, generator=g).to(dev)
        w = (0.05 * torch.randn(N, K, generator=g)).to(dev)
        dy = torch.randn(M, N, generator=g).to(dev)
        for label, fn in (("nt", lambda: gemm.gemm_nt(x, w)), ("nn", lambda: gemm.gemm_nn(dy, w)), ("tn", lambda: gemm.gemm_tn(dy, x))):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            read()
            fn()
            torch.cuda.synchronize()
            c = read()
            steps = max(c[15], 1)
            rec = dict(shape=name, form=label, wave_ksteps=c[15], tiles_x_waves=c[13], clocks_per_kstep=round(c[14] / steps, 1))
            for i, n in enumerate(NAMES):
                rec[n] = round(c[i] / steps, 1)
            rec["epilogue_per_tile"] = round(c[6] / max(c[13], 1), 1)
            print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
