#!/usr/bin/env python3
"""Phase breakdown of the row-lane kernels from a -DSIGMA_RL_PROF=1 build (SIGMA_HIP_LIB=.../libsigma_hip_rlprof.so):
cycles per phase summed over the waves / number of waves.   python tools/rowlane_prof.py [shape ...]"""
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sigma_amd import _capi  # noqa: E402
from sigma_amd import selective_scan_cuda_core as core  # noqa: E402
from tools.scan_bench import SHAPES, make  # noqa: E402

PH = ["prologue", "barrier1", "lds_reads", "scalar_wait", "state_loop", "exch_write", "barrier2", "epilogue",
      "p_dma_wait", "p_raw_reads", "p_requests", "e_exch_reads"]      # 8-11: parts of the backward's prologue / epilogue (0 = the rest)


def read():
    buf = (ctypes.c_uint64 * 16)()
    _capi.check(_capi.load().sigma_scan_debug_read(ctypes.byref(buf)), "debug_read")
    return list(buf)


def main():
    for kv in os.environ.get("OPTS", "").split(","):
        if kv:
            k, v = kv.split("=")
            _capi.set_option(k, int(v))
    for name in sys.argv[1:] or ["enc_s2_b16"]:
        u, delta, A, Bm, Cm, D, bias, dout = make(SHAPES[name])
        out, x = core.fwd_ext(u, delta, A, Bm, Cm, D, bias, True, ckpt_pitch=16)
        torch.cuda.synchronize()
        read()
        core.fwd_ext(u, delta, A, Bm, Cm, D, bias, True, ckpt_pitch=16)
        r = read()
        w = max(r[15], 1)
        print(json.dumps({"shape": name, "kernel": "fwd", "waves": r[15], "total_per_wave": sum(r[:12]) / w,
                          **{PH[i]: round(r[i] / w) for i in range(12)}}))
        core.bwd_ext(u, delta, A, Bm, Cm, D, bias, dout, x, True, ckpt_pitch=16)
        r = read()
        w = max(r[15], 1)
        print(json.dumps({"shape": name, "kernel": "bwd", "waves": r[15], "total_per_wave": sum(r[:12]) / w,
                          **{PH[i]: round(r[i] / w) for i in range(12)}}))


if __name__ == "__main__":
    main()
