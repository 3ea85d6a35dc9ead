"""ctypes front-end of oracle/scan_oracle.c -- TEST INFRASTRUCTURE ONLY.

The oracle is the checker for the HIP selective-scan kernels.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module; the product package ``sigma_amd`` never does (and fails loudly without
its HIP library instead of falling back to anything here).

Restates (reference paths relative to /root/reference):
  * ``selective_scan_ref``  models/encoders/selective_scan/selective_scan/selective_scan_interface.py:86-131
  * the adjoint computed by ``selective_scan_bwd_kernel``
    models/encoders/selective_scan/csrc/selective_scan/selective_scan_bwd_kernel.cuh:141-273

Pinned by tests/test_oracle_scan.py against tests/golden/scan_*.npz, which were
produced by the reference's own ``selective_scan_ref`` + autograd in the build
container (tests/golden/make_golden_scan.py).
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Optional, Tuple

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libscan_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    """Compile the C oracle with gcc (idempotent)."""
    src = os.path.join(_HERE, "scan_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
    return _LIB_PATH


def _load():
    global _lib
    if _lib is None:
        build()
        lib = ctypes.CDLL(_LIB_PATH)
        fp = ctypes.c_void_p
        ci = ctypes.c_int
        lib.scan_oracle_fwd.argtypes = [fp] * 7 + [ci] * 7 + [fp, fp]
        lib.scan_oracle_fwd.restype = ci
        lib.scan_oracle_bwd.argtypes = [fp] * 8 + [ci] * 7 + [fp] * 7
        lib.scan_oracle_bwd.restype = ci
        lib.scan_oracle_num_threads.restype = ci
        _lib = lib
    return _lib


def num_threads() -> int:
    return int(_load().scan_oracle_num_threads())


def _f32(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    if t is None:
        return None
    return t.detach().to(device="cpu", dtype=torch.float32).contiguous()


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _canon(u, delta, A, B, C, D, delta_bias):
    """fp32 upcast + (B,G,N,L) view exactly as selective_scan_ref does (:96-118)."""
    u32, d32, A32 = _f32(u), _f32(delta), _f32(A)
    B32, C32 = _f32(B), _f32(C)
    if B32.dim() == 3:
        B32 = B32.unsqueeze(1)
    if C32.dim() == 3:
        C32 = C32.unsqueeze(1)
    Bsz, Dm, L = u32.shape
    N = A32.shape[1]
    G = B32.shape[1]
    assert C32.shape[1] == G, "oracle expects B and C with the same group count"
    assert A32.shape[0] == Dm and Dm % G == 0
    assert B32.shape == (Bsz, G, N, L) and C32.shape == (Bsz, G, N, L)
    return u32, d32, A32, B32.contiguous(), C32.contiguous(), _f32(D), _f32(delta_bias), (Bsz, Dm, L, N, G)


def selective_scan_oracle(u, delta, A, B, C, D=None, delta_bias=None, delta_softplus=False,
                          acc64: bool = False, return_last_state: bool = False):
    """Forward oracle; same signature/semantics as the reference's selective_scan_ref.

    Output is cast back to ``u.dtype`` (selective_scan_interface.py:129).
    """
    lib = _load()
    u32, d32, A32, B32, C32, D32, b32, (Bsz, Dm, L, N, G) = _canon(u, delta, A, B, C, D, delta_bias)
    out = torch.empty_like(u32)
    last = torch.empty(Bsz, Dm, N, dtype=torch.float32)
    rc = lib.scan_oracle_fwd(_ptr(u32), _ptr(d32), _ptr(A32), _ptr(B32), _ptr(C32), _ptr(D32), _ptr(b32),
                             int(bool(delta_softplus)), Bsz, Dm, L, N, G, int(bool(acc64)),
                             _ptr(out), _ptr(last))
    if rc != 0:
        raise RuntimeError(f"scan_oracle_fwd failed with code {rc}")
    out = out.to(u.dtype)
    return (out, last) if return_last_state else out


def selective_scan_oracle_bwd(u, delta, A, B, C, D, delta_bias, dout, delta_softplus=False
                              ) -> Tuple[torch.Tensor, ...]:
    """Backward oracle: (du, ddelta, dA, dB, dC, dD, ddelta_bias), all float32.

    dB/dC come back with the rank B/C were given in (3-D inputs -> 3-D grads).
    """
    lib = _load()
    squeeze_B = B.dim() == 3
    squeeze_C = C.dim() == 3
    u32, d32, A32, B32, C32, D32, b32, (Bsz, Dm, L, N, G) = _canon(u, delta, A, B, C, D, delta_bias)
    g32 = _f32(dout)
    du = torch.empty_like(u32)
    dd = torch.empty_like(u32)
    dA = torch.empty_like(A32)
    dB = torch.empty_like(B32)
    dC = torch.empty_like(C32)
    dD = torch.empty(Dm, dtype=torch.float32) if D is not None else None
    db = torch.empty(Dm, dtype=torch.float32) if delta_bias is not None else None
    rc = lib.scan_oracle_bwd(_ptr(u32), _ptr(d32), _ptr(A32), _ptr(B32), _ptr(C32), _ptr(D32), _ptr(b32),
                             _ptr(g32), int(bool(delta_softplus)), Bsz, Dm, L, N, G, 1,
                             _ptr(du), _ptr(dd), _ptr(dA), _ptr(dB), _ptr(dC), _ptr(dD), _ptr(db))
    if rc != 0:
        raise RuntimeError(f"scan_oracle_bwd failed with code {rc}")
    if squeeze_B:
        dB = dB.squeeze(1)
    if squeeze_C:
        dC = dC.squeeze(1)
    return du, dd, dA, dB, dC, dD, db


def algorithmic_bytes_fwd(Bsz, KD, L, N, G, itemsize=4, with_checkpoint=False) -> int:
    """u, delta read once; out written once; B, C once per group."""
    b = itemsize * 3 * Bsz * KD * L + itemsize * 2 * Bsz * G * N * L + 4 * (KD * N + 2 * KD)
    if with_checkpoint:
        b += 4 * Bsz * KD * ((L + 2047) // 2048) * 2 * N
    return int(b)


def algorithmic_bytes_bwd(Bsz, KD, L, N, G, itemsize=4) -> int:
    """reads u, delta, dout, B, C; writes du, ddelta, dB, dC."""
    return int(itemsize * 5 * Bsz * KD * L + itemsize * 4 * Bsz * G * N * L)
