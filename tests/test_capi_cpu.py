"""CPU-side checks of the C-ABI library: it loads, exports every symbol that
include/sigma_scan.h declares, and its host-side validation behaves like the reference's
TORCH_CHECKs -- all without touching a GPU (no compute entry point is reached)."""
import ctypes
import os
import re

import pytest

from sigma_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_abi_version():
    lib = _capi.load()
    assert lib.sigma_scan_abi_version() == _capi.SIGMA_SCAN_ABI_VERSION


def test_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "sigma_scan.h")).read()
    declared = set(re.findall(r"^\s*(?:const\s+char\s*\*\s*|int\s+|int64_t\s+)(sigma_\w+)\s*\(", header, flags=re.M))
    assert declared == set(_capi.EXPORTED_SYMBOLS), declared ^ set(_capi.EXPORTED_SYMBOLS)
    lib = _capi.load()
    for name in declared:
        assert getattr(lib, name) is not None


def test_struct_layout_matches_header():
    # 8 int32 + 9 pointers + 14 int64 ; bwd adds 9 pointers + 15 int64
    assert ctypes.sizeof(_capi.FwdParams) == 8 * 4 + 9 * 8 + 14 * 8
    assert ctypes.sizeof(_capi.BwdParams) == ctypes.sizeof(_capi.FwdParams) + 9 * 8 + 15 * 8


def _params(**kw):
    p = _capi.FwdParams()
    p.batch, p.dim, p.seqlen, p.dstate, p.n_groups = 1, 8, 100, 4, 2
    p.n_chunks = 1
    p.io_dtype = 0
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def test_host_validation_without_gpu():
    lib = _capi.load()
    assert lib.sigma_selective_scan_fwd(None, None) == 1                       # NULL params
    assert lib.sigma_selective_scan_fwd(ctypes.byref(_params(io_dtype=7)), None) == 3
    assert lib.sigma_selective_scan_fwd(ctypes.byref(_params(n_groups=3)), None) == 2   # dim % groups
    assert "dividable" in _capi.last_error()
    assert lib.sigma_selective_scan_fwd(ctypes.byref(_params(dstate=257)), None) == 2
    assert lib.sigma_selective_scan_fwd(ctypes.byref(_params(n_chunks=2)), None) == 2
    assert lib.sigma_selective_scan_fwd(ctypes.byref(_params()), None) == 1             # NULL tensors
    # empty problems are a no-op success (nothing is launched)
    assert lib.sigma_selective_scan_fwd(ctypes.byref(_params(seqlen=0, n_chunks=0)), None) == 0
    assert lib.sigma_selective_scan_fwd(ctypes.byref(_params(batch=0)), None) == 0


def test_options_and_launch_plan():
    lib = _capi.load()
    with pytest.raises(RuntimeError):
        _capi.set_option("fwd_items", 5)
    with pytest.raises(RuntimeError):
        _capi.set_option("no_such_option", 1)
    plan = (ctypes.c_int32 * 4)()
    # headline shape (1, 768, 19200), N=16, G=4 -> rows of one workgroup share a group
    p = _params(batch=1, dim=768, seqlen=19200, dstate=16, n_groups=4, n_chunks=10)
    assert lib.sigma_scan_fwd_plan(ctypes.byref(p), ctypes.byref(plan)) == 0
    items, waves, grid, lds = list(plan)
    assert items in (4, 8, 16) and waves in (1, 2, 4, 8, 16)
    assert (768 // 4) % waves == 0 and grid == 768 // waves and lds <= 160 * 1024
    _capi.set_option("fwd_waves", 16)
    try:
        assert lib.sigma_scan_fwd_plan(ctypes.byref(p), ctypes.byref(plan)) == 0
        assert plan[1] == 16 and plan[2] == 48
    finally:
        _capi.set_option("fwd_waves", 0)
    # reference unit-test shape: 24 rows, 2 groups -> 12 rows per group -> 4 rows per workgroup
    p = _params(batch=2, dim=24, seqlen=372, dstate=8, n_groups=2, n_chunks=1)
    assert lib.sigma_scan_fwd_plan(ctypes.byref(p), ctypes.byref(plan)) == 0
    assert plan[1] == 4 and plan[2] == 12
    bp = _capi.BwdParams()
    bp.fwd = p
    assert lib.sigma_scan_bwd_plan(ctypes.byref(bp), ctypes.byref(plan)) == 0
    assert plan[0] in (4, 8) and plan[3] <= 160 * 1024
    # 12 rows per group, 4 per workgroup -> 3 partial slabs of (B, G, N, L) for each of dB, dC
    assert lib.sigma_scan_bwd_workspace_bytes(ctypes.byref(bp)) == 2 * 3 * 2 * 2 * 8 * 372 * 4


def test_operator_module_raises_without_gpu_tensors():
    import torch
    from sigma_amd import selective_scan_cuda_core as core
    u = torch.randn(1, 8, 16)
    A = -torch.rand(8, 4)
    Bm = torch.randn(1, 1, 4, 16)
    with pytest.raises(RuntimeError, match="GPU tensor"):
        core.fwd(u, u, A, Bm, Bm, None, None, False, 1)
