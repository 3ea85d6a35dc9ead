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
    ops = open(os.path.join(ROOT, "include", "sigma_ops.h")).read()
    declared_ops = set(re.findall(r"^\s*int\s+(sigma_\w+)\s*\(", ops, flags=re.M))
    assert declared_ops == set(_capi.OPS_SYMBOLS), declared_ops ^ set(_capi.OPS_SYMBOLS)
    for name in declared_ops:
        assert getattr(lib, name) is not None
    gemm = open(os.path.join(ROOT, "include", "sigma_gemm.h")).read()
    declared_gemm = set(re.findall(r"^\s*(?:int|int64_t)\s+(sigma_\w+)\s*\(", gemm, flags=re.M))
    assert declared_gemm == set(_capi.GEMM_SYMBOLS) | set(_capi.GEMM_AUX_SYMBOLS), declared_gemm ^ set(_capi.GEMM_SYMBOLS)
    for name in declared_gemm:
        assert getattr(lib, name) is not None


def test_struct_layout_matches_header(tmp_path):
    """Compile include/sigma_scan.h with gcc and compare sizeof/offsetof of every field with the
    ctypes mirror in sigma_amd/_capi.py."""
    import subprocess
    lines = []
    structs = (("sigma_scan_fwd_params", _capi.FwdParams), ("sigma_scan_bwd_params", _capi.BwdParams),
               ("sigma_dwconv_params", _capi.DwConvParams), ("sigma_merge_params", _capi.MergeParams),
               ("sigma_layernorm_params", _capi.LayerNormParams), ("sigma_transpose_params", _capi.TransposeParams),
               ("sigma_gemm_params", _capi.GemmParams), ("sigma_gate_bwd_params", _capi.GateBwdParams))
    for cname, cls in structs:
        lines.append(f'printf("%s %zu\\n", "{cname}", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'printf("%s.%s %zu\\n", "{cname}", "{fname}", offsetof({cname}, {fname}));')
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "sigma_scan.h"\n#include "sigma_ops.h"\n#include "sigma_gemm.h"\nint main(void){' + "".join(lines) + "return 0;}")
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = dict(l.split() for l in subprocess.check_output([str(exe)], text=True).splitlines())
    for cname, cls in structs:
        assert int(got[cname]) == ctypes.sizeof(cls)
        for fname, _ in cls._fields_:
            assert int(got[f"{cname}.{fname}"]) == getattr(cls, fname).offset, fname


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
        _capi.set_option("fwd_items", 8)
    with pytest.raises(RuntimeError):
        _capi.set_option("no_such_option", 1)
    with pytest.raises(RuntimeError):
        _capi.set_option("rl_chain_timeouts", 0)          # read-only counter of the chained walk
    assert _capi.get_option("no_such_option") == -1
    plan = (ctypes.c_int32 * 6)()
    # headline shape (1, 768, 19200), N=16, G=4: few rows -> the sequence is split inside the workgroup
    p = _params(batch=1, dim=768, seqlen=19200, dstate=16, n_groups=4, n_chunks=10)
    assert lib.sigma_scan_fwd_plan(ctypes.byref(p), ctypes.byref(plan)) == 0
    items, rows, grid, lds, tiles, nb = list(plan)
    assert items in (4, 5, 10, 20) and 1 <= rows <= 16 and 1 <= rows * tiles <= 16 and nb in (1, 2, 4, 8)
    assert (768 // 4) % rows == 0 and grid == 768 // rows and lds <= 160 * 1024
    assert tiles > 1 and grid >= 128          # one image per GPU still fills the chip
    # the same launch on the row-lane kernels (ckpt_pitch 16): 12 row blocks are cut into segments so that no CU holds a
    # workgroup more than the others (round 6: 48 segments = 576 workgroups, three on 64 CUs and two on the rest, ran 170 us;
    # 40-42 segments = 480-504, two at most, 149-152 us) -- and the dominant training launch is not cut at all
    for b, d, L, n, g in ((1, 768, 19200, 16, 4), (2, 768, 19200, 16, 4), (1, 1536, 4800, 16, 4), (2, 3072, 1200, 16, 4)):
        pr = _params(batch=b, dim=d, seqlen=L, dstate=n, n_groups=g, n_chunks=(L + 2047) // 2048)
        pr.ckpt_pitch = 16
        assert lib.sigma_scan_fwd_plan(ctypes.byref(pr), ctypes.byref(plan)) == 0
        assert plan[0] == 16 and plan[5] == -200 and plan[4] > 1 and plan[2] == b * (d // 64) * plan[4]
        assert 384 <= plan[2] <= 512 or plan[2] % 256 == 0, list(plan)
    pr = _params(batch=16, dim=3072, seqlen=1200, dstate=16, n_groups=4, n_chunks=1)
    pr.ckpt_pitch = 16
    assert lib.sigma_scan_fwd_plan(ctypes.byref(pr), ctypes.byref(plan)) == 0 and plan[4] == 1 and plan[2] == 768
    # a training batch has enough rows: no sequence split; long 16-state rows take 1280-element tiles ...
    pb = _params(batch=16, dim=768, seqlen=19200, dstate=16, n_groups=4, n_chunks=10)
    assert lib.sigma_scan_fwd_plan(ctypes.byref(pb), ctypes.byref(plan)) == 0
    assert plan[0] == 20 and plan[1] == 12 and plan[4] == 1 and plan[2] == 16 * 768 // 12
    # ... short ones 640-element tiles, 8 rows per workgroup, two workgroups per CU
    pc = _params(batch=16, dim=3072, seqlen=1200, dstate=16, n_groups=4, n_chunks=1)
    assert lib.sigma_scan_fwd_plan(ctypes.byref(pc), ctypes.byref(plan)) == 0
    assert plan[0] == 10 and plan[1] == 8 and plan[4] == 1 and 2 * plan[3] <= 160 * 1024
    # few-state scans (fusion / decoder): 16 rows share one B/C stage
    pd = _params(batch=8, dim=768, seqlen=19200, dstate=4, n_groups=4, n_chunks=10)
    assert lib.sigma_scan_fwd_plan(ctypes.byref(pd), ctypes.byref(plan)) == 0
    assert plan[0] == 10 and plan[1] == 16 and plan[4] == 1
    _capi.set_option("fwd_waves", 16)
    try:
        assert lib.sigma_scan_fwd_plan(ctypes.byref(p), ctypes.byref(plan)) == 0
        assert plan[1] == 16 and plan[2] == 48 and plan[4] == 1
    finally:
        _capi.set_option("fwd_waves", 0)
    # reference unit-test shape: 24 rows, 2 groups -> 12 rows per group; L = 372 -> tiles of 5 x 64 or 4 x 64
    p = _params(batch=2, dim=24, seqlen=372, dstate=8, n_groups=2, n_chunks=1)
    assert lib.sigma_scan_fwd_plan(ctypes.byref(p), ctypes.byref(plan)) == 0
    assert 12 % plan[1] == 0 and plan[2] == 48 // plan[1]
    bp = _capi.BwdParams()
    bp.fwd = p
    assert lib.sigma_scan_bwd_plan(ctypes.byref(bp), ctypes.byref(plan)) == 0
    assert plan[0] in (4, 5, 10) and plan[3] <= 160 * 1024 and 12 % plan[1] == 0
    # P = 12 / rows partial slabs of (B, G, N, L) for each of dB, dC (none when one workgroup owns the group)
    P = 12 // plan[1]
    assert lib.sigma_scan_bwd_workspace_bytes(ctypes.byref(bp)) == (0 if P == 1 else 2 * P * 2 * 2 * 8 * 372 * 4)
    # extension fields are validated
    bad = _params(batch=1, dim=8, seqlen=64, dstate=4, n_groups=2, n_chunks=1)
    bad.rev_group_mask = 0b100
    assert lib.sigma_scan_fwd_plan(ctypes.byref(bad), ctypes.byref(plan)) != 0


def test_operator_module_raises_without_gpu_tensors():
    import torch
    from sigma_amd import selective_scan_cuda_core as core
    u = torch.randn(1, 8, 16)
    A = -torch.rand(8, 4)
    Bm = torch.randn(1, 1, 4, 16)
    with pytest.raises(RuntimeError, match="GPU tensor"):
        core.fwd(u, u, A, Bm, Bm, None, None, False, 1)


def test_gemm_workspace_query_is_host_side_planning():
    """sigma_gemm_workspace_bytes needs no GPU: the scratch of the two-stage sums follows from the launch geometry --
    reduction slices x output bytes for a weight gradient (tn), problems per output for shared outputs (c_mod), nothing
    when every work item owns its output; bad arguments answer -1."""
    lib = _capi.load()

    def params(M, N, K, lda, ldb, ldc, **kw):
        p = _capi.GemmParams()
        p.M, p.N, p.K, p.lda, p.ldb, p.ldc = M, N, K, lda, ldb, ldc
        p.A, p.Bt, p.C = 0x10000, 0x20000, 0x30000          # never dereferenced by the query (16-byte aligned non-null)
        p.batch, p.pieces = kw.pop("batch", 1), 2
        for k, v in kw.items():
            setattr(p, k, v)
        return p

    q = lambda p, form: int(lib.sigma_gemm_workspace_bytes(ctypes.byref(p), form))
    # forward / input gradient of a linear layer: every tile owns its output
    assert q(params(19200, 1536, 384, 384, 384, 1536), 0) == 0
    assert q(params(19200, 384, 1536, 1536, 384, 384), 1) == 0
    # weight gradient of the stage-2 in_proj (19200 tokens, 1536 x 384): 36 tiles -> 14 slices in one round of 512 workgroups
    assert q(params(19200, 1536, 384, 1536, 384, 384), 2) == 14 * 1536 * 384 * 4
    # few tokens: one slice, no scratch
    assert q(params(200, 72, 44, 72, 44, 44), 2) == 0
    # the sliced nn form (dW_x = dx^T x with dx channel-major): 768 x 384 output, 19200-long reduction
    need = q(params(768, 384, 19200, 19200, 384, 384, k_slices=1), 1)
    assert need > 0 and need % (768 * 384 * 4) == 0 and need // (768 * 384 * 4) <= 512 // 18
    # shared outputs: 32 problems summed into 2 outputs of 112 x 768 -> 16 parts per output
    sh = params(112, 768, 1200, 1200, 1200, 768, batch=32, c_mod=2, strideA=112 * 1200, strideB=768 * 1200, strideC=112 * 768, accumulate=1)
    assert q(sh, 0) == 2 * 16 * 112 * 768 * 4
    # bad arguments
    assert q(params(19200, 1536, 384, 384, 384, 1536), 7) == -1
    bad = params(19200, 1536, 383, 384, 384, 1536)
    assert q(bad, 0) == -1
