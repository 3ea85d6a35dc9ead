"""Launch planning of libsigma_hip.so on CPU (no kernel runs): invariants of every plan the model's shapes get."""
import ctypes
import itertools

import pytest

from sigma_amd import _capi

MODEL_DIMS = [(768, 16, 4), (1536, 16, 4), (3072, 16, 4), (6144, 16, 4), (192, 4, 1), (384, 4, 2), (768, 4, 4), (1536, 4, 4),
              (3072, 4, 4), (1024, 16, 4), (4096, 16, 4), (8192, 16, 4)]
LENGTHS = [300, 900, 1200, 3600, 4800, 14400, 19200, 38400, 57600]
BATCHES = [1, 2, 8, 16]
LDS_LIMIT = 160 * 1024


def _params(batch, dim, L, N, G, pitch):
    bp = _capi.BwdParams()
    fp = bp.fwd
    fp.batch, fp.dim, fp.seqlen, fp.dstate, fp.n_groups = batch, dim, L, N, G
    fp.n_chunks = (L + _capi.SIGMA_SCAN_CHUNK - 1) // _capi.SIGMA_SCAN_CHUNK
    fp.ckpt_pitch = pitch
    fp.x_row_stride = ((L + pitch - 1) // pitch) * N if pitch else 0
    fp.io_dtype = _capi.DTYPE_F32
    # contiguous tensors: strides matter for the alignment checks of the LDS-DMA staging
    fp.B_dstate_stride = fp.C_dstate_stride = L
    fp.B_group_stride = fp.C_group_stride = N * L
    fp.B_batch_stride = fp.C_batch_stride = G * N * L
    fp.u_d_stride = fp.delta_d_stride = L
    fp.u_batch_stride = fp.delta_batch_stride = dim * L
    return bp


def _plan(fn, bp):
    plan = (ctypes.c_int32 * 6)()
    rc = fn(ctypes.byref(bp if fn is _capi.load().sigma_scan_bwd_plan else bp.fwd), ctypes.byref(plan))
    return rc, list(plan)


@pytest.mark.parametrize("pitch", [0, 640, 320, 160])
def test_every_model_shape_gets_a_legal_plan(pitch):
    lib = _capi.load()
    seen = 0
    for (dim, N, G), L, batch in itertools.product(MODEL_DIMS, LENGTHS, BATCHES):
        bp = _params(batch, dim, L, N, G, pitch)
        rc, f = _plan(lib.sigma_scan_fwd_plan, bp)
        assert rc == 0, (_capi.last_error(), dim, L, batch)
        items, rows, grid, lds = f[0], f[1], f[2], f[3]
        assert 0 < lds <= LDS_LIMIT and grid >= 1 and 1 <= rows <= 16, (f, dim, L, batch)
        rpg = dim // G
        if f[5] == -100:                       # quad-row forward: rows slot = waves of 4 rows, <= 8
            assert pitch == 160 and rows <= 8 and (rpg // 4) % rows == 0 and grid == batch * G * (rpg // 4 // rows)
        rc, b = _plan(lib.sigma_scan_bwd_plan, bp)
        assert rc == 0, (_capi.last_error(), dim, L, batch, pitch)
        items, rows, grid, lds, t4, t5 = b
        assert 0 < lds <= LDS_LIMIT and grid >= 1 and 1 <= rows <= 16, (b, dim, L, batch)
        ws = lib.sigma_scan_bwd_workspace_bytes(ctypes.byref(bp))
        assert ws >= 0
        if t5 <= -100:                         # quad-row backward: W waves x 4 rows x RB row blocks per workgroup
            W, RB, SB = rows, -t4, -t5 - 100
            S = items // 1000 if items >= 1000 else 1              # sequence segments
            assert pitch == 160 and items % 1000 == 10 and SB in (1, 2, 4, 8) and N % SB == 0
            quads = rpg // 4
            assert quads % W == 0 and (quads // W) % RB == 0
            P = quads // W // RB
            assert grid == batch * G * P * S
            ntiles = (L + 159) // 160
            seg_tiles = (ntiles + S - 1) // S
            assert S == 1 or (seg_tiles >= 2 and seg_tiles * (S - 1) < ntiles)
            assert ws == (0 if P == 1 else 2 * P * batch * G * N * L * 4) + (S - 1) * batch * dim * N * 8
        elif t4 < 0:                           # second generation / state-parallel: -t4 row blocks per workgroup
            assert pitch in (640, 320)
        seen += 1
    assert seen == len(MODEL_DIMS) * len(LENGTHS) * len(BATCHES)


def test_quad_row_plan_refuses_what_the_kernel_cannot_take():
    lib = _capi.load()
    for kw in (dict(L=322), dict(N=32), dict(dim=36, G=4)):            # L % 4, dstate, rows per group % 4
        a = dict(batch=2, dim=768, L=1200, N=16, G=4)
        a.update(kw)
        bp = _params(a["batch"], a["dim"], a["L"], a["N"], a["G"], 160)
        rc, _ = _plan(lib.sigma_scan_bwd_plan, bp)
        assert rc != 0 and "160" in _capi.last_error()
        rc, f = _plan(lib.sigma_scan_fwd_plan, bp)                    # the forward falls back to the 64-lane kernel
        assert rc == 0 and f[5] != -100
    bp = _params(2, 768, 1200, 16, 4, 160)
    bp.fwd.io_dtype = _capi.DTYPE_BF16
    rc, _ = _plan(lib.sigma_scan_bwd_plan, bp)
    assert rc != 0


def test_dominant_launch_plans_are_the_documented_ones():
    """DESIGN.md 4.2 / profiles/r02_bwd4_shapes.txt: (16,3072,1200,N16) -> 16 waves x 3 row blocks, P = 4;
    (16,768,19200,N16) -> 12 waves, one row block (256 workgroups)."""
    lib = _capi.load()
    rc, b = _plan(lib.sigma_scan_bwd_plan, _params(16, 3072, 1200, 16, 4, 160))
    assert rc == 0 and b[:3] == [10, 16, 256] and b[4] == -3 and b[5] == -102
    rc, b = _plan(lib.sigma_scan_bwd_plan, _params(16, 768, 19200, 16, 4, 160))
    assert rc == 0 and b[:3] == [10, 12, 256] and b[4] == -1
    rc, f = _plan(lib.sigma_scan_fwd_plan, _params(16, 3072, 1200, 16, 4, 160))
    assert rc == 0 and f[0] == 10 and f[1] == 8 and f[5] == -100


def test_row_lane_plans_of_every_model_shape():
    """ckpt_pitch 16 (csrc/scan_fwdr.hip / scan_bwdr.hip): every model shape with rows per group divisible by 64 gets a
    legal plan -- state waves dividing dstate, segments of >= 2 tiles none of them empty, workspace = slabs + summaries +
    the hand-over slots of the chained walk -- and the others are refused by BOTH entry points."""
    lib = _capi.load()
    seen = 0
    for (dim, N, G), L, batch in itertools.product(MODEL_DIMS, LENGTHS, BATCHES):
        bp = _params(batch, dim, L, N, G, 16)
        rpg = dim // G
        rc, f = _plan(lib.sigma_scan_fwd_plan, bp)
        rcb, b = _plan(lib.sigma_scan_bwd_plan, bp)
        if rpg % 64 != 0:
            assert rc != 0 and rcb != 0 and "16" in _capi.last_error()
            assert lib.sigma_scan_bwd_workspace_bytes(ctypes.byref(bp)) < 0
            continue
        assert rc == 0 and rcb == 0, (_capi.last_error(), dim, L, batch)
        ntiles = (L + 15) // 16
        P = rpg // 64
        for plan, backward in ((f, False), (b, True)):
            items, nw, grid, lds, S, tag = plan
            assert items == 16 and tag == -200 and 0 < lds <= LDS_LIMIT // 2
            assert nw in (4, 8, 16) and N % nw == 0 and (not backward or nw == 4)
            assert 1 <= S <= 64 and grid == batch * G * P * S
            st = (ntiles + S - 1) // S
            assert S == 1 or (st >= 2 and st * (S - 1) < ntiles)
            summ = (S - 1) * batch * dim * N * 2 * 4
            if backward:
                nrb = batch * dim // 64
                chain = (nrb * N * 64 + (nrb + 3) // 4 * 4) * 4
                slabs = 0 if P == 1 else 2 * P * batch * G * N * L * 4
                assert lib.sigma_scan_bwd_workspace_bytes(ctypes.byref(bp)) == slabs + summ + chain
            else:
                assert lib.sigma_scan_fwd_workspace_bytes(ctypes.byref(bp.fwd)) == summ
        seen += 1
    assert seen > 300


def test_row_lane_policy_matches_the_measured_table():
    """ss2d_fused.rowlane_pays against profiles/r04_rowlane_vs_auto.txt (forward + backward time of the row-lane kernels
    against the round-3 planner's kernels on MI355X): the policy picks the faster side on every launch shape of the
    batch-8 and the one-image training step, except where the two are within 5 %."""
    import json
    import os
    from sigma_amd.ss2d_fused import ckpt_pitch_for, rowlane_pays
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles")
    a = {json.loads(l)["shape"]: json.loads(l) for l in open(os.path.join(root, "r04_rowlane_p16_scan_bench.jsonl"))}
    b = {json.loads(l)["shape"]: json.loads(l) for l in open(os.path.join(root, "r04_rowlane_auto_scan_bench.jsonl"))}
    assert len(a) >= 30
    for k in a:
        B, KD, L, N, G = a[k]["dims"]
        ta, tb = a[k]["fwd_us"] + a[k]["bwd_us"], b[k]["fwd_us"] + b[k]["bwd_us"]
        if abs(ta - tb) <= 0.05 * tb:
            continue
        assert rowlane_pays(L, N, B * KD, G) == (ta < tb), (k, ta, tb)
    # the dominant launch of the benchmark step and SURVEY's headline shape go to the row-lane kernels, when they can take them
    assert ckpt_pitch_for(1200, 16, 16 * 3072, True, True, 4) == 16
    assert ckpt_pitch_for(19200, 16, 768, True, True, 4) == 16
    assert ckpt_pitch_for(1200, 16, 16 * 3072, True, False, 4) == 160
    assert ckpt_pitch_for(19200, 16, 16 * 768, True, True, 4) == 160
