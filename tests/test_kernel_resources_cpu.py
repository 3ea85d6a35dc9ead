"""The launch planner's residency assumptions against what THIS toolchain emits (no GPU: hipcc's
-Rpass-analysis=kernel-resource-usage while cross-compiling the kernel files; ~1 min).

csrc/capi.hip plans grids from "workgroups a CU holds" per kernel build (plan_rowlane's candidate tables, the 512 slots of the
row-lane backward, two workgroups per CU of the GEMMs): those numbers are register counts of a particular compiler.  A
toolchain that allocates more registers -- or starts spilling the hot loops to scratch -- silently turns a plan of whole
rounds into one with a half-empty tail; this test makes that loud."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _resources(src):
    from sigma_amd import build as B
    if not os.path.exists(B.HIPCC):
        pytest.skip("no hipcc")
    cmd = [B.HIPCC, *B.FLAGS, "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(B.CSRC, src), "-o", os.devnull]
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows, cur = {}, None
    for line in err.splitlines():
        m = re.search(r"remark:\s+(Function Name|VGPRs|AGPRs|VGPRs Spill|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]): (\S+)", line)
        if not m:
            continue
        k, v = m.groups()
        if k == "Function Name":
            cur = subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()
            rows[cur] = {}
        elif cur:
            rows[cur][k] = int(v)
    assert rows, "hipcc printed no resource remarks"
    return rows


def _pick(rows, pattern):
    hits = {k: v for k, v in rows.items() if re.search(pattern, k)}
    assert hits, pattern
    return hits


def _waves_per_simd(r):
    """hardware allocation granule 8 registers, 512 per lane and SIMD, at most 8 waves (MI355X_MICROARCH.md, register files)"""
    alloc = (r["VGPRs"] + r.get("AGPRs", 0) + 7) // 8 * 8
    return min(8, 512 // alloc)


def test_row_lane_backward_fits_the_two_workgroups_its_lds_allows_without_scratch_in_the_tile_loops(tmp_path_factory):
    """Round 6: the whole-tile walk of the 16-state build sits AT the 256 registers two waves per SIMD allow and spills a few
    dozen prologue / epilogue values; what must not happen is scratch traffic INSIDE the tile loops (the loops that issue the
    LDS-DMA requests) -- checked in the ISA."""
    from tests.test_isa_waits_cpu import _asm, _function, _tile_loops
    rows = _resources("scan_bwdr.hip")
    asm = None
    for ns in (4, 2, 1):
        (name, r), = _pick(rows, rf"scan_bwdr_kernel<{ns}, 0>").items()
        # bwdr_lds_bytes(4) = 64 KB -> two 4-wave workgroups per CU = two waves per SIMD (plan_rowlane: {4, 2})
        assert _waves_per_simd(r) >= 2, (name, r)
        if r["ScratchSize [bytes/lane]"] == 0 and r["VGPRs Spill"] == 0:
            continue
        asm = asm or _asm(tmp_path_factory)
        body = _function(asm, ns, 0)
        loops = _tile_loops(body)
        assert loops, name
        for h, b in loops:
            bad = [body[j].strip() for j in range(h, b) if body[j].lstrip().startswith("scratch_")]
            assert not bad, (name, "scratch accesses inside a tile loop", bad[:4])
    for name, r in _pick(rows, r"scan_bwdr_kernel<\d, 1>").items():   # summary pre-pass: declared for four waves per SIMD
        assert _waves_per_simd(r) >= 4, (name, r)


def test_row_lane_forward_residency_matches_the_planner_table():
    rows = _resources("scan_fwdr.hip")
    # plan_rowlane: cf16 = {4 waves: 3 workgroups per CU, 8: 2, 16: 1}, cf8 = {4: 4, 8: 2}, cf4 = {4: 4}; MODE 0 / 1 / 2 alike
    need = {(4, 4): 3, (2, 8): 4, (1, 16): 4, (2, 4): 4, (1, 8): 4, (1, 4): 4}      # (NS, NW) -> waves per SIMD
    for (ns, nw), waves in need.items():
        for name, r in _pick(rows, rf"scan_fwdr_kernel<{ns}, {nw}, \d>").items():
            assert r["ScratchSize [bytes/lane]"] == 0 and r["VGPRs Spill"] == 0, (name, r)
            assert _waves_per_simd(r) >= waves, (name, r)


def test_quad_row_kernels_keep_their_wave_counts():
    rows = _resources("scan_bwd4.hip")
    (name, r), = _pick(rows, r"scan_bwd4_kernel<12>").items()          # 12 waves per workgroup: three per SIMD, no scratch
    assert _waves_per_simd(r) >= 3 and r["ScratchSize [bytes/lane]"] == 0, (name, r)
    (name, r), = _pick(rows, r"scan_bwd4_kernel<16>").items()          # 16 waves: the 128-register build (its few spills
    assert _waves_per_simd(r) >= 4, (name, r)                          # sit in the tile / row prologue)
    (name, r), = _pick(_resources("scan_fwd4.hip"), r"scan_fwd4_kernel").items()
    assert _waves_per_simd(r) >= 4 and r["ScratchSize [bytes/lane]"] == 0, (name, r)   # two 8-wave workgroups per CU


def test_gemm_kernels_hold_two_workgroups_per_cu_without_spills():
    rows = _pick(_resources("gemm_split.hip"), r"gemm_split3_kernel<")
    assert len(rows) >= 20
    for name, r in rows.items():                                       # __launch_bounds__(256, 2); 64 of the registers are accumulators
        assert r["ScratchSize [bytes/lane]"] == 0 and r["VGPRs Spill"] == 0, (name, r)
        assert _waves_per_simd(r) >= 2, (name, r)
