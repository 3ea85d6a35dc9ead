"""Static check of the hand-counted vector-memory wait of the row-lane backward (csrc/scan_bwdr.hip) in the gfx950 ISA that
THIS toolchain emits -- no GPU needed (hipcc cross-compiles; ~15 s).

The tile loop requests the next tile's u / delta / dout with three ``global_load_lds_dwordx4`` from inline assembly and, one
iteration later, retires them with ``s_waitcnt vmcnt(N)``, N = 3 + 2 NS: vector-memory operations retire in order, so the
requests have landed once at most the K operations issued AFTER them are outstanding -- which is only true while K >= N.  A
compiler that drops, merges or moves one of those younger loads / stores in front of the requests (or an edit that changes N
without changing the loop) makes the wait a no-op for the last request: wrong u / delta / dout, silently and only under
memory pressure.  A run-time self test on an idle chip does not see that (tools/diag/r5_selftest_power.sh: a build with
N + 3 passes sigma_scan_rowlane_selftest three times out of three), so the count is checked where it is decided: in the
ISA, with N read from the instruction itself.  The same build is caught here ("only 11 ... keeps 14")."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VMEM = re.compile(r"^\s+(global_load|global_store|global_atomic|buffer_|scratch_|flat_)")


def _asm(tmp_path_factory, extra=()):
    from sigma_amd import build
    out = tmp_path_factory.mktemp("isa") / "scan_bwdr.s"
    flags = [f for f in build.FLAGS if f != "-fPIC"] + list(extra)
    subprocess.check_call([build.HIPCC, *flags, "--offload-device-only", "-S", os.path.join(build.CSRC, "scan_bwdr.hip"), "-o", str(out)],
                          stderr=subprocess.DEVNULL)
    return out.read_text().split("\n")


@pytest.fixture(scope="module")
def asm(tmp_path_factory):
    from sigma_amd import build
    if not os.path.exists(build.HIPCC):                     # the compiler the build itself would use
        pytest.skip(f"no hipcc at {build.HIPCC}")
    return _asm(tmp_path_factory)


def _function(lines, ns, mode):
    name = f"_ZN5sigma16scan_bwdr_kernelILi{ns}ELi{mode}EEEvNS_7BwdArgsE"
    start = next(i for i, l in enumerate(lines) if l.startswith(name + ":"))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    return lines[start:end]


def _tile_loops(body):
    """(header line, back-edge line) of every loop of `body` that issues LDS-DMA requests"""
    loops = []
    for h, l in enumerate(body):
        if "Loop Header" not in l:
            continue
        label = l.split(":")[0].strip()
        if not label.startswith(".LBB"):                    # nested loops: the label sits on the line above the comment
            label = body[h - 1].split(":")[0].strip()
        back = [j for j in range(h + 1, len(body)) if re.search(rf"s_cbranch_\w+\s+{re.escape(label)}\s*$", body[j].split(";")[0])]
        if back and any("global_load_lds_dwordx4" in body[j] for j in range(h, back[0])):
            inner_headers = [j for j in range(h + 1, back[0]) if "Loop Header" in body[j]]
            if not inner_headers:                           # innermost only
                loops.append((h, back[0]))
    return loops


def check_counted_wait(body):
    """every tile loop: the first vmcnt wait of an iteration (the one in front of the read-back of the landed bytes) keeps N
    operations in flight; at least N vector-memory operations must follow the three requests on the way round the loop"""
    loops = _tile_loops(body)
    assert loops, "no tile loop with LDS-DMA requests found"
    for h, b in loops:
        inner = [j for j in range(h + 1, b) if re.search(r"s_cbranch_|s_branch", body[j].split(";")[0])]
        assert not inner, f"control flow inside the tile loop (line {inner[0]}): the static count below would not hold on every path"
        dma = [j for j in range(h, b) if "global_load_lds_dwordx4" in body[j]]
        assert len(dma) == 3, "three LDS-DMA requests per tile"
        # the read-back of the landed bytes = the first LDS read after the loop header; every vmcnt wait in front of it
        # counts (the compiler may add waits of its own for loop-carried loads): the bytes are there if ONE of them keeps
        # no more operations in flight than vector-memory operations were issued between the requests and that wait
        first_read = next((j for j in range(h, dma[0]) if re.match(r"\s+ds_read", body[j])), dma[0])
        waits = [(j, int(m.group(1))) for j in range(h, first_read) for m in [re.search(r"s_waitcnt.*vmcnt\((\d+)\)", body[j].split(";")[0])] if m]
        assert waits, "no vmcnt wait between the loop header and the read-back of the LDS-DMA bytes"
        report = []
        for w, keep in waits:
            younger = [j for j in list(range(dma[-1] + 1, b)) + list(range(h, w)) if VMEM.match(body[j])]
            report.append((keep, len(younger)))
        assert any(n >= keep for keep, n in report), (
            "no wait in front of the read-back covers the LDS-DMA requests of a tile: (kept in flight, vector-memory operations issued "
            f"after the requests) = {report}: the last request would not be waited for")
    return len(loops)


@pytest.mark.parametrize("ns,mode", [(4, 0), (2, 0), (1, 0), (4, 2)], ids=["N16", "N8", "N4", "N16-chained"])
def test_counted_wait_of_the_row_lane_backward_is_covered_by_younger_operations(asm, ns, mode):
    assert check_counted_wait(_function(asm, ns, mode)) >= 1


def test_a_miscounted_wait_is_refused(tmp_path_factory):
    """the check has teeth: a build whose wait keeps three operations too many in flight (the build a run-time self test on an
    idle chip passes, tools/diag/README.md) fails it"""
    from sigma_amd import build
    if not os.path.exists(build.HIPCC):
        pytest.skip(f"no hipcc at {build.HIPCC}")
    bad = _asm(tmp_path_factory, extra=["-DSIGMA_BWDR_WAIT_SKEW=3"])
    with pytest.raises(AssertionError, match="no wait in front of the read-back covers"):
        check_counted_wait(_function(bad, 4, 0))
