#!/usr/bin/env python3
"""Static view of hipcc's ISA for one kernel: the loops (backward branches) and an instruction-class histogram of each.

    python tools/isa_loops.py file.s <kernel-name-substring> [--dump N]

Used to iterate on the instruction stream of the scan kernels without a GPU (round 6)."""
import re
import sys
from collections import Counter


def classify(op):
    if op.startswith("v_exp") or op.startswith("v_log") or op.startswith("v_rcp") or op.startswith("v_rsq") or op.startswith("v_sqrt"):
        return "trans"
    if op.startswith("v_readlane") or op.startswith("v_writelane") or op.startswith("v_readfirstlane"):
        return "lane_mov"
    if op.startswith("v_permlane"):
        return "permlane"
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("v_accvgpr"):
        return "accvgpr"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "smem"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith("s_nop"):
        return "nop"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("global_load_lds") or op.startswith("buffer_load") and "lds" in op:
        return "dma"
    if op.startswith("global_load") or op.startswith("buffer_load"):
        return "vload"
    if op.startswith("global_store") or op.startswith("buffer_store"):
        return "vstore"
    if op.startswith("global_atomic"):
        return "atomic"
    if op.startswith("scratch_"):
        return "scratch"
    return "other"


def main():
    path, name = sys.argv[1], sys.argv[2]
    dump = int(sys.argv[sys.argv.index("--dump") + 1]) if "--dump" in sys.argv else -1
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*:", l) and name in l)
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    body = lines[start:end]
    labels = {}
    insts = []   # (index in body, op, full)
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = len(insts)
            continue
        s = l.strip()
        if not s or s.startswith(";") or s.startswith("."):
            continue
        op = s.split()[0]
        insts.append((i, op, s))
    loops = []
    for k, (i, op, s) in enumerate(insts):
        if op.startswith("s_cbranch") or op.startswith("s_branch"):
            tgt = s.split()[-1]
            if tgt in labels and labels[tgt] <= k:
                loops.append((labels[tgt], k, tgt))
    print(f"kernel {name}: {len(insts)} instructions, {len(loops)} loops")
    for n, (a, b, tgt) in enumerate(loops):
        c = Counter(classify(op) for _, op, _ in insts[a:b + 1])
        tot = b - a + 1
        print(f"loop {n} {tgt}: {tot} instr  " + "  ".join(f"{k} {v}" for k, v in sorted(c.items(), key=lambda kv: -kv[1])))
        if n == dump:
            for _, _, s in insts[a:b + 1]:
                print("    " + s.split(";")[0].rstrip())


if __name__ == "__main__":
    main()
