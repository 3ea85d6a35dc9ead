#!/usr/bin/env python3
"""Turn the FETCH_SIZE / WRITE_SIZE passes of tools/gpu_pmc.sh into profiles/scan_traffic.json.

    python tools/pmc_traffic.py gpurun_out/<tag> <shape-name> [...]  ->  profiles/scan_traffic.json

FETCH_SIZE / WRITE_SIZE are in KiB... on gfx950 (this image's rocprofv3) FETCH_SIZE reports half the
bytes of a wide coalesced streaming read (MI355X_MICROARCH.md, HBM section), so it is doubled;
WRITE_SIZE is used as reported.  Values are means per dispatch of the named kernel."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.scan_bench import SHAPES  # noqa: E402


def main():
    tag_dir = sys.argv[1]
    out = os.path.join(ROOT, "profiles", "scan_traffic.json")
    try:
        table = json.load(open(out))
    except OSError:
        table = dict(note="HBM bytes per launch from rocprofv3 PMC (FETCH_SIZE x 2 x 1024 + WRITE_SIZE x 1024); "
                          "see tools/pmc_traffic.py", entries=[])
    for name in sys.argv[2:]:
        d = os.path.join(tag_dir, name)
        vals = collections.defaultdict(lambda: collections.defaultdict(list))
        for f in glob.glob(os.path.join(d, "p*", "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
                    kn = r["Kernel_Name"]
                    k = "scan_fwd" if any(t in kn for t in ("scan_fwd_kernel", "scan_fwd4_kernel", "scan_fwdr_kernel")) else \
                        "scan_bwd" if any(t in kn for t in ("scan_bwd_kernel", "scan_bwd2_kernel", "scan_bwd3_kernel", "scan_bwd4_kernel",
                                                            "scan_bwdr_kernel")) else \
                        "reduce_partials" if "reduce_partials" in kn else None
                    if k:
                        vals[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        B, KD, L, N, G = SHAPES[name]
        def mean_of_the_benchmarked_launches(v):
            # the process also runs the load-time self tests (a tiny launch of the same kernels, sigma_scan_rowlane_selftest):
            # only dispatches within a factor of two of the largest one are launches of the benchmarked shape
            keep = [x for x in v if x >= 0.5 * max(v)]
            return sum(keep) / len(keep)
        for k in ("scan_fwd", "scan_bwd"):
            if not vals[k]:
                continue
            fetch = mean_of_the_benchmarked_launches(vals[k]["FETCH_SIZE"])
            write = mean_of_the_benchmarked_launches(vals[k]["WRITE_SIZE"])
            extra = {}
            if k == "scan_bwd" and vals["reduce_partials"]:
                rp = vals["reduce_partials"]
                extra["reduce_partials_bytes_per_launch"] = int((2 * mean_of_the_benchmarked_launches(rp["FETCH_SIZE"]) +
                                                                 mean_of_the_benchmarked_launches(rp["WRITE_SIZE"])) * 1024)
            e = dict(kernel=k, shape=[B, KD, L, N, G], fetch_size_KiB=fetch, write_size_KiB=write,
                     hbm_bytes_per_launch=int((2 * fetch + write) * 1024), source=os.path.relpath(d, ROOT), **extra)
            table["entries"] = [x for x in table["entries"] if not (x["kernel"] == k and x["shape"] == e["shape"])] + [e]
            print(e)
    json.dump(table, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
