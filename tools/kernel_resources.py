#!/usr/bin/env python3
"""Compact per-kernel resource table (VGPRs, spills, scratch, occupancy) from hipcc's
-Rpass-analysis=kernel-resource-usage.   python tools/kernel_resources.py [file.hip ...]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sigma_amd import build as B  # noqa: E402


def main():
    srcs = sys.argv[1:] or [s for s in B.SOURCES if s != "capi.hip"]
    for src in srcs:
        path = src if os.path.exists(src) else os.path.join(B.CSRC, src)
        cmd = [B.HIPCC, *B.FLAGS, "-Rpass-analysis=kernel-resource-usage", "-c", path, "-o", "/dev/null"]
        err = subprocess.run(cmd, capture_output=True, text=True).stderr
        cur = None
        rows = {}
        for line in err.splitlines():
            m = re.search(r"remark:\s+(Function Name|VGPRs|VGPRs Spill|SGPRs Spill|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs): (\S+)", line)
            if not m:
                if "error" in line:
                    print(line)
                continue
            k, v = m.groups()
            if k == "Function Name":
                cur = subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()
                cur = re.sub(r"\(sigma::\w+\)$", "", cur).replace("void sigma::", "").replace("sigma::", "")
                rows[cur] = {}
            elif cur:
                rows[cur][k] = v
        for name, r in rows.items():
            print(f"{name:60s} vgpr {r.get('VGPRs','?'):>4} sgpr {r.get('SGPRs','?'):>4} vspill {r.get('VGPRs Spill','?'):>4} "
                  f"sspill {r.get('SGPRs Spill','?'):>4} scratch {r.get('ScratchSize [bytes/lane]','?'):>4} occ {r.get('Occupancy [waves/SIMD]','?')}")


if __name__ == "__main__":
    main()
