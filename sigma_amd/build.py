"""Build libsigma_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m sigma_amd.build            # incremental
    python -m sigma_amd.build --force

The library is built IN-TREE (sigma_amd/lib/libsigma_hip.so) so that it travels to the
GPU box with the repo snapshot; it is git-ignored.
"""
from __future__ import annotations

import argparse
import concurrent.futures as cf
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "lib", "obj")
LIB = os.path.join(HERE, "lib", "libsigma_hip.so")
SOURCES = ["scan_fwd.hip", "scan_bwd.hip", "selftest.hip", "capi.hip", "dwconv.hip", "merge.hip"]
HEADERS = ["scan_device.h", "scan_launch.h", os.path.join("..", "..", "include", "sigma_scan.h"),
           os.path.join("..", "..", "include", "sigma_ops.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
    "-munsafe-fp-atomics",          # ds_add_f32 / global_atomic_add_f32 instead of CAS loops
    "-ffp-contract=off",            # every fma in the kernels is written explicitly
    "-Wall", "-Wno-unused-function",
]


def _newer(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src: str, force: bool, extra) -> str:
    obj = os.path.join(OBJ, src.replace(".hip", ".o"))
    deps = [os.path.join(CSRC, src)] + [os.path.join(CSRC, h) for h in HEADERS] + [os.path.abspath(__file__)]
    if force or _newer(obj, deps):
        cmd = [HIPCC, *FLAGS, *extra, "-c", os.path.join(CSRC, src), "-o", obj]
        subprocess.check_call(cmd)
    return obj


def build(force: bool = False, verbose: bool = False, extra=()) -> str:
    os.makedirs(OBJ, exist_ok=True)
    with cf.ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        objs = list(ex.map(lambda s: _compile(s, force, list(extra)), SOURCES))
    if force or _newer(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
        subprocess.check_call(cmd)
    if verbose:
        print(f"built {LIB} ({os.path.getsize(LIB) / 1024:.0f} KiB)")
    return LIB


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--resource-usage", action="store_true", help="print per-kernel VGPR/LDS usage")
    a = ap.parse_args()
    extra = ["-Rpass-analysis=kernel-resource-usage"] if a.resource_usage else []
    build(force=a.force or a.resource_usage, verbose=True, extra=extra)
    sys.exit(0)
