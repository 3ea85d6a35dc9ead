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
SOURCES = ["scan_fwd.hip", "scan_fwd4.hip", "scan_fwdr.hip", "scan_bwdr.hip", "scan_bwd.hip", "scan_bwd2.hip", "scan_bwd3.hip", "scan_bwd4.hip", "selftest.hip", "capi.hip", "dwconv.hip", "merge.hip", "layernorm.hip", "gemm_split.hip", "upsample.hip", "pointwise.hip"]
HEADERS = ["scan_device.h", "scan_launch.h", "scan_quad.h", "scan_rowlane.h", os.path.join("..", "..", "include", "sigma_scan.h"),
           os.path.join("..", "..", "include", "sigma_ops.h"), os.path.join("..", "..", "include", "sigma_gemm.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
    "-munsafe-fp-atomics",          # ds_add_f32 / global_atomic_add_f32 instead of CAS loops
    "-ffp-contract=off",            # every fma in the kernels is written explicitly
    "-fno-slp-vectorize",           # v_pk_*_f32 issue at half rate on gfx950 and the packing costs v_movs (tools/ubench)
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


def build(force: bool = False, verbose: bool = False, extra=(), variant: str = "") -> str:
    """variant: experimental build of the same ABI into lib/libsigma_hip_<variant>.so (select it with
    SIGMA_HIP_LIB=...); `extra` flags apply to every translation unit."""
    global OBJ
    lib = LIB if not variant else LIB.replace(".so", f"_{variant}.so")
    obj_saved = OBJ
    if variant:
        OBJ = os.path.join(HERE, "lib", f"obj_{variant}")
    try:
        os.makedirs(OBJ, exist_ok=True)
        with cf.ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
            objs = list(ex.map(lambda s: _compile(s, force, list(extra)), SOURCES))
        if force or _newer(lib, objs):
            cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs]
            subprocess.check_call(cmd)
    finally:
        OBJ = obj_saved
    if verbose:
        print(f"built {lib} ({os.path.getsize(lib) / 1024:.0f} KiB)")
    return lib


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--resource-usage", action="store_true", help="print per-kernel VGPR/LDS usage")
    ap.add_argument("--variant", default="", help="build lib/libsigma_hip_<variant>.so with --flags")
    ap.add_argument("--flags", default="", help="extra hipcc flags (space separated) for a variant build")
    a = ap.parse_args()
    extra = ["-Rpass-analysis=kernel-resource-usage"] if a.resource_usage else []
    extra += a.flags.split()
    build(force=a.force or a.resource_usage, verbose=True, extra=extra, variant=a.variant)
    sys.exit(0)
