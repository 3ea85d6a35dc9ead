"""ctypes binding of libsigma_hip.so (C ABI declared in include/sigma_scan.h).

The library is REQUIRED: importing the operator modules without it raises
``SigmaHipUnavailable`` -- there is no CPU or eager fallback anywhere in the product
path (the CPU oracle under oracle/ is test infrastructure only).
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
# SIGMA_HIP_LIB lets a benchmark A/B an experimental build of the same ABI; default is the in-tree library
LIB_PATH = os.environ.get("SIGMA_HIP_LIB") or os.path.join(_HERE, "lib", "libsigma_hip.so")

SIGMA_SCAN_ABI_VERSION = 10
SIGMA_SCAN_CHUNK = 2048
SIGMA_SCAN_CKPT_PITCH = 1280
SIGMA_SCAN_CKPT_PITCH_FINE = 640
SIGMA_SCAN_CKPT_PITCH_320 = 320
SIGMA_SCAN_CKPT_PITCH_160 = 160
SIGMA_SCAN_CKPT_PITCH_16 = 16
SIGMA_SCAN_MAX_DSTATE = 256

DTYPE_F32, DTYPE_F16, DTYPE_BF16 = 0, 1, 2


class SigmaHipUnavailable(RuntimeError):
    pass


class FwdParams(ctypes.Structure):
    _fields_ = [
        ("batch", ctypes.c_int32), ("dim", ctypes.c_int32), ("seqlen", ctypes.c_int32),
        ("dstate", ctypes.c_int32), ("n_groups", ctypes.c_int32), ("n_chunks", ctypes.c_int32),
        ("io_dtype", ctypes.c_int32), ("delta_softplus", ctypes.c_int32),
        ("rev_group_mask", ctypes.c_uint32), ("u_group_shift", ctypes.c_int32),
        ("ckpt_pitch", ctypes.c_int32), ("param_group_swap", ctypes.c_int32), ("x_row_stride", ctypes.c_int64),
        ("u", ctypes.c_void_p), ("delta", ctypes.c_void_p), ("A", ctypes.c_void_p),
        ("B", ctypes.c_void_p), ("C", ctypes.c_void_p), ("D", ctypes.c_void_p),
        ("delta_bias", ctypes.c_void_p),
        ("out", ctypes.c_void_p), ("x", ctypes.c_void_p),
        ("u_batch_stride", ctypes.c_int64), ("u_d_stride", ctypes.c_int64),
        ("delta_batch_stride", ctypes.c_int64), ("delta_d_stride", ctypes.c_int64),
        ("A_d_stride", ctypes.c_int64), ("A_dstate_stride", ctypes.c_int64),
        ("B_batch_stride", ctypes.c_int64), ("B_group_stride", ctypes.c_int64), ("B_dstate_stride", ctypes.c_int64),
        ("C_batch_stride", ctypes.c_int64), ("C_group_stride", ctypes.c_int64), ("C_dstate_stride", ctypes.c_int64),
        ("out_batch_stride", ctypes.c_int64), ("out_d_stride", ctypes.c_int64),
        ("workspace", ctypes.c_void_p), ("workspace_bytes", ctypes.c_int64),
    ]


class BwdParams(ctypes.Structure):
    _fields_ = [
        ("fwd", FwdParams),
        ("dout", ctypes.c_void_p), ("du", ctypes.c_void_p), ("ddelta", ctypes.c_void_p),
        ("dA", ctypes.c_void_p), ("dB", ctypes.c_void_p), ("dC", ctypes.c_void_p),
        ("dD", ctypes.c_void_p), ("ddelta_bias", ctypes.c_void_p),
        ("workspace", ctypes.c_void_p), ("workspace_bytes", ctypes.c_int64),
        ("dout_group_shift", ctypes.c_int32), ("reserved_", ctypes.c_int32),
        ("dout_batch_stride", ctypes.c_int64), ("dout_d_stride", ctypes.c_int64),
        ("du_batch_stride", ctypes.c_int64), ("du_d_stride", ctypes.c_int64),
        ("ddelta_batch_stride", ctypes.c_int64), ("ddelta_d_stride", ctypes.c_int64),
        ("dA_d_stride", ctypes.c_int64), ("dA_dstate_stride", ctypes.c_int64),
        ("dB_batch_stride", ctypes.c_int64), ("dB_group_stride", ctypes.c_int64), ("dB_dstate_stride", ctypes.c_int64),
        ("dC_batch_stride", ctypes.c_int64), ("dC_group_stride", ctypes.c_int64), ("dC_dstate_stride", ctypes.c_int64),
    ]


class DwConvParams(ctypes.Structure):
    """mirror of sigma_dwconv_params (include/sigma_ops.h)"""
    _fields_ = [
        ("batch", ctypes.c_int32), ("channels", ctypes.c_int32), ("height", ctypes.c_int32), ("width", ctypes.c_int32),
        ("n_orders", ctypes.c_int32), ("reserved_", ctypes.c_int32),
        ("x", ctypes.c_void_p), ("weight", ctypes.c_void_p), ("bias", ctypes.c_void_p),
        ("out2", ctypes.c_void_p),
        ("g2", ctypes.c_void_p), ("gpre", ctypes.c_void_p), ("dweight", ctypes.c_void_p), ("dbias", ctypes.c_void_p),
        ("dx", ctypes.c_void_p),
        ("x_batch_stride", ctypes.c_int64), ("x_channel_stride", ctypes.c_int64),
    ]


class MergeParams(ctypes.Structure):
    """mirror of sigma_merge_params (include/sigma_ops.h)"""
    _fields_ = [
        ("batch", ctypes.c_int32), ("channels", ctypes.c_int32), ("height", ctypes.c_int32), ("width", ctypes.c_int32),
        ("planes4", ctypes.c_void_p), ("planes2", ctypes.c_void_p), ("nhwc", ctypes.c_void_p),
    ]


class TransposeParams(ctypes.Structure):
    """mirror of sigma_transpose_params (include/sigma_ops.h)"""
    _fields_ = [
        ("batch", ctypes.c_int32), ("rows", ctypes.c_int32), ("cols", ctypes.c_int32), ("reserved_", ctypes.c_int32),
        ("src", ctypes.c_void_p), ("dst", ctypes.c_void_p),
        ("src_batch_stride", ctypes.c_int64), ("src_row_stride", ctypes.c_int64),
        ("dst_batch_stride", ctypes.c_int64), ("dst_row_stride", ctypes.c_int64),
    ]


class LayerNormParams(ctypes.Structure):
    """mirror of sigma_layernorm_params (include/sigma_ops.h)"""
    _fields_ = [
        ("rows", ctypes.c_int64), ("channels", ctypes.c_int32), ("eps", ctypes.c_float),
        ("x", ctypes.c_void_p), ("gamma", ctypes.c_void_p), ("beta", ctypes.c_void_p),
        ("y", ctypes.c_void_p), ("mean", ctypes.c_void_p), ("rstd", ctypes.c_void_p),
        ("dy", ctypes.c_void_p), ("dx", ctypes.c_void_p), ("dgamma", ctypes.c_void_p), ("dbeta", ctypes.c_void_p),
        ("workspace", ctypes.c_void_p),
        ("gate", ctypes.c_void_p), ("gate_row_stride", ctypes.c_int64), ("dgate", ctypes.c_void_p),
        ("dgate_row_stride", ctypes.c_int64),
        ("row_scale", ctypes.c_void_p), ("rows_per_scale", ctypes.c_int64),
        ("dx_add", ctypes.c_void_p),
    ]


class GemmParams(ctypes.Structure):
    """mirror of sigma_gemm_params (include/sigma_gemm.h)"""
    _fields_ = [
        ("M", ctypes.c_int64), ("N", ctypes.c_int32), ("K", ctypes.c_int32),
        ("A", ctypes.c_void_p), ("Bt", ctypes.c_void_p), ("C", ctypes.c_void_p), ("bias", ctypes.c_void_p),
        ("lda", ctypes.c_int64), ("ldb", ctypes.c_int64), ("ldc", ctypes.c_int64),
        ("accumulate", ctypes.c_int32), ("batch", ctypes.c_int32),
        ("strideA", ctypes.c_int64), ("strideB", ctypes.c_int64), ("strideC", ctypes.c_int64),
        ("a_mod", ctypes.c_int32), ("pieces", ctypes.c_int32),
        ("c_mod", ctypes.c_int32), ("reserved", ctypes.c_int32),
        ("residual", ctypes.c_void_p), ("residual2", ctypes.c_void_p), ("ldr", ctypes.c_int64), ("strideR", ctypes.c_int64),
        ("Ct", ctypes.c_void_p), ("ldct", ctypes.c_int64), ("t_cols", ctypes.c_int32), ("k_slices", ctypes.c_int32),
        ("workspace", ctypes.c_void_p), ("workspace_bytes", ctypes.c_int64),
    ]


class GateBwdParams(ctypes.Structure):
    """mirror of sigma_gate_bwd_params (include/sigma_ops.h)"""
    _fields_ = [
        ("planes", ctypes.c_int64), ("hw", ctypes.c_int64),
        ("g", ctypes.c_void_p), ("x", ctypes.c_void_p), ("scale", ctypes.c_void_p), ("dmean", ctypes.c_void_p),
        ("dmax", ctypes.c_void_p), ("max", ctypes.c_void_p), ("count", ctypes.c_void_p), ("dx", ctypes.c_void_p),
    ]


SIGMA_CE_BLOCKS = 1024      # include/sigma_ops.h

# every symbol include/sigma_gemm.h declares
GEMM_SYMBOLS = ("sigma_gemm_nt_split3", "sigma_gemm_nn_split3", "sigma_gemm_tn_split3")
GEMM_AUX_SYMBOLS = ("sigma_gemm_selftest", "sigma_gemm_workspace_bytes")

# every symbol include/sigma_ops.h declares
OPS_SYMBOLS = ("sigma_dwconv3x3_silu_fwd", "sigma_dwconv3x3_silu_bwd", "sigma_cross_merge_nhwc", "sigma_cross_split_nhwc",
               "sigma_layernorm_fwd", "sigma_layernorm_bwd", "sigma_layernorm_bwd_partial_rows", "sigma_transpose2d",
               "sigma_pair_sum_add", "sigma_upsample2x_nhwc", "sigma_plane_pool", "sigma_plane_scale",
               "sigma_plane_dot", "sigma_plane_gate_bwd", "sigma_softmax_ce_fwd", "sigma_softmax_ce_bwd", "sigma_colscale_bwd")

# every symbol include/sigma_scan.h declares; tests check the library exports all of them
EXPORTED_SYMBOLS = (
    "sigma_selective_scan_fwd",
    "sigma_selective_scan_bwd",
    "sigma_scan_bwd_workspace_bytes",
    "sigma_scan_fwd_workspace_bytes",
    "sigma_scan_last_error",
    "sigma_scan_abi_version",
    "sigma_scan_set_option",
    "sigma_scan_get_option",
    "sigma_scan_fwd_plan",
    "sigma_scan_bwd_plan",
    "sigma_scan_debug_read",
    "sigma_scan_selftest",
    "sigma_scan_rowlane_selftest",
)

_lib: Optional[ctypes.CDLL] = None


def load() -> ctypes.CDLL:
    """Load the HIP library or raise loudly.  Never falls back to anything."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SigmaHipUnavailable(
            f"{LIB_PATH} is missing: build it with `python -m sigma_amd.build` "
            "(hipcc --offload-arch=gfx950).  sigma_amd has no CPU/eager fallback.")
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:  # e.g. libamdhip64 not found
        raise SigmaHipUnavailable(f"cannot load {LIB_PATH}: {e}") from e
    P = ctypes.POINTER
    lib.sigma_selective_scan_fwd.argtypes = [P(FwdParams), ctypes.c_void_p]
    lib.sigma_selective_scan_fwd.restype = ctypes.c_int
    lib.sigma_selective_scan_bwd.argtypes = [P(BwdParams), ctypes.c_void_p]
    lib.sigma_selective_scan_bwd.restype = ctypes.c_int
    lib.sigma_scan_bwd_workspace_bytes.argtypes = [P(BwdParams)]
    lib.sigma_scan_bwd_workspace_bytes.restype = ctypes.c_int64
    lib.sigma_scan_fwd_workspace_bytes.argtypes = [P(FwdParams)]
    lib.sigma_scan_fwd_workspace_bytes.restype = ctypes.c_int64
    lib.sigma_scan_last_error.argtypes = []
    lib.sigma_scan_last_error.restype = ctypes.c_char_p
    lib.sigma_scan_abi_version.argtypes = []
    lib.sigma_scan_abi_version.restype = ctypes.c_int
    lib.sigma_scan_set_option.argtypes = [ctypes.c_char_p, ctypes.c_int]
    lib.sigma_scan_set_option.restype = ctypes.c_int
    lib.sigma_scan_get_option.argtypes = [ctypes.c_char_p]
    lib.sigma_scan_get_option.restype = ctypes.c_int
    lib.sigma_scan_fwd_plan.argtypes = [P(FwdParams), P(ctypes.c_int32 * 6)]
    lib.sigma_scan_fwd_plan.restype = ctypes.c_int
    lib.sigma_scan_bwd_plan.argtypes = [P(BwdParams), P(ctypes.c_int32 * 6)]
    lib.sigma_scan_bwd_plan.restype = ctypes.c_int
    lib.sigma_scan_selftest.argtypes = [ctypes.c_void_p]
    lib.sigma_scan_selftest.restype = ctypes.c_int
    lib.sigma_scan_rowlane_selftest.argtypes = [ctypes.c_void_p]
    lib.sigma_scan_rowlane_selftest.restype = ctypes.c_int
    lib.sigma_scan_debug_read.argtypes = [P(ctypes.c_uint64 * 16)]
    lib.sigma_scan_debug_read.restype = ctypes.c_int
    for name in OPS_SYMBOLS:
        fn = getattr(lib, name)
        if name == "sigma_layernorm_bwd_partial_rows":
            fn.argtypes = [ctypes.c_int64, ctypes.c_int32]
        elif name == "sigma_pair_sum_add":
            fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p]
        elif name == "sigma_upsample2x_nhwc":
            fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                           ctypes.c_int32, ctypes.c_void_p]
        elif name == "sigma_plane_pool":
            fn.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        elif name in ("sigma_plane_scale", "sigma_plane_dot"):
            fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p]
        elif name == "sigma_plane_gate_bwd":
            fn.argtypes = [P(GateBwdParams), ctypes.c_void_p]
        elif name == "sigma_colscale_bwd":
            fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                           ctypes.c_int32, ctypes.c_void_p]
        elif name == "sigma_softmax_ce_fwd":
            fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p,
                           ctypes.c_void_p, ctypes.c_void_p]
        elif name == "sigma_softmax_ce_bwd":
            fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32,
                           ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]
        else:
            st = (MergeParams if "cross_" in name else LayerNormParams if "layernorm" in name else
                  TransposeParams if "transpose" in name else DwConvParams)
            fn.argtypes = [P(st), ctypes.c_void_p]
        fn.restype = ctypes.c_int
    for name in GEMM_SYMBOLS:
        fn = getattr(lib, name)
        fn.argtypes = [P(GemmParams), ctypes.c_void_p]
        fn.restype = ctypes.c_int
    lib.sigma_gemm_selftest.argtypes = [ctypes.c_void_p]
    lib.sigma_gemm_selftest.restype = ctypes.c_int
    lib.sigma_gemm_workspace_bytes.argtypes = [P(GemmParams), ctypes.c_int]
    lib.sigma_gemm_workspace_bytes.restype = ctypes.c_int64
    if lib.sigma_scan_abi_version() != SIGMA_SCAN_ABI_VERSION:
        raise SigmaHipUnavailable(
            f"ABI mismatch: library {lib.sigma_scan_abi_version()} vs binding {SIGMA_SCAN_ABI_VERSION}; rebuild")
    _lib = lib
    return lib


def last_error() -> str:
    return load().sigma_scan_last_error().decode("utf-8", "replace")


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise RuntimeError(f"{what}: {last_error()} (sigma_status {rc})")


def set_option(name: str, value: int) -> None:
    check(load().sigma_scan_set_option(name.encode(), int(value)), f"sigma_scan_set_option({name})")


def get_option(name: str) -> int:
    return int(load().sigma_scan_get_option(name.encode()))
