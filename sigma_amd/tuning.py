"""Tuned library GEMMs for the projections (in_proj / out_proj / x_proj / dt_proj, PatchMerging, decoder
linears; reference models/encoders/vmamba.py:1067-1089, 612-636).

The projections are plain GEMMs and stay on the vendor libraries (rocBLAS / hipBLASLt through
``torch.matmul``); what the library picks by default for the model's skinny shapes is far from its best
solution -- e.g. the weight-gradient GEMM (1536 x 19200) x (19200 x 384) runs at 44 TFLOP/s of the 157
TFLOP/s fp32 MFMA peak.  ``tools/tune_gemms.py`` searches every solution of both libraries for every GEMM
shape of the training step once (PyTorch TunableOp) on an MI355X; the table is committed under
``sigma_amd/tuning/`` and only LOOKED UP at run time (tuning disabled: no search, no timing noise).
PyTorch validates the table against the ROCm / hipBLASLt / rocBLAS versions recorded in it and ignores
it on a mismatch, in which case the default heuristics apply.

``SIGMA_TUNED_GEMMS=0`` disables the lookup (A/B runs)."""
from __future__ import annotations

import os

_HERE = os.path.dirname(os.path.abspath(__file__))
TABLE = os.path.join(_HERE, "tuning", "tunableop_mi355x.csv")
_state = {"enabled": None}


def enable_tuned_gemms(table: str = TABLE) -> bool:
    """Idempotent; returns True when the lookup table is active."""
    if _state["enabled"] is not None:
        return _state["enabled"]
    ok = False
    try:
        import torch
        if os.environ.get("SIGMA_TUNED_GEMMS", "1") != "0" and torch.cuda.is_available() and os.path.exists(table):
            import torch.cuda.tunable as tun
            tun.set_filename(table)
            tun.tuning_enable(False)          # look up only; unknown shapes use the library default
            tun.enable(True)
            ok = True
    except Exception as e:                    # noqa: BLE001 -- an optimisation, never a reason to fail
        print(f"sigma_amd.tuning: tuned GEMM table not used ({e})")
    _state["enabled"] = ok
    return ok
