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

One variable, one meaning: ``SIGMA_TUNED_GEMMS=0`` forbids the lookup everywhere (A/B runs); entry points (bench.py,
tools/) call ``enable_tuned_gemms()`` themselves; ``SIGMA_TUNED_GEMMS=1`` additionally makes the model constructor call it,
for an unchanged reference ``train.py`` (the constructor alone never flips the process-wide switch)."""
from __future__ import annotations

import os

_HERE = os.path.dirname(os.path.abspath(__file__))
TABLE = os.path.join(_HERE, "tuning", "tunableop_mi355x.csv")
_state = {"enabled": None}


def table_path() -> str:
    return TABLE


def tuned_gemms_requested() -> bool:
    """True once enable_tuned_gemms() has switched the lookup on in this process."""
    return bool(_state["enabled"])


def enable_tuned_gemms(table: str = TABLE) -> bool:
    """Idempotent; returns True when the lookup was switched on (see ``tuned_gemms_active`` for whether PyTorch
    accepted the table)."""
    if _state["enabled"] is not None:
        return _state["enabled"]
    ok = False
    try:
        import torch
        if os.environ.get("SIGMA_TUNED_GEMMS", "1") != "0" and torch.cuda.is_available() and os.path.exists(table):
            import torch.cuda.tunable as tun
            tun.set_filename(table)
            tun.tuning_enable(False)          # look up only; unknown shapes use the library default
            if hasattr(tun, "write_file_on_exit"):
                tun.write_file_on_exit(False)     # the committed table is never rewritten by a run
            tun.enable(True)
            ok = True
            try:                              # explicit read: False when the validators in the file reject this stack
                _state["accepted"] = bool(tun.read_file(table))
            except Exception:                 # noqa: BLE001
                _state["accepted"] = None
    except Exception as e:                    # noqa: BLE001 -- an optimisation, never a reason to fail
        print(f"sigma_amd.tuning: tuned GEMM table not used ({e})")
    _state["enabled"] = ok
    return ok


def tuned_gemms_active() -> bool:
    """True when the lookup is on AND PyTorch accepted the committed table, i.e. it holds results after the
    validators (ROCm / hipBLASLt / rocBLAS versions recorded in the file) were checked.  TunableOp reads the file
    lazily at the first GEMM, so call this after a step has run.  A rejected table costs ~17 % of the batch-8 step
     and used to go unnoticed; bench.py reports this flag as config.tuned_gemms."""
    if not _state["enabled"]:
        return False
    try:
        import torch.cuda.tunable as tun
        if _state.get("accepted") is False:
            return False
        return tun.is_enabled() and len(tun.get_results()) > 0
    except Exception:                         # noqa: BLE001
        return False
