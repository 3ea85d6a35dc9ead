"""Drop-in for the reference's Python package ``selective_scan``
(models/encoders/selective_scan/selective_scan/__init__.py:8): the three public names."""
from .selective_scan_interface import SelectiveScanFn, selective_scan_fn, selective_scan_ref

__all__ = ["SelectiveScanFn", "selective_scan_fn", "selective_scan_ref"]
