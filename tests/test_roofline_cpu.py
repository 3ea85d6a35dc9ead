"""tools/roofline.py reproduces the algorithmic-byte totals of SURVEY.md 8(d) / Appendix A."""
import json
import subprocess
import sys
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args):
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", "roofline.py"), "--json", *args], text=True)
    return json.loads(out)


def test_scan_byte_totals_match_survey():
    small = _run("--backbone", "sigma_small", "--batch", "1")["summary"]
    assert abs(small["scan_fwd_GB"] - 5.39) < 0.05 and abs(small["scan_bwd_GB"] - 8.99) < 0.02
    tiny = _run("--backbone", "sigma_tiny", "--batch", "1")["summary"]
    assert abs(tiny["scan_fwd_GB"] - 3.77) < 0.05 and abs(tiny["scan_bwd_GB"] - 6.30) < 0.02
    base = _run("--backbone", "sigma_base", "--height", "720", "--width", "1280", "--batch", "1", "--classes", "5")["summary"]
    assert abs(base["scan_fwd_GB"] - 21.43) < 0.2 and abs(base["scan_bwd_GB"] - 35.79) < 0.1


def test_headline_shape_bytes():
    rows = _run("--backbone", "sigma_small", "--batch", "1")["scans"]
    enc0 = [r for r in rows if r["site"] == "enc s0"][0]
    # (2, 768, 19200) N=16 G=4: twice the 186.8 MB of the one-modality call (+ checkpoints)
    assert abs(enc0["fwd_MB"] - 2 * 186.8) < 2.5
