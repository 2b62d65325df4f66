"""INTEGRATION.md §1 (unmodified reference solvers + engine operators) — runs only in the build container, where the
reference tree exists; skipped everywhere else (nothing of the reference travels).  See tools/check_dropin.py."""
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("TRIPS_REFERENCE", "/root/reference")


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "trips")), reason="reference tree not present (build container only)")
def test_unmodified_reference_solvers_accept_engine_operators():
    r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "check_dropin.py")], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "drop-in cases agree" in r.stdout
