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


@pytest.mark.skipif(not os.path.isfile(os.path.join(REF, "demos", "demo_Tomo_small_scale.ipynb")),
                    reason="reference tree not present (build container only)")
def test_fanbeam_demo_fixture_regenerates_from_the_reference_notebook(tmp_path):
    """tests/golden/fanbeam_demo_image.npz is what tools/make_fanbeam_demo_golden.py decodes from the notebook's stored outputs."""
    import numpy as np
    out = tmp_path / "fan.npz"
    r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "make_fanbeam_demo_golden.py"), str(out)],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    with np.load(out) as a, np.load(os.path.join(REPO, "tests", "golden", "fanbeam_demo_image.npz")) as b:
        assert sorted(a.files) == sorted(b.files)
        for k in a.files:
            assert np.array_equal(a[k], b[k]), k
