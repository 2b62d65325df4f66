"""Measured agreement of C3 (Hybrid-LSQR 512^2 x 180) and C5 (GKS 32 x 256^2) with the float64 oracle at full size: the numbers the
bars of tests/test_gpu_configs_fullsize.py are set from.  GPU box."""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import test_gpu_configs_fullsize as T  # noqa: E402



def c3(_its):
    """(round 5: C3 is measured at its 100 iterations, against the envelope of the oracle's own steps — T.c3_numbers(solver))"""
    from trips_py_amd import solvers as S
    from trips_py_amd.operators import Radon2DParallel
    m = T.c3_numbers(lambda N, ang, xt, b: S.Hybrid_LSQR(Radon2DParallel(N, ang), b, T.C3_ITS, 1e-2, xt))
    m["relError"] = max(m["relError"])
    m["envelope"] = [float(f"{v:.2e}") for v in m["envelope"]]
    return m


which = sys.argv[1:] or ["c3", "c5"]          # e.g. `configs_parity.py c5:20` for one case
jobs = []
for w in which:
    name, _, k = w.partition(":")
    fn, dflt = {"c3": (c3, (T.C3_ITS,)), "c5": (T.c5_numbers, (8, 20))}[name]
    jobs.append((name, fn, (int(k),) if k else dflt))
for name, fn, its in jobs:
    for k in its:
        m = fn(k)
        m["iterates_max"] = max(m["iterates"])
        m["iterates"] = [float(f"{v:.2e}") for v in m["iterates"]]
        print(name, k, json.dumps(m), flush=True)
