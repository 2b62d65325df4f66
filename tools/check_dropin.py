#!/usr/bin/env python
"""Container-only check of INTEGRATION.md §1: the UNMODIFIED reference solvers driven by trips_py_amd operators.

TEST TOOLING, NOT PRODUCT.  Runs only where /root/reference exists (the build container; no GPU there), so the
operators are `trips_py_amd.operators.LinearOperator` subclasses whose `_apply` is served by the CPU test engine
(tests/cpu_engine.py over the oracle) — what is exercised is the operator SURFACE the reference's solvers touch
(`shape`, `@`, `*`, `.T`, operands (n,), (n,1), (n,k), NumPy in -> NumPy float64 out, `to_pylops()`), not the kernels
(those are covered by `-m gpu` tests against the same goldens).  Nothing from here travels to the GPU box.

Two routes are checked against the goldens the reference itself produced (tests/golden/*.npz):
  raw      the engine operator handed over as it is (duck typing)
  pylops   `op.to_pylops()` = pylops.FunctionOperator(op.matvec, op.rmatvec, nr, nc), the very wrapper the reference's
           own test problems build (Deblurring2D.py:72, io.py:400)

    python tools/check_dropin.py            # prints one line per case, exits non-zero on a mismatch
"""
import contextlib
import io
import os
import sys
import traceback

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REF = os.environ.get("TRIPS_REFERENCE", "/root/reference")
if not os.path.isdir(REF):
    print("check_dropin: no reference tree at", REF, "- nothing to check here")
    sys.exit(0)
sys.path.insert(0, os.path.join(HERE, "oracle_shim"))
sys.path.insert(0, REF)
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))

import numpy as np  # noqa: E402

np.int0 = np.intp

from trips.solvers.CGLS import CGLS  # noqa: E402      (the reference's own, unmodified)
from trips.solvers.GKS import GKS  # noqa: E402
from trips.solvers.MMGKS import MMGKS  # noqa: E402
from trips.solvers.Hybrid_LSQR import Hybrid_LSQR  # noqa: E402
from trips.solvers.Hybrid_GMRES import Hybrid_GMRES  # noqa: E402

from cpu_engine import CpuEngine, OracleOp  # noqa: E402
from oracle import cpu_ref as O  # noqa: E402

GOLD = os.path.join(REPO, "tests", "golden")
eng = CpuEngine()


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        return fn(*a, **k)


def relerr(a, b):
    a, b = np.asarray(a, dtype=float).reshape(-1), np.asarray(b, dtype=float).reshape(-1)
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


def golden(name):
    return np.load(os.path.join(GOLD, name + ".npz"))


def blur_op(g):
    return OracleOp(O.Blur2D(g["psf"], int(g["N"]), int(g["N"])), eng)


def deriv_op(N):
    return OracleOp(O.FirstDerivative2D(N), eng)


def routes(op):
    return (("raw", op), ("pylops", op.to_pylops()))


results = []


def case(name, fn, tol):
    try:
        err = fn()
        ok = err < tol
        results.append((name, ok, f"rel.err {err:.2e} (bar {tol:g})"))
    except Exception as exc:        # noqa: BLE001
        tb = traceback.extract_tb(exc.__traceback__)
        where = next((f"{os.path.relpath(t.filename, REF)}:{t.lineno}" for t in reversed(tb) if t.filename.startswith(REF)), "?")
        results.append((name, False, f"{type(exc).__name__}: {str(exc)[:120]} (reference frame {where})"))


# engine storage is fp32, the goldens float64: 1e-5 is north_star's bar at fixed lambda
g = golden("cgls_blur64_x0zero")
for tag, A in routes(blur_op(g)):
    case(f"CGLS            A={tag}", lambda A=A: relerr(quiet(CGLS, A, g["b"], np.zeros_like(g["x_true"]), int(g["max_iter"]), 0,
                                                             x_true=g["x_true"])[0], g["x"]), 1e-5)
g1 = golden("hybrid_lsqr_blur32_lam1e-2")
for tag, A in routes(blur_op(g1)):
    case(f"Hybrid_LSQR     A={tag}", lambda A=A: relerr(quiet(Hybrid_LSQR, A, g1["b"], int(g1["n_iter"]), 1e-2, g1["x_true"])[0],
                                                             g1["x"]), 1e-5)
g1g = golden("hybrid_lsqr_blur32_gcv")
for tag, A in routes(blur_op(g1g)):
    case(f"Hybrid_LSQR gcv A={tag}", lambda A=A: relerr(quiet(Hybrid_LSQR, A, g1g["b"], int(g1g["n_iter"]), "gcv", g1g["x_true"])[0],
                                                             g1g["x"]), 1e-3)
g2 = golden("hybrid_gmres_blur32_lam1e-2")
for tag, A in routes(blur_op(g2)):
    case(f"Hybrid_GMRES    A={tag}", lambda A=A: relerr(quiet(Hybrid_GMRES, A, g2["b"], int(g2["n_iter"]), 1e-2, g2["x_true"])[0],
                                                             g2["x"]), 1e-5)
g3 = golden("gks_blur32_lam1e-2")
for ta, A in routes(blur_op(g3)):
    for tl, L in routes(deriv_op(int(g3["N"]))):
        case(f"GKS             A={ta} L={tl}", lambda A=A, L=L: relerr(quiet(GKS, A, g3["b"], L, 3, int(g3["n_iter"]), 1e-2,
                                                                             g3["x_true"])[0], g3["x"]), 1e-5)
g3g = golden("gks_blur32_gcv")
for ta, A in routes(blur_op(g3g)):
    L = deriv_op(int(g3g["N"])).to_pylops()
    case(f"GKS gcv         A={ta} L=pylops", lambda A=A, L=L: relerr(quiet(GKS, A, g3g["b"], L, 3, int(g3g["n_iter"]), "gcv",
                                                                           g3g["x_true"])[0], g3g["x"]), 5e-3)
g4 = golden("mmgks_blur32_p2q1_lam1e-2")
for ta, A in routes(blur_op(g4)):
    for tl, L in routes(deriv_op(int(g4["N"]))):
        case(f"MMGKS           A={ta} L={tl}", lambda A=A, L=L: relerr(quiet(MMGKS, A, g4["b"], L, 2, 1, 3, int(g4["n_iter"]), 1e-2,
                                                                             g4["x_true"])[0], g4["x"]), 1e-5)
# a SQUARE regulariser handed over raw: utils.is_identity (utils.py:55) would build an n x n eye for anything that is not a
# pylops LinearOperator; the engine's operators refuse to be taken for an array there (see LinearOperator.__array__)
sq = OracleOp(O.MatrixOp(O.old_time_derivative_operator(32, 32, 1) + O.sp.identity(1024)), eng)
case("GKS square L    L=pylops", lambda: float(not np.all(np.isfinite(quiet(GKS, blur_op(g3), g3["b"], sq.to_pylops(), 3, 3, 1e-2)[0]))), 0.5)


def raw_square_refused():
    try:
        quiet(GKS, blur_op(g3), g3["b"], sq, 3, 3, 1e-2)
    except TypeError as exc:
        return 0.0 if "to_pylops" in str(exc) else 1.0
    return 1.0


case("GKS square L    L=raw -> TypeError naming to_pylops()", raw_square_refused, 0.5)

bad = 0
for name, ok, msg in results:
    print(f"{'ok  ' if ok else 'FAIL'}  {name:54s} {msg}")
    bad += not ok
print(f"{len(results) - bad}/{len(results)} drop-in cases agree with the reference's goldens")
sys.exit(1 if bad else 0)
