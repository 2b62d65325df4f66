#!/usr/bin/env python
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE in this container.

TEST TOOLING, NOT PRODUCT.  Run only where /root/reference exists (the build
container):

    python tools/make_goldens.py

It puts tools/oracle_shim (our stand-ins for the absent pylops / astra / h5py /
resizeimage packages) and /root/reference on sys.path, imports the reference's
own modules, runs them on small seeded inputs and stores plain arrays / scalars
(inputs and expected outputs) as .npz.  No reference source, bytecode or pickled
reference object is written anywhere — only numbers.

Fixture families:
  G1 blur2d_*            Deblurring2D.Gauss + forward_Op  (Deblurring2D.py:48-73), fwd & bwd
  G2 cgls_*              CGLS                              (CGLS.py:16-86)
  G3 gk_update, arnoldi_update, golub_kahan, arnoldi      (decompositions.py:20-255)
  G4 hybrid_lsqr_*, hybrid_gmres_*                        (Hybrid_LSQR.py:25, Hybrid_GMRES.py:23)
  G5 gks_*, mmgks_*                                       (GKS.py:27, MMGKS.py:28)
  G5b mmgks_*_isotv_*, isotv_weights                     (MMGKS.py:61-77 over the shim's FirstDerivative)
  G4c/G5c *_lcurve       regparam='l_curve' through the four projection solvers (Hybrid_LSQR.py:94-98, Hybrid_GMRES.py:67-71,
                         GKS.py:67-68, MMGKS.py:100-101)
  G5d *_framelet_*       GKS / MMGKS with L = create_framelet_operator(32, 32, 2) (operators.py:50-113; the large-scale demos' regulariser)
  G6 deriv_ops                                            (operators.py:24-45)
  G7 regparam_fn                                          (gcv.py, discrepancy_principle.py, l_curve.py)
  G9 sparse_dynamic_*    generate_crossPhantom on a synthetic stand-in file + CGLS / Hybrid_LSQR / GKS / MMGKS on its outputs (io.py:187-229)
  G8 deblur1d_cgls                                        (Deblurring1D.py + CGLS: BASELINE config C1)
"""
import io
import os
import sys
import contextlib

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REF = os.environ.get("TRIPS_REFERENCE", "/root/reference")
sys.path.insert(0, os.path.join(HERE, "oracle_shim"))
sys.path.insert(0, REF)

import numpy as np

np.int0 = np.intp  # removed in NumPy 2; Deblurring1D.py:68,72 still uses it
import scipy.sparse as _sps  # noqa: E402
for _cls in (_sps.csr_matrix, _sps.csc_matrix, _sps.coo_matrix, _sps.lil_matrix):
    if not hasattr(_cls, "H"):   # removed from SciPy sparse matrices; create_framelet_operator (operators.py:107,109) uses it
        _cls.H = property(lambda self: self.conj().T)

from scipy.ndimage import convolve  # noqa: E402
from trips.solvers.CGLS import CGLS  # noqa: E402
from trips.solvers.GKS import GKS  # noqa: E402
from trips.solvers.MMGKS import MMGKS  # noqa: E402
from trips.solvers.Hybrid_LSQR import Hybrid_LSQR  # noqa: E402
from trips.solvers.Hybrid_GMRES import Hybrid_GMRES  # noqa: E402
from trips.solvers.GK_Tikhonov import Golub_Kahan_Tikhonov  # noqa: E402
from trips.solvers.A_Tikhonov import Arnoldi_Tikhonov  # noqa: E402
from trips.solvers.GMRES import GMRES  # noqa: E402
from trips.test_problems.Deblurring2D import Deblurring2D  # noqa: E402
from trips.test_problems.Deblurring1D import Deblurring1D  # noqa: E402
from trips.utilities import decompositions as dec  # noqa: E402
from trips.utilities import operators as refops  # noqa: E402
from trips.utilities.weights import smoothed_holder_weights  # noqa: E402
from trips.utilities.reg_param.gcv import generalized_crossvalidation, gcv_numerator, gcv_denominator  # noqa: E402
from trips.utilities.reg_param.discrepancy_principle import discrepancy_principle  # noqa: E402
from trips.utilities.reg_param.l_curve import l_curve, curvature  # noqa: E402
import scipy.linalg as la  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")
os.makedirs(OUT, exist_ok=True)


def quiet(fn, *a, **k):
    """tqdm / print noise of the reference goes nowhere."""
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        return fn(*a, **k)


def save(name, **arrays):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrays.items()})
    print(f"  {name}.npz  {os.path.getsize(path) / 1024:.1f} KiB")


def test_image(nx, ny, seed):
    """Piecewise-constant rectangles + smooth bump + a little texture (seeded)."""
    rng = np.random.default_rng(seed)
    img = np.zeros((nx, ny))
    for _ in range(4):
        i0, j0 = rng.integers(0, nx - 2), rng.integers(0, ny - 2)
        h, w = rng.integers(2, max(3, nx // 2)), rng.integers(2, max(3, ny // 2))
        img[i0:i0 + h, j0:j0 + w] += rng.uniform(0.3, 1.0)
    ii, jj = np.meshgrid(np.arange(nx), np.arange(ny), indexing="ij")
    img += 0.5 * np.exp(-((ii - nx / 3) ** 2 + (jj - ny / 2) ** 2) / (0.05 * nx * ny))
    img += 0.1 * rng.random((nx, ny))
    return img


def blur_problem(N, seed, noise=0.01, dim=(9, 9), spread=(3, 3)):
    D = Deblurring2D(CommitCrime=True)
    A = D.forward_Op(dim, spread, N, N)
    PSF, center = D.Gauss(dim, spread)
    x_true = test_image(N, N, seed).reshape(-1, 1)
    b_true = A @ x_true
    rng = np.random.default_rng(seed + 1000)
    e = rng.standard_normal(b_true.shape)
    e *= noise * np.linalg.norm(b_true) / np.linalg.norm(e)   # recipe of Deblurring2D.py:142-146, seeded
    b = b_true + e
    return A, PSF, x_true, b, float(np.linalg.norm(e))


# ----------------------------------------------------------------------------------------- G1
def g1_blur():
    print("G1 blur")
    cases = [
        ("g9x9_s3_32x32", (9, 9), (3, 3), 32, 32),
        ("g5x7_s1-2_24x40", (5, 7), (1, 2), 24, 40),
        ("g10x10_s2_16x16", (10, 10), (2, 2), 16, 16),
        ("g9x9_s3_64x64", (9, 9), (3, 3), 64, 64),
        ("g3x3_s1_7x5", (3, 3), (1, 1), 7, 5),        # image smaller than tile, odd sizes
        ("g9x9_s3_6x6", (9, 9), (3, 3), 6, 6),        # PSF larger than image: repeated reflection
    ]
    for name, dim, spread, nx, ny in cases:
        D = Deblurring2D(CommitCrime=True)
        A = D.forward_Op(dim, spread, nx, ny)
        PSF, center = D.Gauss(dim, spread)
        x = test_image(nx, ny, 7).reshape(-1)
        y = test_image(nx, ny, 8).reshape(-1)
        fwd = np.asarray(A @ x.reshape(-1, 1)).reshape(-1)
        bwd = np.asarray(A.T @ y.reshape(-1, 1)).reshape(-1)
        # multi-column operand (GKS.py:37 `A@V`)
        X3 = np.stack([x, y, x - y], axis=1)
        fwd3 = np.asarray(A @ X3)
        save("blur2d_" + name, psf=PSF, center=center, nx=nx, ny=ny, x=x, y=y, fwd=fwd, bwd=bwd, X3=X3, fwd3=fwd3)
    # non-separable, asymmetric PSF through the reference's exact scipy call (Deblurring2D.py:70-71)
    rng = np.random.default_rng(3)
    for name, kh, kw, nx, ny in [("asym7x5_20x28", 7, 5, 20, 28), ("asym4x6_33x17", 4, 6, 33, 17)]:
        PSF = rng.random((kh, kw))
        PSF /= PSF.sum()
        x = test_image(nx, ny, 9)
        y = test_image(nx, ny, 10)
        fwd = convolve(x.reshape([nx, ny]), PSF, mode="reflect").reshape(-1)
        bwd = convolve(y.reshape([nx, ny]), np.flipud(np.fliplr(PSF)), mode="reflect").reshape(-1)
        save("blur2d_" + name, psf=PSF, nx=nx, ny=ny, x=x.reshape(-1), y=y.reshape(-1), fwd=fwd, bwd=bwd)


# ----------------------------------------------------------------------------------------- G2
def g2_cgls():
    print("G2 CGLS")
    N = 64
    A, PSF, x_true, b, delta = blur_problem(N, 11)
    n = N * N
    x, info = quiet(CGLS, A, b, np.zeros((n, 1)), 20, 0, x_true=x_true)
    save("cgls_blur64_x0zero", psf=PSF, N=N, b=b, x_true=x_true, x0=np.zeros((n, 1)), max_iter=20, tol=0.0,
         x=x, relResidual=info["relResidual"], relError=info["relError"], its=info["its"],
         x_it1=info["xHistory"][0], x_it10=info["xHistory"][9])
    x0 = np.asarray(A.T @ b).reshape(-1, 1)
    x, info = quiet(CGLS, A, b, x0, 20, 0, x_true=x_true)
    save("cgls_blur64_x0ATb", psf=PSF, N=N, b=b, x_true=x_true, x0=x0, max_iter=20, tol=0.0,
         x=x, relResidual=info["relResidual"], relError=info["relError"], its=info["its"])
    # early stop through tol (CGLS.py:73-75)
    for tol in (2e-2, 1e-2, 5e-3, 2e-3, 1e-3):
        x, info = quiet(CGLS, A, b, np.zeros((n, 1)), 50, tol)
        if 5 <= info["its"] < 40:
            break
    save("cgls_blur64_tol", psf=PSF, N=N, b=b, x0=np.zeros((n, 1)), max_iter=50, tol=tol,
         x=x, relResidual=info["relResidual"], its=info["its"])


# ----------------------------------------------------------------------------------------- G3
def g3_decomp():
    print("G3 decompositions")
    N = 32
    A, PSF, x_true, b, delta = blur_problem(N, 21)
    # golub_kahan_update x10 exactly as Hybrid_LSQR drives it (Hybrid_LSQR.py:63-74)
    beta = np.linalg.norm(b)
    U = b.reshape((-1, 1)) / beta
    B = np.empty(1)
    V = np.empty((N * N, 1))
    for _ in range(10):
        U, B, V = dec.golub_kahan_update(A, U, B, V)
    save("gk_update_blur32", psf=PSF, N=N, b=b, steps=10, U=U, B=B, V=V)
    # arnoldi_update x10 as Hybrid_GMRES drives it (Hybrid_GMRES.py:41-47)
    Vq = b.reshape((-1, 1)) / beta
    H = np.empty(1)
    for _ in range(10):
        Vq, H = dec.arnoldi_update(A, Vq, H)
    save("arnoldi_update_blur32", psf=PSF, N=N, b=b, steps=10, V=Vq, H=H)
    U, S, V = quiet(dec.golub_kahan, A, b, 3)
    save("golub_kahan_blur32_d3", psf=PSF, N=N, b=b, n_iter=3, U=U, S=S, V=V)
    U, S, V = quiet(dec.golub_kahan, A, b, 8)
    save("golub_kahan_blur32_d8", psf=PSF, N=N, b=b, n_iter=8, U=U, S=S, V=V)
    Q, H = quiet(dec.arnoldi, A, b, 6)
    save("arnoldi_blur32_d6", psf=PSF, N=N, b=b, n_iter=6, Q=Q, H=H)
    # discrepancy-principle stopping inside golub_kahan (decompositions.py:167-195): gk_delta chosen so that it stops early
    _, _, _, bn, dn = blur_problem(N, 21, noise=0.1)
    U, S, V = quiet(dec.golub_kahan, A, bn, 12, True, gk_eta=1.001, gk_delta=dn)
    save("golub_kahan_blur32_dpstop", psf=PSF, N=N, b=bn, n_iter=12, gk_eta=1.001, gk_delta=dn, U=U, S=S, V=V)
    # the same switch in arnoldi (decompositions.py:104-112).  Its residual (normal equations of the square block of H, the
    # normalised b) stays between 0.9 and 1 on this problem: gk_delta = 1 halts after the first step, 0.5 never halts
    for tag, gd in (("stop1", 1.0), ("never", 0.5)):
        Q, H = quiet(dec.arnoldi, A, bn, 7, True, gk_eta=1.001, gk_delta=gd)
        save("arnoldi_blur32_dpstop_" + tag, psf=PSF, N=N, b=bn, n_iter=7, gk_eta=1.001, gk_delta=gd, Q=Q, H=H)
    # (Arnoldi_Tikhonov(dp_stop=True) cannot be pinned: A_Tikhonov.py:70 passes dp_stop both positionally and inside
    #  **kwargs and raises TypeError)


# ----------------------------------------------------------------------------------------- G4
def g4_hybrid():
    print("G4 hybrid")
    N = 32
    A, PSF, x_true, b0, delta0 = blur_problem(N, 31)
    _, _, _, b1, delta1 = blur_problem(N, 31, noise=0.1)
    for tag, rp, kw in [("lam1e-2", 1e-2, {}), ("gcv", "gcv", {}), ("dp", "dp", {"delta": delta1})]:
        b, delta = (b1, delta1) if tag == "dp" else (b0, delta0)
        x, info = quiet(Hybrid_LSQR, A, b, 12, rp, x_true, **kw)
        save("hybrid_lsqr_blur32_" + tag, psf=PSF, N=N, b=b, x_true=x_true, n_iter=12, delta=delta,
             x=x, regParam=info["regParam"], regParam_history=np.array(info["regParam_history"], dtype=float),
             relError=info["relError"], its=info["its"], n_hist=len(info["xHistory"]), x_it1=info["xHistory"][0])
        x, info = quiet(Hybrid_GMRES, A, b, 12, rp, x_true, **kw)
        save("hybrid_gmres_blur32_" + tag, psf=PSF, N=N, b=b, x_true=x_true, n_iter=12, delta=delta,
             x=x, regParam=info["regParam"], regParam_history=np.array(info["regParam_history"], dtype=float),
             relError=info["relError"], relResidual=info["relResidual"], its=info["its"],
             n_hist=len(info["xHistory"]), x_it1=info["xHistory"][0])


# ----------------------------------------------------------------------------------------- G4b
def g4b_oneshot():
    print("G4b one-shot solvers")
    N = 32
    A, PSF, x_true, b, delta = blur_problem(N, 61)
    _, _, _, b1, delta1 = blur_problem(N, 61, noise=0.2)
    out = {"psf": PSF, "N": N, "b": b, "b_dp": b1, "delta_dp": delta1}
    for tag, rp, kw, bb in [("lam", 1e-2, {}, b), ("gcv", "gcv", {}, b), ("dp", "dp", {"delta": delta1}, b1)]:
        x, lam = quiet(Golub_Kahan_Tikhonov, A, bb, 3, rp, **kw)
        out[f"gkt_{tag}_x"], out[f"gkt_{tag}_lam"] = x, lam
        x, lam = quiet(Arnoldi_Tikhonov, A, bb, 6, rp, **kw)
        out[f"at_{tag}_x"], out[f"at_{tag}_lam"] = x, lam
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        out["gmres_x"] = quiet(GMRES, A, b, 5)
    save("oneshot_blur32", **out)


# ----------------------------------------------------------------------------------------- G5
def g5_gks():
    print("G5 GKS / MMGKS")
    N = 32
    A, PSF, x_true, b0, delta0 = blur_problem(N, 41)
    _, _, _, b1, delta1 = blur_problem(N, 41, noise=0.1)
    L = refops.gen_first_derivative_operator_2D(N, N)
    for tag, rp, kw in [("lam1e-2", 1e-2, {}), ("gcv", "gcv", {}), ("dp", "dp", {"delta": delta1})]:
        b, delta = (b1, delta1) if tag == "dp" else (b0, delta0)
        x, info = quiet(GKS, A, b, L, 3, 10, rp, x_true, **kw)
        save("gks_blur32_" + tag, psf=PSF, N=N, b=b, x_true=x_true, projection_dim=3, n_iter=10, delta=delta,
             x=x, regParam=info["regParam"], regParam_history=np.array(info["regParam_history"], dtype=float),
             relError=info["relError"], Residual=info["Residual"], its=info["its"], x_it1=info["xHistory"][0])
    for tag, p, q, rp, kw in [("p2q1_lam1e-2", 2, 1, 1e-2, {}),
                              ("p2q1_gcv", 2, 1, "gcv", {}),
                              ("p1q1_lam1e-2", 1, 1, 1e-2, {}),
                              ("p2q0.5_eps0.01_lam1e-3", 2, 0.5, 1e-3, {"epsilon": 0.01})]:
        b, delta = b0, delta0
        x, info = quiet(MMGKS, A, b, L, p, q, 3, 10, rp, x_true, **kw)
        save("mmgks_blur32_" + tag, psf=PSF, N=N, b=b, x_true=x_true, pnorm=p, qnorm=q, projection_dim=3, n_iter=10,
             epsilon=kw.get("epsilon", 0.1),
             x=x, regParam=info["regParam"], regParam_history=np.array(info["regParam_history"], dtype=float),
             relError=info["relError"], Residual=info["Residual"], its=info["its"], x_it1=info["xHistory"][0])
    # a dynamic (frame-major, block-diagonal) problem with the space-time derivative (config C5 shape, tiny)
    nt, Nf = 3, 16
    Ds = [Deblurring2D(CommitCrime=True) for _ in range(nt)]
    spreads = [(1.0, 1.0), (1.5, 1.0), (2.0, 2.0)]
    ops = [Ds[t].forward_Op((5, 5), spreads[t], Nf, Nf) for t in range(nt)]
    psfs = np.stack([Ds[t].Gauss((5, 5), spreads[t])[0] for t in range(nt)])
    import pylops
    F = pylops.BlockDiag(ops)
    xt = np.concatenate([test_image(Nf, Nf, 50 + t).reshape(-1) for t in range(nt)]).reshape(-1, 1)
    bt = np.asarray(F @ xt).reshape(-1, 1)
    rng = np.random.default_rng(77)
    e = rng.standard_normal(bt.shape)
    bt = bt + 0.01 * np.linalg.norm(bt) / np.linalg.norm(e) * e
    Lst = refops.gen_spacetime_derivative_operator(Nf, Nf, nt)
    x, info = quiet(GKS, F, bt, Lst, 3, 8, 1e-2, xt)
    save("gks_dyn3x16_lam1e-2", psfs=psfs, N=Nf, nt=nt, b=bt, x_true=xt, projection_dim=3, n_iter=8,
         x=x, regParam_history=np.array(info["regParam_history"], dtype=float),
         relError=info["relError"], Residual=info["Residual"], its=info["its"])
    x, info = quiet(MMGKS, F, bt, Lst, 2, 1, 3, 8, 1e-2, xt)
    save("mmgks_dyn3x16_p2q1_lam1e-2", psfs=psfs, N=Nf, nt=nt, b=bt, x_true=xt, pnorm=2, qnorm=1, projection_dim=3,
         n_iter=8, epsilon=0.1, x=x, regParam_history=np.array(info["regParam_history"], dtype=float),
         relError=info["relError"], Residual=info["Residual"], its=info["its"])
    # group-sparsity weights branch (MMGKS.py:45-52,78-91): L is replaced by kron(I_nt, old first-derivative matrix)
    for tag, q, rp in (("q1_lam1e-2", 1, 1e-2), ("q0.5_lam1e-3", 0.5, 1e-3), ("q1_gcv", 1, "gcv")):
        x, info = quiet(MMGKS, F, bt, Lst, 2, q, 3, 8, rp, xt, GS="GS", prob_dims=(Nf, Nf, nt))
        save("mmgks_dyn3x16_gs_" + tag, psfs=psfs, N=Nf, nt=nt, b=bt, x_true=xt, pnorm=2, qnorm=q, projection_dim=3,
             n_iter=8, epsilon=0.1, x=x, regParam_history=np.array(info["regParam_history"], dtype=float),
             relError=info["relError"], Residual=info["Residual"], its=info["its"])


# ----------------------------------------------------------------------------------------- G4c / G5c / G5d
def g45_lcurve():
    """regparam = 'l_curve' end to end: Hybrid_LSQR.py:94-98, Hybrid_GMRES.py:67-71, GKS.py:67-68, MMGKS.py:100-101."""
    print("G4c/G5c l_curve through the solvers")
    N = 32
    A, PSF, x_true, b, delta = blur_problem(N, 31)
    for name, fn in (("hybrid_lsqr", Hybrid_LSQR), ("hybrid_gmres", Hybrid_GMRES)):
        x, info = quiet(fn, A, b, 12, "l_curve", x_true)
        save(name + "_blur32_lcurve", psf=PSF, N=N, b=b, x_true=x_true, n_iter=12, delta=delta,
             x=x, regParam=info["regParam"], regParam_history=np.array(info["regParam_history"], dtype=float),
             relError=info["relError"], relResidual=np.array(info["relResidual"], dtype=float), its=info["its"],
             n_hist=len(info["xHistory"]), x_it1=info["xHistory"][0], x_it2=info["xHistory"][1])
    A, PSF, x_true, b, delta = blur_problem(N, 41)
    L = refops.gen_first_derivative_operator_2D(N, N)
    x, info = quiet(GKS, A, b, L, 3, 10, "l_curve", x_true)
    save("gks_blur32_lcurve", psf=PSF, N=N, b=b, x_true=x_true, projection_dim=3, n_iter=10, delta=delta,
         x=x, regParam=info["regParam"], regParam_history=np.array(info["regParam_history"], dtype=float),
         relError=info["relError"], Residual=info["Residual"], its=info["its"], x_it1=info["xHistory"][0])
    x, info = quiet(MMGKS, A, b, L, 2, 1, 3, 10, "l_curve", x_true)
    save("mmgks_blur32_p2q1_lcurve", psf=PSF, N=N, b=b, x_true=x_true, pnorm=2, qnorm=1, projection_dim=3, n_iter=10,
         epsilon=0.1, x=x, regParam=info["regParam"], regParam_history=np.array(info["regParam_history"], dtype=float),
         relError=info["relError"], Residual=info["Residual"], its=info["its"], x_it1=info["xHistory"][0])


def g5d_framelet():
    """GKS / MMGKS with the framelet analysis operator as regulariser — what demos/demo_2D_Deblurring_large_scale.ipynb:403 and
    demo_Tomo_large_scale.ipynb:687 run (operators.py:50-113)."""
    print("G5d GKS / MMGKS with the framelet regulariser")
    N = 32
    A, PSF, x_true, b, delta = blur_problem(N, 41)
    W = refops.create_framelet_operator(N, N, 2)
    for tag, rp in (("lam1e-2", 1e-2), ("gcv", "gcv")):
        x, info = quiet(GKS, A, b, W, 3, 10, rp, x_true)
        save("gks_blur32_framelet_" + tag, psf=PSF, N=N, level=2, b=b, x_true=x_true, projection_dim=3, n_iter=10, delta=delta,
             x=x, regParam=info["regParam"], regParam_history=np.array(info["regParam_history"], dtype=float),
             relError=info["relError"], Residual=info["Residual"], its=info["its"], x_it1=info["xHistory"][0])
        x, info = quiet(MMGKS, A, b, W, 2, 1, 3, 10, rp, x_true)
        save("mmgks_blur32_framelet_p2q1_" + tag, psf=PSF, N=N, level=2, b=b, x_true=x_true, pnorm=2, qnorm=1, projection_dim=3,
             n_iter=10, epsilon=0.1, x=x, regParam=info["regParam"],
             regParam_history=np.array(info["regParam_history"], dtype=float),
             relError=info["relError"], Residual=info["Residual"], its=info["its"], x_it1=info["xHistory"][0])


# ----------------------------------------------------------------------------------------- G5b
def g5b_isotv():
    """MMGKS isoTV branch (MMGKS.py:61-77) with the operators_old.py PyLops-built regulariser.  NOTE: PyLops is absent, so
    `pylops.FirstDerivative / Kronecker / VStack` here are tools/oracle_shim's restatement of their published semantics:
    these fixtures pin MMGKS.py's and operators_old.py's own logic, not PyLops' arithmetic (parity unpinned there)."""
    print("G5b MMGKS isoTV (reference code over the shim's FirstDerivative)")
    import pylops
    from trips.utilities import operators_old as oo
    nt, Nf = 3, 16
    Ds = [Deblurring2D(CommitCrime=True) for _ in range(nt)]
    spreads = [(1.0, 1.0), (1.5, 1.0), (2.0, 2.0)]
    ops = [Ds[t].forward_Op((5, 5), spreads[t], Nf, Nf) for t in range(nt)]
    psfs = np.stack([Ds[t].Gauss((5, 5), spreads[t])[0] for t in range(nt)])
    F = pylops.BlockDiag(ops)
    xt = np.concatenate([test_image(Nf, Nf, 50 + t).reshape(-1) for t in range(nt)]).reshape(-1, 1)
    bt = np.asarray(F @ xt).reshape(-1, 1)
    rng = np.random.default_rng(77)
    e = rng.standard_normal(bt.shape)
    bt = bt + 0.01 * np.linalg.norm(bt) / np.linalg.norm(e) * e
    L = pylops.VStack((oo.spatial_derivative_operator(Nf, Nf, nt), oo.time_derivative_operator(Nf, Nf, nt)))
    Ld = _sps.csr_matrix(np.asarray(L.todense(), dtype=np.float64))
    for tag, q, rp in (("q1_lam1e-2", 1, 1e-2), ("q0.5_lam1e-3", 0.5, 1e-3), ("q1_gcv", 1, "gcv")):
        x, info = quiet(MMGKS, F, bt, L, 2, q, 3, 8, rp, xt, isoTV="isoTV", prob_dims=(Nf, Nf, nt))
        save("mmgks_dyn3x16_isotv_" + tag, psfs=psfs, N=Nf, nt=nt, b=bt, x_true=xt, pnorm=2, qnorm=q, projection_dim=3,
             n_iter=8, epsilon=0.1, x=x, regParam_history=np.array(info["regParam_history"], dtype=float),
             relError=info["relError"], Residual=info["Residual"], its=info["its"], x_it1=info["xHistory"][0],
             L_data=Ld.data, L_indices=Ld.indices, L_indptr=Ld.indptr, L_shape=np.array(Ld.shape),
             pylops_first_derivative_is_shim=1)
    # the weights function on its own (weights.py:29-40)
    from trips.utilities.weights import iso_TV_weights
    xs = rng.standard_normal((Nf * Nf * nt, 1))
    us = np.asarray(L @ xs).reshape(-1, 1)
    save("isotv_weights_16x3", x=xs, u=us, nx=Nf, ny=Nf, eps=0.1, q=1.0,
         wr=iso_TV_weights(xs, us, Nf, Nf, 0.1, 1.0), pylops_first_derivative_is_shim=1)


# ----------------------------------------------------------------------------------------- G6
def g6_derivs():
    print("G6 derivative operators")
    out = {}
    for n in (4, 5):
        out[f"D1_{n}"] = refops.gen_first_derivative_operator(n).toarray()
        out[f"D2_{n}"] = refops.gen_first_derivative_operator_2D(n, n).toarray()
    out["Dst_4_3"] = refops.gen_spacetime_derivative_operator(4, 4, 3).toarray()
    out["Dst_3_2"] = refops.gen_spacetime_derivative_operator(3, 3, 2).toarray()
    rng = np.random.default_rng(5)
    u = rng.standard_normal(50)
    out["holder_u"] = u
    out["holder_eps0.1_p1"] = smoothed_holder_weights(u, epsilon=0.1, p=1)
    out["holder_eps0.01_p0.5"] = smoothed_holder_weights(u, epsilon=0.01, p=0.5)
    out["holder_eps0.1_p2"] = smoothed_holder_weights(u, epsilon=0.1, p=2)
    save("deriv_ops", **out)
    # framelet analysis operator (operators.py:50-113): action on seeded vectors + a small dense matrix
    fr = {}
    for (n, m, l) in ((8, 6, 2), (12, 12, 1), (16, 10, 3)):
        Wf = refops.create_framelet_operator(n, m, l)
        rng2 = np.random.default_rng(n * 100 + m * 10 + l)
        x = rng2.standard_normal(n * m)
        y = rng2.standard_normal(Wf.shape[0])
        fr[f"x_{n}_{m}_{l}"], fr[f"y_{n}_{m}_{l}"] = x, y
        fr[f"Wx_{n}_{m}_{l}"] = np.asarray(Wf @ x).reshape(-1)
        fr[f"WTy_{n}_{m}_{l}"] = np.asarray(Wf.T @ y).reshape(-1)
    fr["dense_8_6_2"] = np.asarray(refops.create_framelet_operator(8, 6, 2).todense())
    save("framelet_ops", **fr)


# ----------------------------------------------------------------------------------------- G7
def g7_regparam():
    print("G7 reg-param functions")
    rng = np.random.default_rng(6)
    m, k = 60, 6
    AV = rng.standard_normal((m, k)) @ np.diag(np.logspace(0, -3, k))
    LV = rng.standard_normal((80, k))
    b = AV @ rng.standard_normal((k, 1)) + 0.05 * rng.standard_normal((m, 1))
    Q_A, R_A = la.qr(AV, mode="economic")
    _, R_L = la.qr(LV, mode="economic")
    lams = np.array([1e-8, 1e-4, 1e-2, 1.0, 50.0])
    num = np.array([gcv_numerator(l, Q_A, R_A, R_L, b) for l in lams])
    den = np.array([gcv_denominator(l, R_A, R_L, b) for l in lams])
    lam_gcv = generalized_crossvalidation(Q_A, R_A, R_L, b)
    delta = 0.05 * np.sqrt(m)
    lam_dp = discrepancy_principle(Q_A, R_A, R_L, b, delta=float(delta))
    lam_dp_eta = discrepancy_principle(Q_A, R_A, R_L, b, delta=float(delta), eta=1.2)
    curv = np.array([curvature(l, R_A, R_L, Q_A.T @ b) for l in lams])
    lam_lc = l_curve(R_A, R_L, Q_A.T @ b)
    # hybrid flavour: bidiagonal B, 'modified' GCV with fullsize (Hybrid_LSQR.py:81-84)
    kk = 7
    B = np.zeros((kk + 1, kk))
    B[np.arange(kk), np.arange(kk)] = np.logspace(0, -2, kk)
    B[np.arange(1, kk + 1), np.arange(kk)] = 0.5 * np.logspace(0, -2, kk)
    bhat = np.zeros(kk + 1)
    bhat[0] = 3.7
    Qb, sb, _ = la.svd(B, full_matrices=False)
    Rb = np.diag(sb)
    num_mod = np.array([gcv_numerator(l, Qb, Rb, np.eye(kk), bhat, variant="modified") for l in lams])
    den_mod = np.array([gcv_denominator(l, Rb, np.eye(kk), bhat, variant="modified", fullsize=500) for l in lams])
    lam_gcv_mod = generalized_crossvalidation(Qb, Rb, np.eye(kk), bhat, variant="modified", fullsize=500)
    save("regparam_fn", AV=AV, LV=LV, b=b, Q_A=Q_A, R_A=R_A, R_L=R_L, lams=lams, gcv_num=num, gcv_den=den,
         lam_gcv=lam_gcv, delta=delta, lam_dp=lam_dp, lam_dp_eta12=lam_dp_eta, curvature=curv, lam_lcurve=lam_lc,
         B=B, bhat=bhat, gcv_num_mod=num_mod, gcv_den_mod=den_mod, lam_gcv_mod=lam_gcv_mod, fullsize=500)


def g7b_dp_corners():
    """The discrepancy principle's corner branches (discrepancy_principle.py:45-66 and :100-129): a projected regulariser with
    fewer rows than columns; the 'tsvd' / 'tgsvd' truncation indices of the direct solvers.  (A regulariser with an exactly zero
    singular value ends in numpy's LinAlgError in the reference itself — the tests expect the same.)"""
    print("G7b discrepancy principle: wide L, tsvd / tgsvd")
    rng = np.random.default_rng(12)
    m, k = 40, 6
    A = rng.standard_normal((m, k)) @ np.diag(np.logspace(0, -2, k))
    Q, R = la.qr(A, mode="economic")
    b = A @ rng.standard_normal((k, 1)) + 0.05 * rng.standard_normal((m, 1))
    delta = float(0.05 * np.sqrt(m))
    Lw = rng.standard_normal((4, k))
    lam_wide = discrepancy_principle(Q, R, Lw, b, delta=delta)
    lam_wide_eta = discrepancy_principle(Q, R, Lw, b, delta=delta, eta=1.3)
    U, _, _ = la.svd(A)
    out = {}
    for dpt in ("tsvd", "tgsvd"):
        for tag, dl in (("", delta), ("_big", 6.0 * delta), ("_small", 0.05 * delta)):
            out[f"{dpt}{tag}"] = discrepancy_principle(U, A, np.eye(k), b, delta=float(dl), dptype=dpt)
    save("regparam_dp_corners", A=A, Q=Q, R=R, b=b, delta=delta, L_wide=Lw, lam_wide=lam_wide, lam_wide_eta13=lam_wide_eta, U=U,
         **out)


# ----------------------------------------------------------------------------------------- G8
def g8_deblur1d():
    print("G8 1-D deblurring (BASELINE config C1)")
    n = 256
    D1 = Deblurring1D(CommitCrime=True)
    A = D1.forward_Op_1D(parameter=3, nx=n)
    x_true = D1.gen_xtrue(n, "curve0").reshape(-1, 1)
    b_true = np.asarray(A @ x_true).reshape(-1, 1)
    rng = np.random.default_rng(81)
    e = rng.standard_normal(b_true.shape)
    e *= 0.01 * np.linalg.norm(b_true) / np.linalg.norm(e)
    b = b_true + e
    x, info = quiet(CGLS, A, b, np.zeros((n, 1)), 50, 0, x_true=x_true)
    ATb = np.asarray(A.T @ b).reshape(-1)
    save("deblur1d_cgls_n256", psf=D1.PSF, n=n, x_true=x_true, b_true=b_true, b=b, ATb=ATb, max_iter=50, tol=0.0,
         x=x, relResidual=info["relResidual"], relError=info["relError"], its=info["its"])


# ----------------------------------------------------------------------------------------- G9
def sparse_dynamic_vectors(n, m):
    """Deterministic probe vectors (formulas, so that the fixture need not hold them)."""
    return np.sin(0.37 * np.arange(n)) + 0.25 * np.cos(0.011 * np.arange(n)), np.cos(0.53 * np.arange(m)) - 0.1


def g9_sparse_dynamic():
    """The sparse-forward-matrix dynamic path: the reference's OWN loader generate_crossPhantom (io.py:187-229) run on a synthetic
    stand-in for the Zenodo file it downloads (same keys, same layout: A of (mm nn) x (16 128^2), sinogram mm x nn), then its
    solvers on what the loader returns — F = A_small whole (demos/2_demo_dynamic_CrossPhantom.ipynb cell 15, 23) and one frame's
    block Aseq[t] (cell 5).  T = 16, 700 rows and 16384 columns per frame are hard-coded in the loader (:203, :223)."""
    print("G9 sparse dynamic path through generate_crossPhantom")
    import tempfile
    import scipy.io as spio
    from trips.utilities import io as rio
    T, Nf, mm, nn = 16, 128, 140, 240                 # nn / 3 = 80 kept sinogram columns x 140 rays = 11200 = 16 x 700 rows
    npix = Nf * Nf
    rng = np.random.default_rng(91)
    nrows = mm * nn
    rows, cols, vals = [], [], []
    for r in range(nrows):
        col, t = divmod(r, mm)                        # sinogram column, ray
        kept, ii = (col % 3 == 0), col // 3
        f = (ii * mm + t) // 700 if kept else rng.integers(0, T)     # the frame the loader will file this row under
        # a short "ray": 4 pixels of a line through frame f, and one entry OUTSIDE the frame's block (the loader must drop it)
        i0, j0 = rng.integers(8, Nf - 8, size=2)
        di, dj = rng.integers(-2, 3, size=2)
        for k in range(4):
            rows.append(r)
            cols.append(f * npix + (i0 + k * di) * Nf + (j0 + k * dj))
            vals.append(rng.integers(8, 64) / 64.0)
        rows.append(r)
        cols.append(((f + 1 + rng.integers(0, T - 1)) % T) * npix + rng.integers(0, npix))
        vals.append(rng.integers(1, 8) / 64.0)
    A = _sps.csc_matrix((np.array(vals), (np.array(rows), np.array(cols))), shape=(nrows, T * npix))
    xs = []
    for t in range(T):
        img = np.zeros((Nf, Nf))
        img[30 + 2 * t:70 + 2 * t, 20:90] = 1.0
        img[80:110, 10 + 4 * t:40 + 4 * t] = 0.5
        xs.append(img.reshape(-1))
    x_true = np.concatenate(xs)
    sino = np.asarray(A @ x_true).reshape(mm, nn, order="F")
    here = os.getcwd()
    with tempfile.TemporaryDirectory() as d:
        os.makedirs(os.path.join(d, "data", "crossphantom_data"))
        spio.savemat(os.path.join(d, "data", "crossphantom_data", "DataDynamic_128x15.mat"), {"A": A, "sinogram": sino})
        os.chdir(d)
        # (scipy.io.loadmat of SciPy >= 1.15 hands a MATLAB sparse matrix over as COO, which cannot be indexed: the loader's
        #  `A[ind, :]` (:209) was written against the CSC that older SciPy returned — restored here, nothing else touched)
        real_loadmat = rio.spio.loadmat

        def loadmat_csc(*a, **k):
            f = real_loadmat(*a, **k)
            return {key: (_sps.csc_matrix(v) if _sps.issparse(v) else v) for key, v in f.items()}
        rio.spio.loadmat = loadmat_csc
        try:
            F, b, Aseq, B, nx, ny, nt = quiet(rio.generate_crossPhantom, 15)
        finally:
            rio.spio.loadmat = real_loadmat
            os.chdir(here)
    F = _sps.csr_matrix(F)
    F.sum_duplicates()
    F.sort_indices()
    assert (nx, ny, nt) == (Nf, Nf, T) and F.shape == (T * 700, T * npix) and len(Aseq) == T
    xr, yr = sparse_dynamic_vectors(T * npix, T * 700)
    blk_fwd = np.concatenate([np.asarray(Aseq[t] @ xr[t * npix:(t + 1) * npix]).reshape(-1) for t in range(T)])
    blk_adj = np.concatenate([np.asarray(Aseq[t].T @ yr[t * 700:(t + 1) * 700]).reshape(-1) for t in range(T)])
    out = dict(T=T, N=Nf, rows_per_frame=700, F_data=F.data.astype(np.float32), F_indices=F.indices.astype(np.int32),
               F_indptr=F.indptr.astype(np.int32), F_shape=np.array(F.shape), b=np.asarray(b).reshape(-1),
               B_concat=np.concatenate([np.asarray(v).reshape(-1) for v in B]), block_nnz=np.array([a.nnz for a in Aseq]),
               blk_fwd=blk_fwd, blk_adj_s=blk_adj[::37], F_fwd=np.asarray(F @ xr).reshape(-1), F_adj_s=np.asarray(F.T @ yr).reshape(-1)[::37])
    # the solvers on what the loader returned
    bv = np.asarray(b).reshape(-1, 1)
    x, info = quiet(Hybrid_LSQR, F, bv, 10, 1e-2)
    out.update(lsqr_x_s=np.asarray(x).reshape(-1)[::16], lsqr_x_norm=np.linalg.norm(x), lsqr_its=info["its"])
    x, info = quiet(CGLS, F, bv, np.zeros((T * npix, 1)), 12, 0)
    out.update(cgls_x_s=np.asarray(x).reshape(-1)[::16], cgls_x_norm=np.linalg.norm(x), cgls_relResidual=info["relResidual"])
    tf = 5
    L2 = refops.gen_first_derivative_operator_2D(Nf, Nf)
    x, info = quiet(MMGKS, Aseq[tf], np.asarray(B[tf]).reshape(-1, 1), L2, 2, 1, 1, 6, 1e-2, None, epsilon=0.1)
    out.update(frame=tf, mmgks_frame_x=np.asarray(x).reshape(-1), mmgks_frame_Residual=info["Residual"])
    Lst = refops.gen_spacetime_derivative_operator(Nf, Nf, T)
    x, info = quiet(GKS, F, bv, Lst, 2, 4, 1e-2, None)
    out.update(gks_x_s=np.asarray(x).reshape(-1)[::16], gks_x_norm=np.linalg.norm(x), gks_Residual=info["Residual"])
    save("sparse_dynamic_crossphantom_like", **out)


if __name__ == "__main__":
    if len(sys.argv) > 1:                      # python tools/make_goldens.py g5b_isotv [...]: only the named families
        for name in sys.argv[1:]:
            globals()[name]()
        print("done ->", OUT)
        sys.exit(0)
    g1_blur()
    g2_cgls()
    g3_decomp()
    g4_hybrid()
    g4b_oneshot()
    g5_gks()
    g45_lcurve()
    g5d_framelet()
    g5b_isotv()
    g6_derivs()
    g7_regparam()
    g7b_dp_corners()
    g8_deblur1d()
    g9_sparse_dynamic()
    print("done ->", OUT)
