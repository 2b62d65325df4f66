"""Pin the CPU oracle (oracle/cpu_ref.py) against the golden vectors produced by the reference.

CPU-only (`-m "not gpu"`).  Tolerances: the oracle is float64 like the reference, so agreement is
to rounding (1e-10) except where an optimiser (fminbound) sits in the loop (1e-6 on lambda)."""
import numpy as np
import pytest

from conftest import load_golden, relerr
from oracle import cpu_ref as O

def lam_close(ours, ref, rtol):
    """lambda histories agree; where GCV sits on its flat floor near the fminbound lower limit
    (1e-9) the minimiser's position is rounding noise, so only its order of magnitude is compared."""
    ours, ref = np.asarray(ours, dtype=float), np.asarray(ref, dtype=float)
    if ours.shape != ref.shape:
        return False
    floor = ref < 1e-7
    ok_big = np.allclose(ours[~floor], ref[~floor], rtol=rtol, atol=1e-12)
    ok_floor = np.all(ours[floor] < 1e-7)
    return bool(ok_big and ok_floor)


BLUR_GAUSS = [("g9x9_s3_32x32", (9, 9), (3, 3)), ("g5x7_s1-2_24x40", (5, 7), (1, 2)),
              ("g10x10_s2_16x16", (10, 10), (2, 2)), ("g9x9_s3_64x64", (9, 9), (3, 3)),
              ("g3x3_s1_7x5", (3, 3), (1, 1)), ("g9x9_s3_6x6", (9, 9), (3, 3))]


@pytest.mark.parametrize("name,dim,spread", BLUR_GAUSS)
@pytest.mark.parametrize("use_scipy", [True, False])
def test_blur_gauss(name, dim, spread, use_scipy):
    g = load_golden("blur2d_" + name)
    psf, center = O.gauss_psf(dim, spread)
    assert np.allclose(psf, g["psf"], rtol=0, atol=1e-15)
    assert np.array_equal(center, g["center"])
    A = O.Blur2D(psf, int(g["nx"]), int(g["ny"]), use_scipy=use_scipy)
    assert relerr(A @ g["x"], g["fwd"]) < 1e-13
    assert relerr(A.T @ g["y"], g["bwd"]) < 1e-13
    out3 = A @ g["X3"]
    assert out3.shape == g["fwd3"].shape and relerr(out3, g["fwd3"]) < 1e-13
    assert (A @ g["x"].reshape(-1, 1)).shape == (A.shape[0], 1)


@pytest.mark.parametrize("name", ["asym7x5_20x28", "asym4x6_33x17"])
@pytest.mark.parametrize("use_scipy", [True, False])
def test_blur_asymmetric(name, use_scipy):
    g = load_golden("blur2d_" + name)
    A = O.Blur2D(g["psf"], int(g["nx"]), int(g["ny"]), use_scipy=use_scipy)
    assert relerr(A @ g["x"], g["fwd"]) < 1e-13
    assert relerr(A.T @ g["y"], g["bwd"]) < 1e-13


def test_blur_adjoint_identity_symmetric_odd_psf():
    psf, _ = O.gauss_psf((9, 9), (3, 3))
    A = O.Blur2D(psf, 20, 24)
    rng = np.random.default_rng(0)
    x, y = rng.standard_normal(480), rng.standard_normal(480)
    assert abs(np.dot(A @ x, y) - np.dot(x, A.T @ y)) < 1e-12 * np.linalg.norm(x) * np.linalg.norm(y)


@pytest.mark.parametrize("name", ["cgls_blur64_x0zero", "cgls_blur64_x0ATb", "cgls_blur64_tol"])
def test_cgls(name):
    g = load_golden(name)
    N = int(g["N"])
    A = O.Blur2D(g["psf"], N, N)
    xt = g["x_true"] if "x_true" in g else None
    x, info = O.cgls(A, g["b"], g["x0"], int(g["max_iter"]), float(g["tol"]), x_true=xt)
    assert info["its"] == int(g["its"])
    assert relerr(x, g["x"]) < 1e-10
    assert np.allclose(info["relResidual"], g["relResidual"], rtol=1e-9)
    if xt is not None:
        assert np.allclose(info["relError"], g["relError"], rtol=1e-9)
    if "x_it10" in g:
        assert relerr(info["xHistory"][9], g["x_it10"]) < 1e-10
    assert len(info["xHistory"]) == info["its"] and info["regParam"] == []


def test_deblur1d_cgls_config_c1():
    g = load_golden("deblur1d_cgls_n256")
    n = int(g["n"])
    assert np.allclose(O.gauss_psf_1d(n, 3), g["psf"], atol=1e-16)
    for use_scipy in (True, False):
        A = O.Blur1D(g["psf"], use_scipy=use_scipy)
        assert relerr(A @ g["x_true"], g["b_true"]) < 1e-13
        assert relerr(A.T @ g["b"], g["ATb"]) < 1e-13
    x, info = O.cgls(A, g["b"], np.zeros((n, 1)), int(g["max_iter"]), float(g["tol"]), x_true=g["x_true"])
    assert info["its"] == int(g["its"])
    assert np.allclose(info["relError"][:30], g["relError"][:30], rtol=1e-6)
    assert relerr(x, g["x"]) < 1e-5     # 50 CG steps on a numerically rank-deficient 1-D blur amplify rounding


def test_golub_kahan_update_and_arnoldi_update():
    g = load_golden("gk_update_blur32")
    N = int(g["N"])
    A = O.Blur2D(g["psf"], N, N)
    b = g["b"].reshape(-1)
    U, B, V = (b / np.linalg.norm(b))[:, None], None, None
    for _ in range(int(g["steps"])):
        U, B, V = O.golub_kahan_update(A, U, B, V)
    assert B.shape == g["B"].shape and np.allclose(B, g["B"], rtol=1e-10, atol=1e-14)
    assert relerr(U, g["U"]) < 1e-9 and relerr(V, g["V"]) < 1e-9
    g = load_golden("arnoldi_update_blur32")
    Vq, H = (b / np.linalg.norm(b))[:, None], None
    for _ in range(int(g["steps"])):
        Vq, H = O.arnoldi_update(A, Vq, H)
    assert H.shape == g["H"].shape and np.allclose(H, g["H"], rtol=1e-9, atol=1e-13)
    assert relerr(Vq, g["V"]) < 1e-9


@pytest.mark.parametrize("d", [3, 8])
def test_golub_kahan(d):
    g = load_golden(f"golub_kahan_blur32_d{d}")
    N = int(g["N"])
    A = O.Blur2D(g["psf"], N, N)
    U, S, V = O.golub_kahan(A, g["b"], d)
    assert S.shape == g["S"].shape and np.allclose(S, g["S"], rtol=1e-10, atol=1e-14)
    assert relerr(U, g["U"]) < 1e-9 and relerr(V, g["V"]) < 1e-9


def test_arnoldi():
    g = load_golden("arnoldi_blur32_d6")
    N = int(g["N"])
    A = O.Blur2D(g["psf"], N, N)
    Q, H = O.arnoldi(A, g["b"], int(g["n_iter"]))
    assert H.shape == g["H"].shape and np.allclose(H, g["H"], rtol=1e-9, atol=1e-13)
    assert relerr(Q, g["Q"]) < 1e-9


@pytest.mark.parametrize("tag", ["stop1", "never"])
def test_arnoldi_dp_stop(tag):
    """decompositions.py:104-112: halts after the first step for gk_delta = 1, never for 0.5 (its residual stays ~0.95)."""
    g = load_golden("arnoldi_blur32_dpstop_" + tag)
    N = int(g["N"])
    A = O.Blur2D(g["psf"], N, N)
    Q, H = O.arnoldi(A, g["b"], int(g["n_iter"]), True, float(g["gk_eta"]), float(g["gk_delta"]))
    assert Q.shape == g["Q"].shape and H.shape == g["H"].shape
    assert np.allclose(H, g["H"], rtol=1e-9, atol=1e-13) and relerr(Q, g["Q"]) < 1e-9


@pytest.mark.parametrize("tag", ["lam1e-2", "gcv", "dp"])
@pytest.mark.parametrize("solver", ["hybrid_lsqr", "hybrid_gmres"])
def test_hybrid(solver, tag):
    g = load_golden(f"{solver}_blur32_{tag}")
    N = int(g["N"])
    A = O.Blur2D(g["psf"], N, N)
    rp = {"lam1e-2": 1e-2, "gcv": "gcv", "dp": "dp"}[tag]
    fn = getattr(O, solver)
    x, info = fn(A, g["b"], int(g["n_iter"]), rp, g["x_true"], delta=float(g["delta"]))
    assert info["its"] == int(g["its"]) and len(info["xHistory"]) == int(g["n_hist"])
    assert lam_close(info["regParam_history"], g["regParam_history"], 1e-3 if tag == "gcv" else 1e-6)
    assert np.allclose(info["relError"], g["relError"], rtol=1e-6)
    assert relerr(x, g["x"]) < 1e-6
    assert relerr(info["xHistory"][0], g["x_it1"]) < 1e-8
    if solver == "hybrid_gmres":
        assert np.allclose(info["relResidual"], g["relResidual"], rtol=1e-6, atol=1e-12)


@pytest.mark.parametrize("tag", ["lam1e-2", "gcv", "dp"])
def test_gks(tag):
    g = load_golden(f"gks_blur32_{tag}")
    N = int(g["N"])
    A = O.Blur2D(g["psf"], N, N)
    rp = {"lam1e-2": 1e-2, "gcv": "gcv", "dp": "dp"}[tag]
    for L in (O.FirstDerivative2D(N), O.MatrixOp(O.first_derivative_2d(N, N))):
        x, info = O.gks(A, g["b"], L, int(g["projection_dim"]), int(g["n_iter"]), rp, g["x_true"], delta=float(g["delta"]))
        assert info["its"] == int(g["its"])
        assert lam_close(info["regParam_history"], g["regParam_history"], 1e-3 if tag == "gcv" else 1e-6)
        assert np.allclose(info["relError"], g["relError"], rtol=1e-5)
        assert np.allclose(info["Residual"], g["Residual"], rtol=1e-4)
        assert relerr(x, g["x"]) < 1e-5


@pytest.mark.parametrize("tag,p,q,rp,eps", [("p2q1_lam1e-2", 2, 1, 1e-2, 0.1), ("p2q1_gcv", 2, 1, "gcv", 0.1),
                                            ("p1q1_lam1e-2", 1, 1, 1e-2, 0.1),
                                            ("p2q0.5_eps0.01_lam1e-3", 2, 0.5, 1e-3, 0.01)])
def test_mmgks(tag, p, q, rp, eps):
    g = load_golden("mmgks_blur32_" + tag)
    N = int(g["N"])
    assert float(g["epsilon"]) == eps
    A = O.Blur2D(g["psf"], N, N)
    x, info = O.mmgks(A, g["b"], O.FirstDerivative2D(N), p, q, int(g["projection_dim"]), int(g["n_iter"]), rp,
                      g["x_true"], epsilon=eps)
    assert info["its"] == int(g["its"])
    assert lam_close(info["regParam_history"], g["regParam_history"], 1e-3 if rp == "gcv" else 1e-6)
    assert np.allclose(info["relError"], g["relError"], rtol=1e-5)
    assert np.allclose(info["Residual"], g["Residual"], rtol=1e-4)
    assert relerr(x, g["x"]) < 1e-5


def test_dynamic_blockdiag_spacetime():
    g = load_golden("gks_dyn3x16_lam1e-2")
    N, nt = int(g["N"]), int(g["nt"])
    F = O.BlockDiag([O.Blur2D(g["psfs"][t], N, N) for t in range(nt)])
    L = O.SpaceTimeDerivative(N, nt)
    x, info = O.gks(F, g["b"], L, 3, int(g["n_iter"]), 1e-2, g["x_true"])
    assert np.allclose(info["relError"], g["relError"], rtol=1e-6) and relerr(x, g["x"]) < 1e-6
    g = load_golden("mmgks_dyn3x16_p2q1_lam1e-2")
    x, info = O.mmgks(F, g["b"], L, 2, 1, 3, int(g["n_iter"]), 1e-2, g["x_true"])
    assert np.allclose(info["relError"], g["relError"], rtol=1e-6) and relerr(x, g["x"]) < 1e-6
    assert np.allclose(info["Residual"], g["Residual"], rtol=1e-5)


@pytest.mark.parametrize("tag,q,rp", [("q1_lam1e-2", 1, 1e-2), ("q0.5_lam1e-3", 0.5, 1e-3), ("q1_gcv", 1, "gcv")])
def test_mmgks_group_sparsity_branch(tag, q, rp):
    """MMGKS(..., GS='GS', prob_dims=...) (MMGKS.py:45-52,78-91) as run by the reference itself."""
    g = load_golden("mmgks_dyn3x16_gs_" + tag)
    N, nt = int(g["N"]), int(g["nt"])
    F = O.BlockDiag([O.Blur2D(g["psfs"][t], N, N) for t in range(nt)])
    L = O.SpaceTimeDerivative(N, nt)                      # handed over like the reference's call; replaced inside
    x, info = O.mmgks(F, g["b"], L, 2, q, 3, int(g["n_iter"]), rp, g["x_true"], GS=True, prob_dims=(N, N, nt))
    assert info["its"] == int(g["its"])
    lam, lam_ref = np.asarray(info["regParam_history"], dtype=float), np.asarray(g["regParam_history"], dtype=float)
    big = lam_ref > 1e-5        # GCV minima at the lower end of the search interval are flat: lambda itself is not determined there
    assert np.allclose(lam[big], lam_ref[big], rtol=1e-4 if rp == "gcv" else 0) and np.all(lam[~big] < 1e-5)
    assert np.allclose(info["relError"], g["relError"], rtol=1e-5) and relerr(x, g["x"]) < 1e-5
    assert np.allclose(info["Residual"], g["Residual"], rtol=1e-4)


def _isotv_L(g):
    import scipy.sparse as sp
    return sp.csr_matrix((g["L_data"], g["L_indices"], g["L_indptr"]), shape=tuple(g["L_shape"]))


@pytest.mark.parametrize("tag,q,rp", [("q1_lam1e-2", 1, 1e-2), ("q0.5_lam1e-3", 0.5, 1e-3), ("q1_gcv", 1, "gcv")])
def test_mmgks_isotv_branch(tag, q, rp):
    """MMGKS(..., isoTV='isoTV', prob_dims=...) (MMGKS.py:61-77) as run by the reference's own MMGKS.py and
    operators_old.py over tools/oracle_shim's restatement of pylops.FirstDerivative (parity unpinned for that stencil)."""
    import scipy.sparse as sp
    g = load_golden("mmgks_dyn3x16_isotv_" + tag)
    assert int(g["pylops_first_derivative_is_shim"]) == 1
    N, nt = int(g["N"]), int(g["nt"])
    Lm = sp.vstack((O.old_spatial_derivative_operator(N, N, nt), O.old_time_derivative_operator(N, N, nt))).tocsr()
    assert abs(Lm - _isotv_L(g)).max() == 0.0             # operators_old.py:35-61 as the reference assembled them
    F = O.BlockDiag([O.Blur2D(g["psfs"][t], N, N) for t in range(nt)])
    x, info = O.mmgks(F, g["b"], O.MatrixOp(Lm), 2, q, 3, int(g["n_iter"]), rp, g["x_true"], isoTV=True, prob_dims=(N, N, nt))
    assert info["its"] == int(g["its"])
    lam, lam_ref = np.asarray(info["regParam_history"], dtype=float), np.asarray(g["regParam_history"], dtype=float)
    big = lam_ref > 1e-5
    assert np.allclose(lam[big], lam_ref[big], rtol=1e-4 if rp == "gcv" else 0) and np.all(lam[~big] < 1e-5)
    # the reference's PyLops operators round their products to float32 (operators_old.py:31 dtype), the oracle does not
    assert np.allclose(info["relError"], g["relError"], rtol=1e-5) and relerr(x, g["x"]) < 1e-5
    assert np.allclose(info["Residual"], g["Residual"], rtol=1e-4)


def test_isotv_weights_function():
    g = load_golden("isotv_weights_16x3")                 # weights.py:29-40
    w = O.iso_tv_weights(g["x"], g["u"], int(g["nx"]), int(g["ny"]), float(g["eps"]), float(g["q"]))
    assert w.shape == g["wr"].shape and np.allclose(w, g["wr"], rtol=1e-6)


def test_derivative_operators_and_weights():
    g = load_golden("deriv_ops")
    for n in (4, 5):
        assert np.array_equal(O.first_derivative_1d(n).toarray(), g[f"D1_{n}"])
        assert np.array_equal(O.first_derivative_2d(n, n).toarray(), g[f"D2_{n}"])
        assert np.array_equal(O.FirstDerivative2D(n).todense(), g[f"D2_{n}"])
        assert np.array_equal(O.FirstDerivative2D(n).T.todense(), g[f"D2_{n}"].T)
    for (N, nt) in ((4, 3), (3, 2)):
        assert np.array_equal(O.spacetime_derivative(N, N, nt).toarray(), g[f"Dst_{N}_{nt}"])
        assert np.array_equal(O.SpaceTimeDerivative(N, nt).todense(), g[f"Dst_{N}_{nt}"])
        assert np.array_equal(O.SpaceTimeDerivative(N, nt).T.todense(), g[f"Dst_{N}_{nt}"].T)
    u = g["holder_u"]
    assert np.allclose(O.smoothed_holder_weights(u, 0.1, 1), g["holder_eps0.1_p1"], rtol=1e-15)
    assert np.allclose(O.smoothed_holder_weights(u, 0.01, 0.5), g["holder_eps0.01_p0.5"], rtol=1e-15)
    assert np.allclose(O.smoothed_holder_weights(u, 0.1, 2), g["holder_eps0.1_p2"], rtol=1e-15)


def test_regparam_functions():
    g = load_golden("regparam_fn")
    Q_A, R_A, R_L, b = g["Q_A"], g["R_A"], g["R_L"], g["b"]
    for i, lam in enumerate(g["lams"]):
        assert np.isclose(O.gcv_numerator(lam, Q_A, R_A, R_L, b), g["gcv_num"][i], rtol=1e-9)
        assert np.isclose(O.gcv_denominator(lam, R_A, R_L), g["gcv_den"][i], rtol=1e-9)
        assert np.isclose(O.lcurve_curvature(lam, R_A, R_L, Q_A.T @ b), g["curvature"][i], rtol=1e-6)
    assert np.isclose(O.gcv_choose(Q_A, R_A, R_L, b), float(g["lam_gcv"]), rtol=1e-6)
    assert np.isclose(O.discrepancy_choose(Q_A, R_A, R_L, b, float(g["delta"])), float(g["lam_dp"]), rtol=1e-8)
    assert np.isclose(O.discrepancy_choose(Q_A, R_A, R_L, b, float(g["delta"]), eta=1.2), float(g["lam_dp_eta12"]), rtol=1e-8)
    assert np.isclose(O.lcurve_choose(R_A, R_L, Q_A.T @ b), float(g["lam_lcurve"]), rtol=1e-5)
    import scipy.linalg as sla
    Qb, s, _ = sla.svd(g["B"], full_matrices=False)
    k = g["B"].shape[1]
    for i, lam in enumerate(g["lams"]):
        assert np.isclose(O.gcv_numerator(lam, Qb, np.diag(s), np.eye(k), g["bhat"], variant="modified"), g["gcv_num_mod"][i], rtol=1e-9)
        assert np.isclose(O.gcv_denominator(lam, np.diag(s), np.eye(k), "modified", int(g["fullsize"])), g["gcv_den_mod"][i], rtol=1e-9)
    assert np.isclose(O.gcv_choose(Qb, np.diag(s), np.eye(k), g["bhat"], "modified", int(g["fullsize"])), float(g["lam_gcv_mod"]), rtol=1e-6)


# ------------------------------------------------------------------ Radon: invariants (its convention is pinned further down)
def test_radon_oracle_invariants():
    N = 32
    ang = np.linspace(0, np.pi, 12, endpoint=False)
    R = O.Radon2D(N, ang)
    rng = np.random.default_rng(1)
    x, y = rng.standard_normal(R.shape[1]), rng.standard_normal(R.shape[0])
    assert abs(np.dot(R @ x, y) - np.dot(x, R.T @ y)) < 1e-12 * np.linalg.norm(R @ x) * np.linalg.norm(y)
    img = rng.random((N, N))
    sino = (R @ img.reshape(-1)).reshape(len(ang), N) * N
    # axis-aligned views: angle 0 integrates along y (column sums), pi/2 along x (row sums)
    assert np.allclose(sino[0], img.sum(axis=0), rtol=1e-12)
    assert np.allclose(sino[6], img.sum(axis=1)[::-1], rtol=1e-9) or np.allclose(sino[6], img.sum(axis=1), rtol=1e-9)
    # total mass is conserved for every view whose rays all stay inside the detector
    disc = np.zeros((N, N))
    ii, jj = np.meshgrid(np.arange(N) - (N - 1) / 2, np.arange(N) - (N - 1) / 2, indexing="ij")
    disc[ii ** 2 + jj ** 2 <= 10 ** 2] = 1.0
    sd = (R @ disc.reshape(-1)).reshape(len(ang), N) * N
    assert np.allclose(sd.sum(axis=1), disc.sum(), rtol=2e-2)
    # central ray of a centred disc of radius 10 is ~ its diameter
    assert np.all(np.abs(sd[:, N // 2 - 1:N // 2 + 1].mean(axis=1) - 20.0) < 1.0)


def test_oneshot_solvers():
    g = load_golden("oneshot_blur32")
    N = int(g["N"])
    A = O.Blur2D(g["psf"], N, N)
    for tag, rp, bb, kw in [("lam", 1e-2, g["b"], {}), ("gcv", "gcv", g["b"], {}), ("dp", "dp", g["b_dp"], {"delta": float(g["delta_dp"])})]:
        x, lam = O.golub_kahan_tikhonov(A, bb, 3, rp, **kw)
        assert lam_close([lam], [float(g[f"gkt_{tag}_lam"])], 2e-3) and relerr(x, g[f"gkt_{tag}_x"]) < 1e-7
        x, lam = O.arnoldi_tikhonov(A, bb, 6, rp, **kw)
        assert lam_close([lam], [float(g[f"at_{tag}_lam"])], 2e-3) and relerr(x, g[f"at_{tag}_x"]) < (1e-4 if tag == "gcv" else 1e-6)
    assert relerr(O.gmres(A, g["b"], 5), g["gmres_x"]) < 1e-8


def test_fanbeam_oracle_invariants():
    """Fan-beam oracle (pinned to the reference's ASTRA images in test_fanbeam_oracle_matches_the_reference_demo_images): exact adjoint by construction; every ray's weights sum to its chord through the
    image square; geometry defaults of Tomography.define_proj_id (Tomography.py:48-60)."""
    N = 16
    A = O.FanBeam2D(N, np.linspace(0, np.pi, 6, endpoint=False))
    assert A.nd == int(np.sqrt(2) * N) and A.sod == 3 * N and A.odd == N and abs(A.pitch - 4 / 3) < 1e-15
    M = A.matrix()
    chord = np.asarray(M.sum(axis=1)).reshape(6, A.nd)
    assert chord.max() <= np.sqrt(2) * N + 1e-9 and np.all(chord[:, A.nd // 2 - 1:A.nd // 2 + 1] >= N - 1e-9)
    rng = np.random.default_rng(0)
    x, y = rng.standard_normal(N * N), rng.standard_normal(M.shape[0])
    assert abs(np.dot(A @ x, y) - np.dot(x, A.T @ y)) < 1e-12 * np.linalg.norm(x) * np.linalg.norm(y) * N


def test_fanbeam_oracle_vs_the_astra_outputs_the_reference_holds():
    """The only ASTRA outputs in the reference tree: the rendered fan-beam matrix `AA` and noisy sinogram `b` of the tectonic
    32^2, 30-view demo (demos/demo_Tomo_small_scale.ipynb:145,179; geometry Tomography.py:53-88, phantom phantoms.py:67),
    decoded into tests/golden/fanbeam_demo_image.npz.  They pin — to image precision — the rotation sense, the detector order,
    the (views, detectors) sinogram layout, the row-major image, and the side a ray running along a pixel boundary belongs to."""
    from astra_demo_image import check_against_demo_images
    g = load_golden("fanbeam_demo_image")
    assert np.abs(g["phantom"] * 255.0 - g["xtrue_grey"]).max() <= 1.0       # the phantom image is the phantom, as oriented
    A = O.FanBeam2D(int(g["nx"]), np.linspace(0, np.pi, int(g["views"]), endpoint=False))
    assert A.nd == int(g["n_det"])
    out = check_against_demo_images(g, A @ g["phantom"].reshape(-1), np.asarray(A.matrix().todense()))
    assert out["sino_corr"] > 0.9998 and out["dense_corr"] > 0.98, out
    # view 0 / detector 22 runs along the boundary of pixel columns 15 | 16, view 15 / detector 22 along rows 15 | 16
    s = (A @ g["phantom"].reshape(-1)).reshape(30, 45)
    assert abs(s[0, 22] - g["phantom"][:, 16].sum()) < 1e-12 and abs(s[15, 22] - g["phantom"][16].sum()) < 1e-12


def test_parallel_beam_convention_is_the_far_source_limit_of_the_pinned_fan_beam():
    """The parallel-beam geometry has no ASTRA output anywhere in the reference, but it shares ASTRA's angle / detector convention
    with the fan-beam geometry, and THAT is pinned to the reference's images (test above).  With the source a million image
    widths away and unit detector spacing at the origin the fan-beam operator must become the parallel-beam one: same rotation
    sense, detector order and (views, detectors) layout; what is left between the two is the difference of their interpolation
    models (exact ray / pixel lengths against Joseph's linear interpolation): 1e-3 on a smooth image.  The mirrored conventions
    are nowhere near."""
    from astra_demo_image import corr
    g = load_golden("fanbeam_demo_image")
    N, views = int(g["nx"]), int(g["views"])
    ang = np.linspace(0, np.pi, views, endpoint=False)
    sod = 1e6 * N
    F = O.FanBeam2D(N, ang, n_det=N, sod=sod, odd=float(N), pitch=(sod + N) / sod)
    R = O.Radon2D(N, ang, n_det=N, scale=1.0)
    ii, jj = np.meshgrid(np.arange(N), np.arange(N), indexing="ij")
    smooth = np.exp(-((ii - 10) ** 2 + (jj - 20) ** 2) / 30.0) + 0.5 * np.exp(-((ii - 22) ** 2 + (jj - 9) ** 2) / 50.0)
    for img, c_min, d_max in ((g["phantom"], 0.9995, 0.03), (smooth, 0.999995, 0.003)):
        sf, sr = (F @ img.reshape(-1)).reshape(views, N), (R @ img.reshape(-1)).reshape(views, N)
        assert corr(sf, sr) > c_min and relerr(sf, sr) < d_max, (corr(sf, sr), relerr(sf, sr))
        assert max(corr(sf, sr[:, ::-1]), corr(sf, sr[::-1])) < 0.9


def test_discrepancy_principle_corner_branches():
    """discrepancy_principle.py:56-66 (a projected regulariser with fewer rows than columns) and :100-129 (dptype 'tsvd' / 'tgsvd':
    the direct solvers' truncation index) against values the reference itself returned; :45-55 (an exactly zero singular value)
    ends in numpy's LinAlgError in the reference, and here."""
    g = load_golden("regparam_dp_corners")
    Q, R, b, delta = g["Q"], g["R"], g["b"], float(g["delta"])
    assert np.isclose(O.discrepancy_choose(Q, R, g["L_wide"], b, delta), float(g["lam_wide"]), rtol=1e-9)
    assert np.isclose(O.discrepancy_choose(Q, R, g["L_wide"], b, delta, eta=1.3), float(g["lam_wide_eta13"]), rtol=1e-9)
    with pytest.raises(np.linalg.LinAlgError):
        O.discrepancy_choose(Q, R, np.diag([1.0, 2, 3, 4, 5, 0.0]), b, delta)
    Qtb = g["U"].T @ b
    for dpt in ("tsvd", "tgsvd"):
        for tag, dl in (("", delta), ("_big", 6.0 * delta), ("_small", 0.05 * delta)):
            assert O.discrepancy_truncation(Qtb, 6, dl, dptype=dpt) == int(g[f"{dpt}{tag}"]), (dpt, tag)


@pytest.mark.parametrize("solver", ["hybrid_lsqr", "hybrid_gmres", "gks", "mmgks"])
def test_lcurve_through_the_solvers(solver):
    """regparam = 'l_curve' end to end (Hybrid_LSQR.py:94-98, Hybrid_GMRES.py:67-71, GKS.py:67-68, MMGKS.py:100-101)."""
    g = load_golden({"gks": "gks_blur32_lcurve", "mmgks": "mmgks_blur32_p2q1_lcurve"}.get(solver, f"{solver}_blur32_lcurve"))
    N = int(g["N"])
    A = O.Blur2D(g["psf"], N, N)
    if solver.startswith("hybrid"):
        x, info = getattr(O, solver)(A, g["b"], int(g["n_iter"]), "l_curve", g["x_true"])
        assert len(info["xHistory"]) == int(g["n_hist"]) and relerr(info["xHistory"][1], g["x_it2"]) < 1e-8
    elif solver == "gks":
        x, info = O.gks(A, g["b"], O.FirstDerivative2D(N), 3, int(g["n_iter"]), "l_curve", g["x_true"])
    else:
        x, info = O.mmgks(A, g["b"], O.FirstDerivative2D(N), 2, 1, 3, int(g["n_iter"]), "l_curve", g["x_true"])
    assert info["its"] == int(g["its"])
    assert np.allclose(np.array(info["regParam_history"], dtype=float), g["regParam_history"], rtol=1e-5)
    assert np.allclose(info["relError"], g["relError"], rtol=1e-6) and relerr(x, g["x"]) < 1e-6


def test_framelet_operator_restatement():
    """oracle.Framelet2D against create_framelet_operator's own matrix and actions (operators.py:50-113)."""
    g = load_golden("framelet_ops")
    W = O.Framelet2D(8, 6, 2)
    assert np.abs(np.column_stack([W._fwd(e) for e in np.eye(48)]) - g["dense_8_6_2"]).max() < 1e-14
    for (n, m, l) in ((8, 6, 2), (12, 12, 1), (16, 10, 3)):
        W = O.Framelet2D(n, m, l)
        assert W.shape == (n * (2 * l + 1) * m * (2 * l + 1), n * m)
        assert np.abs(W._fwd(g[f"x_{n}_{m}_{l}"]) - g[f"Wx_{n}_{m}_{l}"]).max() < 1e-13
        assert np.abs(W._adj(g[f"y_{n}_{m}_{l}"]) - g[f"WTy_{n}_{m}_{l}"]).max() < 1e-13


@pytest.mark.parametrize("tag", ["lam1e-2", "gcv"])
@pytest.mark.parametrize("solver", ["gks", "mmgks"])
def test_framelet_regulariser_through_the_solvers(solver, tag):
    """GKS / MMGKS with L = create_framelet_operator(32, 32, 2): the large-scale demos' regulariser."""
    g = load_golden("gks_blur32_framelet_" + tag if solver == "gks" else "mmgks_blur32_framelet_p2q1_" + tag)
    N = int(g["N"])
    A, W = O.Blur2D(g["psf"], N, N), O.Framelet2D(N, N, int(g["level"]))
    rp = 1e-2 if tag == "lam1e-2" else "gcv"
    if solver == "gks":
        x, info = O.gks(A, g["b"], W, 3, int(g["n_iter"]), rp, g["x_true"])
    else:
        x, info = O.mmgks(A, g["b"], W, 2, 1, 3, int(g["n_iter"]), rp, g["x_true"])
    assert info["its"] == int(g["its"])
    assert lam_close(info["regParam_history"], g["regParam_history"], 1e-3 if tag == "gcv" else 1e-6)
    assert np.allclose(info["relError"], g["relError"], rtol=1e-6) and relerr(x, g["x"]) < 1e-6
    assert np.allclose(info["Residual"], g["Residual"], rtol=1e-4)


def sparse_dynamic_vectors(n, m):
    """The probe vectors of tools/make_goldens.py g9_sparse_dynamic (formulas, not stored)."""
    return np.sin(0.37 * np.arange(n)) + 0.25 * np.cos(0.011 * np.arange(n)), np.cos(0.53 * np.arange(m)) - 0.1


def sparse_dynamic_golden():
    import scipy.sparse as sp
    g = load_golden("sparse_dynamic_crossphantom_like")
    F = sp.csr_matrix((g["F_data"].astype(np.float64), g["F_indices"], g["F_indptr"]), shape=tuple(g["F_shape"]))
    return g, F


def test_sparse_dynamic_frame_slicing_and_solvers():
    """The sparse-forward-matrix dynamic path against the reference's own loader and solvers run on a synthetic
    stand-in file (make_goldens g9): frame blocks as io.py:223-225 cuts them (the entries outside the blocks dropped), then CGLS /
    Hybrid_LSQR on the whole matrix, MMGKS on one frame's block, GKS with the space-time regulariser."""
    g, F = sparse_dynamic_golden()
    T, N, rpf = int(g["T"]), int(g["N"]), int(g["rows_per_frame"])
    npix = N * N
    AA, B = O.dynamic_frame_blocks(F, g["b"], T, rpf, npix)
    assert np.array_equal(np.array([a.nnz for a in AA]), g["block_nnz"]) and F.nnz > sum(a.nnz for a in AA)
    assert np.array_equal(np.concatenate(B), g["B_concat"])
    xr, yr = sparse_dynamic_vectors(T * npix, T * rpf)
    assert np.allclose(np.concatenate([AA[t] @ xr[t * npix:(t + 1) * npix] for t in range(T)]), g["blk_fwd"], rtol=1e-13, atol=1e-13)
    assert np.allclose(np.concatenate([AA[t].T @ yr[t * rpf:(t + 1) * rpf] for t in range(T)])[::37], g["blk_adj_s"], rtol=1e-13, atol=1e-13)
    Fo = O.MatrixOp(F)
    assert np.allclose(Fo @ xr, g["F_fwd"], rtol=1e-13, atol=1e-13) and np.allclose((Fo.T @ yr)[::37], g["F_adj_s"], rtol=1e-13, atol=1e-13)
    bv = g["b"].reshape(-1, 1)
    x, info = O.hybrid_lsqr(Fo, bv, 10, 1e-2)
    assert info["its"] == int(g["lsqr_its"]) and relerr(x.reshape(-1)[::16], g["lsqr_x_s"]) < 1e-8
    assert abs(np.linalg.norm(x) / float(g["lsqr_x_norm"]) - 1) < 1e-9
    x, info = O.cgls(Fo, bv, np.zeros((T * npix, 1)), 12, 0)
    assert relerr(x.reshape(-1)[::16], g["cgls_x_s"]) < 1e-9 and np.allclose(info["relResidual"], g["cgls_relResidual"], rtol=1e-8)
    tf = int(g["frame"])
    x, info = O.mmgks(O.MatrixOp(AA[tf]), B[tf].reshape(-1, 1), O.FirstDerivative2D(N), 2, 1, 1, 6, 1e-2, None, epsilon=0.1)
    assert relerr(x, g["mmgks_frame_x"]) < 1e-7 and np.allclose(info["Residual"], g["mmgks_frame_Residual"], rtol=1e-6)
    x, info = O.gks(Fo, bv, O.SpaceTimeDerivative(N, T), 2, 4, 1e-2, None)
    assert relerr(x.reshape(-1)[::16], g["gks_x_s"]) < 1e-8 and np.allclose(info["Residual"], g["gks_Residual"], rtol=1e-7)


def test_product_slicing_helper_without_gpu():
    """trips_py_amd.operators.slice_dynamic_frames (host-side, scipy only) = the reference's blocks."""
    from trips_py_amd.operators import slice_dynamic_frames
    g, F = sparse_dynamic_golden()
    T, N, rpf = int(g["T"]), int(g["N"]), int(g["rows_per_frame"])
    AA, B = slice_dynamic_frames(F, g["b"], T, rpf, N * N)
    assert np.array_equal(np.array([a.nnz for a in AA]), g["block_nnz"]) and np.array_equal(np.concatenate(B), g["B_concat"])
    xr, _ = sparse_dynamic_vectors(T * N * N, T * rpf)
    assert np.allclose(np.concatenate([AA[t] @ xr[t * N * N:(t + 1) * N * N] for t in range(T)]), g["blk_fwd"], rtol=1e-13, atol=1e-13)
    with pytest.raises(ValueError):
        slice_dynamic_frames(F, g["b"], T + 1, rpf, N * N)
