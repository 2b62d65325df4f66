"""GPU parity of the projection solvers (HIP kernels through libtrk.so) with the reference's golden outputs.
Bars: fixed lambda -> final x <= 1e-5 relative (north_star); automatic lambda (gcv/dp) -> looser, reported bars
because the selectors' minima are flat (SURVEY §7 hard part 1: the reference itself moves 3e-3 between fp32 and fp64)."""
import numpy as np
import pytest
import torch

from conftest import bar, load_golden, maxrel, relerr
from test_oracle_golden import lam_close

pytestmark = pytest.mark.gpu
TOL = 1e-5
# Automatic-lambda bars (gcv / dp / l_curve): (relError history, x).  Rule (round 6): bar = max(8 x the deviation MEASURED on
# one MI355X against the reference's golden run, 1e-5 = north_star's bar for a fixed lambda) — measured values beside each
# (profiles/r05/bars.txt; the record is still logged with TRK_BARS_LOG).  Round 5 had 2 x measured: the order of the partial sums
# depends on the CU count (grid sizes of the sparse, band and Gram kernels) and the selectors' minima are flat — the reference moves
# itself by 3-4e-3 between fp32 and fp64 inputs — so a tight multiple would flake on another part or ROCm.
AUTO_BAR = {"gks-gcv": (1.4e-4, 6.4e-5),        # measured 1.7e-5, 8.0e-6
            "gks-dp": (2.3e-5, 2.7e-5),         # 2.8e-6, 3.3e-6
            "gks-lcurve": (1e-5, 1e-5),         # 3.1e-7, 1.1e-7
            "mmgks-gcv": (1.2e-5, 3.4e-5),      # 1.5e-6, 4.2e-6
            "mmgks-lcurve": (1e-5, 2e-5),       # 7.8e-7, 2.4e-6
            "mmgks_gs-gcv": (1e-5, 1e-5),       # 1.1e-7, 2.5e-7
            "mmgks_isotv-gcv": (3.5e-5, 2.3e-5),  # 4.3e-6, 2.8e-6
            "gks_framelet-gcv": (1e-5, 1e-5),   # 3.6e-7, 2.6e-7
            "mmgks_framelet-gcv": (1e-5, 1e-5)}  # 1.1e-7, 5.0e-7
HYBRID_AUTO_BAR = {("Hybrid_LSQR", "gcv"): 1e-5, ("Hybrid_LSQR", "dp"): 4.5e-5, ("Hybrid_GMRES", "gcv"): 1e-5, ("Hybrid_GMRES", "dp"): 1e-5,
                   ("Hybrid_LSQR", "lcurve"): 1e-5, ("Hybrid_GMRES", "lcurve"): 1e-5}   # measured 2.8e-7, 5.6e-6, 3.9e-7, 2.0e-7, 7.4e-8, 6.3e-8


def blur(g):
    from trips_py_amd.operators import Blur2D
    N = int(g["N"])
    return Blur2D(g["psf"], N, N)


def test_decompositions():
    from trips_py_amd.decompositions import arnoldi, arnoldi_update, golub_kahan, golub_kahan_update
    g = load_golden("golub_kahan_blur32_d8")
    A = blur(g)
    U, Sm, V = golub_kahan(A, g["b"], 8)
    assert Sm.shape == g["S"].shape and np.allclose(Sm, g["S"], rtol=1e-5, atol=1e-7)
    assert relerr(U, g["U"]) < 1e-4 and relerr(V, g["V"]) < 1e-4      # late Lanczos vectors amplify fp32 rounding
    assert relerr(U[:, :4], g["U"][:, :4]) < TOL and relerr(V[:, :3], g["V"][:, :3]) < TOL
    g = load_golden("gk_update_blur32")
    b = g["b"].reshape(-1, 1)
    U, B, V = b / np.linalg.norm(b), np.empty(1), np.empty((b.size, 1))
    for _ in range(int(g["steps"])):
        U, B, V = golub_kahan_update(A, U, B, V)
    assert np.allclose(np.asarray(B), g["B"], rtol=1e-4, atol=1e-7)
    g = load_golden("arnoldi_update_blur32")
    Vq, H = b / np.linalg.norm(b), np.empty(1)
    for _ in range(int(g["steps"])):
        Vq, H = arnoldi_update(A, Vq, H)
    assert np.allclose(np.asarray(H), g["H"], rtol=1e-3, atol=1e-6)
    assert relerr(np.asarray(Vq)[:, :5], g["V"][:, :5]) < 1e-4
    VtV = np.asarray(Vq).T @ np.asarray(Vq)
    assert np.abs(VtV - np.eye(VtV.shape[0])).max() < 1e-5           # two Gram-Schmidt passes keep V orthonormal in fp32
    g = load_golden("arnoldi_blur32_d6")
    Q, H = arnoldi(A, g["b"], int(g["n_iter"]))
    assert H.shape == g["H"].shape and np.allclose(H, g["H"], rtol=1e-3, atol=1e-5)
    for tag in ("stop1", "never"):                                    # dp_stop inside arnoldi (decompositions.py:104-112)
        g = load_golden("arnoldi_blur32_dpstop_" + tag)
        Q, H = arnoldi(blur(g), g["b"], int(g["n_iter"]), True, gk_eta=float(g["gk_eta"]), gk_delta=float(g["gk_delta"]))
        assert Q.shape == g["Q"].shape and H.shape == g["H"].shape and np.allclose(H, g["H"], rtol=1e-3, atol=1e-5)


@pytest.mark.parametrize("tag", ["lam1e-2", "gcv", "dp"])
@pytest.mark.parametrize("solver", ["Hybrid_LSQR", "Hybrid_GMRES"])
def test_hybrid(solver, tag):
    from trips_py_amd import solvers as S
    g = load_golden(f"{solver.lower()}_blur32_{tag}")
    rp = {"lam1e-2": 1e-2, "gcv": "gcv", "dp": "dp"}[tag]
    kw = {"delta": float(g["delta"])} if tag == "dp" else {}
    x, info = getattr(S, solver)(blur(g), g["b"], int(g["n_iter"]), rp, g["x_true"], **kw)
    assert info["its"] == int(g["its"]) and len(info["xHistory"]) == int(g["n_hist"])
    assert lam_close(info["regParam_history"], g["regParam_history"], 2e-3)
    assert np.allclose(info["relError"], g["relError"], rtol=2e-4)
    bar(f"hybrid[{solver}-{tag}].x", relerr(x, g["x"]), HYBRID_AUTO_BAR[(solver, tag)] if tag != "lam1e-2" else TOL)
    assert relerr(info["xHistory"][0], g["x_it1"]) < TOL
    if solver == "Hybrid_GMRES":
        assert np.allclose(info["relResidual"], g["relResidual"], rtol=1e-4)


@pytest.mark.parametrize("tag", ["lam1e-2", "gcv", "dp"])
def test_gks(tag):
    from trips_py_amd import solvers as S
    from trips_py_amd.operators import FirstDerivative2D
    g = load_golden(f"gks_blur32_{tag}")
    N = int(g["N"])
    rp = {"lam1e-2": 1e-2, "gcv": "gcv", "dp": "dp"}[tag]
    kw = {"delta": float(g["delta"])} if tag == "dp" else {}
    x, info = S.GKS(blur(g), g["b"], FirstDerivative2D(N), int(g["projection_dim"]), int(g["n_iter"]), rp, g["x_true"], **kw)
    assert info["its"] == int(g["its"]) and len(info["xHistory"]) == int(g["n_iter"])
    if tag == "lam1e-2":
        assert relerr(x, g["x"]) < TOL, relerr(x, g["x"])
        assert relerr(info["xHistory"][0], g["x_it1"]) < TOL
        assert np.allclose(info["relError"], g["relError"], rtol=1e-4)
        assert np.allclose(info["Residual"], g["Residual"], rtol=1e-3)
    else:
        bar(f"gks[{tag}].relError", maxrel(info["relError"], g["relError"]), AUTO_BAR[f"gks-{tag}"][0])
        bar(f"gks[{tag}].x", relerr(x, g["x"]), AUTO_BAR[f"gks-{tag}"][1])


@pytest.mark.parametrize("tag,p,q,rp,eps", [("p2q1_lam1e-2", 2, 1, 1e-2, 0.1), ("p1q1_lam1e-2", 1, 1, 1e-2, 0.1),
                                            ("p2q0.5_eps0.01_lam1e-3", 2, 0.5, 1e-3, 0.01), ("p2q1_gcv", 2, 1, "gcv", 0.1)])
def test_mmgks(tag, p, q, rp, eps):
    from trips_py_amd import solvers as S
    from trips_py_amd.operators import FirstDerivative2D
    g = load_golden("mmgks_blur32_" + tag)
    N = int(g["N"])
    x, info = S.MMGKS(blur(g), g["b"], FirstDerivative2D(N), p, q, int(g["projection_dim"]), int(g["n_iter"]), rp, g["x_true"], epsilon=eps)
    assert info["its"] == int(g["its"]) and len(info["xHistory"]) == int(g["n_iter"])
    if rp != "gcv":
        # MM re-weighting feeds fp32 rounding back through the weights: 5e-5 (q = 0.5: powf) ; TV (q = 1) meets 1e-5
        assert relerr(x, g["x"]) < (TOL if q == 1 and p == 2 else 5e-5), relerr(x, g["x"])
        assert np.allclose(info["relError"], g["relError"], rtol=2e-4)
        assert np.allclose(info["Residual"], g["Residual"], rtol=2e-3)
    else:
        bar("mmgks[p2q1_gcv].relError", maxrel(info["relError"], g["relError"]), AUTO_BAR["mmgks-gcv"][0])
        bar("mmgks[p2q1_gcv].x", relerr(x, g["x"]), AUTO_BAR["mmgks-gcv"][1])


@pytest.mark.parametrize("solver", ["Hybrid_LSQR", "Hybrid_GMRES", "GKS", "MMGKS"])
def test_lcurve_through_the_solvers(solver):
    """regparam = 'l_curve' end to end against the reference's own runs (Hybrid_LSQR.py:94-98, Hybrid_GMRES.py:67-71,
    GKS.py:67-68, MMGKS.py:100-101)."""
    from trips_py_amd import solvers as S
    from trips_py_amd.operators import FirstDerivative2D
    g = load_golden({"GKS": "gks_blur32_lcurve", "MMGKS": "mmgks_blur32_p2q1_lcurve"}.get(solver, f"{solver.lower()}_blur32_lcurve"))
    N = int(g["N"])
    if solver.startswith("Hybrid"):
        x, info = getattr(S, solver)(blur(g), g["b"], int(g["n_iter"]), "l_curve", g["x_true"])
        assert len(info["xHistory"]) == int(g["n_hist"])
        bar(f"lcurve[{solver}].x", relerr(x, g["x"]), HYBRID_AUTO_BAR[(solver, "lcurve")])
        key = None
    elif solver == "GKS":
        x, info = S.GKS(blur(g), g["b"], FirstDerivative2D(N), 3, int(g["n_iter"]), "l_curve", g["x_true"])
        key = "gks-lcurve"
    else:
        x, info = S.MMGKS(blur(g), g["b"], FirstDerivative2D(N), 2, 1, 3, int(g["n_iter"]), "l_curve", g["x_true"])
        key = "mmgks-lcurve"
    assert info["its"] == int(g["its"])
    lam, lam_ref = np.array(info["regParam_history"], dtype=float), g["regParam_history"]
    nz = lam_ref != 0                                    # (Hybrid-GMRES reports 0 for its first step)
    bar(f"lcurve[{solver}].lambda", maxrel(lam[nz], lam_ref[nz]), 2e-5)          # measured 5.7e-8 (hybrids) ... 5.6e-6 (GKS)
    if key is not None:
        bar(f"lcurve[{solver}].relError", maxrel(info["relError"], g["relError"]), AUTO_BAR[key][0])
        bar(f"lcurve[{solver}].x", relerr(x, g["x"]), AUTO_BAR[key][1])
    else:
        bar(f"lcurve[{solver}].relError", maxrel(info["relError"], g["relError"]), 1e-5)     # measured 1.7e-8


@pytest.mark.parametrize("tag", ["lam1e-2", "gcv"])
@pytest.mark.parametrize("solver", ["GKS", "MMGKS"])
def test_framelet_regulariser_through_the_solvers(solver, tag):
    """GKS / MMGKS with L = create_framelet_operator(32, 32, 2) — the regulariser of the reference's large-scale demos
    (demos/demo_2D_Deblurring_large_scale.ipynb:403, demo_Tomo_large_scale.ipynb:687; operators.py:50-113) — against its own runs."""
    from trips_py_amd import solvers as S
    from trips_py_amd.operators import create_framelet_operator
    g = load_golden("gks_blur32_framelet_" + tag if solver == "GKS" else "mmgks_blur32_framelet_p2q1_" + tag)
    N = int(g["N"])
    W = create_framelet_operator(N, N, int(g["level"]))
    rp = 1e-2 if tag == "lam1e-2" else "gcv"
    if solver == "GKS":
        x, info = S.GKS(blur(g), g["b"], W, 3, int(g["n_iter"]), rp, g["x_true"])
    else:
        x, info = S.MMGKS(blur(g), g["b"], W, 2, 1, 3, int(g["n_iter"]), rp, g["x_true"])
    assert info["its"] == int(g["its"]) and len(info["xHistory"]) == int(g["n_iter"])
    if tag == "lam1e-2":
        bar(f"framelet[{solver}-lam].x", relerr(x, g["x"]), TOL)
        bar(f"framelet[{solver}-lam].x_it1", relerr(info["xHistory"][0], g["x_it1"]), TOL)
        bar(f"framelet[{solver}-lam].relError", maxrel(info["relError"], g["relError"]), 1e-5)     # measured 8.4e-8 / 4.4e-8
        # (the norm of a residual orthogonalised against the basis: a cancelling sum, which sees the operators' fp32 rounding many
        #  times over — 4.7e-6 / 3.2e-4 with the CSR kernel's fp32 row sums; the other MMGKS goldens hold it to 2e-3 as well)
        bar(f"framelet[{solver}-lam].Residual", maxrel(info["Residual"], g["Residual"]), 1e-3)
    else:
        assert lam_close(info["regParam_history"], g["regParam_history"], 5e-2)      # (GCV sits on its floor near 1e-9 here: lam_close)
        key = "gks_framelet-gcv" if solver == "GKS" else "mmgks_framelet-gcv"
        bar(f"framelet[{solver}-gcv].relError", maxrel(info["relError"], g["relError"]), AUTO_BAR[key][0])
        bar(f"framelet[{solver}-gcv].x", relerr(x, g["x"]), AUTO_BAR[key][1])


def test_dynamic_blockdiag_spacetime():
    """Frame-major block-diagonal operator + space-time derivative (config C5 structure, tiny) vs the reference."""
    from trips_py_amd import solvers as S
    from trips_py_amd.operators import Blur2D, BlockDiagOp, SpaceTimeDerivative
    g = load_golden("gks_dyn3x16_lam1e-2")
    N, nt = int(g["N"]), int(g["nt"])
    frames = [Blur2D(g["psfs"][t], N, N) for t in range(nt)]
    F = BlockDiagOp(frames)
    L = SpaceTimeDerivative(N, nt)
    x, info = S.GKS(F, g["b"], L, 3, int(g["n_iter"]), 1e-2, g["x_true"])
    assert relerr(x, g["x"]) < TOL and np.allclose(info["relError"], g["relError"], rtol=1e-4)
    g = load_golden("mmgks_dyn3x16_p2q1_lam1e-2")
    x, info = S.MMGKS(F, g["b"], L, 2, 1, 3, int(g["n_iter"]), 1e-2, g["x_true"])
    assert relerr(x, g["x"]) < TOL and np.allclose(info["relError"], g["relError"], rtol=2e-4)
    assert np.allclose(info["Residual"], g["Residual"], rtol=1e-3)


@pytest.mark.parametrize("tag,q,rp", [("q1_lam1e-2", 1, 1e-2), ("q0.5_lam1e-3", 0.5, 1e-3), ("q1_gcv", 1, "gcv")])
def test_mmgks_group_sparsity_branch(tag, q, rp):
    """MMGKS(..., GS='GS', prob_dims=(nx, ny, nt)) (MMGKS.py:45-52,78-91) against the reference's own run."""
    from trips_py_amd import solvers as S
    from trips_py_amd.operators import Blur2D, BlockDiagOp, SpaceTimeDerivative
    g = load_golden("mmgks_dyn3x16_gs_" + tag)
    N, nt = int(g["N"]), int(g["nt"])
    F = BlockDiagOp([Blur2D(g["psfs"][t], N, N) for t in range(nt)])
    x, info = S.MMGKS(F, g["b"], SpaceTimeDerivative(N, nt), 2, q, 3, int(g["n_iter"]), rp, g["x_true"], GS="GS",
                      prob_dims=(N, N, nt))
    assert info["its"] == int(g["its"])
    if rp != "gcv":
        assert relerr(x, g["x"]) < 5e-5 and np.allclose(info["relError"], g["relError"], rtol=2e-4)
        assert np.allclose(info["Residual"], g["Residual"], rtol=2e-3)
    else:
        bar("mmgks_gs[gcv].relError", maxrel(info["relError"], g["relError"]), AUTO_BAR["mmgks_gs-gcv"][0])
        bar("mmgks_gs[gcv].x", relerr(x, g["x"]), AUTO_BAR["mmgks_gs-gcv"][1])
    with pytest.raises(TypeError):
        S.MMGKS(F, g["b"], SpaceTimeDerivative(N, nt), 2, q, 3, 2, 1e-2, GS="GS")


@pytest.mark.parametrize("tag,q,rp", [("q1_lam1e-2", 1, 1e-2), ("q0.5_lam1e-3", 0.5, 1e-3), ("q1_gcv", 1, "gcv")])
def test_mmgks_isotv_branch(tag, q, rp):
    """MMGKS(..., isoTV='isoTV', prob_dims=(nx, ny, nt)) (MMGKS.py:61-77) with the operators_old.py regulariser, against
    the reference's own MMGKS.py / operators_old.py run over the restated pylops.FirstDerivative (parity unpinned there)."""
    import scipy.sparse as sp
    from trips_py_amd import solvers as S
    from trips_py_amd.operators import Blur2D, BlockDiagOp, VStack, spatial_derivative_operator, time_derivative_operator
    g = load_golden("mmgks_dyn3x16_isotv_" + tag)
    N, nt = int(g["N"]), int(g["nt"])
    F = BlockDiagOp([Blur2D(g["psfs"][t], N, N) for t in range(nt)])
    L = VStack((spatial_derivative_operator(N, N, nt), time_derivative_operator(N, N, nt)))
    Lg = sp.csr_matrix((g["L_data"], g["L_indices"], g["L_indptr"]), shape=tuple(g["L_shape"]))
    assert L.shape == Lg.shape and abs(L.matrix - Lg).max() == 0.0          # the operator the reference assembled
    x, info = S.MMGKS(F, g["b"], L, 2, q, 3, int(g["n_iter"]), rp, g["x_true"], isoTV="isoTV", prob_dims=(N, N, nt))
    assert info["its"] == int(g["its"])
    if rp != "gcv":
        assert relerr(x, g["x"]) < 5e-5 and np.allclose(info["relError"], g["relError"], rtol=2e-4)
        assert np.allclose(info["Residual"], g["Residual"], rtol=2e-3)
    else:
        bar("mmgks_isotv[gcv].relError", maxrel(info["relError"], g["relError"]), AUTO_BAR["mmgks_isotv-gcv"][0])
        bar("mmgks_isotv[gcv].x", relerr(x, g["x"]), AUTO_BAR["mmgks_isotv-gcv"][1])
    with pytest.raises(TypeError):
        S.MMGKS(F, g["b"], L, 2, q, 3, 2, 1e-2, isoTV="isoTV")


def test_isotv_weights_kernel_vs_reference_golden():
    """trk_isotv_weights against the reference's iso_TV_weights (weights.py:29-40) and a general exponent."""
    import torch
    from trips_py_amd.engine import default_engine
    eng = default_engine()
    g = load_golden("isotv_weights_16x3")
    N, nt = int(g["nx"]), 3
    x = eng.to_vec(g["x"])
    u = eng.to_vec(g["u"])
    out = eng.empty(g["wr"].size)
    eng.isotv_weights(x, N, nt, u[2 * N * N * nt:], float(g["eps"]), float(g["q"]), out)
    assert np.allclose(out.cpu().numpy(), g["wr"].reshape(-1), rtol=2e-5)
    # q = 0.5 (general power), checked against the same formula in float64 on the fp32-rounded inputs
    eng.isotv_weights(x, N, nt, u[2 * N * N * nt:], 0.05, 0.5, out)
    X = x.cpu().numpy().astype(np.float64).reshape(N, N, nt)
    g1, g2 = np.zeros_like(X), np.zeros_like(X)
    g1[:, 1:-1] = 0.5 * X[:, 2:] - 0.5 * X[:, :-2]
    g2[1:-1] = 0.5 * X[2:] - 0.5 * X[:-2]
    w = ((g1 ** 2 + g2 ** 2 + 0.05 ** 2) ** ((0.5 - 2) / 4)).reshape(-1)
    wt = (u.cpu().numpy().astype(np.float64)[2 * N * N * nt:] ** 2 + 0.05 ** 2) ** ((0.5 - 2) / 4)
    assert np.allclose(out.cpu().numpy(), np.concatenate((w, w, wt)), rtol=2e-5)


def test_history_off_and_torch_io():
    import torch
    from trips_py_amd import solvers as S
    from trips_py_amd.operators import FirstDerivative2D
    g = load_golden("gks_blur32_lam1e-2")
    A, N = blur(g), int(g["N"])
    dev = A.engine.device
    bt = torch.from_numpy(g["b"].astype(np.float32)).to(dev)
    x, info = S.GKS(A, bt, FirstDerivative2D(N), 3, int(g["n_iter"]), 1e-2, history=False)
    assert isinstance(x, torch.Tensor) and x.shape == (N * N, 1) and info["xHistory"] == []
    assert relerr(x.cpu().numpy(), g["x"]) < TOL
    x, info = S.MMGKS(A, bt, FirstDerivative2D(N), 2, 1, 3, 10, 1e-2, history=False)
    assert relerr(x.cpu().numpy(), load_golden("mmgks_blur32_p2q1_lam1e-2")["x"]) < TOL
    gh = load_golden("hybrid_lsqr_blur32_lam1e-2")
    x, info = S.Hybrid_LSQR(blur(gh), torch.from_numpy(gh["b"].astype(np.float32)).to(dev), 12, 1e-2, history=False)
    assert relerr(x.cpu().numpy(), gh["x"]) < TOL


def test_oneshot_solvers():
    """Golub_Kahan_Tikhonov, Arnoldi_Tikhonov, GMRES vs the reference's outputs."""
    from trips_py_amd import solvers as S
    g = load_golden("oneshot_blur32")
    A = blur(g)
    for tag, rp, bb, kw in [("lam", 1e-2, g["b"], {}), ("gcv", "gcv", g["b"], {}), ("dp", "dp", g["b_dp"], {"delta": float(g["delta_dp"])})]:
        x, lam = S.Golub_Kahan_Tikhonov(A, bb, 3, rp, **kw)
        assert lam_close([lam], [float(g[f"gkt_{tag}_lam"])], 5e-3) and relerr(x, g[f"gkt_{tag}_x"]) < (TOL if tag == "lam" else 1e-4)
        x, lam = S.Arnoldi_Tikhonov(A, bb, 6, rp, **kw)
        assert lam_close([lam], [float(g[f"at_{tag}_lam"])], 5e-2) and relerr(x, g[f"at_{tag}_x"]) < (TOL if tag == "lam" else 2e-3)
    assert relerr(S.GMRES(A, g["b"], 5), g["gmres_x"]) < 1e-4


@pytest.mark.parametrize("solver", ["GKS", "MMGKS"])
def test_separate_tv_kernels_path_matches_the_golden_too(solver):
    """fused_tv=False keeps L x / w * (L x) / L^T / axpby as separate launches (the path operators without trk_tv_grad
    take); both paths meet the reference golden and agree with each other to fp32 rounding."""
    from trips_py_amd import solvers as S
    from trips_py_amd.operators import FirstDerivative2D
    g = load_golden("gks_blur32_lam1e-2" if solver == "GKS" else "mmgks_blur32_p2q1_lam1e-2")
    N = int(g["N"])
    L = FirstDerivative2D(N)
    out = {}
    for fused in (True, False):
        if solver == "GKS":
            x, info = S.GKS(blur(g), g["b"], L, int(g["projection_dim"]), int(g["n_iter"]), 1e-2, g["x_true"], fused_tv=fused)
        else:
            x, info = S.MMGKS(blur(g), g["b"], L, 2, 1, int(g["projection_dim"]), int(g["n_iter"]), 1e-2, g["x_true"],
                              epsilon=0.1, fused_tv=fused)
        assert relerr(x, g["x"]) < TOL, (fused, relerr(x, g["x"]))
        assert np.allclose(info["Residual"], g["Residual"], rtol=2e-3)
        out[fused] = x
    assert relerr(out[True], out[False]) < 5e-6


def _gk_paths_agree(N, na, strict):
    """GKState on the Radon projector: the three-launch step (forward, band reduction with epilogue + records, tile gather
    with epilogue + transposed copy) against apply / axpby / reduce as separate launches: same U, V, alpha, beta.
    strict: the process runs with TRK_RADON_EPI_F32=1 — the fused half step then combines in trk_axpby's fp32 arithmetic like the
    separate kernels and all twelve vectors agree to rounding.  Otherwise (the default: one rounding of a Op(x) + b z formed in
    float64) the two paths differ by a unit in the last place per step, which un-reorthogonalised Golub-Kahan on noise-free data
    amplifies ~5 x per step (measured 1.2e-5 at the third vector, 2.8e-3 at the sixth): only the first two steps are comparable."""
    from trips_py_amd.krylov import GKState
    from trips_py_amd.operators import Radon2DParallel
    A = Radon2DParallel(N, np.linspace(0, np.pi, na, endpoint=False))
    b = A.apply(torch.rand(N * N, device=A.engine.device, generator=torch.Generator(device=A.engine.device).manual_seed(2)))
    st_f, st_s, st_d = (GKState(A, b, 12, normalized=False) for _ in range(3))
    assert st_f.native_axpby
    st_s.native_axpby = False
    for _ in range(12):
        st_f.step(sync=False)
        st_s.step(sync=False)
        st_d.step(sync=False, defer=True)        # beta^2 finished by the next step's adjoint kernel / the flush in _sync
    # deferred and immediate forms of the SAME arithmetic: the norms are summed over different workgroup partitions (equal to fp64
    # rounding), the vectors equal to fp32 rounding
    assert np.allclose(st_d.alphas, st_f.alphas, rtol=1e-9) and np.allclose(st_d.betas, st_f.betas, rtol=1e-9)
    assert relerr(st_d.V.data[11].cpu().numpy(), st_f.V.data[11].cpu().numpy()) < 5e-6
    for k in range(12 if strict else 2):
        assert relerr(st_f.V.data[k].cpu().numpy(), st_s.V.data[k].cpu().numpy()) < 5e-6, k
        assert relerr(st_f.U.data[k + 1].cpu().numpy(), st_s.U.data[k + 1].cpu().numpy()) < 5e-6, k
    if strict:
        assert np.allclose(st_f.alphas, st_s.alphas, rtol=1e-6) and np.allclose(st_f.betas, st_s.betas, rtol=1e-6)
    else:
        assert np.allclose(st_f.alphas[:2], st_s.alphas[:2], rtol=1e-6) and np.allclose(st_f.betas[:2], st_s.betas[:2], rtol=1e-6)


@pytest.mark.parametrize("N,na", [(64, 30), (512, 180)])
def test_golub_kahan_fused_half_steps_equal_the_separate_kernels(N, na):
    _gk_paths_agree(N, na, strict=False)
    # ... and to rounding on all twelve steps when both paths use the same arithmetic (the switch is read once per process)
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, "-c", f"import sys; sys.path[:0] = [{here!r}, {os.path.dirname(here)!r}]; "
                        f"import test_gpu_solvers as T; T._gk_paths_agree({N}, {na}, True); print('strict ok')"],
                       env=dict(os.environ, TRK_RADON_EPI_F32="1"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "strict ok" in r.stdout, r.stderr[-1500:]


@pytest.mark.parametrize("N,its", [(32, 20), (128, 100)])
def test_hybrid_gmres_device_projected_solve_equals_the_host_one(N, its):
    """trk_hess_tikhonov (H appended from the sweep's device scalars, (H^T H + lam I) y = beta0 H[0,:]^T by Cholesky in LDS, no
    host round trip in the loop) against the host path (download of H's column, stacked lstsq): iterates, lambda history,
    relError and the reference's relResidual quirk — up to k = 100 > 64 KB of LDS."""
    from trips_py_amd.operators import Blur2D
    from trips_py_amd.problems import gauss_psf
    from trips_py_amd.solvers import Hybrid_GMRES
    A = Blur2D(gauss_psf((9, 9), (2, 2))[0], N, N)
    dev = A.engine.device
    xt = torch.rand(N * N, device=dev, generator=torch.Generator(device=dev).manual_seed(3))
    b = A.apply(xt)
    b = b + 0.01 * torch.randn(N * N, device=dev, generator=torch.Generator(device=dev).manual_seed(4)) * b.norm() / N
    xh, ih = Hybrid_GMRES(A, b, its, 1e-2, xt, device_solve=False)
    # the device kernel (c_loop=False), and the default since round 6: the library's one-call-per-iteration loop with the projected
    # problems as fixed-lambda jobs of the host worker threads (trk_hgmres_iter + trk_host_worker_post_hess_fixed)
    for kw in ({"c_loop": False}, {}):
        xd, idv = Hybrid_GMRES(A, b, its, 1e-2, xt, **kw)
        assert float(torch.linalg.norm(xd - xh) / torch.linalg.norm(xh)) < 1e-5, kw
        assert idv["regParam_history"] == ih["regParam_history"] and idv["its"] == ih["its"], kw
        assert np.allclose(idv["relError"], ih["relError"], rtol=1e-5), kw
        assert np.allclose(idv["relResidual"], ih["relResidual"], rtol=1e-6), kw
        for k in (0, its // 2, its - 1):
            a, c = idv["xHistory"][k].reshape(-1), ih["xHistory"][k].reshape(-1)
            assert float(torch.linalg.norm(a - c) / torch.linalg.norm(c)) < 1e-5, (k, kw)


def test_arnoldi_step_last_arriver_kernels_over_many_runs_at_512():
    """k_finalize_cgs and k_scale_fin hand partial sums from all workgroups to the one that draws the last ticket (write-through stores,
    loads past the caches, an agent-scope counter): at 512^2 the sweep's partials come from ~1 000 workgroups on all eight XCDs.  Twenty
    factorisations of 30 steps, each against the five separate calls of the Python step (finalize launches in between): the same H and
    the same basis bit for bit, every time — a stale line or a lost ticket would show as a different sum."""
    from trips_py_amd.operators import Blur2D
    from trips_py_amd.problems import gauss_psf
    from trips_py_amd.krylov import ArnoldiState
    N, steps = 512, 30
    A = Blur2D(gauss_psf((9, 9), (3, 3))[0], N, N)
    eng = A.engine
    b = torch.rand(N * N, device=eng.device, generator=torch.Generator(device=eng.device).manual_seed(21))
    eng.arnoldi_step, keep = None, eng.arnoldi_step
    try:
        five = ArnoldiState(A, b, steps)
        for _ in range(steps):
            five.step()
    finally:
        del eng.arnoldi_step
    assert keep is not None
    Href, Vref = five.H(), five.V.data[:steps + 1].clone()
    for rep in range(20):
        one = ArnoldiState(A, b, steps)
        for _ in range(steps):
            one.step()
        assert np.array_equal(one.H(), Href), rep
        assert torch.equal(one.V.data[:steps + 1], Vref), rep


@pytest.mark.parametrize("N,its,hist", [(64, 30, True), (128, 45, True), (96, 25, 3), (64, 13, False)])
def test_hybrid_gmres_gcv_one_library_call_per_iteration_equals_the_python_loop(N, its, hist):
    """trk_hgmres_iter — absorb the step that ran ahead, enqueue the next, collect the worker's answer for the iterate before, post this
    one, launch x = V y: the host side of a Hybrid-GMRES iteration with regparam='gcv' in one call — against the Python loop that makes
    those calls one by one (c_loop=False): the same lambda history (the same job on the same worker: 1e-10), iterates, relError and the
    reference's relResidual."""
    from trips_py_amd.operators import Blur2D
    from trips_py_amd.problems import gauss_psf
    from trips_py_amd.solvers import Hybrid_GMRES
    A = Blur2D(gauss_psf((9, 9), (2, 2))[0], N, N)
    dev = A.engine.device
    xt = torch.rand(N * N, device=dev, generator=torch.Generator(device=dev).manual_seed(5))
    b = A.apply(xt)
    b = b + 0.01 * torch.randn(N * N, device=dev, generator=torch.Generator(device=dev).manual_seed(6)) * b.norm() / N
    assert hasattr(A.engine.lib, "trk_hgmres_iter")
    xc, ic = Hybrid_GMRES(A, b, its, "gcv", xt, history=hist)
    xp, ip = Hybrid_GMRES(A, b, its, "gcv", xt, history=hist, c_loop=False)
    assert len(ic["regParam_history"]) == len(ip["regParam_history"]) == its and ic["its"] == ip["its"]
    assert np.allclose(ic["regParam_history"], ip["regParam_history"], rtol=1e-10, atol=0)
    assert np.allclose(ic["relResidual"], ip["relResidual"], rtol=1e-10)
    assert np.allclose(ic["relError"], ip["relError"], rtol=1e-6)
    assert float(torch.linalg.norm(xc - xp) / torch.linalg.norm(xp)) < 1e-6
    if hist is True:
        for k in (0, 11, its // 2, its - 1):
            a, c = ic["xHistory"][k].reshape(-1), ip["xHistory"][k].reshape(-1)
            assert float(torch.linalg.norm(a - c) / torch.linalg.norm(c)) < 1e-6, k
    # without x_true (no error partials) the same iterates
    x2, i2 = Hybrid_GMRES(A, b, its, "gcv", history=False)
    assert float(torch.linalg.norm(x2 - xc) / torch.linalg.norm(xc)) < 1e-6 and "relError" not in i2


@pytest.mark.parametrize("N,its,noise", [(64, 30, 0.01), (128, 40, 0.05), (96, 20, 0.001)])
def test_hybrid_gmres_discrepancy_principle_one_library_call_per_iteration(N, its, noise):
    """regparam='dp' through trk_hgmres_iter: V_{k+1}^T b grows by the dot the step's normalising pass takes (k_scale_fin<DOT>: block
    partials, the last workgroup adds them up and posts them with the step's scalars), the jobs are trk_host_worker_post_hess_dp, and a
    job that sets no positive lambda comes back to the interpreter's branches — against the Python loop (c_loop=False): lambda history
    to 1e-8 (the dots are summed in another order: 1e-16 of them), iterates, relError, relResidual."""
    from trips_py_amd.operators import Blur2D
    from trips_py_amd.problems import gauss_psf
    from trips_py_amd.solvers import Hybrid_GMRES
    A = Blur2D(gauss_psf((9, 9), (2, 2))[0], N, N)
    dev = A.engine.device
    g = torch.Generator(device=dev).manual_seed(N)
    xt = torch.rand(N * N, device=dev, generator=g)
    b = A.apply(xt)
    e = torch.randn(N * N, device=dev, generator=g)
    delta = noise * float(b.norm())
    b = b + e * (delta / e.norm())
    xc, ic = Hybrid_GMRES(A, b, its, "dp", xt, delta=delta)
    xp, ip = Hybrid_GMRES(A, b, its, "dp", xt, delta=delta, c_loop=False)
    assert len(ic["regParam_history"]) == len(ip["regParam_history"]) == its
    assert np.allclose(ic["regParam_history"], ip["regParam_history"], rtol=1e-8, atol=1e-300)
    assert np.allclose(ic["relResidual"], ip["relResidual"], rtol=1e-7)
    assert np.allclose(ic["relError"], ip["relError"], rtol=1e-6)
    assert float(torch.linalg.norm(xc - xp) / torch.linalg.norm(xp)) < 1e-6
    for k in (0, 5, its // 2, its - 1):
        a, c = ic["xHistory"][k].reshape(-1), ip["xHistory"][k].reshape(-1)
        assert float(torch.linalg.norm(a - c) / torch.linalg.norm(c)) < 1e-6, k


@pytest.mark.parametrize("reg", ["gcv", "dp"])
@pytest.mark.parametrize("hist,with_xt", [(True, True), (False, True), (False, False)])
def test_hybrid_lsqr_automatic_lambda_host_turn_in_one_library_call(reg, hist, with_xt):
    """trk_hlsqr_select — collect lambda of the step before, post the search for this step, solve the projected problem and launch the
    iterate: the host's turn of a Hybrid-LSQR iteration with gcv / the discrepancy principle in one call — against the four calls of the
    interpreter's loop (one_call=False): the same lambdas to the bit (the same searches on the same worker), the same iterates."""
    from trips_py_amd.operators import Radon2DParallel
    from trips_py_amd.solvers import Hybrid_LSQR
    N, na, its = 96, 40, 30
    A = Radon2DParallel(N, np.linspace(0, np.pi, na, endpoint=False))
    dev = A.engine.device
    g = torch.Generator(device=dev).manual_seed(9)
    xt = torch.rand(N * N, device=dev, generator=g)
    b = A.apply(xt)
    e = torch.randn(b.numel(), device=dev, generator=g)
    delta = 0.01 * float(b.norm())
    b = b + e * (delta / e.norm())
    kw = {"delta": delta} if reg == "dp" else {}
    assert hasattr(A.engine.lib, "trk_hlsqr_select")
    x1, i1 = Hybrid_LSQR(A, b, its, reg, xt if with_xt else None, history=hist, **kw)
    x0, i0 = Hybrid_LSQR(A, b, its, reg, xt if with_xt else None, history=hist, one_call=False, **kw)
    assert i1["regParam_history"] == i0["regParam_history"] and len(i1["regParam_history"]) == its - 1
    assert float(torch.linalg.norm(x1 - x0) / torch.linalg.norm(x0)) < 1e-6
    if with_xt:
        assert np.allclose(i1["relError"], i0["relError"], rtol=1e-6)
    if hist is True:
        for k in (0, its // 2, its - 2):
            a, c = i1["xHistory"][k].reshape(-1), i0["xHistory"][k].reshape(-1)
            assert float(torch.linalg.norm(a - c) / torch.linalg.norm(c)) < 1e-6, k


def test_gks_gram_rows_from_v_equal_the_stored_images_form():
    """GKS on stencil operators keeps no AV / LV: G_A, G_L rows come from one sweep over V with A^T A v_new and L^T L v_new
    (trk_gemv_t2).  Same iterates and lambda history as the stored-images form, on the reference golden and on a 256^2 problem
    with the automatic selector (the host-side Gram path)."""
    from trips_py_amd import solvers as S
    from trips_py_amd.operators import Blur2D, FirstDerivative2D
    from trips_py_amd.problems import gauss_psf
    g = load_golden("gks_blur32_lam1e-2")
    N = int(g["N"])
    L = FirstDerivative2D(N)
    xa, ia = S.GKS(blur(g), g["b"], L, int(g["projection_dim"]), int(g["n_iter"]), 1e-2, g["x_true"])
    xb, ib = S.GKS(blur(g), g["b"], L, int(g["projection_dim"]), int(g["n_iter"]), 1e-2, g["x_true"], gram_from_v=False)
    assert relerr(xa, g["x"]) < TOL and relerr(xb, g["x"]) < TOL and relerr(xa, xb) < 5e-6
    N = 256
    A = Blur2D(gauss_psf((9, 9), (2, 2))[0], N, N)
    dev = A.engine.device
    xt = torch.rand(N * N, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
    b = A.apply(xt)
    b = b + 0.01 * torch.randn(N * N, device=dev, generator=torch.Generator(device=dev).manual_seed(2)) * b.norm() / N
    for rp in (1e-2, "gcv"):
        xa, ia = S.GKS(A, b, FirstDerivative2D(N), 3, 25, rp, xt)
        xb, ib = S.GKS(A, b, FirstDerivative2D(N), 3, 25, rp, xt, gram_from_v=False)
        assert float(torch.linalg.norm(xa - xb) / torch.linalg.norm(xb)) < (2e-5 if rp == 1e-2 else 5e-3), rp
        assert np.allclose(ia["relError"], ib["relError"], rtol=1e-4 if rp == 1e-2 else 1e-2)


@pytest.mark.parametrize("kind", ["blur", "dynamic_tomo"])
def test_gks_gram_rows_from_the_sweep_equal_the_separate_pass(kind):
    """GKS with a numeric regparam: the Gram rows of the next basis vector from the sweep's own pass over V (trk_gemv_tn +
    trk_gram_row_from_sweep) against the separate pass with A^T A v_new / L^T L v_new: same iterates, residual norms and errors
    over 30 iterations — for a stencil A (both Gram matrices from V) and for the Radon A that keeps its images (only G_L)."""
    from trips_py_amd import solvers as S
    from trips_py_amd.operators import Blur2D, BlockDiagOp, FirstDerivative2D, Radon2DParallel, SpaceTimeDerivative
    from trips_py_amd.problems import gauss_psf
    if kind == "blur":
        N = 128
        A = Blur2D(gauss_psf((9, 9), (2, 2))[0], N, N)
        L = FirstDerivative2D(N)
    else:
        N, nt = 64, 4
        A = BlockDiagOp([Radon2DParallel(N, np.deg2rad(5.0 * t + 12.0 * np.arange(15))) for t in range(nt)])
        L = SpaceTimeDerivative(N, nt)
    dev = A.engine.device
    n = A.shape[1]
    xt = torch.rand(n, device=dev, generator=torch.Generator(device=dev).manual_seed(5))
    b = A.apply(xt)
    b = b + 0.01 * torch.randn(b.numel(), device=dev, generator=torch.Generator(device=dev).manual_seed(6)) * b.norm() / b.numel() ** 0.5
    xa, ia = S.GKS(A, b, L, 3, 30, 1e-2, xt)
    xb, ib = S.GKS(A, b, L, 3, 30, 1e-2, xt, gram_rows_from_sweep=False)
    assert float(torch.linalg.norm(xa - xb) / torch.linalg.norm(xb)) < 1e-5
    assert np.allclose(ia["relError"], ib["relError"], rtol=1e-5) and np.allclose(ia["Residual"], ib["Residual"], rtol=1e-3)
    for k in (0, 10, 29):
        u, v = ia["xHistory"][k].reshape(-1), ib["xHistory"][k].reshape(-1)
        assert float(torch.linalg.norm(u - v) / torch.linalg.norm(v)) < 1e-5, k


@pytest.mark.parametrize("kind,rp", [("blur", 1e-2), ("dynamic_tomo", 1e-2), ("blur", "gcv"), ("dynamic_tomo", "gcv"), ("dynamic_tomo", "dp")])
@pytest.mark.parametrize("hist", [True, False])
def test_gks_one_pass_for_new_vector_and_next_iterate_equals_the_two_pass_form(kind, rp, hist):
    """GKS (late round 6): the projected problem of iteration i + 1 solved before r - V c is formed — its Gram rows and rho come from
    the h-sweep's products — so that ONE pass over the basis leaves the new vector and the next iterate (trk_gemv_orth_iterate); against
    the form with a pass each (fused_orth_iterate=False): iterates, residual norms, errors and lambdas over 30 iterations, for a stencil
    A (Gram rows of both sides from V) and for the Radon A that keeps its images (A v_k = (A r - AV c)/rho)."""
    from trips_py_amd import solvers as S
    from trips_py_amd.operators import Blur2D, BlockDiagOp, FirstDerivative2D, Radon2DParallel, SpaceTimeDerivative
    from trips_py_amd.problems import gauss_psf
    if kind == "blur":
        N = 128
        A = Blur2D(gauss_psf((9, 9), (2, 2))[0], N, N)
        L = FirstDerivative2D(N)
    else:
        N, nt = 64, 4
        A = BlockDiagOp([Radon2DParallel(N, np.deg2rad(5.0 * t + 12.0 * np.arange(15))) for t in range(nt)])
        L = SpaceTimeDerivative(N, nt)
    dev = A.engine.device
    n = A.shape[1]
    xt = torch.rand(n, device=dev, generator=torch.Generator(device=dev).manual_seed(5))
    b0 = A.apply(xt)
    e = 0.01 * torch.randn(b0.numel(), device=dev, generator=torch.Generator(device=dev).manual_seed(6)) * b0.norm() / b0.numel() ** 0.5
    b = b0 + e
    kw = {"delta": float(e.norm())} if rp == "dp" else {}
    its = 30
    xa, ia = S.GKS(A, b, L, 3, its, rp, xt, history=hist, **kw)
    xb, ib = S.GKS(A, b, L, 3, its, rp, xt, history=hist, fused_orth_iterate=False, **kw)
    auto = isinstance(rp, str)
    tol = 2e-3 if auto else 1e-5                                     # an automatic lambda amplifies 1e-7 differences of the Gram data
    assert float(torch.linalg.norm(xa - xb) / torch.linalg.norm(xb)) < tol
    assert len(ia["relError"]) == len(ib["relError"]) == its and len(ia["regParam_history"]) == len(ib["regParam_history"]) == its
    assert np.allclose(ia["relError"], ib["relError"], rtol=1e-2 if auto else 1e-5)
    assert np.allclose(ia["Residual"], ib["Residual"], rtol=2e-2 if auto else 1e-3)
    assert np.allclose(ia["regParam_history"], ib["regParam_history"], rtol=5e-2 if auto else 0)
    assert ia["regParam"] == ia["regParam_history"][-1]
    if hist:
        for k in (0, 1, 10, its - 1):
            u, v = ia["xHistory"][k].reshape(-1), ib["xHistory"][k].reshape(-1)
            assert float(torch.linalg.norm(u - v) / torch.linalg.norm(v)) < tol, k
    if not auto:
        # the k x k work between the sweep and the pass in ONE launch (trk_gks_rows_solve) against its three launches: the same bits
        xc, ic = S.GKS(A, b, L, 3, its, rp, xt, history=hist, rows_and_solve_in_one=False)
        assert torch.equal(xa, xc) and ia["Residual"] == ic["Residual"] and ia["relError"] == ic["relError"]


def test_mmgks_pnorm2_unweighted_fidelity_gram_kept_incrementally():
    """MMGKS with pnorm = 2: wf = 1, so (AV)^T AV is unweighted and only grows — kept as in GKS (row k = V^T (A^T A v_k), no images
    A v_j, no weighted-Gram pass over them) while the L side is re-weighted every iteration.  Same iterates, residual norms and
    errors as the form that re-forms both weighted Grams, over 25 iterations."""
    from trips_py_amd import solvers as S
    from trips_py_amd.operators import Blur2D, FirstDerivative2D
    from trips_py_amd.problems import gauss_psf
    N = 160
    A = Blur2D(gauss_psf((9, 9), (2, 2))[0], N, N)
    L = FirstDerivative2D(N)
    dev = A.engine.device
    xt = torch.zeros(N, N, device=dev)
    xt[30:90, 40:120] = 1.0
    xt[100:140, 20:70] = 0.5
    xt = xt.reshape(-1)
    b = A.apply(xt)
    b = b + 0.01 * torch.randn(N * N, device=dev, generator=torch.Generator(device=dev).manual_seed(3)) * b.norm() / N
    xa, ia = S.MMGKS(A, b, L, 2, 1, 3, 25, 1e-2, xt)
    xb, ib = S.MMGKS(A, b, L, 2, 1, 3, 25, 1e-2, xt, unweighted_fidelity_gram=False)
    assert float(torch.linalg.norm(xa - xb) / torch.linalg.norm(xb)) < 1e-5
    # (relError ~ 0.1: iterates 1e-5 apart may move it by 1e-4 of itself)
    assert np.allclose(ia["relError"], ib["relError"], rtol=1e-4) and np.allclose(ia["Residual"], ib["Residual"], rtol=1e-3)
    for k in (0, 12, 24):
        u, v = ia["xHistory"][k].reshape(-1), ib["xHistory"][k].reshape(-1)
        assert float(torch.linalg.norm(u - v) / torch.linalg.norm(v)) < 1e-5, k


@pytest.mark.parametrize("N,its,pq", [(64, 25, (2, 1)), (160, 12, (2, 1)), (96, 20, (1, 1)), (64, 40, (2, 0.5))])
def test_mmgks_tv_gram_from_v_equals_the_stored_images_form(N, its, pq):
    """MMGKS with the 2-D first-difference L: the re-weighted Gram of L V formed from V (trk_wgram_tv, the default when N % 32 == 0
    and the basis stays within 48 vectors) against the form that stores L v_j and reads them back every iteration (MMGKS.py:94-95)."""
    from trips_py_amd import solvers as S
    from trips_py_amd.operators import Blur2D, FirstDerivative2D
    from trips_py_amd.problems import gauss_psf
    A = Blur2D(gauss_psf((9, 9), (2, 2))[0], N, N)
    L = FirstDerivative2D(N)
    dev = A.engine.device
    xt = torch.zeros(N, N, device=dev)
    xt[N // 5:N // 2, N // 4:3 * N // 4] = 1.0
    xt[5 * N // 8:7 * N // 8, N // 8:N // 2] = 0.5
    xt = xt.reshape(-1)
    b = A.apply(xt)
    b = b + 0.01 * torch.randn(N * N, device=dev, generator=torch.Generator(device=dev).manual_seed(3)) * b.norm() / N
    xa, ia = S.MMGKS(A, b, L, pq[0], pq[1], 3, its, 1e-2, xt)
    xb, ib = S.MMGKS(A, b, L, pq[0], pq[1], 3, its, 1e-2, xt, tv_gram_from_v=False)
    assert float(torch.linalg.norm(xa - xb) / torch.linalg.norm(xb)) < 1e-5
    # (a residual of 1e-7 next to ones of 100 is rounding noise: absolute floor)
    assert np.allclose(ia["relError"], ib["relError"], rtol=1e-4)
    assert np.allclose(ia["Residual"], ib["Residual"], rtol=2e-3, atol=1e-5 * max(ib["Residual"]))
    # automatic lambda (the Gram matrices visit the host every iteration; the fidelity images A v_j are kept): same two forms
    xg, ig = S.MMGKS(A, b, L, pq[0], pq[1], 3, 8, "gcv", xt)
    xh, ih = S.MMGKS(A, b, L, pq[0], pq[1], 3, 8, "gcv", xt, tv_gram_from_v=False)
    lg, lh = np.asarray(ig["regParam_history"]), np.asarray(ih["regParam_history"])
    big = np.maximum(lg, lh) > 1e-4                      # (a lambda at the search interval's lower end sits on a flat GCV curve)
    # (GCV minima are flat — SURVEY section 7: the REFERENCE on fp32 vs fp64 inputs picks 0.0340 vs 0.0282 — so two Gram matrices that
    #  agree to 5e-9 (the matrix pipe's bf16-split products against fp32 ones, round 4) may move a minimiser by 2e-3 of itself:
    #  measured 4.483e-3 vs 4.491e-3 on the (p, q) = (1, 1) case)
    assert np.allclose(lg[big], lh[big], rtol=1e-2)
    assert float(torch.linalg.norm(xg - xh) / torch.linalg.norm(xh)) < 3e-4
    if its + 4 > 48:                                     # a basis that outgrows the kernel: the stored-images form is chosen, silently
        xc, ic = S.MMGKS(A, b, L, pq[0], pq[1], 3, 48, 1e-2, xt)
        assert np.all(np.isfinite(ic["relError"]))


@pytest.mark.parametrize("kind,N,its", [("blur", 64, 30), ("radon", 128, 60), ("radon", 96, 100)])
def test_hybrid_lsqr_recurrence_equals_the_combination(kind, N, its):
    """Fixed lambda: the damped-LSQR short recurrence (trk_lsqr_damped_update) against x_k = V_k y_k formed from the projected
    solve (Hybrid_LSQR.py:104-105), every iterate and every relError — the two are the same iterate algebraically."""
    from trips_py_amd.operators import Blur2D, Radon2DParallel
    from trips_py_amd.problems import gauss_psf
    from trips_py_amd.solvers import Hybrid_LSQR
    rng = np.random.default_rng(N + its)
    if kind == "blur":
        A = Blur2D(gauss_psf((9, 9), (2, 2))[0], N, N)
    else:
        A = Radon2DParallel(N, np.linspace(0, np.pi, 45, endpoint=False))
    xt = rng.random(N * N).astype(np.float32)
    b = A.apply(torch.from_numpy(xt).cuda())
    b = (b + 0.01 * torch.randn_like(b) * b.norm() / b.numel() ** 0.5).cpu().numpy()
    for lam in (1e-2, 0.0):
        x1, i1 = Hybrid_LSQR(A, b, its, lam, xt)
        x0, i0 = Hybrid_LSQR(A, b, its, lam, xt, x_by_recurrence=False)
        assert len(i1["xHistory"]) == len(i0["xHistory"]) == its - 1 and i1["regParam_history"] == i0["regParam_history"]
        assert relerr(x1, x0) < TOL, relerr(x1, x0)
        for j in (0, 1, its // 2, its - 2):
            assert relerr(i1["xHistory"][j], i0["xHistory"][j]) < TOL, j
        assert np.allclose(i1["relError"], i0["relError"], rtol=1e-5)
        # the update riding the next Golub-Kahan step's adjoint half (trk_gk_step_lsqr: the default on the projector) against the
        # update in its own launch: the same floats, element by element; the error norms differ in their summation order only
        x4, i4 = Hybrid_LSQR(A, b, its, lam, xt, update_on_the_step=False)
        assert np.array_equal(x1, x4) and all(np.array_equal(p, q) for p, q in zip(i1["xHistory"], i4["xHistory"]))
        assert np.allclose(i1["relError"], i4["relError"], rtol=1e-12)
    if kind == "radon":
        # the same on a dynamic handle (frames of a block-diagonal projector in one launch: the rider indexes pixels per frame),
        # few angles per frame (the adjoint forms its records itself) and 32 x 32 tiles
        from trips_py_amd.operators import BlockDiagOp
        Fd = BlockDiagOp([Radon2DParallel(64, np.deg2rad(3.0 * t + 12.0 * np.arange(15))) for t in range(6)])
        xd = rng.random(Fd.shape[1]).astype(np.float32)
        bd = Fd.apply(torch.from_numpy(xd).cuda()).cpu().numpy()
        xa, ia = Hybrid_LSQR(Fd, bd, 20, 1e-2, xd)
        xb, ib = Hybrid_LSQR(Fd, bd, 20, 1e-2, xd, update_on_the_step=False)
        xc, ic = Hybrid_LSQR(Fd, bd, 20, 1e-2, xd, x_by_recurrence=False)
        assert np.array_equal(xa, xb) and np.allclose(ia["relError"], ib["relError"], rtol=1e-12)
        assert relerr(xa, xc) < TOL and np.allclose(ia["relError"], ic["relError"], rtol=1e-5)
    # no x_true, history kept: the same iterates
    x2, i2 = Hybrid_LSQR(A, b, 12, 1e-2)
    x3, i3 = Hybrid_LSQR(A, b, 12, 1e-2, x_by_recurrence=False)
    assert relerr(x2, x3) < TOL and relerr(i2["xHistory"][0], i3["xHistory"][0]) < TOL


def test_mailbox_downloads_and_one_call_gk_step():
    """trk_mailbox_*: scalars posted behind kernels arrive (every slot, re-used slots, overlapping ranges); trk_gk_step: one call
    = the two trk_op_apply_axpby half steps of krylov.GKState (bit-identical vectors and norms), Radon (native epilogue) and blur
    (generic fallback)."""
    from trips_py_amd.engine import Coef, default_engine
    from trips_py_amd.operators import Blur2D, Radon2DParallel
    from trips_py_amd.problems import gauss_psf
    eng = default_engine()
    S = eng.scalars(40)
    vals = np.arange(40, dtype=np.float64) * 1.5 + 0.25
    S.set(0, vals)
    hs = [S.host_later(i, i + 3) for i in range(0, 36)]           # more posts than slots: early handles wait on later posts
    for i, h in enumerate(hs):
        assert np.array_equal(h.get(), vals[i:i + 3])
    # the other direction (trk_scalars_put: host doubles in the arguments of a launch), one launch and several, with an offset
    big = eng.scalars(700)
    up = np.random.default_rng(0).standard_normal(300 + 131)
    big.set(7, up[:300])
    big.set(400, up[300:])
    big.set(699, 2.5)
    got = big.host(0, 700)
    assert np.array_equal(got[7:307], up[:300]) and np.array_equal(got[400:531], up[300:]) and got[699] == 2.5
    assert not got[:7].any() and not got[307:400].any() and not got[531:699].any()
    with pytest.raises(IndexError):
        big.set(650, up[:100])
    x = torch.rand(1 << 20, device=eng.device)
    eng.nrm2sq(x, S.ref(5))
    h = S.host_later(5, 6)                                        # behind the kernel that writes it
    assert abs(h.get()[0] - float((x.double() ** 2).sum())) < 1e-6 * x.numel()
    for A in (Radon2DParallel(64, np.linspace(0, np.pi, 24, endpoint=False)), Blur2D(gauss_psf((5, 5), (1, 1))[0], 48, 48)):
        m, n = A.shape
        g = torch.Generator(device=eng.device).manual_seed(3)
        b = torch.randn(m, device=eng.device, generator=g)
        U0, V0 = [b.clone()] + [eng.empty(m) for _ in range(3)], [eng.empty(n) for _ in range(3)]
        U1, V1 = [b.clone()] + [eng.empty(m) for _ in range(3)], [eng.empty(n) for _ in range(3)]
        AB0, AB1 = eng.scalars(8), eng.scalars(8)
        for AB in (AB0, AB1):
            eng.nrm2sq(b, AB.ref(0))
        for k in range(3):
            bk2, a2, b2 = AB0.ref(2 * k), AB0.ref(2 * k + 1), AB0.ref(2 * k + 2)
            A.apply_axpby(U0[k], Coef(1.0, den=bk2, sqrt_den=True),
                          0.0 if k == 0 else Coef(-1.0, num=bk2, den=AB0.ref(2 * k - 1), sqrt_num=True, sqrt_den=True),
                          None if k == 0 else V0[k - 1], V0[k], transpose=True, sumsq=a2)
            A.apply_axpby(V0[k], Coef(1.0, den=a2, sqrt_den=True), Coef(-1.0, num=a2, den=bk2, sqrt_num=True, sqrt_den=True), U0[k],
                          U0[k + 1], sumsq=b2)
            eng.gk_step(A._h, k, U1[k], None if k == 0 else V1[k - 1], V1[k], U1[k + 1], AB1, k > 0, True, k < 2)
        A.flush_deferred()
        assert np.array_equal(AB0.host(0, 7), AB1.host(0, 7))
        for k in range(3):
            assert torch.equal(V0[k], V1[k]) and torch.equal(U0[k + 1], U1[k + 1])


@pytest.mark.parametrize("kind", ["radon", "radon_bands", "blur"])
def test_gk_projection_rides_the_forward_pass_and_the_post(kind):
    """krylov.GKState.step_prefetch(project=b): U^T b row by row (the discrepancy principle's projection, discrepancy_principle.py:58).
    On the projector the forward half step's band reduction leaves <U[k+1], b> as block partials (trk_gk_step_proj) and the post of
    the step's norms adds them up (trk_mailbox_post_sum); checked against the products of the stored rows in float64, with the
    norms that travel in the same post, for steps enqueued one and several ahead.  The blur takes the separate-dot path."""
    from trips_py_amd.krylov import GKState
    from trips_py_amd.operators import Blur2D, Radon2DParallel
    from trips_py_amd.problems import gauss_psf
    if kind == "blur":
        A = Blur2D(gauss_psf((5, 5), (1, 1))[0], 48, 48)
    else:
        N = 64 if kind == "radon" else 600            # 600: several bands
        A = Radon2DParallel(N, np.linspace(0, np.pi, 24, endpoint=False))
    m, n = A.shape
    dev = A.engine.device
    b = torch.randn(m, device=dev, generator=torch.Generator(device=dev).manual_seed(5)).abs()
    steps = 9
    seen = {}
    for ahead, riders in ((1, True), (3, True), (3, False)):
        # riders: the post of a step's norms carried by the next step's adjoint kernel (trk_gk_step_post) or launched on its own
        gk = GKState(A, b, steps, normalized=False)
        gk.rider_posts = riders
        pending, n_enq = [], 0
        for k in range(steps):
            while n_enq < steps and len(pending) < ahead:
                pending.extend(gk.step_prefetch(project=b, more_follow=n_enq + 1 < steps))
                n_enq += 1
            gk.absorb(pending.pop(0))
            assert len(gk.uproj) == k + 2 and len(gk._alphas) == k + 1
        assert not pending
        U = gk.U.data[:steps + 1].double()
        want = (U @ b.double()).cpu().numpy()
        assert np.allclose(gk.uproj, want, rtol=1e-6, atol=1e-6 * float(b.norm()) ** 2)
        ab = gk.AB.host(0, 2 * steps + 1)
        assert np.allclose(np.square(gk._alphas), ab[1::2], rtol=1e-15) and np.allclose(np.square(gk._betas), ab[2::2], rtol=1e-15)
        assert np.allclose(ab[2::2], (U[1:] ** 2).sum(1).cpu().numpy(), rtol=1e-5)
        if kind != "blur":
            assert gk._UP is not None                               # the merged path ran
            assert np.array_equal(gk.AB.host(gk._uoff + 2, gk._uoff + steps + 1), np.asarray(gk.uproj[2:]))
        seen[(ahead, riders)] = (list(gk._alphas), list(gk._betas), list(gk.uproj), gk.beta0)
    assert seen[(3, True)] == seen[(3, False)] == seen[(1, True)]   # the same numbers whichever way they travelled


@pytest.mark.parametrize("k", [1, 5, 128, 129, 300])
def test_projected_solve_on_the_host_and_coefficients_in_the_launch_arguments(k):
    """trk_host_bidiag_tikhonov = trk_bidiag_tikhonov (same recurrence, float64) and trk_gemv_n_hosty = trk_gemv_n / trk_gemv_n_err
    with the coefficients taken from host memory: one launch up to 128 rows, several beyond (one fp32 rounding more per group)."""
    from trips_py_amd.engine import default_engine
    eng = default_engine()
    rng = np.random.default_rng(k)
    n = 70001                                                     # not a multiple of 4: vector body + tail
    al, be = 0.5 + rng.random(k), 0.1 + rng.random(k)
    beta0, mu = 3.7, 0.21
    AB = eng.scalars(2 * k + 1)
    ab = np.empty(2 * k + 1)
    ab[0], ab[1::2], ab[2::2] = beta0 ** 2, al ** 2, be ** 2
    AB.set(0, ab)
    for over in (False, True):
        Y = eng.scalars(k)
        eng.bidiag_tikhonov(AB.ref(1), 2, AB.ref(2), 2, k, mu, AB.ref(0), Y.ref(0), None, y_over_alpha=over)
        yh = eng.host_bidiag_tikhonov(al, be, beta0, mu, y_over_alpha=over)
        assert np.allclose(yh, Y.host(0, k), rtol=1e-12, atol=1e-14)
        # against the definition: argmin || [B; mu I] y - beta0 e1 ||
        B = np.zeros((k + 1, k))
        B[np.arange(k), np.arange(k)], B[np.arange(1, k + 1), np.arange(k)] = al, be
        rhs = np.zeros(2 * k + 1)
        rhs[0] = beta0
        yd = np.linalg.lstsq(np.vstack([B, mu * np.eye(k)]), rhs, rcond=None)[0]
        assert np.allclose(yh * (al if over else 1.0), yd, rtol=1e-9, atol=1e-12)
    V = torch.randn(k, n + 3, device=eng.device, generator=torch.Generator(device=eng.device).manual_seed(k))[:, :n]
    ref = torch.randn(n, device=eng.device)
    Y = eng.scalars(k)
    Y.set(0, yh)
    o0, o1, o2 = eng.empty(n), eng.empty(n), eng.empty(n)
    P0, P1 = eng.scalars(1024), eng.scalars(1024)
    n0 = eng.gemv_n_err(V, k, Y.ref(0), o0, ref, P0.ref(0), 1024)
    n1 = eng.gemv_n_hosty(V, k, yh, o1, ref, P1.ref(0), 1024)
    assert eng.gemv_n_hosty(V, k, yh, o2) == 0
    assert n0 == n1 and torch.equal(o1, o2)
    if k <= 128:
        assert torch.equal(o0, o1) and np.array_equal(P0.host(0, n0), P1.host(0, n1))
    else:
        assert float((o0 - o1).norm() / o0.norm()) < 3e-7
        assert np.allclose(P0.host(0, n0).sum(), P1.host(0, n1).sum(), rtol=1e-6)
    want = (torch.from_numpy(yh).to(eng.device) @ V.double()).float()
    assert float((o1 - want).norm() / want.norm()) < 3e-7


def test_hybrid_lsqr_host_projected_solve_equals_the_device_one():
    """Automatic lambda: y_k on the host + coefficients in the launch arguments (the default) against the device solve."""
    from trips_py_amd.operators import Radon2DParallel
    from trips_py_amd.solvers import Hybrid_LSQR
    N = 96
    A = Radon2DParallel(N, np.linspace(0, np.pi, 40, endpoint=False))
    rng = np.random.default_rng(10)
    xt = rng.random(N * N).astype(np.float32)
    b = A.apply(torch.from_numpy(xt).cuda())
    e = torch.randn_like(b)
    delta = 0.02 * float(b.norm())
    b = (b + e * (delta / e.norm())).cpu().numpy()
    for reg, kw in (("gcv", {}), ("dp", {"delta": delta})):
        for xtrue in (xt, None):
            x0, i0 = Hybrid_LSQR(A, b, 30, reg, xtrue, host_projected_solve=False, **kw)
            x1, i1 = Hybrid_LSQR(A, b, 30, reg, xtrue, **kw)
            assert i0["regParam_history"] == i1["regParam_history"]
            assert relerr(x1, x0) < 1e-6
            assert all(relerr(p, q) < 1e-6 for p, q in zip(i1["xHistory"], i0["xHistory"]))
            if xtrue is not None:
                assert np.allclose(i0["relError"], i1["relError"], rtol=1e-6)


@pytest.mark.parametrize("reg", ["gcv", "dp"])
def test_hybrid_lsqr_pipelined_loop_equals_the_plain_one(reg):
    """Automatic lambda: steps enqueued ahead, the search on the worker thread and the iterate formed one trip late give the
    same lambdas and the same iterates as the loop that does everything in order (Hybrid_LSQR.py:69-110)."""
    from trips_py_amd.operators import Radon2DParallel
    from trips_py_amd.solvers import Hybrid_LSQR
    N = 96
    A = Radon2DParallel(N, np.linspace(0, np.pi, 40, endpoint=False))
    rng = np.random.default_rng(9)
    xt = rng.random(N * N).astype(np.float32)
    b = A.apply(torch.from_numpy(xt).cuda())
    e = torch.randn_like(b)
    delta = 0.02 * float(b.norm())
    b = (b + e * (delta / e.norm())).cpu().numpy()
    kw = {"delta": delta} if reg == "dp" else {}
    for its in (2, 3, 25):
        x0, i0 = Hybrid_LSQR(A, b, its, reg, xt, async_search=False, steps_ahead=1, **kw)
        for opts in ({}, {"steps_ahead": 2}, {"async_search": False}, {"steps_ahead": 7}):
            x1, i1 = Hybrid_LSQR(A, b, its, reg, xt, **opts, **kw)
            assert i1["regParam_history"] == i0["regParam_history"] and i1["regParam"] == i0["regParam"], opts
            assert np.array_equal(x1, x0) and np.array_equal(i1["relError"], i0["relError"]), opts
            assert len(i1["xHistory"]) == len(i0["xHistory"]) == its - 1
            assert all(np.array_equal(p, q) for p, q in zip(i1["xHistory"], i0["xHistory"]))
    # nobody looks at the intermediate iterates: only the last is formed
    x2, i2 = Hybrid_LSQR(A, b, 12, reg, history=False, **kw)
    x3, i3 = Hybrid_LSQR(A, b, 12, reg, history=False, async_search=False, steps_ahead=1, **kw)
    assert np.array_equal(x2, x3) and i2["regParam_history"] == i3["regParam_history"]


def arnoldi_orthogonality(N=256, steps=100, by_gram=True):
    """max |V^T V - I| (float64, on the host) after `steps` Arnoldi steps on the 9 x 9 sigma-3 blur of an N x N image, and the
    smallest h_{k+1,k} / ||A v_k|| met on the way (how much of A v_k the orthogonalisation removed)."""
    import torch
    from trips_py_amd.krylov import ArnoldiState
    from trips_py_amd.operators import Blur2D
    from trips_py_amd.problems import gauss_psf
    A = Blur2D(gauss_psf((9, 9), (3, 3))[0], N, N)
    dev = A.engine.device
    g = torch.Generator(device=dev).manual_seed(11)
    xt = torch.rand(N * N, device=dev, generator=g)
    b = A.apply(xt)
    b = b + 0.01 * torch.randn(N * N, device=dev, generator=g) * b.norm() / N
    ar = ArnoldiState(A, b, steps, by_gram=by_gram)
    ratio = 1.0
    for _ in range(steps):
        col = ar.step()
        ratio = min(ratio, float(col[-1] / np.linalg.norm(col)))
    V = ar.V.data[:ar.V.k].double()
    G = (V @ V.T).cpu().numpy()
    return float(np.abs(G - np.eye(G.shape[0])).max()), ratio, ar.H()


def test_arnoldi_orthogonality_by_gram_vs_sweeps():
    """Arnoldi's two Gram-Schmidt sweeps as one pair of passes (krylov.GramSchmidtByGram, the default) against sweep by sweep:
    in exact arithmetic the same vector; in fp32 the second sweep of the literal form also removes the rounding error of the first
    subtraction, which matters when A v_k lies almost inside the span (strong cancellation).  Measured on the ill-conditioned blur
    (100 steps, 256^2): both stay far inside north_star's 1e-5, and the Hessenberg matrices agree."""
    loss_g, ratio_g, Hg = arnoldi_orthogonality(by_gram=True)
    loss_s, ratio_s, Hs = arnoldi_orthogonality(by_gram=False)
    print(f"max |V^T V - I|: by Gram {loss_g:.2e} (min h_k+1,k / ||A v_k|| {ratio_g:.3f}), sweep by sweep {loss_s:.2e} ({ratio_s:.3f})")
    assert loss_s < 2e-6, loss_s
    assert loss_g < 1e-5, loss_g
    assert np.abs(Hg - Hs).max() / np.abs(Hs).max() < 1e-5


def test_arnoldi_step_in_one_library_call_equals_the_five_calls():
    """trk_arnoldi_step (apply, the sweep's two passes over the basis, its k x k recurrence, the normalisation — enqueued by one call)
    against the same five calls made from Python: the same basis and the same Hessenberg matrix, bit for bit."""
    from oracle import cpu_ref as O
    from trips_py_amd.operators import Blur2D
    from trips_py_amd.krylov import ArnoldiState
    psf, _ = O.gauss_psf((9, 9), (2, 2))
    N = 96
    A = Blur2D(psf, N, N)
    eng = A.engine
    b = torch.rand(N * N, device=eng.device, generator=torch.Generator(device=eng.device).manual_seed(3))
    one = ArnoldiState(A, b, 24)
    for _ in range(24):
        one.step()
    eng.arnoldi_step, keep = None, eng.arnoldi_step                     # the instance attribute shadows the method: Python's five calls
    try:
        five = ArnoldiState(A, b, 24)
        for _ in range(24):
            five.step()
    finally:
        del eng.arnoldi_step
    assert keep is not None and eng.arnoldi_step is not None
    assert np.array_equal(one.H(), five.H())
    assert torch.equal(one.V.data[:25], five.V.data[:25])
