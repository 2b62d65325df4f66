"""BASELINE configs C3 and C5 at their FULL sizes against the float64 oracle (the oracle needs ~1.5 minutes for both on the
box's host; C2 and C4's operator are covered at full size by test_gpu_cgls.py / test_gpu_fullsize.py).

Bars = north_star's 1e-5 on every solution the solvers converge to, with ONE stated exception: iterates 5..19 of Hybrid-LSQR at
lambda = 1e-2.  There the projected Tikhonov solution x_k = V_k y_k passes through its semi-convergence transient and is
sensitive to the 6e-8 roundings of ANY fp32-stored iteration.  Shown, not argued (round 4, profiles/r04/c3_parity_epilogue.txt):
  * the ENGINE'S OWN arrangement restated in NumPy (Golub-Kahan on unnormalised vectors + damped LSQR's short recurrence:
    tools/fp32_floor.py c3emul) agrees with the oracle to 9e-8 in float64 arithmetic; with nothing changed but a rounding to
    float32 where the engine STORES u, v, w, x it leaves the oracle by 3.6e-4 at iterate 9 (5.0e-4 with the operator's output
    rounded once more), and from iterate 10 on by the very numbers the engine measures (2.9e-4, 1.7e-4, 8.9e-5, 4.1e-5);
  * the engine: 1.06e-3 at iterate 8 — twice that floor.  Round 3 had 1.57e-3 at iterate 7: the half steps were combined in
    trk_axpby's fp32 arithmetic (rounded coefficients, the same perturbation in every entry of a Lanczos vector); combined in
    float64 that iterate is at 3.5e-4.  What is left above the floor is the projector's own fp32 accumulation, which the
    restatement does not model.
From iterate 21 on <= 8.4e-7, 1.8e-7 at step 60.  The projector itself is within 1e-7 of the oracle (test_gpu_radon_accuracy.py)."""
import numpy as np
import pytest

from conftest import bar, relerr

pytestmark = pytest.mark.gpu

# bars: set from measurements on the MI355X (tools/configs_parity.py) and the fp32-storage floor of the oracle itself
# (tools/fp32_floor.py) — see DESIGN.md section 2
C3_BAR = 1e-5              # final iterate, every iterate from step 21 on, relError of those (measured 1.8e-7 ... 8.4e-7)
C3_TRANSIENT_BAR = 2.5e-3  # iterates 1..20: fp32 storage of this arrangement (NumPy restatement: 5.0e-4; engine 1.06e-3; round 3: 1.6e-3)
C5_BAR = 1e-5              # measured 6e-8 ... 1.2e-7 on every iterate, relError 3e-9
C5_RESIDUAL_BAR = 1e-5     # measured 1.9e-7
C4_1024_BAR = 5e-5         # provisional


def c3_numbers(its=20):
    """Parallel-beam 512^2, 180 angles, Hybrid_LSQR (lambda = 1e-2), `its` iterations, 1 % noise: engine vs float64 oracle."""
    from oracle import cpu_ref as O
    from trips_py_amd import solvers as S
    from trips_py_amd.operators import Radon2DParallel
    N, na = 512, 180
    ang = np.linspace(0, np.pi, na, endpoint=False)
    R, Ro = Radon2DParallel(N, ang), O.Radon2D(N, ang)
    ii, jj = np.meshgrid(np.arange(N), np.arange(N), indexing="ij")
    xt = (((ii - 256) / 180.0) ** 2 + ((jj - 256) / 230.0) ** 2 < 1).astype(np.float64) + 0.5 * ((((ii - 300) / 60.0) ** 2 + ((jj - 200) / 40.0) ** 2) < 1)
    rng = np.random.default_rng(5)
    b = Ro @ xt.reshape(-1)
    e = rng.standard_normal(b.size)
    b = (b + 0.01 * np.linalg.norm(b) / np.linalg.norm(e) * e).astype(np.float32).astype(np.float64)
    x, info = S.Hybrid_LSQR(R, b, its, 1e-2, xt.reshape(-1))
    xo, io = O.hybrid_lsqr(Ro, b.reshape(-1, 1), its, 1e-2, xt.reshape(-1, 1))
    return {"its": (info["its"], io["its"]), "x": relerr(x, xo.reshape(-1)),
            "iterates": [relerr(a, c) for a, c in zip(info["xHistory"], io["xHistory"])],
            "relError": float(np.max(np.abs(np.asarray(info["relError"]) / np.asarray(io["relError"]) - 1))),
            "relError_last": float(abs(info["relError"][-1] / io["relError"][-1] - 1))}


def test_c3_tomo512_hybrid_lsqr_fullsize():
    m = c3_numbers(60)
    assert m["its"][0] == m["its"][1] == 59
    assert m["x"] < C3_BAR and m["relError_last"] < C3_BAR, m
    assert max(m["iterates"][20:]) < C3_BAR, m
    assert max(m["iterates"][:20]) < C3_TRANSIENT_BAR, m
    assert m["relError"] < C3_TRANSIENT_BAR, m


def c5_numbers(its=8):
    """Dynamic parallel-beam tomography, 32 frames of 256^2, 15 angles per frame shifted by one degree per frame,
    space-time derivative regulariser, GKS(projection_dim = 3, lambda = 1e-2), `its` iterations — all frames on one GPU."""
    from oracle import cpu_ref as O
    from trips_py_amd import solvers as S
    from trips_py_amd.operators import BlockDiagOp, Radon2DParallel, SpaceTimeDerivative
    N, nt, na = 256, 32, 15
    angs = [np.deg2rad(t + 12.0 * np.arange(na)) for t in range(nt)]
    F = BlockDiagOp([Radon2DParallel(N, a) for a in angs])
    Fo = O.BlockDiag([O.Radon2D(N, a) for a in angs])
    L, Lo = SpaceTimeDerivative(N, nt), O.SpaceTimeDerivative(N, nt)
    frames = []
    for t in range(nt):
        img = np.zeros((N, N))
        img[60 + 2 * t:100 + 2 * t, 40:200] = 1.0
        img[150:190, 30 + 3 * t:90 + 3 * t] = 0.6
        frames.append(img.reshape(-1))
    xt = np.concatenate(frames)
    rng = np.random.default_rng(9)
    b = Fo @ xt
    e = rng.standard_normal(b.size)
    b = (b + 0.01 * np.linalg.norm(b) / np.linalg.norm(e) * e).astype(np.float32).astype(np.float64)
    x, info = S.GKS(F, b, L, 3, its, 1e-2, xt)
    xo, io = O.gks(Fo, b.reshape(-1, 1), Lo, 3, its, 1e-2, xt.reshape(-1, 1))
    return {"x": relerr(x, xo.reshape(-1)), "iterates": [relerr(a, c) for a, c in zip(info["xHistory"], io["xHistory"])],
            "relError": float(np.max(np.abs(np.asarray(info["relError"]) / np.asarray(io["relError"]) - 1))),
            "Residual": float(np.max(np.abs(np.asarray(info["Residual"]) / np.asarray(io["Residual"]) - 1)))}


def test_c5_dynamic_32_frames_gks_fullsize():
    m = c5_numbers()
    assert m["x"] < C5_BAR and max(m["iterates"]) < C5_BAR, m
    assert m["relError"] < C5_BAR, m
    assert m["Residual"] < C5_RESIDUAL_BAR, m


def test_c5_dynamic_32_frames_cgls_fullsize():
    """CGLS on the C5 data (what `bench.py` times per rank count), both arrangements — the recurrence as written and the
    one-all-reduce form of csrc/cgls_sharded.hip — against the float64 oracle at full size: every one of the first 20 iterates
    within 1e-5 (measured 6e-8 ... 1.1e-6).  Not further: past iterate ~22 un-regularised CGLS on this data amplifies any fp32
    rounding to 1e-3 — the oracle does it to itself with fp32-rounded products (tools/fp32_floor.py c5cgls, 1.2e-4 at iterate 24,
    1e-3 from 26 on), and so do both arrangements here (1.8e-3 at iterate 30, tools/cgls_forms_accuracy.py)."""
    from oracle import cpu_ref as O
    from trips_py_amd import solvers as S
    from trips_py_amd.operators import BlockDiagOp, Radon2DParallel
    N, nt, na, its = 256, 32, 15, 20
    angs = [np.deg2rad(t + 12.0 * np.arange(na)) for t in range(nt)]
    F = BlockDiagOp([Radon2DParallel(N, a) for a in angs])
    Fo = O.BlockDiag([O.Radon2D(N, a) for a in angs])
    rng = np.random.default_rng(0)
    xt = rng.random(F.shape[1])
    b = Fo @ xt
    e = rng.standard_normal(b.size)
    b = (b + 0.01 * np.linalg.norm(b) / np.linalg.norm(e) * e).astype(np.float32).astype(np.float64)
    x0 = np.zeros(F.shape[1])
    xo, io = O.cgls(Fo, b.reshape(-1, 1), x0.reshape(-1, 1), its, 0, xt.reshape(-1, 1))
    for kw in ({"one_reduction": False}, {"one_reduction": True}):
        x, info = S.CGLS(F, b, x0, its, 0, xt, **kw)
        d = [relerr(h, ho) for h, ho in zip(info["xHistory"], io["xHistory"])]
        assert max(d) < 1e-5, (kw, d)
        # relResidual is the norm of a DIFFERENCE of consecutive iterates over ||x||: it carries the iterates' 1e-6 many times over
        assert np.allclose(info["relError"], io["relError"], rtol=1e-5) and np.allclose(info["relResidual"], io["relResidual"], rtol=2e-3), kw


def test_c4_mmgks_tv_1024_vs_oracle_and_4096_path_equivalence():
    """C4 (blur + MMGKS with the TV-like l2-l1 functional).  The float64 oracle needs minutes at 4096^2 (two economic QRs of
    16.8 M x k per iteration), so parity against it is taken at 1024^2; at the full 4096^2 the two product forms of the
    engine — A x / L x formed directly (stencil operators) and through the bases AV, LV as the reference writes them —
    must give the same iterates, which exercises every kernel of the iteration (matrix-core Gram, fused Gram-Schmidt step,
    weights, stencils) at full size."""
    import torch
    from oracle import cpu_ref as O
    from trips_py_amd import solvers as S
    from trips_py_amd.operators import Blur2D, FirstDerivative2D
    from trips_py_amd.problems import gauss_psf, synthetic_image
    psf = gauss_psf((9, 9), (3, 3))[0]
    N = 1024
    xt = synthetic_image(N, 3).reshape(-1)
    Ao = O.Blur2D(psf, N, N)
    rng = np.random.default_rng(4)
    b = Ao @ xt
    e = rng.standard_normal(b.size)
    b = (b + 0.01 * np.linalg.norm(b) / np.linalg.norm(e) * e).astype(np.float32).astype(np.float64)
    x, info = S.MMGKS(Blur2D(psf, N, N), b, FirstDerivative2D(N), 2, 1, 3, 6, 1e-2, xt, epsilon=0.1)
    xo, io = O.mmgks(Ao, b.reshape(-1, 1), O.FirstDerivative2D(N), 2, 1, 3, 6, 1e-2, xt.reshape(-1, 1), epsilon=0.1)
    assert info["its"] == io["its"]
    bar("c4_1024_vs_oracle.x", relerr(x, xo.reshape(-1)), C4_1024_BAR)
    assert np.allclose(info["relError"], io["relError"], rtol=2e-4)
    N = 4096
    A, L = Blur2D(psf, N, N), FirstDerivative2D(N)
    dev = A.engine.device
    xt = torch.rand(N * N, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
    bb = A.apply(xt)
    bb = bb + 0.01 * torch.linalg.norm(bb) / (N * 1.0) * torch.randn(N * N, device=dev, generator=torch.Generator(device=dev).manual_seed(2))
    x1, i1 = S.MMGKS(A, bb, L, 2, 1, 3, 14, 1e-2, history=False)
    A.streaming = L.streaming = False
    x2, i2 = S.MMGKS(A, bb, L, 2, 1, 3, 14, 1e-2, history=False)
    assert float(torch.linalg.norm(x1 - x2) / torch.linalg.norm(x2)) < 2e-5
    assert np.allclose(i1["Residual"], i2["Residual"], rtol=2e-3)
