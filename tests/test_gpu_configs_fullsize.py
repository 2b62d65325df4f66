"""BASELINE configs C3 and C5 at their FULL sizes against the float64 oracle (the oracle needs ~1.5 minutes for both on the
box's host; C2 and C4's operator are covered at full size by test_gpu_cgls.py / test_gpu_fullsize.py).

Bar = north_star's 1e-5 on every solution the solvers converge to.  ONE stated exception, now a CRITERION instead of a constant
(round 5): iterates 5..20 of Hybrid-LSQR at lambda = 1e-2 on C3, where the projected solution x_k = V_k y_k passes through its
semi-convergence transient.  What the float64 instrument (csrc/ref64.hip, tools/r05_c3_instrument.py -> profiles/r05/
c3_instrument.txt) measured there on the MI355X, the same 100 iterations in twelve arithmetics:
  * un-reorthogonalised Golub-Kahan on this data amplifies ANY perturbation ~6.5 x per iteration from iterate 4 on, until it
    reaches the distance between CONSECUTIVE iterates of the iteration itself (the perturbed run is then a fraction of a step
    ahead of or behind the exact one), and follows that distance down as the iteration converges;
  * float64 vectors, float64 projector: 4e-15 at iterate 1 -> 7.2e-6 at iterate 14 -> 5e-15 from iterate 31 on.  Float64's own
    roundings, amplified eleven orders of magnitude and gone again: the transient is the PROBLEM's conditioning;
  * fp32 vectors with an exact (float64-arithmetic) projector: 5.0e-4 at iterate 9 (NumPy restatement of round 4: 5.0e-4);
    with the product's table weights: 2.9e-4; with fp32 partial sums emulated in either or both directions: 4.4e-4, 1.07e-3,
    1.06e-3, 2.9e-4, 5.6e-4 — the peak is wherever the growth meets the envelope, iterate 8 (1.06e-3 = the envelope there, three
    arithmetics to three digits) or iterate 9;  the product: 1.06e-3 at iterate 8.
So every iterate is held to  max(1e-5, the larger of the reference's own steps into and out of that iterate) — "within 1e-5, or
closer to the reference's iterate than the reference's neighbouring iterates are" — and the float64 instantiation of the chain to
1e-9 outside the transient and to the same envelope inside it.

Round 6 (profiles/r06/c3_instrument.txt): the product's iterates 5-8 sat 44-90 x above the fp32-storage floor (chain32/tab), and the
instrument's emulations said why — not the length of the kernels' fp32 sums (both directions summed in fp32 at the product's old
cadence: on the floor) but a MISMATCH between A and A^T that is the same for every entry of an angle: any arithmetic that rounded the
angle's weight in one direction only sat where the product sat.  The product's mismatch of that kind was the adjoint's neighbour-weight
constant 1 - |inv|: formed from the fp32 inv (off by up to 7e-8 for every neighbour weight of the angle), then rounded again inside the
fp32 FMA that forms the weights.  Now it lies on the 2^-24 grid of the fixed-point tables (the FMA is exact) and which grid neighbour
an angle takes is chosen by error diffusion over the angles (radon2d.hip, radon_create_impl): iterates 5 / 6 / 7 / 8 at 3.2e-7 /
2.0e-6 / 1.2e-5 / 6.9e-5 against the floor's 1.2e-7 / 6.5e-7 / 3.9e-6 / 2.0e-5 (round 5: 6.4e-6 / 4.5e-5 / 3.6e-4 / 1.06e-3).  The
criterion is therefore tightened: 1e-5 outright on iterates 1-6 and from 21 on, and inside the transient within SIX times the floor
measured in the same test (and still inside round 5's envelope)."""
import functools

import numpy as np
import pytest

from conftest import bar, relerr

pytestmark = pytest.mark.gpu

# bars: set from measurements on the MI355X (tools/r05_c3_instrument.py, tools/configs_parity.py)
C3_BAR = 1e-5              # final iterate, the first six, every iterate from step 21 on (measured 8e-8 ... 2.0e-6)
C3_ITS = 100
C5_BAR = 1e-5              # measured 6e-8 ... 1.2e-7 on every iterate, relError 3e-9
C5_RESIDUAL_BAR = 1e-5     # measured 1.9e-7
C4_1024_BAR = 1e-5         # measured 1.2e-7


@functools.lru_cache(maxsize=1)
def c3_problem():
    """Parallel-beam 512^2, 180 angles, 1 % noise, and the oracle's C3_ITS iterations of Hybrid_LSQR(lambda = 1e-2) on it."""
    from oracle import cpu_ref as O
    N, na = 512, 180
    ang = np.linspace(0, np.pi, na, endpoint=False)
    Ro = O.Radon2D(N, ang)
    ii, jj = np.meshgrid(np.arange(N), np.arange(N), indexing="ij")
    xt = (((ii - 256) / 180.0) ** 2 + ((jj - 256) / 230.0) ** 2 < 1).astype(np.float64) + 0.5 * ((((ii - 300) / 60.0) ** 2 + ((jj - 200) / 40.0) ** 2) < 1)
    rng = np.random.default_rng(5)
    b = Ro @ xt.reshape(-1)
    e = rng.standard_normal(b.size)
    b = (b + 0.01 * np.linalg.norm(b) / np.linalg.norm(e) * e).astype(np.float32).astype(np.float64)
    xo, io = O.hybrid_lsqr(Ro, b.reshape(-1, 1), C3_ITS, 1e-2, xt.reshape(-1, 1))
    H = [h.reshape(-1) for h in io["xHistory"]]
    step = [float(np.linalg.norm(H[k + 1] - H[k]) / np.linalg.norm(H[k])) for k in range(len(H) - 1)]
    # envelope[k]: the larger of the reference's steps into and out of iterate k
    env = [max(step[max(k - 1, 0)], step[min(k, len(step) - 1)]) for k in range(len(H))]
    return N, ang, xt.reshape(-1), b, H, io, env


def c3_numbers(solve):
    N, ang, xt, b, H, io, env = c3_problem()
    x, info = solve(N, ang, xt, b)
    d = [relerr(a, c) for a, c in zip(info["xHistory"], H)]
    return {"its": (info["its"], io["its"]), "x": relerr(x, H[-1]), "iterates": d, "envelope": env,
            "relError": [abs(a / c - 1) for a, c in zip(info["relError"], io["relError"])]}


def test_c3_tomo512_hybrid_lsqr_fullsize():
    from trips_py_amd import solvers as S
    from trips_py_amd.operators import Radon2DParallel
    m = c3_numbers(lambda N, ang, xt, b: S.Hybrid_LSQR(Radon2DParallel(N, ang), b, C3_ITS, 1e-2, xt))
    assert m["its"][0] == m["its"][1] == C3_ITS - 1
    d, env = m["iterates"], m["envelope"]
    bar("c3.final_x", m["x"], C3_BAR)
    bar("c3.relError_final", m["relError"][-1], C3_BAR)
    bar("c3.first_six", max(d[:6]), C3_BAR)                                     # round 6 (rounds 3-5: the first four; iterate 6 was 4.5e-5)
    bar("c3.from_21_on", max(d[20:]), C3_BAR)
    bar("c3.transient_max", max(d[:20]), 1e-3)                                  # recorded: 6.3e-4 at iterate 9 (round 5: 1.06e-3 at iterate 8)
    # THE criterion, iterate by iterate (round 6): within 1e-5, or within SIX times what fp32 storage alone costs that iterate — the
    # same arrangement on fp32 vectors with the float64-arithmetic projector on the product's table weights (chain32/tab of
    # profiles/r06/c3_instrument.txt), run here beside the product — and never further than the oracle's own steps (round 5's envelope).
    # Measured product / floor: <= 3.4 (iterates 5-9: 3.2e-7 / 2.0e-6 / 1.2e-5 / 6.9e-5 / 6.3e-4 against 1.2e-7 / 6.5e-7 / 3.9e-6 / 2.0e-5 /
    # 2.9e-4); round 5's kernels sat 44-90 x above it there and passed through the envelope alone.
    floor = c3_numbers(lambda N, ang, xt, b: S.Hybrid_LSQR(Radon2DParallel(N, ang), b, C3_ITS, 1e-2, xt, dtype="float64", storage="float32",
                                                           weights="tables64"))["iterates"]
    # (the ratio where it matters: the iterates that are beyond 1e-5 at all — 7 to 14; a second, tiny bump at iterates 16-19, 1e-7 ...
    #  2e-6 in every arithmetic, is pure luck of the roundings: 0.2e-6 ... 2e-6 across the instrument's columns)
    bar("c3.over_the_fp32_storage_floor", max([d[k] / floor[k] for k in range(20) if d[k] > C3_BAR] + [0.0]), 6.0)
    for k in range(len(d)):
        lim = max(C3_BAR, min(env[k], 6.0 * floor[k]))
        assert d[k] < lim, (k + 1, d[k], floor[k], env[k])
        assert m["relError"][k] < max(C3_BAR, env[k]), (k + 1, m["relError"][k], env[k])


def test_c3_float64_instantiation_of_the_chain():
    """The engine's arrangement on float64 vectors with the float64-arithmetic projector (trk_gk_lsqr_chain): exact to 1e-8 wherever
    float64 itself can be (iterates 1-8 and 21-100), and inside the transient no further from the oracle than the oracle's own
    steps — measured 7.2e-6 at iterate 14, i.e. float64's roundings amplified by 1e11 (module docstring)."""
    from trips_py_amd import solvers as S
    from trips_py_amd.operators import Radon2DParallel
    m = c3_numbers(lambda N, ang, xt, b: S.Hybrid_LSQR(Radon2DParallel(N, ang), b, C3_ITS, 1e-2, xt, dtype="float64"))
    d, env = m["iterates"], m["envelope"]
    # (the oracle's own roundings differ from box to box with the host BLAS's thread count — both sides of these comparisons are
    #  float64 runs of an iteration that amplifies 1e-16 by up to 1e11: two visits measured 4.1e-11 / 5.1e-10, 1.0e-9 / 2.3e-9, 7.2e-6 / 1.5e-5)
    bar("c3.chain64.iterates_1_to_8", max(d[:8]), 1e-8)
    bar("c3.chain64.from_21_on", max(d[20:]), 2e-8)
    bar("c3.chain64.transient_max", max(d[8:20]), 1e-4)
    assert all(d[k] < max(2e-8, env[k]) for k in range(len(d))), [(k + 1, d[k], env[k]) for k in range(len(d)) if d[k] >= max(2e-8, env[k])]
    # fp32 vectors, everything else as above: what STORAGE alone costs — the floor no fp32 engine can be under
    m32 = c3_numbers(lambda N, ang, xt, b: S.Hybrid_LSQR(Radon2DParallel(N, ang), b, C3_ITS, 1e-2, xt, dtype="float64", storage="float32"))
    bar("c3.chain32.transient_max", max(m32["iterates"][:20]), 2e-3)                   # measured 5.0e-4
    bar("c3.chain32.from_21_on", max(m32["iterates"][20:]), C3_BAR)                    # measured 4.2e-7


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
