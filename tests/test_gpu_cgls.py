"""GPU parity of trips_py_amd.solvers.CGLS with the reference CGLS outputs (tests/golden/cgls_*.npz) and the oracle.
Bar: final x <= 1e-5 relative (fp32 engine vs fp64 reference), identical `its`, info lists to 1e-4."""
import numpy as np
import pytest
import torch

from conftest import load_golden, relerr

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["cgls_blur64_x0zero", "cgls_blur64_x0ATb", "cgls_blur64_tol"])
def test_cgls_golden(name):
    from trips_py_amd.operators import Blur2D
    from trips_py_amd.solvers import CGLS
    g = load_golden(name)
    N = int(g["N"])
    A = Blur2D(g["psf"], N, N)
    xt = g["x_true"] if "x_true" in g else None
    x, info = CGLS(A, g["b"], g["x0"], int(g["max_iter"]), float(g["tol"]), x_true=xt)
    assert info["its"] == int(g["its"])
    assert isinstance(x, np.ndarray) and x.shape == (N * N, 1)
    assert relerr(x, g["x"]) < 1e-5, relerr(x, g["x"])
    assert np.allclose(info["relResidual"], g["relResidual"], rtol=1e-4)
    assert len(info["xHistory"]) == info["its"] and info["regParam"] == []
    if xt is not None:
        assert np.allclose(info["relError"], g["relError"], rtol=1e-4)
    if "x_it10" in g:
        assert relerr(info["xHistory"][9], g["x_it10"]) < 1e-5
        assert relerr(info["xHistory"][0], g["x_it1"]) < 1e-5


def test_cgls_config_c1_deblur1d():
    """BASELINE config C1 (1-D Gaussian deblur n=256, CGLS 50 its) through the engine."""
    from trips_py_amd.operators import Blur1D
    from trips_py_amd.solvers import CGLS
    g = load_golden("deblur1d_cgls_n256")
    n = int(g["n"])
    A = Blur1D(g["psf"])
    x, info = CGLS(A, g["b"], np.zeros((n, 1)), int(g["max_iter"]), float(g["tol"]), x_true=g["x_true"])
    assert info["its"] == int(g["its"])
    # the 1-D problem is numerically rank deficient: late CG iterates amplify fp32 rounding; early ones must agree
    assert np.allclose(info["relError"][:15], g["relError"][:15], rtol=1e-3)
    assert info["relError"][-1] < 2 * g["relError"][-1] + 1e-3


def test_cgls_config_c2_512_against_oracle_and_history_off():
    """BASELINE config C2 (blur 512^2 fp32, CGLS 100 its) vs the float64 oracle on identical inputs."""
    from oracle import cpu_ref as O
    from trips_py_amd.operators import Blur2D
    from trips_py_amd.problems import add_noise, gauss_psf, synthetic_image
    from trips_py_amd.solvers import CGLS
    N, its = 512, 100
    psf, _ = gauss_psf((9, 9), (3, 3))
    xt = synthetic_image(N, 0).reshape(-1, 1)
    Ao = O.Blur2D(psf, N, N)
    b, _ = add_noise(Ao @ xt, 0.01, 1)
    xo, io = O.cgls(Ao, b, np.zeros((N * N, 1)), its, 0, x_true=xt)
    A = Blur2D(psf, N, N)
    x, info = CGLS(A, b, np.zeros((N * N, 1)), its, 0, x_true=xt, history=False)
    assert info["its"] == its and info["xHistory"] == []
    assert relerr(x, xo) < 1e-5, relerr(x, xo)
    assert np.allclose(info["relError"], io["relError"], rtol=1e-4)
    assert np.allclose(info["relResidual"], io["relResidual"], rtol=1e-3)
    # torch in -> torch out
    bt = torch.from_numpy(b.astype(np.float32)).to(A.engine.device)
    x2, info2 = CGLS(A, bt, torch.zeros(N * N, device=A.engine.device), 10, 0)
    assert isinstance(x2, torch.Tensor) and x2.shape == (N * N, 1) and len(info2["xHistory"]) == 10
    assert relerr(x2.cpu().numpy(), io["xHistory"][9]) < 1e-5


@pytest.mark.parametrize("N", [64, 256, 1000])
def test_fused_and_unfused_cgls_agree(N):
    """The three-launch fused path (trk_op_apply_fused / trk_cgls_x_update) against the generic path (four launches with raw partials)."""
    from trips_py_amd.operators import Blur2D
    from trips_py_amd.problems import add_noise, gauss_psf, synthetic_image
    from trips_py_amd.solvers import CGLS, CGLSRunFused
    psf, _ = gauss_psf((9, 9), (3, 3))
    A = Blur2D(psf, N, N)
    assert CGLSRunFused.usable(A, A.engine)
    xt = synthetic_image(N, 3).reshape(-1, 1)
    b, _ = add_noise(A @ xt, 0.01, 4)
    x0 = np.zeros((N * N, 1))
    xf, inf_f = CGLS(A, b, x0, 25, 0, x_true=xt)
    xu, inf_u = CGLS(A, b, x0, 25, 0, x_true=xt, fused=False)
    assert relerr(xf, xu) < 2e-6
    assert np.allclose(inf_f["relResidual"], inf_u["relResidual"], rtol=1e-5)
    assert np.allclose(inf_f["relError"], inf_u["relError"], rtol=1e-5)
    assert relerr(inf_f["xHistory"][4], inf_u["xHistory"][4]) < 1e-6
    # non-zero start and no x_true, history off
    xs = A.T @ b
    xf, _ = CGLS(A, b, xs, 8, 0, history=False)
    xu, _ = CGLS(A, b, xs, 8, 0, history=False, fused=False)
    assert relerr(xf, xu) < 2e-6


@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("history", [False, True])
def test_c_loop_equals_stepwise(fused, history):
    """trk_cgls_iterate / trk_cgls_iterate_fused (a stretch of iterations driven from C) enqueue exactly the launches of
    step(): iterates and scalar rows are bit-identical, also when a stretch continues a stepwise start."""
    from trips_py_amd.operators import Blur2D
    from trips_py_amd.problems import gauss_psf
    from trips_py_amd.solvers import CGLSRun, CGLSRunFused
    N, iters = 96, 12
    A = Blur2D(gauss_psf((9, 9), (3, 3))[0], N, N)
    rng = np.random.default_rng(5)
    b, x0, xt = rng.standard_normal(N * N), np.zeros(N * N), rng.standard_normal(N * N)

    def make():
        return (CGLSRunFused(A, b, x0, iters, xt, history) if fused
                else CGLSRun(A, b, x0, iters, xt, history, defer_norms=True))
    r1 = make()
    for _ in range(iters):
        r1.step()
    r2 = make()
    r2.step()
    r2.step()
    r2.run(4)
    r2.run(1000)                                   # clipped to max_iter
    assert r2.k == r1.k == iters
    g1, rows1 = r1.rows()
    g2, rows2 = r2.rows()
    assert g1 == g2 and np.array_equal(rows1, rows2)
    assert torch.equal(r1.x_cur, r2.x_cur)
    if history:
        assert torch.equal(r1.X, r2.X)


@pytest.mark.parametrize("grouping", [0, 1])
@pytest.mark.parametrize("N", [64, 520])
def test_raw_partials_form_equals_finalized_form(N, grouping):
    """CGLSRun with the operator's norms left as block partials (trk_op_apply_fused(x2=NULL), trk_cgls_update_xr_src,
    trk_cgls_p_update: 4 launches) against the same recurrence with finished scalars (6 launches + norms on the fly)."""
    from trips_py_amd.operators import Blur2D
    from trips_py_amd.problems import gauss_psf
    from trips_py_amd.solvers import CGLSRun
    A = Blur2D(gauss_psf((9, 9), (3, 3))[0], N, N)
    rng = np.random.default_rng(N)
    b, x0, xt = rng.standard_normal(N * N), np.zeros(N * N), rng.standard_normal(N * N)
    raw = CGLSRun(A, b, x0, 15, x_true=xt, history=True, defer_norms=True, grouping=grouping)
    fin = CGLSRun(A, b, x0, 15, x_true=xt, history=True, defer_norms=False)
    assert raw.raw and raw.grouping == grouping and not fin.raw
    # the C loop and the stepwise driver enqueue the same launches: identical bits
    twin = CGLSRun(A, b, x0, 15, x_true=xt, history=True, defer_norms=True, grouping=grouping)
    for _ in range(15):
        twin.step()
    raw.run(15)
    assert np.array_equal(raw.rows()[1], twin.rows()[1]) and bool((raw.X[14] == twin.X[14]).all())
    for _ in range(15):
        fin.step()
    g0r, Rr = raw.rows()
    g0f, Rf = fin.rows()
    assert g0r == g0f
    assert np.allclose(Rr, Rf, rtol=2e-6, atol=0)
    xr, xf = raw.X[14].cpu().numpy(), fin.X[14].cpu().numpy()
    assert relerr(xr, xf) < 2e-6


@pytest.mark.parametrize("nx,ny,dim,spread", [(64, 64, (9, 9), (3, 3)), (96, 80, (9, 9), (3, 3)), (40, 100, (5, 5), (1, 2)),
                                             (512, 512, (9, 9), (3, 3)), (17, 16, (3, 3), (1, 1)), (1000, 1000, (7, 7), (2, 2)),
                                             # short axes where a partial last tile's halo loads fold past the far edge
                                             # (n < (i0 + 40) / 2: the fold is clamped, cgls_tiled.hip refl())
                                             (16, 16, (9, 9), (3, 3)), (19, 16, (9, 9), (3, 3)), (33, 20, (9, 9), (3, 3)),
                                             (35, 36, (9, 9), (3, 3)), (36, 16, (5, 5), (1, 1))])
@pytest.mark.parametrize("form", [1, 2])
def test_tiled_two_launch_cgls_equals_the_streaming_form(nx, ny, dim, spread, form):
    """trk_cgls_iterate_tiled (form 1: a workgroup per 32 x 32 tile recomputing its halo of p and A p, four blurs per iteration) and
    trk_cgls_iterate_tiled2 (form 2: w = A p kept as a vector and advanced as A t + beta w, two blurs) against the four-launch
    streaming form on the same problem: sizes that are not multiples of the tile, non-square images, PSFs smaller than 9 x 9 and
    rectangular, with and without x_true / history, zero and non-zero x0."""
    import torch
    from trips_py_amd.operators import Blur2D
    from trips_py_amd.problems import gauss_psf
    from trips_py_amd.solvers import CGLS
    from trips_py_amd.solvers.CGLS import CGLSRunFused
    A = Blur2D(gauss_psf(dim, spread)[0], nx, ny)
    assert CGLSRunFused.tiled_usable(A, A.engine)
    dev = A.engine.device
    n = nx * ny
    xt = torch.rand(n, device=dev, generator=torch.Generator(device=dev).manual_seed(7))
    b = A.apply(xt)
    b = b + 0.01 * torch.randn(n, device=dev, generator=torch.Generator(device=dev).manual_seed(8)) * b.norm() / n ** 0.5
    x0 = torch.zeros(n, device=dev)
    # images of a few hundred pixels under a 9 x 9 blur are so ill-conditioned that fp32 CGLS itself falls apart after a handful
    # of iterations (16 x 16: the two forms' roundings differ by 3e-6 at iterate 6 and grow ninefold per iteration): what the short
    # axes are here for — the clamped halo folds — is exercised from the first iteration on, so they run 5
    its = 25 if n >= 4096 else 5
    xa, ia = CGLS(A, b, x0, its, 0, xt, tiled=False, fused=False)
    xb, ib = CGLS(A, b, x0, its, 0, xt, tiled=form)
    for k in range(its):
        ra, rb = ia["xHistory"][k].reshape(-1), ib["xHistory"][k].reshape(-1)
        assert float(torch.linalg.norm(ra - rb) / torch.linalg.norm(ra)) < 2e-6, k
    assert np.allclose(ia["relError"], ib["relError"], rtol=1e-5) and np.allclose(ia["relResidual"], ib["relResidual"], rtol=1e-5)
    xc, ic = CGLS(A, b, x0, its, 0, tiled=form, history=False)
    assert float(torch.linalg.norm(xc - xb) / torch.linalg.norm(xb)) == 0.0 and ic["xHistory"] == []
    # a start that is not zero (r0 = b - A x0; form 2's w buffer holds A x0 when the first iteration begins), stepwise driver
    x1 = 0.5 * torch.rand(n, device=dev, generator=torch.Generator(device=dev).manual_seed(9))
    xd, idd = CGLS(A, b, x1, 5, 0, xt, tiled=False, fused=False)
    run = CGLSRunFused(A, b, x1, 5, xt, True, tiled=form)
    for _ in range(5):
        run.step()
    g0, rows = run.rows()
    assert float(torch.linalg.norm(run.x_cur - xd.reshape(-1)) / torch.linalg.norm(xd)) < 2e-6
    xe, ie = CGLS(A, b, x1, 5, 0, xt, tiled=form)
    assert float(torch.linalg.norm(xe.reshape(-1) - run.x_cur) / torch.linalg.norm(xe)) == 0.0      # C loop == stepwise
    assert np.allclose(idd["relError"], ie["relError"], rtol=1e-5)


@pytest.mark.parametrize("form", [1, 2])
def test_tiled_cgls_against_the_reference_golden(form):
    from trips_py_amd.solvers import CGLS
    from trips_py_amd.operators import Blur2D
    g = load_golden("cgls_blur64_x0zero")
    A = Blur2D(g["psf"], int(g["N"]), int(g["N"]))
    x, info = CGLS(A, g["b"], g["x0"], int(g["max_iter"]), 0, g["x_true"], tiled=form)
    assert relerr(x, g["x"]) < 1e-5 and np.allclose(info["relError"], g["relError"], rtol=1e-4)


@pytest.mark.parametrize("case", ["static", "dynamic", "large_tiles"])
def test_radon_cgls_raw_partials_form_equals_finalized_form(case):
    """The projector's one-operand fused apply (trk_op_apply_fused with x2 = NULL: ||A p||^2, ||A^T r||^2 left as the block
    partials of the band reduction / tile gather, added up by the CGLS update kernels — four launches, no reduction launch)
    against the same recurrence with finished scalars; the C loop against the stepwise driver; trk_op_fused_caps = 2."""
    from trips_py_amd.operators import BlockDiagOp, Radon2DParallel
    from trips_py_amd.solvers import CGLSRun
    from trips_py_amd.solvers.CGLS import CGLSRunFused
    if case == "static":
        A = Radon2DParallel(128, np.linspace(0, np.pi, 60, endpoint=False))
    elif case == "dynamic":
        A = BlockDiagOp([Radon2DParallel(64, np.deg2rad(5.0 * t + 12.0 * np.arange(15))) for t in range(4)])
    else:                        # 1100^2: more adjoint tiles than CGLSRun.PCAP partial slots -> one finished value instead
        A = Radon2DParallel(2200, np.linspace(0, np.pi, 6, endpoint=False))
    eng = A.engine
    assert eng.op_can_fuse(A._h) == 2 and not CGLSRunFused.usable(A, eng)
    m, n = A.shape
    g = torch.Generator(device=eng.device).manual_seed(9)
    xt = torch.rand(n, device=eng.device, generator=g)
    b = A.apply(xt)
    b = b + 0.02 * torch.randn(m, device=eng.device, generator=g) * b.norm() / m ** 0.5
    x0 = torch.zeros(n, device=eng.device)
    its = 12
    raw = CGLSRun(A, b, x0, its, x_true=xt, history=True, defer_norms=True)
    fin = CGLSRun(A, b, x0, its, x_true=xt, history=True, defer_norms=False)
    twin = CGLSRun(A, b, x0, its, x_true=xt, history=True, defer_norms=True)
    assert raw.raw and not fin.raw
    raw.run(its)
    for _ in range(its):
        fin.step()
        twin.step()
    assert np.array_equal(raw.rows()[1], twin.rows()[1]) and torch.equal(raw.X[its - 1], twin.X[its - 1])
    g0r, Rr = raw.rows()
    g0f, Rf = fin.rows()
    # (two fp32 recurrences whose scalars are summed in different orders: every iterate agrees to fp32 rounding; delta, gamma
    #  and ||dx||^2 can pass through stretches where CG amplifies that rounding to 1e-4 ... 1e-3 of their current — by then tiny —
    #  values, so the scalar rows are compared where they are well conditioned: the first four iterations)
    assert np.isclose(g0r, g0f, rtol=1e-12) and np.allclose(Rr[:4], Rf[:4], rtol=1e-6, atol=0)
    for k in range(its):
        assert relerr(raw.X[k].cpu().numpy(), fin.X[k].cpu().numpy()) < 5e-6, k
    assert relerr(raw.X[its - 1].cpu().numpy(), fin.X[its - 1].cpu().numpy()) < 5e-6


def test_carried_vectors_of_the_recurrence_forms_do_not_drift_over_long_solves():
    """the default arrangements advance w = A p (and r) by fp32 recurrences and never refresh them from a real product
    (tiled form 2 on one rank: w_k = A t + beta w_{k-1}; the one-all-reduce form on ranks: w_k = q + beta w_{k-1}).  Over the longest
    solves anything here runs (bench: 100 iterations; this test: 1000 and 400) the carried vectors stay at rounding distance from
    b - A x and A p — measured on the MI355X (tools/cgls_drift.py): r 3.4e-7 |b| after 1000 iterations (the form that forms A p every
    iteration: 2.7e-7), w 2.8e-6 |A p|; sharded form after 400: 7.7e-8 / 8.3e-7.  No periodic refresh is needed below these counts;
    `tiled=1` / `one_reduction=False` select the arrangements that do refresh."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import cgls_drift as D
    dr, dw = D.blur_case(256, 1000, 2)
    assert dr < 2e-6 and dw < 2e-5, (dr, dw)
    dr, dw = D.sharded_case(400)
    assert dr < 1e-6 and dw < 1e-5, (dr, dw)
