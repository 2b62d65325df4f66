"""The float64 instrument (csrc/ref64.hip; SURVEY section 7 hard part 2): the Golub-Kahan / damped-LSQR chain
instantiated on the element type, and the parallel-beam projector with float64 arithmetic.  It is what the fast path is checked
AGAINST on the hardware, so it is itself pinned to the float64 oracle here — at 1e-12, not at an fp32 bar."""
import numpy as np
import pytest
import torch

from conftest import bar, relerr

pytestmark = pytest.mark.gpu


def _geom(N=48, na=37, nd=61):
    from oracle import cpu_ref as O
    from trips_py_amd.operators import Radon2DParallel
    ang = np.linspace(0.0, np.pi, na, endpoint=False) + 0.0131          # no axis-aligned ray: every tap has a fraction
    ang = np.concatenate((ang, [0.0, np.pi / 2, np.pi / 4, 3 * np.pi / 4, 2.5]))   # ... and the aligned / diagonal ones too
    return Radon2DParallel(N, ang, n_det=nd), O.Radon2D(N, ang, n_det=nd)


def test_float64_projector_is_the_oracle_matrix():
    R, Ro = _geom()
    dev = R.engine.device
    rng = np.random.default_rng(0)
    x, y = rng.standard_normal(R.shape[1]), rng.standard_normal(R.shape[0])
    M = Ro.matrix()
    for tr, v, ref in ((False, x, M @ x), (True, y, M.T @ y)):
        got = R.apply_ref(torch.from_numpy(v).to(dev), tr, "float64").cpu().numpy()
        bar(f"ref64.projector[{'adj' if tr else 'fwd'}]", relerr(got, ref), 1e-12)
        # the product's fixed-point weights, summed in float64: the product OPERATOR (its 2^-24 weight grid), not its rounding
        tab = R.apply_ref(torch.from_numpy(v).to(dev), tr, "tables64").cpu().numpy()
        bar(f"ref64.tables[{'adj' if tr else 'fwd'}]", relerr(tab, ref), 2e-7)
        # fp32 vectors through the same arithmetic: one rounding of the result
        v32 = torch.from_numpy(v.astype(np.float32)).to(dev)
        r32 = (M.T if tr else M) @ v32.cpu().numpy().astype(np.float64)
        got32 = R.apply_ref(v32, tr, "float64").cpu().numpy()
        assert np.max(np.abs(got32 - r32.astype(np.float32))) <= np.spacing(np.abs(r32).max().astype(np.float32))
    # matched pair, exactly: <A x, y> = <x, A^T y> to float64 rounding — for both weight sources
    for wsrc in ("float64", "tables64"):
        Ax = R.apply_ref(torch.from_numpy(x).to(dev), False, wsrc).cpu().numpy()
        Aty = R.apply_ref(torch.from_numpy(y).to(dev), True, wsrc).cpu().numpy()
        assert abs(Ax @ y - x @ Aty) <= 1e-12 * np.linalg.norm(Ax) * np.linalg.norm(y), wsrc


def test_table_weights_are_the_product_kernels_weights():
    """arithmetic 'tables64' = the product's operator with float64 sums: the product's own output is that, rounded the way its
    fp32 partial sums round (a few units in the last place), on any vector — while the float64-geometry operator differs from both
    by the weight grid."""
    R, _ = _geom(64, 45, 64)
    dev = R.engine.device
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn(R.shape[1], device=dev, generator=g)
    y = torch.randn(R.shape[0], device=dev, generator=g)
    for tr, v in ((False, x), (True, y)):
        prod = R.apply(v, transpose=tr).double()
        tab = R.apply_ref(v.double(), tr, "tables64")
        bar(f"ref64.product_vs_tables[{'adj' if tr else 'fwd'}]", float((prod - tab).norm() / tab.norm()), 3e-7)


def test_set_arithmetic_routes_every_entry_point():
    """trk_radon2d_set_arithmetic: apply, the fused half step (trk_op_apply_axpby, riders not taken) and the one-call Golub-Kahan step
    all run the float64-arithmetic kernels; switching back restores the product's bits."""
    from trips_py_amd.engine import Coef
    from trips_py_amd.krylov import GKState
    R, Ro = _geom(64, 45, 64)
    eng, dev = R.engine, R.engine.device
    g = torch.Generator(device=dev).manual_seed(2)
    x = torch.rand(R.shape[1], device=dev, generator=g)
    z = torch.rand(R.shape[0], device=dev, generator=g)
    y0 = R.apply(x).clone()
    R.set_arithmetic("float64")
    y1 = R.apply(x).clone()
    assert torch.equal(y1, R.apply_ref(x, False, "float64")) and not torch.equal(y0, y1)
    # the half step: float64 coefficients and products, one rounding
    out, ss = eng.empty(R.shape[0]), eng.scalars(1)
    R.apply_axpby(x, 0.37, -1.25, z, out, sumsq=ss.ref(0))
    want = (0.37 * y1.double() + (-1.25) * z.double()).float()
    assert torch.equal(out, want)
    assert abs(ss.host(0, 1)[0] / float((want.double() ** 2).sum()) - 1) < 1e-13
    # Golub-Kahan through the library's one-call step == apply + float64 combination, vector for vector
    b = torch.from_numpy((Ro @ x.cpu().numpy().astype(np.float64)).astype(np.float32)).to(dev)
    st = GKState(R, b, 4, normalized=False)
    for _ in range(4):
        st.step(sync=False)
    M = Ro.matrix()
    u = b.cpu().numpy().astype(np.float64)
    v_prev, beta, alpha_prev = None, np.linalg.norm(u), None
    for k in range(2):
        v = (M.T @ u) / beta - (0.0 if v_prev is None else (beta / alpha_prev) * v_prev)
        v = v.astype(np.float32).astype(np.float64)
        assert relerr(st.V.data[k].cpu().numpy(), v) < 2e-7, k
        alpha = np.linalg.norm(v)
        un = ((M @ v) / alpha - (alpha / beta) * u).astype(np.float32).astype(np.float64)
        u, v_prev, alpha_prev, beta = un, v, alpha, np.linalg.norm(un)
    R.set_arithmetic("product")
    assert torch.equal(R.apply(x), y0)


def test_generic_half_step_is_the_fused_epilogues_arithmetic():
    """k_ref_axpby<float> (the kernel whose double instantiation the float64 chain runs) against the product projector's fused half
    step on the same Op(x): identical bits in every element, the norm to float64 rounding."""
    from trips_py_amd import _lib
    from trips_py_amd.engine import Coef
    from trips_py_amd.operators import Radon2DParallel
    R = Radon2DParallel(128, np.linspace(0, np.pi, 60, endpoint=False))
    eng, dev = R.engine, R.engine.device
    g = torch.Generator(device=dev).manual_seed(3)
    for tr in (False, True):
        nin, nout = (R.shape[0], R.shape[1]) if tr else (R.shape[1], R.shape[0])
        x = torch.randn(nin, device=dev, generator=g)
        z = torch.randn(nout, device=dev, generator=g)
        y = R.apply(x, transpose=tr).clone()
        fused, s1 = eng.empty(nout), eng.scalars(2)
        s1.set(0, np.array([3.7, 0.0]))
        a, b = Coef(1.0, den=s1.ref(0), sqrt_den=True), Coef(-0.61, num=s1.ref(0), sqrt_num=True)
        R.apply_axpby(x, a, b, z, fused, transpose=tr, sumsq=s1.ref(1))
        gen, s2 = eng.empty(nout), eng.scalars(1)
        # flags: TRK_SQRT_DEN = 2, TRK_SQRT_NUM = 1 (include/trk.h)
        rc = eng.lib.trk_ref_axpby(4, nout, 1.0, None, s1.ref(0), 2, y.data_ptr(),
                                   -0.61, s1.ref(0), None, 1, z.data_ptr(),
                                   gen.data_ptr(), s2.ref(0), eng.stream())
        _lib.check(rc, "trk_ref_axpby")
        assert torch.equal(fused, gen), tr
        assert abs(s1.host(1, 2)[0] / s2.host(0, 1)[0] - 1) < 1e-13


@pytest.mark.parametrize("storage", ["float64", "float32"])
def test_float64_chain_small_problem_vs_oracle(storage):
    """trk_gk_lsqr_chain on a 64^2 x 45 problem: with float64 vectors every iterate within 1e-10 of the oracle's dense lstsq + V y
    (Hybrid_LSQR.py:104-105) — the engine's arrangement IS the reference's iteration; with fp32 vectors what storage alone costs."""
    from oracle import cpu_ref as O
    from trips_py_amd import solvers as S
    from trips_py_amd.operators import Radon2DParallel
    N, na, its = 64, 45, 25
    ang = np.linspace(0, np.pi, na, endpoint=False)
    R, Ro = Radon2DParallel(N, ang), O.Radon2D(N, ang)
    rng = np.random.default_rng(4)
    xt = np.zeros((N, N))
    xt[16:40, 20:50] = 1.0
    xt[30:55, 8:24] = 0.5
    xt = xt.reshape(-1)
    b = Ro @ xt
    e = rng.standard_normal(b.size)
    b = (b + 0.01 * np.linalg.norm(b) / np.linalg.norm(e) * e).astype(np.float32).astype(np.float64)
    xo, io = O.hybrid_lsqr(Ro, b.reshape(-1, 1), its, 1e-2, xt.reshape(-1, 1))
    x, info = S.Hybrid_LSQR(R, b, its, 1e-2, xt, dtype="float64", storage=storage)
    assert info["its"] == io["its"] and len(info["xHistory"]) == len(io["xHistory"])
    d = [relerr(h, ho) for h, ho in zip(info["xHistory"], io["xHistory"])]
    # What can be held: the first steps (every kernel and every coefficient path of the chain has run by step 3) and the converged
    # end.  In between the iteration ITSELF amplifies any rounding ~6 x per step until the perturbation reaches the distance between
    # consecutive iterates — float64's 1e-16 grows to 4.5e-5 here (7e-6 at iterate 14 of C3, profiles/r05/c3_instrument.txt), fp32
    # storage's 6e-8 to 2.5e-3: the conditioning of un-reorthogonalised Golub-Kahan on this data, not a property of any kernel.
    if storage == "float64":
        bar("ref64.chain64_small.first_steps", max(d[:3]), 1e-12)
        bar("ref64.chain64_small.final", d[-1], 1e-9)
        bar("ref64.chain64_small.transient", max(d), 2e-4)
        bar("ref64.chain64_small.relError_final", abs(info["relError"][-1] / io["relError"][-1] - 1), 1e-9)
    else:
        bar("ref64.chain32_small.first_steps", max(d[:3]), 2e-7)
        bar("ref64.chain32_small.final", d[-1], 1e-5)
        bar("ref64.chain32_small.transient", max(d), 1e-2)
    with pytest.raises(NotImplementedError):
        S.Hybrid_LSQR(R, b, its, "gcv", xt, dtype="float64")
