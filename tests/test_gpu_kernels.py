"""GPU parity of the vector kernels against NumPy float64 on the same seeded inputs.
All calls go through the C ABI (libtrk.so) via HipEngine.  fp32 storage / fp64 accumulation: tolerances are stated per test."""
import numpy as np
import pytest
import torch

from conftest import bar, relerr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from trips_py_amd.engine import default_engine
    return default_engine()


def dev(eng, a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(eng.device)


def f32(a):
    return np.asarray(a, dtype=np.float32).astype(np.float64)


SIZES = [1, 3, 64, 1000, 4097, 262144, 1_000_003]


@pytest.mark.parametrize("n", SIZES)
def test_dot_nrm2_diff(eng, n):
    rng = np.random.default_rng(n)
    x, y = rng.standard_normal(n), rng.standard_normal(n)
    dx, dy = dev(eng, x), dev(eng, y)
    S = eng.scalars(3)
    eng.dot(dx, dy, S[0:1])
    eng.nrm2sq(dx, S[1:2])
    eng.diff_nrm2sq(dx, dy, S[2:3])
    got = eng.to_host(S)
    x32, y32 = f32(x), f32(y)
    # fp64 accumulation of fp32 products: agreement with NumPy float64 on the same fp32 inputs to ~1e-13
    assert np.isclose(got[0], np.dot(x32, y32), rtol=1e-11, atol=1e-9)
    assert np.isclose(got[1], np.dot(x32, x32), rtol=1e-12)
    assert np.isclose(got[2], np.sum((x32 - y32) ** 2), rtol=1e-12)


def test_reductions_misaligned_and_empty(eng):
    rng = np.random.default_rng(5)
    x = rng.standard_normal(10_001)
    big = dev(eng, x)
    S = eng.scalars(2)
    v = big[1:]                               # 4-byte aligned only -> scalar path
    eng.nrm2sq(v, S[0:1])
    eng.dot(v, v, S[1:2])
    got = eng.to_host(S)
    ref = np.dot(f32(x)[1:], f32(x)[1:])
    assert np.isclose(got[0], ref, rtol=1e-12) and np.isclose(got[1], ref, rtol=1e-12)
    e = eng.empty(0)
    eng.nrm2sq(e, S[0:1])
    assert eng.to_host(S)[0] == 0.0


@pytest.mark.parametrize("n", [5, 4096, 100_003])
def test_axpby_with_device_coefficients(eng, n):
    from trips_py_amd.engine import Coef
    rng = np.random.default_rng(n)
    x, y = rng.standard_normal(n), rng.standard_normal(n)
    dx, dy, out = dev(eng, x), dev(eng, y), eng.empty(n)
    S = eng.scalars(4)
    S[0], S[1] = 9.0, 4.0
    # out = (2 * sqrt(S0) / S1) x + (-1/sqrt(S1)) y
    eng.axpby(Coef(2.0, num=S[0:1], den=S[1:2], sqrt_num=True), dx, Coef(-1.0, den=S[1:2], sqrt_den=True), dy, out, sumsq=S[2:3])
    ref = (f32(x) * np.float32(1.5) + f32(y) * np.float32(-0.5)).astype(np.float32)
    assert np.allclose(out.cpu().numpy(), ref, rtol=2e-7, atol=1e-7)
    assert np.isclose(eng.to_host(S)[2], np.sum(out.cpu().numpy().astype(np.float64) ** 2), rtol=1e-12)
    # in place, y = None (scaling)
    eng.scale(Coef(1.0, den=S[0:1], sqrt_den=True), dx, dx)
    assert np.allclose(dx.cpu().numpy(), f32(x) / 3.0, rtol=2e-7)
    # aliasing out == y
    eng.axpby(1.0, out, 2.0, dy, dy)
    assert np.allclose(dy.cpu().numpy(), ref + 2 * f32(y), rtol=3e-7, atol=1e-6)


@pytest.mark.parametrize("n", [1, 5, 4096, 100_003, 1_000_001])
def test_scale_with_the_dot_of_its_output(eng, n):
    """trk_scale_dot: out = a x (in place as the solvers call it) and <out, z> in one pass — the same floats as scale, the dot of
    exactly those floats (MMGKS.py:121-123 with the new entry v . A^T b of the projected right-hand side)."""
    from trips_py_amd.engine import Coef
    rng = np.random.default_rng(n)
    x, z = rng.standard_normal(n), rng.standard_normal(n)
    dx, dz = dev(eng, x), dev(eng, z)
    for view in (slice(None), slice(1, None)):                          # 16-byte aligned and not
        a, b = dx[view].clone() if view.start else dx.clone(), dz[view]
        if view.start:
            buf = eng.empty(n + 3)
            a = buf[1:n]
            a.copy_(dx[view])
        if a.numel() == 0:
            continue
        S = eng.scalars(2)
        S[0] = 16.0
        want = a.clone()
        eng.scale(Coef(1.0, den=S[0:1], sqrt_den=True), want, want)
        eng.scale_dot(Coef(1.0, den=S[0:1], sqrt_den=True), a, a, b, S[1:2])
        assert torch.equal(a, want)
        ref = float(np.dot(a.cpu().numpy().astype(np.float64), b.cpu().numpy().astype(np.float64)))
        got = float(eng.to_host(S)[1])
        assert abs(got - ref) <= 1e-12 * float(np.linalg.norm(a.cpu().numpy().astype(np.float64)) * np.linalg.norm(b.cpu().numpy().astype(np.float64))) + 1e-300


@pytest.mark.parametrize("p,eps", [(1.0, 0.1), (2.0, 0.1), (0.5, 0.01), (1.5, 0.3)])
def test_mm_weights_and_mul(eng, p, eps):
    rng = np.random.default_rng(3)
    n = 50_001
    x, y = rng.standard_normal(n), rng.standard_normal(n)
    dx, dy, out = dev(eng, x), dev(eng, y), eng.empty(n)
    eng.mm_weights(dx, dy, eps, p, out)
    v = (f32(x) - f32(y)).astype(np.float32).astype(np.float64)
    ref = (v ** 2 + eps ** 2) ** (p / 2 - 1)                        # weights.py:66-68
    assert np.allclose(out.cpu().numpy(), ref, rtol=5e-6)          # powf / rsqrt in fp32
    eng.mm_weights(dx, None, eps, p, out)
    assert np.allclose(out.cpu().numpy(), (f32(x) ** 2 + eps ** 2) ** (p / 2 - 1), rtol=5e-6)
    eng.mul(dx, dy, out)
    assert np.allclose(out.cpu().numpy(), (f32(x) * f32(y)).astype(np.float32), rtol=1e-7)


@pytest.mark.parametrize("key,eps,p", [("holder_eps0.1_p1", 0.1, 1.0), ("holder_eps0.01_p0.5", 0.01, 0.5),
                                       ("holder_eps0.1_p2", 0.1, 2.0)])
def test_mm_weights_reference_golden(eng, key, eps, p):
    """The reference's own smoothed_holder_weights values (weights.py:66-68), tests/golden/deriv_ops.npz."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "deriv_ops.npz"))
    u = g["holder_u"]
    du, out = dev(eng, u), eng.empty(u.size)
    eng.mm_weights(du, None, eps, p, out)
    # the kernel sees u rounded to fp32: the weight's sensitivity to that rounding is part of the bar
    assert np.allclose(out.cpu().numpy(), g[key], rtol=1e-5)


@pytest.mark.parametrize("n,m", [(1000, 1000), (4099, 777), (262144, 92160)])
def test_cgls_update(eng, n, m):
    rng = np.random.default_rng(n + m)
    x, p, xt = rng.standard_normal(n), rng.standard_normal(n), rng.standard_normal(n)
    r, w = rng.standard_normal(m), rng.standard_normal(m)
    dx, dp, dxt, dr, dw = (dev(eng, a) for a in (x, p, xt, r, w))
    xn = eng.empty(n)
    S = eng.scalars(5)
    S[0], S[1] = 3.0, 7.0
    eng.cgls_update(S[0:1], S[1:2], dx, dp, xn, dr, dw, dxt, S[2:5])
    step = np.float32(3.0 / 7.0)
    d = (step * f32(p).astype(np.float32)).astype(np.float32)
    xr = (f32(x).astype(np.float32) + d).astype(np.float32)
    assert np.allclose(xn.cpu().numpy(), xr, rtol=1e-6, atol=1e-7)
    assert np.allclose(dr.cpu().numpy(), f32(r) - step * f32(w), rtol=1e-6, atol=1e-6)
    got = eng.to_host(S)[2:5]
    xn64 = xn.cpu().numpy().astype(np.float64)
    assert np.isclose(got[0], np.sum(xn64 ** 2), rtol=1e-12)
    assert np.isclose(got[1], np.sum(d.astype(np.float64) ** 2), rtol=1e-6)
    assert np.isclose(got[2], np.sum((xn64 - f32(xt)) ** 2), rtol=1e-12)


@pytest.mark.parametrize("k,n", [(1, 100), (3, 4096), (8, 10_001), (13, 65_536), (37, 30_000)])
def test_gemv_t_and_gemv_n(eng, k, n):
    rng = np.random.default_rng(k * n)
    V = rng.standard_normal((k + 2, n))
    r, w2 = rng.standard_normal(n), rng.random(n)
    dV, dr, dw = dev(eng, V), dev(eng, r), dev(eng, w2)
    H = eng.scalars(2 * k)
    eng.gemv_t(dV, k, dr, H[0:k])
    eng.gemv_t(dV, k, dr, H[k:2 * k], w2=dw)
    got = eng.to_host(H)
    V32, r32, w32 = f32(V)[:k], f32(r), f32(w2)
    assert np.allclose(got[:k], V32 @ r32, rtol=1e-10, atol=1e-8)
    assert np.allclose(got[k:], V32 @ (r32 * w32).astype(np.float32).astype(np.float64), rtol=1e-10, atol=1e-8)
    # gemv_n: out = a*base + s*sum y_j V_j
    y = rng.standard_normal(k)
    Y = eng.scalars(k + 1)
    Y[:k] = torch.from_numpy(y).to(eng.device)
    base, out = dev(eng, r), eng.empty(n)
    eng.gemv_n(dV, k, Y[0:k], out, a=0.5, base=base, s=-2.0, sumsq=Y[k:k + 1])
    ref = 0.5 * r32 - 2.0 * (y @ V32)
    assert np.allclose(out.cpu().numpy(), ref, rtol=1e-6, atol=1e-6)
    assert np.isclose(eng.to_host(Y)[k], np.sum(out.cpu().numpy().astype(np.float64) ** 2), rtol=1e-12)
    eng.gemv_n(dV, k, Y[0:k], out)                       # x = V@y form
    assert np.allclose(out.cpu().numpy(), y @ V32, rtol=1e-6, atol=1e-6)
    # in place: r -= V h  (GKS.py:86-88)
    eng.gemv_n(dV, k, Y[0:k], base, a=1.0, base=base, s=-1.0)
    assert np.allclose(base.cpu().numpy(), r32 - y @ V32, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("k,m", [(1, 64), (3, 5000), (4, 4096), (7, 10_001), (13, 33_000), (14, 96), (20, 40_004), (30, 8_200),
                                 (31, 4_100), (33, 70_001), (45, 12_004), (61, 20_000), (90, 9_000)])
@pytest.mark.parametrize("weighted", [False, True])
def test_wgram(eng, k, m, weighted):
    rng = np.random.default_rng(k + m)
    W = rng.standard_normal((k + 1, m))
    w, b = rng.random(m) + 0.5, rng.standard_normal(m)
    dW, dw, db = dev(eng, W), dev(eng, w), dev(eng, b)
    G = eng.scalars(k * k + 2 * k)
    eng.wgram(dW, k, dw if weighted else None, db, G[0:k * k], G[k * k:k * k + k], G[k * k + k:])
    got = eng.to_host(G)
    W32, w32, b32 = f32(W)[:k], (f32(w) if weighted else np.ones(m)), f32(b)
    Gref = (W32 * w32 ** 2) @ W32.T
    assert np.allclose(got[:k * k].reshape(k, k), Gref, rtol=2e-6, atol=1e-6 * np.abs(Gref).max())
    assert np.allclose(got[:k * k].reshape(k, k), got[:k * k].reshape(k, k).T)
    assert np.allclose(got[k * k:k * k + k], W32 @ (w32 * b32), rtol=1e-6, atol=1e-6 * m ** 0.5)
    assert np.allclose(got[k * k + k:], W32 @ (w32 ** 2 * b32), rtol=1e-6, atol=1e-6 * m ** 0.5)


@pytest.mark.parametrize("N,nt", [(16, 1), (100, 1), (257, 1), (512, 1), (48, 5), (64, 3)])
def test_tv_grad_with_the_dot_of_its_output(eng, N, nt):
    """trk_tv_grad_dot: the same `out` as trk_tv_grad, and <out, dotv> from the same pass — widths that are not a multiple of the
    workgroup (threads beyond the image stay for the block sum), with and without weights / r_in, the space-time operator."""
    from trips_py_amd.operators import FirstDerivative2D, SpaceTimeDerivative
    L = FirstDerivative2D(N, engine=eng) if nt == 1 else SpaceTimeDerivative(N, nt, engine=eng)
    n, p = L.shape[1], L.shape[0]
    g = torch.Generator(device=eng.device).manual_seed(N + nt)
    x, rin, dv = (torch.randn(n, device=eng.device, generator=g) for _ in range(3))
    w = torch.rand(p, device=eng.device, generator=g) + 0.5
    S = eng.scalars(1)
    for ww in (None, w):
        for rr in (None, rin):
            o0, o1 = eng.empty(n), eng.empty(n)
            L.tv_grad(x, ww, rr, 0.7, o0)
            L.tv_grad(x, ww, rr, 0.7, o1, dot_with=dv, dot_out=S.ref(0))
            assert torch.equal(o0, o1)
            want = float(o0.double() @ dv.double())
            assert abs(float(eng.to_host(S)[0]) - want) <= 1e-12 * float(o0.double().norm() * dv.double().norm())
            # ... and <x, x> behind it (trk_tv_grad_dot_xsq: GKS's one-pass form takes r . r from the pass that forms L^T L r)
            S2, o2 = eng.scalars(2), eng.empty(n)
            L.tv_grad(x, ww, rr, 0.7, o2, dot_with=dv, dot_out=S2.ref(0), xsq_out=S2.ref(1))
            got = eng.to_host(S2)
            assert torch.equal(o0, o2) and got[0] == float(eng.to_host(S)[0])
            assert abs(got[1] - float(x.double() @ x.double())) <= 1e-12 * float(x.double() @ x.double())


@pytest.mark.parametrize("k,n", [(1, 1000), (7, 10_001), (8, 4096), (9, 70_000), (17, 33_333), (40, 20_000)])
def test_gemv_t_with_one_more_row(eng, k, n):
    """trk_gemv_t_x: h = V r and xrow . r from one pass (the row is not part of the basis; tile counts that change with it)."""
    rng = np.random.default_rng(7 * k + n)
    V, r, xr = rng.standard_normal((k, n)), rng.standard_normal(n), rng.standard_normal(n)
    dV, dr, dx = dev(eng, V), dev(eng, r), dev(eng, xr)
    H = eng.scalars(2 * k + 2)
    eng.gemv_t_x(dV, k, dr, dx, H[0:k], H[k:k + 1])
    eng.gemv_t(dV, k, dr, H[k + 1:2 * k + 1])
    got = eng.to_host(H)
    V32, r32, x32 = f32(V), f32(r), f32(xr)
    assert np.allclose(got[:k], V32 @ r32, rtol=1e-12, atol=1e-12 * n ** 0.5)
    assert np.isclose(got[k], x32 @ r32, rtol=1e-12, atol=1e-12 * n ** 0.5)
    assert np.allclose(got[k + 1:2 * k + 1], got[:k], rtol=1e-13, atol=1e-13 * n ** 0.5)


@pytest.mark.parametrize("N,k", [(32, 1), (32, 5), (64, 16), (64, 17), (96, 7), (96, 32), (160, 33), (128, 48), (512, 20), (1024, 3),
                                 # a multiple of four strips: the workgroup's waves in step, the right-neighbour column through LDS
                                 (128, 16), (256, 13), (384, 30), (256, 25),
                                 # 256-column tiles staged through LDS (k_wgram_tv_lds: N a multiple of 256 from 2048 on, k <= 32):
                                 # one and two tiles of vectors, tile counts that do and do not deal evenly to the XCDs; 33: the register-fed kernel
                                 (2048, 5), (2048, 16), (2048, 17), (2048, 24), (2048, 32), (2304, 20), (2048, 33)])
def test_wgram_tv_from_v_equals_the_gram_of_the_stored_images(eng, N, k):
    """trk_wgram_tv (the weighted Gram of L V formed from V, L the 2-D first difference) against (i) the float64 definition on the
    oracle's L and (ii) trk_wgram over the stored images L v_j: all tile counts, bands that do not divide N, the image's right
    and bottom edges, weights as trk_tv_weights lays them out."""
    from trips_py_amd.operators import FirstDerivative2D
    L = FirstDerivative2D(N, engine=eng)
    n, p = N * N, 2 * N * (N - 1)
    g = torch.Generator(device=eng.device).manual_seed(100 * N + k)
    V = torch.randn(k + 1, n + 4, device=eng.device, generator=g)[:k, :n]      # rows 16-byte aligned, not contiguous
    w = torch.rand(p, device=eng.device, generator=g) + 0.25
    LV = torch.empty(k, p, device=eng.device)
    for j in range(k):
        L.apply(V[j].contiguous(), out=LV[j])
    G = eng.scalars(3 * k * k + k)
    eng.wgram_tv(V, k, N, w, G[0:k * k])
    eng.wgram(LV, k, w, None, G[k * k:2 * k * k])
    # ... and with one more image z: the same Gram, and V z from the same pass (trk_wgram_tv_z)
    z = torch.randn(n, device=eng.device, generator=g)
    eng.wgram_tv(V, k, N, w, G[2 * k * k:3 * k * k], z=z, h=G[3 * k * k:3 * k * k + k])
    got = eng.to_host(G)
    a, b = got[:k * k].reshape(k, k), got[k * k:2 * k * k].reshape(k, k)
    assert np.array_equal(got[2 * k * k:3 * k * k].reshape(k, k), a)
    hz = (V.double() @ z.double()).cpu().numpy()
    assert np.allclose(got[3 * k * k:], hz, rtol=1e-12, atol=1e-12 * float(z.double().norm() * V.double().norm(dim=1).max()))
    ref = ((LV.double() * w.double() ** 2) @ LV.double().T).cpu().numpy()
    scale = np.abs(ref).max()
    assert np.allclose(a, a.T) and np.allclose(a, ref, rtol=2e-6, atol=1e-6 * scale) and np.allclose(a, b, rtol=2e-6, atol=1e-6 * scale)
    # and the layout of the differences themselves: the oracle's sparse L
    if N <= 96:
        from oracle.cpu_ref import first_derivative_2d
        Ls = first_derivative_2d(N, N)
        LVo = (Ls @ V.double().cpu().numpy().T).T
        refo = (LVo * w.double().cpu().numpy() ** 2) @ LVo.T
        assert np.allclose(a, refo, rtol=2e-6, atol=1e-6 * scale)


def _adversarial_basis(kind, k, N, dev):
    """Rows that make the roundings of a split-operand product CORRELATED: the same few values over and
    over, so that whatever a bf16 split drops is the same number in millions of terms."""
    g = torch.Generator(device=dev).manual_seed(7 * k + N)
    if kind == "piecewise8":
        # piecewise-constant images of <= 8 distinct values on a coarse block grid (what a TV-regularised iterate looks like)
        levels = torch.tensor([0.0, 0.2137, 0.3931, 0.5009, 0.6877, 0.8113, 0.9371, 1.0], device=dev)
        V = torch.empty(k, N * N, device=dev)
        for j in range(k):
            bs = 32 * (1 + j % 4)
            idx = torch.randint(0, 8, (N // bs + 1, N // bs + 1), device=dev, generator=g)
            img = levels[idx].repeat_interleave(bs, 0).repeat_interleave(bs, 1)[:N, :N]
            V[j] = img.reshape(-1) * (1.0 + 0.01 * j)
        return V
    if kind == "constant_steps":
        # every difference is one of two numbers: a ramp in x times a ramp in y with steps just above a bf16 rounding boundary
        step = 1.0 + 2.0 ** -8 + 2.0 ** -16 + 2.0 ** -17 + 2.0 ** -22      # bf16 keeps 8 bits, two pieces 16: the rest is dropped
        ii = torch.arange(N, device=dev, dtype=torch.float64)
        V = torch.empty(k, N * N, device=dev)
        for j in range(k):
            a, b = step * (1 + j % 3), step * (1 + (j // 3) % 4)
            V[j] = ((a * ii)[:, None] * 2.0 ** -12 + (b * ii)[None, :] * 2.0 ** -12).float().reshape(-1)
        return V
    if kind == "constant":
        V = torch.ones(k, N * N, device=dev) * torch.linspace(0.3, 1.7, k, device=dev)[:, None]
        V[:, ::N] += 0.5004883                                                # one step per row, the same everywhere
        return V
    raise ValueError(kind)


@pytest.mark.parametrize("kind", ["piecewise8", "constant_steps", "constant"])
@pytest.mark.parametrize("with_z", [False, True])
@pytest.mark.parametrize("mode,limit", [("auto", 1e-6), ("bf16x2", 1.2e-5), ("bf16x3", 1e-6), ("fp32", 1e-6)])
def test_wgram_tv_split_products_on_adversarial_images(eng, kind, with_z, mode, limit):
    """trk_wgram_tv(_z) — Gram tiles through the bf16 matrix pipe with split operands — on images whose differences repeat a few
    values millions of times, unit weights and the weights trk_tv_weights makes of such an image: every entry within 1e-6 of the
    float64 Gram relative to sqrt(G_aa G_bb), in the four arithmetic modes of trk_wgram_tv_precision — the accuracy contract of
    include/trk.h: the two-piece split loses up to 2^-16 of each operand, all of one sign on such images (measured 5.8e-6 on
    `constant_steps`; bar 2 x that), three pieces or the fp32 pipe stay within 1e-6 (measured 4.7e-7); the default, 'auto', measures
    the loss on a sample per call and runs the fp32 pipe where it exceeds 1e-6 (it does on every image of this test that needs it)."""
    from trips_py_amd.operators import FirstDerivative2D
    N, k = 2048, 24
    dev = eng.device
    L = FirstDerivative2D(N, engine=eng)
    n, p = N * N, 2 * N * (N - 1)
    V = _adversarial_basis(kind, k, N, dev)
    ws = [torch.ones(p, device=dev)]
    wt = torch.empty(p, device=dev)
    L.tv_weights(V[0].contiguous(), 0.1, 1.0, wt)                             # ((L x)^2 + eps^2)^(-1/2) of a piecewise image
    ws.append(wt)
    z = torch.randn(n, device=dev, generator=torch.Generator(device=dev).manual_seed(5))
    G = eng.scalars(k * k + k)
    worst = 0.0
    was = eng.wgram_tv_precision(mode)
    for w in ws:
        if with_z:
            eng.wgram_tv(V, k, N, w, G[0:k * k], z=z, h=G[k * k:k * k + k])
        else:
            eng.wgram_tv(V, k, N, w, G[0:k * k])
        got = eng.to_host(G)[:k * k].reshape(k, k)
        # float64 reference from the fp32 operands as the kernel forms them: d = fl32(fl32(x_i - x_j) * w)
        ref = torch.zeros(k, k, dtype=torch.float64, device=dev)
        for lo in range(0, N, 256):                                            # row blocks: 24 x 2048 x 256 doubles at a time
            hi = min(lo + 256, N)
            X = V.reshape(k, N, N)[:, lo:min(hi + 1, N), :]
            dh = ((X[:, :hi - lo, :-1] - X[:, :hi - lo, 1:]) * w[:N * (N - 1)].reshape(N, N - 1)[lo:hi]).double().reshape(k, -1)
            ref += dh @ dh.T
            nv = min(hi, N - 1) - lo
            if nv > 0:
                dv = ((X[:, :nv, :] - X[:, 1:nv + 1, :]) * w[N * (N - 1):].reshape(N - 1, N)[lo:lo + nv]).double().reshape(k, -1)
                ref += dv @ dv.T
        ref = ref.cpu().numpy()
        dg = np.sqrt(np.abs(np.diag(ref)))
        scale = np.maximum(np.outer(dg, dg), 1e-300)
        worst = max(worst, float(np.max(np.abs(got - ref) / scale)))
    verdict, sampled = eng.wgram_tv_last_probe() if mode == "auto" else (None, None)
    assert eng.wgram_tv_precision(was) == mode
    bar(f"wgram_tv.adversarial[{kind}{'-z' if with_z else ''}-{mode}]", worst, limit)
    if mode == "auto":
        # (the probe's estimate of the last call — the weights of trk_tv_weights — next to what the two-piece split really loses)
        assert verdict in (0, 1) and sampled >= 0.0
        if verdict == 0:
            assert sampled <= 3e-7


def test_wgram_tv_auto_mode_keeps_the_two_piece_form_on_noisy_data(eng):
    """'auto' on the data the solvers meet (Krylov vectors: noise-like): the probe's sampled deviation is ~1e-9, the verdict 0, and
    the Gram is the two-piece Gram to the bit — the probe costs three near-empty launches, not accuracy or the matrix pipe's speed."""
    from trips_py_amd.operators import FirstDerivative2D
    N, k = 1024, 20
    n, p = N * N, 2 * N * (N - 1)
    g = torch.Generator(device=eng.device).manual_seed(3)
    V = torch.randn(k, n, device=eng.device, generator=g)
    w = torch.rand(p, device=eng.device, generator=g) + 0.25
    z = torch.randn(n, device=eng.device, generator=g)
    G = eng.scalars(2 * (k * k + k))
    was = eng.wgram_tv_precision("auto")
    try:
        eng.wgram_tv(V, k, N, w, G[0:k * k], z=z, h=G[k * k:k * k + k])
        verdict, sampled = eng.wgram_tv_last_probe()
        eng.wgram_tv_precision("bf16x2")
        eng.wgram_tv(V, k, N, w, G[k * k + k:2 * k * k + k], z=z, h=G[2 * k * k + k:2 * k * k + 2 * k])
    finally:
        eng.wgram_tv_precision(was)
    got = eng.to_host(G)
    assert verdict == 0 and sampled < 1e-7, (verdict, sampled)
    assert np.array_equal(got[:k * k + k], got[k * k + k:])
    # ... and an image of repeated values flips it (the verdict is per call, nothing sticks)
    N2, k2 = 2048, 24
    V2 = _adversarial_basis("constant_steps", k2, N2, eng.device)
    G2 = eng.scalars(k2 * k2)
    w2 = torch.empty(2 * N2 * (N2 - 1), device=eng.device)
    FirstDerivative2D(N2, engine=eng).tv_weights(V2[0].contiguous(), 0.1, 1.0, w2)       # (the weighting under which two pieces lose 5.8e-6)
    eng.wgram_tv_precision("auto")
    try:
        eng.wgram_tv(V2, k2, N2, w2, G2[0:k2 * k2])
        verdict2, sampled2 = eng.wgram_tv_last_probe()
        eng.wgram_tv(V, k, N, w, G[0:k * k])
        verdict3, _ = eng.wgram_tv_last_probe()
    finally:
        eng.wgram_tv_precision(was)
    assert verdict2 == 1 and sampled2 > 1e-6 and verdict3 == 0, (verdict2, sampled2, verdict3)


@pytest.mark.parametrize("where", ["last4", "last4_but_newest", "last1"])
def test_wgram_tv_auto_probe_sees_a_basis_that_turns_piecewise_constant_at_its_end(eng, where):
    """a basis whose NEWEST vectors alone are images of repeated values (MMGKS late in a TV solve), the rest
    noise-like — the probe samples four vectors spread over the basis and, since round 6, the newest four with all their pairs: the
    verdict is 1 and the Gram within the contract's 1e-6 whether the repeated-value images are the last four, the three before the
    newest, or the newest alone."""
    from trips_py_amd.operators import FirstDerivative2D
    N, k = 2048, 24
    dev = eng.device
    n, p = N * N, 2 * N * (N - 1)
    V = torch.randn(k, n, device=dev, generator=torch.Generator(device=dev).manual_seed(11))
    adv = _adversarial_basis("constant_steps", 4, N, dev)
    rows = {"last4": [k - 4, k - 3, k - 2, k - 1], "last4_but_newest": [k - 4, k - 3, k - 2], "last1": [k - 1]}[where]
    for j, r in enumerate(rows):
        V[r] = adv[j]
    w = torch.empty(p, device=dev)
    FirstDerivative2D(N, engine=eng).tv_weights(adv[0].contiguous(), 0.1, 1.0, w)
    G = eng.scalars(2 * k * k)
    was = eng.wgram_tv_precision("auto")
    try:
        eng.wgram_tv(V, k, N, w, G[0:k * k])
        verdict, sampled = eng.wgram_tv_last_probe()
        eng.wgram_tv_precision("fp32")
        eng.wgram_tv(V, k, N, w, G[k * k:2 * k * k])
    finally:
        eng.wgram_tv_precision(was)
    got = eng.to_host(G)
    a, ref = got[:k * k].reshape(k, k), got[k * k:].reshape(k, k)
    dg = np.sqrt(np.abs(np.diag(ref)))
    worst = float(np.max(np.abs(a - ref) / np.maximum(np.outer(dg, dg), 1e-300)))
    bar(f"wgram_tv.auto_late_basis[{where}]", worst, 1e-6)
    assert verdict == 1 and sampled > 3e-7, (verdict, sampled)


@pytest.mark.gpu
@pytest.mark.parametrize("k,mu", [(1, 0.3), (2, 0.0), (7, 1e-2), (100, 1e-3), (1000, 0.5)])
def test_bidiag_tikhonov_matches_stacked_lstsq(k, mu):
    """trk_bidiag_tikhonov against the reference's formulation (Hybrid_LSQR.py:104): lstsq on [B; mu I], [beta0 e1; 0]."""
    from trips_py_amd.engine import default_engine
    eng = default_engine()
    rng = np.random.default_rng(k)
    al, be, b0 = rng.random(k) + 0.05, rng.random(k) + 0.05, 1.7
    AB = eng.scalars(2 * k + 1)
    ab = np.zeros(2 * k + 1)
    ab[0], ab[1::2], ab[2::2] = b0 ** 2, al ** 2, be ** 2
    AB.set(0, ab)
    Y = eng.scalars(k)
    eng.bidiag_tikhonov(AB.ref(1), 2, AB.ref(2), 2, k, mu, AB.ref(0), Y.ref(0))
    B = np.zeros((k + 1, k))
    B[np.arange(k), np.arange(k)] = al
    B[np.arange(1, k + 1), np.arange(k)] = be
    rhs = np.zeros(2 * k + 1)
    rhs[0] = b0
    want = np.linalg.lstsq(np.vstack((B, mu * np.eye(k))), rhs, rcond=None)[0]
    got = Y.host()
    assert np.abs(got - want).max() <= 1e-10 * max(1.0, np.abs(want).max()) * np.linalg.cond(np.vstack((B, mu * np.eye(k))))
    # coefficients of un-normalised basis vectors alpha_j v_j
    eng.bidiag_tikhonov(AB.ref(1), 2, AB.ref(2), 2, k, mu, AB.ref(0), Y.ref(0), y_over_alpha=True)
    assert np.allclose(Y.host() * al, got, rtol=1e-13, atol=0)
    # resumable form: growing k with the same mu (one column or several at a time), then a change of mu, then a smaller k
    W = eng.scalars(3 * (k + 1) + 4)
    for kk, m in [(max(1, k // 3), mu), (max(1, k // 3) + 1 if k > 3 else k, mu), (k, mu), (k, mu + 0.25), (max(1, k - 1), mu + 0.25)]:
        kk = min(kk, k)
        eng.bidiag_tikhonov(AB.ref(1), 2, AB.ref(2), 2, kk, m, AB.ref(0), Y.ref(0), W)
        Bk = B[:kk + 1, :kk]
        rk = np.zeros(2 * kk + 1)
        rk[0] = b0
        wk = np.linalg.lstsq(np.vstack((Bk, m * np.eye(kk))), rk, rcond=None)[0]
        assert np.abs(Y.host(0, kk) - wk).max() <= 1e-10 * max(1.0, np.abs(wk).max()) * np.linalg.cond(np.vstack((Bk, m * np.eye(kk)))), (kk, m)


@pytest.mark.gpu
@pytest.mark.parametrize("k,n", [(1, 100), (3, 4097), (8, 10_000), (9, 65_536), (12, 50_000), (16, 30_001)])
def test_gemv_nt_fused_gram_schmidt_step(k, n):
    """trk_gemv_nt: w_out = w - V^T-combination with the old coefficients AND the new dot products in one pass."""
    from trips_py_amd.engine import default_engine
    eng = default_engine()
    rng = np.random.default_rng(k * 7 + n)
    V, w, h = rng.standard_normal((k, n)), rng.standard_normal(n), rng.standard_normal(k) * 0.3
    dV = torch.from_numpy(V.astype(np.float32)).to(eng.device)
    dw = eng.to_vec(w)
    H = eng.scalars(2 * k)
    H.set(0, h)
    out = eng.empty(n)
    eng.gemv_nt(dV, k, H.ref(0), dw, out, H.ref(k))
    V32, w32 = V.astype(np.float32).astype(np.float64), w.astype(np.float32).astype(np.float64)
    want = w32 - h @ V32
    got = out.cpu().numpy().astype(np.float64)
    assert np.allclose(got, want, rtol=1e-6, atol=1e-6)
    assert np.allclose(H.host(k, 2 * k), V32 @ got, rtol=1e-10, atol=1e-9)
    eng.gemv_nt(dV, k, H.ref(0), dw, dw, H.ref(k))                 # in place
    assert torch.equal(dw, out)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(64, 64), (96, 40)])
def test_golub_kahan_unnormalised_storage_is_the_same_factorisation(shape):
    """GKState(normalized=False) keeps beta_j u_j / alpha_j v_j: same alpha, beta, and the same vectors after division."""
    import torch
    from trips_py_amd.krylov import GKState
    from trips_py_amd.operators import Blur2D
    from oracle import cpu_ref as O
    rng = np.random.default_rng(5)
    A = Blur2D(O.gauss_psf((7, 7), 1.2)[0], shape[0], shape[1])
    b = rng.standard_normal(A.shape[0]).astype(np.float32)
    g1, g2 = GKState(A, b, 12), GKState(A, b, 12, normalized=False)
    for _ in range(12):
        g1.step()
        g2.step(sync=False)
    assert np.allclose(g2.alphas, g1.alphas, rtol=2e-5) and np.allclose(g2.betas, g1.betas, rtol=2e-5)
    assert abs(g2.beta0 - g1.beta0) <= 1e-6 * g1.beta0
    V1, V2 = g1.V.numpy(), g2.V.numpy() / np.asarray(g2.alphas)
    U1, U2 = g1.U.numpy(), g2.U.numpy() / np.concatenate(([g2.beta0], g2.betas))
    assert np.abs(V1 - V2).max() <= 2e-4 * np.abs(V1).max() and np.abs(U1 - U2).max() <= 2e-4 * np.abs(U1).max()



@pytest.mark.gpu
@pytest.mark.parametrize("k,n", [(1, 1000), (5, 4099), (40, 65_536)])
def test_gemv_n_err_partials(k, n):
    """trk_gemv_n_err: x = V y and the raw partials of ||x - ref||^2, summed later by trk_finalize_batched."""
    import torch
    from trips_py_amd.engine import default_engine
    eng = default_engine()
    rng = np.random.default_rng(k)
    V = eng.empty_basis(k + 1, n)
    Vh = rng.standard_normal((k, n)).astype(np.float32)
    V[:k].copy_(torch.from_numpy(Vh))
    yh = rng.standard_normal(k)
    Y = eng.scalars(k)
    Y.set(0, yh)
    ref = torch.from_numpy(rng.standard_normal(n).astype(np.float32)).to(eng.device)
    out, out2 = eng.empty(n), eng.empty(n)
    EP, E = eng.scalars(2 * 1024), eng.scalars(2)
    nb = eng.gemv_n_err(V, k, Y.ref(0), out, ref, EP.ref(0), 1024)
    nb2 = eng.gemv_n_err(V, k, Y.ref(0), out, ref, EP.ref(nb), 1024)
    assert nb == nb2 and 1 <= nb <= 1024
    eng.finalize_batched(EP.ref(0), nb, 1, 2, E.ref(0), 1)
    eng.gemv_n(V, k, Y.ref(0), out2)
    assert torch.equal(out, out2)
    want = float(np.sum((out.cpu().numpy().astype(np.float64) - ref.cpu().numpy().astype(np.float64)) ** 2))
    got = E.host()
    assert abs(got[0] - want) <= 1e-12 * want and got[0] == got[1]


@pytest.mark.gpu
@pytest.mark.parametrize("k,n,with_x", [(1, 1000, True), (5, 4099, True), (13, 70_001, False), (40, 262_144, True)])
def test_gemv_orth_iterate_one_pass_for_the_new_vector_and_the_next_iterate(k, n, with_x):
    """trk_cgs_coeffs_rho + trk_gemv_orth_iterate: rho^2 = ||w - V c||^2 by algebra from the sweep's products, vn = (w - V c)/rho and
    x' = V y'[:k] + y'[k] vn in ONE pass — against float64 NumPy, against the pass's own computed norm (chk), and x' bit for bit
    against trk_gemv_n over the k + 1 stored vectors (the sum it replaces, term for term)."""
    import torch
    from trips_py_amd.engine import default_engine
    eng = default_engine()
    rng = np.random.default_rng(100 + k)
    V = eng.empty_basis(k + 1, n)
    Q, _ = np.linalg.qr(rng.standard_normal((n, k)))
    Vh = np.ascontiguousarray(Q.T).astype(np.float32)                 # orthonormal to fp32 rounding
    V[:k].copy_(torch.from_numpy(Vh))
    w64 = rng.standard_normal(n)
    w64 -= Q @ (Q.T @ w64) * (1.0 - 1e-4)                             # nearly orthogonal to V, as the solvers' residuals are
    w = torch.from_numpy(w64.astype(np.float32)).to(eng.device)
    wd, Vd = w.cpu().numpy().astype(np.float64), Vh.astype(np.float64)
    W, G, C, RR = eng.scalars(4 * (k + 1)), eng.scalars((k + 1) * (k + 1)), eng.scalars(k + 1), eng.scalars(3)
    for j in range(k):                                                # G = V^T V, row by row as GramSchmidtByGram installs it
        eng.gemv_t(V, j + 1, V[j], W.ref(0))
        eng.cgs_coeffs(G.ref(0), k + 1, None, W.ref(0), j + 1, 0, None)
    eng.gemv_t(V, k, w, W.ref(0))
    eng.nrm2sq(w, RR.ref(0))
    eng.cgs_coeffs_rho(G.ref(0), k + 1, W.ref(0), None, k, 3, C.ref(0), RR.ref(0), RR.ref(1))
    c = C.host()[:k]
    o64 = wd - Vd.T @ c
    rho2 = RR.host()[1]
    assert abs(rho2 - float(o64 @ o64)) <= 1e-9 * float(o64 @ o64)
    yh = rng.standard_normal(k + 1)
    Y = eng.scalars(k + 1)
    Y.set(0, yh)
    ref = torch.from_numpy(rng.standard_normal(n).astype(np.float32)).to(eng.device)
    x, x2 = eng.empty(n), eng.empty(n)
    EP, E = eng.scalars(2048), eng.scalars(1)
    if with_x:
        nb = eng.gemv_orth_iterate(V, k, w, C.ref(0), RR.ref(1), V[k], y_next=Y.ref(0), x_next=x, ref=ref, partials=EP.ref(0),
                                   capacity=2048, chk=RR.ref(2))
        assert 1 <= nb <= 2048
    else:
        assert eng.gemv_orth_iterate(V, k, w, C.ref(0), RR.ref(1), V[k], chk=RR.ref(2)) == 0
    vn = V[k].cpu().numpy().astype(np.float64)
    want = o64 / np.sqrt(rho2)
    assert np.abs(vn - want).max() <= 1.2e-7 * np.abs(want).max() + 1e-12
    assert abs(float(vn @ vn) - 1.0) < 1e-6
    chk = RR.host()[2]
    assert abs(chk - rho2) <= 1e-9 * rho2                             # the computed norm and the algebraic one
    if with_x:
        eng.gemv_n(V, k + 1, Y.ref(0), x2)
        assert torch.equal(x, x2)
        eng.finalize_batched(EP.ref(0), nb, 1, 1, E.ref(0), 1)
        e = float(np.sum((x.cpu().numpy().astype(np.float64) - ref.cpu().numpy().astype(np.float64)) ** 2))
        assert abs(E.host()[0] - e) <= 1e-12 * e


@pytest.mark.gpu
@pytest.mark.parametrize("k,n", [(12, 4096), (13, 64), (40, 122_880), (53, 122_880), (100, 262_144), (11, 4096), (40, 122_882)])
def test_gemv_n_on_short_vectors_rows_split_over_the_waves(k, n):
    """k_gemv_n_split (a projector's m-length images: few columns, many rows — the four waves of a workgroup take a quarter of the rows
    each) through trk_gemv_n with and without a base and through trk_gemv_orth_iterate's image form (A v_k = (A r - AV c) / rho), against
    float64 NumPy to fp32 rounding; the last two shapes fall back to k_gemv_n (k < 12, n % 4 != 0)."""
    import torch
    from trips_py_amd.engine import default_engine
    eng = default_engine()
    rng = np.random.default_rng(k * 7 + n)
    V = eng.empty_basis(k + 1, n)
    Vh = rng.standard_normal((k, n)).astype(np.float32)
    V[:k].copy_(torch.from_numpy(Vh))
    yh = rng.standard_normal(k)
    Y, R2 = eng.scalars(k), eng.scalars(1)
    Y.set(0, yh)
    R2.set(0, np.array([7.25]))
    bh = rng.standard_normal(n).astype(np.float32)
    b = torch.from_numpy(bh).to(eng.device)
    out = eng.empty(n)
    Vd = Vh.astype(np.float64)
    comb = Vd.T @ yh
    eng.gemv_n(V, k, Y.ref(0), out)
    assert np.abs(out.cpu().numpy() - comb).max() <= 1.2e-7 * np.abs(comb).max()
    eng.gemv_n(V, k, Y.ref(0), out, a=-1.0, base=b, s=1.0)
    want = comb - bh.astype(np.float64)
    assert np.abs(out.cpu().numpy() - want).max() <= 1.2e-7 * np.abs(want).max()
    assert eng.gemv_orth_iterate(V, k, b, Y.ref(0), R2.ref(0), V[k]) == 0
    want = (bh.astype(np.float64) - comb) / np.sqrt(7.25)
    assert np.abs(V[k].cpu().numpy() - want).max() <= 1.2e-7 * np.abs(want).max()
    assert torch.equal(V[:k].cpu(), torch.from_numpy(Vh))                   # the rows read are untouched


@pytest.mark.gpu
@pytest.mark.parametrize("groups,glen,copies,expo", [(1, 1, 1, -0.5), (37, 3, 3, -0.5), (1000, 7, 2, -0.75), (5000, 32, 1, 0.0)])
def test_group_weights(groups, glen, copies, expo):
    """trk_group_weights: (sum of squares over each group of consecutive entries + add)^expo, tiled `copies` times."""
    import torch
    from trips_py_amd.engine import default_engine
    eng = default_engine()
    rng = np.random.default_rng(groups + glen)
    d = rng.standard_normal(groups * glen).astype(np.float32)
    dd = torch.from_numpy(d).to(eng.device)
    out = eng.empty(groups * copies)
    add = float(np.exp(2))
    eng.group_weights(dd, groups, glen, add, expo, copies, out)
    want = np.tile(((d.astype(np.float64).reshape(groups, glen) ** 2).sum(axis=1) + add) ** expo, copies)
    assert np.allclose(out.cpu().numpy(), want, rtol=2e-7, atol=0)


@pytest.mark.gpu
@pytest.mark.parametrize("n,with_xt", [(8, False), (4099 * 4, True), (100_000, True)])
def test_cgls_regrouped_update_kernels(n, with_xt):
    """trk_cgls_r_update and trk_cgls_xp_update against the formulas of CGLS.py:64-72 (scalars as block partials)."""
    import torch
    from trips_py_amd.engine import default_engine
    eng = default_engine()
    rng = np.random.default_rng(n)
    f = lambda: torch.from_numpy(rng.standard_normal(n).astype(np.float32)).to(eng.device)
    x, p, t, r, w, xt = f(), f(), f(), f(), f(), f()
    x_new = eng.empty(n)
    S = eng.scalars(8)                      # [gamma_old, delta_pub, gamma_pub]
    parts = rng.random(5) + 0.1             # block partials of delta / gamma_new
    PD, PG = eng.scalars(5), eng.scalars(5)
    PD.set(0, parts)
    PG.set(0, parts[::-1] * 2.0)
    S.set(0, [1.7])
    delta, gnew, gold = parts.sum(), (parts[::-1] * 2.0).sum(), 1.7
    r0, p0 = r.cpu().numpy().copy(), p.cpu().numpy().copy()
    eng.cgls_r_update(S.ref(0), PD.ref(0), 5, r, w, S.ref(1))
    step = np.float32(gold / delta)
    assert np.allclose(r.cpu().numpy(), r0 - step * w.cpu().numpy(), rtol=1e-6, atol=1e-6)
    NP = eng.scalars(3 * 1024)
    nb = eng.cgls_xp_update(S.ref(0), S.ref(1), PG.ref(0), 5, x, p, t, x_new, xt if with_xt else None, S.ref(2), NP.ref(0), 1024)
    sh = S.host(0, 3)
    assert abs(sh[1] - delta) <= 1e-14 * delta and abs(sh[2] - gnew) <= 1e-14 * gnew
    xn = x.cpu().numpy() + step * p0
    assert np.allclose(x_new.cpu().numpy(), xn, rtol=1e-6, atol=1e-6)
    beta = np.float32(gnew / gold)
    assert np.allclose(p.cpu().numpy(), t.cpu().numpy() + beta * p0, rtol=1e-6, atol=1e-6)
    sums = NP.host(0, 3 * nb).reshape(nb, 3).sum(axis=0)
    xn64 = x_new.cpu().numpy().astype(np.float64)
    assert abs(sums[0] - (xn64 ** 2).sum()) <= 1e-10 * (xn64 ** 2).sum()
    if with_xt:
        e = ((xn64 - xt.cpu().numpy()) ** 2).sum()
        assert abs(sums[2] - e) <= 1e-10 * e


@pytest.mark.parametrize("N", [2, 3, 17, 64, 257, 520, 1030])
@pytest.mark.parametrize("q", [1.0, 0.7, 2.0])
def test_fused_tv_weights_and_gradient(eng, N, q):
    """trk_tv_weights / trk_tv_grad (no L x written out) against the separate kernels they replace: L @ x -> weights
    (MMGKS.py:60,93) bit for bit; w * (L x) -> L^T -> r + lam * rb (MMGKS.py:116-118) to fp32 rounding of the last sum;
    and against the float64 sparse matrix of the oracle."""
    from oracle import cpu_ref as O
    from trips_py_amd.operators import FirstDerivative2D
    L = FirstDerivative2D(N, engine=eng)
    g = torch.Generator(device=eng.device).manual_seed(N)
    x = torch.randn(N * N, device=eng.device, generator=g)
    r = torch.randn(N * N, device=eng.device, generator=g)
    lam, eps = 0.37, 0.1
    lx = L.apply(x)
    w_ref = eng.empty(L.shape[0])
    eng.mm_weights(lx, None, eps, q, w_ref)
    w = eng.empty(L.shape[0])
    L.tv_weights(x, eps, q, w)
    assert torch.equal(w, w_ref)
    tp, rb = eng.empty(L.shape[0]), eng.empty(N * N)
    eng.mul(w_ref, lx, tp)
    L.apply(tp, out=rb, transpose=True)
    want = eng.empty(N * N)
    eng.axpby(1.0, r, lam, rb, want)
    got = eng.empty(N * N)
    L.tv_grad(x, w, r, lam, out=got)
    assert relerr(got.cpu().numpy(), want.cpu().numpy()) < 3e-7
    with pytest.raises(ValueError):
        L.tv_grad(x, w, r, lam, out=r)                       # in place is refused
    # unit weights, no r_in: lam * L^T L x
    L.tv_grad(x, None, None, lam, out=got)
    Lm = O.first_derivative_2d(N, N).astype(np.float64)
    x64 = x.cpu().numpy().astype(np.float64)
    assert relerr(got.cpu().numpy(), lam * (Lm.T @ (Lm @ x64))) < 2e-6
    w64 = ((Lm @ x64) ** 2 + eps ** 2) ** (q / 2 - 1)
    assert relerr(w.cpu().numpy(), w64) < 2e-6


@pytest.mark.parametrize("kmax,lam", [(5, 0.3), (40, 1e-2), (120, 1e-3)])
def test_hess_tikhonov_modes_against_stacked_lstsq(eng, kmax, lam):
    """trk_hess_tikhonov: column by column of a random Hessenberg matrix, y = argmin ||H_k y - beta0 e1||^2 + lam ||y||^2
    (Hybrid_GMRES.py:69-77) by the bordering update of the inverse (mode 1, chain started by mode 2 with lam = 0 at k = 1 as
    the reference does) and by Cholesky from scratch (mode 0) against numpy's stacked least-squares solve."""
    rng = np.random.default_rng(kmax)
    Hm = np.triu(rng.standard_normal((kmax + 1, kmax)), -1)
    Hm[np.arange(1, kmax + 1), np.arange(kmax)] = np.abs(Hm[np.arange(1, kmax + 1), np.arange(kmax)]) + 0.5
    beta0 = 2.7
    Hd, Gd, Mi = eng.scalars((kmax + 1) * kmax), eng.scalars(kmax * kmax), eng.scalars(kmax * kmax)
    H0, G0 = eng.scalars((kmax + 1) * kmax), eng.scalars(kmax * kmax)
    S = eng.scalars(2 * kmax + 2)
    Y, Y0 = eng.scalars(kmax), eng.scalars(kmax)
    for k in range(1, kmax + 1):
        col = Hm[:k + 1, k - 1]
        half = rng.standard_normal(k)                                  # the two sweeps' coefficient sets add up to the column
        S.set(0, np.concatenate(([col[k] ** 2], half, col[:k] - half)))
        lk = 0.0 if k == 1 else lam
        eng.hess_tikhonov(Hd.ref(0), kmax + 1, Gd.ref(0), Mi.ref(0), kmax, S.ref(1), S.ref(1 + k), S.ref(0), beta0, k, lk,
                          2 if k <= 2 else 1, Y.ref(0))
        eng.hess_tikhonov(H0.ref(0), kmax + 1, G0.ref(0), None, kmax, S.ref(1), S.ref(1 + k), S.ref(0), beta0, k, lk, 0, Y0.ref(0))
        if k in (1, 2, 3, kmax // 2, kmax):
            Hk = Hm[:k + 1, :k]
            rhs = np.zeros(2 * k + 1)
            rhs[0] = beta0
            want = np.linalg.lstsq(np.vstack((Hk, np.sqrt(lk) * np.eye(k))), rhs, rcond=None)[0]
            # normal equations in float64 on a random (worse than Arnoldi's) H: cond(H^T H + lam I) <= ||H||^2 / lam ~ 1e5 at
            # k = 120, and the bordered inverse carries k updates — both far below the fp32 vectors the result multiplies
            assert relerr(Y.host(0, k), want) < 1e-7, (k, "bordering")
            assert relerr(Y0.host(0, k), want) < 1e-8, (k, "cholesky")
    assert np.allclose(Hd.host(0, (kmax + 1) * kmax).reshape(kmax, kmax + 1).T, Hm, rtol=1e-12, atol=1e-13)


@pytest.mark.parametrize("kmax,lam", [(6, 0.5), (60, 1e-2), (150, 1e-2)])
def test_gram_tikhonov_bordered_inverse_against_solve(eng, kmax, lam):
    """trk_gram_tikhonov with Minv: G_A, G_L grow by one row and column per call (GKS), the inverse of G_A + lam G_L is
    bordered — started from a 3 x 3 block as GKS starts from projection_dim = 3, k up to 150 (> the Cholesky form's LDS
    limit) — against numpy.linalg.solve and, where it applies, against the Cholesky form."""
    rng = np.random.default_rng(kmax)
    Wa, Wl = rng.standard_normal((kmax, 2 * kmax + 5)), rng.standard_normal((kmax, 3 * kmax))
    GA, GL = Wa @ Wa.T, Wl @ Wl.T
    c = rng.standard_normal(kmax)
    GA_d, GL_d, c_d = eng.scalars(kmax * kmax), eng.scalars(kmax * kmax), eng.scalars(kmax)
    GA_d.set(0, GA.reshape(-1))
    GL_d.set(0, GL.reshape(-1))
    c_d.set(0, c)
    Minv, Y, Y0 = eng.scalars(kmax * kmax), eng.scalars(kmax), eng.scalars(kmax)
    k_inv = 0
    for k in range(3, kmax + 1):
        eng.gram_tikhonov(GA_d.ref(0), kmax, GL_d.ref(0), kmax, c_d.ref(0), k, lam, Y.ref(0), Minv=Minv.ref(0), ldm=kmax, k_from=k_inv)
        k_inv = k
        if k in (3, 4, kmax // 2, kmax):
            want = np.linalg.solve(GA[:k, :k] + lam * GL[:k, :k], c[:k])
            assert relerr(Y.host(0, k), want) < 1e-9, k
            if k <= eng.GRAM_TIKHONOV_MAX_K:
                eng.gram_tikhonov(GA_d.ref(0), kmax, GL_d.ref(0), kmax, c_d.ref(0), k, lam, Y0.ref(0))
                assert relerr(Y0.host(0, k), want) < 1e-9, k



@pytest.mark.parametrize("N,nt", [(16, 3), (33, 2), (64, 5), (256, 4)])
@pytest.mark.parametrize("q", [1.0, 0.7])
def test_fused_tv_forms_of_the_spacetime_operator(eng, N, nt, q):
    """trk_tv_weights / trk_tv_grad on the space-time first-difference operator (one rank owns the time axis): weights bit for
    bit with L @ x -> trk_mm_weights; the weighted and the unit-weight gradient against w * (L x) -> L^T -> r + lam * rb."""
    from trips_py_amd.operators import SpaceTimeDerivative
    L = SpaceTimeDerivative(N, nt, engine=eng)
    assert L.fused_tv
    n = N * N * nt
    g = torch.Generator(device=eng.device).manual_seed(N + nt)
    x = torch.randn(n, device=eng.device, generator=g)
    r = torch.randn(n, device=eng.device, generator=g)
    lam, eps = 0.41, 0.1
    lx = L.apply(x)
    w_ref, w = eng.empty(L.shape[0]), eng.empty(L.shape[0])
    eng.mm_weights(lx, None, eps, q, w_ref)
    L.tv_weights(x, eps, q, w)
    assert torch.equal(w, w_ref)
    tp, rb, want, got = eng.empty(L.shape[0]), eng.empty(n), eng.empty(n), eng.empty(n)
    for ww in (w_ref, None):
        if ww is None:
            tp.copy_(lx)
        else:
            eng.mul(ww, lx, tp)
        L.apply(tp, out=rb, transpose=True)
        eng.axpby(1.0, r, lam, rb, want)
        L.tv_grad(x, ww, r, lam, out=got)
        assert relerr(got.cpu().numpy(), want.cpu().numpy()) < 3e-7, ww is None
