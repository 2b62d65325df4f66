"""Degenerate and ragged sizes through the C ABI, against the oracle: images smaller than the PSF, single rows / columns,
one-pixel images, one angle, one frame, one-column bases, vectors shorter than a wave."""
import numpy as np
import pytest
import torch

from conftest import relerr

pytestmark = pytest.mark.gpu

f32 = lambda a: np.asarray(a, dtype=np.float64).astype(np.float32).astype(np.float64)


@pytest.mark.parametrize("nx,ny,kh,kw", [(1, 1, 3, 3), (1, 7, 9, 9), (7, 1, 9, 9), (2, 3, 9, 9), (4, 4, 9, 9), (5, 8, 3, 5),
                                         (8, 8, 9, 9), (3, 260, 9, 9), (260, 4, 9, 9), (16, 12, 2, 4)])
def test_blur_tiny_and_ragged_images(nx, ny, kh, kw):
    """scipy.ndimage.convolve(mode='reflect') semantics hold when the reflection wraps more than once (image shorter than
    the PSF half-width) and for even PSFs."""
    from oracle import cpu_ref as O
    from trips_py_amd.operators import Blur2D
    rng = np.random.default_rng(nx * 100 + ny)
    psf = rng.random((kh, kw))
    psf /= psf.sum()
    A, Ao = Blur2D(psf, nx, ny), O.Blur2D(psf, nx, ny)
    x = rng.standard_normal(nx * ny)
    assert relerr(A @ x, Ao @ f32(x)) < 1e-5
    assert relerr(A.T @ x, Ao.T @ f32(x)) < 1e-5
    X = rng.standard_normal((nx * ny, 3))                     # (n, k) operand: GKS.py:37
    assert relerr(A @ X, Ao @ f32(X)) < 1e-5


@pytest.mark.parametrize("N,na,nd", [(1, 1, 1), (2, 1, 2), (3, 2, 5), (4, 3, 4), (5, 1, 9), (8, 4, 3)])
def test_radon_tiny(N, na, nd):
    from oracle import cpu_ref as O
    from trips_py_amd.operators import Radon2DParallel
    ang = np.linspace(0.2, 2.9, na)
    R, Ro = Radon2DParallel(N, ang, n_det=nd), O.Radon2D(N, ang, n_det=nd)
    rng = np.random.default_rng(N)
    x, y = rng.random(N * N), rng.standard_normal(na * nd)
    assert np.allclose(R @ x, Ro @ f32(x), rtol=1e-5, atol=1e-6)
    assert np.allclose(R.T @ y, Ro.T @ f32(y), rtol=1e-5, atol=1e-6)


def test_derivatives_smallest_sizes():
    from oracle import cpu_ref as O
    from trips_py_amd.operators import FirstDerivative2D, SpaceTimeDerivative
    rng = np.random.default_rng(0)
    for N in (2, 3):
        L, Lo = FirstDerivative2D(N), O.FirstDerivative2D(N)
        x, y = rng.standard_normal(N * N), rng.standard_normal(Lo.shape[0])
        assert np.allclose(L @ x, Lo @ f32(x), atol=1e-6) and np.allclose(L.T @ y, Lo.T @ f32(y), atol=1e-6)
    for N, nt in ((2, 1), (2, 2), (3, 1)):
        L, Lo = SpaceTimeDerivative(N, nt), O.SpaceTimeDerivative(N, nt)
        assert L.shape == Lo.shape
        x, y = rng.standard_normal(Lo.shape[1]), rng.standard_normal(Lo.shape[0])
        assert np.allclose(L @ x, Lo @ f32(x), atol=1e-6) and np.allclose(L.T @ y, Lo.T @ f32(y), atol=1e-6)


@pytest.mark.parametrize("n", [1, 2, 3, 63, 65])
def test_vector_kernels_shorter_than_a_wave(n):
    from trips_py_amd.engine import Coef, default_engine
    eng = default_engine()
    rng = np.random.default_rng(n)
    x, y = rng.standard_normal(n), rng.standard_normal(n)
    dx, dy = eng.to_vec(x), eng.to_vec(y)
    S = eng.scalars(4)
    eng.dot(dx, dy, S.ref(0))
    eng.nrm2sq(dx, S.ref(1))
    out = eng.empty(n)
    eng.axpby(2.0, dx, Coef(-1.0, num=S.ref(1), sqrt_num=True), dy, out, sumsq=S.ref(2))
    h = S.host()
    xf, yf = f32(x), f32(y)
    assert np.isclose(h[0], xf @ yf, rtol=1e-12, atol=1e-12) and np.isclose(h[1], xf @ xf, rtol=1e-12)
    want = 2.0 * xf - np.sqrt(xf @ xf) * yf
    assert np.allclose(out.cpu().numpy(), want, rtol=2e-6, atol=1e-6)
    assert np.isclose(h[2], float((out.double() ** 2).sum()), rtol=1e-10)
    # one-column basis
    V = eng.empty_basis(1, n)
    V[0].copy_(dx)
    H = eng.scalars(1)
    eng.gemv_t(V, 1, dy, H.ref(0))
    assert np.isclose(H.host()[0], xf @ yf, rtol=1e-12, atol=1e-12)
    eng.gemv_n(V, 1, H.ref(0), out, a=1.0, base=dy, s=-1.0)
    assert np.allclose(out.cpu().numpy(), yf - (xf @ yf) * xf, rtol=1e-5, atol=1e-6)


def test_solvers_on_a_tiny_problem():
    """CGLS / Hybrid_LSQR / GKS on a 6x6 image: every kernel sees sizes far below its tile shapes."""
    from oracle import cpu_ref as O
    from trips_py_amd import solvers as S
    from trips_py_amd.operators import Blur2D, FirstDerivative2D
    N = 6
    rng = np.random.default_rng(1)
    psf = np.outer([0.25, 0.5, 0.25], [0.25, 0.5, 0.25])
    A, Ao = Blur2D(psf, N, N), O.Blur2D(psf, N, N)
    xt = rng.random(N * N)
    b = f32(Ao @ xt) + 1e-3 * rng.standard_normal(N * N)
    x, info = S.CGLS(A, b, np.zeros(N * N), 6, 0)
    xo, _ = O.cgls(Ao, b.reshape(-1, 1), np.zeros((N * N, 1)), 6, 0)
    assert relerr(x, xo.reshape(-1)) < 1e-4
    x, info = S.Hybrid_LSQR(A, b, 5, 1e-2)
    xo, _ = O.hybrid_lsqr(Ao, b.reshape(-1, 1), 5, 1e-2)
    assert relerr(x, xo.reshape(-1)) < 1e-4
    L, Lo = FirstDerivative2D(N), O.FirstDerivative2D(N)
    x, info = S.GKS(A, b, L, 2, 4, 1e-2)
    xo, _ = O.gks(Ao, b.reshape(-1, 1), Lo, 2, 4, 1e-2)
    assert relerr(x, xo.reshape(-1)) < 1e-4


def test_scalar_block_download_by_mailbox_equals_the_tensor_copy():
    """DevScalars.host() goes through the block's pinned mailbox for up to HOST_BY_MAILBOX_MAX doubles (one small launch + a poll instead of a
    staged tensor copy) and through torch beyond: the same numbers either way, for whole blocks, slices, negative and open bounds."""
    from trips_py_amd.engine import default_engine
    eng = default_engine()
    rng = np.random.default_rng(5)
    for n in (1, 7, 64, 256, 257, 1000, 5000):
        S = eng.scalars(n)
        v = rng.standard_normal(n)
        S.t.copy_(torch.as_tensor(v))
        ref = S.t.detach().to("cpu").numpy()
        assert np.array_equal(S.host(), ref)
        assert np.array_equal(S.host(0, n), ref)
        if n >= 7:
            assert np.array_equal(S.host(2, 6), ref[2:6])
            assert np.array_equal(S.host(3, None), ref[3:])
            assert np.array_equal(S.host(1, -1), ref[1:-1])
        a = S.host(0, 1)
        a[0] = 123.0                                   # a copy: the block is untouched
        assert S.host(0, 1)[0] == ref[0]
