"""Seeded random shapes through every operator, against the oracle: sizes that are not multiples of the kernels' tile
shapes, PSFs of every parity, detectors wider / narrower than the image, tiny frame counts."""
import numpy as np
import pytest

from conftest import relerr

pytestmark = pytest.mark.gpu

f32 = lambda a: np.asarray(a, dtype=np.float64).astype(np.float32).astype(np.float64)


@pytest.mark.parametrize("seed", range(24))
def test_blur_random_shapes(seed):
    from oracle import cpu_ref as O
    from trips_py_amd.operators import Blur2D
    rng = np.random.default_rng(1000 + seed)
    nx, ny = int(rng.integers(1, 300)), int(rng.integers(1, 300))
    if seed % 4 == 0:
        ny = 4 * int(rng.integers(2, 80))                        # the sliding-window kernel's column condition
    kh, kw = int(rng.integers(1, 12)), int(rng.integers(1, 12))
    if seed % 3 == 0:                                             # separable odd PSF <= 9: the fast kernels
        k = int(rng.choice([3, 5, 7, 9]))
        a, b = rng.random(k) + 0.1, rng.random(k) + 0.1
        psf = np.outer(a, b)
    else:
        psf = rng.random((kh, kw)) + 0.01
    psf = psf / psf.sum()
    A, Ao = Blur2D(psf, nx, ny), O.Blur2D(psf, nx, ny)
    x = rng.standard_normal(nx * ny)
    assert relerr(A @ x, Ao @ f32(x)) < 1e-5, (nx, ny, psf.shape)
    assert relerr(A.T @ x, Ao.T @ f32(x)) < 1e-5, (nx, ny, psf.shape)


@pytest.mark.parametrize("seed", range(16))
def test_radon_random_shapes(seed):
    from oracle import cpu_ref as O
    from trips_py_amd.operators import Radon2DParallel
    rng = np.random.default_rng(2000 + seed)
    N = int(rng.integers(1, 200)) if seed % 4 else 4 * int(rng.integers(256, 300))   # every 4th: the window-sharing kernel
    na = int(rng.integers(1, 10))
    nd = int(rng.integers(max(1, N // 2), 2 * N + 2))
    ang = rng.uniform(-1.0, 4.0, na) if seed % 2 else np.sort(rng.uniform(0, np.pi, na))
    R, Ro = Radon2DParallel(N, ang, n_det=nd), O.Radon2D(N, ang, n_det=nd)
    ii, jj = np.meshgrid(np.arange(N), np.arange(N), indexing="ij")
    x = (np.cos(ii / 7.0) * np.sin(jj / 5.0) + 0.1 * rng.random((N, N))).reshape(-1)
    y = rng.standard_normal(na * nd)
    tol = max(2e-5, 2e-7 * N)
    assert np.linalg.norm(R @ x - Ro @ f32(x)) <= tol * max(np.linalg.norm(Ro @ f32(x)), 1e-12), (N, na, nd)
    assert np.linalg.norm(R.T @ y - Ro.T @ f32(y)) <= tol * max(np.linalg.norm(Ro.T @ f32(y)), 1e-12), (N, na, nd)


@pytest.mark.parametrize("seed", range(8))
def test_derivative_random_shapes(seed):
    from oracle import cpu_ref as O
    from trips_py_amd.operators import FirstDerivative2D, SpaceTimeDerivative
    rng = np.random.default_rng(3000 + seed)
    N, nt = int(rng.integers(2, 90)), int(rng.integers(1, 6))
    for L, Lo in ((FirstDerivative2D(N), O.FirstDerivative2D(N)), (SpaceTimeDerivative(N, nt), O.SpaceTimeDerivative(N, nt))):
        x, y = rng.standard_normal(Lo.shape[1]), rng.standard_normal(Lo.shape[0])
        assert np.allclose(L @ x, Lo @ f32(x), rtol=1e-5, atol=1e-5) and np.allclose(L.T @ y, Lo.T @ f32(y), rtol=1e-5, atol=1e-5)
