"""How far the fp32 ray coordinates put the projector from the float64 oracle: measured bounds, so that a
regression in the coordinate arithmetic shows up.  The crossing position is one fp32 FMA on values up to N, i.e. an
interpolation weight is off by up to ~6e-8 N; on a smooth image the errors average out, on white noise they do not."""
import numpy as np
import pytest

from conftest import relerr

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N", [512, 2048])
def test_projector_accuracy_against_float64_oracle(N):
    from oracle import cpu_ref as O
    from trips_py_amd.operators import Radon2DParallel
    ang = np.array([0.3, 1.1, 2.0])
    R, Ro = Radon2DParallel(N, ang), O.Radon2D(N, ang)
    rng = np.random.default_rng(0)
    ii, jj = np.meshgrid(np.arange(N), np.arange(N), indexing="ij")
    smooth = (np.exp(-((ii - N / 2.5) ** 2 + (jj - N / 1.7) ** 2) / (0.02 * N * N)) + 0.05 * rng.random((N, N))).reshape(-1)
    f = lambda v: v.astype(np.float32).astype(np.float64)
    assert relerr(R @ smooth, Ro @ f(smooth)) < 1e-6                       # measured 1.7e-7 (512), 1.8e-7 (2048), 3.3e-7 (4096)
    noise = rng.standard_normal(Ro.shape[0])
    e = relerr(R.T @ noise, Ro.T @ f(noise))
    assert e < 1e-7 * N, e                                                  # measured 2.5e-5 (512), 7.6e-5 (2048), 1.9e-4 (4096)
    xn = rng.standard_normal(N * N)
    e = relerr(R @ xn, Ro @ f(xn))
    assert e < 1e-7 * N, e
