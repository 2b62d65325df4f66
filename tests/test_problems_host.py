"""The demos' host-side data helpers (gen_data / add_noise, Deblurring2D.py:123-159) — tools/demo_helpers.py, outside the package and
outside the hot-path scope — still follow the reference's recipes: no GPU."""
import os
import sys
import types

import numpy as np
from scipy.ndimage import convolve

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
from demo_helpers import demo_classes  # noqa: E402

_D1, _D2, _T = demo_classes()
P = types.SimpleNamespace(Deblurring1D=_D1, Deblurring2D=_D2, Tomography=_T)


def test_gen_data_and_add_noise_follow_the_reference_recipe():
    nx, ny = 24, 20
    rng = np.random.default_rng(0)
    x = rng.random(nx * ny)
    for crime in (False, True):
        D = P.Deblurring2D(CommitCrime=crime)
        D.nx, D.ny = nx, ny                                   # what forward_Op records (it needs the GPU engine)
        psf, centre = D.Gauss((7, 7), (2, 3))
        assert np.isclose(psf.sum(), 1.0) and tuple(centre) == (3, 3)
        b = D.gen_data(x)
        assert b.shape == (nx * ny, 1)
        if crime:
            want = convolve(x.reshape(nx, ny), psf, mode="reflect")
        else:
            big = np.zeros((2 * nx, 2 * ny))
            big[nx // 2:nx // 2 + nx, ny // 2:ny // 2 + ny] = x.reshape(nx, ny)
            want = convolve(big, psf, mode="constant")[nx // 2:nx // 2 + nx, ny // 2:ny // 2 + ny]
        assert np.array_equal(b.reshape(nx, ny), want)
    np.random.seed(5)
    bm, delta = D.add_noise(b, "Gaussian", 0.02)
    assert bm.shape == (nx, ny)
    assert np.isclose(np.linalg.norm(bm.reshape(-1, 1) - b), delta) and np.isclose(delta / np.linalg.norm(b), 0.02)
    bm, delta = D.add_noise(b, "Poisson", 0.0)
    assert bm.shape == (nx, ny) and delta == 0.0


def test_deblurring1d_helpers_reproduce_config_c1_inputs():
    """Deblurring1D.gen_xtrue / gen_data (Deblurring1D.py:104-197) against the golden of BASELINE config C1, which was produced
    by the reference itself (n = 256, 'curve0', sigma = 3, inverse crime)."""
    from conftest import load_golden
    g = load_golden("deblur1d_cgls_n256")
    n = int(g["n"])
    D = P.Deblurring1D(CommitCrime=True)
    x = D.gen_xtrue(n, "curve0")
    assert np.array_equal(x, g["x_true"].reshape(-1))
    b = D.gen_data(x, parameter=3.0)
    assert np.allclose(b.reshape(-1), g["b_true"].reshape(-1), rtol=0, atol=1e-15)
    assert np.allclose(D.PSF, g["psf"].reshape(-1))
    for test, shape in (("sigma", (n,)), ("piecewise", (n,)), ("curve1", (n, 1)), ("curve2", (n, 1)), ("curve3", (n, 1))):
        assert D.gen_xtrue(n, test).shape == shape
    np.random.seed(0)
    bm, delta = D.add_noise(b, "Gaussian", 0.05)
    assert np.isclose(delta / np.linalg.norm(b), 0.05) and bm.shape == b.shape
