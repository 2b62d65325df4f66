"""Full-size (BASELINE 4096^2) parity of the engine's CGLS against an independent float64 restatement that runs on the
GPU through torch (F.conv2d on a reflect-padded image = scipy.ndimage.convolve(mode='reflect'); CGLS.py:45-80 loop).
The oracle proper (NumPy/SciPy, oracle/cpu_ref.py) is too slow at this size for a unit test (1.7 s per apply), so the
torch float64 path is first pinned to the oracle at 256^2, then used as the checker at 4096^2."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import relerr

pytestmark = pytest.mark.gpu


def blur64(img, psf_t, flip=False):
    """float64 reflect-boundary convolution on the GPU.  img: (nx,ny) float64 tensor."""
    k = psf_t.flip(0, 1) if not flip else psf_t          # conv2d correlates: convolution = correlate with flipped PSF
    kh, kw = k.shape
    top, left = kh - 1 - kh // 2, kw - 1 - kw // 2
    p = F.pad(img[None, None], (left, kw // 2, top, kh // 2), mode="reflect") if False else None
    # torch 'reflect' excludes the edge sample; scipy 'reflect' repeats it = torch 'symmetric' -> build by hand
    x = torch.cat([img[:top].flip(0), img, img[-(kh // 2):].flip(0)] if kh // 2 else [img[:top].flip(0), img], 0)
    x = torch.cat([x[:, :left].flip(1), x, x[:, -(kw // 2):].flip(1)] if kw // 2 else [x[:, :left].flip(1), x], 1)
    return F.conv2d(x[None, None], k[None, None])[0, 0]


def cgls64(psf, N, b, iters, x_true):
    psf_t = torch.from_numpy(psf).to(b.device, torch.float64)
    A = lambda v: blur64(v.reshape(N, N), psf_t).reshape(-1)
    AT = lambda v: blur64(v.reshape(N, N), psf_t, flip=True).reshape(-1)
    x = torch.zeros_like(b)
    r = b - A(x)
    t = AT(r)
    p = t.clone()
    gamma = torch.dot(t, t)
    rel = []
    hist = {}
    for k in range(1, iters + 1):
        w = A(p)
        step = gamma / torch.dot(w, w)
        x = x + step * p
        r = r - step * w
        t = AT(r)
        g2 = torch.dot(t, t)
        p = t + (g2 / gamma) * p
        gamma = g2
        rel.append(float(torch.linalg.norm(x - x_true) / torch.linalg.norm(x)))
        if k in (10, 30, iters):
            hist[k] = x.clone()
    return hist, rel


def make_problem(N, dev):
    from trips_py_amd.problems import gauss_psf
    psf, _ = gauss_psf((9, 9), (3, 3))
    g = torch.Generator(device="cpu").manual_seed(0)
    img = torch.zeros((N, N), dtype=torch.float64)
    rr = torch.randint(0, N - N // 8, (8, 2), generator=g)
    hw = torch.randint(N // 16, N // 3, (8, 2), generator=g)
    for q in range(8):
        img[rr[q, 0]:rr[q, 0] + hw[q, 0], rr[q, 1]:rr[q, 1] + hw[q, 1]] += 0.2 + 0.1 * q
    img += 0.1 * torch.rand((N, N), generator=g, dtype=torch.float64)
    xt = img.reshape(-1).to(dev)
    psf_t = torch.from_numpy(psf).to(dev, torch.float64)
    b = blur64(xt.reshape(N, N), psf_t).reshape(-1)
    e = torch.randn(N * N, generator=g, dtype=torch.float64).to(dev)
    b = b + e * (0.01 * torch.linalg.norm(b) / torch.linalg.norm(e))
    return psf, xt, b


def test_torch64_checker_is_the_oracle_at_256():
    from oracle import cpu_ref as O
    dev = torch.device("cuda")
    N = 256
    psf, xt, b = make_problem(N, dev)
    psf_t = torch.from_numpy(psf).to(dev, torch.float64)
    Ao = O.Blur2D(psf, N, N)
    assert relerr(blur64(xt.reshape(N, N), psf_t).cpu().numpy(), (Ao @ xt.cpu().numpy()).reshape(N, N)) < 1e-13
    rng = np.random.default_rng(0)
    psf_a = rng.random((4, 7))
    v = rng.standard_normal((40, 52))
    got = blur64(torch.from_numpy(v).to(dev), torch.from_numpy(psf_a).to(dev)).cpu().numpy()
    assert relerr(got, O.blur2d_scipy(v, psf_a)) < 1e-13
    hist, rel = cgls64(psf, N, b, 30, xt)
    xo, io = O.cgls(Ao, b.cpu().numpy(), np.zeros((N * N, 1)), 30, 0, x_true=xt.cpu().numpy())
    assert relerr(hist[30].cpu().numpy(), xo) < 1e-9
    assert np.allclose(rel, io["relError"], rtol=1e-9)


@pytest.mark.parametrize("N,iters", [(1024, 60), (4096, 60)])
def test_cgls_fullsize_vs_float64(N, iters):
    from trips_py_amd.operators import Blur2D
    from trips_py_amd.solvers import CGLS
    dev = torch.device("cuda")
    psf, xt, b = make_problem(N, dev)
    hist, rel = cgls64(psf, N, b, iters, xt)
    A = Blur2D(psf, N, N)
    x, info = CGLS(A, b.float(), torch.zeros(N * N, device=dev), iters, 0, x_true=xt.float())
    for k in (10, 30, iters):
        e = float(torch.linalg.norm(info["xHistory"][k - 1].reshape(-1).double() - hist[k]) / torch.linalg.norm(hist[k]))
        assert e < 1e-5, (k, e)
    assert np.allclose(info["relError"], rel, rtol=1e-4)
