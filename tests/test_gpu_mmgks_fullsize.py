"""BASELINE config C4 at its real size: MMGKS (l2-l1, TV-like) on the 4096^2 blur against an independent float64
restatement of trips/solvers/MMGKS.py:37-128 that runs on the GPU through torch (economic QRs by torch.linalg.qr, the
blur as F.conv2d on a symmetric-padded image, the 2-D first-derivative matrix of operators.py:30-36 as slicing).  The
oracle proper (oracle/cpu_ref.py) needs minutes at this size, so the torch float64 path is first pinned to the oracle at
256^2 (1e-9), then used as the checker at 4096^2 (bar: 1e-5 on every iterate, north_star's tolerance at fixed lambda)."""
import numpy as np
import pytest
import torch

from conftest import relerr
from test_gpu_fullsize import blur64, make_problem

pytestmark = pytest.mark.gpu


def d2_fwd(x, N):
    """gen_first_derivative_operator_2D (operators.py:30-36): rows x[i,j] - x[i,j+1] (i-major), then x[i,j] - x[i+1,j]."""
    X = x.reshape(N, N)
    return torch.cat([(X[:, :-1] - X[:, 1:]).reshape(-1), (X[:-1, :] - X[1:, :]).reshape(-1)])


def d2_adj(y, N):
    a, b = y[:N * (N - 1)].reshape(N, N - 1), y[N * (N - 1):].reshape(N - 1, N)
    out = torch.zeros((N, N), dtype=y.dtype, device=y.device)
    out[:, :-1] += a
    out[:, 1:] -= a
    out[:-1, :] += b
    out[1:, :] -= b
    return out.reshape(-1)


def mv(M, y):
    """M @ y without BLAS (rocBLAS dgemv refuses 16.8 M-row operands)."""
    return (M * y[None, :]).sum(dim=1)


def mtv(M, v):
    return (M * v[:, None]).sum(dim=0)


def mmgks64(psf, N, b, d, n_iter, lam, eps, pnorm, qnorm):
    """MMGKS.py:37-128, plain smoothed-Holder weights, numeric regparam; returns the list of iterates (float64, device)."""
    psf_t = torch.from_numpy(psf).to(b.device, torch.float64)
    A = lambda v: blur64(v.reshape(N, N), psf_t).reshape(-1)
    AT = lambda v: blur64(v.reshape(N, N), psf_t, flip=True).reshape(-1)
    # golub_kahan(A, b, d): decompositions.py:118-205 without reorthogonalisation
    u = b / torch.linalg.norm(b)
    V = []
    beta_prev, v_prev = None, None
    for k in range(d):
        v = AT(u)
        if k:
            v = v - beta_prev * v_prev
        alpha = torch.linalg.norm(v)
        v = v / alpha
        un = A(v) - alpha * u
        beta_prev = torch.linalg.norm(un)
        u = un / beta_prev
        v_prev = v
        V.append(v)
    V = torch.stack(V, 1)
    x = AT(b)                                                                  # :43
    AV = torch.stack([A(V[:, j]) for j in range(V.shape[1])], 1)
    LV = torch.stack([d2_fwd(V[:, j], N) for j in range(V.shape[1])], 1)
    hist = []
    for ii in range(n_iter):
        wf = ((A(x) - b) ** 2 + eps ** 2) ** (pnorm / 2 - 1)                   # :56-57
        Q_A, R_A = torch.linalg.qr(AV * wf[:, None])
        wr = (d2_fwd(x, N) ** 2 + eps ** 2) ** (qnorm / 2 - 1)                 # :60,93
        _, R_L = torch.linalg.qr(LV * wr[:, None])
        k = R_A.shape[0]
        M = torch.cat([R_A, np.sqrt(lam) * R_L])
        rhs = torch.cat([mtv(Q_A, b), torch.zeros(k, dtype=b.dtype, device=b.device)])   # UNWEIGHTED b (:106)
        y = torch.linalg.lstsq(M, rhs[:, None]).solution[:, 0]
        x = mv(V, y)
        hist.append(x.clone())
        if ii >= R_L.shape[0]:
            break
        r = AT(wf * (mv(AV, y) - b)) + lam * d2_adj(wr * mv(LV, y), N)            # :114-118
        for _ in range(2):
            r = r - mv(V, mtv(V, r))
        vn = r / torch.linalg.norm(r)
        V = torch.cat([V, vn[:, None]], 1)
        AV = torch.cat([AV, A(vn)[:, None]], 1)
        LV = torch.cat([LV, d2_fwd(vn, N)[:, None]], 1)
    return hist


def test_torch64_mmgks_checker_is_the_oracle_at_256():
    from oracle import cpu_ref as O
    dev = torch.device("cuda")
    N = 256
    psf, xt, b = make_problem(N, dev)
    v = torch.randn(N * N, dtype=torch.float64, device=dev, generator=torch.Generator(device=dev).manual_seed(3))
    Lo = O.FirstDerivative2D(N)
    assert relerr(d2_fwd(v, N).cpu().numpy(), Lo @ v.cpu().numpy()) < 1e-14
    w = torch.randn(2 * N * (N - 1), dtype=torch.float64, device=dev, generator=torch.Generator(device=dev).manual_seed(4))
    assert relerr(d2_adj(w, N).cpu().numpy(), Lo.T @ w.cpu().numpy()) < 1e-14
    for q in (1, 0.5):
        hist = mmgks64(psf, N, b, 3, 8, 1e-2, 0.1, 2, q)
        xo, io = O.mmgks(O.Blur2D(psf, N, N), b.cpu().numpy().reshape(-1, 1), Lo, 2, q, 3, 8, 1e-2, epsilon=0.1)
        assert len(hist) == len(io["xHistory"])
        for k in range(len(hist)):
            assert relerr(hist[k].cpu().numpy(), io["xHistory"][k].reshape(-1)) < 1e-9, (q, k)


@pytest.mark.parametrize("N,n_iter", [(1024, 8), (4096, 30)])
def test_c4_mmgks_fullsize_vs_float64(N, n_iter):
    """(4096, 30) is the solve bench.py times for C4: the basis grows from 3 to 33 vectors, so the two- and three-tile
    instantiations of the re-weighted Gram kernels (k_wgram_tv<2>, <3>, trk_wgram_tv_z) meet the float64 checker at the size they are
    timed at.  Measured (tools/c4_parity.py, profiles/r04/c4_parity_30.txt): 6e-8 ... 1.6e-6 over the 30 iterates; the checker needs
    ~2 minutes and 44 GiB of device memory (float64 QRs of 33.5 M x 33)."""
    from trips_py_amd import solvers as S
    from trips_py_amd.operators import Blur2D, FirstDerivative2D
    dev = torch.device("cuda")
    psf, xt, b = make_problem(N, dev)
    b32 = b.float()
    hist = mmgks64(psf, N, b32.double(), 3, n_iter, 1e-2, 0.1, 2, 1)
    x, info = S.MMGKS(Blur2D(psf, N, N), b32, FirstDerivative2D(N), 2, 1, 3, n_iter, 1e-2, epsilon=0.1)
    assert len(info["xHistory"]) == len(hist) == n_iter
    for k in range(n_iter):
        e = float(torch.linalg.norm(info["xHistory"][k].reshape(-1).double() - hist[k]) / torch.linalg.norm(hist[k]))
        assert e < 1e-5, (k, e)
    del hist
    torch.cuda.empty_cache()


def test_gks_4096_one_pass_form_equals_the_pass_each_form():
    """GKS on the C4 problem at 4096^2 (the leg bench.py times beside MMGKS): the iteration with ONE pass over the basis for the new vector
    and the next iterate (trk_gemv_orth_iterate — the next projected problem solved first, from the h-sweep's products) against the
    reference's order with a pass each, 30 iterations, basis of 3 .. 33 vectors of 67 MB: every 5th iterate, the residual norms and the
    errors.  (Both forms were held to the float64 oracle at the sizes it reaches: tests/test_gpu_solvers.py, test_gpu_configs_fullsize.py.)"""
    from trips_py_amd import solvers as S
    from trips_py_amd.operators import Blur2D, FirstDerivative2D
    dev = torch.device("cuda")
    N = 4096
    psf, xt, b = make_problem(N, dev)
    b32, xt32 = b.float(), xt.float()
    del b, xt
    A, L = Blur2D(psf, N, N), FirstDerivative2D(N)
    keep = {}
    for tag, kw in (("one", {}), ("each", {"fused_orth_iterate": False})):
        x, info = S.GKS(A, b32, L, 3, 30, 1e-2, xt32, history=5, **kw)
        keep[tag] = (x.clone(), [h.reshape(-1).clone() for h in info["xHistory"]], np.array(info["Residual"]), np.array(info["relError"]))
        del x, info
        torch.cuda.empty_cache()
    (xa, ha, ra, ea), (xb, hb, rb, eb) = keep["one"], keep["each"]
    assert float(torch.linalg.norm(xa - xb) / torch.linalg.norm(xb)) < 1e-6
    assert len(ha) == len(hb) and len(ha) >= 6
    for u, v in zip(ha, hb):
        assert float(torch.linalg.norm(u - v) / torch.linalg.norm(v)) < 1e-6
    assert np.allclose(ra, rb, rtol=1e-4) and np.allclose(ea, eb, rtol=1e-5)


def gks64(psf, N, b, d, n_iter, lam):
    """GKS.py:36-96 with a numeric regparam and the 2-D first-derivative L, float64 on the device (the QRs of AV and LV from scratch every
    iteration, three Gram-Schmidt sweeps :86-88); returns (iterates, residual norms)."""
    psf_t = torch.from_numpy(psf).to(b.device, torch.float64)
    A = lambda v: blur64(v.reshape(N, N), psf_t).reshape(-1)
    AT = lambda v: blur64(v.reshape(N, N), psf_t, flip=True).reshape(-1)
    u = b / torch.linalg.norm(b)                               # golub_kahan(A, b, d): decompositions.py:118-205 without reorthogonalisation
    V, beta_prev, v_prev = [], None, None
    for k in range(d):
        v = AT(u)
        if k:
            v = v - beta_prev * v_prev
        alpha = torch.linalg.norm(v)
        v = v / alpha
        un = A(v) - alpha * u
        beta_prev = torch.linalg.norm(un)
        u = un / beta_prev
        v_prev = v
        V.append(v)
    V = torch.stack(V, 1)
    AV = torch.stack([A(V[:, j]) for j in range(V.shape[1])], 1)
    LV = torch.stack([d2_fwd(V[:, j], N) for j in range(V.shape[1])], 1)
    hist, res = [], []
    for ii in range(n_iter):
        Q_A, R_A = torch.linalg.qr(AV)
        _, R_L = torch.linalg.qr(LV)
        k = R_A.shape[0]
        M = torch.cat([R_A, np.sqrt(lam) * R_L])
        rhs = torch.cat([mtv(Q_A, b), torch.zeros(k, dtype=b.dtype, device=b.device)])
        y = torch.linalg.lstsq(M, rhs[:, None]).solution[:, 0]
        x = mv(V, y)
        hist.append(x.clone())
        r = AT(mv(AV, y) - b) + lam * d2_adj(mv(LV, y), N)     # :81-85
        for _ in range(3):
            r = r - mv(V, mtv(V, r))
        nr = torch.linalg.norm(r)
        res.append(float(nr))
        vn = r / nr
        V = torch.cat([V, vn[:, None]], 1)
        AV = torch.cat([AV, A(vn)[:, None]], 1)
        LV = torch.cat([LV, d2_fwd(vn, N)[:, None]], 1)
    return hist, res


def test_torch64_gks_checker_is_the_oracle_at_128():
    from oracle import cpu_ref as O
    dev = torch.device("cuda")
    N = 128
    psf, xt, b = make_problem(N, dev)
    hist, res = gks64(psf, N, b, 3, 8, 1e-2)
    xo, io = O.gks(O.Blur2D(psf, N, N), b.cpu().numpy().reshape(-1, 1), O.FirstDerivative2D(N), 3, 8, 1e-2)
    assert len(hist) == len(io["xHistory"]) == 8
    for k in range(8):
        assert relerr(hist[k].cpu().numpy(), io["xHistory"][k].reshape(-1)) < 1e-9, k
    assert np.allclose(res, io["Residual"], rtol=1e-9)


@pytest.mark.parametrize("N,n_iter", [(1024, 30), (2048, 20)])
def test_gks_blur_tv_large_vs_float64(N, n_iter):
    """GKS on the blur with the 2-D first-derivative regulariser at 1024^2 / 2048^2 — the stencil-operator path: no images A v_j, L v_j kept,
    both Gram matrices' rows from the orthogonalisation sweep's products, the next projected problem solved before the new vector exists and
    ONE pass over the basis for it and the next iterate (docs/kernels/projected_gks_mmgks.md 4.3c) — against the float64 restatement of
    GKS.py on the device, pinned to the oracle above: 1e-5 on every iterate (north_star's tolerance at fixed lambda), the residual norms to 1e-3."""
    from trips_py_amd import solvers as S
    from trips_py_amd.operators import Blur2D, FirstDerivative2D
    dev = torch.device("cuda")
    psf, xt, b = make_problem(N, dev)
    b32 = b.float()
    hist, res = gks64(psf, N, b32.double(), 3, n_iter, 1e-2)
    x, info = S.GKS(Blur2D(psf, N, N), b32, FirstDerivative2D(N), 3, n_iter, 1e-2)
    assert len(info["xHistory"]) == len(hist) == n_iter
    worst = 0.0
    for k in range(n_iter):
        e = float(torch.linalg.norm(info["xHistory"][k].reshape(-1).double() - hist[k]) / torch.linalg.norm(hist[k]))
        worst = max(worst, e)
        assert e < 1e-5, (k, e)
    assert np.allclose(info["Residual"], res, rtol=1e-3)
    print(f"GKS {N}^2: worst iterate distance from float64 {worst:.2e}")
    del hist
    torch.cuda.empty_cache()
