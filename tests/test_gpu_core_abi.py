"""The CORE of the C ABI: twenty entry points that alone run all five solvers of the path.

What a maintainer binding libtrk.so for the first time aims at.  This file binds exactly those functions with its own ctypes
declarations (nothing of trips_py_amd is imported: no engine, no operator classes, no fused fast paths) and writes the
reference's five solver loops on top of them the way trips/solvers/*.py write them over NumPy — vectors and bases are device
buffers (torch is used as the allocator only), k-sized float64 work stays on the host — then checks every solver against the float64
oracle.  Every other entry point of include/trk.h is an OPTIONAL fused form of something done here in several calls
(discoverable through trk_op_fused_caps / trk_op_axpby_caps / trk_cgls_tiled_caps).

Reference loops restated below: CGLS.py:16-86, decompositions.py:207-255 (arnoldi_update, golub_kahan_update),
Hybrid_LSQR.py:55-114, Hybrid_GMRES.py:23-87, GKS.py:27-105, MMGKS.py:28-137 (plain-weights branch)."""
import ctypes
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
c_vp, c_i64, c_int, c_dbl = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_double
P = ctypes.POINTER

# name: (restype, argtypes) — the whole binding
CORE = {
    "trk_last_error": (ctypes.c_char_p, []),
    # operators: create / shape / apply / destroy
    "trk_blur2d_create": (c_int, [P(c_dbl), c_int, c_int, c_int, c_int, P(c_vp)]),
    "trk_radon2d_create": (c_int, [c_int, c_int, P(c_dbl), c_int, c_dbl, P(c_vp)]),
    "trk_deriv2d_create": (c_int, [c_int, P(c_vp)]),
    "trk_spacetime_create": (c_int, [c_int, c_int, c_int, c_int, P(c_vp)]),
    "trk_op_shape": (c_int, [c_vp, P(c_i64), P(c_i64)]),
    "trk_op_apply": (c_int, [c_vp, c_int, c_vp, c_i64, c_vp, c_i64, c_int, c_vp, c_vp]),
    "trk_op_destroy": (c_int, [c_vp]),
    # vectors
    "trk_dot": (c_int, [c_vp, c_vp, c_i64, c_vp, c_vp]),
    "trk_nrm2sq": (c_int, [c_vp, c_i64, c_vp, c_vp]),
    "trk_axpby": (c_int, [c_i64, c_dbl, c_vp, c_vp, c_int, c_vp, c_dbl, c_vp, c_vp, c_int, c_vp, c_vp, c_vp, c_vp]),
    "trk_mul": (c_int, [c_i64, c_vp, c_vp, c_vp, c_vp]),
    "trk_mm_weights": (c_int, [c_i64, c_vp, c_vp, c_dbl, c_dbl, c_vp, c_vp]),
    # tall-skinny bases
    "trk_gemv_n": (c_int, [c_vp, c_i64, c_int, c_i64, c_vp, c_dbl, c_vp, c_dbl, c_vp, c_vp, c_vp]),
    "trk_gemv_t": (c_int, [c_vp, c_i64, c_int, c_i64, c_vp, c_vp, c_vp, c_vp]),
    "trk_wgram": (c_int, [c_vp, c_i64, c_int, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    # ranks (declared: the symbols must exist; a one-GPU box cannot run two RCCL ranks — tests/test_gpu_dist.py covers them)
    "trk_comm_unique_id": (c_int, [c_vp]),
    "trk_comm_init": (c_int, [c_vp, c_int, c_int, P(c_vp)]),
    "trk_allreduce_f64": (c_int, [c_vp, c_vp, c_int, c_vp]),
    "trk_halo_exchange2": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
}


class Core:
    def __init__(self):
        import torch  # noqa: F401  (torch's HIP runtime first: INTEGRATION.md section 4)
        self.lib = ctypes.CDLL(os.path.join(REPO, "trips_py_amd", "csrc", "libtrk.so"))
        for name, (res, args) in CORE.items():
            f = getattr(self.lib, name)
            f.restype, f.argtypes = res, args
        self.dev = torch.device("cuda", 0)
        self.S = torch.zeros(64, dtype=torch.float64, device=self.dev)      # device scalars the reductions write

    def ck(self, rc):
        if rc:
            raise RuntimeError(self.lib.trk_last_error().decode())

    def stream(self):
        return torch.cuda.current_stream().cuda_stream

    def vec(self, a):
        return torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.float32).reshape(-1))).to(self.dev)

    def empty(self, *shape):
        return torch.empty(shape, dtype=torch.float32, device=self.dev)

    def host(self, t):
        return t.detach().to("cpu").numpy().astype(np.float64)

    # ---- operators
    def blur(self, psf, nx, ny):
        psf = np.ascontiguousarray(psf, dtype=np.float64)
        h = c_vp()
        self.ck(self.lib.trk_blur2d_create(psf.ctypes.data_as(P(c_dbl)), psf.shape[0], psf.shape[1], nx, ny, ctypes.byref(h)))
        return Op(self, h)

    def deriv2d(self, N):
        h = c_vp()
        self.ck(self.lib.trk_deriv2d_create(N, ctypes.byref(h)))
        return Op(self, h)

    # ---- vector algebra (float results come back through one device double)
    def nrm2(self, x):
        self.ck(self.lib.trk_nrm2sq(x.data_ptr(), x.numel(), self.S.data_ptr(), self.stream()))
        return float(np.sqrt(self.S[0].item()))

    def dot(self, x, y):
        self.ck(self.lib.trk_dot(x.data_ptr(), y.data_ptr(), x.numel(), self.S.data_ptr(), self.stream()))
        return float(self.S[0].item())

    def axpby(self, a, x, b, y, out):
        """out = a x + b y (y may be None)."""
        self.ck(self.lib.trk_axpby(x.numel(), float(a), None, None, 0, x.data_ptr(), float(b), None, None, 0,
                                   None if y is None else y.data_ptr(), out.data_ptr(), None, self.stream()))

    def mul(self, x, y, out):
        self.ck(self.lib.trk_mul(x.numel(), x.data_ptr(), y.data_ptr(), out.data_ptr(), self.stream()))

    def mm_weights(self, x, y, eps, p, out):
        self.ck(self.lib.trk_mm_weights(x.numel(), x.data_ptr(), None if y is None else y.data_ptr(), float(eps), float(p),
                                        out.data_ptr(), self.stream()))

    def combine(self, V, k, y, out, a=0.0, base=None, s=1.0):
        """out = a base + s V[:k]^T y  (V row-per-vector, y a host k-vector)."""
        yd = torch.from_numpy(np.ascontiguousarray(y, dtype=np.float64).reshape(-1)).to(self.dev)
        self.ck(self.lib.trk_gemv_n(V.data_ptr(), V.stride(0), k, V.shape[1], yd.data_ptr(), float(a),
                                    None if base is None else base.data_ptr(), float(s), out.data_ptr(), None, self.stream()))

    def project(self, V, k, r):
        """V[:k] r as a host k-vector."""
        h = torch.zeros(k, dtype=torch.float64, device=self.dev)
        self.ck(self.lib.trk_gemv_t(V.data_ptr(), V.stride(0), k, V.shape[1], r.data_ptr(), None, h.data_ptr(), self.stream()))
        return self.host(h)

    def wgram(self, W, k, w, b1=None):
        """G = W diag(w^2) W^T (k x k) and, with b1, c1 = W (w .* b1), c2 = W (w^2 .* b1)."""
        G = torch.zeros(k * k + 2 * k, dtype=torch.float64, device=self.dev)
        self.ck(self.lib.trk_wgram(W.data_ptr(), W.stride(0), k, W.shape[1], None if w is None else w.data_ptr(),
                                   None if b1 is None else b1.data_ptr(), G.data_ptr(),
                                   None if b1 is None else G[k * k:].data_ptr(), None if b1 is None else G[k * k + k:].data_ptr(),
                                   self.stream()))
        g = self.host(G)
        return g[:k * k].reshape(k, k), g[k * k:k * k + k], g[k * k + k:]


class Op:
    def __init__(self, core, h):
        self.c, self.h = core, h
        m, n = c_i64(), c_i64()
        core.ck(core.lib.trk_op_shape(h, ctypes.byref(m), ctypes.byref(n)))
        self.shape = (m.value, n.value)

    def apply(self, x, out, transpose=False):
        self.c.ck(self.c.lib.trk_op_apply(self.h, int(transpose), x.data_ptr(), 0, out.data_ptr(), 0, 1, None, self.c.stream()))
        return out

    def __del__(self):
        try:
            self.c.lib.trk_op_destroy(self.h)
        except Exception:      # noqa: BLE001
            pass


# ------------------------------------------------------------------------------------------------ the five loops, core calls only
def cgls(c, A, b, x0, iters):                                                       # CGLS.py:45-80, tol = 0
    m, n = A.shape
    x, r, t, w, p = x0.clone(), c.empty(m), c.empty(n), c.empty(m), c.empty(n)
    A.apply(x, r)
    c.axpby(1.0, b, -1.0, r, r)
    A.apply(r, t, True)
    p.copy_(t)
    gamma = c.dot(t, t)
    for _ in range(iters):
        A.apply(p, w)
        alpha = gamma / c.dot(w, w)
        c.axpby(1.0, x, alpha, p, x)
        c.axpby(1.0, r, -alpha, w, r)
        A.apply(r, t, True)
        g_new = c.dot(t, t)
        c.axpby(1.0, t, g_new / gamma, p, p)
        gamma = g_new
    return c.host(x)


class GolubKahan:                                                                    # decompositions.py:230-255
    def __init__(self, c, A, b, kmax):
        self.c, self.A = c, A
        m, n = A.shape
        self.U, self.V = c.empty(kmax + 2, m), c.empty(kmax + 1, n)
        self.beta0 = c.nrm2(b)
        c.axpby(1.0 / self.beta0, b, 0.0, None, self.U[0])
        self.alphas, self.betas, self.k = [], [], 0

    def step(self):
        c, A, k = self.c, self.A, self.k
        v, u = self.V[k], self.U[k + 1]
        A.apply(self.U[k], v, True)
        if k:
            c.axpby(1.0, v, -self.betas[-1], self.V[k - 1], v)
        alpha = c.nrm2(v)
        c.axpby(1.0 / alpha, v, 0.0, None, v)
        A.apply(v, u)
        c.axpby(1.0, u, -alpha, self.U[k], u)
        beta = c.nrm2(u)
        c.axpby(1.0 / beta, u, 0.0, None, u)
        self.alphas.append(alpha)
        self.betas.append(beta)
        self.k += 1

    def B(self):
        k = self.k
        B = np.zeros((k + 1, k))
        for j in range(k):
            B[j, j], B[j + 1, j] = self.alphas[j], self.betas[j]
        return B


def tik(Bm, Lm, lam, rhs):                                                          # lstsq([B; sqrt(lam) L], [rhs; 0])
    M = np.vstack((Bm, np.sqrt(lam) * Lm))
    return np.linalg.lstsq(M, np.concatenate((rhs, np.zeros(Lm.shape[0]))), rcond=None)[0]


def hybrid_lsqr(c, A, b, iters, lam):                                               # Hybrid_LSQR.py:69-110, numeric regparam
    gk = GolubKahan(c, A, b, iters)
    x = c.empty(A.shape[1])
    for ii in range(iters):
        gk.step()
        if ii == 0:
            continue
        k = gk.k
        bhat = np.zeros(k + 1)
        bhat[0] = gk.beta0
        y = tik(gk.B(), np.eye(k), lam, bhat)
        c.combine(gk.V, k, y, x)
    return c.host(x)


def hybrid_gmres(c, A, b, iters, lam):                                              # Hybrid_GMRES.py:45-83 + decompositions.py:207-228
    n = A.shape[1]
    V, w, x = c.empty(iters + 1, n), c.empty(n), c.empty(n)
    beta0 = c.nrm2(b)
    c.axpby(1.0 / beta0, b, 0.0, None, V[0])
    H = np.zeros((iters + 1, iters))
    for k in range(1, iters + 1):
        A.apply(V[k - 1], w)
        h = np.zeros(k)
        for _ in range(2):                                   # classical Gram-Schmidt twice = the reference's modified sweep to rounding
            hh = c.project(V, k, w)
            c.combine(V, k, hh, w, a=1.0, base=w, s=-1.0)
            h += hh
        nw = c.nrm2(w)
        c.axpby(1.0 / nw, w, 0.0, None, V[k])
        H[:k, k - 1], H[k, k - 1] = h, nw
        if k == 1:
            continue                                          # the reference forms an iterate at its first step too, with lambda = 0;
        bhat = np.zeros(k + 1)                                # the last one is what is compared
        bhat[0] = beta0
        y = tik(H[:k + 1, :k], np.eye(k), lam, bhat)
        c.combine(V, k, y, x)
    return c.host(x)


def chol_r(G):
    return np.linalg.cholesky(G + 1e-14 * np.trace(G) * np.eye(G.shape[0])).T        # R with R^T R = G (the R of the economic QR)


def gks(c, A, L, b, d, iters, lam, weights=None, eps=0.1, q=1.0):                   # GKS.py:36-96; weights = "mm": MMGKS.py:43-128 (pnorm 2)
    m, n = A.shape
    p = L.shape[0]
    kmax = d + iters + 1
    gk = GolubKahan(c, A, b, d)
    for _ in range(d):
        gk.step()
    V, AV, LV = c.empty(kmax, n), c.empty(kmax, m), c.empty(kmax, p)
    V[:d].copy_(gk.V[:d])
    k = d
    for j in range(k):
        A.apply(V[j], AV[j])
        L.apply(V[j], LV[j])
    x, tm, tp, r, rb, wr, lx = c.empty(n), c.empty(m), c.empty(p), c.empty(n), c.empty(n), c.empty(p), c.empty(p)
    if weights:
        A.apply(b, x, True)                                                           # MMGKS.py:43
    for ii in range(iters):
        if weights:                                                                   # wf = 1 (pnorm = 2); wr from L x (:60,93)
            L.apply(x, lx)
            c.mm_weights(lx, None, eps, q, wr)
        GA, c1, _ = c.wgram(AV, k, None, b)
        GL, _, _ = c.wgram(LV, k, wr if weights else None)
        R_A, R_L = chol_r(GA), chol_r(GL)
        rhs = np.linalg.solve(R_A.T, c1)                                              # Q_A^T b
        y = tik(R_A, R_L, lam, rhs)
        c.combine(V, k, y, x)
        if weights and ii >= k:
            break
        c.combine(AV, k, y, tm, a=-1.0, base=b, s=1.0)                                # (AV) y - b
        A.apply(tm, r, True)
        c.combine(LV, k, y, tp)
        if weights:
            c.mul(wr, tp, tp)                                                         # to the FIRST power here (MMGKS.py:116), squared in the QR (:94-95)
        L.apply(tp, rb, True)
        c.axpby(1.0, r, lam, rb, r)
        for _ in range(2 if weights else 3):                                          # (:86-88 / MMGKS.py:119-120)
            c.combine(V, k, c.project(V, k, r), r, a=1.0, base=r, s=-1.0)
        c.axpby(1.0 / c.nrm2(r), r, 0.0, None, V[k])
        A.apply(V[k], AV[k])
        L.apply(V[k], LV[k])
        k += 1
    return c.host(x)


# ------------------------------------------------------------------------------------------------ against the oracle
@pytest.fixture(scope="module")
def setup():
    from oracle import cpu_ref as O
    c = Core()
    N = 48
    psf, _ = O.gauss_psf((7, 7), (1.6, 2.1))
    Ao, Lo = O.Blur2D(psf, N, N), O.FirstDerivative2D(N)
    ii, jj = np.meshgrid(np.arange(N), np.arange(N), indexing="ij")
    xt = ((np.abs(ii - 20) < 9) & (np.abs(jj - 26) < 12)).astype(np.float64) + 0.5 * (((ii - 34) ** 2 + (jj - 12) ** 2) < 40)
    rng = np.random.default_rng(7)
    b = Ao @ xt.reshape(-1)
    e = rng.standard_normal(b.size)
    b = (b + 0.01 * np.linalg.norm(b) / np.linalg.norm(e) * e).astype(np.float32).astype(np.float64)
    return c, c.blur(psf, N, N), c.deriv2d(N), Ao, Lo, c.vec(b), b, N


def rel(a, b):
    return float(np.linalg.norm(np.ravel(a) - np.ravel(b)) / np.linalg.norm(np.ravel(b)))


def test_core_symbols_are_all_there_and_are_twenty():
    lib = ctypes.CDLL(os.path.join(REPO, "trips_py_amd", "csrc", "libtrk.so"))
    assert len(CORE) == 20
    for name in CORE:
        assert hasattr(lib, name), name


def test_cgls_on_the_core(setup):
    from oracle import cpu_ref as O
    c, A, L, Ao, Lo, bd, b, N = setup
    x = cgls(c, A, bd, torch.zeros(N * N, device=c.dev), 25)
    xo, _ = O.cgls(Ao, b.reshape(-1, 1), np.zeros((N * N, 1)), 25, 0)
    assert rel(x, xo) < 1e-5


def test_hybrid_lsqr_on_the_core(setup):
    from oracle import cpu_ref as O
    c, A, L, Ao, Lo, bd, b, N = setup
    xo, _ = O.hybrid_lsqr(Ao, b.reshape(-1, 1), 20, 1e-2)
    assert rel(hybrid_lsqr(c, A, bd, 20, 1e-2), xo) < 1e-5


def test_hybrid_gmres_on_the_core(setup):
    from oracle import cpu_ref as O
    c, A, L, Ao, Lo, bd, b, N = setup
    xo, _ = O.hybrid_gmres(Ao, b.reshape(-1, 1), 15, 1e-2)
    assert rel(hybrid_gmres(c, A, bd, 15, 1e-2), xo) < 1e-5


def test_gks_on_the_core(setup):
    from oracle import cpu_ref as O
    c, A, L, Ao, Lo, bd, b, N = setup
    xo, _ = O.gks(Ao, b.reshape(-1, 1), Lo, 3, 10, 1e-2)
    assert rel(gks(c, A, L, bd, 3, 10, 1e-2), xo) < 1e-5


def test_mmgks_on_the_core(setup):
    from oracle import cpu_ref as O
    c, A, L, Ao, Lo, bd, b, N = setup
    xo, _ = O.mmgks(Ao, b.reshape(-1, 1), Lo, 2, 1, 3, 8, 1e-2, epsilon=0.1)
    assert rel(gks(c, A, L, bd, 3, 8, 1e-2, weights="mm", eps=0.1, q=1.0), xo) < 2e-5
