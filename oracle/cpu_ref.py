"""CPU ORACLE — a NumPy/SciPy float64 restatement of TRIPs-Py's Krylov hot path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT.  Only `tests/`, `__graft_entry__.smoke()`
and the `cpu_baseline` leg of `bench.py` may import it, and only as the checker /
the reported CPU baseline.  Nothing under `trips_py_amd/` imports it; the product
path runs on hand-written HIP kernels and fails loudly without them.

Pinning status
  * blur, CGLS, Golub-Kahan / Arnoldi, Hybrid-LSQR, Hybrid-GMRES, GKS, MMGKS, the
    derivative operators, the MM weights and the GCV / discrepancy / L-curve
    selectors are PINNED: tests/test_oracle_golden.py checks every function here
    against tests/golden/*.npz, which tools/make_goldens.py produced by running the
    reference itself (/root/reference, 2025-08-24 snapshot) in the build container.
  * The blur arithmetic of the reference lives in third-party SciPy
    (`scipy.ndimage.convolve`, un-pinned in the reference's setup.py:5; 1.15.3 here).
    `blur2d_scipy` calls it with the reference's exact arguments; `blur2d_numpy`
    restates its published algorithm (correlate with the mirrored kernel, origin
    shifted for even sizes, half-sample-symmetric 'reflect' extension) from scratch,
    and the two are checked against each other and against the goldens.
  * Radon (parallel beam): the arithmetic lives in astra-toolbox (un-pinned, setup.py:11;
    not installable here).  `Radon2D` follows the call-site contract of
    trips/utilities/io.py:392-399 with a Joseph (linear-interpolation) projector.
    PINNED: its CONVENTION — rotation sense, detector order, (views, detectors) layout,
    row-major image — to the two ASTRA outputs the reference holds as rendered images
    (demos/demo_Tomo_small_scale.ipynb:145,179, decoded into
    tests/golden/fanbeam_demo_image.npz), through the fan-beam operator whose far-source
    limit it is (tests/test_oracle_golden.py: correlation 0.9999 with the ASTRA sinogram,
    every mirrored convention <= 0.95; parallel vs far-source fan: 1e-3, the two
    interpolation models); the adjoint identity, analytic line integrals and the
    axis-aligned views.  NOT pinned: the interpolation weights against ASTRA's own
    numbers (no parallel-beam output exists in the reference) — they follow Joseph's
    published kernel.
  * MMGKS isoTV weights (MMGKS.py:61-77): PARITY UNPINNED in the one piece that lives in
    PyLops (un-pinned, setup.py:6; absent): `pylops.FirstDerivative`'s centered stencil is
    restated from its published definition.  Everything around it (reshape, exponent,
    weight layout, solver loop) is checked against goldens made by running the reference's
    own MMGKS.py / operators_old.py over that same restatement (tools/oracle_shim).

Every function cites the reference lines it follows (paths relative to /root/reference).
"""
import numpy as np
import scipy.linalg as sla
import scipy.optimize as sopt
import scipy.sparse as sp

try:  # the reference's own blur kernel (third-party)
    from scipy.ndimage import convolve as _ndi_convolve
except Exception:  # pragma: no cover
    _ndi_convolve = None


# =====================================================================================
# small operator protocol (what the reference's solvers require of `A`: SURVEY §8b)
# =====================================================================================
class _Op:
    """Duck-typed operator: shape, `@`, `*`, `.T`; operands (n,), (n,1) or (n,k) keep their rank."""

    shape = (0, 0)

    def _fwd(self, x):  # x: (n,) -> (m,)
        raise NotImplementedError

    def _adj(self, y):  # y: (m,) -> (n,)
        raise NotImplementedError

    def _apply(self, fn, x, nin):
        x = np.asarray(x, dtype=np.float64)
        if x.ndim == 1:
            assert x.shape[0] == nin
            return fn(x)
        assert x.ndim == 2 and x.shape[0] == nin
        return np.stack([fn(x[:, j]) for j in range(x.shape[1])], axis=1)

    def matvec(self, x):
        return self._apply(self._fwd, x, self.shape[1])

    def rmatvec(self, y):
        return self._apply(self._adj, y, self.shape[0])

    def __matmul__(self, x):
        return self.matvec(x)

    __mul__ = __matmul__

    @property
    def T(self):
        return _AdjOp(self)

    def todense(self):
        return self.matvec(np.eye(self.shape[1]))


class _AdjOp(_Op):
    def __init__(self, op):
        self.op = op
        self.shape = (op.shape[1], op.shape[0])

    def _fwd(self, x):
        return self.op._adj(x)

    def _adj(self, y):
        return self.op._fwd(y)

    @property
    def T(self):
        return self.op


class MatrixOp(_Op):
    """Wrap an ndarray / scipy.sparse matrix in the same protocol."""

    def __init__(self, M):
        self.M = M
        self.shape = M.shape

    def _fwd(self, x):
        return np.asarray(self.M @ x).reshape(-1)

    def _adj(self, y):
        return np.asarray(self.M.T @ y).reshape(-1)


# =====================================================================================
# a1  Gaussian PSF                                  trips/test_problems/Deblurring2D.py:48-64
# =====================================================================================
def gauss_psf(dim, spread):
    m, n = int(dim[0]), int(dim[1])
    s1, s2 = (spread, spread) if np.isscalar(spread) else (spread[0], spread[1])
    gx = np.arange(-np.fix(n / 2), np.ceil(n / 2))      # along columns, uses s1 (:57,60)
    gy = np.arange(-np.fix(m / 2), np.ceil(m / 2))      # along rows, uses s2 (:58,60)
    X, Y = np.meshgrid(gx, gy)
    psf = np.exp(-0.5 * (X ** 2 / s1 ** 2 + Y ** 2 / s2 ** 2))
    psf /= psf.sum()
    r, c = np.where(psf == psf.max())
    return psf, np.array([r[0], c[0]], dtype=int)


# =====================================================================================
# a2  2-D blur with 'reflect' boundary                trips/test_problems/Deblurring2D.py:66-73
# =====================================================================================
def reflect_index(i, n):
    """Half-sample-symmetric extension (d c b a | a b c d | d c b a), any distance."""
    i = np.mod(i, 2 * n)
    return np.where(i >= n, 2 * n - 1 - i, i)


def blur2d_numpy(img, psf):
    """y[i,j] = sum_{a,b} psf[a,b] * xr[i + kh//2 - a, j + kw//2 - b],  xr = reflect extension.

    This is what scipy.ndimage.convolve(img, psf, mode='reflect') computes (convolution
    = correlation with the mirrored kernel; for even sizes the mirrored kernel's origin
    moves by one, which the formula above absorbs)."""
    img = np.asarray(img, dtype=np.float64)
    psf = np.asarray(psf, dtype=np.float64)
    nx, ny = img.shape
    kh, kw = psf.shape
    out = np.zeros_like(img)
    rows = np.arange(nx)
    cols = np.arange(ny)
    for a in range(kh):
        ri = reflect_index(rows + kh // 2 - a, nx)
        for b in range(kw):
            ci = reflect_index(cols + kw // 2 - b, ny)
            out += psf[a, b] * img[np.ix_(ri, ci)]
    return out


def blur2d_scipy(img, psf):
    """The reference's literal call (Deblurring2D.py:70)."""
    return _ndi_convolve(np.asarray(img, dtype=np.float64), np.asarray(psf, dtype=np.float64), mode="reflect")


class Blur2D(_Op):
    """fwd = convolve(X.reshape(nx,ny), PSF, 'reflect'); 'transpose' := the same with
    flipud(fliplr(PSF))  (Deblurring2D.py:70-71) — exact adjoint only for odd, symmetric PSFs."""

    def __init__(self, psf, nx, ny, use_scipy=True):
        self.psf = np.asarray(psf, dtype=np.float64)
        self.psf_flip = self.psf[::-1, ::-1].copy()
        self.nx, self.ny = int(nx), int(ny)
        self.shape = (self.nx * self.ny, self.nx * self.ny)
        self._conv = blur2d_scipy if (use_scipy and _ndi_convolve is not None) else blur2d_numpy

    def _fwd(self, x):
        return self._conv(x.reshape(self.nx, self.ny), self.psf).reshape(-1)

    def _adj(self, y):
        return self._conv(y.reshape(self.nx, self.ny), self.psf_flip).reshape(-1)


class Blur1D(_Op):
    """1-D blur with a length-n PSF (Deblurring1D.py:56-62,93-102): column vector nx x 1."""

    def __init__(self, psf, use_scipy=True):
        n = len(psf)
        self._b = Blur2D(np.asarray(psf, dtype=np.float64).reshape(n, 1), n, 1, use_scipy)
        self.shape = (n, n)

    def _fwd(self, x):
        return self._b._fwd(x)

    def _adj(self, y):
        return self._b._adj(y)


def gauss_psf_1d(n, sigma):
    """Deblurring1D.py:63-69."""
    x = np.arange(-np.fix(n / 2), np.ceil(n / 2))
    p = np.exp(-0.5 * (x ** 2) / sigma ** 2)
    return p / p.sum()


# =====================================================================================
# a3  parallel-beam Radon, Joseph / 'linear'              trips/utilities/io.py:392-400
#     convention pinned to the reference's ASTRA images, weights = Joseph's published kernel (see module docstring)
# =====================================================================================
class Radon2D(_Op):
    """Sinogram (n_ang, n_det) row-major <- image (N, N) row-major, times `scale` (=1/N at io.py:397).

    Geometry (recorded convention; any self-consistent one satisfies the pinned tests):
      pixel (i,j) centre at (x, y) = (j - (N-1)/2, (N-1)/2 - i); detector bin d at
      s = d - (n_det-1)/2 along (cos t, sin t); rays run along (sin t, -cos t).
      |cos t| >= |sin t|: march rows, column coordinate c = (s - y_i sin t)/cos t + (N-1)/2,
      taps floor(c), floor(c)+1 with weights (1-f), f, each times 1/|cos t|;
      otherwise march columns with row coordinate r = (N-1)/2 - (s - x_j cos t)/sin t and 1/|sin t|.
    The matrix is assembled explicitly (scipy.sparse), so `.T` is the exact matched adjoint."""

    def __init__(self, N, angles, n_det=None, scale=None):
        self.N = int(N)
        self.angles = np.asarray(angles, dtype=np.float64).reshape(-1)
        self.n_det = int(n_det) if n_det is not None else self.N
        self.scale = (1.0 / self.N) if scale is None else float(scale)
        self.shape = (len(self.angles) * self.n_det, self.N * self.N)
        self._M = None

    def matrix(self):
        """CSR, written row by row: sinogram row (a, d) holds its 2 N taps in marching order (a tap outside the image is kept as
        an explicit zero on column 0, so every row has the same length and nothing has to be sorted or compacted — assembling
        512^2 x 180 takes seconds instead of the half minute of a COO build)."""
        if self._M is None:
            N, nd = self.N, self.n_det
            half = (N - 1) / 2.0
            s = np.arange(nd) - (nd - 1) / 2.0
            k = np.arange(N)
            na = len(self.angles)
            idx = np.empty((na, nd, N, 2), dtype=np.int32 if N * N < 2 ** 31 else np.int64)
            val = np.empty((na, nd, N, 2), dtype=np.float64)
            for a, th in enumerate(self.angles):
                ct, st = np.cos(th), np.sin(th)
                if abs(ct) >= abs(st):
                    y = half - k                                        # row i -> y
                    q = (s[:, None] - y[None, :] * st) / ct + half      # (nd, N rows): column coordinate
                    w = 1.0 / abs(ct)
                else:
                    x = k - half                                        # col j -> x
                    q = half - (s[:, None] - x[None, :] * ct) / st      # (nd, N cols): row coordinate
                    w = 1.0 / abs(st)
                q0 = np.floor(q)
                f = q - q0
                for off, wt in ((0, 1.0 - f), (1, f)):
                    t = (q0 + off).astype(np.int64)
                    ok = (t >= 0) & (t < N)
                    t = np.where(ok, t, 0)
                    lin = (k[None, :] * N + t) if abs(ct) >= abs(st) else (t * N + k[None, :])
                    idx[a, :, :, off] = np.where(ok, lin, 0)
                    val[a, :, :, off] = np.where(ok, w * wt, 0.0)
            val *= self.scale
            indptr = np.arange(na * nd + 1, dtype=np.int64) * (2 * N)
            self._M = sp.csr_matrix((val.reshape(-1), idx.reshape(-1), indptr), shape=self.shape)
        return self._M

    def _fwd(self, x):
        return self.matrix() @ x

    def _adj(self, y):
        return self.matrix().T @ y


# =====================================================================================
# a4  frame-block (block-diagonal) operator     trips/utilities/io.py:223-225,420 (pylops.BlockDiag)
# =====================================================================================
class BlockDiag(_Op):
    def __init__(self, ops):
        self.ops = list(ops)
        self._ro = np.cumsum([0] + [o.shape[0] for o in self.ops])
        self._co = np.cumsum([0] + [o.shape[1] for o in self.ops])
        self.shape = (int(self._ro[-1]), int(self._co[-1]))

    def _fwd(self, x):
        return np.concatenate([o._fwd(x[self._co[t]:self._co[t + 1]]) for t, o in enumerate(self.ops)])

    def _adj(self, y):
        return np.concatenate([o._adj(y[self._ro[t]:self._ro[t + 1]]) for t, o in enumerate(self.ops)])


def dynamic_frame_blocks(A, b, nt, rows_per_frame, cols_per_frame):
    """The per-frame blocks the reference's loaders cut out of a dynamic problem's sparse forward matrix and data
    (io.py:223-225 generate_crossPhantom, :160-162 generate_emoji): AA[ii] = A[r ii : r (ii+1), c ii : c (ii+1)], B[ii] = b[r ii : r (ii+1)]."""
    A = sp.csr_matrix(A)
    b = np.asarray(b).reshape(-1)
    r, c = int(rows_per_frame), int(cols_per_frame)
    AA = [A[r * ii:r * (ii + 1), c * ii:c * (ii + 1)] for ii in range(int(nt))]
    B = [b[r * ii:r * (ii + 1)] for ii in range(int(nt))]
    return AA, B


# =====================================================================================
# a14 derivative regularisers (as sparse matrices)         trips/utilities/operators.py:24-45
# =====================================================================================
def first_derivative_1d(n):
    """(n-1) x n, row i = x[i] - x[i+1]   (operators.py:24-28)."""
    return (sp.identity(n, format="csr") - sp.diags(np.ones(n - 1), 1, format="csr"))[:-1, :]


def first_derivative_2d(nx, ny):
    """vstack(kron(I_nx, D_nx), kron(D_ny, I_ny))   (operators.py:30-36; meaningful for nx == ny)."""
    return sp.vstack((sp.kron(sp.identity(nx), first_derivative_1d(nx)),
                      sp.kron(first_derivative_1d(ny), sp.identity(ny)))).tocsr()


def spacetime_derivative(nx, ny, nt):
    """vstack(kron(I_nt, D2), kron(D_nt, I_{nx^2}))   (operators.py:39-45)."""
    return sp.vstack((sp.kron(sp.identity(nt), first_derivative_2d(nx, ny)),
                      sp.kron(first_derivative_1d(nt), sp.identity(nx ** 2)))).tocsr()


class FirstDerivative2D(_Op):
    """Matrix-free form of first_derivative_2d for an N x N image (row-major):
    first N(N-1) rows x[i,j]-x[i,j+1] (i-major), then (N-1)N rows x[i,j]-x[i+1,j]."""

    def __init__(self, N):
        self.N = int(N)
        self.shape = (2 * self.N * (self.N - 1), self.N * self.N)

    def _fwd(self, x):
        X = x.reshape(self.N, self.N)
        return np.concatenate(((X[:, :-1] - X[:, 1:]).reshape(-1), (X[:-1, :] - X[1:, :]).reshape(-1)))

    def _adj(self, y):
        N = self.N
        H = y[:N * (N - 1)].reshape(N, N - 1)
        Vv = y[N * (N - 1):].reshape(N - 1, N)
        out = np.zeros((N, N))
        out[:, :-1] += H
        out[:, 1:] -= H
        out[:-1, :] += Vv
        out[1:, :] -= Vv
        return out.reshape(-1)


class SpaceTimeDerivative(_Op):
    """Matrix-free form of spacetime_derivative for nt frame-major N x N frames."""

    def __init__(self, N, nt):
        self.N, self.nt = int(N), int(nt)
        self.D2 = FirstDerivative2D(N)
        self.ps = self.D2.shape[0]
        self.shape = (self.nt * self.ps + (self.nt - 1) * self.N * self.N, self.nt * self.N * self.N)

    def _fwd(self, x):
        X = x.reshape(self.nt, -1)
        spat = np.concatenate([self.D2._fwd(X[t]) for t in range(self.nt)])
        return np.concatenate((spat, (X[:-1] - X[1:]).reshape(-1)))

    def _adj(self, y):
        n2 = self.N * self.N
        out = np.stack([self.D2._adj(y[t * self.ps:(t + 1) * self.ps]) for t in range(self.nt)])
        Tm = y[self.nt * self.ps:].reshape(self.nt - 1, n2)
        out[:-1] += Tm
        out[1:] -= Tm
        return out.reshape(-1)


def framelet_filters_1d(n, lev):
    """The three n x n filter matrices of one framelet level (operators.py:50-85): taps (1, 2, 1)/4, (-1, 0, 1)*sqrt(2)/4 and
    (-1, 2, -1)/4 at offsets (-lev, 0, +lev); the corrections of :57-59, :67-69, :79-81 are the half-sample-symmetric
    reflection of the taps that fall off either end (index -1-j -> j, n+j -> n-1-j)."""
    taps = (((1.0, 2.0, 1.0), 0.25), ((-1.0, 0.0, 1.0), np.sqrt(2.0) / 4.0), ((-1.0, 2.0, -1.0), 0.25))
    out = []
    for coef, scale in taps:
        H = np.zeros((n, n))
        for i in range(n):
            for off, c in zip((-lev, 0, lev), coef):
                H[i, reflect_index(i + off, n)] += c
        out.append(H * scale)
    return out


def framelet_analysis_1d(n, l):
    """operators.py:88-103, recursion included as written: the deepest level returns its three filters WITHOUT the product of
    the low-pass filters above it (:90-91 returns before `* w`); every other level is vstack(deeper, H1, H2) * w with w the
    low-pass filter of the level above (w = 1 at the top)."""
    def rec(level, w):
        H0, H1, H2 = framelet_filters_1d(n, level)
        if level == l:
            return np.vstack((H0, H1, H2))
        return np.vstack((rec(level + 1, H0), H1, H2)) @ w
    return rec(1, np.eye(n))


class Framelet2D(_Op):
    """create_framelet_operator(n, m, l) (operators.py:105-113): x -> W_n X W_m^H with COLUMN-major reshapes of x (n x m)
    and of the result (n(2l+1) x m(2l+1))."""

    def __init__(self, n, m, l):
        self.n, self.m = int(n), int(m)
        self.Wn, self.Wm = framelet_analysis_1d(self.n, l), framelet_analysis_1d(self.m, l)
        self.shape = (self.Wn.shape[0] * self.Wm.shape[0], self.n * self.m)

    def _fwd(self, x):
        return (self.Wn @ (x.reshape(self.n, self.m, order="F") @ self.Wm.T)).reshape(-1, order="F")

    def _adj(self, y):
        return (self.Wn.T @ (y.reshape(self.Wn.shape[0], self.Wm.shape[0], order="F") @ self.Wm)).reshape(-1, order="F")


# =====================================================================================
# a13 MM weights                                           trips/utilities/weights.py:66-68
# =====================================================================================
def smoothed_holder_weights(x, epsilon, p):
    return (np.asarray(x, dtype=np.float64) ** 2 + epsilon ** 2) ** (p / 2 - 1)


# =====================================================================================
# a6/a8/a10  Krylov factorisations                 trips/utilities/decompositions.py:20-255
# =====================================================================================
def _col(v):
    return np.asarray(v, dtype=np.float64).reshape(-1)


def golub_kahan_update(A, U, B, V):
    """One GK step, no reorthogonalisation (decompositions.py:230-255).
    First call: U = b/||b|| (m x 1), B = None (or shape-(1,) placeholder), V = None."""
    first = B is None or np.ndim(B) < 2
    u_last = U[:, -1]
    v = _col(A.T @ u_last)
    if not first:
        k = B.shape[0]
        v = v - B[k - 1, k - 2] * V[:, k - 2]
    alpha = np.linalg.norm(v)
    v = v / alpha
    u = _col(A @ v) - alpha * u_last
    beta = np.linalg.norm(u)
    u = u / beta
    U = np.hstack((U, u[:, None]))
    if first:
        return U, np.array([[alpha], [beta]]), v[:, None]
    k = B.shape[0]
    Bn = np.zeros((k + 1, k))
    Bn[:k, :k - 1] = B
    Bn[k - 1, k - 1] = alpha
    Bn[k, k - 1] = beta
    return U, Bn, np.hstack((V, v[:, None]))


def golub_kahan(A, b, n_iter):
    """n_iter GK steps from b (decompositions.py:118-205, dp_stop=False path): U m x (d+1), S (d+1) x d, V n x d."""
    b = _col(b)
    U = (b / np.linalg.norm(b))[:, None]
    B = V = None
    for _ in range(n_iter):
        U, B, V = golub_kahan_update(A, U, B, V)
    return U, B, V


def arnoldi_update(A, V, H):
    """One Arnoldi step with modified Gram-Schmidt against ALL previous vectors (decompositions.py:207-228).
    First call: V = b/||b|| (n x 1), H = None."""
    k = V.shape[1]
    w = _col(A @ V[:, -1])
    h = np.zeros(k + 1)
    for j in range(k):
        h[j] = np.dot(V[:, j], w)
        w = w - h[j] * V[:, j]
    h[k] = np.linalg.norm(w)
    Hn = np.zeros((k + 1, k))
    if H is not None and np.ndim(H) == 2:
        Hn[:k, :k - 1] = H
    Hn[:, k - 1] = h
    return np.hstack((V, (w / h[k])[:, None])), Hn


def arnoldi(A, b, n_iter, dp_stop=False, gk_eta=1.001, gk_delta=0.001):
    """decompositions.py:20-116.  NOTE the reference orthogonalises step ii only against the first `ii` vectors... which
    is all of them but the newest (jj < iterations == ii): H[ii,ii] is never written and the new vector is not
    orthogonalised against Q[:,ii].  dp_stop (:104-112): after every step the NORMALISED b is projected, y solves
    (H_k^T H_k) y = Q_k^T b with the square top block H_k of H, and the factorisation halts before the next step once
    ||A Q_k y - b/||b|| || <= gk_eta * gk_delta (defaults 1.001, 0.001; not the solvers' delta)."""
    b = _col(b)
    n = b.shape[0]
    bn = b / np.linalg.norm(b)
    Q = np.zeros((n, 2))
    H = np.zeros((2, 1))
    Q[:, 0] = bn
    res_norm = np.inf
    for ii in range(n_iter):
        if dp_stop and res_norm <= gk_eta * gk_delta:
            break
        if ii != 0:
            Q = np.pad(Q, ((0, 0), (0, 1)))
            H = np.pad(H, ((0, 1), (0, 1)))
        w = _col(A @ Q[:, ii])
        for jj in range(ii):
            H[jj, ii] = np.dot(Q[:, jj], w)
            w = w - H[jj, ii] * Q[:, jj]
        H[ii + 1, ii] = np.linalg.norm(w)
        if H[ii + 1, ii] == 0:
            return Q, H
        Q[:, ii + 1] = w / H[ii + 1, ii]
        if dp_stop:
            bhat = Q[:, :-1].T @ bn
            Hk = H[:-1, :]
            y = np.linalg.lstsq(Hk.T @ Hk, bhat, rcond=None)[0]
            res_norm = np.linalg.norm(_col(A @ (Q[:, :-1] @ y)) - bn)
    return Q, H


# =====================================================================================
# a16  regularisation-parameter selectors (k-sized problems; host fp64)
# =====================================================================================
def _dense(M):
    return M.todense() if hasattr(M, "todense") and not isinstance(M, np.ndarray) else np.asarray(M)


def gcv_numerator(lam, Q_A, R_A, R_L, b, variant="standard"):
    """reg_param/gcv.py:25-48."""
    rhs = Q_A.T @ b
    xl = sla.solve(R_A.T @ R_A + lam * (R_L.T @ R_L), R_A.T @ rhs)
    val = np.linalg.norm(R_A @ xl - rhs) ** 2
    if variant == "modified":
        val = val + np.linalg.norm(b - Q_A @ rhs) ** 2
    return val


def gcv_denominator(lam, R_A, R_L, variant="standard", fullsize=None):
    """reg_param/gcv.py:50-78."""
    inv = sla.solve(R_A.T @ R_A + lam * (R_L.T @ R_L), R_A.T)
    tr = np.trace(R_A @ inv)
    return ((fullsize if variant == "modified" else R_A.shape[0]) - tr) ** 2


def gcv_choose(Q_A, R_A, R_L, b, variant="standard", fullsize=None):
    """reg_param/gcv.py:80-95, gcvtype='tikhonov'.  The reference forwards **kwargs only to the
    DENOMINATOR (:94): the numerator is always the 'standard' one."""
    fun = lambda lam: gcv_numerator(lam, Q_A, R_A, R_L, b) / gcv_denominator(lam, R_A, R_L, variant, fullsize)
    return sopt.fminbound(fun, 1e-9, 1e2, xtol=1e-12, maxfun=1000, disp=0)


def discrepancy_choose(Q, A, L, b, delta, eta=1.01, L_is_identity=False, explicitProj=False):
    """reg_param/discrepancy_principle.py:19-99, dptype='tikhonov', for the shapes the solvers
    produce: (i) hybrid: Q=U m x (k+1), A=B (k+1) x k, L=I;  (ii) GKS/MMGKS: Q=Q_A, A=R_A, L=R_L (square,
    nonsingular).  Returns alpha (=lambda); 0 when the discrepancy is not yet reachable."""
    bfull = b
    bp = Q.T @ b
    bp0 = bp                                                             # the reference's `b` (= Q^T b) where it forms b - Q Q^T b
    if L_is_identity:
        Anew = A
    else:
        _, SL, VL = sla.svd(L)
        if L.shape[0] >= L.shape[1] and SL[-1] != 0:                     # (:42-44)
            Anew = A @ (VL.T @ np.diag(SL ** (-1.0)))
        else:                                                            # (:45-66) null space of L: exact zeros / fewer rows
            W = (VL[np.where(SL == 0), :].reshape((-1, 1)) if L.shape[0] >= L.shape[1] else VL[L.shape[0] - L.shape[1]:, :].T)
            Q_AW, R_AW = np.linalg.qr(A @ W, mode="reduced")
            Q_LT, R_LT = np.linalg.qr(L.T, mode="reduced")
            P = (np.eye(L.shape[1]) - (W @ np.linalg.inv(R_AW) @ Q_AW.T @ A)) @ Q_LT @ np.linalg.inv(R_LT.T)
            Anew = A @ P
            bp = bp - A @ (W @ np.linalg.inv(R_AW) @ Q_AW.T @ bp)
    U, S, _ = sla.svd(Anew)
    sv = S ** 2
    bhat = (U.T @ bp).reshape(-1, 1)
    r, c = Anew.shape
    if r > c:
        sv = np.append(sv, np.zeros(r - c))
        testzero = np.linalg.norm(bhat[c - r:, :]) ** 2 - (eta * delta) ** 2
        if explicitProj:
            testzero += np.linalg.norm(bfull - Q @ bp0) ** 2
    else:
        testzero = np.linalg.norm(bfull - Q @ bp0) ** 2 - (eta * delta) ** 2
    sv = sv.reshape(-1, 1)
    if not testzero < 0:
        return 0
    beta, it, alpha = 1e-8, 0, None
    extra = np.linalg.norm(bfull - Q @ bp0) ** 2 if explicitProj else 0.0
    while it < 30 or (it <= 100 and abs(alpha) < 1e-16):
        z = bhat / (sv * beta + 1)
        f = np.linalg.norm(z) ** 2 + extra - (eta * delta) ** 2
        w = z / (sv * beta + 1)
        fp = 2 / beta * (z.T @ (w - z))
        beta_new = beta - f / fp
        if abs(beta_new - beta) < 1e-12 * beta:
            break
        beta = beta_new
        alpha = 1 / beta_new[0, 0]
        it += 1
    return alpha


def discrepancy_truncation(Qtb, n, delta, eta=1.01, dptype="tsvd"):
    """reg_param/discrepancy_principle.py:100-129 (dptype 'tsvd' / 'tgsvd'): truncation index from bhat = Q^T b, n = L.shape[1]."""
    bhat = np.asarray(Qtb, dtype=np.float64).reshape(-1, 1)
    m, alpha = bhat.shape[0], n
    if dptype == "tsvd":
        f = np.ones((m, 1))
        for i in range(n):
            f[n - (i + 1), ] = 0
            fvar = np.concatenate((1 - f[:n, ], f[n:, ]))
            if np.sum((fvar * bhat) ** 2) - (eta * delta) ** 2 < 0:
                alpha = n - (i + 1)
            else:
                break
        return alpha
    coeff = np.square(bhat)
    for i in range(n):
        coeff[n - (i + 1), ] = 0
        if np.sum(coeff) - (eta * delta) ** 2 >= 0:
            alpha = i
        else:
            break
    return alpha


def _lc_terms(lam, A, L, b):
    """x_l, x_l', x_l'' of the Tikhonov solution (reg_param/l_curve.py:23-87, d = 0)."""
    C, D = A.T @ A, L.T @ L
    M = C + lam * D
    x = np.linalg.lstsq(M, A.T @ b, rcond=None)[0]
    dx = -np.linalg.lstsq(M, D @ x, rcond=None)[0]
    i4 = np.linalg.lstsq(M, D @ x, rcond=None)[0]
    d2x = 2 * np.linalg.lstsq(M, D @ dx - D @ i4, rcond=None)[0]
    return x, dx, d2x


def lcurve_curvature(lam, A, L, b):
    """reg_param/l_curve.py:171-189 (fidelity f = ||Ax-b||^2, regulariser g = ||Lx||^2)."""
    x, dx, d2x = _lc_terms(lam, A, L, b)
    fr, gr = A @ x - b, L @ x
    f1 = (2 * fr.T @ (A @ dx)).item()
    g1 = (2 * gr.T @ (L @ dx)).item()
    f2 = (2 * ((A @ dx).T @ (A @ dx) + fr.T @ (A @ d2x))).item()
    g2 = (2 * ((L @ dx).T @ (L @ dx) + gr.T @ (L @ d2x))).item()
    return (-g1 * f2 + f1 * g2) / (g1 ** 2 + f1 ** 2) ** 1.5


def lcurve_choose(A, L, b):
    """reg_param/l_curve.py:190-203."""
    return sopt.fminbound(lambda l: -lcurve_curvature(l, A, L, b), 1e-9, 2, xtol=1e-12, maxfun=1000, disp=0)


# =====================================================================================
# a5  CGLS                                                       trips/solvers/CGLS.py:16-86
# =====================================================================================
def cgls(A, b, x0, max_iter, tol, x_true=None):
    b = np.asarray(b, dtype=np.float64).reshape(-1, 1)
    x = np.asarray(x0, dtype=np.float64).reshape(-1, 1)
    r = b - A @ x
    t = A.T @ r
    p = t
    nt0 = np.linalg.norm(t)
    gamma = nt0 ** 2
    hist, relres, relerr = [], [], []
    k, stop = 0, False
    while k < max_iter and not stop:
        x_old = x
        k += 1
        w = A @ p
        delta = np.linalg.norm(w) ** 2
        step = gamma / delta                       # the reference calls this `beta` (:64)
        x = x + step * p
        hist.append(x)
        r = r - step * w
        t = A.T @ r
        gamma_old, nt = gamma, np.linalg.norm(t)
        gamma = nt ** 2
        p = t + (gamma / gamma_old) * p
        nx = np.linalg.norm(x)
        stop = (nt <= nt0 * tol) or (nx * tol >= 1)
        relres.append(np.linalg.norm(x - x_old) / nx)
        if x_true is not None:
            relerr.append(np.linalg.norm(x - np.asarray(x_true).reshape(-1, 1)) / nx)   # sic: / ||x|| (:79)
    info = {"xHistory": hist, "regParam": [], "relResidual": relres, "its": k}
    if x_true is not None:
        info["relError"] = relerr
    return x, info


# =====================================================================================
# helpers shared by the projection solvers
# =====================================================================================
def _tik_lstsq(M, L, lam, rhs):
    """y = argmin ||M y - rhs||^2 + lam ||L y||^2 via the stacked least-squares problem
    (Hybrid_LSQR.py:104, GKS.py:74)."""
    top = np.asarray(rhs, dtype=np.float64).reshape(-1, 1)
    return np.linalg.lstsq(np.vstack((M, np.sqrt(lam) * L)), np.vstack((top, np.zeros((L.shape[0], 1)))), rcond=None)[0]


def _rre(hist, x_true):
    xt = np.asarray(x_true, dtype=np.float64).reshape(-1, 1)
    return [np.linalg.norm(x - xt) / np.linalg.norm(xt) for x in hist]


# =====================================================================================
# a7  Hybrid LSQR                                         trips/solvers/Hybrid_LSQR.py:25-114
# =====================================================================================
def hybrid_lsqr(A, b, n_iter=100, regparam="gcv", x_true=None, delta=None, eta=1.01):
    if regparam == "dp" and delta is None:
        raise Exception("A value for the noise level delta was not provided")
    b = np.asarray(b, dtype=np.float64).reshape(-1, 1)
    beta = np.linalg.norm(b)
    U, B, V = b / beta, None, None
    bhat = np.array([beta])
    hist, lams = [], []
    lam = 0
    x = None
    for ii in range(n_iter):
        U, B, V = golub_kahan_update(A, U, B, V)
        bhat = np.append(bhat, 0)
        k = B.shape[1]
        if ii == 0:
            lam = 0
            continue                                              # no x at the first step (:77-78)
        if regparam == "gcv":
            Qb, s, _ = sla.svd(B, full_matrices=False)
            lam = gcv_choose(Qb, np.diag(s), np.eye(k), bhat, variant="modified", fullsize=A.shape[0])
        elif regparam == "dp":
            lam = discrepancy_choose(U, B, np.eye(k), b, delta, eta, L_is_identity=True)
        elif regparam == "l_curve":
            Qb, s, _ = sla.svd(B, full_matrices=False)
            lam = lcurve_choose(np.diag(s), np.eye(k), Qb.T @ bhat.reshape(-1, 1))
        else:
            lam = regparam
        lams.append(lam)
        y = _tik_lstsq(B, np.eye(k), lam, bhat)
        x = (V @ y).reshape(-1, 1)
        hist.append(x)
    info = {"xHistory": hist, "regParam": lam, "regParam_history": lams, "relResidual": [], "its": n_iter - 1}
    if x_true is not None:
        info["relError"] = _rre(hist, x_true)
    return x, info


# =====================================================================================
# a9  Hybrid GMRES                                       trips/solvers/Hybrid_GMRES.py:23-87
# =====================================================================================
def hybrid_gmres(A, b, n_iter, regparam="gcv", x_true=None, delta=None, eta=1.01):
    if regparam == "dp" and delta is None:
        raise Exception("A value for the noise level delta was not provided")
    if A.shape[0] != A.shape[1]:
        raise Exception("A should be square in order to apply hybrid GMRES")
    b = np.asarray(b, dtype=np.float64).reshape(-1, 1)
    beta = np.linalg.norm(b)
    V, H = b / beta, None
    bhat = np.array([beta])
    hist, lams, res = [], [], []
    lam = 0
    x = None
    for ii in range(n_iter):
        V, H = arnoldi_update(A, V, H)
        bhat = np.append(bhat, 0)
        k = H.shape[1]
        if ii == 0:
            lam = 0
        elif regparam == "gcv":
            Qh, s, _ = sla.svd(H, full_matrices=False)
            lam = gcv_choose(Qh, np.diag(s), np.eye(k), bhat)     # 'standard' variant here (:58)
        elif regparam == "dp":
            lam = discrepancy_choose(V, H, np.eye(k), b, delta, eta, L_is_identity=True)
        elif regparam == "l_curve":
            Qh, s, _ = sla.svd(H, full_matrices=False)
            lam = lcurve_choose(np.diag(s), np.eye(k), Qh.T @ bhat.reshape(-1, 1))
        else:
            lam = regparam
        lams.append(lam)
        y = _tik_lstsq(H, np.eye(k), lam, bhat)
        x = (V[:, :-1] @ y).reshape(-1, 1)
        hist.append(x)
        # reference quirk (:80): bhat is 1-D and H@y is a column, so `bhat - H@y` broadcasts to a
        # (k+1) x (k+1) matrix and la.norm is its Frobenius norm — reproduced, it is what info holds
        res.append(np.linalg.norm(bhat.reshape(1, -1) - (H @ y).reshape(-1, 1)))
    info = {"xHistory": hist, "regParam": lam, "regParam_history": lams, "relResidual": res, "its": n_iter - 1}
    if x_true is not None:
        info["relError"] = _rre(hist, x_true)
    return x, info


# =====================================================================================
# a11 GKS                                                          trips/solvers/GKS.py:27-105
# =====================================================================================
def _select_lambda(regparam, Q_A, R_A, R_L, b, delta, eta):
    if regparam == "gcv":
        return gcv_choose(Q_A, R_A, R_L, b)
    if regparam == "dp":
        return discrepancy_choose(Q_A, R_A, R_L, b, delta, eta)
    if regparam == "l_curve":
        return lcurve_choose(R_A, R_L, Q_A.T @ b)
    return regparam


def gks(A, b, L, projection_dim=3, n_iter=50, regparam="gcv", x_true=None, delta=None, eta=1.01):
    """General-L branch (QR of AV and LV from scratch every iteration, :54-56)."""
    if regparam == "dp" and delta is None:
        raise Exception("A value for the noise level delta was not provided")
    b = np.asarray(b, dtype=np.float64).reshape(-1, 1)
    _, _, V = golub_kahan(A, b, projection_dim)
    AV, LV = A @ V, L @ V
    hist, lams, res = [], [], []
    lam = None
    for ii in range(n_iter):
        Q_A, R_A = sla.qr(AV, mode="economic")
        _, R_L = sla.qr(LV, mode="economic")
        lam = _select_lambda(regparam, Q_A, R_A, R_L, b, delta, eta)
        lams.append(lam)
        y = _tik_lstsq(R_A, R_L, lam, Q_A.T @ b)
        x = V @ y
        hist.append(x)
        r = (A.T @ (AV @ y - b)) + lam * (L.T @ (LV @ y))
        for _ in range(3):                                       # :86-88
            r = r - V @ (V.T @ r)
        nr = np.linalg.norm(r)
        res.append(nr)
        vn = r / nr
        V = np.column_stack((V, vn))
        AV = np.column_stack((AV, A @ vn))
        LV = np.column_stack((LV, L @ vn))
    info = {"xHistory": hist, "regParam": lam, "regParam_history": lams, "Residual": res, "its": n_iter - 1}
    if x_true is not None:
        info["relError"] = _rre(hist, x_true)
    return x, info


# =====================================================================================
# a12 MMGKS (plain smoothed-Holder weights, group-sparsity and isoTV branches)   trips/solvers/MMGKS.py:28-137
# =====================================================================================
def old_first_derivative_matrix(n):
    """trips/utilities/operators_old.py:66-72: rows 0..n-2 of I - subdiag(1), i.e. row 0 = x[0], row i = x[i] - x[i-1]."""
    import scipy.sparse as sp
    D = sp.spdiags(np.ones(n - 1), -1, n, n)
    return (sp.identity(n) - D).tocsr()[0:-1, :]


def old_first_derivative_2d_matrix(nx, ny):
    """operators_old.py:75-85: vstack(kron(I_nx, D_nx), kron(D_ny, I_ny))."""
    import scipy.sparse as sp
    Dx, Dy = old_first_derivative_matrix(nx), old_first_derivative_matrix(ny)
    return sp.vstack((sp.kron(sp.identity(nx), Dx), sp.kron(Dy, sp.identity(ny)))).tocsr()


def group_sparsity_weights(x, Ls, nx, ny, qnorm):
    """MMGKS.py:78-91 (GS branch), literally: the frame-major iterate is reshaped (nx*ny, nt) in C order (:85), the
    smoothing constant is exp(2) (:87), one weight per row of Ls over its nt entries, tiled nt times (:90)."""
    nt = int(x.reshape(-1, 1).shape[0] / (nx * ny))
    D = Ls.dot(np.reshape(x, (nx * ny, nt)))
    wr = (np.linalg.norm(D[:2 * nx * (ny - 1), :], axis=1) ** 2 + np.exp(2)) ** (qnorm / 2 - 1)
    return np.kron(np.ones((nt, 1)), wr.reshape(-1, 1))


# ---- isoTV branch (MMGKS.py:61-77).  PARITY UNPINNED: its spatial operator is operators_old.py:22-45, built from
# pylops.FirstDerivative / Kronecker / VStack; PyLops is an un-pinned dependency of the reference (setup.py:6) that is
# absent from /root/reference and not installable here.  What follows restates PyLops' PUBLISHED semantics:
# FirstDerivative(n) default = centered 3-point stencil, y[i] = (x[i+1] - x[i-1]) / 2 for 1 <= i <= n-2, first and last
# row zero (edge=False); Kronecker(A, B) = kron(A, B); VStack = row stacking.  (PyLops creates FirstDerivative's output
# in the operator's dtype, float32 at operators_old.py:31; the restatement stays in float64 — a 6e-8 relative rounding
# of the products with L that is part of the stated tolerance.)
def pylops_first_derivative_matrix(n):
    import scipy.sparse as sp
    D = sp.lil_matrix((n, n))
    for i in range(1, n - 1):
        D[i, i - 1], D[i, i + 1] = -0.5, 0.5
    return D.tocsr()


def old_first_derivative_operator_2d(nx, ny):
    """operators_old.py:35-45: VStack(Kronecker(I_nx, D_nx), Kronecker(D_ny, I_ny)) — (nx^2 + ny^2) x nx^2, needs nx = ny."""
    import scipy.sparse as sp
    Dx, Dy = pylops_first_derivative_matrix(nx), pylops_first_derivative_matrix(ny)
    return sp.vstack((sp.kron(sp.identity(nx), Dx), sp.kron(Dy, sp.identity(ny)))).tocsr()


def old_spatial_derivative_operator(nx, ny, nt):
    """operators_old.py:47-53: Kronecker(I_nt, D_spatial)."""
    import scipy.sparse as sp
    return sp.kron(sp.identity(nt), old_first_derivative_operator_2d(nx, ny)).tocsr()


def old_time_derivative_operator(nx, ny, nt):
    """operators_old.py:55-61: Kronecker(D_nt, I_{nx^2})."""
    import scipy.sparse as sp
    return sp.kron(pylops_first_derivative_matrix(nt), sp.identity(nx ** 2)).tocsr()


def iso_tv_weights(x, u, nx, ny, epsilon, qnorm):
    """MMGKS.py:61-77 (= weights.py:29-40), literally: the frame-major iterate is reshaped (nx^2, nt) in C order (:71), the
    two directional derivatives of that array share one weight (LsX1^2 + LsX2^2 + eps^2)^((q-2)/4) (:75: exponent (q-2)/4,
    sic), laid out as [weightx.flatten(); weightx.flatten()] (:76) in front of the temporal weights of u[2*spacen*nt:]
    (:77) — whatever row order the caller's L has."""
    x = np.asarray(x, dtype=np.float64)
    nt = int(x.reshape(-1, 1).shape[0] / (nx * ny))
    Ls = old_first_derivative_operator_2d(nx, ny)
    spacen = int(Ls.shape[0] / 2)
    spacent = spacen * nt
    LsX = Ls @ x.reshape(nx ** 2, nt)
    wx = (LsX[:spacen, :] ** 2 + LsX[spacen:2 * spacen, :] ** 2 + epsilon ** 2) ** ((qnorm - 2) / 4)
    wx = np.concatenate((wx.flatten(), wx.flatten()))
    wt = (np.asarray(u, dtype=np.float64).reshape(-1, 1)[2 * spacent:] ** 2 + epsilon ** 2) ** ((qnorm - 2) / 4)
    return np.concatenate((wx.reshape(-1, 1), wt))


def mmgks(A, b, L, pnorm=2, qnorm=1, projection_dim=3, n_iter=5, regparam="gcv", x_true=None,
          epsilon=0.1, delta=None, eta=1.01, GS=False, prob_dims=None, isoTV=False):
    b = np.asarray(b, dtype=np.float64).reshape(-1, 1)
    _, _, V = golub_kahan(A, b, projection_dim)
    x = A.T @ b                                                  # :43
    Ls = None
    if GS:                                                       # :45-52: L is REPLACED by kron(I_nt, Ls)
        import scipy.sparse as sp
        nx, ny, nt = prob_dims
        Ls = old_first_derivative_2d_matrix(nx, ny)
        L = MatrixOp(sp.kron(sp.identity(nt), Ls).tocsr())
    AV, LV = A @ V, L @ V
    hist, lams, res = [], [], []
    lam = None
    its = 0
    for ii in range(n_iter):
        its = ii
        wf = smoothed_holder_weights(A @ x - b, epsilon, pnorm)          # :56-57
        Q_A, R_A = sla.qr(AV * wf, mode="economic")
        if isoTV:                                                # tested BEFORE the GS option (:61 / :78)
            wr = iso_tv_weights(x, L @ x, prob_dims[0], prob_dims[1], epsilon, qnorm)
        elif GS:
            wr = group_sparsity_weights(x, Ls, prob_dims[0], prob_dims[1], qnorm)   # :78-91
        else:
            wr = smoothed_holder_weights(L @ x, epsilon, qnorm).reshape(-1, 1)   # :60,93
        _, R_L = sla.qr(LV * wr, mode="economic")
        lam = _select_lambda(regparam, Q_A, R_A, R_L, wf * b, delta, eta)    # weighted b for the selector (:97-99)
        lams.append(lam)
        y = _tik_lstsq(R_A, R_L, lam, Q_A.T @ b)                 # UNWEIGHTED b here (:106, reference quirk)
        x = V @ y
        hist.append(x)
        if ii >= R_L.shape[0]:                                   # :109-110
            break
        r = (A.T @ (wf * (AV @ y - b))) + lam * (L.T @ (wr * (LV @ y)))
        for _ in range(2):                                       # :119-120
            r = r - V @ (V.T @ r)
        nr = np.linalg.norm(r)
        vn = r / nr
        V = np.column_stack((V, vn))
        AV = np.column_stack((AV, A @ vn))
        LV = np.column_stack((LV, L @ vn))
        res.append(nr)
    info = {"xHistory": hist, "regParam": lam, "regParam_history": lams, "Residual": res, "its": its}
    if x_true is not None:
        info["relError"] = _rre(hist, x_true)
    return x, info


# =====================================================================================
# SURVEY §8f rank 2 — one-shot projection solvers
# =====================================================================================
def golub_kahan_tikhonov(A, b, n_iter=3, regparam="gcv", delta=None, eta=1.01):
    """trips/solvers/GK_Tikhonov.py:23-76.  NOTE the reference ignores n_iter: golub_kahan(A, b, n_iter=3) (:60)."""
    b = np.asarray(b, dtype=np.float64).reshape(-1, 1)
    U, B, V = golub_kahan(A, b, 3)
    bhat = U.T @ b
    k = B.shape[1]
    if regparam == "gcv":
        Qb, s, _ = sla.svd(B, full_matrices=False)
        lam = gcv_choose(Qb, np.diag(s), np.eye(k), bhat, variant="modified", fullsize=A.shape[0])
    elif regparam == "dp":
        lam = discrepancy_choose(U, B, np.eye(k), b, delta, eta, L_is_identity=True)
    else:
        lam = regparam
    y = _tik_lstsq(B, np.eye(k), lam, bhat)
    return V @ y, lam


def arnoldi_tikhonov(A, b, n_iter=3, regparam="gcv", delta=None, eta=1.01):
    """trips/solvers/A_Tikhonov.py:23-97 (uses the quirky `arnoldi`, decompositions.py:20-116).  dp_stop=True is not
    restated: the reference call `arnoldi(A, b, n_iter, dp_stop, **kwargs)` (:70) then raises TypeError."""
    if A.shape[0] != A.shape[1]:
        raise ValueError("The observation matrix A must be square for this method.")
    b = np.asarray(b, dtype=np.float64).reshape(-1, 1)
    Q, H = arnoldi(A, b, n_iter)
    bhat = Q.T @ b
    k = H.shape[1]
    if regparam == "gcv":
        Qh, s, _ = sla.svd(H, full_matrices=False)
        lam = gcv_choose(Qh, np.diag(s), np.eye(k), bhat)
        y = sla.solve(H.T @ H + lam * np.eye(k), H.T @ bhat)
    elif regparam == "dp":
        lam = discrepancy_choose(Q, H, np.eye(k), b, delta, eta, L_is_identity=True)
        y = _tik_lstsq(H, np.eye(k), lam, bhat)
    else:
        lam = regparam
        y = sla.solve(H.T @ H + lam * np.eye(k), H.T @ bhat)
    return Q[:, :-1] @ y, lam


def gmres(A, b, n_iter=3):
    """trips/solvers/GMRES.py:19-51.  NOTE the reference ignores n_iter (arnoldi(A, b_vec, n_iter=5), :46) and solves
    lstsq(H.T, H.T @ bhat), i.e. the minimum-norm y in R^{k+1}, then x = V_{k+1} y (:49-50)."""
    if A.shape[0] != A.shape[1]:
        raise ValueError("Arnoldi can not be used. The operator is not square")
    b = np.asarray(b, dtype=np.float64).reshape(-1, 1)
    Q, H = arnoldi(A, b, 5)
    bhat = Q.T @ b
    y = np.linalg.lstsq(H.T, H.T @ bhat, rcond=None)[0]
    return Q @ y


# =====================================================================================
# SURVEY §8f rank 1 — fan-beam line projector (pinned to the reference's two rendered ASTRA outputs: tests/test_oracle_golden.py)
# =====================================================================================
class FanBeam2D(_Op):
    """astra 'fanflat' + 'line_fanflat' of trips/test_problems/Tomography.py:53-88, restated by brute force: the weight of
    pixel (r,c) for ray (angle a, detector d) is the length of the segment source -> detector-pixel centre inside the
    pixel square, obtained by clipping the ray against EVERY pixel (no traversal logic to get wrong).  Convention as in
    csrc/fanbeam2d.hip: source (SOD sin t, -SOD cos t), detector centre (-ODD sin t, ODD cos t), detector axis (cos t, sin t)."""

    def __init__(self, N, angles, n_det=None, sod=None, odd=None, pitch=None):
        self.N = int(N)
        self.angles = np.asarray(angles, dtype=np.float64).reshape(-1)
        self.nd = int(np.sqrt(2) * self.N) if n_det is None else int(n_det)
        self.sod = 3.0 * self.N if sod is None else float(sod)
        self.odd = 1.0 * self.N if odd is None else float(odd)
        self.pitch = (self.sod + self.odd) / self.sod if pitch is None else float(pitch)
        self.shape = (len(self.angles) * self.nd, self.N * self.N)
        self._M = None

    def matrix(self):
        if self._M is None:
            N, nd, half = self.N, self.nd, 0.5 * self.N
            rr, cc = np.meshgrid(np.arange(N), np.arange(N), indexing="ij")
            x0, x1 = (cc - half).reshape(-1), (cc + 1 - half).reshape(-1)
            y1 = (half - rr).reshape(-1)
            y0 = y1 - 1.0
            rows, cols, vals = [], [], []
            for a, th in enumerate(self.angles):
                ct, st = np.cos(th), np.sin(th)
                sx, sy = self.sod * st, -self.sod * ct
                for d in range(nd):
                    off = (d - 0.5 * (nd - 1)) * self.pitch
                    ex, ey = -self.odd * st + off * ct, self.odd * ct + off * st
                    dx, dy = ex - sx, ey - sy
                    L = np.hypot(dx, dy)
                    t0 = np.zeros(N * N)
                    t1 = np.ones(N * N)
                    for (lo, hi, s0, dd, axis) in ((x0, x1, sx, dx, 0), (y0, y1, sy, dy, 1)):
                        if abs(dd) > 1e-14:
                            ta, tb = (lo - s0) / dd, (hi - s0) / dd
                            t0 = np.maximum(t0, np.minimum(ta, tb))
                            t1 = np.minimum(t1, np.maximum(ta, tb))
                        elif axis == 0:     # axis-parallel ray: a ray running exactly ALONG a pixel boundary belongs to ONE
                            # side — the pixel with the larger column (row) index, as a `floor(c + 0.5)` pixel choice gives
                            # it; the reference's ASTRA sinogram of the 32^2 demo (tests/golden/fanbeam_demo_image.npz: view 0,
                            # detector 22 runs along x = 0) shows one column's mass there, not two
                            t1 = np.where((s0 < lo) | (s0 >= hi), -1.0, t1)
                        else:               # rows grow downwards: the larger row index is the pixel BELOW the boundary
                            t1 = np.where((s0 <= lo) | (s0 > hi), -1.0, t1)
                    ln = (t1 - t0) * L
                    nz = np.nonzero(ln > 1e-12)[0]
                    rows.append(np.full(nz.size, a * nd + d))
                    cols.append(nz)
                    vals.append(ln[nz])
            self._M = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=self.shape).tocsr()
        return self._M

    def _fwd(self, x):
        return self.matrix() @ x

    def _adj(self, y):
        return self.matrix().T @ y
