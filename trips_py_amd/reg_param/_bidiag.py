"""Singular values of the Golub-Kahan bidiagonal B_k and the first row of its left singular vectors — all that the
hybrid solvers' GCV needs of  svd(B_k)  (Hybrid_LSQR.py:81-84:  Q_A^T (beta0 e1) = beta0 * U[0, :k]).

LAPACK's bidiagonal QR (dbdsqr) applied to B_k directly, with the rotations accumulated on the single row e1^T: O(k^2)
instead of the dense (k+1) x k SVD.  SciPy does not wrap dbdsqr in scipy.linalg.lapack but exports it, like every LAPACK
routine, through the C API of scipy.linalg.cython_lapack (the capsule table other compiled extensions bind to); it is
called from here through ctypes.  If that table is absent, the dense SVD is used (same numbers, slower)."""
import ctypes

import numpy as np
import scipy.linalg as sla

_dbdsqr = None
_INT_P, _DBL_P = ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_double)


def _bind():
    global _dbdsqr
    if _dbdsqr is None:
        try:
            import scipy.linalg.cython_lapack as cl
            cap = cl.__pyx_capi__["dbdsqr"]
            api = ctypes.pythonapi
            api.PyCapsule_GetName.restype, api.PyCapsule_GetName.argtypes = ctypes.c_char_p, [ctypes.py_object]
            api.PyCapsule_GetPointer.restype = ctypes.c_void_p
            api.PyCapsule_GetPointer.argtypes = [ctypes.py_object, ctypes.c_char_p]
            ptr = api.PyCapsule_GetPointer(cap, api.PyCapsule_GetName(cap))
            # dbdsqr(uplo, n, ncvt, nru, ncc, d, e, vt, ldvt, u, ldu, c, ldc, work, info) — Fortran: everything by reference
            proto = ctypes.CFUNCTYPE(None, ctypes.c_char_p, _INT_P, _INT_P, _INT_P, _INT_P, _DBL_P, _DBL_P, _DBL_P, _INT_P,
                                     _DBL_P, _INT_P, _DBL_P, _INT_P, _DBL_P, _INT_P)
            _dbdsqr = proto(ptr)
        except Exception:                     # noqa: BLE001  (no capsule table in this SciPy build)
            _dbdsqr = False
    return _dbdsqr or None


def bidiag_svd_first_row(alphas, betas):
    """B = lower bidiagonal (k+1) x k, diagonal alphas[0..k), sub-diagonal betas[0..k)  ->  (s, u0):
    s = singular values (descending), u0[i] = first component of the i-th left singular vector (sign arbitrary)."""
    s, proj = bidiag_svd_project(alphas, betas)
    return s, proj[:-1]


def bidiag_svd_project(alphas, betas, row=None):
    """(s, row^T U) for the FULL (k+1) x (k+1) left factor U of B: k singular values (descending) and k+1 projections,
    the last one on the left null vector of B.  row = None means e1.  Signs of the projections are arbitrary."""
    al = np.asarray(alphas, dtype=np.float64)
    be = np.asarray(betas, dtype=np.float64)
    k = al.size
    fn = _bind()
    if fn is None:
        B = np.zeros((k + 1, k))
        B[np.arange(k), np.arange(k)] = al
        B[np.arange(1, k + 1), np.arange(k)] = be
        U, s, _ = sla.svd(B)
        return s, (U[0].copy() if row is None else U.T @ np.asarray(row, dtype=np.float64).reshape(-1))
    # square it with a zero last column: (k+1) x (k+1) lower bidiagonal, d = [alphas, 0]; the extra singular value is 0
    n = k + 1
    d = np.zeros(n)
    d[:k] = al
    e = be.copy()
    if row is None:
        u = np.zeros(n)
        u[0] = 1.0                             # 1 x n "U": on exit e1^T U_B
    else:
        u = np.array(row, dtype=np.float64).reshape(-1)
        if u.size != n:
            raise ValueError("row must have k + 1 entries")
    work = np.empty(4 * n)
    dummy = np.zeros(1)
    one, zero, nn, info = ctypes.c_int(1), ctypes.c_int(0), ctypes.c_int(n), ctypes.c_int(0)
    fn(b"L", ctypes.byref(nn), ctypes.byref(zero), ctypes.byref(one), ctypes.byref(zero), d.ctypes.data_as(_DBL_P),
       e.ctypes.data_as(_DBL_P), dummy.ctypes.data_as(_DBL_P), ctypes.byref(one), u.ctypes.data_as(_DBL_P),
       ctypes.byref(one), dummy.ctypes.data_as(_DBL_P), ctypes.byref(one), work.ctypes.data_as(_DBL_P), ctypes.byref(info))
    if info.value != 0:
        raise np.linalg.LinAlgError(f"dbdsqr: info = {info.value}")
    return d[:k], u


# --------------------------------------------------------------------------------------------------------------------
# Hybrid-GMRES: the projected problem  min || H y - beta0 e1 ||^2 + lam || y ||^2  (H the (k+1) x k Arnoldi Hessenberg matrix,
# Hybrid_GMRES.py:54-77) brought to the Golub-Kahan form the LSQR path already solves without an SVD.
# LAPACK's dgebrd on the (k+1) x (k+1) matrix M = [beta0 e1 | H]:  M = Q B P^T, B upper bidiagonal (d, e).  The first column of M is
# a multiple of e1, so the first left reflector is the identity and Q^T (beta0 e1) = d[0] e1; P = diag(1, P') acts on H's columns only.
# Hence  H = Q B[:, 1:] P'^T  with  B[:, 1:]  LOWER bidiagonal (k+1) x k, diagonal e[0..k), sub-diagonal d[1..k], and with y = P' z:
#     || H y - beta0 e1 || = || B[:, 1:] z - d[0] e1 || ,   || y || = || z ||
# — the same singular values and the same products u_i . bhat as the SVD of H gives, so GCV ('standard': fullsize k + 1,
# Hybrid_GMRES.py:58), the discrepancy principle and the Tikhonov minimiser follow from (e, d) in O(k) per evaluation, and the one
# O(k^3) step per iteration is the bidiagonalisation (a third of the dense SVD's cost: no vectors, no iteration).
_gebrd = None


def _bind_gebrd():
    global _gebrd
    if _gebrd is None:
        try:
            import scipy.linalg.cython_lapack as cl
            api = ctypes.pythonapi
            api.PyCapsule_GetName.restype, api.PyCapsule_GetName.argtypes = ctypes.c_char_p, [ctypes.py_object]
            api.PyCapsule_GetPointer.restype = ctypes.c_void_p
            api.PyCapsule_GetPointer.argtypes = [ctypes.py_object, ctypes.c_char_p]

            def fn(name, *argtypes):
                cap = cl.__pyx_capi__[name]
                return ctypes.CFUNCTYPE(None, *argtypes)(api.PyCapsule_GetPointer(cap, api.PyCapsule_GetName(cap)))
            # dgebrd(m, n, a, lda, d, e, tauq, taup, work, lwork, info); dormbr(vect, side, trans, m, n, k, a, lda, tau, c, ldc, work, lwork, info)
            _gebrd = (fn("dgebrd", _INT_P, _INT_P, _DBL_P, _INT_P, _DBL_P, _DBL_P, _DBL_P, _DBL_P, _DBL_P, _INT_P, _INT_P),
                      fn("dormbr", ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, _INT_P, _INT_P, _INT_P, _DBL_P, _INT_P, _DBL_P,
                         _DBL_P, _INT_P, _DBL_P, _INT_P, _INT_P))
        except Exception:                     # noqa: BLE001  (no capsule table in this SciPy build)
            _gebrd = False
    return _gebrd or None


_raw_ptrs = {}


def lapack_pointer(name):
    """SciPy's LAPACK routine `name` (scipy.linalg.cython_lapack's capsule table) as a plain C address, or None."""
    if name not in _raw_ptrs:
        try:
            import scipy.linalg.cython_lapack as cl
            api = ctypes.pythonapi
            api.PyCapsule_GetName.restype, api.PyCapsule_GetName.argtypes = ctypes.c_char_p, [ctypes.py_object]
            api.PyCapsule_GetPointer.restype = ctypes.c_void_p
            api.PyCapsule_GetPointer.argtypes = [ctypes.py_object, ctypes.c_char_p]
            cap = cl.__pyx_capi__[name]
            _raw_ptrs[name] = api.PyCapsule_GetPointer(cap, api.PyCapsule_GetName(cap))
        except Exception:                     # noqa: BLE001  (no capsule table in this SciPy build)
            _raw_ptrs[name] = None
    return _raw_ptrs[name]


def lapack_pointers():
    """(dgebrd, dormbr) as plain C addresses for libtrk's worker thread (trk_host_worker_set_lapack), or None."""
    fns = _bind_gebrd()
    if fns is None:
        return None
    return tuple(ctypes.cast(f, ctypes.c_void_p).value for f in fns)


class HessenbergBidiag:
    """The reduction above for one H: `.alphas`, `.betas`, `.beta0` (the Golub-Kahan triple of B[:, 1:] and Q^T bhat) and `.back(z)`
    = P' z.  `available()` is False when SciPy's LAPACK capsule table is missing (callers keep their SVD path)."""

    @staticmethod
    def available():
        return _bind_gebrd() is not None

    def __init__(self, H, beta0):
        gebrd, self._ormbr = _bind_gebrd()
        H = np.asarray(H, dtype=np.float64)
        k = H.shape[1]
        n = k + 1
        M = np.zeros((n, n), order="F")
        M[0, 0] = beta0
        M[:, 1:] = H
        d, e, tq, tp = np.empty(n), np.empty(max(1, n - 1)), np.empty(n), np.empty(n)
        lwork = 64 * n
        work = np.empty(lwork)
        ci = ctypes.c_int
        nn, lw, info = ci(n), ci(lwork), ci(0)
        p = lambda a: a.ctypes.data_as(_DBL_P)      # noqa: E731
        gebrd(ctypes.byref(nn), ctypes.byref(nn), p(M), ctypes.byref(nn), p(d), p(e), p(tq), p(tp), p(work), ctypes.byref(lw),
              ctypes.byref(info))
        if info.value != 0:
            raise RuntimeError(f"dgebrd failed (info = {info.value})")
        self.k, self._M, self._tp, self._tq, self._work = k, M, tp, tq, work
        self.alphas, self.betas, self.beta0 = e[:k].copy(), d[1:n].copy(), float(d[0])

    def back(self, z):
        """y = P' z (k-vector)."""
        n = self.k + 1
        c = np.zeros(n)
        c[1:] = z
        ci = ctypes.c_int
        nn, one, lw, info = ci(n), ci(1), ci(self._work.size), ci(0)
        p = lambda a: a.ctypes.data_as(_DBL_P)      # noqa: E731
        self._ormbr(b"P", b"L", b"N", ctypes.byref(nn), ctypes.byref(one), ctypes.byref(nn), p(self._M), ctypes.byref(nn), p(self._tp),
                    p(c), ctypes.byref(nn), p(self._work), ctypes.byref(lw), ctypes.byref(info))
        if info.value != 0:
            raise RuntimeError(f"dormbr failed (info = {info.value})")
        return c[1:].copy()

    def left_t(self, p_vec):
        """Q^T p for a (k+1)-vector p in the Arnoldi basis' coordinates (e.g. V_{k+1}^T b of the discrepancy principle): its
        coordinates in the left Golub-Kahan basis of the bidiagonal form."""
        n = self.k + 1
        c = np.array(p_vec, dtype=np.float64).reshape(-1).copy()
        if c.size != n:
            raise ValueError(f"left_t: expected {n} entries")
        ci = ctypes.c_int
        nn, one, lw, info = ci(n), ci(1), ci(self._work.size), ci(0)
        p = lambda a: a.ctypes.data_as(_DBL_P)      # noqa: E731
        self._ormbr(b"Q", b"L", b"T", ctypes.byref(nn), ctypes.byref(one), ctypes.byref(nn), p(self._M), ctypes.byref(nn), p(self._tq),
                    p(c), ctypes.byref(nn), p(self._work), ctypes.byref(lw), ctypes.byref(info))
        if info.value != 0:
            raise RuntimeError(f"dormbr failed (info = {info.value})")
        return c


def bidiag_tikhonov_host(alphas, betas, beta0, mu):
    """argmin || [B; mu I] z - beta0 e1 || for B lower bidiagonal (diagonal alphas, sub-diagonal betas), on the host in O(k)
    (libtrk's trk_host_bidiag_tikhonov: Givens recurrence + back substitution; no GPU involved)."""
    from .. import _lib
    lib = _lib.load()
    al = np.ascontiguousarray(alphas, dtype=np.float64)
    be = np.ascontiguousarray(betas, dtype=np.float64)
    z = np.empty(al.size, dtype=np.float64)
    rc = lib.trk_host_bidiag_tikhonov(al.ctypes.data, be.ctypes.data, int(al.size), float(beta0), float(mu), 0, z.ctypes.data)
    _lib.check(rc, "trk_host_bidiag_tikhonov")
    return z
