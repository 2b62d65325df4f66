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
