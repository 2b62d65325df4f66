"""Pieces shared by the projection solvers (host side, float64): stacked Tikhonov least squares, Gram -> triangular
factor, and the lambda dispatcher with the reference's keyword names."""
import numpy as np
import scipy.linalg as sla

from ..reg_param import discrepancy_principle, generalized_crossvalidation, l_curve

NO_DELTA_MSG = ("A value for the noise level delta was not provided and the discrepancy principle cannot be applied. \n"
                "                    Please supply a value of delta based on the estimated noise level of the problem, or choose "
                "the regularization parameter according to gcv or a different stopping criterion.")


_blas_controller = None


def small_host_blas(fn=None, *, when=None):
    """Run a solver with the host BLAS/LAPACK pools limited to one thread.  The host side of these solvers is k-sized
    (k <= a few hundred) SVD / least squares / Cholesky: on a many-core host the threaded OpenBLAS spends ~1 ms per call
    waking its pool for them (measured: 1.3 ms per 100x100 SVD on the MI355X box, 0.1 ms single-threaded).  No-op
    when threadpoolctl is not installed — and when `regparam` is a number: those paths keep the projected problem on the
    device and call no host BLAS, while resizing a 128-thread pool on the way in and out can cost tens of milliseconds
    (measured: a 50-iteration GKS of 18 ms took 92 ms right after another NumPy LAPACK call had populated the pool)."""
    import functools
    import inspect
    import os
    if fn is None:                                              # @small_host_blas(when=...)
        return lambda f: small_host_blas(f, when=when)
    sig = inspect.signature(fn)
    needs_limit = when if when is not None else (lambda rp: isinstance(rp, str))
    disabled = os.environ.get("TRK_NO_BLAS_LIMIT") is not None

    @functools.wraps(fn)
    def wrapped(*args, **kwargs):
        global _blas_controller
        try:
            bound = sig.bind_partial(*args, **kwargs)
            rp = bound.arguments.get("regparam", sig.parameters["regparam"].default if "regparam" in sig.parameters else "gcv")
        except TypeError:
            rp = "gcv"
        # `when`: which regparam values run threaded host LAPACK at all (Hybrid_LSQR: only 'l_curve' — gcv / dp go through
        # dbdsqr and the C searches); changing the pool size is not free: now and then the next BLAS call re-creates the pool
        # (74 ms measured on the 64-thread MI355X host), so the limit is set only where it pays
        if disabled or (not needs_limit(rp) and kwargs.get("device_solve", True)):
            return fn(*args, **kwargs)
        if _blas_controller is None:
            try:
                from threadpoolctl import ThreadpoolController
                _blas_controller = ThreadpoolController()
            except Exception:                                   # pragma: no cover
                _blas_controller = False
        if not _blas_controller:
            return fn(*args, **kwargs)
        with _blas_controller.limit(limits=1, user_api="blas"):
            return fn(*args, **kwargs)
    return wrapped


def check_delta(regparam, kwargs):
    """Same precondition (and exception type) as Hybrid_LSQR.py:55-61, Hybrid_GMRES.py:25-31, GKS.py:29-34."""
    delta = kwargs.get("delta", None)
    dp_stop = kwargs.get("dp_stop", False)
    if (isinstance(regparam, str) and regparam == "dp" or dp_stop is not False) and delta is None:
        raise Exception(NO_DELTA_MSG)
    return delta


def tikhonov_lstsq(M, L, lam, rhs):
    """argmin ||M y - rhs||^2 + lam ||L y||^2 through the stacked least-squares problem, as the reference does
    (Hybrid_LSQR.py:104, Hybrid_GMRES.py:76, GKS.py:74, MMGKS.py:106)."""
    top = np.asarray(rhs, dtype=np.float64).reshape(-1, 1)
    stack = np.vstack((M, np.sqrt(lam) * L))
    full = np.vstack((top, np.zeros((L.shape[0], 1))))
    # same minimiser as the reference's np.linalg.lstsq (SVD driver); the pivoted-QR driver is ~5x faster at k ~ 100
    return sla.lstsq(stack, full, lapack_driver="gelsy", check_finite=False)[0].reshape(-1)


def gram_factor(G):
    """Upper-triangular R with R^T R = G (G symmetric positive definite up to rounding).  Replaces the economic QR of
    the m x k matrix (GKS.py:54-56, MMGKS.py:58-59,94-95): R differs from the Householder R by row signs only, which no
    downstream formula sees.  Falls back to a symmetric eigen-factor when G is numerically semi-definite."""
    G = 0.5 * (G + G.T)
    try:
        return sla.cholesky(G, lower=False)
    except sla.LinAlgError:
        w, Q = sla.eigh(G)
        w = np.maximum(w, w.max() * 1e-15)
        return np.sqrt(w)[:, None] * Q.T


def project_rhs(R, c):
    """Q^T b = R^{-T} (W^T b) for Q R = W."""
    return sla.solve(R.T, np.asarray(c, dtype=np.float64).reshape(-1))


_GRAM_GCV = None


def gram_gcv_host(GA, GL, c_select, c_solve):
    """(lambda, y) of GKS / MMGKS's projected problem with regparam='gcv' from the Gram data, in ONE library call
    (trk_host_gram_gcv: Cholesky factors, Q_A^T b, GCV through the singular values of R_A R_L^-1, the stacked least-squares solve — the sequence
    gram_factor / project_rhs / choose_lambda / tikhonov_lstsq runs through SciPy's wrappers with the device idle), or None where the
    library, SciPy's LAPACK capsules or a factorisation are not to be had — the caller then runs that sequence."""
    global _GRAM_GCV
    import ctypes
    if _GRAM_GCV is None:
        try:
            from .. import _lib
            from ..reg_param._bidiag import lapack_pointer
            lib = _lib.load()
            ptrs = [lapack_pointer(nm) for nm in ("dpotrf", "dtrtrs", "dgebrd", "dormbr", "dbdsqr", "dgelsy")]
            _GRAM_GCV = (lib, (ctypes.c_void_p * 6)(*ptrs)) if (all(ptrs) and hasattr(lib, "trk_host_gram_gcv")) else False
        except Exception:                     # noqa: BLE001
            _GRAM_GCV = False
    if not _GRAM_GCV:
        return None
    lib, ptrs = _GRAM_GCV
    GA = np.asarray(GA, dtype=np.float64)
    GL = np.asarray(GL, dtype=np.float64)
    k = GA.shape[0]
    if GA.shape != (k, k) or GL.shape != (k, k) or GA.strides[1] != 8 or GL.strides[1] != 8 or GA.strides[0] != GL.strides[0]:
        GA, GL = np.ascontiguousarray(GA), np.ascontiguousarray(GL)
    cs = np.ascontiguousarray(c_select, dtype=np.float64).reshape(-1)
    cb = cs if c_solve is c_select else np.ascontiguousarray(c_solve, dtype=np.float64).reshape(-1)
    y = np.empty(k)
    lam, ok = ctypes.c_double(0.0), ctypes.c_int(0)
    rc = lib.trk_host_gram_gcv(ptrs, GA.ctypes.data, GL.ctypes.data, GA.strides[0] // 8, cs.ctypes.data,
                               cb.ctypes.data, int(k), float(k), ctypes.byref(lam), y.ctypes.data, ctypes.byref(ok))
    if rc != 0 or not ok.value:
        return None
    return lam.value, y


def choose_lambda(regparam, R_A, R_L, rhs, resid2, kwargs, variant="standard", fullsize=None, L_is_identity=False,
                  dp_A=None, dp_bproj=None):
    """regparam in {'gcv','dp','l_curve', number}  ->  lambda  (branches of GKS.py:60-70 / Hybrid_LSQR.py:80-100)."""
    if isinstance(regparam, str):
        if regparam == "gcv":
            return generalized_crossvalidation(R_A, R_L, rhs, variant=variant, fullsize=fullsize)
        if regparam == "dp":
            extra = {k: kwargs[k] for k in ("eta", "explicitProj") if k in kwargs}
            return discrepancy_principle(R_A if dp_A is None else dp_A, R_L, rhs if dp_bproj is None else dp_bproj, resid2,
                                         delta=kwargs.get("delta"), L_is_identity=L_is_identity, **extra)
        if regparam == "l_curve":
            return l_curve(R_A, R_L, rhs)
        raise ValueError(f"unknown regparam {regparam!r}")
    return regparam
