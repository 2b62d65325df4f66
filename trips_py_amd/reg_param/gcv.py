"""Generalised cross validation on projected factors (reference: trips/utilities/reg_param/gcv.py:25-95)."""
import numpy as np
import scipy.linalg as sla
import scipy.optimize as sopt


def gcv_function(lam, R_A, R_L, rhs, variant="standard", fullsize=None):
    """G(lambda) = ||R_A x_l - rhs||^2 / (m_eff - trace(R_A (R_A^T R_A + l R_L^T R_L)^-1 R_A^T))^2.

    The reference evaluates the NUMERATOR without its keyword arguments (gcv.py:94: `gcv_numerator(reg_param, Q_A,
    R_A, R_L, b)`), i.e. always in its 'standard' form; only the denominator sees `variant` / `fullsize`
    ('modified': m_eff = fullsize, the Hybrid_LSQR call at Hybrid_LSQR.py:84; otherwise m_eff = rows of R_A)."""
    M = R_A.T @ R_A + lam * (R_L.T @ R_L)
    sol = sla.solve(M, np.column_stack((R_A.T @ rhs, R_A.T)))
    xl, inv = sol[:, 0], sol[:, 1:]
    num = np.linalg.norm(R_A @ xl - rhs) ** 2
    m_eff = fullsize if variant == "modified" else R_A.shape[0]
    return num / (m_eff - np.trace(R_A @ inv)) ** 2


def gcv_function_diag(lam, s, rhs, variant="standard", fullsize=None):
    """The same G(lambda) when R_A = diag(s) (k values, possibly fewer than len(rhs)) and R_L = I — the hybrid solvers'
    case (SVD of the projected matrix, Hybrid_LSQR.py:81-84, Hybrid_GMRES.py:55-58): O(k) instead of a k x k solve.
        x_l = s*rhs/(s^2+lam)  ->  R_A x_l - rhs = -lam/(s^2+lam) * rhs ;  trace = sum s^2/(s^2+lam)."""
    f = s * s / (s * s + lam)
    num = float(np.sum(((1.0 - f) * rhs) ** 2))
    m_eff = fullsize if variant == "modified" else len(s)
    return num / (m_eff - float(np.sum(f))) ** 2


def _host_lib():
    """libtrk.so's host-side minimiser (trk_host_gcv_fminbound).  Loading failures propagate: there is no second
    implementation behind it."""
    from .. import _lib
    return _lib.load()


def fminbound_gcv_diag(s, rhs, m_eff, x1=1e-9, x2=1e2, xtol=1e-12, maxfun=1000):
    """argmin of gcv_function_diag over [x1, x2] — bounded Brent search with the settings of gcv.py:94-95."""
    s = np.ascontiguousarray(s, dtype=np.float64)
    rhs = np.ascontiguousarray(rhs, dtype=np.float64)
    if s.size == 0:
        return sopt.fminbound(lambda lam: gcv_function_diag(lam, s, rhs, "modified", m_eff), x1, x2, xtol=xtol,
                              maxfun=maxfun, disp=0)
    import ctypes
    lib = _host_lib()
    lam = ctypes.c_double(0.0)
    rc = lib.trk_host_gcv_fminbound(s.ctypes.data, rhs.ctypes.data, int(s.size), float(m_eff), float(x1), float(x2),
                                    float(xtol), int(maxfun), ctypes.byref(lam), None, None)
    if rc != 0:
        raise RuntimeError("trk_host_gcv_fminbound failed")
    return lam.value


def fminbound_gcv_bidiag(alphas, betas, beta0, m_eff, x1=1e-9, x2=1e2, xtol=1e-12, maxfun=1000):
    """The same minimiser for the Golub-Kahan projected problem (B_k lower bidiagonal with diagonal `alphas`, sub-diagonal
    `betas`; bhat = beta0 e1; 'modified' GCV with fullsize m_eff: Hybrid_LSQR.py:81-84) WITHOUT the SVD of B_k: G(lam) through
    one LDL^T of a k x k tridiagonal matrix per evaluation (trk_host_gcv_bidiag)."""
    import ctypes
    al = np.ascontiguousarray(alphas, dtype=np.float64)
    be = np.ascontiguousarray(betas, dtype=np.float64)
    if al.size == 0 or al.size != be.size:
        raise ValueError("fminbound_gcv_bidiag: need k >= 1 diagonal and k sub-diagonal entries")
    lib = _host_lib()
    lam = ctypes.c_double(0.0)
    rc = lib.trk_host_gcv_bidiag(al.ctypes.data, be.ctypes.data, int(al.size), float(beta0), float(m_eff), float(x1), float(x2),
                                 float(xtol), int(maxfun), ctypes.byref(lam), None, None)
    if rc != 0:
        raise RuntimeError("trk_host_gcv_bidiag failed")
    return lam.value


def _diagonalise(R_A, R_L, rhs):
    """(R_A, R_L) -> (s, U^T rhs) with G unchanged: substitute z = R_L y, M = R_A R_L^-1 = U diag(s) W^T; then
    R_A (R_A^T R_A + lam R_L^T R_L)^-1 R_A^T = U diag(s^2/(s^2+lam)) U^T.  None if R_L is (numerically) singular."""
    k = R_A.shape[0]
    if R_A.shape != (k, k) or R_L.shape != (k, k) or rhs.size != k:
        return None
    d = np.abs(np.diag(R_L))
    tri = not np.any(np.tril(R_L, -1))
    if tri and (d.min() <= 1e-12 * d.max()):
        return None
    try:
        # (check_finite=False: the same LAPACK calls without SciPy's scans of the operands — NaN / inf surface as a failed SVD or as
        #  non-finite singular values, both answered with None below; this pair of calls runs once per GKS / MMGKS iteration with the
        #  device idle)
        if not (np.all(np.isfinite(R_A)) and np.all(np.isfinite(R_L))):
            return None
        M = (sla.solve_triangular(R_L, R_A.T, trans="T", lower=False, check_finite=False).T if tri
             else sla.solve(R_L.T, R_A.T, check_finite=False).T)
        U, sig, _ = sla.svd(M, check_finite=False)
    except (sla.LinAlgError, ValueError):
        return None
    if not np.all(np.isfinite(sig)):
        return None
    return sig, U.T @ rhs


def generalized_crossvalidation(R_A, R_L, rhs, variant="standard", fullsize=None, **_ignored):
    """lambda = argmin G over [1e-9, 1e2] by bounded Brent search, same settings as gcv.py:94-95.  The pair is first
    brought to (diag(s), I) — it already is in the hybrid solvers — so that every one of the ~60 evaluations is O(k)."""
    rhs = np.asarray(rhs, dtype=np.float64).reshape(-1)
    R_A, R_L = np.asarray(R_A, dtype=np.float64), np.asarray(R_L, dtype=np.float64)
    k = R_A.shape[0]
    m_eff = fullsize if variant == "modified" else R_A.shape[0]
    if (R_A.shape == (k, k) and R_L.shape == (k, k) and not np.any(R_A - np.diag(np.diag(R_A)))
            and not np.any(R_L - np.eye(k))):
        return fminbound_gcv_diag(np.diag(R_A).copy(), rhs, m_eff)
    red = _diagonalise(R_A, R_L, rhs)
    if red is not None:
        return fminbound_gcv_diag(red[0], red[1], m_eff)
    fun = lambda lam: gcv_function(lam, R_A, R_L, rhs, variant, fullsize)
    return sopt.fminbound(fun, 1e-9, 1e2, xtol=1e-12, maxfun=1000, disp=0)
