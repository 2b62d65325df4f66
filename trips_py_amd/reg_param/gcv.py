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


def generalized_crossvalidation(R_A, R_L, rhs, variant="standard", fullsize=None, **_ignored):
    """lambda = argmin G over [1e-9, 1e2] by scipy's bounded Brent search, same settings as gcv.py:94-95."""
    rhs = np.asarray(rhs, dtype=np.float64).reshape(-1)
    R_A, R_L = np.asarray(R_A), np.asarray(R_L)
    k = R_A.shape[0]
    if (R_A.shape == (k, k) and R_L.shape == (k, k) and not np.any(R_A - np.diag(np.diag(R_A)))
            and not np.any(R_L - np.eye(k))):
        s = np.diag(R_A).copy()
        fun = lambda lam: gcv_function_diag(lam, s, rhs, variant, fullsize)
    else:
        fun = lambda lam: gcv_function(lam, R_A, R_L, rhs, variant, fullsize)
    return sopt.fminbound(fun, 1e-9, 1e2, xtol=1e-12, maxfun=1000, disp=0)
