"""L-curve corner (maximum curvature) on projected factors (reference: trips/utilities/reg_param/l_curve.py:23-203)."""
import numpy as np
import scipy.optimize as sopt


def l_curve_curvature(lam, A, L, b):
    """kappa(lambda) of the curve (||A x_l - b||^2, ||L x_l||^2), with the reference's derivative formulas
    (x' = -(C+lD)^-1 D x,  x'' = 2 (C+lD)^-1 (D x' - D (C+lD)^-1 D x)), least-squares solves as at l_curve.py:44,64,84-85."""
    b = np.asarray(b, dtype=np.float64).reshape(-1, 1)
    C, D = A.T @ A, L.T @ L
    M = C + lam * D
    solve = lambda rhs: np.linalg.lstsq(M, rhs, rcond=None)[0]
    x = solve(A.T @ b)
    Dx = D @ x
    dx = -solve(Dx)
    d2x = 2 * solve(D @ dx - D @ solve(Dx))
    fr, gr = A @ x - b, L @ x
    Adx, Ldx = A @ dx, L @ dx
    f1 = (2 * fr.T @ Adx).item()
    g1 = (2 * gr.T @ Ldx).item()
    f2 = (2 * (Adx.T @ Adx + fr.T @ (A @ d2x))).item()
    g2 = (2 * (Ldx.T @ Ldx + gr.T @ (L @ d2x))).item()
    return (-g1 * f2 + f1 * g2) / (g1 ** 2 + f1 ** 2) ** 1.5


def l_curve(A, L, b, **_ignored):
    """lambda = argmax curvature over [1e-9, 2] (l_curve.py:201-202)."""
    return sopt.fminbound(lambda l: -l_curve_curvature(l, A, L, b), 1e-9, 2, xtol=1e-12, maxfun=1000, disp=0)
