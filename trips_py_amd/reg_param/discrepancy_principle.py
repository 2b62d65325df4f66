"""Discrepancy principle on projected factors (reference: trips/utilities/reg_param/discrepancy_principle.py:19-99,
dptype='tikhonov')."""
import numpy as np
import scipy.linalg as sla


def discrepancy_principle(A, L, bproj, resid2, delta=None, eta=1.01, L_is_identity=False, explicitProj=False, **_ignored):
    """alpha with ||A x_alpha - b||^2 = (eta*delta)^2 by the reference's Newton iteration on beta = 1/alpha.

    A       projected operator: B_k ((k+1) x k, hybrid solvers) or R_A (k x k, GKS / MMGKS)
    L       projected regulariser (ignored when L_is_identity)
    bproj   Q^T b  (k+1 or k entries)
    resid2  ||b - Q Q^T b||^2  (used where the reference uses it: square `A`, or explicitProj)
    Returns 0 when the discrepancy cannot be reached yet (:76 `testzero >= 0` branch)."""
    if not isinstance(delta, (float, int)):
        raise Exception("A value for the noise level delta was not provided and the discrepancy principle cannot be applied. "
                        "Please supply a value of delta based on the estimated noise level of the problem, or choose the "
                        "regularization parameter according to gcv.")
    A = np.asarray(A, dtype=np.float64)
    bp = np.asarray(bproj, dtype=np.float64).reshape(-1, 1)
    if L_is_identity:
        Anew = A
    else:
        L = np.asarray(L, dtype=np.float64)
        _, SL, VL = sla.svd(L)
        if not (L.shape[0] >= L.shape[1] and SL[-1] != 0):
            raise NotImplementedError("projected regulariser with a null space (discrepancy_principle.py:45-66)")
        Anew = A @ (VL.T @ np.diag(SL ** (-1.0)))
    U, S, _ = sla.svd(Anew)
    sv = (S ** 2)
    bhat = U.T @ bp
    r, c = Anew.shape
    target = (eta * delta) ** 2
    if r > c:
        sv = np.append(sv, np.zeros(r - c))
        testzero = np.linalg.norm(bhat[c - r:, :]) ** 2 - target + (resid2 if explicitProj else 0.0)
    else:
        testzero = resid2 - target
    if not testzero < 0:
        return 0
    sv = sv.reshape(-1, 1)
    extra = resid2 if explicitProj else 0.0
    beta, it, alpha = 1e-8, 0, None
    while it < 30 or (it <= 100 and abs(alpha) < 1e-16):
        z = bhat / (sv * beta + 1)
        f = np.linalg.norm(z) ** 2 + extra - target
        w = z / (sv * beta + 1)
        fp = 2 / beta * (z.T @ (w - z))
        beta_new = beta - f / fp
        if abs(beta_new - beta) < 1e-12 * beta:
            break
        beta = beta_new
        alpha = 1 / beta_new[0, 0]
        it += 1
    return alpha
