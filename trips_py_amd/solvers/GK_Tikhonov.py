"""Golub-Kahan-Tikhonov (one-shot) on the HIP engine — trips/solvers/GK_Tikhonov.py:23-76."""
import numpy as np
import scipy.linalg as sla

from .._io import Formatter, as_operator
from ..krylov import GKState
from ._common import check_delta, choose_lambda, tikhonov_lstsq, small_host_blas


@small_host_blas
def Golub_Kahan_Tikhonov(A, b, n_iter=3, regparam="gcv", **kwargs):
    """Returns (x, lambda).  NOTE the reference ignores `n_iter` and always takes 3 Golub-Kahan steps
    (`golub_kahan(A, b, n_iter=3, dp_stop=0)`, :60); reproduced."""
    A = as_operator(A)
    check_delta(regparam, kwargs)
    eng = A.engine
    m, n = A.shape
    fmt = Formatter(b)
    bv = eng.to_vec(b, m)
    gk = GKState(A, bv, 3)
    for _ in range(3):
        gk.step()
    k = 3
    P = eng.scalars(k + 1)
    eng.gemv_t(gk.U.data, k + 1, bv, P.ref(0))                     # bhat = U.T @ b (:62)
    eng.allreduce(P, 0, k + 1)
    bhat = P.host(0, k + 1)
    B = gk.B()
    if isinstance(regparam, str) and regparam == "gcv":
        Qb, s, _ = sla.svd(B, full_matrices=False)
        lam = choose_lambda("gcv", np.diag(s), np.eye(k), Qb.T @ bhat, 0.0, kwargs, variant="modified", fullsize=m)
    elif isinstance(regparam, str) and regparam == "dp":
        lam = choose_lambda("dp", None, None, None, 0.0, kwargs, L_is_identity=True, dp_A=B, dp_bproj=bhat)
    else:
        lam = regparam
    Y = eng.scalars(k)
    Y.set(0, tikhonov_lstsq(B, np.eye(k), lam, bhat))
    x = eng.empty(n)
    eng.gemv_n(gk.V.data, k, Y.ref(0), x)
    return fmt.vec(x), lam
