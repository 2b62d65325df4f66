"""Arnoldi-Tikhonov (one-shot) on the HIP engine — trips/solvers/A_Tikhonov.py:23-97."""
import numpy as np
import scipy.linalg as sla

from .._io import Formatter, as_operator
from ..decompositions import arnoldi_device
from ._common import check_delta, choose_lambda, tikhonov_lstsq, small_host_blas


@small_host_blas
def Arnoldi_Tikhonov(A, b, n_iter=3, regparam="gcv", **kwargs):
    """Returns (x, lambda).  Built on the reference's `arnoldi` (with its skipped-newest-vector quirk)."""
    A = as_operator(A)
    if A.shape[0] != A.shape[1]:
        raise ValueError("The observation matrix A must be square for this method.")
    if "dp_stop" in kwargs:
        # the reference passes dp_stop positionally AND inside **kwargs (A_Tikhonov.py:69-70) and fails the same way
        raise TypeError("arnoldi() got multiple values for argument 'dp_stop'")
    if isinstance(regparam, str) and regparam == "dp" and kwargs.get("delta") is None:
        check_delta(regparam, kwargs)
    eng = A.engine
    n = A.shape[0]
    fmt = Formatter(b)
    bv = eng.to_vec(b, n)
    Q, H = arnoldi_device(A, bv, n_iter)
    k = H.shape[1]
    P = eng.scalars(k + 1)
    eng.gemv_t(Q.data, k + 1, bv, P.ref(0))                        # bhat = Vdp1.T @ b (:73)
    eng.allreduce(P, 0, k + 1)
    bhat = P.host(0, k + 1)
    if isinstance(regparam, str) and regparam == "gcv":
        Qh, s, _ = sla.svd(H, full_matrices=False)
        lam = choose_lambda("gcv", np.diag(s), np.eye(k), Qh.T @ bhat, 0.0, kwargs)
        y = sla.solve(H.T @ H + lam * np.eye(k), H.T @ bhat)       # normal equations here (:86), lstsq for 'dp' (:91)
    elif isinstance(regparam, str) and regparam == "dp":
        lam = choose_lambda("dp", None, None, None, 0.0, kwargs, L_is_identity=True, dp_A=H, dp_bproj=bhat)
        y = tikhonov_lstsq(H, np.eye(k), lam, bhat)
    else:
        lam = regparam
        y = sla.solve(H.T @ H + lam * np.eye(k), H.T @ bhat)
    Y = eng.scalars(k)
    Y.set(0, y)
    x = eng.empty(n)
    eng.gemv_n(Q.data, k, Y.ref(0), x)                             # x = Vd @ y
    return fmt.vec(x), lam
