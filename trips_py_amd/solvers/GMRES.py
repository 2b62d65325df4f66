"""Basic GMRES (one-shot) on the HIP engine — trips/solvers/GMRES.py:19-51."""
import numpy as np

from .._io import Formatter, as_operator
from ..decompositions import arnoldi_device
from ._common import small_host_blas


@small_host_blas
def GMRES(A, b, n_iter=3, dp_stop=0, **kwargs):
    """Returns x.  NOTE the reference ignores `n_iter` (`arnoldi(A, b_vec, n_iter=5)`, :46) and solves
    `lstsq(H.T, H.T @ bhat)` — the minimum-norm y in R^{k+1} — then x = V_{k+1} y (:49-50); reproduced."""
    A = as_operator(A)
    if A.shape[0] != A.shape[1]:
        raise ValueError("Arnoldi can not be used. The operator is not square")
    eng = A.engine
    n = A.shape[0]
    fmt = Formatter(b)
    bv = eng.to_vec(b, n)
    Q, H = arnoldi_device(A, bv, 5)
    k1 = H.shape[0]
    P = eng.scalars(k1)
    eng.gemv_t(Q.data, k1, bv, P.ref(0))
    eng.allreduce(P, 0, k1)
    bhat = P.host(0, k1).reshape(-1, 1)
    y = np.linalg.lstsq(H.T, H.T @ bhat, rcond=None)[0].reshape(-1)
    Y = eng.scalars(k1)
    Y.set(0, y)
    x = eng.empty(n)
    eng.gemv_n(Q.data, k1, Y.ref(0), x)
    return fmt.vec(x)
