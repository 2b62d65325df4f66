"""Krylov factorisations with the reference's names and return shapes (trips/utilities/decompositions.py:20-255),
computed by the engine (trips_py_amd/krylov.py).

    golub_kahan(A, b, n_iter)        -> (U  m x (n_iter+1),  S  (n_iter+1) x n_iter,  V  n x n_iter)      (:118-205)
    arnoldi(A, b, n_iter)            -> (Q  n x (n_iter+1),  H  (n_iter+1) x n_iter)                       (:20-116)
    golub_kahan_update(A, U, S, V)   -> one more GK step                                                    (:230-255)
    arnoldi_update(A, V, H)          -> one more Arnoldi step                                               (:207-228)

The `*_update` functions accept either the reference's arrays (NumPy, n x k) — then the bases are uploaded, one step is
taken and NumPy arrays come back, O(k n) traffic like the reference's own hstack — or the handles they returned the
previous time (`KrylovArrays`), in which case the bases stay on the GPU and the step costs two operator applies.
"""
import numpy as np
import torch

from ._io import as_operator
from .engine import Coef
from .krylov import ArnoldiState, DeviceBasis, GKState


def _fmt_like(b):
    return not isinstance(b, torch.Tensor)


def golub_kahan_device(A, b, n_iter, dp_stop=False, **kwargs):
    """`golub_kahan` on the device -> GKState.  With dp_stop the factorisation halts once the projected least-squares
    solution meets the discrepancy principle ||A x_k - b|| <= gk_eta * gk_delta (decompositions.py:148-149,167-195;
    defaults 1.001 and 0.001 — NOT the solvers' `delta`)."""
    A = as_operator(A)
    n_iter = int(n_iter)
    gk = GKState(A, b, n_iter)
    if not dp_stop:
        for _ in range(n_iter):
            gk.step(sync=False)                 # B_k is downloaded once, when the caller asks for it
        return gk
    eta = kwargs.get("gk_eta", 1.001)
    delta = kwargs.get("gk_delta", 0.001)
    eng = A.engine
    m, n = A.shape
    bv = eng.to_vec(b, m)
    P, Y, R = eng.scalars(n_iter + 2), eng.scalars(n_iter + 1), eng.scalars(1)
    x, ax = eng.empty(n), eng.empty(m)
    res_norm = np.inf
    for _ in range(n_iter):
        if res_norm <= eta * delta:
            print("discrepancy principle satisfied, halting early.")
            break
        gk.step()
        k = gk.V.k
        eng.gemv_t(gk.U.data, k + 1, bv, P.ref(0))                  # bhat = U.T @ b (:189)
        eng.allreduce(P, 0, k + 1)
        y = np.linalg.lstsq(gk.B(), P.host(0, k + 1).reshape(-1, 1), rcond=None)[0].reshape(-1)
        Y.set(0, y)
        eng.gemv_n(gk.V.data, k, Y.ref(0), x)                       # x = V @ y (:193)
        A.apply(x, out=ax)
        eng.diff_nrm2sq(ax, bv, R.ref(0))                           # ||A x - b|| (:195)
        eng.allreduce(R, 0, 1)
        res_norm = float(np.sqrt(R.host(0, 1)[0]))
    return gk


def golub_kahan(A, b, n_iter, dp_stop=False, **kwargs):
    gk = golub_kahan_device(A, b, n_iter, dp_stop, **kwargs)
    if _fmt_like(b):
        return gk.U.numpy(), gk.B(), gk.V.numpy()
    return gk.U.torch_cols(), gk.B(), gk.V.torch_cols()


def arnoldi_device(A, b, n_iter, dp_stop=False, **kwargs):
    """The reference's `arnoldi` on the device: returns (DeviceBasis Q with k+1 vectors, H (k+1) x k host float64).
    NOTE the reference (unlike arnoldi_update) orthogonalises step ii only against Q[:, :ii] — the newest vector
    Q[:, ii] is skipped and H[ii, ii] stays 0 (decompositions.py:88-94, `range(0, iterations)`).  Reproduced.
    dp_stop (:104-112): after every step y solves (H_k^T H_k) y = Q_k^T b^ with the square top block of H and the
    NORMALISED b^, and the factorisation halts before the next step once ||A Q_k y - b^|| <= gk_eta * gk_delta
    (defaults 1.001 / 0.001 — not the solvers' delta)."""
    A = as_operator(A)
    if A.shape[0] != A.shape[1]:
        raise ValueError("Arnoldi can not be used. The operator is not square")
    eng, n, n_iter = A.engine, A.shape[0], int(n_iter)
    eta, delta = kwargs.get("gk_eta", 1.001), kwargs.get("gk_delta", 0.001)
    Q = DeviceBasis(eng, n, n_iter + 1)
    S = eng.scalars(2 * n_iter + 2)
    H = np.zeros((n_iter + 1, n_iter))
    bv = eng.to_vec(b, n)
    eng.nrm2sq(bv, S.ref(0))
    eng.allreduce(S, 0, 1)
    eng.scale(Coef(1.0, den=S.ref(0), sqrt_den=True), bv, Q.next_slot())
    Q.commit()
    w = eng.empty(n)
    if dp_stop:
        P, Y, R = eng.scalars(n_iter + 1), eng.scalars(n_iter + 1), eng.scalars(1)
        x, ax = eng.empty(n), eng.empty(n)
    res_norm, done = np.inf, 0
    for ii in range(n_iter):
        if dp_stop and res_norm <= eta * delta:
            print("discrepancy principle satisfied, stopping early.")
            break
        A.apply(Q[ii], out=w)
        # literal modified Gram-Schmidt against Q[:, :ii] only: with the newest vector skipped the basis is not
        # orthonormal, so block (classical) Gram-Schmidt would NOT give the same numbers
        for jj in range(ii):
            eng.dot(Q[jj], w, S.ref(1 + jj))
            eng.allreduce(S, 1 + jj, 2 + jj)
            eng.axpby(1.0, w, Coef(-1.0, num=S.ref(1 + jj)), Q[jj], w)
        eng.nrm2sq(w, S.ref(0))
        eng.allreduce(S, 0, 1)
        h = S.host(0, 1 + ii)
        H[:ii, ii] = h[1:1 + ii]
        H[ii + 1, ii] = np.sqrt(h[0])
        done = ii + 1
        if H[ii + 1, ii] == 0:
            break
        eng.scale(Coef(1.0, den=S.ref(0), sqrt_den=True), w, Q.next_slot())
        Q.commit()
        if dp_stop:
            k = ii + 1
            eng.gemv_t(Q.data, k, Q[0], P.ref(0))                     # bhat = Q[:, :-1].T @ (b / ||b||)   (:106)
            eng.allreduce(P, 0, k)
            Hk = H[:k, :k]
            y = np.linalg.lstsq(Hk.T @ Hk, P.host(0, k), rcond=None)[0]   # (:108)
            Y.set(0, y)
            eng.gemv_n(Q.data, k, Y.ref(0), x)                        # x = Q[:, :-1] @ y                  (:110)
            A.apply(x, out=ax)
            eng.diff_nrm2sq(ax, Q[0], R.ref(0))                       # ||A x - b^||                       (:112)
            eng.allreduce(R, 0, 1)
            res_norm = float(np.sqrt(R.host(0, 1)[0]))
    return Q, H[:done + 1, :done]


def arnoldi(A, b, n_iter, dp_stop=False, **kwargs):
    Q, H = arnoldi_device(A, b, n_iter, dp_stop, **{k_: v_ for k_, v_ in kwargs.items() if k_ in ("gk_eta", "gk_delta")})
    return (Q.numpy() if _fmt_like(b) else Q.torch_cols()), H


class KrylovArrays(np.ndarray):
    """A NumPy array (the reference's n x k layout) that remembers the device state it was downloaded from, so that a
    chain of *_update calls keeps working on the GPU."""

    def __new__(cls, arr, state):
        obj = np.asarray(arr).view(cls)
        obj._trk_state = state
        return obj

    def __array_finalize__(self, obj):
        self._trk_state = getattr(obj, "_trk_state", None)


def golub_kahan_update(A, U, S, V):
    A = as_operator(A)
    st = getattr(U, "_trk_state", None)
    if not isinstance(st, GKState) or st.A is not A or st.V.k != (0 if np.ndim(S) < 2 else np.shape(S)[1]):
        # cold start from the reference's arrays: upload the vectors and the bidiagonal
        U = np.asarray(U, dtype=np.float64)
        k = 0 if np.ndim(S) < 2 else np.shape(S)[1]
        Sm = np.asarray(S, dtype=np.float64)
        st = GKState.resume(A, [U[:, j] for j in range(U.shape[1])], [np.asarray(V)[:, j] for j in range(k)],
                            [] if k == 0 else list(np.diag(Sm[:k, :k])), [] if k == 0 else list(np.diag(Sm[1:, :k])))
    st.step()
    return KrylovArrays(st.U.numpy(), st), st.B(), KrylovArrays(st.V.numpy(), st)


def arnoldi_update(A, V, H):
    A = as_operator(A)
    st = getattr(V, "_trk_state", None)
    if not isinstance(st, ArnoldiState) or st.A is not A or st.V.k != np.shape(V)[1]:
        Vn = np.asarray(V, dtype=np.float64)
        st = ArnoldiState.__new__(ArnoldiState)
        st.A, st.eng = A, A.engine
        eng, n = A.engine, A.shape[0]
        st.V = DeviceBasis(eng, n, Vn.shape[1] + 1)
        for j in range(Vn.shape[1]):
            st.V.next_slot().copy_(eng.to_vec(Vn[:, j], n))
            st.V.commit()
        st.w, st.S = eng.empty(n), eng.scalars(2 * (Vn.shape[1] + 1) + 2)
        Hm = np.asarray(H, dtype=np.float64)
        st.Hcols = [] if Hm.ndim < 2 else [Hm[:j + 2, j].copy() for j in range(Hm.shape[1])]
        st.beta0 = None
        st.gram, st.capacity, st.by_gram = None, None, False   # a cold start from host arrays: sweep by sweep (the basis may grow)
    st.step()
    return KrylovArrays(st.V.numpy(), st), st.H()
