"""Discrepancy principle on projected factors (reference: trips/utilities/reg_param/discrepancy_principle.py:19-99,
dptype='tikhonov')."""
import ctypes

import numpy as np
import scipy.linalg as sla


def _newton(sv, bhat, target, extra):
    """The loop of discrepancy_principle.py:80-99 in libtrk.so (trk_host_dp_newton): ~16 steps of k-sized vector work,
    ~10 us there against ~150 us as NumPy calls."""
    import ctypes
    from .. import _lib
    lib = _lib.load()
    sv = np.ascontiguousarray(sv, dtype=np.float64).reshape(-1)
    bhat = np.ascontiguousarray(bhat, dtype=np.float64).reshape(-1)
    alpha, have = ctypes.c_double(0.0), ctypes.c_int(0)
    _lib.check(lib.trk_host_dp_newton(sv.ctypes.data, bhat.ctypes.data, int(sv.size), float(target), float(extra),
                                      ctypes.byref(alpha), ctypes.byref(have), None), "trk_host_dp_newton")
    return alpha.value if have.value else None


def discrepancy_principle_bidiag(alphas, betas, bproj, delta=None, eta=1.01, resid2=0.0, explicitProj=False, **_ignored):
    """`discrepancy_principle` for the Golub-Kahan projected problem (B_k lower bidiagonal: diagonal `alphas`, sub-diagonal `betas`;
    bproj = U^T b with k+1 entries; L = I) without the SVD of B_k (trk_host_dp_bidiag): same Newton iteration, same branches."""
    if not isinstance(delta, (float, int)):
        raise Exception("A value for the noise level delta was not provided and the discrepancy principle cannot be applied. "
                        "Please supply a value of delta based on the estimated noise level of the problem, or choose the "
                        "regularization parameter according to gcv.")
    al = np.ascontiguousarray(alphas, dtype=np.float64)
    be = np.ascontiguousarray(betas, dtype=np.float64)
    bp = np.ascontiguousarray(np.asarray(bproj, dtype=np.float64).reshape(-1))
    if al.size < 1 or be.size != al.size or bp.size != al.size + 1:
        raise ValueError("discrepancy_principle_bidiag: need k diagonal, k sub-diagonal and k + 1 projected entries")
    from .. import _lib
    alpha, have = ctypes.c_double(0.0), ctypes.c_int(0)
    _lib.check(_lib.load().trk_host_dp_bidiag(al.ctypes.data, be.ctypes.data, int(al.size), bp.ctypes.data, float((eta * delta) ** 2),
                                              float(resid2 if explicitProj else 0.0), ctypes.byref(alpha), ctypes.byref(have),
                                              None, None), "trk_host_dp_bidiag")
    return alpha.value if have.value else None


def _null_space_branch(A, L, bp, W):
    """discrepancy_principle.py:47-66: A restricted to the A-weighted pseudo-inverse of L, b without its component along the null
    space W of L (mirrored literally, `np.linalg.inv` of the triangular factors included)."""
    AW = A @ W
    Q_AW, R_AW = np.linalg.qr(AW, mode="reduced")
    Q_LT, R_LT = np.linalg.qr(L.T, mode="reduced")
    LAwpinv = (np.eye(L.shape[1]) - (W @ np.linalg.inv(R_AW) @ Q_AW.T @ A)) @ Q_LT @ np.linalg.inv(R_LT.T)
    xnull = W @ np.linalg.inv(R_AW) @ Q_AW.T @ bp
    return A @ LAwpinv, bp - A @ xnull


def _truncation_index(bproj, n, target, dptype):
    """dptype = 'tsvd' / 'tgsvd' (discrepancy_principle.py:100-129): the truncation index of the direct solvers' filter, from
    bhat = Q^T b (Q square there) — the two loops as written, quirks included (tgsvd returns the LAST index that still
    satisfies the discrepancy, counted from the other end)."""
    bhat = np.asarray(bproj, dtype=np.float64).reshape(-1, 1)
    m = bhat.shape[0]
    alpha = n
    if dptype == "tsvd":
        f = np.ones((m, 1))
        for i in range(n):
            f[n - (i + 1), ] = 0
            fvar = np.concatenate((1 - f[:n, ], f[n:, ]))
            if np.sum((fvar * bhat) ** 2) - target < 0:
                alpha = n - (i + 1)
            else:
                break
        return alpha
    coeff = np.square(bhat)
    for i in range(n):
        coeff[n - (i + 1), ] = 0
        if np.sum(coeff) - target >= 0:
            alpha = i
        else:
            break
    return alpha


def discrepancy_principle(A, L, bproj, resid2, delta=None, eta=1.01, L_is_identity=False, explicitProj=False,
                          spectrum=None, dptype="tikhonov", **_ignored):
    """alpha with ||A x_alpha - b||^2 = (eta*delta)^2 by the reference's Newton iteration on beta = 1/alpha.

    A       projected operator: B_k ((k+1) x k, hybrid solvers) or R_A (k x k, GKS / MMGKS)
    L       projected regulariser (ignored when L_is_identity)
    bproj   Q^T b  (k+1 or k entries)
    resid2  ||b - Q Q^T b||^2  (used where the reference uses it: square `A`, or explicitProj)
    spectrum (engine-only) = (S, U^T bproj, (rows, cols)) of the matrix the reference would decompose, when the caller
              has it cheaper than a dense SVD (Hybrid_LSQR: the bidiagonal B_k); A and L are then not looked at
    dptype  'tikhonov' (default: the iterative solvers'), 'tsvd' / 'tgsvd' (the direct solvers' truncation index, :100-129; `bproj`
            is then Q^T b for the square Q, `L` only gives the column count)
    Returns 0 when the discrepancy cannot be reached yet (:76 `testzero >= 0` branch)."""
    if not isinstance(delta, (float, int)):
        raise Exception("A value for the noise level delta was not provided and the discrepancy principle cannot be applied. "
                        "Please supply a value of delta based on the estimated noise level of the problem, or choose the "
                        "regularization parameter according to gcv.")
    if dptype in ("tsvd", "tgsvd"):
        return _truncation_index(bproj, np.shape(L)[1], (eta * delta) ** 2, dptype)
    if dptype != "tikhonov":
        raise UnboundLocalError(f"dptype={dptype!r}: the reference leaves alpha unassigned (discrepancy_principle.py:131)")
    if spectrum is not None:
        S, bhat, (r, c) = spectrum
        S, bhat = np.asarray(S, dtype=np.float64), np.asarray(bhat, dtype=np.float64).reshape(-1, 1)
    else:
        A = np.asarray(A, dtype=np.float64)
        bp = np.asarray(bproj, dtype=np.float64).reshape(-1, 1)
        if L_is_identity:
            Anew = A
        else:
            L = np.asarray(L, dtype=np.float64)
            _, SL, VL = sla.svd(L)
            if L.shape[0] >= L.shape[1] and SL[-1] != 0:
                Anew = A @ (VL.T @ np.diag(SL ** (-1.0)))
            elif L.shape[0] >= L.shape[1]:
                # an exactly zero singular value (:45-55): mirrored as written — like the reference it ends in numpy's LinAlgError,
                # since R_LT of a rank-deficient (or tall) L is singular (not square)
                W = VL[np.where(SL == 0), :].reshape((-1, 1))
                Anew, bp = _null_space_branch(A, L, bp, W)
            else:                                                       # fewer rows than columns (:56-66)
                W = VL[L.shape[0] - L.shape[1]:, :].T
                Anew, bp = _null_space_branch(A, L, bp, W)
        U, S, _ = sla.svd(Anew)
        bhat = U.T @ bp
        r, c = Anew.shape
    sv = (S ** 2)
    target = (eta * delta) ** 2
    if r > c:
        sv = np.append(sv, np.zeros(r - c))
        testzero = np.linalg.norm(bhat[c - r:, :]) ** 2 - target + (resid2 if explicitProj else 0.0)
    else:
        testzero = resid2 - target
    if not testzero < 0:
        return 0
    return _newton(sv, bhat, target, resid2 if explicitProj else 0.0)
