"""Hybrid LSQR on the HIP engine — signature, iteration structure and `info` of trips/solvers/Hybrid_LSQR.py:25-114.

Device: Golub-Kahan steps (2 operator applies + fused axpby/norm kernels), the projected Tikhonov solve on the
bidiagonal (trk_bidiag_tikhonov), x = V y (one tall-skinny GEMV over the row-per-vector basis), ||x - x_true||.
Host (float64, k-sized): lambda selection only.  With a numeric `regparam` nothing is read back inside the loop: the
whole solve is enqueued asynchronously.
"""
import numpy as np
import scipy.linalg as sla

from .. import _trace
from .._io import Formatter, History, as_operator
from ..krylov import GKState
from ..reg_param._bidiag import bidiag_svd_first_row, bidiag_svd_project
from ..reg_param.discrepancy_principle import discrepancy_principle, discrepancy_principle_bidiag
from ..reg_param.gcv import fminbound_gcv_diag, fminbound_gcv_bidiag
from ._common import check_delta, choose_lambda, small_host_blas


import atexit
import threading

_SEARCHER_LOCK = threading.Lock()


class _Searcher:
    """The library's worker thread for the lambda searches (trk_host_worker_*): post copies B_k's entries and returns, collect
    waits for the result.  One per solve."""

    def __init__(self, lib):
        import ctypes
        self._ct, self.lib = ctypes, lib
        self.h = ctypes.c_void_p()
        if lib.trk_host_worker_create(ctypes.byref(self.h)) != 0:
            raise RuntimeError("trk_host_worker_create failed")

    def post_gcv(self, alphas, betas, beta0, m_eff, x1=1e-9, x2=1e2, xtol=1e-12, maxfun=1000):
        al = np.ascontiguousarray(alphas, dtype=np.float64)
        be = np.ascontiguousarray(betas, dtype=np.float64)
        if self.lib.trk_host_worker_post_gcv_bidiag(self.h, al.ctypes.data, be.ctypes.data, int(al.size), float(beta0), float(m_eff),
                                                    float(x1), float(x2), float(xtol), int(maxfun)) != 0:
            raise RuntimeError("trk_host_worker_post_gcv_bidiag failed")

    def post_dp(self, alphas, betas, bproj, delta, eta=1.01):
        al = np.ascontiguousarray(alphas, dtype=np.float64)
        be = np.ascontiguousarray(betas, dtype=np.float64)
        bp = np.ascontiguousarray(np.asarray(bproj, dtype=np.float64).reshape(-1))
        if self.lib.trk_host_worker_post_dp_bidiag(self.h, al.ctypes.data, be.ctypes.data, int(al.size), bp.ctypes.data,
                                                   float((eta * delta) ** 2), 0.0) != 0:
            raise RuntimeError("trk_host_worker_post_dp_bidiag failed")

    def post_hess_gcv(self, H, beta0, k, x1=1e-9, x2=1e2, xtol=1e-12, maxfun=1000):
        """Hybrid-GMRES: the whole projected problem of iterate k (bidiagonalisation, GCV, Tikhonov solve, back-transformation) as one
        job; H: a float64 array view of the (k+1) x k Hessenberg matrix (any strides: copied by the library before this returns)."""
        self._need_lapack()
        if H.dtype != np.float64 or H.shape != (k + 1, k):
            raise ValueError("post_hess_gcv: H must be a float64 (k+1) x k array")
        if self.lib.trk_host_worker_post_hess_gcv(self.h, H.ctypes.data, H.strides[0] // 8, H.strides[1] // 8, int(k), float(beta0),
                                                  float(k), float(x1), float(x2), float(xtol), int(maxfun)) != 0:
            raise RuntimeError("trk_host_worker_post_hess_gcv failed")

    def _need_lapack(self):
        if not getattr(self, "_lapack", False):
            from ..reg_param._bidiag import lapack_pointers
            ptrs = lapack_pointers()
            if ptrs is None or self.lib.trk_host_worker_set_lapack(self.h, ptrs[0], ptrs[1]) != 0:
                raise RuntimeError("trk_host_worker_set_lapack failed")
            self._lapack = True

    def post_hess_dp(self, H, beta0, k, bproj, delta, eta=1.01):
        """The same job with the discrepancy principle: bproj = V_{k+1}^T b (k + 1 values, copied)."""
        self._need_lapack()
        bp = np.ascontiguousarray(np.asarray(bproj, dtype=np.float64).reshape(-1))
        if H.dtype != np.float64 or H.shape != (k + 1, k) or bp.size != k + 1:
            raise ValueError("post_hess_dp: H must be a float64 (k+1) x k array, bproj k + 1 values")
        if self.lib.trk_host_worker_post_hess_dp(self.h, H.ctypes.data, H.strides[0] // 8, H.strides[1] // 8, int(k), float(beta0),
                                                 bp.ctypes.data, float((eta * delta) ** 2), 0.0) != 0:
            raise RuntimeError("trk_host_worker_post_hess_dp failed")

    def collect_vec(self, k):
        """(lambda or None, y, relResidual); y and the residual are meaningful only for a positive lambda."""
        lam, have, r = self._ct.c_double(0.0), self._ct.c_int(0), self._ct.c_double(0.0)
        y = np.empty(k, dtype=np.float64)
        if self.lib.trk_host_worker_collect_vec(self.h, self._ct.byref(lam), self._ct.byref(have), y.ctypes.data, int(k),
                                                self._ct.byref(r)) != 0:
            raise RuntimeError("the projected problem failed on the worker thread")
        return (lam.value if have.value else None), y, r.value

    def collect(self):
        lam, have = self._ct.c_double(0.0), self._ct.c_int(0)
        if self.lib.trk_host_worker_collect(self.h, self._ct.byref(lam), self._ct.byref(have)) != 0:
            raise RuntimeError("lambda search failed on the worker thread")
        return lam.value if have.value else None

    def close(self):
        h, self.h = self.h, None
        if h:
            self.lib.trk_host_worker_destroy(h)

    # a solve borrows a worker and hands it back: creating and joining a thread per solve was ~0.1 ms of a 7 ms solve
    _idle = []
    _idle_max = 4

    @classmethod
    def borrow(cls, lib):
        with _SEARCHER_LOCK:
            while cls._idle:
                w = cls._idle.pop()
                if w.h and w.lib is lib:
                    return w
        return cls(lib)

    def give_back(self):
        """After a clean solve (nothing posted and not collected); anything else: close()."""
        with _SEARCHER_LOCK:
            keep = bool(self.h) and len(_Searcher._idle) < _Searcher._idle_max
            if keep:
                _Searcher._idle.append(self)
        if not keep:
            self.close()


@small_host_blas(when=lambda rp: rp == "l_curve")
def Hybrid_LSQR(A, b, n_iter=100, regparam="gcv", x_true=None, **kwargs):
    """Returns (x, info); info keys: xHistory (n_iter-1 iterates: none is formed at the first step, :77-78), regParam,
    regParam_history, relError (if x_true), relResidual (empty list, as in the reference), its (= n_iter-1).
    Engine-only kwarg: history (True, False, a stride, 'host' or a .npy path: _io.History)."""
    A = as_operator(A)
    delta = check_delta(regparam, kwargs)
    if kwargs.get("dtype") is not None and np.dtype(kwargs["dtype"]) != np.dtype("float32"):
        return _hybrid_lsqr_float64(A, b, int(n_iter), regparam, x_true, kwargs)
    if kwargs.get("dp_stop", False):
        # the reference's dp_stop branch multiplies V[:, :-1] (k-1 columns) by a k-vector and raises (:87-93 / :60-66)
        raise NotImplementedError("dp_stop=True: the reference branch is shape-inconsistent; not reproduced")
    eng = A.engine
    m, n = A.shape
    n_iter = int(n_iter)
    fmt = Formatter(b)
    xt = None if x_true is None else eng.to_vec(x_true, n)

    gk = GKState(A, b, n_iter, normalized=False)     # U[j] = beta_j u_j, V[j] = alpha_j v_j (krylov.GKState)
    bv = eng.to_vec(b, m) if (isinstance(regparam, str) and regparam == "dp") else None
    H = History(eng, kwargs.get("history", True), max(1, n_iter - 1), n, "Hybrid_LSQR xHistory")
    keep = H.keeps_any
    Y = eng.scalars(max(1, n_iter))          # projected solution
    W = eng.scalars(3 * (n_iter + 1) + 4)    # rotation state of the projected solve, resumed while lambda stays the same
    E = eng.scalars(max(1, n_iter) + 1)      # E[0] = ||x_true||^2, E[i] = ||x_i - x_true||^2
    P = eng.scalars(n_iter + 2)              # U^T b for the discrepancy principle
    if xt is not None:
        eng.nrm2sq(xt, E.ref(0))
        eng.allreduce(E, 0, 1)
    # ||x_i - x_true||^2 as raw block partials of the kernel that forms x_i, summed once after the loop
    err_fused = xt is not None and hasattr(eng, "gemv_n_err")
    host_y = (isinstance(regparam, str) and hasattr(eng, "gemv_n_hosty") and kwargs.get("host_projected_solve", True)
              and (xt is None or err_fused))
    EP = eng.scalars(1024 * max(1, n_iter)) if err_fused else None
    n_ep = 0

    lams, lam, nx_done, x_dev = [], 0, 0, None
    on_host = isinstance(regparam, str)          # lambda selection needs B_k on the host
    # fixed lambda with every iterate formed: x_k = V_k y_k is the damped-LSQR iterate, which follows from x_{k-1} by Paige &
    # Saunders' short recurrence (an identity in B_k, whatever the orthogonality of V): one pass over three vectors per iterate
    # instead of the projected solve plus a k-term combination (4 k n bytes) — and nothing reads AB, so no norm is flushed
    recur = (not on_host and hasattr(eng, "lsqr_damped_update") and kwargs.get("x_by_recurrence", True)
             and (keep or xt is not None) and n_iter >= 2 and float(regparam) >= 0.0)
    if recur:
        lsqr_w, lsqr_x1, lsqr_st = eng.empty(n), eng.empty(n), eng.scalars(8)
        x_prev = None
        fuse_update = bool(kwargs.get("update_on_the_step", True)) and gk.native_axpby and hasattr(eng, "gk_step_lsqr")
    ub_vec = bv if (isinstance(regparam, str) and regparam == "dp") else None   # the discrepancy principle wants U^T b
    # The Golub-Kahan steps do not depend on lambda: they are enqueued `ahead` steps in front of the iterate the host is choosing
    # lambda for, each followed by the download of its two norms.  One step ahead, the device idled every iteration between the
    # kernels of x_k and the enqueue of step k+2, which waited for step k+1's norms (gcv: 115 us per iteration for 75 us of kernels).
    ahead = max(1, int(kwargs.get("steps_ahead", 3)))
    pending, n_enq = [], 0
    while on_host and n_enq < n_iter and len(pending) < ahead:
        pending.extend(gk.step_prefetch(project=ub_vec, more_follow=n_enq + 1 < n_iter))     # (a step's download starts behind the
        n_enq += 1                                                                           #  next step: its last norm rides there)
    # gcv / dp: the search for lambda_k runs on a worker thread of the library (trk_host_worker_*) while this thread enqueues the
    # next step; the iterate of step k is formed one trip later, when its lambda is collected.  (The searches were 20-45 us of
    # the ~100 us this loop spent per iteration at 512^2 x 180 — more than the device needs for an iteration's kernels.)
    searcher = None
    if (on_host and regparam in ("gcv", "dp") and kwargs.get("async_search", True) and hasattr(eng, "lib")
            and not kwargs.get("gcv_by_svd", False) and not kwargs.get("dp_by_svd", False) and n_iter > 2):
        searcher = _Searcher.borrow(eng.lib)
    waiting = None                           # the step whose lambda the worker is looking for
    clean = False
    # B_k's entries (and U^T b in the left basis' scale) as float64 ARRAYS grown in place: the calls below take views of them — the lists
    # of krylov.GKState converted to arrays four times per iteration were O(k) interpreter work each (20 us of a 70 us iteration at k = 100)
    al_np, be_np, bp_np = np.empty(n_iter + 1), np.empty(n_iter + 1), np.empty(n_iter + 2)
    n_ab, n_bp = 0, 0
    # kwarg one_call (default on): the host's turn of an iteration as one library call (trk_hlsqr_select)
    one_call = (searcher is not None and host_y and kwargs.get("one_call", True) and hasattr(getattr(eng, "lib", None), "trk_hlsqr_select")
                and (regparam == "gcv" or kwargs.get("delta") is not None))
    if one_call:
        import ctypes as ct
        from .. import _lib
        c_nblk, c_lam, c_have = ct.c_int(0), ct.c_double(0.0), ct.c_int(0)
        dp_target = float((kwargs.get("eta", 1.01) * kwargs["delta"]) ** 2) if regparam == "dp" else 0.0

    def grow_host_arrays():
        nonlocal n_ab, n_bp
        while n_ab < len(gk._alphas):
            al_np[n_ab], be_np[n_ab] = gk._alphas[n_ab], gk._betas[n_ab]
            n_ab += 1
        if gk.uproj is not None:
            while n_bp < len(gk.uproj) and n_bp <= n_ab:                 # rows of U are beta_j u_j (beta_0 := beta0)
                bp_np[n_bp] = gk.uproj[n_bp] / (gk.beta0 if n_bp == 0 else be_np[n_bp - 1])
                n_bp += 1

    def form_iterate(k, lam):
        nonlocal nx_done, n_ep, x_dev
        lams.append(lam)
        if not keep and xt is None and k < n_iter:
            return                           # nobody looks at this iterate (history=False, no x_true)
        # y = lstsq([B; sqrt(lam) I], [beta0 e1; 0]) (:104), on the device from the squared norms in gk.AB
        # (as y_j / alpha_j: the rows of V are alpha_j v_j)
        if host_y:
            # automatic lambda: B_k is on the host already (lambda_k was chosen from it), and so is the projected solve — O(k) on a
            # CPU core, where one GPU lane spent 5-25 us on the dependent square roots and divisions of a recurrence that restarts
            # whenever lambda moves; the k coefficients reach the device in the arguments of the launch that forms x_k
            yk = eng.host_bidiag_tikhonov(al_np[:k], be_np[:k], gk.beta0, np.sqrt(lam), y_over_alpha=True)
            x_dev = H.row(nx_done)
            if xt is not None:
                n_ep = eng.gemv_n_hosty(gk.V.data, k, yk, x_dev, xt, EP.ref(n_ep * nx_done), 1024)
            else:
                eng.gemv_n_hosty(gk.V.data, k, yk, x_dev)
            H.pushed(nx_done)
            nx_done += 1
            return
        if not on_host:
            gk.flush()                       # (automatic lambda: the norms of step k were downloaded, hence finished, before this)
        eng.bidiag_tikhonov(gk.AB.ref(1), 2, gk.AB.ref(2), 2, k, np.sqrt(lam), gk.AB.ref(0), Y.ref(0), W, y_over_alpha=True)
        x_dev = H.row(nx_done)
        if err_fused:
            n_ep = eng.gemv_n_err(gk.V.data, k, Y.ref(0), x_dev, xt, EP.ref(n_ep * nx_done), 1024)
        else:
            eng.gemv_n(gk.V.data, k, Y.ref(0), x_dev)
        H.pushed(nx_done)
        nx_done += 1
        if xt is not None and not err_fused:
            eng.diff_nrm2sq(x_dev, xt, E.ref(nx_done))

    try:
        for ii in _trace.progress(range(n_iter), "running Golub-Kahan bidiagonalization algorithm...", kwargs.get("progress")):   # (Hybrid_LSQR.py:73)
            _trace.mark("Hybrid_LSQR: Golub-Kahan step, projected problem, iterate")
            k = ii + 1
            rode = False
            if on_host:
                gk.absorb(pending.pop(0))        # alpha_k, beta_{k+1}; the steps behind it run while the host chooses lambda_k
                while n_enq < n_iter and len(pending) < ahead:
                    pending.extend(gk.step_prefetch(project=ub_vec, more_follow=n_enq + 1 < n_iter))
                    n_enq += 1
                grow_host_arrays()
            else:
                # fixed lambda: nothing on the host needs B_k.  Iterates that nobody looks at are not formed and the step's last norm
                # stays inside the operator (defer); when every iterate IS formed, step k+1 is enqueued BEFORE x_k — its adjoint kernel
                # finishes beta_{k+1}^2 on the way (GKState.step(defer=True)), which the projected solve of x_k reads — so that no
                # reduction launch is spent on it
                if gk.V.k < k:
                    gk.step(sync=False, defer=True)
                rode = False
                if k < n_iter and (keep or xt is not None):
                    # step k+1 ahead of x_k — and x_k's own update on that step's adjoint half, whose second operand is the V[k-1] the
                    # update needs (trk_gk_step_lsqr: the iteration is three launches on the projector)
                    req = None
                    if recur and fuse_update:
                        x_dev = lsqr_x1 if ii == 0 else H.row(nx_done)
                        req = (lsqr_w, x_prev, x_dev, None if ii == 0 else xt,
                               None if (ii == 0 or xt is None) else EP.ref(1024 * nx_done), 1024, np.sqrt(float(regparam)),
                               lsqr_st.ref(4 * ((ii + 1) & 1)), lsqr_st.ref(4 * (ii & 1)))
                    gk.step(sync=False, defer=True, lsqr=req)
                    rode = req is not None and gk.lsqr_taken
                    if rode:
                        got = gk.lsqr_blocks
            if recur and not rode:
                # step k of the recurrence (the reference reports no iterate for k = 1, the recurrence needs it all the same);
                # alpha_k^2 = AB[2k-1] and beta_{k+1}^2 = AB[2k] are final: the step enqueued ahead finished the latter
                if k == n_iter:
                    gk.flush()                       # no step was enqueued ahead of the last iterate
                x_dev = lsqr_x1 if ii == 0 else H.row(nx_done)
                got = eng.lsqr_damped_update(gk.V[ii], lsqr_w, x_prev, x_dev, gk.AB.ref(2 * k - 1), gk.AB.ref(2 * k), gk.AB.ref(0),
                                             np.sqrt(float(regparam)), lsqr_st.ref(4 * ((ii + 1) & 1)), lsqr_st.ref(4 * (ii & 1)), ii == 0,
                                             ref=None if ii == 0 else xt, partials=None if (ii == 0 or xt is None) else EP.ref(1024 * nx_done),
                                             capacity=1024)
            if recur:
                x_prev = x_dev
                if ii == 0:
                    lam = 0
                    x_dev = None
                    continue
                if xt is not None:
                    n_ep = got
                lam = regparam
                lams.append(lam)
                H.pushed(nx_done)
                nx_done += 1
                continue
            if ii == 0:
                lam = 0
                continue
            if searcher is not None and one_call:
                # collect lambda of the step before, post the search for lambda_k, solve and launch the iterate of the step before: ONE
                # library call (trk_hlsqr_select) for what the branch below does in four
                kd = waiting if waiting is not None else 0
                form = kd > 0 and (keep or xt is not None or kd >= n_iter)
                row = H.row(nx_done) if form else None
                if regparam == "dp" and n_bp < k + 1:
                    raise RuntimeError("Hybrid_LSQR: U^T b is behind the bidiagonal")
                _lib.check(eng.lib.trk_hlsqr_select(
                    searcher.h, 0 if regparam == "gcv" else 1, al_np.ctypes.data, be_np.ctypes.data, k, float(gk.beta0),
                    float(m) if regparam == "gcv" else dp_target, bp_np.ctypes.data, 0.0, kd, gk.V.data.data_ptr(), gk.V.data.stride(0), n,
                    None if row is None else row.data_ptr(), None if (row is None or xt is None) else xt.data_ptr(),
                    None if (row is None or xt is None) else EP.ref(n_ep * nx_done), 1024, ct.byref(c_nblk), ct.byref(c_lam), ct.byref(c_have),
                    eng.stream()), "trk_hlsqr_select")
                if kd > 0:
                    if not c_have.value:
                        raise RuntimeError("Hybrid_LSQR: the search on the worker thread set no lambda")
                    lam = c_lam.value
                    lams.append(lam)
                    if form:
                        x_dev = row
                        if xt is not None:
                            n_ep = c_nblk.value
                        H.pushed(nx_done)
                        nx_done += 1
                waiting = k
                continue
            if searcher is not None:
                if waiting is not None:
                    lam = searcher.collect()
                # post the search for lambda_k first, then form the iterate of the step before it
                if regparam == "gcv":
                    searcher.post_gcv(al_np[:k], be_np[:k], gk.beta0, m)
                else:
                    if n_bp < k + 1:
                        raise RuntimeError("Hybrid_LSQR: U^T b is behind the bidiagonal")
                    searcher.post_dp(al_np[:k], be_np[:k], bp_np[:k + 1], kwargs.get("delta"), kwargs.get("eta", 1.01))
                if waiting is not None:
                    form_iterate(waiting, lam)
                waiting = k
                continue
            if isinstance(regparam, str) and regparam == "gcv":
                # svd(B) (:81) enters GCV through s and Q_A^T bhat = beta0 * (first row of the left vectors) only — and G(lam) is a
                # resolvent of the tridiagonal B B^T: evaluated without the SVD (trk_host_gcv_bidiag); variant 'modified', fullsize = m (:84)
                if kwargs.get("gcv_by_svd", False):
                    s, u0 = bidiag_svd_first_row(gk._alphas[:k], gk._betas[:k])
                    lam = fminbound_gcv_diag(s, gk.beta0 * u0, m)
                else:
                    lam = fminbound_gcv_bidiag(gk._alphas[:k], gk._betas[:k], gk.beta0, m)
            elif isinstance(regparam, str) and regparam == "l_curve":
                bhat = np.zeros(k + 1)
                bhat[0] = gk.beta0
                Qb, s, _ = sla.svd(gk.B(k), full_matrices=False)
                lam = choose_lambda(regparam, np.diag(s), np.eye(k), Qb.T @ bhat, 0.0, kwargs, variant="modified", fullsize=m)
            elif isinstance(regparam, str) and regparam == "dp":
                # discrepancy_principle(U, B, L, b): projects b on the (no longer exactly orthonormal) computed U (:86)
                # U^T b row by row, downloaded with each step's norms (krylov.GKState.step_prefetch): no pass over U, no blocking copy
                bproj = np.asarray(gk.uproj[:k + 1]) / np.concatenate(([gk.beta0], gk._betas[:k]))   # rows of U are beta_j u_j
                extra = {key: kwargs[key] for key in ("eta", "explicitProj") if key in kwargs}
                if kwargs.get("dp_by_svd", False):
                    s, proj = bidiag_svd_project(gk._alphas[:k], gk._betas[:k], bproj)       # svd(B_k), U^T bproj (dp :68-70)
                    lam = discrepancy_principle(None, None, None, 0.0, delta=kwargs.get("delta"), L_is_identity=True,
                                                spectrum=(s, proj, (k + 1, k)), **extra)
                else:                                # the same Newton iteration on the tridiagonal resolvent: no SVD of B_k
                    lam = discrepancy_principle_bidiag(gk._alphas[:k], gk._betas[:k], bproj, delta=kwargs.get("delta"), **extra)
            else:
                lam = regparam
            form_iterate(k, lam)
        if waiting is not None:
            lam = searcher.collect()
            form_iterate(waiting, lam)
        clean = True
    finally:
        _trace.mark(None)
        if searcher is not None:
            if clean:
                searcher.give_back()
            else:
                searcher.close()             # (a job may still be posted: destroy waits for it)
    if x_dev is None:
        raise UnboundLocalError("Hybrid_LSQR with n_iter < 2 forms no iterate (the reference fails the same way, "
                                "Hybrid_LSQR.py:114)")
    info = {"xHistory": H.collect(fmt, nx_done), "regParam": lam, "regParam_history": lams,
            "relResidual": [], "its": n_iter - 1}
    if xt is not None:
        if err_fused:
            # (the recurrence's iterates come from two kernels with different partial counts — the adjoint's tiles or the update's own
            #  grid: each iterate owns 1024 zero-initialised places, all of them are summed)
            eng.finalize_batched(EP.ref(0), 1024 if recur else n_ep, 1, nx_done, E.ref(1), 1)
        eng.allreduce(E, 1, nx_done + 1)
        e = E.host(0, nx_done + 1)
        info["relError"] = list(np.sqrt(e[1:] / e[0]))
    return fmt.vec(x_dev), info


def _hybrid_lsqr_float64(A, b, n_iter, regparam, x_true, kwargs):
    """Hybrid_LSQR(..., dtype='float64'): the float64 INSTRUMENT (csrc/ref64.hip, trk_gk_lsqr_chain) — the engine's own arrangement
    of the solver (Golub-Kahan on unnormalised vectors, the iterate by damped LSQR's short recurrence) with its vectors stored in
    float64 and the projector's weights and sums in float64.  For checking the fast path against, not for speed: a numeric
    `regparam` and a parallel-beam operator only.  kwargs: storage ('float64' | 'float32': the element type of the vectors; the
    arithmetic stays float64), weights ('float64' | 'tables64': see Radon2DParallel.set_arithmetic)."""
    import torch
    from .. import _lib
    from ..operators import Radon2DParallel
    if np.dtype(kwargs["dtype"]) != np.dtype("float64"):
        raise ValueError("Hybrid_LSQR: dtype must be 'float32' (the engine) or 'float64' (the instrument)")
    if isinstance(regparam, str) or not isinstance(A, Radon2DParallel):
        raise NotImplementedError("Hybrid_LSQR(dtype='float64') is the diagnostic chain: numeric regparam, Radon2DParallel operator")
    if n_iter < 2:
        raise UnboundLocalError("Hybrid_LSQR with n_iter < 2 forms no iterate (the reference fails the same way, Hybrid_LSQR.py:114)")
    eng = A.engine
    m, n = A.shape
    tdt = torch.float64 if kwargs.get("storage", "float64") == "float64" else torch.float32
    fmt = Formatter(b)
    bv = torch.as_tensor(np.asarray(b, dtype=np.float64).reshape(-1) if not isinstance(b, torch.Tensor) else b.reshape(-1)).to(device=eng.device, dtype=tdt)
    if bv.numel() != m:
        raise ValueError(f"dimension mismatch: b has {bv.numel()} entries, the operator {m} rows")
    X = torch.empty((n_iter, n), dtype=tdt, device=eng.device)
    work = torch.empty(2 * m + 3 * n, dtype=tdt, device=eng.device)
    AB = torch.zeros(2 * n_iter + 1, dtype=torch.float64, device=eng.device)
    st = torch.zeros(8, dtype=torch.float64, device=eng.device)
    rc = eng.lib.trk_gk_lsqr_chain(A._h, bv.element_size(), {"float64": 0, "tables64": 1}[kwargs.get("weights", "float64")], bv.data_ptr(),
                                   n_iter, float(regparam), X.data_ptr(), work.data_ptr(), AB.data_ptr(), st.data_ptr(), eng.stream())
    _lib.check(rc, "trk_gk_lsqr_chain")
    hist = [X[k].to("cpu").numpy().astype(np.float64).reshape(-1, 1) for k in range(1, n_iter)]   # none at the first step (:77-78)
    info = {"xHistory": hist, "regParam": regparam, "regParam_history": [regparam] * (n_iter - 1), "relResidual": [],
            "its": n_iter - 1, "B_squared": AB.to("cpu").numpy()}
    if x_true is not None:
        xt = (x_true.detach().to("cpu").numpy() if isinstance(x_true, torch.Tensor) else np.asarray(x_true)).astype(np.float64).reshape(-1, 1)
        info["relError"] = [float(np.linalg.norm(h - xt) / np.linalg.norm(xt)) for h in hist]
    return (hist[-1] if fmt.numpy else X[n_iter - 1].clone().reshape(-1, 1)), info


@atexit.register
def _close_idle_searchers():
    with _SEARCHER_LOCK:
        idle, _Searcher._idle = _Searcher._idle, []
    for w in idle:
        try:
            w.close()
        except Exception:      # noqa: BLE001  (interpreter shutdown)
            pass
