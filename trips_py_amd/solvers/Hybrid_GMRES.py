"""Hybrid GMRES on the HIP engine — signature, iteration structure and `info` of trips/solvers/Hybrid_GMRES.py:23-87.

Device: Arnoldi steps (1 operator apply + two passes of block Gram-Schmidt against the whole basis = 2 tall-skinny
GEMV-T / GEMV-N pairs), x = V_k y.  Host: H_k, lambda selection, stacked Tikhonov solve."""
import numpy as np
import scipy.linalg as sla

from .._io import Formatter, History, as_operator
from ..krylov import ArnoldiState, _plain_handle_apply
from ..reg_param._bidiag import HessenbergBidiag, bidiag_tikhonov_host
from ..reg_param.gcv import fminbound_gcv_bidiag
from ._common import check_delta, choose_lambda, tikhonov_lstsq, small_host_blas

# basis size from which Hybrid-GMRES's GCV goes through the bidiagonal form instead of the dense SVD (below it the SVD is the
# cheaper call: 20 us at k = 10 against 25-60 us of ctypes and NumPy overhead around dgebrd)
BIDIAG_FROM_K = 12
# (the same threshold in the one-call-per-iteration loop, kwarg worker_from_k: the two routes evaluate one GCV function a rounding
# apart, and below k ~ 10 its minimum can be flat enough for the two answers to differ visibly — 56 % at one iterate of the 512^2 blur
# with worker_from_k=2 — so the reference's route, the SVD, keeps the small iterates)
WORKER_FROM_K = BIDIAG_FROM_K


@small_host_blas
def Hybrid_GMRES(A, b, n_iter, regparam="gcv", x_true=None, **kwargs):
    """Returns (x, info); info keys: xHistory (n_iter iterates), regParam, regParam_history (first entry 0),
    relError (if x_true), relResidual, its (= n_iter-1).  Engine-only kwarg: history (True, False, a stride, 'host' or a .npy path: _io.History)."""
    A = as_operator(A)
    delta = check_delta(regparam, kwargs)
    if kwargs.get("dp_stop", False):
        # the reference's dp_stop branch multiplies V[:, :-1] (k-1 columns) by a k-vector and raises (:87-93 / :60-66)
        raise NotImplementedError("dp_stop=True: the reference branch is shape-inconsistent; not reproduced")
    eng = A.engine
    m, n = A.shape
    if m != n:
        raise Exception("Please check the size of the matrx A: it should be square in order to apply hybrid GMRES")
    n_iter = int(n_iter)
    fmt = Formatter(b)
    xt = None if x_true is None else eng.to_vec(x_true, n)

    ar = ArnoldiState(A, b, n_iter, by_gram=kwargs.get("gram_sweeps", True))
    bv = eng.to_vec(b, m) if (isinstance(regparam, str) and regparam == "dp") else None
    Hs = History(eng, kwargs.get("history", True), max(1, n_iter), n, "Hybrid_GMRES xHistory")
    Y = eng.scalars(max(1, n_iter))
    E = eng.scalars(max(1, n_iter) + 1)
    P = eng.scalars(n_iter + 2)
    if xt is not None:
        eng.nrm2sq(xt, E.ref(0))
        eng.allreduce(E, 0, 1)
    # ||x_i - x_true||^2 as raw block partials of the kernel that forms x_i, summed once after the loop
    err_fused = xt is not None and hasattr(eng, "gemv_n_err")
    EP = eng.scalars(1024 * max(1, n_iter)) if err_fused else None
    n_ep = 0

    lams, res, lam, x_dev = [], [], 0, None
    # numeric regparam: the projected problem stays on the device (trk_hess_tikhonov appends the new column of H from the
    # sweep's scalars and solves the k x k normal equations) — nothing visits the host inside the loop; H and every y are
    # downloaded once at the end for `relResidual`
    # ... unless the library's one-call-per-iteration loop runs (below: c_loop): then the projected problems are jobs of the host worker
    # threads, with the caller's lambda instead of a search, and the device runs nothing but the Arnoldi steps and x = V y (the
    # ~8 us k_hess_tikhonov left every iteration's critical path: 17-18 k -> 20 k iterations/s on the 512^2 blur)
    c_ok = (kwargs.get("c_loop", True) and n_iter >= 2 and getattr(eng, "world", 1) == 1 and bool(getattr(A, "_h", None))
            and _plain_handle_apply(A) and ar.by_gram and hasattr(eng, "cgs_coeffs") and hasattr(eng, "gemv_n_hosty")
            and hasattr(getattr(eng, "lib", None), "trk_hgmres_create") and ar.V.data.stride(0) >= n
            and (xt is None or err_fused) and HessenbergBidiag.available())
    c_fixed = (c_ok and not isinstance(regparam, str) and float(regparam) >= 0.0 and kwargs.get("device_solve", True)
               and (regparam > 0 or n_iter < eng.GRAM_TIKHONOV_MAX_K))
    on_dev = (not isinstance(regparam, str)) and hasattr(eng, "hess_tikhonov") and hasattr(eng, "cgs_coeffs") \
        and 0 < n_iter and (regparam > 0 or n_iter < eng.GRAM_TIKHONOV_MAX_K) and kwargs.get("device_solve", True) and not c_fixed
    if on_dev:
        kmax = n_iter
        Hd = eng.scalars((kmax + 1) * kmax)          # column-major, column stride kmax + 1
        Gd = eng.scalars(kmax * kmax)
        Mi = eng.scalars(kmax * kmax)                # (H^T H + lam I)^-1, bordered by one row and column per step
        Yall = eng.scalars(kmax * kmax)              # y of iteration ii in row ii
        for ii in range(n_iter):
            k = ar._enqueue()                        # Arnoldi step k = ii + 1: its coefficients sit in ar.S[1 .. 1+2k), ||w||^2 in S[0]
            second = ar.S.ref(1 + k) if ar.gram is None else None          # sweep-by-sweep form: two coefficient sets
            lam = 0 if ii == 0 else regparam                                # (:55-56: the first projected problem is unregularised)
            lams.append(lam)
            # steps 1 (lam = 0) and 2 start the chain, every later step borders the inverse it inherits (same lam)
            mode = 2 if k <= 2 else (1 if lam > 0 else 0)
            eng.hess_tikhonov(Hd.ref(0), kmax + 1, Gd.ref(0), Mi.ref(0), kmax, ar.S.ref(1), second, ar.S.ref(0), ar.beta0, k,
                              lam, mode, Yall.ref(ii * kmax))
            x_dev = Hs.row(ii)
            if err_fused:
                n_ep = eng.gemv_n_err(ar.V.data, k, Yall.ref(ii * kmax), x_dev, xt, EP.ref(n_ep * ii), 1024)
            else:
                eng.gemv_n(ar.V.data, k, Yall.ref(ii * kmax), x_dev)       # x = V[:, :-1] @ y (:77)
                if xt is not None:
                    eng.diff_nrm2sq(x_dev, xt, E.ref(ii + 1))
            Hs.pushed(ii)
        Hh = Hd.host(0, (kmax + 1) * kmax).reshape(kmax, kmax + 1).T         # (kmax+1) x kmax
        Yh = Yall.host(0, kmax * kmax).reshape(kmax, kmax)
        ar.Hcols = [Hh[:j + 2, j].copy() for j in range(n_iter)]
        for ii in range(n_iter):
            k = ii + 1
            bhat = np.zeros(k + 1)
            bhat[0] = ar.beta0
            hy = (Hh[:k + 1, :k] @ Yh[ii, :k]).reshape(-1, 1)
            res.append(float(np.linalg.norm(bhat.reshape(1, -1) - hy)))    # the reference's broadcast quirk (:80), as below
    def projected(k, H, first):
        """Everything the host owes iterate k once H_k is known — lambda_k, y_k, the reference's relResidual — as a function of H_k
        alone.  (Round 5 measured the obvious next step — several k at once on a small thread pool, LAPACK releasing the interpreter
        lock — and it LOSES: 3.3 k iterations/s against 5.2 k on the 512^2 blur, the hand-over of the interpreter lock between the
        pool and the enqueueing thread costs more than the SVDs overlap.  The SVD of a 61 x 60 H is 0.45 ms on a host core, 0.17 ms on
        average over a 60-step solve, against 66 us of kernels per iteration: that is the 5.2 k with gcv against 15.4 k with a number.)"""
        bhat = np.zeros(k + 1)
        bhat[0] = ar.beta0
        svd = None
        if first:
            lam = 0
        elif regparam == "gcv" and k >= BIDIAG_FROM_K and kwargs.get("gcv_by_bidiag", True) and HessenbergBidiag.available():
            # no SVD: [bhat | H] bidiagonalised once (LAPACK dgebrd, a third of the dense SVD's cost), then the O(k)-per-evaluation
            # GCV search and the O(k) Tikhonov solve of the Golub-Kahan form (reg_param/_bidiag.HessenbergBidiag); 'standard' GCV
            # on the k x k diag(s) of (:58) = fullsize k.  Same singular values, same products with bhat: lambda agrees with the
            # SVD route to ~3e-8 at an interior minimum (two evaluations of one smooth function a rounding apart), y to 1e-14.
            hb = HessenbergBidiag(H, ar.beta0)
            lam = fminbound_gcv_bidiag(hb.alphas, hb.betas, hb.beta0, k)
            y = hb.back(bidiag_tikhonov_host(hb.alphas, hb.betas, hb.beta0, np.sqrt(lam)))
            hy = (H @ y).reshape(-1, 1)
            return lam, y, float(np.linalg.norm(bhat.reshape(1, -1) - hy))
        elif regparam in ("gcv", "l_curve"):
            Qh, sv, Vh = sla.svd(H, full_matrices=False, check_finite=False)
            qb = Qh.T @ bhat
            lam = choose_lambda(regparam, np.diag(sv), np.eye(k), qb, 0.0, kwargs)   # 'standard' GCV here (:58)
            svd = (sv, Vh, qb)
        else:
            lam = regparam
        if svd is not None and lam > 0 and kwargs.get("solve_by_svd", True):
            # the Tikhonov minimiser of (:76) from the SVD the selector needed anyway: y = V diag(s / (s^2 + lam)) U^T bhat — O(k^2)
            # where the stacked least-squares problem is another O(k^3) factorisation per iteration
            sv, Vh, qb = svd
            y = Vh.T @ ((sv / (sv * sv + lam)) * qb)
        else:
            y = tikhonov_lstsq(H, np.eye(k), lam, bhat)
        # reference quirk (:80): `bhat - H@y` broadcasts a (k+1,) against a (k+1,1) -> Frobenius norm of a matrix
        hy = (H @ y).reshape(-1, 1)
        return lam, y, float(np.linalg.norm(bhat.reshape(1, -1) - hy))

    def form(ii, lam, y, r):
        nonlocal n_ep, x_dev
        k = ii + 1
        lams.append(lam)
        res.append(r)
        if c_loop:                                 # (the one-call loop: every iterate through the same kernel and partial count)
            yk = np.ascontiguousarray(y, dtype=np.float64).reshape(-1)
            x_dev = Hs.row(ii)
            if err_fused:
                n_ep = eng.gemv_n_hosty(ar.V.data, k, yk, x_dev, xt, EP.ref(n_ep * ii), 1024)
            else:
                eng.gemv_n_hosty(ar.V.data, k, yk, x_dev)
            Hs.pushed(ii)
            return
        Y.set(0, y)
        x_dev = Hs.row(ii)
        if err_fused:
            n_ep = eng.gemv_n_err(ar.V.data, k, Y.ref(0), x_dev, xt, EP.ref(n_ep * ii), 1024)
        else:
            eng.gemv_n(ar.V.data, k, Y.ref(0), x_dev)                  # x = V[:, :-1] @ y (:77)
        Hs.pushed(ii)
        if xt is not None and not err_fused:
            eng.diff_nrm2sq(x_dev, xt, E.ref(ii + 1))

    is_dp = isinstance(regparam, str) and regparam == "dp"
    # gcv through the bidiagonal form: the search itself on the library's worker thread, one iteration behind (kwarg async_search)
    async_gcv = (regparam == "gcv" and not on_dev and kwargs.get("gcv_by_bidiag", True) and kwargs.get("async_search", True)
                 and HessenbergBidiag.available() and n_iter > BIDIAG_FROM_K)
    searcher, waiting = None, None
    if async_gcv:
        from .. import _lib
        from .Hybrid_LSQR import _Searcher
        searcher = _Searcher.borrow(_lib.load())           # (host-only entry points: no GPU involved)

    # dp the same way: V_{k+1}^T b grows by ONE entry per step (the earlier basis vectors do not change): one dot per step, enqueued
    # behind the step and downloaded with it, instead of a sweep over the whole basis and a blocking read per iteration
    dp_async = (is_dp and not on_dev and kwargs.get("dp_by_bidiag", True) and kwargs.get("async_search", True)
                and "explicitProj" not in kwargs and HessenbergBidiag.available() and n_iter > BIDIAG_FROM_K)
    if dp_async and searcher is None:
        from .. import _lib
        from .Hybrid_LSQR import _Searcher
        searcher = _Searcher.borrow(_lib.load())
    Ph = np.zeros(n_iter + 2)
    # gcv, one rank, a plain library operator: the host side of an iteration is ONE library call (trk_hgmres_iter: absorb the step that
    # ran ahead, enqueue the next, collect the worker's answer for the iterate before, post this one, launch x = V y) — kwarg c_loop
    c_dp = c_ok and dp_async and kwargs.get("delta") is not None
    c_loop = (async_gcv and c_ok) or c_fixed or c_dp
    if c_fixed and searcher is None:
        from .. import _lib
        from .Hybrid_LSQR import _Searcher
        searcher = _Searcher.borrow(_lib.load())
    pend = ar.step_prefetch() if (n_iter > 0 and not on_dev and not c_loop) else None

    def enqueue_proj(j0, j1):
        for j in range(j0, j1):
            eng.dot(ar.V[j], bv, P.ref(j))
        eng.allreduce(P, j0, j1)
        return j0, j1, P.host_later(j0, j1)

    ppend = enqueue_proj(0, 2) if (is_dp and pend is not None) else None

    def collect_waiting():
        nonlocal waiting
        jj = waiting
        waiting = None
        return (jj,) + searcher.collect_vec(jj + 1)

    def dp_inline(ii, k, H, Pk):
        """Iterate k with the discrepancy principle in this thread (the first steps, and whatever the worker hands back: the
        reference's unassigned / not-reachable-yet branches)."""
        bhat = np.zeros(k + 1)
        bhat[0] = ar.beta0
        svd = None
        if ii == 0:
            lam = 0
        else:
            hb = None
            if k >= BIDIAG_FROM_K and kwargs.get("dp_by_bidiag", True) and HessenbergBidiag.available() and "explicitProj" not in kwargs:
                # the same Newton iteration on the bidiagonal form of [bhat | H] (as 'gcv' above): V^T b in the left basis'
                # coordinates; lambda agrees with the SVD route to 1e-15 (tests/test_host_regparam.py)
                from ..reg_param.discrepancy_principle import discrepancy_principle_bidiag
                hb = HessenbergBidiag(np.array(H, copy=True), ar.beta0)
                extra = {key: kwargs[key] for key in ("eta",) if key in kwargs}
                lam = discrepancy_principle_bidiag(hb.alphas, hb.betas, hb.left_t(Pk), delta=kwargs.get("delta"), **extra)
                if lam is None or not lam > 0:
                    hb = None
            if hb is not None:
                y = hb.back(bidiag_tikhonov_host(hb.alphas, hb.betas, hb.beta0, np.sqrt(lam)))
                hy = (H @ y).reshape(-1, 1)
                form(ii, lam, y, float(np.linalg.norm(bhat.reshape(1, -1) - hy)))
                return
            if kwargs.get("solve_by_svd", True):
                # the SVD discrepancy_principle() takes of H (discrepancy_principle.py:68-70), taken here so that the
                # Tikhonov solve below can share it
                from ..reg_param.discrepancy_principle import discrepancy_principle
                Uf, sv, Vh = sla.svd(H)
                extra = {key: kwargs[key] for key in ("eta", "explicitProj") if key in kwargs}
                lam = discrepancy_principle(None, None, None, 0.0, delta=kwargs.get("delta"), L_is_identity=True,
                                            spectrum=(sv, Uf.T @ np.array(Pk).reshape(-1, 1), (k + 1, k)), **extra)
                svd = (sv, Vh, Uf[:, :k].T @ bhat)
            else:
                lam = choose_lambda("dp", None, None, None, 0.0, kwargs, L_is_identity=True, dp_A=H, dp_bproj=np.array(Pk))
        if svd is not None and lam > 0 and kwargs.get("solve_by_svd", True):
            sv, Vh, qb = svd
            y = Vh.T @ ((sv / (sv * sv + lam)) * qb)
        else:
            y = tikhonov_lstsq(H, np.eye(k), lam, bhat)
        hy = (H @ y).reshape(-1, 1)
        form(ii, lam, y, float(np.linalg.norm(bhat.reshape(1, -1) - hy)))

    def finish_dp(done):
        jj, lam_w, y, r = done
        if lam_w is not None and lam_w > 0:
            form(jj, lam_w, y, r)
        else:
            dp_inline(jj, jj + 1, ar.H_view()[:jj + 2, :jj + 1], Ph[:jj + 2])

    def host_loop():
        nonlocal pend, ppend, waiting, lam
        for ii in range(0 if not on_dev else n_iter, n_iter):
            k = ii + 1
            ar.absorb(pend)                      # column k of H; step k+1 runs while the host works on the projected problem
            if ppend is not None:
                j0, j1, hnd = ppend
                Ph[j0:j1] = hnd.get()
            pend = ar.step_prefetch() if k < n_iter else None
            ppend = enqueue_proj(k + 1, k + 2) if (is_dp and pend is not None) else None
            H = ar.H_view()
            if is_dp:
                if dp_async and ii > 0 and k >= BIDIAG_FROM_K:
                    done = collect_waiting() if waiting is not None else None
                    searcher.post_hess_dp(H, ar.beta0, k, Ph[:k + 1], kwargs.get("delta"), kwargs.get("eta", 1.01))
                    waiting = ii
                    if done is not None:
                        finish_dp(done)
                    continue
                if waiting is not None:
                    finish_dp(collect_waiting())
                dp_inline(ii, k, H, Ph[:k + 1])
                continue
            if async_gcv and k >= BIDIAG_FROM_K:
                # iterate k's whole projected problem — bidiagonalisation of [bhat | H_k], the GCV search, the Tikhonov solve, y = P' z —
                # is ONE job of the library's worker thread (trk_host_worker_post_hess_gcv), collected one iteration later: this thread
                # only absorbs a step, enqueues the next and launches x_{k-1} = V y_{k-1} while the worker and the device run
                done = collect_waiting() if waiting is not None else None
                searcher.post_hess_gcv(H, ar.beta0, k)           # (H is copied before this returns)
                waiting = ii
                if done is not None:
                    form(*done)
                continue
            if waiting is not None:
                form(*collect_waiting())
            form(ii, *projected(k, H, ii == 0))
        if waiting is not None:
            if is_dp:
                finish_dp(collect_waiting())
            else:
                form(*collect_waiting())

    def c_gcv_loop():
        nonlocal n_ep, x_dev
        import ctypes as ct
        from .. import _lib
        from ..krylov import GramSchmidtByGram
        lib = eng.lib
        searcher._need_lapack()
        ar.gram = GramSchmidtByGram(eng, ar.V, ar.capacity + 1)          # (installs the Gram row of V[0])
        drv = ct.c_void_p()
        # worker threads: the projected problems of consecutive iterates side by side (each O(k^3): ~150 us at k = 60 against ~55 us
        # of kernels per step on the 512^2 blur), collected in order — kwarg search_workers
        nw = max(1, min(16, int(kwargs.get("search_workers", 2))))
        from .Hybrid_LSQR import _Searcher
        while len(more_searchers) < nw - 1:                              # (borrowed like the first one; handed back by the caller)
            more_searchers.append(_Searcher.borrow(lib))
        for sx in more_searchers:
            sx._need_lapack()
        wh = (ct.c_void_p * nw)(searcher.h, *[sx.h for sx in more_searchers])
        cap_mb = 64
        while cap_mb < 2 * (2 * n_iter + 4):
            cap_mb *= 2
        mb, mb_view = eng._mailbox_take(cap_mb, 32)                      # pinned memory from the engine's pool
        try:
            _lib.check(lib.trk_hgmres_create(A._h, ar.V.data.data_ptr(), ar.V.data.stride(0), n_iter, ar.w.data_ptr(), ar.gram.G.ref(0),
                                             ar.gram.kmax, ar.gram.W.ref(0), ar.S.ref(0), mb, wh, nw, float(ar.beta0), eng.stream(),
                                             ct.byref(drv)), "trk_hgmres_create")
        except Exception:
            eng._mailbox_give(cap_mb, mb, mb_view)
            raise
        try:
            Hp, ldh, ncol = ct.POINTER(ct.c_double)(), ct.c_int(), ct.c_int()
            _lib.check(lib.trk_hgmres_hessenberg(drv, ct.byref(Hp), ct.byref(ldh), ct.byref(ncol)), "trk_hgmres_hessenberg")
            Ht = np.ctypeslib.as_array(Hp, shape=(n_iter + 1, ldh.value))      # Ht[j, i] = H[i, j]: the library's array, filled as steps arrive
            d_ii, d_lam, d_res, d_blk = ct.c_int(), ct.c_double(), ct.c_double(), ct.c_int()
            ref = xt.data_ptr() if xt is not None else None
            posted = []                                                  # iterates whose jobs the workers hold, oldest first
            from_k = max(2, int(kwargs.get("worker_from_k", WORKER_FROM_K)))
            if c_fixed:                                                  # a number: no search, no flat minimum — every iterate but the first
                from_k = 2
                _lib.check(lib.trk_hgmres_fixed_lambda(drv, float(regparam)), "trk_hgmres_fixed_lambda")
            bp = None
            if c_dp:
                # V_{k+1}^T b: the first entry here, every further one by the step's own normalising pass (trk_arnoldi_step_post_dot)
                eng.dot(ar.V[0], bv, P.ref(0))
                bpp = ct.POINTER(ct.c_double)()
                _lib.check(lib.trk_hgmres_dp(drv, bv.data_ptr(), float(P.host(0, 1)[0]), float((kwargs.get("eta", 1.01) * kwargs["delta"]) ** 2),
                                             0.0, ct.byref(bpp)), "trk_hgmres_dp")
                bp = np.ctypeslib.as_array(bpp, shape=(n_iter + 2,))
                from_k = max(2, int(kwargs.get("worker_from_k", BIDIAG_FROM_K)))   # (from 2: measured, no gain — the early iterates come back unassigned)

            _lib.check(lib.trk_hgmres_start(drv), "trk_hgmres_start")

            def one(absorb, more, post, collect):
                nonlocal n_ep, x_dev
                jj = posted.pop(0) if collect else None
                row = Hs.row(jj) if collect else None
                ep = EP.ref(n_ep * jj) if (err_fused and collect) else None
                _lib.check(lib.trk_hgmres_iter(drv, absorb, more, post, None if row is None else row.data_ptr(), ref, ep, 1024,
                                               ct.byref(d_ii), ct.byref(d_lam), ct.byref(d_res), ct.byref(d_blk)), "trk_hgmres_iter")
                if collect:
                    if d_ii.value != jj:
                        raise RuntimeError("trk_hgmres_iter: the collected iterate is not the oldest one posted")
                    if d_blk.value < 0:                                  # dp: no positive lambda from the worker — the reference's other branches, here
                        dp_inline(jj, jj + 1, np.ascontiguousarray(Ht[:jj + 1, :jj + 2].T), bp[:jj + 2].copy())
                        return
                    lams.append(d_lam.value)
                    res.append(d_res.value)
                    if err_fused:
                        n_ep = d_blk.value
                    x_dev = row
                    Hs.pushed(jj)

            for ii in range(n_iter):
                k = ii + 1
                post = k >= from_k
                one(1, 1, int(post), post and len(posted) == nw)          # (the library keeps two steps ahead, n_iter in all)
                if post:
                    posted.append(ii)
                if not post:                                             # the first iterates: in this thread (SVD route), as before
                    # (row-major copies: the layout the Python loop hands to LAPACK — the discrepancy principle's lambda is ill-conditioned
                    #  enough at small k for a transposed input's other rounding to show at 1e-6)
                    Hk = np.ascontiguousarray(Ht[:k, :k + 1].T)
                    if c_dp:
                        dp_inline(ii, k, Hk, bp[:k + 1].copy())
                    else:
                        form(ii, *projected(k, Hk, ii == 0))
            while posted:
                one(0, 0, 0, True)
            if "host_phases" in kwargs:                                  # (a list the caller wants the library's phase timers in)
                t5 = (ct.c_double * 5)()
                lib.trk_hgmres_stats(drv, t5)
                kwargs["host_phases"][:] = list(t5)
        finally:
            lib.trk_hgmres_destroy(drv)                                  # (waits for what is still posted: then the mailbox may go back)
            eng._mailbox_give(cap_mb, mb, mb_view)

    more_searchers = []
    clean = False
    try:
        if c_loop:
            c_gcv_loop()
        else:
            host_loop()
        clean = True
    finally:
        for sx in ([searcher] if searcher is not None else []) + more_searchers:
            if clean:
                sx.give_back()
            else:
                sx.close()                   # (a job may still be posted: destroy waits for it)
    if lams:
        lam = lams[-1]
    if x_dev is None:
        raise UnboundLocalError("Hybrid_GMRES with n_iter < 1 forms no iterate")
    info = {"xHistory": Hs.collect(fmt, n_iter), "regParam": lam, "regParam_history": lams,
            "relResidual": res, "its": n_iter - 1}
    if xt is not None:
        if err_fused:
            eng.finalize_batched(EP.ref(0), n_ep, 1, n_iter, E.ref(1), 1)
        eng.allreduce(E, 1, n_iter + 1)
        e = E.host(0, n_iter + 1)
        info["relError"] = list(np.sqrt(e[1:] / e[0]))
    return fmt.vec(x_dev), info
