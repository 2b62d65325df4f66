"""Generalized Krylov Subspace method on the HIP engine — signature and `info` of trips/solvers/GKS.py:27-105.

    min ||A x - b||^2 + lambda ||L x||^2   over   x in span(V_k),   V grown by the normalised normal-equation residual.

What moved where
  * V, AV = A V, LV = L V live on the GPU as row-per-vector bases; a new column is a write (GKS.py:91-96 re-copies all).
  * The from-scratch economic QRs of AV and LV (:54-56) are replaced by their Gram matrices, kept INCREMENTALLY:
    one tall-skinny GEMV-T per new column (k+1 fp64-accumulated dots), then R = chol(G) on the host (k x k).
    Q_A^T b = R_A^{-T} (AV)^T b, ||b - Q_A Q_A^T b||^2 = ||b||^2 - ||Q_A^T b||^2.
  * residual r = A^T(AV y - b) + lambda L^T(LV y) and its 3 re-orthogonalisation passes (:81-88) run on the device with
    device-resident coefficients; the host sees k-sized data only.
"""
import numpy as np

from .. import _trace
from .._io import Formatter, History, as_operator
from ..engine import Coef
from ..decompositions import golub_kahan_device
from ..krylov import DeviceBasis, GramSchmidtByGram, orthogonalize
from ..operators import is_identity
from ._common import check_delta, choose_lambda, gram_factor, gram_gcv_host, project_rhs, tikhonov_lstsq, small_host_blas


import os as _os
_TVDOT = _os.environ.get("TRK_GKS_TVDOT", "1") != "0"       # r . L^T L r from the pass that forms L^T L r (trk_tv_grad_dot)


class _HaloTrack:
    """Time-sharded space-time regulariser: the neighbour ranks' boundary frames ("halo", [prev's last | next's first]) of every
    basis vector.  The fused stencil L^T L x needs them of its operand (operators.SpaceTimeDerivative.tv_grad, trk_tv_halo);
    everything the solver hands to it is a combination of vectors whose halos are known here:
        x = V y            ->  halo(x) = VH y                       (a 2 N^2-long combination, no exchange)
        r (the residual)   ->  ONE two-sided exchange per iteration  (`of_residual`)
        v_new = (r - V c) / rho  ->  halo(v_new) = (halo(r) - VH c) / rho   (`push_from_sweep`)
    so a GKS iteration costs one exchange (GKS.py:81-96; operators.py:39-45 couple frame t with t+1 only)."""

    def __init__(self, L, V, kmax):
        from ..krylov import DeviceBasis
        self.L, self.eng = L, L.engine
        n2 = 2 * L.npix
        self.VH = DeviceBasis(self.eng, n2, kmax)
        self.VH.data.zero_()                      # the half of a rank without that neighbour is never exchanged — and never read
        self.xh, self.rh = self.eng.zeros(n2), self.eng.zeros(n2)
        for j in range(V.k):
            self.push_exchanged(V[j])

    def push_exchanged(self, v):
        self.L.halo_frames(v, out=self.VH.next_slot())
        self.VH.commit()

    def of_iterate(self, k, y):
        self.eng.gemv_n(self.VH.data, k, y, self.xh)
        return self.xh

    def of_residual(self, r):
        return self.L.halo_frames(r, out=self.rh)

    def push_from_sweep(self, k, c, rho2):
        slot = self.VH.next_slot()
        self.eng.gemv_n(self.VH.data, k, c, slot, a=1.0, base=self.rh, s=-1.0)
        self.eng.scale(Coef(1.0, den=rho2, sqrt_den=True), slot, slot)
        self.VH.commit()

    def __getitem__(self, j):
        return self.VH[j]


class _ProjectedBases:
    """V and the incrementally maintained Gram data  G_A = (AV)^T AV, G_L = (LV)^T LV, c = (AV)^T b.

    from_v_A / from_v_L (stencil operators, whose products with the iterate are formed directly from x): the images AV, LV are
    never stored.  G_A[i][j] = v_i . (A^T A v_j), so the new row is V^T z_A with z_A = A^T (A v_new) — one pass over V (n floats
    per vector) instead of one over AV (m) — and likewise G_L from z_L = L^T L v_new instead of a pass over LV (p = 2n .. 3n floats
    per vector); both right-hand sides share ONE sweep over V (trk_gemv_t2), and c[j] = v_j . (A^T b).  Otherwise (the Radon
    projector: m is small and the residual wants (AV) y) the images are kept as row-per-vector bases as before."""

    def __init__(self, A, L, bv, V0, kmax, on_device=False, from_v_A=False, from_v_L=False, use_L=True):
        self.A, self.L, self.eng, self.bv = A, L, A.engine, bv
        eng = self.eng
        # on_device: the Gram data stays on the device (rows installed by a tiny kernel) and nothing is downloaded — the
        # projected problem is solved there too (trk_gram_tikhonov: numeric regparam); else host copies for the selectors
        self.on_device = bool(on_device)
        self.kmax = int(kmax)
        if self.on_device:
            # (one block: an automatic selector downloads G_A, G_L and c with ONE copy per iteration — download_grams)
            from ..engine import DevScalars
            kq = self.kmax * self.kmax
            blk = eng.scalars(2 * kq + self.kmax)
            if isinstance(blk, DevScalars):
                self._gram_block = blk
                self.GA_d, self.GL_d = DevScalars(blk.t[:kq], eng), DevScalars(blk.t[kq:2 * kq], eng)
                self.c_d = DevScalars(blk.t[2 * kq:], eng)
            else:                                   # (another engine's scalar blocks — the CPU test engine: three blocks, three downloads)
                self._gram_block = None
                self.GA_d, self.GL_d, self.c_d = eng.scalars(kq), eng.scalars(kq), eng.scalars(self.kmax)
        m, n = A.shape
        p = L.shape[0]
        self.V = V0
        self.V.reserve(kmax)
        # use_L = False (MMGKS with pnorm = 2: only the A side is unweighted and grows; the caller keeps the weighted L side)
        self.use_L = bool(use_L)
        self.from_v_A, self.from_v_L = bool(from_v_A), bool(from_v_L) and self.use_L
        self.AV = None if self.from_v_A else DeviceBasis(eng, m, kmax)
        self.LV = None if (self.from_v_L or not self.use_L) else DeviceBasis(eng, p, kmax)
        if self.from_v_A:
            self.tA, self.zA, self.atb = eng.empty(m), eng.empty(n), eng.empty(n)
            A.apply(bv, out=self.atb, transpose=True)
        if self.from_v_L:
            self.zL = eng.empty(n)
            self.tL = None if getattr(L, "fused_tv", False) else eng.empty(p)
        self.GA = np.zeros((kmax, kmax))
        self.GL = np.zeros((kmax, kmax))
        self.c = np.zeros(kmax)
        self.S = eng.scalars(2 * kmax + 8)
        # time-sharded fused L: the basis vectors' boundary frames of the neighbour ranks (one exchange per start vector here)
        self.halo = _HaloTrack(L, V0, kmax) if (self.from_v_L and self.tL is None and getattr(L, "sharded", False)) else None
        self.r_halo = None
        for j in range(V0.k):
            self._push_images(j)

    def _push_images(self, j, v_pass=None):
        """Row / column j of the Gram data (and, where they are kept, AV[j] = A V[j], LV[j] = L V[j]).
        v_pass(z, out): a pass over V the caller needs anyway, which also leaves V^T z in `out` (MMGKS: the next iteration's
        re-weighted Gram, trk_wgram_tv_z) — taken instead of the sweep of V for the A-side Gram row (from_v_A without L)."""
        eng, S = self.eng, self.S
        v = self.V[j]
        k = j + 1
        # sharded: c_j travels with the Gram rows (one all-reduce of 2k + 1 doubles, then a device copy into c_d[j])
        c_out = self.c_d.ref(j) if (self.on_device and eng.world == 1) else S.ref(2 * k)
        if self.from_v_A:
            self.A.apply(v, out=self.tA)
            self.A.apply(self.tA, out=self.zA, transpose=True)            # z_A = A^T A v
            if getattr(self, "_c_ready", -1) != j:                        # (else: left by normalise_new, with the scaling pass)
                eng.dot(v, self.atb, c_out)                               # c_j = (A v_j) . b = v_j . (A^T b)
        else:
            av = self.AV.next_slot()
            self.A.apply(v, out=av)
            self.AV.commit()
            # c_j = (A v_j) . b: with the Gram row below when the engine can (b as one more row of that pass), else a dot of its own
            c_with_row = hasattr(eng, "gemv_t_x")
            if not c_with_row:
                eng.dot(av, self.bv, c_out)
        if not self.use_L:
            pass
        elif self.from_v_L:
            if self.tL is None:
                hk = {} if self.halo is None else {"halo": self.halo[j]}
                self.L.tv_grad(v, None, None, 1.0, out=self.zL, **hk)     # z_L = L^T L v in one stencil pass
            else:
                self.L.apply(v, out=self.tL)
                self.L.apply(self.tL, out=self.zL, transpose=True)
        else:
            lv = self.LV.next_slot()
            self.L.apply(v, out=lv)
            self.LV.commit()
        def a_row_from_images():
            if c_with_row:
                eng.gemv_t_x(self.AV.data, k, av, self.bv, S.ref(0), c_out)
            else:
                eng.gemv_t(self.AV.data, k, av, S.ref(0))
        if not self.use_L:
            if self.from_v_A:
                if v_pass is not None:
                    v_pass(self.zA, S.ref(0))
                else:
                    eng.gemv_t(self.V.data, k, self.zA, S.ref(0))
            else:
                a_row_from_images()
        elif self.from_v_A and self.from_v_L:
            eng.gemv_t2(self.V.data, k, self.zA, self.zL, S.ref(0))       # both Gram rows from one sweep over V
        else:
            if self.from_v_A:
                eng.gemv_t(self.V.data, k, self.zA, S.ref(0))
            else:
                a_row_from_images()
            if self.from_v_L:
                eng.gemv_t(self.V.data, k, self.zL, S.ref(k))
            else:
                eng.gemv_t(self.LV.data, k, lv, S.ref(k))
        if self.on_device:
            if eng.world > 1:
                eng.allreduce(S, 0, 2 * k + 1)
                eng.copy_scalars(S, 2 * k, self.c_d, j, 1)
            eng.cgs_coeffs(self.GA_d.ref(0), self.kmax, None, S.ref(0), k, 0, None)     # install row / column j
            if self.use_L:
                eng.cgs_coeffs(self.GL_d.ref(0), self.kmax, None, S.ref(k), k, 0, None)
            return
        eng.allreduce(S, 0, 2 * k + 1)
        h = S.host(0, 2 * k + 1)
        self.GA[j, :k] = self.GA[:k, j] = h[:k]
        self.GL[j, :k] = self.GL[:k, j] = h[k:2 * k]
        self.c[j] = h[2 * k]

    def download_grams(self, k):
        """(G_A[:k, :k], G_L[:k, :k], c[:k]) of the device-resident Gram data as host views of one download (row stride kmax)."""
        kq = self.kmax * self.kmax
        if self._gram_block is None:
            return (self.GA_d.host(0, kq).reshape(self.kmax, self.kmax)[:k, :k], self.GL_d.host(0, kq).reshape(self.kmax, self.kmax)[:k, :k],
                    self.c_d.host(0, k))
        h = self._gram_block.host()
        return (h[:kq].reshape(self.kmax, self.kmax)[:k, :k], h[kq:2 * kq].reshape(self.kmax, self.kmax)[:k, :k], h[2 * kq:2 * kq + k])

    def normalise_new(self, coef, vn):
        """vn <- coef * vn for the basis vector about to be committed (v = r / ||r||).  Where c_j = v_j . (A^T b) is a dot of its
        own (A-side Gram from V, one rank, scalars on the device) it rides this pass (trk_scale_dot): one launch and one read of
        v less per iteration; the same bits as scale followed by dot up to the order of the block sums."""
        eng = self.eng
        j = self.V.k
        if self.from_v_A and self.on_device and eng.world == 1 and hasattr(eng, "scale_dot"):
            eng.scale_dot(coef, vn, vn, self.atb, self.c_d.ref(j))
            self._c_ready = j
        else:
            eng.scale(coef, vn, vn)

    def append(self, v_pass=None):
        """The caller has written the new basis vector into V.next_slot() and committed it."""
        self._push_images(self.V.k - 1, v_pass=v_pass if (self.from_v_A and not self.use_L) else None)
        return v_pass is not None and self.from_v_A and not self.use_L          # whether v_pass ran

    # ---- the Gram rows of the next vector WITHOUT a pass over the basis of their own (on_device only) ----
    # The new vector is v_k = (r - V c)/rho; G[i][k] = v_i . M v_k follows from a = V^T (M r), which rides on the sweep that
    # orthogonalises r (krylov.GramSchmidtByGram.sweep(extra=...), trk_gemv_tn), and from c, r . M r, rho^2
    # (trk_gram_row_from_sweep).  Per iteration GKS then passes over V three times (x = V y, the sweep, r - V c), not four.
    def sweep_operands(self, r, scal=None, off=0, rr=False):
        """The vectors whose V^T products the sweep must also take, and the scalars r . M r (and r . A^T b): call before the sweep.
        scal / off: put the three scalars at scal[off .. off + 3) and leave their sum over ranks to the caller (GKS puts them right
        behind the sweep's own products, so that ONE all-reduce carries both: three exchanges per iteration on ranks, not four).
        rr: also r . r at scal[off + 3] (GKS's one-pass form) — from the stencil pass that forms z_L where it can, else a pass of its own."""
        eng = self.eng
        own = scal is None
        S, o = (self.S, 0) if own else (scal, int(off))
        self._sw = (S, o)
        extra = []
        rr_done = not rr
        if self.from_v_A:
            self.A.apply(r, out=self.tA, sumsq=S.ref(o))                  # ||A r||^2 = r . A^T A r
            self.A.apply(self.tA, out=self.zA, transpose=True)
            eng.dot(r, self.atb, S.ref(o + 2))
            extra.append(self.zA)
        if self.from_v_L:
            hk = {} if self.halo is None else {"halo": self.r_halo}
            if self.tL is None and rr and hasattr(getattr(eng, "lib", None), "trk_tv_grad_dot_xsq") and _TVDOT:
                self.L.tv_grad(r, None, None, 1.0, out=self.zL, dot_with=r, dot_out=S.ref(o + 1), xsq_out=S.ref(o + 3), **hk)
                rr_done = True
            elif self.tL is None and hasattr(getattr(eng, "lib", None), "trk_tv_grad_dot") and _TVDOT:
                self.L.tv_grad(r, None, None, 1.0, out=self.zL, dot_with=r, dot_out=S.ref(o + 1), **hk)   # z_L = L^T L r and r . z_L, one pass
            else:
                if self.tL is None:
                    self.L.tv_grad(r, None, None, 1.0, out=self.zL, **hk)
                else:
                    self.L.apply(r, out=self.tL)
                    self.L.apply(self.tL, out=self.zL, transpose=True)
                eng.dot(r, self.zL, S.ref(o + 1))                         # r . L^T L r
            extra.append(self.zL)
        if not rr_done:
            eng.nrm2sq(r, S.ref(o + 3))
        if own:
            eng.allreduce(S, 0, 4 if rr else 3)
        return extra

    def append_from_sweep(self, gs, k, c, rho2, r_early=None, solve=None):
        """After the sweep over k vectors (coefficients c, rho^2 = ||r - V c||^2) and the commit of v_k: rows k of the Gram data.
        r_early: v_k = (r - V c) / rho is NOT formed yet (GKS's one-pass form, trk_gemv_orth_iterate) — where the images of A are kept,
        A v_k = (A r - AV c) / rho comes from one product with r and the same kernel on the m-length images.
        solve = (lam, Minv, ldm, k_from, Y) (one rank, numeric lambda): the rows AND the projected solve over k + 1 vectors in one launch
        (trk_gks_rows_solve); returns True when it ran."""
        eng, S = self.eng, self.S
        Sw, so = getattr(self, "_sw", (self.S, 0))                          # where sweep_operands left r . M r
        q = 0
        one = solve is not None and self.from_v_L and eng.world == 1 and hasattr(eng, "gks_rows_solve")
        if one and self.from_v_A:
            lam, Minv, ldm, k_from, Y = solve
            eng.gks_rows_solve(self.GA_d.ref(0), self.GL_d.ref(0), self.kmax, k, c, rho2, self.c_d.ref(0), lam, Minv, ldm, k_from, Y.ref(0),
                               gs.extra_ref(1, k), Sw.ref(so + 1), a_A=gs.extra_ref(0, k), s_A=Sw.ref(so), tb=Sw.ref(so + 2))
            return True
        if self.from_v_A:
            eng.gram_row_from_sweep(self.GA_d.ref(0), self.kmax, k, gs.extra_ref(q, k), c, Sw.ref(so), rho2, rhs=self.c_d.ref(0), tb=Sw.ref(so + 2))
            q += 1
        else:                                                             # the images of A are kept (small m): as in _push_images
            av = self.AV.next_slot()
            if r_early is not None:
                if getattr(self, "_tA_r", None) is None:
                    self._tA_r = eng.empty(self.A.shape[0])
                self.A.apply(r_early, out=self._tA_r)
                eng.gemv_orth_iterate(self.AV.data, k, self._tA_r, c, rho2, av)
            else:
                self.A.apply(self.V[k], out=av)
            self.AV.commit()
            with_row = hasattr(eng, "gemv_t_x")       # c_k = (A v_k) . b from the Gram row's own pass (b as one more row)
            if eng.world > 1:                                             # c_k behind the Gram row: one exchange
                if with_row:
                    eng.gemv_t_x(self.AV.data, k + 1, av, self.bv, S.ref(4), S.ref(4 + k + 1))
                else:
                    eng.dot(av, self.bv, S.ref(4 + k + 1))
                    eng.gemv_t(self.AV.data, k + 1, av, S.ref(4))
                eng.allreduce(S, 4, 4 + k + 2)
                eng.copy_scalars(S, 4 + k + 1, self.c_d, k, 1)
            elif with_row:
                eng.gemv_t_x(self.AV.data, k + 1, av, self.bv, S.ref(4), self.c_d.ref(k))
            else:
                eng.dot(av, self.bv, self.c_d.ref(k))
                eng.gemv_t(self.AV.data, k + 1, av, S.ref(4))
            if one:
                lam, Minv, ldm, k_from, Y = solve
                eng.gks_rows_solve(self.GA_d.ref(0), self.GL_d.ref(0), self.kmax, k, c, rho2, self.c_d.ref(0), lam, Minv, ldm, k_from,
                                   Y.ref(0), gs.extra_ref(q, k), Sw.ref(so + 1), ga_new=S.ref(4))
                return True
            eng.cgs_coeffs(self.GA_d.ref(0), self.kmax, None, S.ref(4), k + 1, 0, None)
        if self.from_v_L:
            eng.gram_row_from_sweep(self.GL_d.ref(0), self.kmax, k, gs.extra_ref(q, k), c, Sw.ref(so + 1), rho2)
        return False


@small_host_blas
def GKS(A, b, L, projection_dim=3, n_iter=50, regparam="gcv", x_true=None, **kwargs):
    """Returns (x, info); info keys xHistory, regParam, regParam_history, relError (if x_true), Residual, its (= n_iter-1).
    Engine-only kwarg: history (True, False, a stride, 'host' or a .npy path: _io.History)."""
    A, L = as_operator(A), as_operator(L, "L")
    check_delta(regparam, kwargs)
    if is_identity(L):
        raise NotImplementedError("GKS with L = Identity: the reference's SVD branch (GKS.py:44-50) sets R_L to a pylops Identity and then "
                                  "fails in np.concatenate((R_A, sqrt(lam) * R_L)) (:74); not reproduced — use Hybrid_LSQR for L = I")
    eng = A.engine
    m, n = A.shape
    n_iter, d = int(n_iter), int(projection_dim)
    fmt = Formatter(b)
    bv = eng.to_vec(b, m)
    xt = None if x_true is None else eng.to_vec(x_true, n)
    kmax = d + n_iter + 1

    gk = golub_kahan_device(A, bv, d, kwargs.get("dp_stop", False),                 # GKS.py:36 / MMGKS.py:37
                            **{k_: v_ for k_, v_ in kwargs.items() if k_ in ("gk_eta", "gk_delta")})
    # numeric regparam: the projected problem is solved on the device from device-resident Gram data (no host round trip per
    # iteration); the automatic selectors need the factors on the host
    on_dev = (not isinstance(regparam, str)) and hasattr(eng, "gram_tikhonov") and hasattr(eng, "cgs_coeffs") \
        and (kmax <= eng.GRAM_TIKHONOV_MAX_K or kwargs.get("border_inverse", True)) and kwargs.get("device_solve", True)
    dA, dL = bool(getattr(A, "streaming", False)), bool(getattr(L, "streaming", False))
    from_v = hasattr(eng, "gemv_t2") and kwargs.get("gram_from_v", True)
    # an automatic regparam ('gcv', the reference's default, 'dp', 'l_curve'; late round 6): the Gram data is kept on the device all the same — its rows then come from the
    # orthogonalisation sweep's own pass over V (gram_rows_from_sweep) instead of two sweeps of their own — and the selector downloads it
    # with one copy per iteration (kwarg device_gram)
    dev_gram = on_dev or (isinstance(regparam, str) and hasattr(eng, "gram_tikhonov") and hasattr(eng, "cgs_coeffs")
                          and kwargs.get("device_solve", True) and kwargs.get("device_gram", True))
    pb = _ProjectedBases(A, L, bv, gk.V, kmax, on_device=dev_gram, from_v_A=dA and from_v, from_v_L=dL and from_v)
    Hs = History(eng, kwargs.get("history", True), n_iter, n, "GKS xHistory")
    Y = eng.scalars(kmax)
    H = eng.scalars(3 * kmax)
    E = eng.scalars(n_iter + 3)             # E[0] = ||x_true||^2, E[1] = ||b||^2, E[2+i] = ||x_i - x_true||^2
    R = eng.scalars(n_iter + 1)             # ||r_i||^2
    tm, tp, r, rb = eng.empty(m), eng.empty(L.shape[0]), eng.empty(n), eng.empty(n)
    eng.nrm2sq(bv, E.ref(1))
    if xt is not None:
        eng.nrm2sq(xt, E.ref(0))
    eng.allreduce(E, 0, 2)
    b2 = float(E.host(1, 2)[0])

    fusedL = dL and getattr(L, "fused_tv", False) and kwargs.get("fused_tv", True)
    gs_gram = GramSchmidtByGram(eng, pb.V, kmax) if (hasattr(eng, "cgs_coeffs") and kwargs.get("gram_sweeps", True)) else None
    lams, lam, x_dev = [], None, None
    Minv, k_inv = (eng.scalars(kmax * kmax) if (on_dev and kwargs.get("border_inverse", True)) else None), 0
    # ||x_i - x_true||^2 rides the pass that forms x_i = V y (trk_gemv_n_err) as raw block partials
    err_fused = xt is not None and hasattr(eng, "gemv_n_err") and kwargs.get("fused_error_norm", True)
    EP_CAP = 2048
    EP, n_ep = (eng.scalars(EP_CAP * max(1, n_iter)) if err_fused else None), 0
    # One pass over the basis for the new vector AND the next iterate (late round 6; kwarg fused_orth_iterate): the projected problem of
    # iteration ii + 1 needs of v_k only its Gram rows and its norm, and both follow from the h-sweep's products — so it is solved
    # BEFORE r - V c is formed, and that pass leaves x_{ii+1} = V y' as well (trk_gemv_orth_iterate): two passes over V per iteration
    # instead of three.  Device-resident Gram data whose rows come from the sweep (`merged` below); on ranks r . r travels with the sweep's
    # products (the all-reduce of ||r - V c||^2 is gone) and the new vector's boundary frames come from the residual's, as before.
    early = (dev_gram and gs_gram is not None and pb.from_v_L and hasattr(eng, "gemv_orth_iterate") and hasattr(eng, "gram_row_from_sweep")
             and kmax < 1024 and kwargs.get("gram_rows_from_sweep", True) and kwargs.get("fused_orth_iterate", True))
    x_ready = False

    def projected_problem(k):
        nonlocal lam, k_inv
        if on_dev:
            lam = regparam
            lams.append(lam)
            # (:74) the Gram matrices only grow: the inverse of G_A + lam G_L is bordered by the rows new since the last call
            eng.gram_tikhonov(pb.GA_d.ref(0), kmax, pb.GL_d.ref(0), kmax, pb.c_d.ref(0), k, lam, Y.ref(0),
                              Minv=Minv, ldm=kmax, k_from=k_inv)
            k_inv = k
            return
        hGA, hGL, hc = pb.download_grams(k) if dev_gram else (pb.GA[:k, :k], pb.GL[:k, :k], pb.c[:k])
        one = gram_gcv_host(hGA, hGL, hc, hc) if (regparam == "gcv" and kwargs.get("host_solve_in_c", True)) else None
        if one is not None:                   # the whole projected problem in one library call (trk_host_gram_gcv)
            lam, y = one
            lams.append(lam)
        else:
            R_A, R_L = gram_factor(hGA), gram_factor(hGL)
            rhs = project_rhs(R_A, hc)
            lam = choose_lambda(regparam, R_A, R_L, rhs, max(b2 - float(rhs @ rhs), 0.0), kwargs)
            lams.append(lam)
            y = tikhonov_lstsq(R_A, R_L, lam, rhs)
        Y.set(0, y)

    for ii in _trace.progress(range(n_iter), "running GKS...", kwargs.get("progress")):       # (GKS.py:42)
        k = pb.V.k
        if not x_ready:
            _trace.mark("GKS: projected problem")
            projected_problem(k)
            _trace.mark("GKS: iterate x = V y")
            x_dev = Hs.row(ii)
            if err_fused:      # x = V y (:76) with ||x - x_true||^2 as block partials of the same pass, summed once after the loop
                n_ep = eng.gemv_n_err(pb.V.data, k, Y.ref(0), x_dev, xt, EP.ref(n_ep * ii), EP_CAP)
            else:
                eng.gemv_n(pb.V.data, k, Y.ref(0), x_dev)                               # x = V y (:76)
            Hs.pushed(ii)
            if xt is not None and not err_fused:
                eng.diff_nrm2sq(x_dev, xt, E.ref(2 + ii))
        x_ready = False
        # r = A^T (A x - b) + lam L^T (L x), A x = (AV) y and L x = (LV) y in the reference (:81-85); stencil operators
        # form them directly from x (8n-12n bytes instead of k basis vectors)
        _trace.mark("GKS: residual")
        if dA:
            A.apply(x_dev, out=tm)
            eng.axpby(1.0, tm, -1.0, bv, tm)
        else:
            eng.gemv_n(pb.AV.data, k, Y.ref(0), tm, a=-1.0, base=bv, s=1.0)
        A.apply(tm, out=r, transpose=True)
        if fusedL:
            # time-sharded: the iterate's boundary frames of the neighbour ranks are the same combination of the basis vectors' (no exchange)
            hk = {} if pb.halo is None else {"halo": pb.halo.of_iterate(k, Y.ref(0))}
            L.tv_grad(x_dev, None, r, float(lam), out=rb, **hk)                      # r + lam L^T L x in one stencil pass
            r, rb = rb, r
            if pb.halo is not None:
                pb.r_halo = pb.halo.of_residual(r)                                   # the ONE neighbour exchange of the iteration
        else:
            if dL:
                L.apply(x_dev, out=tp)
            else:
                eng.gemv_n(pb.LV.data, k, Y.ref(0), tp)
            L.apply(tp, out=rb, transpose=True)
            eng.axpby(1.0, r, float(lam), rb, r)
        _trace.mark("GKS: orthogonalise, new basis vector")
        vn = pb.V.next_slot()
        merged = (dev_gram and gs_gram is not None and pb.from_v_L and hasattr(eng, "gram_row_from_sweep")
                  and gs_gram.in_G == k - 1 and kwargs.get("gram_rows_from_sweep", True))
        cc = None
        if merged and early:
            n_extra = int(pb.from_v_A) + int(pb.from_v_L)
            off = (2 + n_extra) * k
            # (r . r behind the three scalars of sweep_operands: from the stencil pass that forms L^T L r)
            cc = gs_gram.sweep(k, r, 3, None, extra=pb.sweep_operands(r, scal=gs_gram.W, off=off, rr=True), tail=4,
                               rr=gs_gram.W.ref(off + 3), rho2=R.ref(ii))            # (:86-88) h, the Gram rows' products, c, rho^2
            if pb.halo is not None:
                pb.halo.push_from_sweep(k, cc, R.ref(ii))                            # the new vector's boundary frames
            _trace.mark("GKS: Gram rows")
            last = ii + 1 >= n_iter
            # rows k of the Gram data, without v_k (:92-96) — with a numeric lambda on one rank together with the solve of the NEXT
            # iteration's projected problem (:74) in one launch
            merged_solve = (on_dev and Minv is not None and not last and kwargs.get("rows_and_solve_in_one", True))
            solved = pb.append_from_sweep(gs_gram, k, cc, R.ref(ii), r_early=r,
                                          solve=(float(regparam), Minv, kmax, k_inv, Y) if merged_solve else None)
            if solved:
                lam = regparam
                lams.append(lam)
                k_inv = k + 1
            elif not last:
                _trace.mark("GKS: projected problem")
                projected_problem(k + 1)                                             # (:74 of the NEXT iteration)
            _trace.mark("GKS: new basis vector and next iterate, one pass")
            if last:
                eng.gemv_orth_iterate(pb.V.data, k, r, cc, R.ref(ii), vn)            # v_k = (r - V c)/||.|| (:86-91)
            else:
                x_dev = Hs.row(ii + 1)
                if err_fused:
                    n_now = eng.gemv_orth_iterate(pb.V.data, k, r, cc, R.ref(ii), vn, y_next=Y.ref(0), x_next=x_dev, ref=xt,
                                                  partials=EP.ref(n_ep * (ii + 1)), capacity=EP_CAP)
                    if n_now != n_ep:
                        raise RuntimeError("GKS: the error partials of the one-pass form do not match trk_gemv_n_err's layout")
                else:
                    eng.gemv_orth_iterate(pb.V.data, k, r, cc, R.ref(ii), vn, y_next=Y.ref(0), x_next=x_dev)
                Hs.pushed(ii + 1)
                if xt is not None and not err_fused:
                    eng.diff_nrm2sq(x_dev, xt, E.ref(2 + ii + 1))
                x_ready = True
            pb.V.commit()
            continue
        if merged:
            # the sweep's pass over V also takes V^T (A^T A r), V^T (L^T L r): the next vector's Gram rows need no pass of their own
            # (the three scalars r . M r ride behind the sweep's products: one all-reduce for both)
            n_extra = int(pb.from_v_A) + int(pb.from_v_L)
            off = (2 + n_extra) * k
            cc = gs_gram.sweep(k, r, 3, vn, sumsq=R.ref(ii), extra=pb.sweep_operands(r, scal=gs_gram.W, off=off), tail=3)
        elif gs_gram is not None:
            cc = gs_gram.sweep(k, r, 3, vn, sumsq=R.ref(ii))                         # (:86-88) three sweeps, ||r||^2 fused
        else:
            orthogonalize(eng, pb.V, k, r, H, 0, passes=3, out=vn, sumsq=R.ref(ii))
        eng.allreduce(R, ii, ii + 1)
        if merged:
            eng.scale(Coef(1.0, den=R.ref(ii), sqrt_den=True), vn, vn)               # vn = r/||r|| (:89-91)
        else:
            pb.normalise_new(Coef(1.0, den=R.ref(ii), sqrt_den=True), vn)
        pb.V.commit()
        if pb.halo is not None:                                                      # the new vector's boundary frames: from the
            if cc is not None and pb.r_halo is not None:                             # residual's and the sweep's coefficients
                pb.halo.push_from_sweep(k, cc, R.ref(ii))
            else:
                pb.halo.push_exchanged(pb.V[k])
        _trace.mark("GKS: Gram rows")
        if merged:
            pb.append_from_sweep(gs_gram, k, cc, R.ref(ii))
        else:
            pb.append()                                                              # AV, LV, Gram rows (:92-96)
    _trace.mark(None)
    info = {"xHistory": Hs.collect(fmt, n_iter), "regParam": lam, "regParam_history": lams,
            "Residual": list(np.sqrt(R.host(0, n_iter))), "its": n_iter - 1}
    if getattr(L, "sharded", False):
        info["fused_tv"] = bool(fusedL)                 # engine-only: the kernels of the one-rank solve ran on every rank
    if xt is not None:
        if err_fused:
            eng.finalize_batched(EP.ref(0), n_ep, 1, n_iter, E.ref(2), 1)
        eng.allreduce(E, 2, 2 + n_iter)
        e = E.host(0, 2 + n_iter)
        info["relError"] = list(np.sqrt(e[2:] / e[0]))
    return fmt.vec(x_dev), info
