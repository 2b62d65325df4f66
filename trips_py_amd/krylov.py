"""Engine-level Krylov building blocks: a growable row-per-vector device basis and the Golub-Kahan / Arnoldi steps
(reference: trips/utilities/decompositions.py:207-255) expressed in libtrk kernels.

Scalars produced by a step (alpha^2, beta^2, the Gram-Schmidt coefficients) stay in device doubles and are consumed by
the next kernel through device-evaluated coefficients; the host reads them once per step (it needs them for the
projected problem anyway).  Sharded problems all-reduce the same doubles (`engine.allreduce`).
"""
import numpy as np

from .engine import Coef



def _plain_handle_apply(A):
    """True when applying A is nothing but the library call on its handle (operators whose Python `_apply` does more — the
    time-sharded space-time operator exchanges halos first — must keep going through it)."""
    from .operators import _HandleOperator
    return isinstance(A, _HandleOperator) and type(A)._apply is _HandleOperator._apply


class DeviceBasis:
    """k vectors of length n stored row-per-vector in one [capacity, n] fp32 device tensor.

    Appending is a write into the next row (the reference re-copies the whole basis with hstack / column_stack / pad at
    every step: decompositions.py:172-175,243-254; GKS.py:91-96).  `.shape` is the reference's (n, k)."""

    def __init__(self, engine, n, capacity):
        self.engine, self.n = engine, int(n)
        self.data = engine.empty_basis(max(1, int(capacity)), n)
        self.k = 0

    @property
    def shape(self):
        return (self.n, self.k)

    def reserve(self, capacity):
        if capacity > self.data.shape[0]:
            new = self.engine.empty_basis(max(capacity, 2 * self.data.shape[0]), self.n)
            new[:self.k].copy_(self.data[:self.k])
            self.data = new

    def next_slot(self):
        """The row the next vector will be written into (not yet counted)."""
        self.reserve(self.k + 1)
        return self.data[self.k]

    def commit(self):
        self.k += 1

    def __getitem__(self, j):
        if j < 0:
            j += self.k
        if not 0 <= j < self.k:
            raise IndexError(j)
        return self.data[j]

    def rows(self, k=None):
        return self.data[:self.k if k is None else k]

    def numpy(self, k=None):
        """(n, k) float64 array, the reference's layout."""
        return self.rows(k).detach().to("cpu").numpy().astype(np.float64).T.copy()

    def torch_cols(self, k=None):
        return self.rows(k).T


# --------------------------------------------------------------------------------------------------------------------
class GKState:
    """Golub-Kahan bidiagonalisation A V_k = U_{k+1} B_k, one step at a time (decompositions.py:230-255, no
    reorthogonalisation).  U, V live on the device, and so do the squared norms that define B_k:
    AB[0] = beta0^2 = ||b||^2, AB[2j+1] = alpha_j^2, AB[2j+2] = beta_{j+1}^2 (global sums after `allreduce`).
    `alphas`, `betas` are their host copies; `step(sync=False)` skips the download, so a caller that needs B_k only
    on the device (fixed-lambda Hybrid_LSQR: trk_bidiag_tikhonov) never synchronises.

    normalized=False keeps the vectors as the recurrence produces them, U[j] = beta_j u_j (beta_0 := beta0, U[0] = b)
    and V[j] = alpha_j v_j, and folds the divisions into the coefficients of the next step:
        V[k]   = (1/beta_k)  A^T U[k] - (beta_k/alpha_{k-1}) V[k-1]        ||V[k]||   = alpha_k
        U[k+1] = (1/alpha_k) A V[k]   - (alpha_k/beta_k)     U[k]          ||U[k+1]|| = beta_{k+1}
    (same alpha, beta; one rounding less per element) — two kernels and 8 (n + m) bytes less per step.  Consumers
    divide by the norms themselves: x = V y takes y_j / alpha_j, U^T b gives beta_j (u_j . b)."""

    PROJ_PARTIALS = 4096            # room for the block partials of <U[k+1], project> (512^2 x 180 angles: 363)

    def __init__(self, A, b, capacity, normalized=True):
        self.A, self.eng = A, A.engine
        self._proj, self._proj_n = None, 0
        self.normalized = bool(normalized)
        # operators whose own output pass takes the vector update and the norm of a half step (the Radon projector)
        self.native_axpby = bool(getattr(A, "native_axpby", False)) and not self.normalized
        self._chained = False                   # U[k] came out of this state's previous fused forward apply
        self._UB, self.uproj = None, None       # U^T b, one entry per row of U (step_prefetch(project=b))
        self._late, self._posted_any = None, False   # step_prefetch: a step whose download waits for the next step
        self.rider_posts = True                      # ... and is carried by that step's adjoint kernel where the operator can
        m, n = A.shape
        eng = self.eng
        self.U = DeviceBasis(eng, m, capacity + 1)
        self.V = DeviceBasis(eng, n, capacity)
        self.tmp_n, self.tmp_m = eng.empty(n), eng.empty(m)
        # behind the 2 capacity + 1 squared norms: capacity + 2 places for U^T b (step_prefetch(project=...)), so that one mailbox
        # post carries a step's norms and its new U^T b entry together
        self._ab_cap = 2 * max(1, capacity) + 1
        self._uoff = self._ab_cap
        self._UP, self._merged = None, False    # block partials of <U[k+1], project>, two halves by step parity
        self.AB = eng.scalars(self._ab_cap + max(1, capacity) + 2)
        self._alphas, self._betas = [], []
        bv = eng.to_vec(b, m)
        eng.nrm2sq(bv, self.AB.ref(0))
        eng.allreduce(self.AB, 0, 1)
        if self.normalized:
            eng.scale(Coef(1.0, den=self.AB.ref(0), sqrt_den=True), bv, self.U.next_slot())
        else:
            self.U.next_slot().copy_(bv)
        self.U.commit()
        self._beta0 = None

    @classmethod
    def resume(cls, A, U_cols, V_cols, alphas, betas):
        """Cold start from host arrays (golub_kahan_update handed the reference's U, S, V): upload the vectors and the
        bidiagonal so that the next step continues the factorisation."""
        st = cls.__new__(cls)
        st.A, st.eng = A, A.engine
        st.normalized = True
        st.native_axpby, st._chained = False, False
        st._UB, st.uproj = None, None
        st._proj, st._proj_n = None, 0
        st._late, st._posted_any = None, False
        st.rider_posts = False
        eng = A.engine
        m, n = A.shape
        k = len(alphas)
        st.U, st.V = DeviceBasis(eng, m, k + 2), DeviceBasis(eng, n, k + 1)
        for u in U_cols:
            st.U.next_slot().copy_(eng.to_vec(u, m))
            st.U.commit()
        for v in V_cols:
            st.V.next_slot().copy_(eng.to_vec(v, n))
            st.V.commit()
        st.tmp_n, st.tmp_m = eng.empty(n), eng.empty(m)
        st._ab_cap = 2 * (k + 1) + 1
        st._uoff, st._UP, st._merged = None, None, False
        st.AB = eng.scalars(st._ab_cap)
        if k:
            ab = np.zeros(2 * k + 1)
            ab[1::2], ab[2::2] = np.square(alphas), np.square(betas)
            st.AB.set(0, ab)
        st._alphas, st._betas = [float(a) for a in alphas], [float(b) for b in betas]
        st._beta0 = float("nan")                # unknown: the reference's arrays do not carry ||b||
        return st

    # host copies, downloaded on demand
    def _sync(self, need=None):
        k = self.V.k if need is None else need
        if len(self._alphas) < k:
            self.flush()
            k = self.V.k
            h = self.AB.host(0, 2 * k + 1)
            if self._beta0 is None:
                self._beta0 = float(np.sqrt(h[0]))
            self._alphas, self._betas = list(np.sqrt(h[1::2])), list(np.sqrt(h[2::2]))

    @property
    def alphas(self):
        self._sync()
        return self._alphas

    @property
    def betas(self):
        self._sync()
        return self._betas

    @property
    def beta0(self):
        if self._beta0 is None:
            self._beta0 = float(np.sqrt(self.AB.host(0, 1)[0]))
        return self._beta0

    def step_prefetch(self, project=None, more_follow=False):
        """One step, enqueued, plus the start of the download of its two norms: `absorb()` later waits for that copy
        only, so a caller can enqueue the NEXT step before it looks at this one's numbers (the hybrid solvers choose
        lambda_k on the host while the device already runs step k+1, which does not depend on it).
        Returns the LIST of downloads started by this call, oldest first, each to be handed to `absorb()` in order.
        more_follow=True: the caller promises another step_prefetch before anything else touches this state — beta_{k+1}^2 then
        stays inside the operator (step(defer=True)) until the NEXT step's adjoint kernel finishes it, and this step's download
        is started by that next call (none by this one: the list may be empty); the last call (more_follow=False) starts both.
        project: a vector b whose products with the rows of U the caller wants (the discrepancy principle's U^T b):
        the new row's U[k+1] . b (U[0] . b as well on the first step) is taken right behind the step and downloaded with
        the norms — `self.uproj` grows by one entry per absorbed step — instead of a pass over all of U and a blocking
        download per iteration."""
        k = self.V.k
        eng = self.eng
        j0 = None
        # one rank, an operator the library steps in one call: the forward half step's own output pass leaves <U[k+1], project> as
        # block partials (trk_gk_step_proj), and the post of the step's norms adds them up on the way to the host
        # (trk_mailbox_post_sum) — a dot, its reduction launch and a second post less per step.  The first step (which also owes
        # U[0] . project) and everything else take the dots separately.
        merged = (project is not None and k > 0 and hasattr(eng, "gk_step_proj") and getattr(eng, "world", 1) == 1
                  and self.native_axpby and getattr(self.A, "_h", None) and self._uoff is not None and self._UB is not None
                  and 2 * k + 3 <= self._ab_cap)
        if merged:
            if self._UP is None:
                self._UP = eng.scalars(2 * self.PROJ_PARTIALS)
            self._proj = (project, self._UP.ref(self.PROJ_PARTIALS * (k & 1)), self.PROJ_PARTIALS)
        # the post of the step before (its last norm is finished by THIS step's adjoint kernel) rides that kernel where it can
        late, rider = self._late, None
        late_lo = None if late is None else self._post_lo(late[0])
        if (late is not None and k >= 1 and self.native_axpby and hasattr(eng, "gk_step_post") and getattr(eng, "world", 1) == 1
                and getattr(self.A, "_h", None) and (late[1] is None or isinstance(late[1], tuple)) and self.rider_posts):
            rider = self._post_args(late[0], late[1], late_lo)
        self.step(sync=False, defer=bool(more_follow), post=None if rider is None else rider[0])
        n_part = self._proj_n if merged else 0
        self._proj = None
        if merged:
            j0 = ("merged", self._UP.ref(self.PROJ_PARTIALS * (k & 1)), n_part)
        elif project is not None:
            if self._UB is None:
                self._UB = eng.scalars(self.U.data.shape[0] + 1)
                self.uproj = []
            j0 = 0 if k == 0 else k + 1
            for j in range(j0, k + 2):
                eng.dot(self.U[j], project, self._UB.ref(j))
            eng.allreduce(self._UB, j0, k + 2)
        started = []
        self._late = None
        if late is not None:
            if rider is not None and self.post_taken:
                started.append(rider[1])
            else:
                started.append(self._post_step(late[0], late[1], late_lo))   # the step before: its beta^2 was finished by this step's first kernel
        if more_follow:
            self._late = (k, j0)
        else:
            started.append(self._post_step(k, j0))
        return started

    def _post_lo(self, k):
        """Where the post of step k starts: the first one also carries beta0^2."""
        first = not self._posted_any
        self._posted_any = True
        return 0 if first else 2 * k + 1

    def _post_args(self, k, j0, lo):
        """_post_step as a rider: (arguments for trk_gk_step_post, the (k, lo, handle, extra) tuple absorb() takes)."""
        if isinstance(j0, tuple):
            _, part, n_part = j0
            args, h, hs = self.AB.rider_post(lo, 2 * k + 3, part, n_part, self._uoff + k + 1)
        else:
            args, h, hs = self.AB.rider_post(lo, 2 * k + 3)
        return args, (k, lo, h, hs)

    def _post_step(self, k, j0, lo=None):
        if lo is None:
            lo = self._post_lo(k)
        if isinstance(j0, tuple):                   # the norms and the new U^T b entry (AB[uoff + k + 1]) in one post
            _, part, n_part = j0
            at = self._uoff + k + 1
            handle, extra = self.AB.host_later_sum(lo, 2 * k + 3, part, n_part, at)
            return k, lo, handle, extra
        extra = None if j0 is None else self._UB.host_later(j0, k + 2)
        return k, lo, self.AB.host_later(lo, 2 * k + 3), extra

    def absorb(self, pending):
        k, lo, handle, extra = pending
        if extra is not None:
            self.uproj.extend(float(t) for t in extra.get())
        if len(self._alphas) > k:
            return                                  # a full download overtook it
        v = np.sqrt(handle.get())
        if lo == 0:
            self._beta0 = float(v[0])
            v = v[2 * k + 1:]
        self._alphas.append(float(v[0]))
        self._betas.append(float(v[1]))

    def flush(self):
        """Finish a norm a `step(defer=True)` left inside the operator (before anything reads AB)."""
        if self.native_axpby:
            self.A.flush_deferred()

    def step(self, sync=True, defer=False, lsqr=None, post=None):
        """defer=True (with sync=False): beta_{k+1}^2 may stay unfinished inside the operator until the next step or
        `flush()` — for loops that look at AB only at the end (fixed-lambda Hybrid_LSQR without history)."""
        A, eng = self.A, self.eng
        k = self.V.k
        self.lsqr_taken, self.lsqr_blocks, self.post_taken = False, 0, False
        defer = bool(defer) and not sync
        if self._ab_cap < 2 * k + 3:
            self.flush()
            new = eng.scalars(4 * k + 8)
            new.view(0, 2 * k + 1).copy_(self.AB.view(0, 2 * k + 1))
            self.AB = new
            self._ab_cap, self._uoff = 4 * k + 8, None       # (no U^T b places behind the regrown block: separate dots from here on)
        AB = self.AB
        u = self.U[k]
        a2, b2 = AB.ref(2 * k + 1), AB.ref(2 * k + 2)
        if not self.normalized:
            bk2 = AB.ref(2 * k)                                     # ||U[k]||^2
            v = self.V.next_slot()
            ca_v = Coef(1.0, den=bk2, sqrt_den=True)
            cb_v = None if k == 0 else Coef(-1.0, num=bk2, den=AB.ref(2 * k - 1), sqrt_num=True, sqrt_den=True)
            ca_u = Coef(1.0, den=a2, sqrt_den=True)
            cb_u = Coef(-1.0, num=a2, den=bk2, sqrt_num=True, sqrt_den=True)
            if self.native_axpby:
                # the operator's output pass carries the vector update and the norm (trk_op_apply_axpby): no vector kernel,
                # no reduction launch; U[k] / V[k] go from one half step straight into the other, so what the next apply
                # derives from its input (detector records, transposed image) is left behind by the one that writes it
                feeds, takes = A.OUT_FEEDS_OPPOSITE, A.INPUT_FROM_OPPOSITE
                # single rank: alpha_k^2 stays block partials until the forward apply below adds them up for its own
                # coefficients (no reduction launch); beta_{k+1}^2 likewise until the next step, if the caller allows
                local = getattr(eng, "world", 1) == 1
                if local and hasattr(eng, "gk_step") and getattr(A, "_h", None):
                    # the whole step in one call of the library (trk_gk_step: the two half steps below, same coefficients,
                    # same hints) — the Python side of a step was a fifth of a 512^2 Hybrid-LSQR iteration's host time
                    un = self.U.next_slot()
                    if post is not None and k >= 1 and hasattr(eng, "gk_step_post"):
                        # the mailbox post of the step before rides this step's adjoint half (trk_gk_step_post)
                        n_pr = eng.gk_step_post(A._h, k, u, self.V[k - 1], v, un, AB, self._chained, True, defer, post,
                                                proj=getattr(self, "_proj", None))
                        if getattr(self, "_proj", None) is not None:
                            self._proj_n = n_pr
                        self.post_taken = True
                    elif getattr(self, "_proj", None) is not None:
                        self._proj_n = eng.gk_step_proj(A._h, k, u, self.V[k - 1], v, un, AB, self._chained, True, defer, *self._proj)
                    elif lsqr is not None and k >= 1 and hasattr(eng, "gk_step_lsqr"):
                        # lsqr = (w, x_in, x_out, ref, partials, capacity, damp, state_in, state_out): damped LSQR's update of the
                        # iterate that V[k-1] belongs to, carried by this step's adjoint half (trk_gk_step_lsqr)
                        self.lsqr_blocks = eng.gk_step_lsqr(A._h, k, u, self.V[k - 1], v, un, AB, self._chained, True, defer, *lsqr)
                        self.lsqr_taken = True
                    else:
                        eng.gk_step(A._h, k, u, None if k == 0 else self.V[k - 1], v, un, AB, self._chained, True, defer)
                    self.V.commit()
                    self.U.commit()
                    self._chained = True
                    return self._finish_step(k, sync)
                later = A.SUMSQ_DEFERRED if local else 0
                A.apply_axpby(u, ca_v, 0.0 if k == 0 else cb_v, None if k == 0 else self.V[k - 1], v, transpose=True,
                              sumsq=a2, hints=feeds | (takes if self._chained else 0) | later)
                eng.allreduce(AB, 2 * k + 1, 2 * k + 2)
                self.V.commit()
                A.apply_axpby(v, ca_u, cb_u, u, self.U.next_slot(), sumsq=b2, hints=feeds | takes | (later if defer else 0))
                eng.allreduce(AB, 2 * k + 2, 2 * k + 3)
                self.U.commit()
                self._chained = True
                return self._finish_step(k, sync)
            if getattr(A, "_h", None) and hasattr(A, "apply_axpby") and _plain_handle_apply(A):
                # a handle without the hinted chains: the half steps still go through trk_op_apply_axpby — apply + trk_axpby inside the
                # library for most operators, the operator's own store where it has one (separable blurs: k_blur_slide<.., EPI>); the
                # same bits either way, no temporary, one call per half step
                A.apply_axpby(u, ca_v, 0.0 if k == 0 else cb_v, None if k == 0 else self.V[k - 1], v, transpose=True, sumsq=a2)
                eng.allreduce(AB, 2 * k + 1, 2 * k + 2)
                self.V.commit()
                A.apply_axpby(v, ca_u, cb_u, u, self.U.next_slot(), sumsq=b2)
                eng.allreduce(AB, 2 * k + 2, 2 * k + 3)
                self.U.commit()
                return self._finish_step(k, sync)
            A.apply(u, out=self.tmp_n, transpose=True)
            if k == 0:
                eng.scale(ca_v, self.tmp_n, v, sumsq=a2)
            else:
                eng.axpby(ca_v, self.tmp_n, cb_v, self.V[k - 1], v, sumsq=a2)
            eng.allreduce(AB, 2 * k + 1, 2 * k + 2)
            self.V.commit()
            A.apply(v, out=self.tmp_m)
            eng.axpby(ca_u, self.tmp_m, cb_u, u, self.U.next_slot(), sumsq=b2)
            eng.allreduce(AB, 2 * k + 2, 2 * k + 3)
            self.U.commit()
            return self._finish_step(k, sync)
        # v = A^T u_k - beta_k v_{k-1} ; alpha = ||v||        (beta_k^2 sits in AB[2k])
        if k == 0:
            A.apply(u, out=self.tmp_n, transpose=True, sumsq=a2)
        else:
            A.apply(u, out=self.tmp_n, transpose=True)
            eng.axpby(1.0, self.tmp_n, Coef(-1.0, num=AB.ref(2 * k), sqrt_num=True), self.V[k - 1], self.tmp_n, sumsq=a2)
        eng.allreduce(AB, 2 * k + 1, 2 * k + 2)
        v = self.V.next_slot()
        eng.scale(Coef(1.0, den=a2, sqrt_den=True), self.tmp_n, v)
        self.V.commit()
        # u' = A v_k - alpha u_k ; beta = ||u'||
        A.apply(v, out=self.tmp_m)
        eng.axpby(1.0, self.tmp_m, Coef(-1.0, num=a2, sqrt_num=True), u, self.tmp_m, sumsq=b2)
        eng.allreduce(AB, 2 * k + 2, 2 * k + 3)
        eng.scale(Coef(1.0, den=b2, sqrt_den=True), self.tmp_m, self.U.next_slot())
        self.U.commit()
        return self._finish_step(k, sync)

    def _finish_step(self, k, sync):
        if not sync:
            return None
        h = np.sqrt(self.AB.host(2 * k + 1, 2 * k + 3))
        if len(self._alphas) == k:
            self._alphas.append(float(h[0]))
            self._betas.append(float(h[1]))
        else:
            self._sync()
        return float(h[0]), float(h[1])

    def B(self, k=None):
        """(k+1) x k lower-bidiagonal projected matrix (k: leading part, host copies permitting without a download)."""
        self._sync(k)
        k = len(self._alphas) if k is None else k
        alphas, betas = self._alphas[:k], self._betas[:k]
        B = np.zeros((k + 1, k))
        B[np.arange(k), np.arange(k)] = alphas
        B[np.arange(1, k + 1), np.arange(k)] = betas
        return B


# --------------------------------------------------------------------------------------------------------------------
def orthogonalize(eng, V, k, w, H, off, passes=2, out=None, sumsq=None):
    """w <- w - V_k (V_k^T w), `passes` times (block classical Gram-Schmidt; GKS.py:86-88 uses 3 passes, MMGKS.py:119-120
    two; for Arnoldi two passes of CGS match the reference's modified Gram-Schmidt to rounding).
    Pass p leaves its k coefficients in H[off + p*k : off + (p+1)*k] (device doubles, all-reduced).
    The last pass may write its result to `out` instead of `w` (e.g. straight into the next basis slot) and leave the
    LOCAL sum of its squares in `sumsq` — fused into that kernel, no extra pass."""
    fuse = passes >= 2 and k <= getattr(eng, "GEMV_NT_MAX_K", 0) and hasattr(eng, "gemv_nt")
    if fuse:
        # pass p's update and pass p+1's dot products read the same rows of V: one sweep (trk_gemv_nt) instead of two
        eng.gemv_t(V.data, k, w, H.ref(off))
        eng.allreduce(H, off, off + k)
        for p in range(passes - 1):
            lo, nx = off + p * k, off + (p + 1) * k
            eng.gemv_nt(V.data, k, H.ref(lo), w, w, H.ref(nx))
            eng.allreduce(H, nx, nx + k)
        lo = off + (passes - 1) * k
        eng.gemv_n(V.data, k, H.ref(lo), out if out is not None else w, a=1.0, base=w, s=-1.0, sumsq=sumsq)
        return
    for p in range(passes):
        lo = off + p * k
        last = p == passes - 1
        eng.gemv_t(V.data, k, w, H.ref(lo))
        eng.allreduce(H, lo, lo + k)
        eng.gemv_n(V.data, k, H.ref(lo), out if (last and out is not None) else w, a=1.0, base=w, s=-1.0,
                   sumsq=sumsq if last else None)


class GramSchmidtByGram:
    """The repeated classical Gram-Schmidt sweeps of the projection solvers (GKS.py:86-88 three, MMGKS.py:119-120 and
    decompositions.py:216-218 two) with TWO passes over the basis instead of two per sweep: `passes` sweeps of
    r <- r - V (V^T r) equal r - V c, c from h = V^T r and the Gram matrix G = V^T V by a k x k recurrence
    (trk_cgs_coeffs).  G lives on the device and grows by one row per appended vector; that row (V^T v_new) is formed by the
    SAME sweep that forms the next h (trk_gemv_t2).  Any basis: the d Golub-Kahan start vectors are not re-orthogonalised,
    G simply says so.  "Equal" is exact-arithmetic algebra: sweeping literally, the later sweeps also project out the rounding
    error of the first fp32 subtraction, which the single r - V c here carries once.  Measured where that could matter most —
    Arnoldi on the 9 x 9 blur, 100 steps, h_{k+1,k} / ||A v_k|| down to 0.03: max |V^T V - I| 9.65e-8 this way, 9.71e-8 sweep by
    sweep (tests/test_gpu_solvers.py::test_arnoldi_orthogonality_by_gram_vs_sweeps); `by_gram=False` / `gram_sweeps=False` select
    the literal form.

    At 4096^2 a pass over k = 18 basis vectors is 1.2 GB: MMGKS goes from 4 (k > 16) / 3 to 2 passes per iteration for the
    sweeps, GKS from 6 / 4 to 2."""

    def __init__(self, eng, V, kmax):
        self.eng, self.V, self.kmax = eng, V, int(kmax)
        self.G = eng.scalars(self.kmax * self.kmax)
        self.W = eng.scalars(5 * self.kmax)           # h (k) | g_new (k) | [extra products (k each)] ... c at 4 kmax
        self.in_G = 0                                 # vectors whose Gram rows are installed
        for j in range(V.k):                          # the start basis: one sweep per vector (d of them)
            eng.gemv_t(V.data, j + 1, V[j], self.W.ref(0))
            eng.allreduce(self.W, 0, j + 1)
            eng.cgs_coeffs(self.G.ref(0), self.kmax, None, self.W.ref(0), j + 1, 0, None)
        self.in_G = V.k

    def sweep(self, k, w, passes, out, sumsq=None, c_out=None, extra=(), tail=0, rr=None, rho2=None):
        """out = w orthogonalised against V[0..k) by `passes` sweeps; LOCAL sum(out^2) into `sumsq` (fused).  Returns the
        DevScalars reference of the k combined coefficients.  extra: one or two more vectors z whose products V^T z ride on
        the same pass over the basis (trk_gemv_tn); they are left at `self.extra_ref(q, k)`.
        rr / rho2 (with extra; one rank): rr = w . w on the device -> rho2 = ||w - V c||^2 by algebra (trk_cgs_coeffs_rho) and NO
        pass forms `out` here (pass out=None): the caller's trk_gemv_orth_iterate forms it together with the next iterate."""
        eng, V, W, K = self.eng, self.V, self.W, self.kmax
        if k > K:
            raise ValueError("GramSchmidtByGram: basis larger than planned")
        c = W.ref(4 * K) if c_out is None else c_out
        if extra:
            if self.in_G != k - 1 or len(extra) > 2:
                raise RuntimeError("GramSchmidtByGram: extra right-hand sides ride on the sweep that installs the newest vector's row")
            eng.gemv_tn(V.data, k, [w, V[k - 1]] + list(extra), W.ref(0))
            # tail: scalars the caller has put right behind the products (GKS: r . A^T A r, r . L^T L r, r . A^T b) share the exchange
            if (2 + len(extra)) * k + tail > 4 * K:
                raise ValueError("GramSchmidtByGram: no room behind the sweep's products")
            eng.allreduce(W, 0, (2 + len(extra)) * k + tail)
            if rr is not None:
                eng.cgs_coeffs_rho(self.G.ref(0), K, W.ref(0), W.ref(k), k, passes, c, rr, rho2)
                self.in_G = k
                return c
            eng.cgs_coeffs(self.G.ref(0), K, W.ref(0), W.ref(k), k, passes, c)
            self.in_G = k
        elif self.in_G == k - 1:                      # the newest vector's Gram row rides along with h
            eng.gemv_t2(V.data, k, w, V[k - 1], W.ref(0))
            eng.allreduce(W, 0, 2 * k)
            eng.cgs_coeffs(self.G.ref(0), K, W.ref(0), W.ref(k), k, passes, c)
            self.in_G = k
        elif self.in_G == k:
            eng.gemv_t(V.data, k, w, W.ref(0))
            eng.allreduce(W, 0, k)
            eng.cgs_coeffs(self.G.ref(0), K, W.ref(0), None, k, passes, c)
        else:
            raise RuntimeError("GramSchmidtByGram: more than one vector appended since the last sweep")
        eng.gemv_n(V.data, k, c, out, a=1.0, base=w, s=-1.0, sumsq=sumsq)
        return c

    def extra_ref(self, q, k):
        """Where the last sweep over k vectors left V^T extra[q]."""
        return self.W.ref((2 + q) * k)


class ArnoldiState:
    """Arnoldi A V_k = V_{k+1} H_k (decompositions.py:207-228): orthogonalisation against ALL previous vectors."""

    def __init__(self, A, b, capacity, by_gram=True):
        """by_gram: the two Gram-Schmidt sweeps of a step as ONE pair of passes over the basis (GramSchmidtByGram) where the
        engine has the kernels; False: sweep by sweep (two pairs of passes; the second sweep then also removes the rounding error
        of the first subtraction — see tests/test_gpu_solvers.py::test_arnoldi_orthogonality_by_gram_vs_sweeps for what that buys)."""
        self.A, self.eng = A, A.engine
        self.by_gram = bool(by_gram)
        m, n = A.shape
        if m != n:
            raise ValueError("Arnoldi can not be used. The operator is not square")
        eng = self.eng
        self.V = DeviceBasis(eng, n, capacity + 1)
        self.w = eng.empty(n)
        self.S = eng.scalars(2 * (capacity + 1) + 2)
        self.Hcols = []
        self._Hfull, self._Hn = np.zeros((int(capacity) + 2, int(capacity) + 1)), 0      # the columns absorbed so far, in place (H_view)
        self.gram, self.capacity = None, int(capacity)
        bv = eng.to_vec(b, n)
        eng.nrm2sq(bv, self.S.ref(0))
        eng.allreduce(self.S, 0, 1)
        eng.scale(Coef(1.0, den=self.S.ref(0), sqrt_den=True), bv, self.V.next_slot())
        self.V.commit()
        self.beta0 = float(np.sqrt(self.S.host(0, 1)[0]))

    def _enqueue(self, post=False):
        """The next step, enqueued; returns its index k — or, with post=True, (k, handle): the download of S[0 .. 1+2k) started with
        the step, riding on the step's last kernel where the library's one-call form runs (handle None: the caller posts it)."""
        if post:
            self._post_handle = None
            k = self._enqueue_impl(True)
            return k, self._post_handle
        return self._enqueue_impl(False)

    def _enqueue_impl(self, post):
        A, eng, S, V = self.A, self.eng, self.S, self.V
        k = V.k
        if len(S) < 2 * k + 2:
            self.S = S = eng.scalars(4 * k + 2)
        slot = V.next_slot()
        # two Gram-Schmidt sweeps (= the reference's modified Gram-Schmidt to rounding), written straight into the next slot with
        # ||.||^2 in S[0]: by Gram matrix (two passes over the basis) where the engine has the kernels, else sweep by sweep
        if self.gram is None and self.by_gram and hasattr(eng, "cgs_coeffs") and self.capacity is not None:
            self.gram = GramSchmidtByGram(eng, V, self.capacity + 1)
        if (self.gram is not None and self.gram.in_G == k - 1 and getattr(eng, "world", 1) == 1
                and getattr(eng, "arnoldi_step", None) is not None and getattr(A, "_h", None) and _plain_handle_apply(A) and V.data.stride(0) >= A.shape[0]):
            # the whole step in one call of the library (trk_arnoldi_step: the very calls below, same arguments, same results; the
            # Python side of a step was a third of a Hybrid-GMRES iteration on the 512^2 blur)
            if post and getattr(eng, "arnoldi_step_post", None) is not None and getattr(S, "rider_post", None) is not None:
                args, self._post_handle, _ = S.rider_post(0, 1 + 2 * k)
                eng.arnoldi_step_post(A._h, V.data, k, self.w, self.gram.G.ref(0), self.gram.kmax, self.gram.W.ref(0), S.ref(0), args)
            else:
                eng.arnoldi_step(A._h, V.data, k, self.w, self.gram.G.ref(0), self.gram.kmax, self.gram.W.ref(0), S.ref(0))
            self.gram.in_G = k
            V.commit()
            return k
        A.apply(V[k - 1], out=self.w)
        if self.gram is not None:
            # the combined coefficients of both sweeps go to S[1 .. 1+k) (column k of H); S[1+k .. 1+2k) stays zero
            self.gram.sweep(k, self.w, 2, slot, sumsq=S.ref(0), c_out=S.ref(1))
        else:
            orthogonalize(eng, V, k, self.w, S, 1, passes=2, out=slot, sumsq=S.ref(0))
        eng.allreduce(S, 0, 1)
        eng.scale(Coef(1.0, den=S.ref(0), sqrt_den=True), slot, slot)
        V.commit()
        return k

    def _column(self, k, h):
        col = np.zeros(k + 1)
        col[:k] = h[1:1 + k] + h[1 + k:1 + 2 * k]
        col[k] = np.sqrt(h[0])
        self.Hcols.append(col)
        if getattr(self, "_Hfull", None) is not None and self._Hn == k - 1 and k < self._Hfull.shape[1]:
            self._Hfull[:k + 1, k - 1] = col
            self._Hn = k
        return col

    def step(self):
        k = self._enqueue()
        return self._column(k, self.S.host(0, 1 + 2 * k))

    def step_prefetch(self):
        """The step enqueued and the download of its column of H started; `absorb()` waits for that copy only (see
        GKState.step_prefetch).  Absorb a pending step before prefetching the next: both use the same scalars."""
        k, handle = self._enqueue(post=True)
        return k, (handle if handle is not None else self.S.host_later(0, 1 + 2 * k))

    def absorb(self, pending):
        k, handle = pending
        return self._column(k, handle.get())

    def H_view(self):
        """H_k as a read-only VIEW of the array the steps fill in place (Hybrid-GMRES looks at H_k every iteration: rebuilding it from
        its columns was O(k) Python statements per iteration); falls back to H() when the columns were set from outside."""
        k = len(self.Hcols)
        if getattr(self, "_Hfull", None) is None or self._Hn != k:
            return self.H()
        v = self._Hfull[:k + 1, :k]
        v.flags.writeable = False
        return v

    def H(self):
        k = len(self.Hcols)
        H = np.zeros((k + 1, k))
        for j, c in enumerate(self.Hcols):
            H[:j + 2, j] = c
        return H
