"""CGLS on the HIP engine — same signature, stopping rule and `info` keys as trips/solvers/CGLS.py:16-86.

Per iteration (all on the GPU; scalars never visit the host unless tol > 0):
    w = A p            ; delta = ||w||^2  fused into the operator kernel            (CGLS.py:60-61)
    x += (gamma/delta) p ; r -= (gamma/delta) w ; ||x||^2, ||x-x_old||^2, ||x-x_true||^2 in the same pass   (:64-67,76-80)
    t = A^T r          ; gamma' = ||t||^2 fused                                       (:68-70)
    p = t + (gamma'/gamma) p                                                          (:72)
Every iterate is written straight into its slot of an on-device history (the reference's xHistory, :66), so
keeping the history costs no extra traffic.  Sharded problems all-reduce 1 + 4 doubles per iteration.
"""
import numpy as np

from .._io import Formatter, History, as_operator
from ..engine import Coef


class _ShiftedRows:
    """A [rows, n] device block seen `shift` rows further on: what the C iteration loops take when iterate j is to land in
    ring slot (j + shift) instead of row j (they only ask for the base address and the row stride)."""

    def __init__(self, t, shift):
        self.t, self.shift = t, int(shift)

    def data_ptr(self):
        return self.t.data_ptr() + self.shift * self.t.stride(0) * self.t.element_size()

    def stride(self, dim):
        return self.t.stride(dim)

    def row(self, j):
        return self.t[j + self.shift]


def _row_of(X, j, keep):
    """Row of the block the C loop wrote iterate j (0-based) to."""
    if isinstance(X, _ShiftedRows):
        return X.row(j)
    return X[j] if keep else X[j & 1]


class CGLSRun:
    """The CGLS recurrence as an object: `step()` enqueues one iteration (no host sync), `rows()` downloads the
    per-iteration scalars.  `CGLS()` below and bench.py both drive this one implementation."""

    PCAP = 4096                            # room for an operator's raw block partials

    def __init__(self, A, b, x0, max_iter, x_true=None, history=True, defer_norms=False, grouping=None):
        self.A = A = as_operator(A)
        self.eng = eng = A.engine
        m, n = A.shape
        self.max_iter = max_iter = int(max_iter)
        self.bv = eng.to_vec(b, m)
        self.xt = None if x_true is None else eng.to_vec(x_true, n)
        x_start = eng.to_vec(x0, n)
        self.hist = History(eng, history, max_iter, n, "CGLS xHistory")       # True / False / stride / 'host' / '<file>.npy'
        self.keep = self.hist.keeps_any
        self.X = self.hist.X
        # tol = 0 on a single rank: nothing needs ||x||, ||dx||, ||x-x_true|| before the end -> keep them as block
        # partials and sum all iterations in one launch afterwards (one reduction-finalize launch less per iteration)
        self.defer = bool(defer_norms) and eng.world == 1 and hasattr(eng, "cgls_update_deferred")
        self.NP = eng.scalars(3 * 1024 * max_iter) if self.defer else None
        self.n_np = 0
        # operators that can leave ||A p||^2 / ||A^T r||^2 as raw block partials (the blur, the Radon projector): the consumers add them up and
        # the two reduction-finalize launches of the iteration disappear (six launches -> four)
        self.raw = bool(self.defer and hasattr(eng, "cgls_p_update") and hasattr(A, "_h") and eng.op_can_fuse(A._h))
        self.PG = eng.scalars(self.PCAP) if self.raw else None
        self.PD = eng.scalars(self.PCAP) if self.raw else None
        # raw form only: 0 = [x, r] / [p] update kernels, 1 = [r] / [x, p] (p read once); None: the library's rule by size
        self.grouping = (int(grouping) if grouping is not None else eng.cgls_update_grouping(n)) if self.raw else 0
        self._final = 0
        self.r, self.t, self.w, self.p = eng.empty(m), eng.empty(n), eng.empty(m), eng.empty(n)
        # scalar layout: S[0] = gamma_0 = ||t_0||^2 ; row k (1-based) at 5k: [delta, gamma, ||x||^2, ||dx||^2, ||x-xt||^2]
        self.S = S = eng.scalars(5 * (max_iter + 1))
        self.dist = eng.world > 1
        self.k = 0
        # r = b - A x0 ; t = A^T r ; p = t                                                (CGLS.py:45-47)
        A.apply(x_start, out=self.r)
        eng.axpby(1.0, self.bv, -1.0, self.r, self.r)
        A.apply(self.r, out=self.t, transpose=True, sumsq=S.ref(0))
        eng.allreduce(S, 0, 1)
        self.p.copy_(self.t)
        self.x_cur = x_start

    def slot(self, k):
        """The device row holding iterate k (0-based)."""
        return self.X[self.hist.slot(k)]

    def step(self):
        self._step()
        self.hist.pushed(self.k - 1)

    def _run_c_loop(self, n_steps, call):
        """`call(k_first, n, X, keep)` enqueues n iterations writing iterate j (1-based) to X + (j-1) rows (keep) — all at
        once, or, when the history is streamed through a ring of device slots, in chunks that stay inside one half of the
        ring (the copies of one half drain while the other half is being written)."""
        h = self.hist
        if h.mode != "stream":
            call(self.k + 1, n_steps, self.X, 1 if h.mode == "device" else 0)
            self.k += n_steps
        else:
            while n_steps > 0:
                k0 = self.k
                c = min(n_steps, h.R // 2, h.R - k0 % h.R)
                for j in range(c):
                    h.row(k0 + j)                                  # the slots' previous copies must be out
                call(k0 + 1, c, _ShiftedRows(self.X, k0 % h.R - k0), 1)
                for j in range(c):
                    h.pushed(k0 + j)
                self.k += c
                n_steps -= c
        self.x_cur = self.slot(self.k - 1)

    def _step(self):
        eng, A, S = self.eng, self.A, self.S
        self.k += 1
        k = self.k
        b = 5 * k                          # row k: delta, gamma, ||x||^2, ||dx||^2, ||x-xt||^2
        delta, gamma = S.ref(b), S.ref(b + 1)
        gamma_old = S.ref(0) if k == 1 else S.ref(b - 4)
        x_new = self.hist.row(k - 1)
        if self.raw and self.grouping == 1:
            # [r] after A p, [x, p] after A^T r: p is read once (trk_cgls_update_grouping: measured rule by size)
            n_d = eng.op_apply_fused(A._h, False, self.p, None, 0.0, None, 0, None, 0, None, self.w, self.PD.ref(0), self.PCAP)
            eng.cgls_r_update(gamma_old, self.PD.ref(0), n_d, self.r, self.w, delta)
            n_g = eng.op_apply_fused(A._h, True, self.r, None, 0.0, None, 0, None, 0, None, self.t, self.PG.ref(0), self.PCAP)
            self.n_np = eng.cgls_xp_update(gamma_old, delta, self.PG.ref(0), n_g, self.x_cur, self.p, self.t, x_new, self.xt,
                                           gamma, self.NP.ref(3 * self.n_np * (k - 1)), 1024)
            self.x_cur = x_new
            return
        if self.raw:
            n_d = eng.op_apply_fused(A._h, False, self.p, None, 0.0, None, 0, None, 0, None, self.w, self.PD.ref(0), self.PCAP)
            self.n_np = eng.cgls_update_src(gamma_old, 1, self.PD.ref(0), n_d, self.x_cur, self.p, x_new, self.r, self.w,
                                            self.xt, delta, self.NP.ref(3 * self.n_np * (k - 1)), 1024)
            n_g = eng.op_apply_fused(A._h, True, self.r, None, 0.0, None, 0, None, 0, None, self.t, self.PG.ref(0), self.PCAP)
            eng.cgls_p_update(self.t, self.p, self.PG.ref(0), n_g, gamma_old, gamma)
            self.x_cur = x_new
            return
        A.apply(self.p, out=self.w, sumsq=delta)
        if self.dist:
            eng.allreduce(S, b, b + 1)
        if self.defer:
            self.n_np = eng.cgls_update_deferred(gamma_old, delta, self.x_cur, self.p, x_new, self.r, self.w, self.xt,
                                                 self.NP.ref(3 * self.n_np * (k - 1)), 1024)
        else:
            eng.cgls_update(gamma_old, delta, self.x_cur, self.p, x_new, self.r, self.w, self.xt, S.ref(b + 2))
        A.apply(self.r, out=self.t, transpose=True, sumsq=gamma)
        if self.dist:
            eng.allreduce(S, b + 1, b + 5)
        eng.axpby(1.0, self.t, Coef(1.0, num=gamma, den=gamma_old), self.p, self.p)
        self.x_cur = x_new

    def run(self, n_steps):
        """Enqueue `n_steps` iterations.  With deferred norms on the HIP engine the whole stretch is one library call
        (trk_cgls_iterate: the same launches as `step()`, driven from C instead of the interpreter)."""
        n_steps = min(int(n_steps), self.max_iter - self.k)
        if n_steps <= 0:
            return
        eng = self.eng
        if self.defer and not self.dist and hasattr(self.A, "_h") and hasattr(eng, "cgls_iterate"):
            def call(k_first, n, X, keep):
                self.n_np = eng.cgls_iterate(self.A._h, k_first, n, self.p, self.r, self.t, self.w, X, keep,
                                             self.x_cur, self.xt, self.S.ref(0), self.NP.ref(0), 1024, self.n_np,
                                             None if not self.raw else self.PG.ref(0), None if not self.raw else self.PD.ref(0),
                                             self.PCAP if self.raw else 0, self.grouping)
                self.x_cur = _row_of(X, k_first + n - 2, keep)
            self._run_c_loop(n_steps, call)
        else:
            for _ in range(n_steps):
                self.step()

    def row(self, k):
        """[delta, gamma, ||x||^2, ||dx||^2, ||x-xt||^2] of iteration k (host sync)."""
        return self.S.host(5 * k, 5 * k + 5)

    def rows(self):
        if self.defer and self._final != self.k and self.k > 0:
            self.eng.finalize_batched(self.NP.ref(0), self.n_np, 3, self.k, self.S.ref(7), 5)
            self._final = self.k
        Sh = self.S.host()
        return Sh[0], Sh[5:5 * (self.k + 1)].reshape(self.k, 5)


class CGLSRunSharded(CGLSRun):
    """The recurrence with ONE all-reduce per iteration, for unknowns spread over ranks (frames of a dynamic problem, io.py:420)
    and tol = 0.  The reference's two global sums per iteration (CGLS.py:61 ||A p||^2, :70 ||A^T r||^2; the second waits for the
    first) become one exchange of FOUR doubles {gamma_{k-1}, ||q||^2, <q, w_{k-1}>, ||w_{k-1}||^2}: with q = A t_{k-1} formed
    explicitly, w_k = A p_k = q + beta w_{k-1} and delta_k = ||q||^2 + 2 beta <q, w_{k-1}> + beta^2 ||w_{k-1}||^2, the last norm
    taken from the stored vector rather than carried as beta^2 delta_{k-1} (include/trk.h, csrc/cgls_sharded.hip).
    `one_reduction=False` selects the two-reduction recurrence as written on ranks; `tiled=1` the four-blur tiled form on one
    rank — the two arrangements that refresh w = A p from a real product every iteration.  The three norms the
    reference records per iterate are only reported, so each rank keeps its share and they are summed over the ranks once, in
    `rows()`.  With libtrk's own communicator (dist.RcclComm) — or on one rank — the whole stretch is one library call
    (trk_cgls_iterate_sharded: the all-reduces are enqueued from C); with torch's communicator the same kernels are stepped from
    Python.  `allreduces` counts the exchanges issued inside the iterations."""

    NPC = 1024

    def __init__(self, A, b, x0, max_iter, x_true=None, history=True):
        self.A = A = as_operator(A)
        self.eng = eng = A.engine
        m, n = A.shape
        self.max_iter = max_iter = int(max_iter)
        self.bv = eng.to_vec(b, m)
        self.xt = None if x_true is None else eng.to_vec(x_true, n)
        x_start = eng.to_vec(x0, n)
        self.hist = History(eng, history, max_iter, n, "CGLS xHistory")
        self.keep = self.hist.keeps_any
        self.X = self.hist.X
        self.defer, self.raw, self.grouping, self._final = False, False, 0, 0
        self.NP = eng.scalars(3 * self.NPC * max_iter)
        self.n_np = 0
        self.r, self.t, self.w, self.p, self.q = eng.empty(m), eng.empty(n), eng.zeros(m), eng.zeros(n), eng.empty(m)
        self.S = eng.scalars(5 * (max_iter + 1))
        self.G = eng.scalars(4)                       # [gamma_{k-1}, ||q||^2, <q, w_{k-1}>, ||w_{k-1}||^2]: an iteration's one exchange
        self.dist = eng.world > 1
        self.k = 0
        self.allreduces = 0
        self._rows = None
        # operators that can leave ||A^T r||^2 as raw block partials (the Radon projector, the blur): the kernel that forms the
        # iteration's other sums adds them up too — no reduction-finalize launch behind the adjoint apply (trk_cgls_sharded_scalars)
        self.rawg = bool(hasattr(eng, "cgls_sharded_scalars") and hasattr(A, "_h") and eng.op_can_fuse(A._h))
        self.PG = eng.scalars(self.PCAP) if self.rawg else None
        self.n_g = 0
        A.apply(x_start, out=self.r)                                         # r = b - A x0 ; t = A^T r          (CGLS.py:45-46)
        eng.axpby(1.0, self.bv, -1.0, self.r, self.r)
        self._adjoint()                                                      # this rank's ||t_0||^2: summed by iteration 1
        self.x_cur = x_start

    def _adjoint(self):
        """t = A^T r with this rank's ||t||^2 as raw block partials in PG (n_g of them) or, without a fused apply, finished in G[0]."""
        if self.rawg:
            self.n_g = self.eng.op_apply_fused(self.A._h, True, self.r, None, 0.0, None, 0, None, 0, None, self.t, self.PG.ref(0),
                                               self.PCAP)
        else:
            self.A.apply(self.r, out=self.t, transpose=True, sumsq=self.G.ref(0))

    def _step(self):
        eng, A, S, G = self.eng, self.A, self.S, self.G
        self.k += 1
        k = self.k
        b = 5 * k
        x_new = self.hist.row(k - 1)
        A.apply(self.t, out=self.q)                                          # q = A t_{k-1}
        if self.rawg:
            eng.cgls_sharded_scalars(self.q, None if k == 1 else self.w, self.PG.ref(0), self.n_g, G.ref(0))
        else:
            eng.dot_pair(self.q, None if k == 1 else self.w, G.ref(1))
        eng.allreduce(G, 0, 4)               # the iteration's one exchange (a no-op on one rank)
        self.allreduces += self.dist
        self.n_np = eng.cgls_sharded_update(G.ref(0), S.ref(0) if k <= 2 else S.ref(b - 9), k == 1, self.x_cur, self.p,
                                            self.t, x_new, self.r, self.q, self.w, self.xt, S.ref(b),
                                            S.ref(0) if k == 1 else S.ref(b - 4), self.NP.ref(3 * self.n_np * (k - 1)), self.NPC)
        self._adjoint()                                                      # t_k = A^T r_k, this rank's ||t_k||^2
        self.x_cur = x_new

    def run(self, n_steps):
        n_steps = min(int(n_steps), self.max_iter - self.k)
        if n_steps <= 0:
            return
        eng = self.eng
        comm_h = getattr(eng.comm, "_h", None) if self.dist else None       # libtrk's own communicator (dist.RcclComm)
        if hasattr(eng, "cgls_iterate_sharded") and hasattr(self.A, "_h") and (not self.dist or comm_h is not None):
            def call(k_first, n, X, keep):
                self.n_np, self.n_g = eng.cgls_iterate_sharded(self.A._h, comm_h, k_first, n, self.p, self.r, self.t, self.q, self.w, X,
                                                               keep, self.x_cur, self.xt, self.S.ref(0), self.G.ref(0), self.NP.ref(0),
                                                               self.NPC, self.n_np, None if not self.rawg else self.PG.ref(0),
                                                               self.PCAP if self.rawg else 0, self.n_g)
                self.x_cur = _row_of(X, k_first + n - 2, keep)
                eng.reduction_points += n
                if self.dist:
                    self.allreduces += n
            self._run_c_loop(n_steps, call)
        else:
            for _ in range(n_steps):
                self.step()

    def rows(self):
        """(gamma_0, [delta, gamma, ||x||^2, ||dx||^2, ||x - x_true||^2] per iteration): the rank's norm shares are summed over
        blocks, then — with the last gamma, still local — over the ranks: one exchange per solve."""
        k = self.k
        if k == 0:
            return self.S.host(0, 1)[0], np.zeros((0, 5))
        if self._rows is None or self._rows[0] != k:
            eng = self.eng
            N3 = eng.scalars(3 * k + 1)
            eng.finalize_batched(self.NP.ref(0), self.n_np, 3, k, N3.ref(0), 3)
            if self.rawg:                                             # gamma_k: formed by the last A^T r, not yet summed / exchanged
                eng.finalize_batched(self.PG.ref(0), self.n_g, 1, 1, N3.ref(3 * k), 1)
            else:
                N3.set(3 * k, self.G.host(0, 1))
            eng.allreduce(N3, 0, 3 * k + 1)
            Sh, Nh = self.S.host(), N3.host()
            rows = Sh[5:5 * (k + 1)].reshape(k, 5).copy()
            rows[:, 2:5] = Nh[:3 * k].reshape(k, 3)
            rows[k - 1, 1] = Nh[3 * k]
            self._rows = (k, Sh[0], rows)
        return self._rows[1], self._rows[2]

    def row(self, k):
        return self.rows()[1][k - 1]


class CGLSRunFused(CGLSRun):
    """The same recurrence in three launches per iteration and no reduction-finalize launches, for operators with a
    fused apply (trk_op_apply_fused: the separable blur):

        K1  w = A (t + (gamma_k/gamma_{k-1}) p_old)  [p_new written out]   + ||w||^2 partials
        K2  x += (gamma/delta) p_new                                        + norm partials; publishes delta, gamma
        K3  t = A^T (r_old - (gamma/delta) w)        [r_new written out]   + ||t||^2 partials

    p and r are double-buffered (a band of the blur kernel reads its neighbours' halo rows of the old vector while they
    write the new one).  11 vector passes per iteration instead of 13, 3 launches instead of 7.  Single rank, tol = 0
    (nothing visits the host until the end)."""

    PCAP = 4096                            # room for the producers' block partials
    # measured (MI355X, 9x9 blur, iterations/s fused vs the four-launch generic form whose consumers add up the block
    # partials): 512^2 36.3 k vs 40.2 k, 1024^2 28.3 k vs 30.0 k, 1536^2 21.9 k vs 24.2 k | 1792^2 21.5 k vs 20.8 k,
    # 2048^2 18.7 k vs 18.1 k, 2560^2 14.7 k vs 13.9 k | 2816^2 11.9 k vs 12.3 k, 3072^2 10.3 k vs 10.7 k, 4096^2 6.2 k vs
    # 6.9 k: the two-operand blur kernel pays only in the middle range, so CGLS() picks it there unless told otherwise
    AUTO_MIN_N = 3 * 2 ** 20
    AUTO_MAX_N = 7 * 2 ** 20

    @classmethod
    def auto(cls, n):
        return cls.AUTO_MIN_N <= n <= cls.AUTO_MAX_N

    @staticmethod
    def usable(A, eng):
        return (getattr(eng, "is_native", False) and eng.world == 1 and hasattr(A, "_h") and A.shape[0] == A.shape[1]
                and eng.op_can_fuse(A._h) == 1)

    # Small images (tiled = True): the same recurrence in TWO launches per iteration (trk_cgls_iterate_tiled: a workgroup per
    # 32 x 32 tile recomputes its halo of p and w in LDS instead of waiting for its neighbours at a kernel boundary).
    # Measured crossover against the streaming forms: see TILED_MAX_N.
    TILED_MAX_N = 1 << 20
    TILED_DEFAULT = 2                      # which tiled form CGLS() picks: 2 (two blurs per iteration, trk_cgls_iterate_tiled2: faster at
                                           # every size, tools/cgls_small_sizes.py) or 1 (four; trk_cgls_iterate_tiled)

    @classmethod
    def tiled_usable(cls, A, eng):
        return (cls.usable(A, eng) and hasattr(eng, "cgls_tiled_caps") and A.shape[1] <= cls.TILED_MAX_N
                and eng.cgls_tiled_caps(A._h, 1024, cls.PCAP))

    def __init__(self, A, b, x0, max_iter, x_true=None, history=True, tiled=False):
        self.A = A = as_operator(A)
        self.eng = eng = A.engine
        m, n = A.shape
        self.tiled = int(tiled)            # 0: streaming kernels; 1: tiled, four blurs per iteration; 2: tiled, two (w by recurrence)
        self.max_iter = max_iter = int(max_iter)
        self.bv = eng.to_vec(b, m)
        self.xt = None if x_true is None else eng.to_vec(x_true, n)
        x_start = eng.to_vec(x0, n)
        self.hist = History(eng, history, max_iter, n, "CGLS xHistory")
        self.keep = self.hist.keeps_any
        self.X = self.hist.X
        self.R = eng.empty_basis(2, m)     # r ping-pong
        self.P = eng.empty_basis(2, n)     # p ping-pong
        if self.tiled != 2:                # (the two-blur tiled form does not read p or w in its first iteration)
            self.P.zero_()                 # the first K1 multiplies p_old by 0: it must be finite
        self.t, self.w = eng.empty(n), eng.empty(m)
        self.S = S = eng.scalars(5 * (max_iter + 1))
        uninit = getattr(eng, "scalars_uninit", eng.scalars)
        self.PG = uninit(self.PCAP)        # ||t||^2 partials (gamma): the counted entries are written before they are read
        self.PD = uninit(self.PCAP)        # ||w||^2 partials (delta)
        self.NP = uninit(3 * 1024 * max_iter)        # norm partials of every iteration, summed once at the end
        self.dist = False
        self.k = 0
        self.n_g = self.n_np = 0
        # r0 = b - A x0 ; t0 = A^T r0 with raw ||t0||^2 partials: K3 form with x1 = b, x2 = A x0, cb = -1
        A.apply(x_start, out=self.w)
        self.n_g = eng.op_apply_fused(A._h, True, self.bv, self.w, -1.0, None, 0, None, 0, self.R[0], self.t,
                                      self.PG.ref(0), self.PCAP)
        self.x_cur = x_start
        self._final = False

    def _step(self):
        eng, A, S = self.eng, self.A, self.S
        if self.tiled:
            x_new = self.hist.row(self.k)
            keep = self.hist.mode == "device"
            X = self.X if self.hist.mode != "stream" else _ShiftedRows(self.X, self.hist.slot(self.k) - self.k)
            self._tiled_call(self.k + 1, 1, X, keep or self.hist.mode == "stream")
            self.k += 1
            self.x_cur = x_new
            return
        self.k += 1
        k = self.k
        b = 5 * k
        p_old, p_new = self.P[(k - 1) & 1], self.P[k & 1]
        r_old, r_new = self.R[(k - 1) & 1], self.R[k & 1]
        gprev = S.ref(0) if k <= 2 else S.ref(5 * (k - 2) + 1)     # gamma_{k-2}, published by K2 of iteration k-1
        # K1: p_k = t + (gamma_{k-1}/gamma_{k-2}) p_{k-1} ; w = A p_k
        n_d = eng.op_apply_fused(A._h, False, self.t, p_old, 0.0 if k == 1 else 1.0, self.PG.ref(0), self.n_g, gprev, 1,
                                 p_new, self.w, self.PD.ref(0), self.PCAP)
        # K2: x_k = x_{k-1} + (gamma_{k-1}/delta_k) p_k ; publishes delta_k -> S[5k], gamma_{k-1} -> S[5(k-1)+1] (S[0] for k = 1)
        x_new = self.hist.row(k - 1)
        gpub = S.ref(0) if k == 1 else S.ref(b - 4)
        self.n_np = eng.cgls_x_update(self.PG.ref(0), self.n_g, self.PD.ref(0), n_d, self.x_cur, p_new, x_new, self.xt,
                                      S.ref(b), gpub, self.NP.ref(3 * self.n_np * (k - 1)), 1024)
        # K3: r_k = r_{k-1} - (gamma_{k-1}/delta_k) w ; t = A^T r_k ; ||t||^2 partials = gamma_k
        self.n_g = eng.op_apply_fused(A._h, True, r_old, self.w, -1.0, gpub, 1, S.ref(b), 1, r_new, self.t,
                                      self.PG.ref(0), self.PCAP)
        self.x_cur = x_new

    def _tiled_call(self, k_first, n, X, keep):
        eng = self.eng
        if self.tiled == 2:
            self.n_g, self.n_np = eng.cgls_iterate_tiled2(self.A._h, k_first, n, self.P[0], self.w, self.R, self.t, X, keep,
                                                          self.x_cur, self.xt, self.S.ref(0), self.PG.ref(0), self.PD.ref(0),
                                                          self.PCAP, self.NP.ref(0), 1024, self.n_g, self.n_np)
        else:
            self.n_g, self.n_np = eng.cgls_iterate_tiled(self.A._h, k_first, n, self.P, self.R, self.t, X, keep, self.x_cur, self.xt,
                                                         self.S.ref(0), self.PG.ref(0), self.PD.ref(0), self.PCAP, self.NP.ref(0),
                                                         1024, self.n_g, self.n_np)

    def run(self, n_steps):
        """Enqueue `n_steps` iterations in one library call (trk_cgls_iterate_fused)."""
        n_steps = min(int(n_steps), self.max_iter - self.k)
        if n_steps <= 0:
            return
        def call(k_first, n, X, keep):
            if self.tiled:
                self._tiled_call(k_first, n, X, keep)
            else:
                self.n_g, self.n_np = self.eng.cgls_iterate_fused(self.A._h, k_first, n, self.P, self.R, self.t, self.w,
                                                                  X, keep, self.x_cur, self.xt, self.S.ref(0),
                                                                  self.PG.ref(0), self.PD.ref(0), self.PCAP, self.NP.ref(0), 1024,
                                                                  self.n_g, self.n_np)
            self.x_cur = _row_of(X, k_first + n - 2, keep)
        self._run_c_loop(n_steps, call)

    def _finish(self):
        """Sum the norm partials of all iterations (one launch) and the last gamma (one launch)."""
        if self.k == 0 or self._final == self.k:
            return
        eng, S, k = self.eng, self.S, self.k
        eng.finalize_batched(self.NP.ref(0), self.n_np, 3, k, S.ref(7), 5)        # -> S[5i+2 .. 5i+4], i = 1..k
        eng.finalize_batched(self.PG.ref(0), self.n_g, 1, 1, S.ref(5 * k + 1), 1)  # gamma_k
        self._final = k

    def row(self, k):
        raise RuntimeError("the fused CGLS path publishes its norms after the solve (tol = 0 only)")

    def rows(self):
        self._finish()
        Sh = self.S.host()
        return Sh[0], Sh[5:5 * (self.k + 1)].reshape(self.k, 5)


def CGLS(A, b, x0, max_iter, tol, x_true=None, **kwargs):
    """Conjugate Gradient Least Squares.

    A: LinearOperator; b: (m,) / (m,1); x0: (n,) / (n,1); returns (x, info) with info keys
    xHistory, regParam (empty), relResidual, its and, if x_true is given, relError (sic: divided by ||x||, :79).
    Engine-only kwarg: history — True (every iterate on the device, like the reference's x_history), False, a stride s
    (every s-th iterate), "host" (every iterate streamed to host memory through a ring of device slots) or a path ending in
    .npy (the same into a memory-mapped file): trips_py_amd._io.History.
    """
    if int(max_iter) <= 0:
        # the reference would fail at `shrink = norm_x/xmax` (:84) with norm_x undefined
        raise UnboundLocalError("CGLS with max_iter <= 0: the reference leaves norm_x undefined (CGLS.py:84)")
    fmt = Formatter(b)
    A = as_operator(A)
    sync_each = (tol != 0)
    want = kwargs.get("fused", None)          # None: automatic by size; True / False: forced
    fused = (not sync_each) and CGLSRunFused.usable(A, A.engine) and \
        (want if want is not None else CGLSRunFused.auto(A.shape[1]))
    tiled = (not sync_each) and want is None and kwargs.get("tiled", True) and CGLSRunFused.tiled_usable(A, A.engine)
    if tiled:
        # tiled=True (what earlier versions took) means "the default tiled form", not form 1
        tk = kwargs.get("tiled", True)
        tiled = CGLSRunFused.TILED_DEFAULT if tk is True else int(tk)       # 1: four blurs per iteration, 2: two (w by recurrence)
    # unknowns spread over ranks, tol = 0: the one-all-reduce form (one_reduction=False keeps the two reductions of the recurrence
    # as written; one_reduction=True also selects it on a single rank, where it is the same arithmetic without the exchange)
    one_red = kwargs.get("one_reduction", None)
    sharded = (not sync_each) and (one_red if one_red is not None else A.engine.world > 1) and \
        hasattr(A.engine, "cgls_sharded_update")
    if sharded:
        run = CGLSRunSharded(A, b, x0, max_iter, x_true, kwargs.get("history", True))
    elif tiled or fused:
        run = CGLSRunFused(A, b, x0, max_iter, x_true, kwargs.get("history", True), tiled=tiled)
    else:
        run = CGLSRun(A, b, x0, max_iter, x_true, kwargs.get("history", True), defer_norms=not sync_each)
    nt0 = None
    stop = False
    if not sync_each:
        run.run(run.max_iter)                   # tol = 0: nothing is read back inside the loop -> one enqueue
    while run.k < run.max_iter and not stop:
        run.step()
        if sync_each:
            if nt0 is None:
                nt0 = float(np.sqrt(run.S.host(0, 1)[0]))
            h = run.row(run.k)
            stop = (np.sqrt(h[1]) <= nt0 * tol) or (np.sqrt(h[2]) * tol >= 1)            # (:73-75)
    _g0, rows = run.rows()
    k, x_fin = run.k, run.x_cur
    if not sync_each:
        # tol == 0: the reference stops only when ||t|| is exactly zero (:75); honour that after the fact
        zero = np.nonzero(rows[:, 1] <= 0.0)[0]
        if zero.size and int(zero[0]) + 1 < k:
            k = int(zero[0]) + 1
            if run.hist.mode == "device":
                rows, x_fin = rows[:k], run.slot(k - 1)
            else:
                # the iterate of the breakdown step is gone (no / streamed history) and everything after it is 0/0: solve
                # again for exactly k iterations (deterministic: the same iterates)
                return CGLS(A, b, x0, k, tol, x_true, **kwargs)
    norm_x = np.sqrt(rows[:, 2])
    info = {"xHistory": run.hist.collect(fmt, k), "regParam": [],
            "relResidual": list(np.sqrt(rows[:, 3]) / norm_x), "its": k}
    if run.xt is not None:
        info["relError"] = list(np.sqrt(rows[:, 4]) / norm_x)
    if sharded:
        info["allreduces_per_iteration"] = run.allreduces / max(1, run.k)      # engine-only key: exchanges inside the iterations
    return fmt.vec(x_fin), info
