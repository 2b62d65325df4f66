"""HipEngine — the Python face of libtrk.so: device vectors are fp32 torch tensors, device scalars are
float64 torch tensors, every call goes through the C ABI (include/trk.h) on torch's current HIP stream.

PyTorch is plumbing here (device memory, streams, torch.distributed); all arithmetic on vectors is done by the
hand-written HIP kernels.  There is no CPU fallback: constructing a HipEngine without a visible GPU or without
libtrk.so raises.

Reductions leave LOCAL sums in device doubles; `allreduce()` turns them into global sums when the problem is
sharded across ranks (one process per GPU, RCCL through torch.distributed), and is a no-op otherwise.
"""
import ctypes

import numpy as np
import torch

from . import _lib

SQRT_NUM = 1
SQRT_DEN = 2


class Coef:
    """A coefficient evaluated ON THE DEVICE when the kernel runs:  c * f(num) / g(den).

    `num` / `den` are one-element views of a float64 device tensor (or None = 1); sqrt_* apply a square root to the
    loaded value (norms are stored squared)."""

    __slots__ = ("c", "num", "den", "flags", "_keep")

    def __init__(self, c=1.0, num=None, den=None, sqrt_num=False, sqrt_den=False):
        self.c = float(c)
        self.num = _ptr(num)
        self.den = _ptr(den)
        self.flags = (SQRT_NUM if sqrt_num else 0) | (SQRT_DEN if sqrt_den else 0)
        self._keep = (num, den)


# torch's C entry point for "current stream of device i" (what torch.cuda.current_stream() wraps): ~0.2 us instead of
# ~3 us through the Python Stream object, which matters when an iteration is a handful of 5-40 us kernels
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
if _raw_stream is None:                                   # pragma: no cover  (older / different torch builds)
    def _raw_stream(index):
        return torch.cuda.current_stream(index).cuda_stream


def _ptr(t):
    if t is None:
        return None
    if isinstance(t, int):
        return t
    if hasattr(t, "data_ptr"):
        return t.data_ptr()
    return t          # an opaque scalar reference of another engine implementation


class DevScalars:
    """A block of float64 device scalars.  `ref(i)` is the opaque handle kernels take for scalar i (here: its device
    address as a plain int, so the hot loops create no tensor views); `view(i, j)` is a tensor view for collectives
    and downloads."""

    __slots__ = ("t", "base", "_pin", "_eng", "_mb", "_mb_np", "_mb_slot", "__weakref__")

    MAILBOX_SLOTS = 32

    def __init__(self, t, eng=None):
        self.t = t
        self.base = t.data_ptr()
        self._pin = None
        self._eng, self._mb, self._mb_np, self._mb_slot = eng, None, None, 0

    def ref(self, i):
        return self.base + 8 * i

    def view(self, i=0, j=None):
        return self.t[i:j]

    HOST_BY_MAILBOX_MAX = 256          # (measured: MMGKS's 2 k^2 + 2 k Gram doubles per iteration travel no faster by mailbox than by tensor copy)

    def host(self, i=0, j=None):
        """A blocking download of scalars [i, j) as a float64 array.  Up to HOST_BY_MAILBOX_MAX doubles go through the block's mailbox
        (one small launch that copies into pinned memory, then a poll: ~10 us) — the tensor route below (`.to('cpu')`: a staged copy and a
        stream synchronisation) costs ~150 us, which GKS / MMGKS with an automatic lambda paid once per iteration."""
        n = self.t.numel()
        i0 = 0 if i is None else (i + n if i < 0 else i)
        j0 = n if j is None else (j + n if j < 0 else min(j, n))
        if self._eng is not None and 0 < j0 - i0 <= self.HOST_BY_MAILBOX_MAX and getattr(self._eng, "lib", None) is not None \
                and hasattr(self._eng.lib, "trk_mailbox_post") and self.t.is_cuda:
            return self.host_later(i0, j0).get()
        return self.t[i:j].detach().to("cpu").numpy().astype(np.float64, copy=False)

    def host_later(self, i, j):
        """Start the download of scalars [i, j) at this point of the stream and return a handle; `get()` waits for THAT
        copy only, so work enqueued in between overlaps with whatever the host does before it asks."""
        eng = self._eng
        if eng is not None:
            # trk_mailbox: one hipMemcpyAsync + one hipEventRecord; the tensor-copy + Event-object route below costs the host
            # ~12 us per download, a tenth of a Hybrid-LSQR iteration at 512^2
            self._ensure_mailbox()
            slot = self._mb_slot
            self._mb_slot = (slot + 1) % self.MAILBOX_SLOTS
            _lib.check(eng.lib.trk_mailbox_post(self._mb, slot, self.base + 8 * i, int(i), int(j - i), eng.stream()), "trk_mailbox_post")
            return _Posted(eng.lib, self._mb, slot, self._mb_np, i, j, self)
        if self._pin is None:
            self._pin = torch.empty(self.t.numel(), dtype=torch.float64, pin_memory=True)
        self._pin[i:j].copy_(self.t[i:j], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        return _Pending(ev, self._pin, i, j)

    def _ensure_mailbox(self):
        """This block's mailbox, taken from the engine's pool on first use — nothing is posted and nothing is waited for."""
        if self._mb is None:
            import weakref
            # borrowed from the engine's pool and handed back when this block dies: pinned memory is never freed while the
            # engine lives (hipHostFree synchronises the device and takes milliseconds — run by the garbage collector in the
            # middle of somebody else's solve it cost the bench's C5 CGLS 85 ms)
            eng = self._eng
            cap = 64
            while cap < self.t.numel():
                cap *= 2
            self._mb, self._mb_np = eng._mailbox_take(cap, self.MAILBOX_SLOTS)
            weakref.finalize(self, eng._mailbox_give, cap, self._mb, self._mb_np)

    def host_later_sum(self, i, j, partials, n_partials, at):
        """host_later(i, j) that also publishes the sum of `n_partials` block partials (device address `partials`) as scalar `at`
        of this block (outside [i, j)), on the device and on the host, with the same launch (trk_mailbox_post_sum).
        Returns two handles: the copied range and the one summed value."""
        eng = self._eng
        self._ensure_mailbox()                 # (no post, no wait: a hidden host synchronisation per solve otherwise)
        slot = self._mb_slot
        self._mb_slot = (slot + 1) % self.MAILBOX_SLOTS
        _lib.check(eng.lib.trk_mailbox_post_sum(self._mb, slot, self.base + 8 * i, int(i), int(j - i), _ptr(partials), int(n_partials),
                                                self.base + 8 * at, int(at), eng.stream()), "trk_mailbox_post_sum")
        return (_Posted(eng.lib, self._mb, slot, self._mb_np, i, j, self), _Posted(eng.lib, self._mb, slot, self._mb_np, at, at + 1, self))

    def rider_post(self, i, j, partials=None, n_partials=0, at=None):
        """The arguments of a post of scalars [i, j) (and, with `partials`, of their sum as scalar `at`) that a kernel is to carry
        (trk_gk_step_post), and the handles to collect it with: ((mailbox, slot, src, offset, count, partials, n, sum_dev,
        sum_offset), range handle, sum handle or None)."""
        eng = self._eng
        self._ensure_mailbox()
        slot = self._mb_slot
        self._mb_slot = (slot + 1) % self.MAILBOX_SLOTS
        args = (self._mb, slot, self.base + 8 * i, int(i), int(j - i), _ptr(partials), int(n_partials),
                None if at is None else self.base + 8 * at, 0 if at is None else int(at))
        h = _Posted(eng.lib, self._mb, slot, self._mb_np, i, j, self)
        hs = None if at is None else _Posted(eng.lib, self._mb, slot, self._mb_np, at, at + 1, self)
        return args, h, hs

    def set(self, i, values):
        a = np.ascontiguousarray(np.atleast_1d(np.asarray(values, dtype=np.float64)).reshape(-1))
        eng = self._eng
        if eng is not None and a.size <= 4096:
            # in the arguments of a one-wave launch (trk_scalars_put): stream-ordered like the copy below, but no staging of
            # pageable memory and no synchronisation — GKS / MMGKS with automatic lambda upload a k-vector per iteration
            if i < 0 or i + a.size > self.t.numel():
                raise IndexError("DevScalars.set: range outside the block")
            _lib.check(eng.lib.trk_scalars_put(self.base + 8 * i, a.ctypes.data, int(a.size), eng.stream()), "trk_scalars_put")
            return
        v = torch.as_tensor(a)
        self.t[i:i + v.numel()].copy_(v, non_blocking=False)

    def __len__(self):
        return self.t.numel()

    # tensor-like conveniences used by tests
    def __getitem__(self, k):
        return self.t[k]

    def __setitem__(self, k, v):
        self.t[k] = v

    def data_ptr(self):
        return self.base


class _Posted:
    # `owner`: the DevScalars whose mailbox this handle reads — kept alive, so that the mailbox cannot go back to the engine's pool
    # (and be handed to another block) while a posted download is still to be collected
    __slots__ = ("lib", "mb", "slot", "host", "i", "j", "owner")

    def __init__(self, lib, mb, slot, host, i, j, owner=None):
        self.lib, self.mb, self.slot, self.host, self.i, self.j, self.owner = lib, mb, slot, host, i, j, owner

    def get(self):
        _lib.check(self.lib.trk_mailbox_wait(self.mb, self.slot), "trk_mailbox_wait")
        return self.host[self.i:self.j].copy()


class _Pending:
    __slots__ = ("ev", "pin", "i", "j")

    def __init__(self, ev, pin, i, j):
        self.ev, self.pin, self.i, self.j = ev, pin, i, j

    def get(self):
        self.ev.synchronize()
        return self.pin[self.i:self.j].numpy().copy()


def _as_coef(a):
    return a if isinstance(a, Coef) else Coef(a)


def host_cpu_budget():
    """CPUs this process may actually use: its affinity mask, cut down to the cgroup's CPU quota (cpu.max / cfs_quota_us)."""
    import math
    import os
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:                                      # pragma: no cover
        n = os.cpu_count() or 1
    for quota_file, period_file in (("/sys/fs/cgroup/cpu.max", None),
                                    ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us")):
        try:
            if period_file is None:
                q, p = open(quota_file).read().split()[:2]
            else:
                q, p = open(quota_file).read().strip(), open(period_file).read().strip()
            if q not in ("max", "-1") and int(p) > 0:
                n = min(n, max(1, math.ceil(int(q) / int(p))))
            break
        except (OSError, ValueError):
            continue
    return n


_host_pools_checked = False


def tame_host_thread_pools():
    """Once per process: if the host thread pools (torch's intra-op pool, the BLAS behind NumPy) are larger than the CPU quota
    of the container, shrink them to half of it.  Measured on the MI355X box (256 logical CPUs visible, cgroup quota 16 CPUs per
    100 ms): any CPU-side tensor op or LAPACK call wakes a 128-thread pool whose idle spinning exhausts the quota, the kernel then
    parks EVERY thread of the container until the period ends, and a solve that is enqueueing 5-us kernels stops for 70-95 ms
    (four of twelve C5 solves in `tools/c5_stall_hunt3.py`; none with 8 threads).  TRK_KEEP_HOST_THREADS=1 leaves the pools alone."""
    global _host_pools_checked
    import os
    if _host_pools_checked or os.environ.get("TRK_KEEP_HOST_THREADS"):
        return
    _host_pools_checked = True
    budget = host_cpu_budget()
    try:
        ranks_here = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))       # one process per GPU: the quota is shared
    except ValueError:
        ranks_here = 1
    budget = max(1, budget // ranks_here)
    want = max(1, budget // 2)
    changed = []
    try:
        if torch.get_num_threads() > budget:
            changed.append(f"torch intra-op threads {torch.get_num_threads()} -> {want}")
            torch.set_num_threads(want)
    except RuntimeError:                                        # pragma: no cover
        pass
    try:
        from threadpoolctl import threadpool_info, threadpool_limits
        big = [p for p in threadpool_info() if int(p.get("num_threads", 1)) > budget]
        if big:
            changed += [f"{p.get('internal_api', p.get('user_api', 'pool'))} threads {p.get('num_threads')} -> {want}" for p in big]
            threadpool_limits(limits=want)                      # (called, not entered: stays in force)
    except Exception:                                           # pragma: no cover  (threadpoolctl missing)
        pass
    if changed:
        # a library changing process-wide state says so, once
        import sys
        print(f"[trips_py_amd] host thread pools exceed this process's CPU budget ({budget} CPUs: affinity / cgroup quota"
              f"{', shared by ' + str(ranks_here) + ' local ranks' if ranks_here > 1 else ''}); shrunk for the life of the process: "
              + "; ".join(changed) + ".  TRK_KEEP_HOST_THREADS=1 leaves them alone (DESIGN.md 6.1).", file=sys.stderr)


class HipEngine:
    """One per (device, communicator).  All methods enqueue on torch's current stream and return immediately."""

    is_native = True

    def __init__(self, device=None, comm=None):
        if not torch.cuda.is_available():
            raise _lib.TrkError("HipEngine needs a visible GPU (torch.cuda.is_available() is False); "
                                "the engine has no CPU fallback")
        self.lib = _lib.load()
        tame_host_thread_pools()
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self._dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        if self.device.type != "cuda" or self._dev_index != torch.cuda.current_device():
            # libtrk allocates operator tables / scratch on the process's CURRENT device (one process per GPU is the
            # deployment model): an engine on any other device would put them on the wrong GPU
            raise ValueError(f"HipEngine(device={self.device}) must be the current device "
                             f"(cuda:{torch.cuda.current_device()}); call torch.cuda.set_device() first")
        self.comm = comm
        self.world = 1 if comm is None else comm.world
        self.rank = 0 if comm is None else comm.rank
        self.reduction_points = 0           # calls of allreduce() (+ the all-reduces a C iteration loop enqueues itself)
        self.halo_exchanges = 0             # neighbour exchanges of the time-sharded space-time regulariser (counted on one rank too)

    # ------------------------------------------------------------------ memory
    def empty(self, n):
        return torch.empty(int(n), dtype=torch.float32, device=self.device)

    def zeros(self, n):
        return torch.zeros(int(n), dtype=torch.float32, device=self.device)

    def empty_basis(self, k, n):
        """Row-per-vector basis storage [k, n] (a new vector is a write, never a re-copy)."""
        return torch.empty((int(k), int(n)), dtype=torch.float32, device=self.device)

    def scalars(self, n):
        return DevScalars(torch.zeros(int(n), dtype=torch.float64, device=self.device), self)

    def scalars_uninit(self, n):
        """A block every entry of which is written before it is read (block partials): no zero-fill launch."""
        return DevScalars(torch.empty(int(n), dtype=torch.float64, device=self.device), self)

    def _mailbox_take(self, cap, slots):
        """A trk_mailbox of `cap` doubles from the pool (created on demand), with the NumPy view of its pinned block."""
        pool = self.__dict__.setdefault("_mailbox_pool", {})
        free = pool.setdefault(cap, [])
        if free:
            return free.pop()
        mb = ctypes.c_void_p()
        _lib.check(self.lib.trk_mailbox_create(int(cap), int(slots), ctypes.byref(mb)), "trk_mailbox_create")
        hp = ctypes.c_void_p()
        _lib.check(self.lib.trk_mailbox_host(mb, ctypes.byref(hp)), "trk_mailbox_host")
        return mb, np.ctypeslib.as_array(ctypes.cast(hp, ctypes.POINTER(ctypes.c_double)), shape=(int(cap),))

    MAILBOX_POOL_MAX = 32          # free mailboxes kept per size (a solver holds 1-3 at a time): beyond that they are destroyed

    def _mailbox_give(self, cap, mb, view):
        free = self.__dict__.setdefault("_mailbox_pool", {}).setdefault(cap, [])
        if len(free) < self.MAILBOX_POOL_MAX:
            free.append((mb, view))
        else:                          # bounded pinned memory: pays the hipHostFree the pool exists to avoid, in pathological use only
            try:
                self.lib.trk_mailbox_destroy(mb)
            except Exception:          # noqa: BLE001  (interpreter shutdown)
                pass

    def to_vec(self, a, n=None):
        """numpy / torch, shape (n,), (n,1) -> contiguous fp32 device vector (a copy unless already one)."""
        if isinstance(a, torch.Tensor):
            t = a.detach().reshape(-1).to(device=self.device, dtype=torch.float32).contiguous()
        else:
            t = torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.float32).reshape(-1))).to(self.device)
        if n is not None and t.numel() != n:
            raise ValueError(f"vector has {t.numel()} entries, expected {n}")
        return t

    def to_host(self, t):
        """Device doubles -> numpy float64 (synchronises the current stream)."""
        if isinstance(t, DevScalars):
            return t.host()
        return t.detach().to("cpu").numpy().astype(np.float64, copy=False)

    def stream(self):
        """Raw handle of torch's CURRENT stream on this device (looked up on every call: a `torch.cuda.stream(...)`
        context around a solver is honoured)."""
        return _raw_stream(self._dev_index)

    def synchronize(self):
        torch.cuda.current_stream(self.device).synchronize()

    # ------------------------------------------------------------------ collectives
    def allreduce(self, scal, i=0, j=None):
        """Sum scalars [i, j) of a DevScalars block (or a float64 tensor view) over all ranks, in place.
        No-op for a single rank."""
        self.reduction_points += 1          # counted on one rank too: what a sharded run of the same solve would exchange
        if self.comm is not None and self.world > 1:
            self.comm.allreduce_sum_(scal.view(i, j) if isinstance(scal, DevScalars) else scal)
        return scal

    def copy_scalars(self, src, i, dst, j, n=1):
        """dst[j .. j+n) = src[i .. i+n) between DevScalars blocks, on the stream (no host visit)."""
        dst.t[j:j + n].copy_(src.t[i:i + n], non_blocking=True)

    # ------------------------------------------------------------------ operators
    def op_apply(self, handle, transpose, x, y, batch=1, ldx=0, ldy=0, sumsq=None):
        rc = self.lib.trk_op_apply(handle, int(bool(transpose)), x.data_ptr(), int(ldx), y.data_ptr(), int(ldy),
                                   int(batch), _ptr(sumsq), self.stream())
        _lib.check(rc, "trk_op_apply")

    def op_apply_axpby(self, handle, transpose, x, a, b, z, out, sumsq=None, hints=0):
        """out = a * Op(x) + b * z (+ ||out||^2); a, b: float or Coef; z may be None."""
        a, b = _as_coef(a), _as_coef(b)
        rc = self.lib.trk_op_apply_axpby(handle, int(bool(transpose)), x.data_ptr(), a.c, a.num, a.den, a.flags, b.c, b.num,
                                         b.den, b.flags, None if z is None else z.data_ptr(), out.data_ptr(), _ptr(sumsq),
                                         int(hints), self.stream())
        _lib.check(rc, "trk_op_apply_axpby")

    # ------------------------------------------------------------------ reductions (local sums)
    def dot(self, x, y, out):
        _lib.check(self.lib.trk_dot(x.data_ptr(), y.data_ptr(), x.numel(), _ptr(out), self.stream()), "trk_dot")

    def nrm2sq(self, x, out):
        _lib.check(self.lib.trk_nrm2sq(x.data_ptr(), x.numel(), _ptr(out), self.stream()), "trk_nrm2sq")

    def diff_nrm2sq(self, x, y, out):
        _lib.check(self.lib.trk_diff_nrm2sq(x.data_ptr(), y.data_ptr(), x.numel(), _ptr(out), self.stream()),
                   "trk_diff_nrm2sq")

    # ------------------------------------------------------------------ axpy family
    def axpby(self, a, x, b, y, out, sumsq=None):
        """out = a*x + b*y  (a, b: float or Coef; y may be None)."""
        a, b = _as_coef(a), _as_coef(b)
        rc = self.lib.trk_axpby(x.numel(), a.c, a.num, a.den, a.flags, x.data_ptr(), b.c, b.num, b.den, b.flags,
                                None if y is None else y.data_ptr(), out.data_ptr(), _ptr(sumsq), self.stream())
        _lib.check(rc, "trk_axpby")

    def scale(self, a, x, out, sumsq=None):
        self.axpby(a, x, 0.0, None, out, sumsq)

    def arnoldi_step(self, op_handle, V, k, w, G, ldg, W, S):
        """One Arnoldi step enqueued by one call (trk_arnoldi_step: apply, the sweep's two passes, the normalisation)."""
        rc = self.lib.trk_arnoldi_step(op_handle, V.data_ptr(), V.stride(0), int(k), w.data_ptr(), _ptr(G), int(ldg), _ptr(W), _ptr(S),
                                       self.stream())
        _lib.check(rc, "trk_arnoldi_step")

    def arnoldi_step_post(self, op_handle, V, k, w, G, ldg, W, S, post):
        """arnoldi_step whose last kernel carries the mailbox post `post` (DevScalars.rider_post()[0]: trk_arnoldi_step_post)."""
        mb, slot, _src, offset, count = post[:5]
        rc = self.lib.trk_arnoldi_step_post(op_handle, V.data_ptr(), V.stride(0), int(k), w.data_ptr(), _ptr(G), int(ldg), _ptr(W), _ptr(S),
                                            mb, int(slot), int(offset), int(count), self.stream())
        _lib.check(rc, "trk_arnoldi_step_post")

    def scale_dot(self, a, x, out, z, dot_out):
        """out = a*x and dot_out = <out, z> in one pass (trk_scale_dot: the new basis vector with its entry of the projected
        right-hand side)."""
        a = _as_coef(a)
        rc = self.lib.trk_scale_dot(x.numel(), a.c, a.num, a.den, a.flags, x.data_ptr(), out.data_ptr(), z.data_ptr(), _ptr(dot_out),
                                    self.stream())
        _lib.check(rc, "trk_scale_dot")

    def copy(self, x, out):
        out.copy_(x)

    def mul(self, x, y, out):
        _lib.check(self.lib.trk_mul(x.numel(), x.data_ptr(), y.data_ptr(), out.data_ptr(), self.stream()), "trk_mul")

    def mul_diff(self, w, x, y, out):
        """out = w * (x - y)."""
        _lib.check(self.lib.trk_mul_diff(x.numel(), w.data_ptr(), x.data_ptr(), y.data_ptr(), out.data_ptr(), self.stream()),
                   "trk_mul_diff")

    def mm_weights(self, x, y, eps, p, out):
        """out = ((x - y)^2 + eps^2)^(p/2 - 1); y may be None."""
        rc = self.lib.trk_mm_weights(x.numel(), x.data_ptr(), None if y is None else y.data_ptr(), float(eps), float(p),
                                     out.data_ptr(), self.stream())
        _lib.check(rc, "trk_mm_weights")

    def group_weights(self, d, groups, group_len, add, expo, copies, out):
        """out[c*groups + i] = (sum_t d[i*group_len + t]^2 + add)^expo, c < copies (MMGKS group-sparsity weights)."""
        rc = self.lib.trk_group_weights(d.data_ptr(), int(groups), int(group_len), float(add), float(expo), int(copies),
                                        out.data_ptr(), self.stream())
        _lib.check(rc, "trk_group_weights")

    def isotv_weights(self, x, N, nt, u_tail, eps, q, out):
        """MMGKS isotropic-TV weights (MMGKS.py:61-77): spatial part from the centered gradient of x viewed as [N][N][nt],
        written twice, then (u_tail^2 + eps^2)^((q-2)/4); u_tail may be None / empty."""
        nt_tail = 0 if u_tail is None else u_tail.numel()
        rc = self.lib.trk_isotv_weights(x.data_ptr(), int(N), int(nt), None if nt_tail == 0 else u_tail.data_ptr(), nt_tail,
                                        float(eps), float(q), out.data_ptr(), self.stream())
        _lib.check(rc, "trk_isotv_weights")

    def sparse_operator(self, M):
        """A scipy.sparse matrix as an operator of this engine (device CSR SpMV)."""
        from .operators import SparseOp
        return SparseOp(M, engine=self)

    def cgls_update(self, gamma, delta, x, p, x_new, r, w, x_true, sums):
        rc = self.lib.trk_cgls_update_xr(x.numel(), r.numel(), _ptr(gamma), _ptr(delta), x.data_ptr(), p.data_ptr(),
                                         x_new.data_ptr(), r.data_ptr(), w.data_ptr(),
                                         None if x_true is None else x_true.data_ptr(), _ptr(sums), self.stream())
        _lib.check(rc, "trk_cgls_update_xr")

    def cgls_update_deferred(self, gamma, delta, x, p, x_new, r, w, x_true, partials, capacity):
        n = ctypes.c_int(0)
        rc = self.lib.trk_cgls_update_xr_deferred(x.numel(), r.numel(), _ptr(gamma), _ptr(delta), x.data_ptr(), p.data_ptr(),
                                                  x_new.data_ptr(), r.data_ptr(), w.data_ptr(),
                                                  None if x_true is None else x_true.data_ptr(), _ptr(partials),
                                                  int(capacity), ctypes.byref(n), self.stream())
        _lib.check(rc, "trk_cgls_update_xr_deferred")
        return n.value

    # ------------------------------------------------------------------ fused CGLS fast path (include/trk.h)
    def op_can_fuse(self, handle):
        can = ctypes.c_int(0)
        _lib.check(self.lib.trk_op_fused_caps(handle, ctypes.byref(can)), "trk_op_fused_caps")
        return can.value                   # 0: no; 1: two-operand form and raw partials; 2: raw partials only

    def op_apply_fused(self, handle, transpose, x1, x2, sign, num, num_n, den, den_n, comb, y, partials, capacity):
        """x2 = None: the plain one-operand apply with ||y||^2 left as raw block partials."""
        n = ctypes.c_int(0)
        rc = self.lib.trk_op_apply_fused(handle, int(bool(transpose)), x1.data_ptr(), None if x2 is None else x2.data_ptr(),
                                         float(sign), _ptr(num), int(num_n), _ptr(den), int(den_n),
                                         None if comb is None else comb.data_ptr(), y.data_ptr(), _ptr(partials),
                                         int(capacity), ctypes.byref(n), self.stream())
        _lib.check(rc, "trk_op_apply_fused")
        return n.value

    def cgls_update_src(self, gamma, gamma_n, delta, delta_n, x, p, x_new, r, w, x_true, pub_delta, partials, capacity):
        n = ctypes.c_int(0)
        rc = self.lib.trk_cgls_update_xr_src(x.numel(), r.numel(), _ptr(gamma), int(gamma_n), _ptr(delta), int(delta_n),
                                             x.data_ptr(), p.data_ptr(), x_new.data_ptr(), r.data_ptr(), w.data_ptr(),
                                             None if x_true is None else x_true.data_ptr(), _ptr(pub_delta), _ptr(partials),
                                             int(capacity), ctypes.byref(n), self.stream())
        _lib.check(rc, "trk_cgls_update_xr_src")
        return n.value

    def cgls_update_grouping(self, n):
        """1: the raw-partials CGLS iteration regroups its updates as [r] / [x, p] (p read once) for vectors of n floats."""
        return int(self.lib.trk_cgls_update_grouping(int(n)))

    def cgls_r_update(self, gamma_old, delta, delta_n, r, w, pub_delta):
        rc = self.lib.trk_cgls_r_update(r.numel(), _ptr(gamma_old), _ptr(delta), int(delta_n), r.data_ptr(), w.data_ptr(),
                                        _ptr(pub_delta), self.stream())
        _lib.check(rc, "trk_cgls_r_update")

    def cgls_xp_update(self, gamma_old, delta, gamma_new, gamma_new_n, x, p, t, x_new, x_true, pub_gamma, partials, capacity):
        n = ctypes.c_int(0)
        rc = self.lib.trk_cgls_xp_update(x.numel(), _ptr(gamma_old), _ptr(delta), _ptr(gamma_new), int(gamma_new_n),
                                         x.data_ptr(), p.data_ptr(), t.data_ptr(), x_new.data_ptr(),
                                         None if x_true is None else x_true.data_ptr(), _ptr(pub_gamma), _ptr(partials),
                                         int(capacity), ctypes.byref(n), self.stream())
        _lib.check(rc, "trk_cgls_xp_update")
        return n.value

    def cgls_p_update(self, t, p, gamma_new, gamma_new_n, gamma_old, pub_gamma):
        rc = self.lib.trk_cgls_p_update(p.numel(), t.data_ptr(), p.data_ptr(), _ptr(gamma_new), int(gamma_new_n),
                                        _ptr(gamma_old), _ptr(pub_gamma), self.stream())
        _lib.check(rc, "trk_cgls_p_update")

    def cgls_x_update(self, gamma, gamma_n, delta, delta_n, x, p, x_new, x_true, pub_delta, pub_gamma, partials, capacity):
        n = ctypes.c_int(0)
        rc = self.lib.trk_cgls_x_update(x.numel(), _ptr(gamma), int(gamma_n), _ptr(delta), int(delta_n), x.data_ptr(),
                                        p.data_ptr(), x_new.data_ptr(), None if x_true is None else x_true.data_ptr(),
                                        _ptr(pub_delta), _ptr(pub_gamma), _ptr(partials), int(capacity), ctypes.byref(n),
                                        self.stream())
        _lib.check(rc, "trk_cgls_x_update")
        return n.value

    def cgls_iterate(self, handle, k_first, n_iters, p, r, t, w, X, keep, x_prev, x_true, S, NP, np_cap, n_np,
                     PG=None, PD=None, pcap=0, grouping=-1):
        """n_iters generic CGLS iterations in one library call; returns the partial-block count.  PG / PD: buffers for the
        operator's raw ||t||^2 / ||w||^2 block partials (four launches per iteration instead of six)."""
        c = ctypes.c_int(int(n_np))
        rc = self.lib.trk_cgls_iterate(handle, int(k_first), int(n_iters), p.data_ptr(), r.data_ptr(), t.data_ptr(),
                                       w.data_ptr(), X.data_ptr(), X.stride(0), int(bool(keep)), x_prev.data_ptr(),
                                       None if x_true is None else x_true.data_ptr(), _ptr(S), _ptr(NP), int(np_cap),
                                       ctypes.byref(c), _ptr(PG), _ptr(PD), int(pcap), int(grouping), self.stream())
        _lib.check(rc, "trk_cgls_iterate")
        return c.value

    def cgls_iterate_fused(self, handle, k_first, n_iters, P, R, t, w, X, keep, x_prev, x_true, S, PG, PD, pcap, NP, np_cap,
                           n_g, n_np):
        """n_iters fused (3-launch) CGLS iterations in one library call; returns (n_g, n_np)."""
        cg, cn = ctypes.c_int(int(n_g)), ctypes.c_int(int(n_np))
        rc = self.lib.trk_cgls_iterate_fused(handle, int(k_first), int(n_iters), P.data_ptr(), P.stride(0), R.data_ptr(),
                                             R.stride(0), t.data_ptr(), w.data_ptr(), X.data_ptr(), X.stride(0),
                                             int(bool(keep)), x_prev.data_ptr(), None if x_true is None else x_true.data_ptr(),
                                             _ptr(S), _ptr(PG), _ptr(PD), int(pcap), _ptr(NP), int(np_cap), ctypes.byref(cg),
                                             ctypes.byref(cn), self.stream())
        _lib.check(rc, "trk_cgls_iterate_fused")
        return cg.value, cn.value

    def cgls_tiled_caps(self, handle, np_cap, pcap):
        can = ctypes.c_int(0)
        _lib.check(self.lib.trk_cgls_tiled_caps(handle, int(np_cap), int(pcap), ctypes.byref(can)), "trk_cgls_tiled_caps")
        return bool(can.value)

    def cgls_iterate_tiled(self, handle, k_first, n_iters, P, R, t, X, keep, x_prev, x_true, S, PG, PD, pcap, NP, np_cap,
                           n_g, n_np):
        """n_iters tiled (2-launch) CGLS iterations on a small blur problem in one library call; returns (n_g, n_np)."""
        cg, cn = ctypes.c_int(int(n_g)), ctypes.c_int(int(n_np))
        rc = self.lib.trk_cgls_iterate_tiled(handle, int(k_first), int(n_iters), P.data_ptr(), P.stride(0), R.data_ptr(),
                                             R.stride(0), t.data_ptr(), X.data_ptr(), X.stride(0), int(bool(keep)),
                                             x_prev.data_ptr(), None if x_true is None else x_true.data_ptr(), _ptr(S),
                                             _ptr(PG), _ptr(PD), int(pcap), _ptr(NP), int(np_cap), ctypes.byref(cg),
                                             ctypes.byref(cn), self.stream())
        _lib.check(rc, "trk_cgls_iterate_tiled")
        return cg.value, cn.value

    def cgls_iterate_tiled2(self, handle, k_first, n_iters, p, w, R, t, X, keep, x_prev, x_true, S, PG, PD, pcap, NP, np_cap, n_g, n_np):
        """The tiled iteration with two blurs instead of four (trk_cgls_iterate_tiled2); returns (n_g, n_np)."""
        cg, cn = ctypes.c_int(int(n_g)), ctypes.c_int(int(n_np))
        rc = self.lib.trk_cgls_iterate_tiled2(handle, int(k_first), int(n_iters), p.data_ptr(), w.data_ptr(), R.data_ptr(), R.stride(0),
                                              t.data_ptr(), X.data_ptr(), X.stride(0), int(bool(keep)), x_prev.data_ptr(),
                                              None if x_true is None else x_true.data_ptr(), _ptr(S), _ptr(PG), _ptr(PD), int(pcap),
                                              _ptr(NP), int(np_cap), ctypes.byref(cg), ctypes.byref(cn), self.stream())
        _lib.check(rc, "trk_cgls_iterate_tiled2")
        return cg.value, cn.value

    # ------------------------------------------------------------------ CGLS with one all-reduce per iteration (cgls_sharded.hip)
    def dot_pair(self, q, w, out3):
        """out3[0] = <q, q>, out3[1] = <q, w>, out3[2] = <w, w> (w None: 0, 0): this rank's sums."""
        _lib.check(self.lib.trk_dot_pair(q.data_ptr(), None if w is None else w.data_ptr(), q.numel(), _ptr(out3), self.stream()),
                   "trk_dot_pair")

    def cgls_sharded_update(self, G4, gamma_prev, first, x, p, t, x_new, r, q, w, x_true, pub_delta, pub_gamma, partials, capacity):
        n = ctypes.c_int(0)
        rc = self.lib.trk_cgls_sharded_update(x.numel(), r.numel(), _ptr(G4), _ptr(gamma_prev), int(bool(first)),
                                              x.data_ptr(), p.data_ptr(), t.data_ptr(), x_new.data_ptr(), r.data_ptr(), q.data_ptr(),
                                              w.data_ptr(), None if x_true is None else x_true.data_ptr(), _ptr(pub_delta),
                                              _ptr(pub_gamma), _ptr(partials), int(capacity), ctypes.byref(n), self.stream())
        _lib.check(rc, "trk_cgls_sharded_update")
        return n.value

    def cgls_sharded_scalars(self, q, w, gamma_partials, n_gamma, G4):
        """G4[1..3] = <q,q>, <q,w>, <w,w>; G4[0] = sum of the n_gamma raw partials (n_gamma = 0: left as it is)."""
        rc = self.lib.trk_cgls_sharded_scalars(q.data_ptr(), None if w is None else w.data_ptr(), q.numel(), _ptr(gamma_partials),
                                               int(n_gamma), _ptr(G4), self.stream())
        _lib.check(rc, "trk_cgls_sharded_scalars")

    def cgls_iterate_sharded(self, handle, comm_handle, k_first, n_iters, p, r, t, q, w, X, keep, x_prev, x_true, S, G4, NP, np_cap,
                             n_np, PG=None, pcap=0, n_g=0):
        """Returns (norm-partial blocks per iteration, gamma partials left behind by the last adjoint apply)."""
        n = ctypes.c_int(int(n_np))
        ng = ctypes.c_int(int(n_g))
        rc = self.lib.trk_cgls_iterate_sharded(handle, comm_handle, int(k_first), int(n_iters), p.data_ptr(), r.data_ptr(),
                                               t.data_ptr(), q.data_ptr(), w.data_ptr(), X.data_ptr(), X.stride(0), int(keep),
                                               x_prev.data_ptr(), None if x_true is None else x_true.data_ptr(), _ptr(S), _ptr(G4),
                                               _ptr(NP), int(np_cap), ctypes.byref(n), _ptr(PG), int(pcap),
                                               ctypes.byref(ng) if PG is not None else None, self.stream())
        _lib.check(rc, "trk_cgls_iterate_sharded")
        return n.value, ng.value

    def finalize_batched(self, partials, nblocks, nvals, batches, out, out_stride):
        rc = self.lib.trk_finalize_batched(_ptr(partials), int(nblocks), int(nvals), int(batches), _ptr(out), int(out_stride),
                                           self.stream())
        _lib.check(rc, "trk_finalize_batched")

    # ------------------------------------------------------------------ tall-skinny basis ops (row-per-vector V[k_max, n])
    def gemv_t_x(self, V, k, r, xrow, out_h, out_x):
        """out_h[j] = V[j] . r for j < k and out_x = xrow . r from the same pass over r (trk_gemv_t_x; local sums)."""
        rc = self.lib.trk_gemv_t_x(V.data_ptr(), V.stride(0), int(k), r.numel(), r.data_ptr(), xrow.data_ptr(), _ptr(out_h),
                                   _ptr(out_x), self.stream())
        _lib.check(rc, "trk_gemv_t_x")

    def gemv_t(self, V, k, r, out_h, w2=None):
        """out_h[j] = sum_i w2[i] V[j,i] r[i]  for j < k (local sums)."""
        rc = self.lib.trk_gemv_t(V.data_ptr(), V.stride(0), int(k), r.numel(), r.data_ptr(),
                                 None if w2 is None else w2.data_ptr(), _ptr(out_h), self.stream())
        _lib.check(rc, "trk_gemv_t")

    GEMV_NT_MAX_K = 16

    def gemv_t2(self, V, k, r, r2, out_h2k):
        """out[j] = V[j] . r, out[k + j] = V[j] . r2 for j < k, one pass over V (local sums)."""
        rc = self.lib.trk_gemv_t2(V.data_ptr(), V.stride(0), int(k), r.numel(), r.data_ptr(), r2.data_ptr(), _ptr(out_h2k),
                                  self.stream())
        _lib.check(rc, "trk_gemv_t2")

    def gemv_tn(self, V, k, rhs, out):
        """out[q*k + j] = V[j] . rhs[q] for 3 or 4 right-hand sides, one pass over V (trk_gemv_tn; local sums)."""
        arr = (ctypes.c_void_p * len(rhs))(*[r.data_ptr() for r in rhs])
        rc = self.lib.trk_gemv_tn(V.data_ptr(), V.stride(0), int(k), rhs[0].numel(), arr, len(rhs), _ptr(out), self.stream())
        _lib.check(rc, "trk_gemv_tn")

    def gram_row_from_sweep(self, G, ldg, k, a, c, s_rr, rho2, rhs=None, tb=None):
        """Row / column k of G = V^T M V for v_k = (r - V c)/rho from a = V^T (M r), c, s = r.M r, rho^2 (trk_gram_row_from_sweep)."""
        rc = self.lib.trk_gram_row_from_sweep(_ptr(G), int(ldg), int(k), _ptr(a), _ptr(c), _ptr(s_rr), _ptr(rho2), _ptr(rhs),
                                              _ptr(tb), self.stream())
        _lib.check(rc, "trk_gram_row_from_sweep")

    GRAM_TIKHONOV_MAX_K = 139          # the k x (k+1) factor lives in LDS (160 KB per workgroup on gfx950)

    def gram_tikhonov(self, GA, lda, GL, ldl, c, k, lam, y, Minv=None, ldm=0, k_from=0):
        """y = (G_A + lam G_L)^-1 c on the device.  Minv = None: Cholesky from scratch (k <= GRAM_TIKHONOV_MAX_K); else the
        inverse kept in Minv (valid for the leading k_from rows, same lam) is bordered by rows k_from .. k-1."""
        rc = self.lib.trk_gram_tikhonov(_ptr(GA), int(lda), _ptr(GL), int(ldl), _ptr(c), int(k), float(lam), _ptr(Minv), int(ldm),
                                        int(k_from), _ptr(y), self.stream())
        _lib.check(rc, "trk_gram_tikhonov")

    def hess_tikhonov(self, H, ldh, G, Minv, ldg, coef, coef2, nrm2_sq, beta0, k, lam, mode, y):
        """Arnoldi step k: append column k-1 of H (coef (+ coef2), sqrt(*nrm2_sq)), extend G = H^T H, solve
        (G + lam I) y = beta0 H[0,:]^T on the device (trk_hess_tikhonov; mode 0 Cholesky, 1 bordering update of Minv, 2 start)."""
        rc = self.lib.trk_hess_tikhonov(_ptr(H), int(ldh), _ptr(G), _ptr(Minv), int(ldg), _ptr(coef), _ptr(coef2),
                                        _ptr(nrm2_sq), float(beta0), int(k), float(lam), int(mode), _ptr(y), self.stream())
        _lib.check(rc, "trk_hess_tikhonov")

    def cgs_coeffs(self, G, ldg, h, g_new, k, passes, c):
        """c = coefficients of `passes` Gram-Schmidt sweeps from h = V^T r and the Gram matrix G (device doubles); g_new: Gram
        row of the newest vector, installed into G first (None: G is complete)."""
        rc = self.lib.trk_cgs_coeffs(_ptr(G), int(ldg), _ptr(h), _ptr(g_new), int(k), int(passes), _ptr(c), self.stream())
        _lib.check(rc, "trk_cgs_coeffs")

    def gks_rows_solve(self, GA, GL, ldg, k, c_sweep, rho2, c_rhs, lam, Minv, ldm, k_from, y, a_L, s_L, ga_new=None, a_A=None, s_A=None,
                       tb=None):
        """Rows k of G_A (installed from ga_new, or from the sweep's products a_A, s_A, tb) and of G_L (a_L, s_L), then the bordered
        solve over k + 1 vectors, in one launch (trk_gks_rows_solve)."""
        rc = self.lib.trk_gks_rows_solve(_ptr(GA), _ptr(GL), int(ldg), int(k), _ptr(ga_new), _ptr(a_A), _ptr(s_A), _ptr(tb), _ptr(a_L),
                                         _ptr(s_L), _ptr(c_sweep), _ptr(rho2), _ptr(c_rhs), float(lam), _ptr(Minv), int(ldm), int(k_from),
                                         _ptr(y), self.stream())
        _lib.check(rc, "trk_gks_rows_solve")

    def cgs_coeffs_rho(self, G, ldg, h, g_new, k, passes, c, rr, rho2):
        """cgs_coeffs and, from rr = r . r, rho2 = ||r - V c||^2 by algebra (trk_cgs_coeffs_rho): the norm of the vector the next pass forms."""
        rc = self.lib.trk_cgs_coeffs_rho(_ptr(G), int(ldg), _ptr(h), _ptr(g_new), int(k), int(passes), _ptr(c), _ptr(rr), _ptr(rho2),
                                         self.stream())
        _lib.check(rc, "trk_cgs_coeffs_rho")

    def gemv_orth_iterate(self, V, k, w, c, rho2, vn, y_next=None, x_next=None, ref=None, partials=None, capacity=0, chk=None):
        """vn = (w - V[0..k) c) / sqrt(rho2) and (with y_next, k + 1 device doubles) x_next = V[0..k) y_next[:k] + y_next[k] vn in ONE pass
        over the basis (trk_gemv_orth_iterate); with `ref`, raw block partials of ||x_next - ref||^2 — returns their count (0 without)."""
        n = ctypes.c_int(0)
        rc = self.lib.trk_gemv_orth_iterate(V.data_ptr(), V.stride(0), int(k), vn.numel(), w.data_ptr(), _ptr(c), _ptr(rho2), _ptr(y_next),
                                            vn.data_ptr(), None if x_next is None else x_next.data_ptr(),
                                            None if ref is None else ref.data_ptr(), _ptr(partials), int(capacity), ctypes.byref(n),
                                            _ptr(chk), self.stream())
        _lib.check(rc, "trk_gemv_orth_iterate")
        return n.value

    def gemv_n_err(self, V, k, y, out, ref, partials, capacity):
        """out = sum_j y[j] V[j]; raw block partials of ||out - ref||^2 into `partials`; returns their count."""
        n = ctypes.c_int(0)
        rc = self.lib.trk_gemv_n_err(V.data_ptr(), V.stride(0), int(k), out.numel(), _ptr(y), out.data_ptr(), ref.data_ptr(),
                                     _ptr(partials), int(capacity), ctypes.byref(n), self.stream())
        _lib.check(rc, "trk_gemv_n_err")
        return n.value

    def host_bidiag_tikhonov(self, alphas, betas, beta0, mu, y_over_alpha=False):
        """argmin || [B_k; mu I] y - beta0 e_1 || on the host (trk_host_bidiag_tikhonov) from B_k's entries; a float64 array."""
        al = np.ascontiguousarray(alphas, dtype=np.float64)
        be = np.ascontiguousarray(betas, dtype=np.float64)
        y = np.empty(al.size, dtype=np.float64)
        rc = self.lib.trk_host_bidiag_tikhonov(al.ctypes.data, be.ctypes.data, int(al.size), float(beta0), float(mu),
                                               int(bool(y_over_alpha)), y.ctypes.data)
        _lib.check(rc, "trk_host_bidiag_tikhonov")
        return y

    def gemv_n_hosty(self, V, k, y_host, out, ref=None, partials=None, capacity=0):
        """out = sum_j y_host[j] V[j], the coefficients a float64 HOST array (they ride in the launch's arguments: trk_gemv_n_hosty);
        with `ref`, raw block partials of ||out - ref||^2 into `partials` — returns their count (0 without)."""
        n = ctypes.c_int(0)
        rc = self.lib.trk_gemv_n_hosty(V.data_ptr(), V.stride(0), int(k), out.numel(), y_host.ctypes.data, out.data_ptr(),
                                       None if ref is None else ref.data_ptr(), _ptr(partials), int(capacity), ctypes.byref(n),
                                       self.stream())
        _lib.check(rc, "trk_gemv_n_hosty")
        return n.value

    def gk_step(self, handle, k, u_k, v_prev, v_k, u_next, AB, chained, defer_alpha, defer_beta):
        """One Golub-Kahan step on unnormalised vectors in one call (trk_gk_step): both half steps with their norms."""
        rc = self.lib.trk_gk_step(handle, int(k), u_k.data_ptr(), None if v_prev is None else v_prev.data_ptr(), v_k.data_ptr(),
                                  u_next.data_ptr(), AB.base, int(bool(chained)), int(bool(defer_alpha)), int(bool(defer_beta)),
                                  self.stream())
        _lib.check(rc, "trk_gk_step")

    def gk_step_lsqr(self, handle, k, u_k, v_prev, v_k, u_next, AB, chained, defer_alpha, defer_beta, w, x_in, x_out, ref, partials,
                     capacity, damp, state_in, state_out):
        """gk_step (k >= 1) that also advances damped LSQR's iterate by the step of v_prev (trk_gk_step_lsqr): on the projector the
        update rides the adjoint half step's pixel pass.  Returns the number of error partials written (0 without `ref`)."""
        n = ctypes.c_int(0)
        rc = self.lib.trk_gk_step_lsqr(handle, int(k), u_k.data_ptr(), v_prev.data_ptr(), v_k.data_ptr(), u_next.data_ptr(), AB.base,
                                       int(bool(chained)), int(bool(defer_alpha)), int(bool(defer_beta)), w.data_ptr(),
                                       None if x_in is None else x_in.data_ptr(), x_out.data_ptr(),
                                       None if ref is None else ref.data_ptr(), _ptr(partials), int(capacity), ctypes.byref(n),
                                       float(damp), _ptr(state_in), _ptr(state_out), self.stream())
        _lib.check(rc, "trk_gk_step_lsqr")
        return n.value

    def gk_step_post(self, handle, k, u_k, v_prev, v_k, u_next, AB, chained, defer_alpha, defer_beta, post, proj=None):
        """gk_step (k >= 1) whose adjoint half carries the mailbox post `post` (DevScalars.rider_post()[0]) — trk_gk_step_post; with
        proj = (vector, partials, capacity) also the projection of gk_step_proj, whose partial count is returned (else 0)."""
        n = ctypes.c_int(0)
        pv, pp, pc = (None, None, 0) if proj is None else (proj[0].data_ptr(), _ptr(proj[1]), int(proj[2]))
        mb, slot, src, off, cnt, sp, sn, sd, so = post
        rc = self.lib.trk_gk_step_post(handle, int(k), u_k.data_ptr(), v_prev.data_ptr(), v_k.data_ptr(), u_next.data_ptr(), AB.base,
                                       int(bool(chained)), int(bool(defer_alpha)), int(bool(defer_beta)), pv, pp, pc, ctypes.byref(n),
                                       mb, int(slot), src, int(off), int(cnt), sp, int(sn), sd, int(so), self.stream())
        _lib.check(rc, "trk_gk_step_post")
        return n.value

    def gk_step_proj(self, handle, k, u_k, v_prev, v_k, u_next, AB, chained, defer_alpha, defer_beta, proj, partials, cap):
        """gk_step that also leaves <u_next, proj> as block partials at `partials` (trk_gk_step_proj); returns their count."""
        n = ctypes.c_int(0)
        rc = self.lib.trk_gk_step_proj(handle, int(k), u_k.data_ptr(), None if v_prev is None else v_prev.data_ptr(), v_k.data_ptr(),
                                       u_next.data_ptr(), AB.base, int(bool(chained)), int(bool(defer_alpha)), int(bool(defer_beta)),
                                       proj.data_ptr(), _ptr(partials), int(cap), ctypes.byref(n), self.stream())
        _lib.check(rc, "trk_gk_step_proj")
        return n.value

    def lsqr_damped_update(self, vk, w, x_in, x_out, alpha_sq, beta_next_sq, beta0_sq, damp, state_in, state_out, first,
                           ref=None, partials=None, capacity=0):
        """One step of damped LSQR's short recurrence (trk_lsqr_damped_update): w and the iterate from the previous ones and
        vk = alpha_k v_k; with `ref`, raw block partials of ||x_out - ref||^2 go to `partials` and their count is returned."""
        nb = ctypes.c_int(0)
        rc = self.lib.trk_lsqr_damped_update(vk.data_ptr(), w.data_ptr(), None if x_in is None else x_in.data_ptr(), x_out.data_ptr(),
                                             x_out.numel(), None if ref is None else ref.data_ptr(), _ptr(partials), int(capacity),
                                             ctypes.byref(nb), _ptr(alpha_sq), _ptr(beta_next_sq), _ptr(beta0_sq), float(damp),
                                             _ptr(state_in), _ptr(state_out), int(bool(first)), self.stream())
        _lib.check(rc, "trk_lsqr_damped_update")
        return nb.value

    def gemv_nt(self, V, k, h, w_in, w_out, g):
        """w_out = w_in - sum_j h[j] V[j] and g[j] = V[j] . w_out (local sums), one pass over V; k <= GEMV_NT_MAX_K."""
        rc = self.lib.trk_gemv_nt(V.data_ptr(), V.stride(0), int(k), w_in.numel(), _ptr(h), w_in.data_ptr(), w_out.data_ptr(),
                                  _ptr(g), self.stream())
        _lib.check(rc, "trk_gemv_nt")

    def gemv_n(self, V, k, y, out, a=0.0, base=None, s=1.0, sumsq=None):
        """out = a*base + s * sum_j y[j] V[j]   (y: k device doubles)."""
        rc = self.lib.trk_gemv_n(V.data_ptr(), V.stride(0), int(k), out.numel(), _ptr(y), float(a),
                                 None if base is None else base.data_ptr(), float(s), out.data_ptr(), _ptr(sumsq),
                                 self.stream())
        _lib.check(rc, "trk_gemv_n")

    def bidiag_tikhonov(self, alpha_sq, alpha_stride, beta_sq, beta_stride, k, mu, beta0_sq, y, work=None,
                        y_over_alpha=False):
        """y = argmin ||B_k y - beta0 e1||^2 + mu^2 ||y||^2 for the Golub-Kahan bidiagonal given as squared device norms.
        work: a zero-initialised DevScalars block kept between calls (resumes the rotation recurrence when only columns
        were appended and mu is unchanged).  y_over_alpha: write y_j / alpha_j (coefficients of un-normalised vectors)."""
        rc = self.lib.trk_bidiag_tikhonov(_ptr(alpha_sq), int(alpha_stride), _ptr(beta_sq), int(beta_stride), int(k),
                                          float(mu), _ptr(beta0_sq), _ptr(y), int(bool(y_over_alpha)),
                                          None if work is None else work.ref(0),
                                          0 if work is None else len(work), self.stream())
        _lib.check(rc, "trk_bidiag_tikhonov")

    WGRAM_TV_MAX_K = 48

    WGRAM_TV_MODES = {"auto": 1, "bf16x2": 2, "bf16x3": 3, "fp32": 0}

    def wgram_tv_precision(self, mode=None):
        """The arithmetic of trk_wgram_tv's tile products (accuracy contract: include/trk.h): 'auto' (default: two bf16 pieces unless a
        per-call probe finds the data's roundings correlated, then the fp32 pipe — decided on the device), 'bf16x2' (fastest; up to
        1.2e-5 per entry on images that repeat a few values), 'bf16x3' or 'fp32' (<= 1e-6 on any image).  Process-wide.  Returns the name
        in force before the call; None only queries."""
        code = -1 if mode is None else self.WGRAM_TV_MODES[mode]
        was = self.lib.trk_wgram_tv_precision(code)
        return {v: n for n, v in self.WGRAM_TV_MODES.items()}[was]

    def wgram_tv_last_probe(self):
        """(verdict, sampled deviation) of the last 'auto' call of wgram_tv (synchronises; diagnostics)."""
        out = (ctypes.c_double * 2)()
        _lib.check(self.lib.trk_wgram_tv_last_probe(out), "trk_wgram_tv_last_probe")
        return int(out[0]), float(out[1])

    def wgram_tv(self, V, k, N, w, G, z=None, h=None):
        """G = (L V) diag(w^2) (L V)^T for the 2-D first-difference L of an N x N image, from V itself (trk_wgram_tv: N % 32 == 0,
        k <= WGRAM_TV_MAX_K; local sums).  With z (an image) the same pass also leaves h[j] = V[j] . z (trk_wgram_tv_z)."""
        if z is not None:
            rc = self.lib.trk_wgram_tv_z(V.data_ptr(), V.stride(0), int(k), int(N), w.data_ptr(), _ptr(G), z.data_ptr(), _ptr(h),
                                         self.stream())
            _lib.check(rc, "trk_wgram_tv_z")
            return
        rc = self.lib.trk_wgram_tv(V.data_ptr(), V.stride(0), int(k), int(N), w.data_ptr(), _ptr(G), self.stream())
        _lib.check(rc, "trk_wgram_tv")

    def wgram(self, W, k, w, b1, G, c1=None, c2=None):
        """G = W diag(w^2) W^T (k x k), c1 = W (w*b1), c2 = W (w^2*b1)  (local sums; w may be None)."""
        rc = self.lib.trk_wgram(W.data_ptr(), W.stride(0), int(k), W.shape[1], None if w is None else w.data_ptr(),
                                None if b1 is None else b1.data_ptr(), _ptr(G), _ptr(c1), _ptr(c2), self.stream())
        _lib.check(rc, "trk_wgram")


_default_engines = {}


def default_engine(comm=None):
    """The process-wide engine of the current device (created on first use; raises without GPU / library)."""
    if not torch.cuda.is_available():
        raise _lib.TrkError("no GPU visible: trips_py_amd runs only on the HIP engine (no CPU fallback)")
    key = (torch.cuda.current_device(), id(comm))
    if key not in _default_engines:
        _default_engines[key] = HipEngine(comm=comm)
    return _default_engines[key]
