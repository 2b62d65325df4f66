"""Small host-side helpers shared by the solvers: operand conversion and result formatting.

Reference solvers take and return NumPy column vectors (n,1); the engine keeps fp32 device vectors.  `numpy in ->
numpy (float64) out`, `torch in -> torch (device, fp32) out` — decided by the type of `b`."""
import numpy as np
import torch

from .operators import LinearOperator


def as_operator(A, role="A"):
    if isinstance(A, LinearOperator):
        return A
    try:
        import scipy.sparse as sp
        if sp.issparse(A):
            # a matrix the reference built with scipy.sparse (e.g. gen_first_derivative_operator_2D): device CSR SpMV
            from .operators import SparseOp
            return SparseOp(A)
    except ImportError:        # pragma: no cover
        pass
    raise TypeError(f"{role} must be a trips_py_amd LinearOperator (Blur2D, Radon2DParallel, BlockDiagOp, "
                    f"FirstDerivative2D, ...); got {type(A).__name__}.  The engine has no host/NumPy operator path.")


class Formatter:
    """Remembers whether the caller speaks NumPy or torch and formats vectors accordingly."""

    def __init__(self, like):
        self.numpy = not isinstance(like, torch.Tensor)

    def vec(self, t):
        """device fp32 vector -> (n,1) column in the caller's flavour (a copy)."""
        if self.numpy:
            return t.detach().to("cpu").numpy().astype(np.float64).reshape(-1, 1)
        return t.detach().clone().reshape(-1, 1)

    def hist(self, H, count):
        """[count, n] device history -> list of (n,1) columns."""
        if self.numpy:
            Hh = H[:count].detach().to("cpu").numpy().astype(np.float64)
            return [Hh[i].reshape(-1, 1) for i in range(count)]
        # (one call that makes the `count` views: a Python loop of slice + reshape cost 2 us per iterate — 0.2 ms of a 1.5 ms CGLS
        #  solve at 512^2)
        return list(H[:count].unsqueeze(-1).unbind(0))


def history_fits(engine, count, n, what):
    """Refuse (loudly) an on-device history that cannot fit; the reference keeps every iterate (CGLS.py:66)."""
    if engine.device.type != "cuda":
        return
    need = int(count) * int(n) * 4
    if need < (256 << 20):
        return                       # (the query below is a driver call of tens of microseconds: not per small solve)
    free, _total = torch.cuda.mem_get_info(engine.device)
    if need > 0.8 * free:
        raise MemoryError(f"{what}: keeping {count} iterates of {n} floats needs {need / 2**30:.1f} GiB of HBM "
                          f"({free / 2**30:.1f} GiB free); pass history=False")


class HistoryView:
    """`info['xHistory']` of a streamed history: a read-only sequence of (n,1) float64 columns (NumPy callers) over the fp32
    rows of a host array or a memory-mapped .npy file, converted on access — the reference returns a list of such columns
    (CGLS.py:66, GKS.py:77) and the demos index it."""

    def __init__(self, rows, iters=None, torch_out=False):
        self.rows = rows                     # [count, n] float32 (numpy array, np.memmap or torch tensor)
        self.iterations = list(range(len(rows))) if iters is None else list(iters)   # which iterates the rows are
        self.torch_out = torch_out           # torch callers get (n,1) float32 host TENSORS from host / file rows, not NumPy

    def __len__(self):
        return len(self.rows)

    def __getitem__(self, k):
        if isinstance(k, slice):
            return [self[i] for i in range(*k.indices(len(self)))]
        r = self.rows[k]
        if isinstance(r, torch.Tensor):
            return r.reshape(-1, 1)
        if self.torch_out:
            return torch.from_numpy(np.array(r, dtype=np.float32)).reshape(-1, 1)
        return np.asarray(r, dtype=np.float64).reshape(-1, 1)

    def __iter__(self):
        return (self[i] for i in range(len(self)))


class History:
    """Where a solver's iterates go (the reference keeps every one of them: CGLS.py:66, GKS.py:77, MMGKS.py:108 — 67 MB each at
    4096^2).  `spec` is the solvers' engine-only `history=` kwarg:

        True / 1        every iterate in an on-device [count, n] block, each written in place into its slot (default)
        False / 0 / None  none (two scratch slots)
        int s >= 2      every s-th iterate (s-1, 2s-1, ...) and the last one the solver formed (also when it stopped early), on
                        the device
        "host"          every iterate, streamed to host memory while the solver runs ahead: the device holds a ring of
                        `ring` slots only
        "<path>.npy"    the same, into a memory-mapped .npy file of shape [count, n] float32 (np.load(path, mmap_mode='r')); a
                        solver that stops early leaves a file of exactly the iterates it formed (header and length rewritten)

    NumPy callers read (n,1) float64 columns from a streamed history, torch callers (n,1) float32 tensors (host memory for
    "host" / file, the device for a stride).

    Protocol: `row(k)` = the device vector iterate k is to be written into (call it right before enqueuing the kernels that
    write it), `pushed(k)` once they are enqueued, `collect(fmt, count)` at the end."""

    def __init__(self, eng, spec, count, n, what, ring=None):
        self.eng, self.count, self.n = eng, int(count), int(n)
        if isinstance(spec, (np.bool_, np.integer)):
            spec = spec.item()                    # NumPy scalars as their Python counterparts
        self.mode = "device" if spec is True else "none" if (spec is False or spec is None) else None
        self.stride = 0
        self.dest = None
        self._events = None
        self._path = None
        if isinstance(spec, bool) or spec is None:
            pass
        elif isinstance(spec, int):
            if spec < 0:
                raise ValueError("history=<int>: the stride must be >= 0 (0 = keep nothing, 1 = keep everything)")
            self.mode, self.stride = ("none", 0) if spec == 0 else ("device", 0) if spec == 1 else ("stream", int(spec))
        elif isinstance(spec, str):
            self.mode = "stream"
        else:
            raise TypeError(f"history={spec!r}: expected True, False, a stride, 'host' or a path ending in .npy")
        self.keeps_any = self.mode != "none"
        cuda = eng.device.type == "cuda"
        if self.mode == "device":
            history_fits(eng, self.count, n, what)
            self.X = eng.empty_basis(max(1, self.count), n)
        elif self.mode == "none":
            self.X = eng.empty_basis(2, n)
        else:
            if ring is None:                      # about 256 MB of device slots, at least 4, an even number
                ring = max(4, min(64, (256 << 20) // (4 * self.n)))
            self.R = max(2, min(int(ring) & ~1, 2 * ((self.count + 1) // 2)))
            self.X = eng.empty_basis(self.R, n)
            if self.stride:                       # kept iterates: s-1, 2s-1, ... and the last one
                self.kept = sorted(set(list(range(self.stride - 1, self.count, self.stride)) + [self.count - 1]))
                history_fits(eng, len(self.kept) + 1, n, what)
                self.dest = eng.empty_basis(len(self.kept) + 1, n)      # + a spare row: the last iterate of an early stop
                self._index = {k: i for i, k in enumerate(self.kept)}
            else:
                self.kept = list(range(self.count))
                self._index = None
                if spec == "host":
                    self.dest = np.empty((self.count, self.n), dtype=np.float32)
                elif spec.endswith(".npy"):
                    self.dest = np.lib.format.open_memmap(spec, mode="w+", dtype=np.float32, shape=(self.count, self.n))
                    self._path = spec
                else:
                    raise ValueError(f"history={spec!r}: expected 'host' or a path ending in .npy")
                # pinned staging rows, one per ring slot; the copy engine fills them, the host drains them into `dest`
                self._pin = torch.empty((self.R, self.n), dtype=torch.float32, pin_memory=cuda)
            if cuda:
                self._copy = torch.cuda.Stream(device=eng.device)
                self._events = [None] * self.R    # copy-done event of the iterate last sent from each ring slot
            self._pending = []                    # (iterate, ring slot) staged, not yet drained into dest

    # ---- solver side
    def slot(self, k):
        """Index of iterate k's row in `self.X`."""
        if self.mode == "device":
            return k
        if self.mode == "none":
            return k & 1
        return k % self.R

    def row(self, k):
        s = self.slot(k)
        if self.mode == "stream" and self._events is not None and self._events[s] is not None:
            torch.cuda.current_stream(self.eng.device).wait_event(self._events[s])     # the slot's last copy must be out
        return self.X[s]

    def pushed(self, k):
        if self.mode != "stream":
            return
        s = k % self.R
        if self._index is not None:               # strided: device-to-device copy of the kept iterates only
            if k in self._index:
                self.dest[self._index[k]].copy_(self.X[s])
            return
        if self._events is None:                  # CPU test engine: copy at once
            self.dest[k] = self.X[s].numpy()
            return
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.eng.device))
        with torch.cuda.stream(self._copy):
            self._copy.wait_event(ev)
            self._pin[s].copy_(self.X[s], non_blocking=True)
            done = torch.cuda.Event()
            done.record(self._copy)
        self._events[s] = done
        self._pending.append((k, s, done))
        # drain what must be out of the pinned ring before its slot is staged again (the solver stays R/2 iterates ahead)
        while len(self._pending) > self.R // 2:
            self._drain_one()

    def _drain_one(self):
        k, s, done = self._pending.pop(0)
        done.synchronize()
        self.dest[k] = self._pin[s].numpy()

    def collect(self, fmt, count):
        """`info['xHistory']` for the first `count` iterates."""
        if self.mode == "none":
            return []
        if self.mode == "device":
            return fmt.hist(self.X, count)
        if self._index is not None:
            idx = [i for i, k in enumerate(self.kept) if k < count]
            iters = [self.kept[i] for i in idx]
            if count >= 1 and count - 1 not in self._index:
                # a solver that stopped early (CGLS tol > 0, MMGKS break): its last iterate is not on the stride, but it is still
                # in its ring slot — nothing was written after it — and goes into the row behind the kept ones
                self.dest[len(idx)].copy_(self.X[(count - 1) % self.R])
                iters.append(count - 1)
            rows = self.dest[:len(iters)]
            return HistoryView(rows.detach().to("cpu").numpy() if fmt.numpy else rows, iters)
        while self._pending:
            self._drain_one()
        if isinstance(self.dest, np.memmap):
            self.dest.flush()
            if count < self.count:                # early stop: the file holds exactly the iterates that exist
                self.dest = None
                _truncate_npy(self._path, count, self.n)
                self.dest = np.load(self._path, mmap_mode="r+") if count else np.empty((0, self.n), dtype=np.float32)
        return HistoryView(self.dest[:count], torch_out=not fmt.numpy)


def _truncate_npy(path, rows, n):
    """Shrink a [count, n] float32 .npy file to its first `rows` rows in place: same header length (the shape's text only gets
    shorter; the padding absorbs it), the data cut behind the last kept row."""
    with open(path, "r+b") as f:
        version = np.lib.format.read_magic(f)
        size = 2 if version == (1, 0) else 4
        hlen = int.from_bytes(f.read(size), "little")
        start = f.tell()
        txt = "{'descr': '<f4', 'fortran_order': False, 'shape': (%d, %d), }" % (rows, n)
        if len(txt) + 1 > hlen:
            raise ValueError(f"{path}: header too short to rewrite")
        f.write((txt + " " * (hlen - len(txt) - 1) + "\n").encode("latin1"))
        f.truncate(start + hlen + rows * n * 4)
