"""Linear operators with the PyLops `LinearOperator` surface the reference's solvers rely on
(`A.shape`, `A @ x`, `A * x`, `A.T @ y`, operands (n,), (n,1), (n,k)), backed by libtrk.so.

  Blur2D(psf, nx, ny)                <- Deblurring2D.forward_Op            trips/test_problems/Deblurring2D.py:66-73
  Blur1D(psf)                        <- Deblurring1D.forward_Op_1D         trips/test_problems/Deblurring1D.py:93-102
  Radon2DParallel(N, angles, ...)    <- parallel-beam OpTomo wrapper       trips/utilities/io.py:392-400
  BlockDiagOp([A_0, ..])             <- pylops.BlockDiag / frame slicing    trips/utilities/io.py:420, :223-225
  FirstDerivative2D(N)               <- gen_first_derivative_operator_2D   trips/utilities/operators.py:30-36
  SpaceTimeDerivative(N, nt)         <- gen_spacetime_derivative_operator  trips/utilities/operators.py:39-45
  Identity(n)                        <- pylops.Identity                    trips/solvers/Hybrid_LSQR.py:76
  first_derivative_operator(_2d), spatial_/time_derivative_operator, VStack
                                     <- the PyLops-built regularisers      trips/utilities/operators_old.py:22-61

NumPy in -> NumPy (float64) out, so the unmodified reference solvers can be handed one of these; torch device
tensors in -> torch device tensors out.  The engine's own solvers use `apply()` on fp32 device vectors and
row-per-vector bases and never leave the GPU.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from .engine import default_engine


class LinearOperator:
    """Base class: subclasses provide `_apply(x2d, y2d, transpose, sumsq)` on [batch, n] fp32 device tensors."""

    dtype = np.dtype("float32")
    # True when one apply moves about as many bytes as reading two or three vectors (stencils): the projection solvers
    # then form A x / L x directly instead of (AV) y / (LV) y, which reads k basis vectors
    streaming = False

    def __init__(self, shape, engine=None):
        self.shape = (int(shape[0]), int(shape[1]))
        self.engine = engine if engine is not None else default_engine()

    # ---- engine-native entry point -----------------------------------------------------------
    def apply(self, x, out=None, transpose=False, sumsq=None):
        """y = A x (or A^T x) for a 1-D fp32 device vector, or a [batch, n] row-per-vector block.
        `sumsq` (1-element float64 device view) receives the LOCAL sum(y*y) fused into the kernel."""
        nin = self.shape[0] if transpose else self.shape[1]
        nout = self.shape[1] if transpose else self.shape[0]
        one = x.dim() == 1
        x2 = x.reshape(1, -1) if one else x
        if x2.shape[1] != nin:
            raise ValueError(f"dimension mismatch: operator expects {nin}, got {x2.shape[1]}")
        if out is None:
            out = torch.empty((x2.shape[0], nout), dtype=torch.float32, device=x.device)
        o2 = out.reshape(1, -1) if out.dim() == 1 else out
        if x2.stride(1) != 1 or o2.stride(1) != 1:
            raise ValueError("apply() needs unit-stride vectors")
        self._apply(x2, o2, bool(transpose), sumsq)
        return o2[0] if one else o2

    def _apply(self, x2, y2, transpose, sumsq):
        raise NotImplementedError

    # ---- PyLops-style surface ----------------------------------------------------------------
    def _dot(self, x, transpose):
        nin = self.shape[0] if transpose else self.shape[1]
        is_np = not isinstance(x, torch.Tensor)
        if is_np:
            xa = np.asarray(x)
            if xa.ndim not in (1, 2) or xa.shape[0] != nin:
                raise ValueError(f"dimension mismatch: operator is {self.shape}, operand is {xa.shape}")
            t = torch.from_numpy(np.ascontiguousarray(xa.reshape(nin, -1).T, dtype=np.float32)).to(self.engine.device)
            y = self.apply(t, transpose=transpose)                       # [k, nout]
            out = y.T.to("cpu").numpy().astype(np.float64)
            return out.reshape(-1) if xa.ndim == 1 else out
        if x.dim() not in (1, 2) or x.shape[0] != nin:
            raise ValueError(f"dimension mismatch: operator is {self.shape}, operand is {tuple(x.shape)}")
        t = x.to(device=self.engine.device, dtype=torch.float32)
        rows = t.reshape(nin, -1).T.contiguous()                         # (n,k) column view -> row-per-vector
        y = self.apply(rows, transpose=transpose).T
        return y.reshape(-1) if x.dim() == 1 else y

    def matvec(self, x):
        return self._dot(x, False)

    def rmatvec(self, y):
        return self._dot(y, True)

    matmat = matvec
    rmatmat = rmatvec

    def dot(self, x):
        return self._dot(x, False)

    def __matmul__(self, x):
        return self._dot(x, False)

    def __mul__(self, x):          # `L * v` means matvec in the reference (GKS.py:95, MMGKS.py:127)
        return self._dot(x, False)

    def __call__(self, x):
        return self._dot(x, False)

    @property
    def T(self):
        return _Transposed(self)

    H = T

    def adjoint(self):
        return _Transposed(self)

    transpose = adjoint

    def __array__(self, *args, **kwargs):
        # NumPy asking for an array means a caller took this operator for a matrix — in the reference that is
        # utils.is_identity's `np.allclose(A, np.eye(n))` on any SQUARE operand that is not a pylops LinearOperator
        # (utils.py:55; an n x n eye).  Fail with the remedy instead of an obscure ufunc error.
        raise TypeError(f"{type(self).__name__} is a matrix-free operator, not an array; to hand it to code that tests "
                        "`isinstance(op, pylops.LinearOperator)` (trips/utilities/utils.py:55) pass op.to_pylops()")

    def to_pylops(self):
        """This operator as a `pylops.FunctionOperator(matvec, rmatvec, nr, nc)` — the wrapper the reference's own test
        problems build around their callables (Deblurring2D.py:72, Tomography.py:83, io.py:400).  Use it to hand an engine
        operator to the UNMODIFIED reference solvers: it then passes their `isinstance(.., pylops.LinearOperator)` tests
        (utils.py:55, gcv.py:33) like the operators it replaces.  Needs PyLops (a dependency of the reference)."""
        import pylops
        return pylops.FunctionOperator(self.matvec, self.rmatvec, self.shape[0], self.shape[1])

    def todense(self):
        """Dense float64 matrix (small operators only; Tikhonov.py:20, demo_1D_deblurring)."""
        n = self.shape[1]
        if n > 1 << 14:
            raise MemoryError("todense() is for small operators")
        return self._dot(np.eye(n), False)

    def __repr__(self):
        return f"<{self.shape[0]}x{self.shape[1]} {type(self).__name__} on {self.engine.device}>"


class _Transposed(LinearOperator):
    def __init__(self, op):
        self.op = op
        self.shape = (op.shape[1], op.shape[0])
        self.engine = op.engine

    def apply(self, x, out=None, transpose=False, sumsq=None):
        return self.op.apply(x, out, not transpose, sumsq)

    def _dot(self, x, transpose):
        return self.op._dot(x, not transpose)

    @property
    def T(self):
        return self.op

    H = T


class _HandleOperator(LinearOperator):
    """An operator that owns a trk_op handle."""

    def __init__(self, handle, engine):
        self._h = handle
        rows, cols = ctypes.c_int64(), ctypes.c_int64()
        _lib.check(engine.lib.trk_op_shape(handle, ctypes.byref(rows), ctypes.byref(cols)), "trk_op_shape")
        super().__init__((rows.value, cols.value), engine)

    def _apply(self, x2, y2, transpose, sumsq):
        self.engine.op_apply(self._h, transpose, x2, y2, batch=x2.shape[0], ldx=x2.stride(0), ldy=y2.stride(0), sumsq=sumsq)

    #: hints of apply_axpby (trk.h TRK_HINT_*)
    OUT_FEEDS_OPPOSITE, INPUT_FROM_OPPOSITE, SUMSQ_DEFERRED = 1, 2, 4

    def apply_axpby(self, x, a, b, z, out, transpose=False, sumsq=None, hints=0):
        """out = a * Op(x) + b * z and ||out||^2 in the operator's own output pass (trk_op_apply_axpby: one Golub-Kahan half
        step; a, b: float or engine.Coef; z may be None).  Operators without a native form run apply + axpby in C."""
        self.engine.op_apply_axpby(self._h, transpose, x, a, b, z, out, sumsq, hints)

    def flush_deferred(self):
        """Finish a norm an apply_axpby(..., hints=SUMSQ_DEFERRED) left as block partials (trk_op_flush)."""
        _lib.check(self.engine.lib.trk_op_flush(self._h, self.engine.stream()), "trk_op_flush")

    @property
    def native_axpby(self):
        n = ctypes.c_int(0)
        _lib.check(self.engine.lib.trk_op_axpby_caps(self._h, ctypes.byref(n)), "trk_op_axpby_caps")
        return bool(n.value)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                self.engine.lib.trk_op_destroy(h)
            except Exception:
                pass


def _dbl_array(a):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float64))
    return a, a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


class Blur2D(_HandleOperator):
    """y = scipy.ndimage.convolve(x.reshape(nx,ny), psf, mode='reflect'); `.T` = the same with the flipped PSF."""

    def __init__(self, psf, nx, ny, engine=None):
        engine = engine if engine is not None else default_engine()
        psf = np.asarray(psf, dtype=np.float64)
        if psf.ndim != 2:
            raise ValueError("psf must be 2-D")
        self.psf, self.nx, self.ny = psf, int(nx), int(ny)
        arr, p = _dbl_array(psf)
        h = ctypes.c_void_p()
        _lib.check(engine.lib.trk_blur2d_create(p, psf.shape[0], psf.shape[1], self.nx, self.ny, ctypes.byref(h)), "trk_blur2d_create")
        super().__init__(h, engine)
        self.streaming = max(psf.shape) <= 15          # the sliding-window / LDS-strip kernels (csrc/blur2d.hip)


class Blur1D(Blur2D):
    """1-D blur with a (possibly length-n) PSF: an n x 1 image and a k x 1 PSF."""

    def __init__(self, psf, n=None, engine=None):
        psf = np.asarray(psf, dtype=np.float64).reshape(-1)
        n = len(psf) if n is None else int(n)
        super().__init__(psf.reshape(-1, 1), n, 1, engine)


class Radon2DParallel(_HandleOperator):
    """Parallel-beam Joseph projector; sinogram (n_ang, n_det) row-major; scale defaults to 1/N (io.py:397-399)."""

    def __init__(self, N, angles, n_det=None, scale=None, engine=None):
        engine = engine if engine is not None else default_engine()
        self.N = int(N)
        self.angles = np.asarray(angles, dtype=np.float64).reshape(-1)
        self.n_det = self.N if n_det is None else int(n_det)
        self.scale = 1.0 / self.N if scale is None else float(scale)
        arr, p = _dbl_array(self.angles)
        h = ctypes.c_void_p()
        _lib.check(engine.lib.trk_radon2d_create(self.N, self.n_det, p, len(self.angles), self.scale, ctypes.byref(h)), "trk_radon2d_create")
        super().__init__(h, engine)

    # ---- the float64 instrument (csrc/ref64.hip; diagnostics, not a fast path) ----
    ARITHMETIC = {"product": 0, "float64": 1, "tables64": 2}

    def set_ref_sums(self, chunk_fwd=0, chunk_adj=0):
        """The instrument's kernels with emulated fp32 partial sums (trk_radon2d_set_ref_sums): fp32 running sums moved to a float64
        total every chunk_fwd marching steps (forward) / chunk_adj angles (adjoint); 0 = float64 sums."""
        _lib.check(self.engine.lib.trk_radon2d_set_ref_sums(self._h, int(chunk_fwd), int(chunk_adj)), "trk_radon2d_set_ref_sums")
        return self

    def set_arithmetic(self, mode):
        """The arithmetic every apply of this operator runs in (trk_radon2d_set_arithmetic): 'product' — the fast kernels;
        'float64' — interpolation weights from the geometry in float64, float64 sums; 'tables64' — the fast kernels' fixed-point
        weights, float64 sums.  Vectors stay fp32; solvers work unchanged (fused riders run in launches of their own)."""
        _lib.check(self.engine.lib.trk_radon2d_set_arithmetic(self._h, self.ARITHMETIC[mode]), "trk_radon2d_set_arithmetic")
        return self

    def apply_ref(self, x, transpose=False, weights="float64"):
        """Op(x) by the float64-arithmetic kernels on a float32 OR float64 device vector (trk_radon2d_apply_ref)."""
        if x.dtype not in (torch.float32, torch.float64) or not x.is_contiguous():
            raise ValueError("apply_ref: contiguous float32 / float64 device vector expected")
        nin, nout = (self.shape[0], self.shape[1]) if transpose else (self.shape[1], self.shape[0])
        if x.numel() != nin:
            raise ValueError(f"dimension mismatch: operator expects {nin}, got {x.numel()}")
        y = torch.empty(nout, dtype=x.dtype, device=x.device)
        _lib.check(self.engine.lib.trk_radon2d_apply_ref(self._h, int(bool(transpose)), x.element_size(), {"float64": 0, "tables64": 1}[weights],
                                                         x.data_ptr(), y.data_ptr(), self.engine.stream()), "trk_radon2d_apply_ref")
        return y


class FanBeam2D(_HandleOperator):
    """Fan-beam flat-detector line projector with the geometry defaults of Tomography.define_proj_id
    (trips/test_problems/Tomography.py:48-67): p = int(sqrt(2) N) detector pixels, source-origin 3N, origin-detector N,
    detector pitch (SOD+ODD)/SOD, theta = linspace(0, pi, views, endpoint=False)."""

    def __init__(self, N, views=None, angles=None, n_det=None, source_origin=None, origin_detector=None, det_pitch=None,
                 engine=None):
        engine = engine if engine is not None else default_engine()
        self.N = int(N)
        self.angles = (np.linspace(0, np.pi, int(views), endpoint=False) if angles is None
                       else np.asarray(angles, dtype=np.float64).reshape(-1))
        self.n_det = int(np.sqrt(2) * self.N) if n_det is None else int(n_det)
        self.sod = 3.0 * self.N if source_origin is None else float(source_origin)
        self.odd = 1.0 * self.N if origin_detector is None else float(origin_detector)
        self.pitch = (self.sod + self.odd) / self.sod if det_pitch is None else float(det_pitch)
        arr, p = _dbl_array(self.angles)
        h = ctypes.c_void_p()
        _lib.check(engine.lib.trk_fanbeam2d_create(self.N, self.n_det, self.pitch, self.sod, self.odd, p, len(self.angles),
                                                   ctypes.byref(h)), "trk_fanbeam2d_create")
        super().__init__(h, engine)


class _FusedTV:
    """Fused forms of a first-difference regulariser for the re-weighted solvers (trk_tv_weights / trk_tv_grad): no vector of
    the length of L x in between.  `fused_tv` says whether this operator instance has them."""
    fused_tv = True

    def tv_weights(self, x, eps, q, out):
        """out = ((L x)^2 + eps^2)^(q/2-1) (MMGKS.py:60,93)."""
        _lib.check(self.engine.lib.trk_tv_weights(self._h, x.data_ptr(), float(eps), float(q), out.data_ptr(),
                                                  self.engine.stream()), "trk_tv_weights")

    def tv_grad(self, x, w, r_in, lam, out, dot_with=None, dot_out=None, xsq_out=None):
        """out = r_in + lam * L^T (w .* (L x)) (MMGKS.py:116-118); w None: unit weights (GKS.py:81-84); r_in None: 0.
        dot_with / dot_out: also <out, dot_with> into the device scalar dot_out, from the same pass (trk_tv_grad_dot);
        xsq_out (with them): and <x, x> (trk_tv_grad_dot_xsq)."""
        if dot_with is not None and xsq_out is not None:
            rc = self.engine.lib.trk_tv_grad_dot_xsq(self._h, x.data_ptr(), None if w is None else w.data_ptr(),
                                                     None if r_in is None else r_in.data_ptr(), float(lam), out.data_ptr(),
                                                     dot_with.data_ptr(), dot_out if isinstance(dot_out, int) else dot_out.data_ptr(),
                                                     xsq_out if isinstance(xsq_out, int) else xsq_out.data_ptr(), self.engine.stream())
            _lib.check(rc, "trk_tv_grad_dot_xsq")
            return
        if dot_with is not None:
            rc = self.engine.lib.trk_tv_grad_dot(self._h, x.data_ptr(), None if w is None else w.data_ptr(),
                                                 None if r_in is None else r_in.data_ptr(), float(lam), out.data_ptr(),
                                                 dot_with.data_ptr(), dot_out if isinstance(dot_out, int) else dot_out.data_ptr(),
                                                 self.engine.stream())
            _lib.check(rc, "trk_tv_grad_dot")
            return
        rc = self.engine.lib.trk_tv_grad(self._h, x.data_ptr(), None if w is None else w.data_ptr(),
                                         None if r_in is None else r_in.data_ptr(), float(lam), out.data_ptr(),
                                         self.engine.stream())
        _lib.check(rc, "trk_tv_grad")


class FirstDerivative2D(_FusedTV, _HandleOperator):
    def __init__(self, N, engine=None):
        engine = engine if engine is not None else default_engine()
        self.N = int(N)
        h = ctypes.c_void_p()
        _lib.check(engine.lib.trk_deriv2d_create(self.N, ctypes.byref(h)), "trk_deriv2d_create")
        super().__init__(h, engine)
        self.streaming = True


class SpaceTimeDerivative(_FusedTV, _HandleOperator):
    """Space-time first differences over this rank's `nt_local` frames (trips/utilities/operators.py:39-45).

    Time-sharded (engine.world > 1), two ways to meet the rows x_t - x_{t+1} that cross a rank boundary:
      * the fused forms `tv_weights` / `tv_grad` (what GKS / MMGKS use; the kernels one rank runs): the operand's two boundary
        frames of the neighbour ranks — `halo_frames(x)`, ONE two-sided exchange — and the rank forms its own pixels of
        L^T (w .* L x) completely.  `halo=` takes frames the caller already has (GKS derives them for x = V y and v_new from the
        basis vectors' halos: one exchange per iteration);
      * plain `L @ x` / `L.T @ y`: one frame moves to the neighbour before each apply (forward: first frame of the next rank;
        transpose: last temporal block of the previous rank)."""

    def __init__(self, N, nt, engine=None):
        engine = engine if engine is not None else default_engine()
        self.N, self.nt_global = int(N), int(nt)
        w, r = engine.world, engine.rank
        if self.nt_global % w:
            raise ValueError(f"{nt} frames do not shard evenly over {w} ranks")
        self.nt_local = self.nt_global // w
        self.has_next, self.has_prev = r < w - 1, r > 0
        h = ctypes.c_void_p()
        _lib.check(engine.lib.trk_spacetime_create(self.N, self.nt_local, int(self.has_next), int(self.has_prev), ctypes.byref(h)),
                   "trk_spacetime_create")
        super().__init__(h, engine)
        npix = self.N * self.N
        self.npix = npix
        self._halo_next = engine.empty(npix) if self.has_next else None
        self._halo_prev = engine.empty(npix) if self.has_prev else None
        self._xh = engine.empty(2 * npix) if w > 1 else None       # [previous rank's last frame | next rank's first frame]
        self._ps = 2 * self.N * (self.N - 1)
        self.sharded = w > 1
        self.streaming = True
        self.fused_tv = True
        # weights of the fused forms: the rows of L, plus (has_prev) the previous rank's boundary row recomputed here
        self.tv_weights_len = self.shape[0] + (npix if self.has_prev else 0)

    # ---- fused forms over a time-sharded vector
    def halo_frames(self, x, out=None):
        """The neighbour ranks' boundary frames of the time-sharded vector x, [prev's last | next's first] (2 N^2 floats; a half
        without a neighbour is left as it is): ONE two-sided exchange."""
        eng, npix = self.engine, self.npix
        out = self._xh if out is None else out
        eng.halo_exchanges += 1
        if eng.world > 1:
            eng.comm.exchange2(x[:npix], out[:npix], x[(self.nt_local - 1) * npix:self.nt_local * npix], out[npix:2 * npix])
        return out

    def _give_halo(self, x, halo):
        if not self.sharded:
            return
        hf = self.halo_frames(x) if halo is None else halo
        _lib.check(self.engine.lib.trk_tv_halo(self._h, hf.data_ptr() if self.has_prev else None,
                                               hf[self.npix:].data_ptr() if self.has_next else None), "trk_tv_halo")

    def tv_weights(self, x, eps, q, out, halo=None):
        if out.numel() < self.tv_weights_len:
            # (a rank with a previous neighbour also writes that neighbour's boundary row: N^2 more than L has rows)
            raise ValueError(f"tv_weights: out holds {out.numel()} floats, this operator writes tv_weights_len = {self.tv_weights_len}")
        self._give_halo(x, halo)
        super().tv_weights(x, eps, q, out)

    def tv_grad(self, x, w, r_in, lam, out, dot_with=None, dot_out=None, halo=None, xsq_out=None):
        if w is not None and w.numel() < self.tv_weights_len:
            raise ValueError(f"tv_grad: w holds {w.numel()} floats, this operator reads tv_weights_len = {self.tv_weights_len}")
        self._give_halo(x, halo)
        super().tv_grad(x, w, r_in, lam, out, dot_with=dot_with, dot_out=dot_out, xsq_out=xsq_out)

    def _apply(self, x2, y2, transpose, sumsq):
        eng = self.engine
        if eng.world > 1:
            if x2.shape[0] != 1:
                raise NotImplementedError("sharded space-time operator applies one vector at a time")
            eng.halo_exchanges += 1
            spacetime_halo_exchange(eng, x2[0], transpose, self.N, self.nt_local, self._halo_next, self._halo_prev)
            _lib.check(eng.lib.trk_spacetime_set_halo(self._h, None if self._halo_next is None else self._halo_next.data_ptr(),
                                                      None if self._halo_prev is None else self._halo_prev.data_ptr()),
                       "trk_spacetime_set_halo")
        super()._apply(x2, y2, transpose, sumsq)


def spacetime_halo_exchange(eng, x, transpose, N, nt_local, halo_next, halo_prev):
    """The one-frame neighbour exchange of the time-sharded space-time regulariser (operators.py:39-45 of the reference
    couples frame t with t+1 through the rows x_t - x_{t+1}).

    forward  : every rank but the first sends its FIRST frame to the previous rank (which owns the row coupling its
               last frame to it) and receives the next rank's first frame into `halo_next`.
    transpose: every rank but the last sends its LAST temporal block (the row it owns on behalf of the boundary) to the
               next rank and receives the previous rank's into `halo_prev`.
    x is the rank-local operand (frame-major image block for forward; [spatial rows | temporal rows] for transpose)."""
    npix, ps = N * N, 2 * N * (N - 1)
    has_next, has_prev = eng.rank < eng.world - 1, eng.rank > 0
    if not transpose:
        eng.comm.shift(send=x[:npix] if has_prev else None, send_to=eng.rank - 1,
                       recv=halo_next, recv_from=eng.rank + 1 if has_next else None)
    else:
        last = x[nt_local * ps + (nt_local - 1) * npix:nt_local * ps + nt_local * npix] if has_next else None
        eng.comm.shift(send=last, send_to=eng.rank + 1,
                       recv=halo_prev, recv_from=eng.rank - 1 if has_prev else None)


class BlockDiagOp(_HandleOperator):
    """F = blkdiag(A_0 .. A_{T-1}) over THIS RANK's frames; x and b are frame-major."""

    def __init__(self, ops, engine=None):
        engine = engine if engine is not None else ops[0].engine
        self.ops = list(ops)
        h = ctypes.c_void_p()
        first = self.ops[0]
        same_radon = all(isinstance(o, Radon2DParallel) and o.N == first.N and o.n_det == first.n_det
                         and o.scale == first.scale and len(o.angles) == len(first.angles) for o in self.ops)
        if len(self.ops) > 1 and all(type(o) is SparseOp for o in self.ops):
            # frames given as sparse matrices (the blocks io.py:223-225 cuts out of the real data's forward matrix): ONE CSR of the
            # block-diagonal matrix, one launch per apply whatever the number of frames
            import scipy.sparse as sp
            merged = SparseOp(sp.block_diag([o.matrix for o in self.ops], format="csr"), engine=engine)
            self.matrix = merged.matrix
            h, merged._h = merged._h, None
        elif same_radon and len(self.ops) > 1:
            # frames of one dynamic tomography problem: a single handle runs all frames per launch
            ang = np.concatenate([o.angles for o in self.ops])
            arr, p = _dbl_array(ang)
            _lib.check(engine.lib.trk_radon2d_dynamic_create(first.N, first.n_det, p, len(self.ops), len(first.angles),
                                                             first.scale, ctypes.byref(h)), "trk_radon2d_dynamic_create")
        else:
            arr = (ctypes.c_void_p * len(self.ops))(*[o._h for o in self.ops])
            _lib.check(engine.lib.trk_blockdiag_create(arr, len(self.ops), ctypes.byref(h)), "trk_blockdiag_create")
        super().__init__(h, engine)


class SparseOp(_HandleOperator):
    """An operator given as a (scipy.sparse or small dense) matrix: CSR SpMV on the device, both directions as gathers.
    This is what a reference-built regulariser (`gen_first_derivative_operator_2D`, `gen_spacetime_derivative_operator`,
    the framelet analysis matrices: trips/utilities/operators.py) or a precomputed sparse forward matrix
    (io.py:197-229) becomes when handed to the engine's solvers."""

    def __init__(self, M, engine=None):
        import scipy.sparse as sp
        engine = engine if engine is not None else default_engine()
        A = sp.csr_matrix(M)
        A.sum_duplicates()
        At = A.T.tocsr()
        At.sum_duplicates()
        self.matrix = A

        def arrs(C):
            return (np.ascontiguousarray(C.indptr, dtype=np.int64), np.ascontiguousarray(C.indices, dtype=np.int32),
                    np.ascontiguousarray(C.data, dtype=np.float32))
        ip, ix, dv = arrs(A)
        tp, tx, tv = arrs(At)
        h = ctypes.c_void_p()
        _lib.check(engine.lib.trk_csr_create(A.shape[0], A.shape[1], A.nnz, ip.ctypes.data, ix.ctypes.data, dv.ctypes.data,
                                             tp.ctypes.data, tx.ctypes.data, tv.ctypes.data, ctypes.byref(h)), "trk_csr_create")
        super().__init__(h, engine)


def slice_dynamic_frames(A, b, nt, rows_per_frame, cols_per_frame):
    """The per-frame blocks of a dynamic problem's forward matrix and data, exactly as the reference's loaders cut them
    (trips/utilities/io.py:223-225, generate_crossPhantom: `AA[ii] = A_small[700*ii:700*(ii+1), 16384*ii:16384*(ii+1)]`,
    `B[ii] = b[700*ii:700*(ii+1)]`; io.py:160-162 does the same for the emoji data): frame ii owns rows
    [ii rows_per_frame, (ii+1) rows_per_frame) and columns [ii cols_per_frame, (ii+1) cols_per_frame); whatever the matrix
    holds outside those blocks is dropped.  Returns (list of scipy.sparse CSR blocks, list of data blocks)."""
    import scipy.sparse as sp
    A = sp.csr_matrix(A)
    nt, rpf, cpf = int(nt), int(rows_per_frame), int(cols_per_frame)
    if A.shape[0] < nt * rpf or A.shape[1] < nt * cpf:
        raise ValueError(f"slice_dynamic_frames: a {A.shape[0]} x {A.shape[1]} matrix has no {nt} blocks of {rpf} x {cpf}")
    bb = None if b is None else np.asarray(b).reshape(-1)
    AA = [A[rpf * ii:rpf * (ii + 1), cpf * ii:cpf * (ii + 1)].tocsr() for ii in range(nt)]
    B = None if bb is None else [bb[rpf * ii:rpf * (ii + 1)] for ii in range(nt)]
    return AA, B


class SparseBlockDiag(SparseOp):
    """F = blkdiag(A_0 .. A_{T-1}) of sparse frame matrices as ONE CSR operator on the device: x and b frame-major, one launch per
    apply (pylops.BlockDiag over per-frame matrices, io.py:420, for the sparse real-data problems of io.py:197-229).  On several
    ranks every rank holds the blocks of its own frames (dist.frame_range)."""

    def __init__(self, blocks, engine=None):
        import scipy.sparse as sp
        self.blocks = [sp.csr_matrix(B) for B in blocks]
        super().__init__(sp.block_diag(self.blocks, format="csr"), engine=engine)

    @classmethod
    def from_matrix(cls, A, nt, rows_per_frame, cols_per_frame, engine=None):
        """From the whole forward matrix, sliced as the reference slices it (slice_dynamic_frames); with engine.world > 1 only this
        rank's frames are kept."""
        from .dist import frame_range
        engine = engine if engine is not None else default_engine()
        AA, _ = slice_dynamic_frames(A, None, nt, rows_per_frame, cols_per_frame)
        lo, hi = frame_range(int(nt), getattr(engine, "world", 1), getattr(engine, "rank", 0))
        return cls(AA[lo:hi], engine=engine)


def create_framelet_operator(n, m, l, engine=None):
    """The reference's framelet analysis operator (trips/utilities/operators.py:50-113) as ONE sparse matrix on the device:
    vec_F(W_n X W_m^H) = kron(W_m, W_n) vec_F(X) for the reference's column-major reshapes (:106-108); rows
    n(2l+1) * m(2l+1), columns n*m."""
    import scipy.sparse as sp

    def construct_H(lev, nn):                       # operators.py:50-85
        e = np.ones((nn,))
        H0 = (sp.spdiags(e, -lev, nn, nn) + sp.spdiags(2 * e, 0, nn, nn) + sp.spdiags(e, lev, nn, nn)).tolil()
        H1 = (sp.spdiags(-e, -lev, nn, nn) + sp.spdiags(e, lev, nn, nn)).tolil()
        H2 = (sp.spdiags(-e, -lev, nn, nn) + sp.spdiags(2 * e, 0, nn, nn) + sp.spdiags(-e, lev, nn, nn)).tolil()
        for jj in range(lev):
            H0[jj, lev - jj - 1] += 1
            H0[-jj - 1, -lev + jj] += 1
            H1[jj, lev - jj - 1] -= 1
            H1[-jj - 1, -lev + jj] += 1
            H2[jj, lev - jj - 1] -= 1
            H2[-jj - 1, -lev + jj] -= 1
        return H0.tocsr() / 4, H1.tocsr() * (np.sqrt(2) / 4), H2.tocsr() / 4

    def analysis(nn, level, w):                     # operators.py:88-103
        if level == l:
            return sp.vstack(construct_H(level, nn))
        H0, H1, H2 = construct_H(level, nn)
        return sp.vstack((analysis(nn, level + 1, H0), H1, H2)) * w

    W_n, W_m = analysis(n, 1, 1), analysis(m, 1, 1)
    return SparseOp(sp.kron(W_m, W_n, format="csr"), engine=engine)


# --------------------------------------------------------------------------------------------------------------------
# The PyLops-built regularisers of trips/utilities/operators_old.py:22-61 (what MMGKS's isoTV branch expects of `L`,
# MMGKS.py:61-77).  PyLops is an un-pinned, absent dependency of the reference: `pylops.FirstDerivative(n)` is restated
# here from its published default (kind="centered", 3-point, edge=False): row i = (x[i+1] - x[i-1]) / 2 for
# 1 <= i <= n-2, first and last row zero — PARITY UNPINNED for that stencil.
def _centered_first_derivative_matrix(n):
    import scipy.sparse as sp
    i = np.arange(1, n - 1)
    return sp.csr_matrix((np.concatenate((np.full(i.size, -0.5), np.full(i.size, 0.5))),
                          (np.concatenate((i, i)), np.concatenate((i - 1, i + 1)))), shape=(n, n))


def _first_derivative_2d_matrix(nx, ny):
    import scipy.sparse as sp
    if nx != ny:      # the reference's VStack of an nx^2- and an ny^2-column operator fails as well (operators_old.py:43)
        raise ValueError("first_derivative_operator_2d: the reference's operator exists for nx == ny only")
    return sp.vstack((sp.kron(sp.identity(nx), _centered_first_derivative_matrix(nx)),
                      sp.kron(_centered_first_derivative_matrix(ny), sp.identity(ny)))).tocsr()


def first_derivative_operator(n, engine=None):
    """operators_old.py:22-33 (`pylops.FirstDerivative(n)`): n x n, centered, zero first/last row."""
    return SparseOp(_centered_first_derivative_matrix(n), engine=engine)


def first_derivative_operator_2d(nx, ny, engine=None):
    """operators_old.py:35-45: VStack(Kronecker(I_nx, D_nx), Kronecker(D_ny, I_ny)), 2 nx^2 x nx^2."""
    return SparseOp(_first_derivative_2d_matrix(nx, ny), engine=engine)


def spatial_derivative_operator(nx, ny, nt, engine=None):
    """operators_old.py:47-53: Kronecker(I_nt, first_derivative_operator_2d) on a frame-major vector."""
    import scipy.sparse as sp
    return SparseOp(sp.kron(sp.identity(nt), _first_derivative_2d_matrix(nx, ny)).tocsr(), engine=engine)


def time_derivative_operator(nx, ny, nt, engine=None):
    """operators_old.py:55-61: Kronecker(D_nt, I_{nx^2})."""
    import scipy.sparse as sp
    return SparseOp(sp.kron(_centered_first_derivative_matrix(nt), sp.identity(nx ** 2)).tocsr(), engine=engine)


class VStack(LinearOperator):
    """pylops.VStack(ops): operators stacked by rows (the way the dynamic demos assemble space + time regularisers).
    Sparse-matrix operators are merged into one CSR operator; anything else is applied block by block."""

    def __new__(cls, ops, engine=None):
        ops = list(ops)
        if ops and all(isinstance(o, SparseOp) for o in ops):
            import scipy.sparse as sp
            return SparseOp(sp.vstack([o.matrix for o in ops]).tocsr(), engine=engine if engine is not None else ops[0].engine)
        return super().__new__(cls)

    def __init__(self, ops, engine=None):
        self.ops = list(ops)
        if not self.ops or any(o.shape[1] != self.ops[0].shape[1] for o in self.ops):
            raise ValueError("VStack: operators must share their column count")
        super().__init__((sum(o.shape[0] for o in self.ops), self.ops[0].shape[1]),
                         engine if engine is not None else self.ops[0].engine)
        self._ro = np.cumsum([0] + [o.shape[0] for o in self.ops])
        self._tmp = None

    def _apply(self, x2, y2, transpose, sumsq):
        eng = self.engine
        for j in range(x2.shape[0]):
            if not transpose:
                for i, o in enumerate(self.ops):
                    o.apply(x2[j], out=y2[j, self._ro[i]:self._ro[i + 1]])
            else:
                if self._tmp is None:
                    self._tmp = eng.empty(self.shape[1])
                for i, o in enumerate(self.ops):
                    seg = x2[j, self._ro[i]:self._ro[i + 1]]
                    if i == 0:
                        o.apply(seg, out=y2[j], transpose=True)
                    else:
                        o.apply(seg, out=self._tmp, transpose=True)
                        eng.axpby(1.0, y2[j], 1.0, self._tmp, y2[j])
        if sumsq is not None:
            if x2.shape[0] != 1:
                raise ValueError("VStack: fused sum of squares for single vectors only")
            eng.nrm2sq(y2[0], sumsq)


class Identity(LinearOperator):
    def __init__(self, n, engine=None):
        super().__init__((n, n), engine)

    def _apply(self, x2, y2, transpose, sumsq):
        y2.copy_(x2)
        if sumsq is not None:
            self.engine.nrm2sq(y2.reshape(-1), sumsq)


def is_identity(A):
    """Host predicate selecting the L = I branch (trips/utilities/utils.py:47-62)."""
    if isinstance(A, Identity):
        return True
    if isinstance(A, LinearOperator):
        return False
    if hasattr(A, "shape") and len(A.shape) == 2 and A.shape[0] == A.shape[1]:
        try:
            import scipy.sparse as sp
            if sp.issparse(A):
                return abs((A - sp.eye(A.shape[0])).sum()) < 1e-6
            return bool(np.allclose(np.asarray(A), np.eye(A.shape[0])))
        except Exception:
            return False
    return False
