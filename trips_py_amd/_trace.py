"""Tracing hooks of the solver loops: roctx ranges around solver phases and the reference's tqdm progress.

  * `phase(name)`: a context manager that opens / closes a roctx range (`roctxRangePushA` / `roctxRangePop` of
    librocprofiler-sdk-roctx / libroctx64, resolved at run time) so that `rocprofv3 --marker-trace` shows which kernels belong to
    which phase of an iteration (projected solve, iterate, residual, orthogonalisation, Gram rows ...).  Off unless TRK_TRACE=1:
    the loops enqueue microsecond kernels and even an empty Python context manager is not free.
  * `progress(iterable, desc)`: the reference wraps its loops in tqdm (GKS.py:42, Hybrid_LSQR.py:73, decompositions.py:76,165).  The
    engine's loops run ahead of the GPU, so a bar would show enqueueing, not solving; it is therefore opt-in (TRK_PROGRESS=1 or
    the solvers' `progress=True`) and falls back to the bare iterable when tqdm is absent.
"""
import contextlib
import ctypes
import os

_ON = os.environ.get("TRK_TRACE", "0") not in ("", "0")
_lib = None


def _roctx():
    global _lib
    if _lib is None:
        _lib = False
        for name in ("librocprofiler-sdk-roctx.so", "libroctx64.so"):
            try:
                lib = ctypes.CDLL(name)
                lib.roctxRangePushA.argtypes = [ctypes.c_char_p]
                lib.roctxRangePushA.restype = ctypes.c_int
                lib.roctxRangePop.restype = ctypes.c_int
                _lib = lib
                break
            except (OSError, AttributeError):
                continue
    return _lib


@contextlib.contextmanager
def _range(name):
    lib = _roctx()
    if lib:
        lib.roctxRangePushA(name.encode())
    try:
        yield
    finally:
        if lib:
            lib.roctxRangePop()


_NULL = contextlib.nullcontext()


def phase(name):
    """roctx range `name` around the enclosed enqueues (TRK_TRACE=1), else a no-op."""
    return _range(name) if _ON else _NULL


_open = [False]


def mark(name):
    """End the phase opened by the previous mark() and, with a name, open the next one: what a loop body calls at its phase
    boundaries (no re-indentation of the loop).  mark(None) after the loop closes the last.  A no-op unless TRK_TRACE=1."""
    if not _ON:
        return
    lib = _roctx()
    if not lib:
        return
    if _open[0]:
        lib.roctxRangePop()
        _open[0] = False
    if name is not None:
        lib.roctxRangePushA(name.encode())
        _open[0] = True


def enabled():
    return _ON


def progress(iterable, desc, want=None):
    """tqdm(iterable, desc) as the reference's loops show it, if asked for (want=True or TRK_PROGRESS=1) and tqdm is importable."""
    if want is None:
        want = os.environ.get("TRK_PROGRESS", "0") not in ("", "0")
    if not want:
        return iterable
    try:
        from tqdm import tqdm
    except ImportError:
        return iterable
    return tqdm(iterable, desc)
