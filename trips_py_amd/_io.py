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
        return [H[i].reshape(-1, 1) for i in range(count)]


def history_fits(engine, count, n, what):
    """Refuse (loudly) an on-device history that cannot fit; the reference keeps every iterate (CGLS.py:66)."""
    if engine.device.type != "cuda":
        return
    need = int(count) * int(n) * 4
    free, _total = torch.cuda.mem_get_info(engine.device)
    if need > 0.8 * free:
        raise MemoryError(f"{what}: keeping {count} iterates of {n} floats needs {need / 2**30:.1f} GiB of HBM "
                          f"({free / 2**30:.1f} GiB free); pass history=False")
