"""Krylov / projection solvers with the reference's signatures (trips/solvers/*.py), running on the HIP engine."""
from .CGLS import CGLS, CGLSRun, CGLSRunFused  # noqa: F401
from .Hybrid_LSQR import Hybrid_LSQR  # noqa: F401
from .Hybrid_GMRES import Hybrid_GMRES  # noqa: F401
from .GKS import GKS  # noqa: F401
from .MMGKS import MMGKS  # noqa: F401
