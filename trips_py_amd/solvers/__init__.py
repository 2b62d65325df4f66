"""Krylov / projection solvers with the reference's signatures (trips/solvers/*.py), running on the HIP engine."""
from .CGLS import CGLS, CGLSRun, CGLSRunFused, CGLSRunSharded  # noqa: F401
from .Hybrid_LSQR import Hybrid_LSQR  # noqa: F401
from .Hybrid_GMRES import Hybrid_GMRES  # noqa: F401
from .GKS import GKS  # noqa: F401
from .MMGKS import MMGKS  # noqa: F401
from .GK_Tikhonov import Golub_Kahan_Tikhonov  # noqa: F401
from .A_Tikhonov import Arnoldi_Tikhonov  # noqa: F401
from .GMRES import GMRES  # noqa: F401
