"""Krylov / projection solvers with the reference's signatures (trips/solvers/*.py), running on the HIP engine."""
from .CGLS import CGLS, CGLSRun  # noqa: F401
