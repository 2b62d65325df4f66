"""trips_py_amd — an MI355X-native (gfx950) engine for TRIPs-Py's Krylov regularization hot path.

Operators (`trips_py_amd.operators`) expose the PyLops matvec/rmatvec surface; solvers
(`trips_py_amd.solvers`) keep the `trips.solvers.*` signatures and `info` dictionaries; all vector
arithmetic runs in hand-written HIP kernels behind the C ABI of include/trk.h (libtrk.so).
"""
__version__ = "0.1.0"

from ._lib import TrkError, build, lib_path  # noqa: F401
