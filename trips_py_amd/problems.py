"""Problem builders: the operator constructors of the reference's test problems plus seeded synthetic data
(the reference's own generators are unseeded: Deblurring2D.py:143, Tomography.py:206).

  gauss_psf(dim, spread)              <- Deblurring2D.Gauss            trips/test_problems/Deblurring2D.py:48-64
  gauss_psf_1d(n, sigma)              <- Deblurring1D.Gauss1D          trips/test_problems/Deblurring1D.py:63-69
  Deblurring2D().forward_Op(...)      <- Deblurring2D.forward_Op       :66-73   (returns a trips_py_amd Blur2D)
  synthetic_image / add_noise         seeded versions of the data recipe (:141-146): what bench.py / smoke() / the tests use

In scope here: the PSF and the operator constructors.  The demos' host-side data preparation — gen_xtrue,
gen_data, the classes' unseeded add_noise (OUT OF SCOPE) — is not part of the package: tools/demo_helpers.py
holds it as mixins (`demo_classes()`), for a notebook that wants the reference's whole class surface.
"""
import numpy as np

from .operators import Blur1D, Blur2D, BlockDiagOp, FanBeam2D, Radon2DParallel


def gauss_psf(dim, spread):
    """PSF = exp(-(X^2/s1^2 + Y^2/s2^2)/2) on X = arange(-fix(n/2), ceil(n/2)), normalised; centre = argmax."""
    m, n = int(dim[0]), int(dim[1])
    if isinstance(spread, (int, float)):
        s1 = s2 = float(spread)
    else:
        s1, s2 = float(spread[0]), float(spread[1])
    xs = np.arange(-np.fix(n / 2), np.ceil(n / 2))
    ys = np.arange(-np.fix(m / 2), np.ceil(m / 2))
    X, Y = np.meshgrid(xs, ys)
    psf = np.exp(-0.5 * ((X ** 2) / (s1 ** 2) + (Y ** 2) / (s2 ** 2)))
    psf /= psf.sum()
    mm, nn = np.where(psf == psf.max())
    return psf, np.array([mm[0], nn[0]]).astype(int)


def gauss_psf_1d(n, sigma):
    x = np.arange(-np.fix(n / 2), np.ceil(n / 2))
    psf = np.exp(-0.5 * ((x ** 2) / (sigma ** 2)))
    return psf / psf.sum()


class Deblurring2D:
    """Operator-constructor subset of trips.test_problems.Deblurring2D (same method names)."""

    def __init__(self, **kwargs):
        self.nx = self.ny = None
        self.CommitCrime = kwargs.get("CommitCrime", False)

    def Gauss(self, PSFdim, PSFspread):
        self.dim, self.spread = PSFdim, PSFspread
        return gauss_psf(PSFdim, PSFspread)

    def forward_Op(self, dim, spread, nx, ny, engine=None):
        self.nx, self.ny = nx, ny
        psf, _ = self.Gauss(dim, spread)
        return Blur2D(psf, nx, ny, engine=engine)


class Deblurring1D:
    """trips.test_problems.Deblurring1D: the operator constructor on the engine (Deblurring1D.py:63-69, 93-102) — BASELINE config C1
    (n = 256, sigma = 3)."""

    def __init__(self, **kwargs):
        self.grid_points = self.ny = self.parameter = self.boundary_condition = None
        self.CommitCrime = kwargs.get("CommitCrime", False)

    def Gauss1D(self, grid_points, parameter):
        self.grid_points = grid_points
        psf = gauss_psf_1d(grid_points, parameter)
        return psf, int(np.where(psf == psf.max())[0][0])

    def forward_Op_1D(self, parameter, nx, boundary_condition="reflect", engine=None):
        if boundary_condition != "reflect":
            raise NotImplementedError("only the 'reflect' boundary is implemented on the engine")
        self.parameter, self.boundary_condition = parameter, boundary_condition
        self.PSF, self.center = self.Gauss1D(nx, parameter)
        return Blur1D(self.PSF, nx, engine=engine)


class Tomography:
    """Operator-constructor subset of trips.test_problems.Tomography (same method name and return convention)."""

    def __init__(self, **kwargs):
        self.nx = self.ny = None
        self.CommitCrime = kwargs.get("CommitCrime", False)

    def forward_Op(self, nx, ny, views, engine=None):
        """Fan-beam operator of Tomography.py:78-88.  With CommitCrime=False the reference also returns a second operator
        whose angles are shifted by 1e-8 (:61-65); the same triple / pair is returned here."""
        self.nx, self.ny, self.q = nx, ny, views
        self.p = int(np.sqrt(2) * nx)
        A = FanBeam2D(nx, views=views, engine=engine)
        if not self.CommitCrime:
            A_mis = FanBeam2D(nx, angles=A.angles + 1e-8, engine=engine)
            return A, A, A_mis
        return A, A


def parallel_beam_frames(N, angle_sets, engine=None):
    """One Radon2DParallel per time frame (io.py:391-420), combined frame-major with BlockDiagOp."""
    ops = [Radon2DParallel(N, ang, engine=engine) for ang in angle_sets]
    return BlockDiagOp(ops, engine=engine), ops


# ---------------------------------------------------------------------------- seeded synthetic data
def synthetic_image(N, seed=0):
    """N x N float32 test image: piecewise-constant rectangles + 0.1*U(0,1) texture."""
    rng = np.random.default_rng(seed)
    img = np.zeros((N, N), dtype=np.float64)
    for _ in range(8):
        i0, j0 = rng.integers(0, max(1, N - N // 8), size=2)
        h, w = rng.integers(max(1, N // 16), max(2, N // 3), size=2)
        img[i0:i0 + h, j0:j0 + w] += rng.uniform(0.2, 1.0)
    img += 0.1 * rng.random((N, N))
    return img


def add_noise(b_true, level, seed=1):
    """b = b_true + e, e ~ N(0,1) scaled to ||e|| = level*||b_true||; returns (b, delta=||e||)."""
    rng = np.random.default_rng(seed)
    e = rng.standard_normal(b_true.shape)
    e *= level * np.linalg.norm(b_true) / np.linalg.norm(e)
    return b_true + e, float(np.linalg.norm(e))
