"""Problem builders: the operator constructors of the reference's test problems plus seeded synthetic data
(the reference's own generators are unseeded: Deblurring2D.py:143, Tomography.py:206).

  gauss_psf(dim, spread)              <- Deblurring2D.Gauss            trips/test_problems/Deblurring2D.py:48-64
  gauss_psf_1d(n, sigma)              <- Deblurring1D.Gauss1D          trips/test_problems/Deblurring1D.py:63-69
  Deblurring2D().forward_Op(...)      <- Deblurring2D.forward_Op       :66-73   (returns a trips_py_amd Blur2D)
  synthetic_image / add_noise         seeded versions of the data recipe (:141-146)
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

    # The two data-generation helpers of the demos, on the HOST (one-time data preparation, not the hot path): same
    # arithmetic as Deblurring2D.py:123-159 so that a demo keeps working with this class swapped in.  Images come from the
    # caller (the reference's gen_true reads ./data/image_data/*.mat: dataset handling is out of scope).
    def gen_data(self, x):
        """b = blurred x.  CommitCrime=False (:125-137): blur on a zero-padded 2nx x 2ny canvas with 'constant' boundary and cut
        the centre out (the data then do not come from the reflective operator); True: the operator's own convolution."""
        from scipy.ndimage import convolve
        psf, _ = gauss_psf(self.dim, self.spread)
        im = np.asarray(x, dtype=np.float64).reshape((self.nx, self.ny))
        if self.CommitCrime is False:
            big = np.zeros((2 * self.nx, 2 * self.ny))
            i0, j0 = self.nx // 2, self.ny // 2
            big[i0:i0 + self.nx, j0:j0 + self.ny] = im
            b = convolve(big, psf, mode="constant")[i0:i0 + self.nx, j0:j0 + self.ny]
        else:
            b = convolve(im, psf, mode="reflect")
        return b.reshape((-1, 1))

    def add_noise(self, b_true, opt, noise_level):
        """(b_meas as an nx x ny image, delta) — :141-159 (unseeded, like the reference; seeded variant: problems.add_noise)."""
        b_true = np.asarray(b_true, dtype=np.float64)
        if opt == "Gaussian":
            e = np.random.randn(self.nx * self.ny, 1)
            sig = noise_level * np.linalg.norm(b_true) / np.linalg.norm(e)
            b_meas, delta = b_true + sig * e, np.linalg.norm(sig * e)
        elif opt == "Poisson":
            b_meas, delta = np.random.poisson(lam=b_true + 1), 0.0
        elif opt == "Laplace":
            e = np.random.laplace(self.nx * self.ny, 1)
            sig = noise_level * np.linalg.norm(b_true) / np.linalg.norm(e)
            b_meas, delta = b_true + sig * e, np.linalg.norm(sig * e)
        else:
            raise ValueError(f"unknown noise option {opt!r}")
        return np.asarray(b_meas).reshape((self.nx, self.ny)), delta


class Deblurring1D:
    """trips.test_problems.Deblurring1D: the operator constructor on the engine plus the demo's host-side data helpers
    (Deblurring1D.py:63-69, 104-143, 144-197, 199-216) — BASELINE config C1 (n = 256, 'curve0', sigma = 3)."""

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

    def gen_xtrue(self, N, test):
        """The test signals of :144-197."""
        self.grid_points, self.ny = N, 1
        if test == "sigma":
            x = np.linspace(-2.5, 2.5, N)
            return np.piecewise(x, [x < 0, x >= 0], [-1, 1])
        if test == "piecewise":
            xx = np.linspace(0, 1, N)
            edges = [0, 0.10, 0.15, 0.20, 0.25, 0.35, 0.38, 0.45, 0.55, 0.75, 0.8]
            values = [0, 1, 0, 0, 0, 0, 0, 0.25, 0, 1, 0]
            conds = [(edges[i] <= xx) & (xx < edges[i + 1]) for i in range(10)] + [(0.8 <= xx) & (xx <= 1)]
            return np.piecewise(xx, conds, values)
        if test == "curve0":
            h = np.pi / N
            t = -np.pi / 2 + np.arange(0.5, N, 1) * h
            return 2 * np.exp(-6 * (t - 0.8) ** 2) + np.exp(-2 * (t + 0.5) ** 2)
        h = 1.0 / N
        sqh = np.sqrt(h)
        i = np.arange(N, dtype=np.float64)
        if test == "curve1":
            return (h * sqh * (i + 0.5)).reshape(-1, 1)
        if test == "curve2":
            return ((np.exp((i + 1) * h) - np.exp(i * h)) / sqh).reshape(-1, 1)
        if test == "curve3":
            d = (((i + 1) * h) ** 2 - (i * h) ** 2) / 2
            first = np.arange(N) < int(N / 2 + 1)
            return (np.where(first, d, h - d) / sqh).reshape(-1, 1)
        raise ValueError(f"unknown test signal {test!r}")

    def gen_data(self, x, **kwargs):
        """b = blurred x (:104-143): on a zero-padded 2N grid unless CommitCrime; parameter defaults to 0.3 as in the reference."""
        from scipy.ndimage import convolve1d
        if "parameter" in kwargs:
            self.parameter, self.boundary_condition = kwargs["parameter"], "reflect"
        elif self.parameter is None:
            self.parameter = 0.3
            self.boundary_condition = kwargs.get("boundary_condition", self.boundary_condition or "reflect")
        n = self.grid_points
        self.PSF, self.center = self.Gauss1D(n, self.parameter)
        if self.CommitCrime is False:
            pad = np.zeros((2 * n, 1))
            pad[n // 2:n // 2 + n, :] = np.asarray(x, dtype=np.float64).reshape((n, 1))
            b = convolve1d(pad, self.PSF, mode=self.boundary_condition)      # (axis -1 of an (2n, 1) array, as the reference)
            return b[n // 2:n // 2 + n, :].reshape((-1, 1))
        return convolve1d(np.asarray(x, dtype=np.float64), self.PSF, mode=self.boundary_condition).reshape((-1, 1))

    def add_noise(self, b_true, opt, noise_level):
        """(b_meas, delta) — :199-216 (unseeded, like the reference)."""
        b_true = np.asarray(b_true, dtype=np.float64)
        if opt == "Gaussian":
            e = np.random.randn(self.grid_points, 1)
            sig = noise_level * np.linalg.norm(b_true) / np.linalg.norm(e)
            return b_true + sig * e, np.linalg.norm(sig * e)
        if opt == "Poisson":
            return np.random.poisson(lam=b_true + 1), 0
        if opt == "Laplace":
            e = np.random.laplace(self.grid_points)
            sig = noise_level * np.linalg.norm(b_true) / np.linalg.norm(e)
            return b_true + sig * e, np.linalg.norm(sig * e)
        raise ValueError(f"unknown noise option {opt!r}")


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

    def gen_data(self, x, nx, ny, views, engine=None):
        """(A, b, p, q, AforMatrixOperation) of Tomography.py:153-168: b from the angle-shifted operator unless CommitCrime;
        NOTE the reference then overwrites p with `views` and q with rows / views (:166-167); reproduced."""
        ops = self.forward_Op(nx, ny, views, engine=engine)
        xv = np.asarray(x, dtype=np.float64).reshape(-1)
        b = np.asarray((ops[2] if not self.CommitCrime else ops[0]) @ xv).reshape((-1, 1))
        self.p = views
        self.q = int(b.shape[0] / views)
        return ops[0], b, self.p, self.q, ops[1]

    def add_noise(self, b_true, opt, noise_level):
        """(b_meas as a p x q array, delta) — Tomography.py:203-227 (unseeded like the reference)."""
        b_true = np.asarray(b_true, dtype=np.float64)
        if opt == "Gaussian":
            noise = np.random.randn(b_true.shape[0]).reshape((-1, 1))
            e = noise_level * np.linalg.norm(b_true) / np.linalg.norm(noise) * noise
            b_meas, delta = b_true.reshape((-1, 1)) + e, np.linalg.norm(e)
        elif opt == "Poisson":
            b_meas, delta = np.random.poisson(lam=b_true + 1), 0
        else:
            e = np.random.laplace(self.p * self.q)
            sig = noise_level * np.linalg.norm(b_true) / np.linalg.norm(e)
            b_meas, delta = b_true + sig * e, np.linalg.norm(sig * e)
        return np.asarray(b_meas).reshape((self.p, self.q)), delta


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
