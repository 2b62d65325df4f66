"""Demo conveniences — OUT OF THE HOT-PATH SCOPE, and since round 5 OUT OF THE PACKAGE (tools/).

Host-side, one-time data preparation of the reference's demo classes (test signals, blurred data on a padded canvas, unseeded
noise: trips/test_problems/Deblurring1D.py:104-216, Deblurring2D.py:123-159, Tomography.py:153-227), restated so that a notebook
keeps running when `trips_py_amd.problems.{Deblurring1D, Deblurring2D, Tomography}` replace the reference's classes.  Not GPU
code, not benchmarked, not part of the parity claims of the path; the engine, the solvers and bench.py never import anything from
here (they use the seeded `problems.synthetic_image` / `problems.add_noise`).  Mixins: `demo_classes()` below combines them with the
operator-constructor classes of trips_py_amd.problems for a notebook that wants the reference's whole class surface:

    sys.path.insert(0, "<repo>/tools"); from demo_helpers import demo_classes
    Deblurring1D, Deblurring2D, Tomography = demo_classes()
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _gauss_psf(dim, spread):
    from trips_py_amd.problems import gauss_psf
    return gauss_psf(dim, spread)


class Deblurring2DData:
    # Same arithmetic as Deblurring2D.py:123-159.  Images come from the caller (the reference's gen_true reads
    # ./data/image_data/*.mat: dataset handling is out of scope altogether).
    def gen_data(self, x):
        """b = blurred x.  CommitCrime=False (:125-137): blur on a zero-padded 2nx x 2ny canvas with 'constant' boundary and cut
        the centre out (the data then do not come from the reflective operator); True: the operator's own convolution."""
        from scipy.ndimage import convolve
        psf, _ = _gauss_psf(self.dim, self.spread)
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


class Deblurring1DData:
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


class TomographyData:
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


def demo_classes():
    """(Deblurring1D, Deblurring2D, Tomography): trips_py_amd.problems' operator constructors + the demo data helpers above."""
    from trips_py_amd import problems as P

    class Deblurring1D(P.Deblurring1D, Deblurring1DData):
        pass

    class Deblurring2D(P.Deblurring2D, Deblurring2DData):
        pass

    class Tomography(P.Tomography, TomographyData):
        pass

    return Deblurring1D, Deblurring2D, Tomography
