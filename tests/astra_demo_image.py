"""Comparing a fan-beam operator with the ASTRA outputs the reference holds as rendered images
(tests/golden/fanbeam_demo_image.npz, made by tools/make_fanbeam_demo_golden.py from
/root/reference/demos/demo_Tomo_small_scale.ipynb:145,179).  Test infrastructure.

An image fixes its data up to the affine grey map of `imshow` (vmin / vmax = data min / max) and 8-bit rounding; the sinogram
additionally carries the demo's 1 % Gaussian noise (unseeded) and was made with the angles shifted by 1e-8 (Tomography.py:61).
So the comparison is a correlation plus the residual of the best affine map, in grey levels."""
import numpy as np


def corr(a, b):
    a = np.asarray(a, dtype=np.float64).ravel()
    b = np.asarray(b, dtype=np.float64).ravel()
    a, b = a - a.mean(), b - b.mean()
    return float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b)))


def affine_residual(data, grey):
    """Residual (grey levels) of the least-squares map grey ~ c0 * data + c1, and (c0, c1)."""
    d = np.asarray(data, dtype=np.float64).ravel()
    A = np.stack([d, np.ones_like(d)], axis=1)
    c, *_ = np.linalg.lstsq(A, np.asarray(grey, dtype=np.float64).ravel(), rcond=None)
    return (np.asarray(grey, dtype=np.float64).ravel() - A @ c).reshape(np.shape(grey)), c


def box_raster(M, H, W):
    """The matrix M drawn on an H x W raster: every screen pixel the mean of the entries it covers (fractional edges)."""
    M = np.asarray(M, dtype=np.float64)
    R, C = M.shape
    cs = np.zeros((R + 1, C + 1))
    cs[1:, 1:] = M.cumsum(0).cumsum(1)

    def at(edges, n):            # linear interpolation of the cumulative sum along one axis
        i = np.minimum(np.floor(edges).astype(int), n - 1)
        return i, edges - i

    re, ce = np.linspace(0, R, H + 1), np.linspace(0, C, W + 1)
    ri, rf = at(re, R)
    ci, cf = at(ce, C)
    rows = cs[ri] * (1 - rf)[:, None] + cs[ri + 1] * rf[:, None]
    I = rows[:, ci] * (1 - cf)[None, :] + rows[:, ci + 1] * cf[None, :]
    return (I[1:, 1:] - I[:-1, 1:] - I[1:, :-1] + I[:-1, :-1]) / ((R / H) * (C / W))


def check_against_demo_images(g, sino, dense=None, sino_corr=0.999, dense_corr=0.97):
    """`sino`: the operator applied to g['phantom'] (row-major), any shape of views * n_det numbers in the operator's own row
    order; `dense`: the operator as a (views * n_det) x N^2 matrix or None.  Returns the measured figures."""
    views, nd = int(g["views"]), int(g["n_det"])
    grey = g["sino_grey"]
    s = np.asarray(sino, dtype=np.float64).reshape(views, nd)          # add_noise's reshape((self.p, self.q)) after gen_data
    out = {"sino_corr": corr(s, grey)}
    res, c = affine_residual(s, grey)
    out["sino_resid_rms_grey"] = float(np.sqrt(np.mean(res ** 2)))
    out["sino_resid_max_grey"] = float(np.abs(res).max())
    # the wrong conventions must be far away, or the image pins nothing
    out["sino_corr_wrong"] = max(corr(s[:, ::-1], grey), corr(s[::-1], grey), corr(s.reshape(-1).reshape(nd, views).T, grey))
    assert out["sino_corr"] >= sino_corr, out
    assert out["sino_corr_wrong"] < 0.9, out
    # 1 % noise is 0.8 grey levels rms here, 8-bit rounding 0.29: a matched operator leaves about 0.9; a ray that counts a pixel
    # boundary twice (view 0, detector 22 runs along x = 0) alone leaves 93
    assert out["sino_resid_rms_grey"] < 1.5 and out["sino_resid_max_grey"] < 6.0, out
    if dense is not None:
        H, W = g["AA_grey"].shape
        M = np.asarray(dense, dtype=np.float64)
        assert M.shape == tuple(g["AA_shape"])
        out["dense_corr"] = corr(box_raster(M, H, W), g["AA_grey"])
        n_img = int(g["nx"])
        Mr, Mi = M.reshape(views, nd, -1), M.reshape(M.shape[0], n_img, n_img)
        wrong = [Mr[:, ::-1].reshape(M.shape), Mr[::-1].reshape(M.shape), Mr.transpose(1, 0, 2).reshape(M.shape),
                 Mi.transpose(0, 2, 1).reshape(M.shape), Mi[:, ::-1].reshape(M.shape), Mi[:, :, ::-1].reshape(M.shape)]
        out["dense_corr_wrong"] = max(corr(box_raster(w, H, W), g["AA_grey"]) for w in wrong)
        assert out["dense_corr"] >= dense_corr, out
        assert out["dense_corr_wrong"] < 0.5, out
    return out
