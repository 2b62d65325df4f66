"""Decode the ASTRA fan-beam OUTPUTS the reference holds as rendered images into a data fixture (build container only).

/root/reference/demos/demo_Tomo_small_scale.ipynb keeps the cell outputs of one run of the reference with ASTRA installed:
    cell `plt.imshow(AA)`   (:145)  AA = A.todense(), the 1350 x 1024 fan-beam matrix of Tomography.forward_Op(32, 32, 30)
    cell `plt.imshow(b)`    (:179)  b  = the noisy (1 %) sinogram of phantoms.tectonic(32), reshaped (30, 45) by add_noise
    cell `plt.imshow(x_true.reshape((nx, ny)))`  the phantom itself
all with `plt.set_cmap('gray')`, `plt.axis('off')`, default `imshow` (origin upper, vmin / vmax = data min / max).  They are the
only ASTRA outputs anywhere in the reference tree (ASTRA itself cannot be installed here), so they are what the fan-beam
convention — rotation sense, detector order, sinogram layout — can be pinned to.  This script reads the PNG bytes out of the
notebook's JSON, crops the axes area (the opaque pixels), and

  * for b and x_true (drawn with >= 7 screen pixels per data pixel, nearest neighbour): samples the centre of every data
    pixel -> grey level arrays (30, 45) and (32, 32), uint8;
  * for AA (6.2 matrix entries per screen pixel, matplotlib's anti-aliased down-sampling): keeps the cropped grey image as
    it is (218 x 165, uint8) — the test renders the candidate matrix to the same raster.

Output: tests/golden/fanbeam_demo_image.npz — numbers only (grey levels decoded from images + the phantom array that
`trips.utilities.phantoms.tectonic(32)` returns when the reference is imported here).
usage: python3 tools/make_fanbeam_demo_golden.py [out.npz]"""
import base64
import io
import json
import os
import sys

import numpy as np
from PIL import Image

REF = "/root/reference"
NB = os.path.join(REF, "demos", "demo_Tomo_small_scale.ipynb")
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def cell_png(nb, source_starts):
    for c in nb["cells"]:
        if c["cell_type"] == "code" and "".join(c["source"]).startswith(source_starts):
            for o in c.get("outputs", []):
                if "image/png" in o.get("data", {}):
                    return np.asarray(Image.open(io.BytesIO(base64.b64decode(o["data"]["image/png"]))).convert("RGBA"))
    raise KeyError(source_starts)


def crop_axes(rgba):
    """The axes area = the opaque pixels (figure background is transparent, `axis('off')` leaves nothing else)."""
    opaque = rgba[..., 3] == 255
    r = np.where(opaque.any(1))[0]
    c = np.where(opaque.any(0))[0]
    box = rgba[r.min():r.max() + 1, c.min():c.max() + 1]
    assert np.all(box[..., 3] == 255)
    assert np.all(box[..., 0] == box[..., 1]) and np.all(box[..., 1] == box[..., 2]), "grey colormap expected"
    return box[..., 0].copy()


def sample_centres(grey, rows, cols):
    H, W = grey.shape
    ri = np.floor((np.arange(rows) + 0.5) * H / rows).astype(int)
    ci = np.floor((np.arange(cols) + 0.5) * W / cols).astype(int)
    out = grey[np.ix_(ri, ci)]
    # every data pixel must be a flat block on screen (nearest-neighbour up-sampling): its centre 3 x 3 is constant
    for dr in (-1, 1):
        for dc in (-1, 1):
            assert np.array_equal(out, grey[np.ix_(ri + dr, ci + dc)]), "data pixels are not flat blocks"
    return out


def main(out):
    nb = json.load(open(NB))
    g_b = crop_axes(cell_png(nb, "plt.imshow(b)"))
    g_x = crop_axes(cell_png(nb, "plt.imshow(x_true.reshape((nx, ny)))"))
    g_A = crop_axes(cell_png(nb, "plt.imshow(AA)"))
    views, nx = 30, 32
    p = int(np.sqrt(2) * nx)
    sino = sample_centres(g_b, views, p)          # add_noise reshapes to (self.p, self.q) = (views, rows / views) after gen_data
    xim = sample_centres(g_x, nx, nx)
    sys.path.insert(0, REF)
    from trips.utilities import phantoms           # imports without pylops / astra
    phantom = np.asarray(phantoms.tectonic(nx), dtype=np.float64)
    np.savez_compressed(out, sino_grey=sino, xtrue_grey=xim, AA_grey=g_A, phantom=phantom,
                        views=np.int64(views), nx=np.int64(nx), n_det=np.int64(p), AA_shape=np.array([views * p, nx * nx]))
    print(f"wrote {out}: sino_grey {sino.shape}, xtrue_grey {xim.shape}, AA_grey {g_A.shape}; "
          f"phantom levels {np.unique(phantom)}, image levels {np.unique(xim)}")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(REPO, "tests", "golden", "fanbeam_demo_image.npz"))
