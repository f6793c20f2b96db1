"""Measure object silhouettes in the reference's only golden image (docs/src/assets/shadows-sppm-1024x1024_mio.png, an
RNG-dependent SPPM render) and store them as a small fixture.  Run in the build container (needs /root/reference):
    python tests/golden/make_golden_geometry.py
The fixture pins the scene transcription + the load-bearing camera matrices geometrically; it is data
derived from the image, not a copy of it."""
import json
import os

import numpy as np
from PIL import Image

SRC = "/root/reference/docs/src/assets/shadows-sppm-1024x1024_mio.png"


def extents(mask):
    ys, xs = np.nonzero(mask)
    return {"x_min": int(xs.min()), "x_max": int(xs.max()), "y_min": int(ys.min()), "y_max": int(ys.max()), "count": int(mask.sum())}


def masks(img):
    """img: H x W x 3 float in [0,1], image coordinates (row 0 = top)."""
    r, g, b = img[..., 0], img[..., 1], img[..., 2]
    rows, cols = np.arange(img.shape[0])[:, None], np.arange(img.shape[1])[None, :]
    blue = (b > 0.25) & (r < 0.5 * b) & (cols < 300) & (rows > 600) & (rows < 860)          # matte blue sphere, directly visible
    red = (r > 0.25) & (g < 0.45 * r) & (b < 0.45 * r) & (rows > 765) & (cols > 520)        # matte red sphere, directly visible
    return {"blue_sphere": blue, "red_sphere": red}


if __name__ == "__main__":
    img = np.asarray(Image.open(SRC).convert("RGB")).astype(np.float32) / 255
    out = {k: extents(m) for k, m in masks(img).items()}
    out["source"] = "docs/src/assets/shadows-sppm-1024x1024_mio.png (pxl-th/Trace.jl @ 2024_10_08)"
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "shadows_golden_geometry.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))
