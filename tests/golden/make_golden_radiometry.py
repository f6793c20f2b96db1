#!/usr/bin/env python
"""Radiometric fixture from the reference's only end-to-end output, docs/src/assets/shadows-sppm-1024x1024_mio.png (the scene of
docs/src/shadows.md:8-107 rendered by SPPMIntegrator, integrators/sppm.jl:132-173, and written by save(film), film.jl:204-222).
Run in the build container (needs /root/reference):

    python tests/golden/make_golden_radiometry.py

Writes tests/golden/shadows_golden_radiometry.npz: the 8-bit image reduced to 128 x 128 block means (8 x 8 pixels each, per channel, stored
as uint16 = mean x 64: exact), the fraction of pixels that are exactly black, and the per-channel means — statistics of the image, not a copy of
it (1/64 of its samples).  tests/test_gpu_golden_radiometry.py renders the same scene with the GPU SPPM integrator and compares."""
import os

import numpy as np
from PIL import Image

SRC = "/root/reference/docs/src/assets/shadows-sppm-1024x1024_mio.png"

if __name__ == "__main__":
    img = np.asarray(Image.open(SRC).convert("RGB")).astype(np.uint32)  # (1024, 1024, 3), row 0 = top of the picture
    assert img.shape == (1024, 1024, 3)
    blocks = img.reshape(128, 8, 128, 8, 3).sum(axis=(1, 3))          # sums of 64 pixels: mean x 64, at most 255 * 64 < 2^16
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "shadows_golden_radiometry.npz")
    np.savez_compressed(out, block_sum64=blocks.astype(np.uint16), black_fraction=np.float64((img.sum(-1) == 0).mean()), channel_mean=img.reshape(-1, 3).mean(0),
                        source=np.array("docs/src/assets/shadows-sppm-1024x1024_mio.png (pxl-th/Trace.jl @ 2024_10_08)"))
    print(out, os.path.getsize(out), "bytes; black fraction", (img.sum(-1) == 0).mean(), "channel means", img.reshape(-1, 3).mean(0))
