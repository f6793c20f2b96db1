"""The one reference-PRODUCED witness of the composite (run with -m gpu): SPPM + materials + point light + film + save.

docs/src/assets/shadows-sppm-1024x1024_mio.png is the only end-to-end output the reference holds: docs/src/shadows.md:8-107's scene rendered by
SPPMIntegrator (integrators/sppm.jl:132-173) and written by save(film) (film.jl:204-222: XYZ -> RGB, / weight, clamp, scale, vertical flip, 8 bit,
no gamma).  The reference's render is RNG-dependent (its camera pass draws from Julia's global RNG, sppm.jl:194) and its iteration count is not
recorded (the doc's code asks for 10 at 341^2; the file name says 1024^2), so the comparison is STATISTICAL: the same scene rendered here by the GPU
SPPM integrator (r0 = 0.025, depth 5 as docs/src/shadows.md:106; 100 iterations at 1024^2), converted by trhip_film_to_rgb + the 8-bit quantisation of
save(), reduced to the fixture's 128 x 128 block means (tests/golden/make_golden_radiometry.py).

Tolerances (8-bit units), and why: SPPM is consistent, not unbiased — a pixel's estimate blurs the flux over its current search radius, which shrinks
with the iteration count — so blocks crossed by a caustic or shadow EDGE differ between two iteration counts by tens of units, while flat regions agree to
the photon noise of a 64-pixel mean (~ 1 unit at 100 iterations).  Hence: the MEDIAN block difference <= 0.5, 90 % of the blocks within 2, 99 % within 8;
per-channel means of the whole picture within 0.3; fraction of exactly black pixels within 0.2 %; named flat regions (wall, direct-lit floor, sphere
interiors, the mirror sphere's black reflection, the empty right margin) within 3.  Measured when the test was written (profiles/r3/r3b_golden_radiometry.txt):
median 0.08, p90 0.61, p99 2.2, max 14.0 (a caustic edge); channel means 96.90 / 94.94 / 99.20 against the picture's 96.89 / 94.94 / 99.19; black fraction
0.33336 against 0.33356 — the bounds leave a factor ~4 for another seed / iteration count.
A wrong BSDF constant, light falloff, film weight, colour matrix, clamp or flip moves these numbers by tens of units."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "shadows_golden_radiometry.npz")

# flat regions of the picture, in 8 x 8-pixel block coordinates (row 0 = top), well inside their object's silhouette
REGIONS = {
    "back wall, upper left": (slice(4, 24), slice(4, 56)),
    "empty right margin (nothing is there: exactly black)": (slice(0, 128), slice(118, 128)),
    "mirror sphere: black reflection": (slice(52, 70), slice(68, 100)),
    "blue matte sphere, lit side": (slice(92, 100), slice(16, 26)),
    "red matte sphere": (slice(100, 110), slice(76, 86)),
    "floor in the big sphere's shadow": (slice(90, 98), slice(40, 52)),
    "back wall, middle left (soft gradient)": (slice(36, 60), slice(2, 28)),
}


def render_8bit(T, ctx, iterations=100):
    scene, cam = T.scenes.shadows_scene(), T.scenes.shadows_camera(1024)
    integ = T.SPPMIntegrator(cam, 0.025, 5, iterations, -1, seed=0x5EED0001)  # docs/src/shadows.md:106 (there: 10 iterations at 341^2)
    integ.render(scene, ctx)
    rgb = cam.film.to_rgb(ctx)  # trhip_film_to_rgb: save() up to the encoder
    img = np.clip(np.rint(rgb[::-1] * 255.0), 0, 255).astype(np.uint8)  # rows flipped, 8 bit (api.save writes exactly this array)
    return img, integ.stats


def test_gpu_sppm_render_matches_the_reference_png_statistically(T, ctx):
    fx = np.load(GOLDEN)
    ref_blocks = fx["block_sum64"].astype(np.float64) / 64.0
    img, st = render_8bit(T, ctx)
    assert img.shape == (1024, 1024, 3)
    blocks = img.astype(np.uint32).reshape(128, 8, 128, 8, 3).sum(axis=(1, 3)).astype(np.float64) / 64.0
    d = np.abs(blocks - ref_blocks).max(-1)  # per block: the worst channel
    stats = {"median": float(np.median(d)), "p90": float(np.percentile(d, 90)), "p99": float(np.percentile(d, 99)), "max": float(d.max()),
             "channel_mean": img.reshape(-1, 3).mean(0).round(3).tolist(), "ref_channel_mean": fx["channel_mean"].round(3).tolist(),
             "black_fraction": float((img.astype(np.uint32).sum(-1) == 0).mean()), "ref_black_fraction": float(fx["black_fraction"]), "ms": round(st.ms_total, 1)}
    print("golden radiometry:", stats)
    assert stats["median"] <= 0.5, stats
    assert stats["p90"] <= 2.0, stats
    assert stats["p99"] <= 8.0, stats
    assert np.all(np.abs(img.reshape(-1, 3).mean(0) - fx["channel_mean"]) <= 0.3), stats
    assert abs(stats["black_fraction"] - stats["ref_black_fraction"]) <= 0.002, stats
    for name, (rows, cols) in REGIONS.items():
        a, b = blocks[rows, cols].mean((0, 1)), ref_blocks[rows, cols].mean((0, 1))
        assert np.all(np.abs(a - b) <= 3.0), f"{name}: {a.round(2)} vs the reference's {b.round(2)}"
    # the margin right of the back wall is EXACTLY black in both (no geometry, no light: Ld = tau = 0 -> 0 after the clamp)
    assert blocks[:, 118:].max() == 0.0 and ref_blocks[:, 118:].max() == 0.0


def test_the_comparison_has_teeth(T, ctx):
    """The same statistics on deliberately wrong pictures: unflipped rows, a gamma curve, 80 % brightness — each must fail the bounds above."""
    fx = np.load(GOLDEN)
    ref_blocks = fx["block_sum64"].astype(np.float64) / 64.0
    img, _ = render_8bit(T, ctx, iterations=30)
    def med(im):
        b = im.astype(np.uint32).reshape(128, 8, 128, 8, 3).sum(axis=(1, 3)).astype(np.float64) / 64.0
        return float(np.median(np.abs(b - ref_blocks).max(-1)))
    assert med(img) <= 2.5
    assert med(img[::-1]) > 10.0                                                                  # rows not flipped
    assert med(np.clip(np.rint(255.0 * (img / 255.0) ** (1 / 2.2)), 0, 255).astype(np.uint8)) > 10.0  # a display gamma applied
    assert med(np.clip(np.rint(img * 0.8), 0, 255).astype(np.uint8)) > 4.0                        # 20 % darker
