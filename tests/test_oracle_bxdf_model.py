"""The oracle's scattering code (oracle/orc_scatter.h: what the HIP kernels are held to bit for bit) against an INDEPENDENT Float64 model
written from the Julia text (tests/bxdf_model.py) — VERDICT r4 weak #6 / next #5.  The reference's own tests pin the BxDFs at normal
incidence with u = (0, 0) only (test/test_materials.jl:27-68); OrenNayar, the microfacet lobes off-normal, `sample_wh`, the Glass and
Plastic lobe assembly and the BSDF's component choice were pinned by reading alone.

Per material x `allow_multiple_lobes` x flag set: 10 000 random frames (geometric normal, shading normal tilted off it, tangent) with random
(wo, wi) for f / pdf and random (wo, u) for sample_f.  The oracle computes in Float32, the model in Float64: values must agree within
Float32 rounding of the formulas (a relative tolerance that scales with the condition of the expression, stated per check), the sampled
lobe TYPE must be the same, and "zero / not zero" must be the same — except where the model reports that the sample sits within 2e-5 of a
decision threshold (there a Float32 and a Float64 evaluation may legitimately take different branches); those must stay a small fraction.
"""
import math
import zlib

import numpy as np
import pytest

import bxdf_model as M

N = 10_000
f32 = np.float32


def unit(v):
    return v / np.linalg.norm(v, axis=-1, keepdims=True)


def frames(rng, n):
    """ng, ns (tilted up to ~25 degrees off ng, as interpolated vertex normals are), dpdu (any vector not parallel to ns: the BSDF normalises it, bsdf.jl:44-45).
    Returned as Float32 values (what both sides receive)."""
    ng = unit(rng.normal(size=(n, 3)))
    ns = unit(ng + 0.45 * rng.uniform(-1, 1, size=(n, 3)))
    t = rng.normal(size=(n, 3))
    t = t - ns * np.sum(t * ns, axis=1, keepdims=True)  # orthogonal to ns, arbitrary length: ∂p∂u of a triangle is (shading.∂p∂u = the geometric one projected)
    t = t * rng.uniform(0.2, 3.0, size=(n, 1))
    return np.concatenate([ng, ns, t], axis=1).astype(f32)


def dirs(rng, n):
    return unit(rng.normal(size=(n, 3))).astype(f32)


MATERIALS = {
    # name: (oracle kind, oracle params, model constructor taking (frame, multi))
    "matte": (0, [0.2, 0.5, 0.7, 0.0], lambda fr, mu: M.matte(fr, (0.2, 0.5, 0.7), 0.0, mu)),
    "matte_oren_nayar_20": (0, [0.6, 0.3, 0.1, 20.0], lambda fr, mu: M.matte(fr, (0.6, 0.3, 0.1), 20.0, mu)),
    "matte_oren_nayar_75": (0, [0.9, 0.9, 0.4, 75.0], lambda fr, mu: M.matte(fr, (0.9, 0.9, 0.4), 75.0, mu)),
    "mirror": (1, [0.9, 0.8, 0.7], lambda fr, mu: M.mirror(fr, (0.9, 0.8, 0.7), mu)),
    "glass_specular": (2, [1.0, 0.9, 0.8, 0.7, 0.9, 1.0, 0.0, 0.0, 1.5, 1.0], lambda fr, mu: M.glass(fr, (1.0, 0.9, 0.8), (0.7, 0.9, 1.0), 0.0, 0.0, 1.5, True, mu)),
    "glass_rough_remapped": (2, [1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 0.3, 0.3, 1.5, 1.0], lambda fr, mu: M.glass(fr, (1.0, 1.0, 1.0), (1.0, 1.0, 1.0), 0.3, 0.3, 1.5, True, mu)),
    "glass_rough_aniso_raw": (2, [0.8, 0.8, 0.8, 0.6, 0.7, 0.9, 0.15, 0.4, 1.33, 0.0], lambda fr, mu: M.glass(fr, (0.8, 0.8, 0.8), (0.6, 0.7, 0.9), 0.15, 0.4, 1.33, False, mu)),
    "glass_reflect_only_rough": (2, [0.8, 0.8, 0.8, 0.0, 0.0, 0.0, 0.2, 0.2, 1.5, 0.0], lambda fr, mu: M.glass(fr, (0.8, 0.8, 0.8), (0.0, 0.0, 0.0), 0.2, 0.2, 1.5, False, mu)),
    "plastic_remapped": (3, [0.3, 0.5, 0.2, 0.6, 0.6, 0.6, 0.2, 1.0], lambda fr, mu: M.plastic(fr, (0.3, 0.5, 0.2), (0.6, 0.6, 0.6), 0.2, True, mu)),
    "plastic_raw_smooth": (3, [0.5, 0.1, 0.1, 0.9, 0.9, 0.9, 0.05, 0.0], lambda fr, mu: M.plastic(fr, (0.5, 0.1, 0.1), (0.9, 0.9, 0.9), 0.05, False, mu)),
    "plastic_specular_only": (3, [0.0, 0.0, 0.0, 0.7, 0.7, 0.7, 0.3, 1.0], lambda fr, mu: M.plastic(fr, (0.0, 0.0, 0.0), (0.7, 0.7, 0.7), 0.3, True, mu)),
}
FLAG_SETS = [M.BSDF_ALL, M.BSDF_ALL & ~M.BSDF_SPECULAR, M.BSDF_REFLECTION | M.BSDF_DIFFUSE | M.BSDF_GLOSSY, M.BSDF_TRANSMISSION | M.BSDF_GLOSSY | M.BSDF_SPECULAR]


def params64(params):
    """The oracle stores its parameters as Float32: the model gets the same numbers."""
    return [float(f32(p)) for p in params]


def build_model(name, frame_row, multi):
    kind, params, _ = MATERIALS[name]
    p = params64(params)
    fr = (tuple(float(x) for x in frame_row[0:3]), tuple(float(x) for x in frame_row[3:6]), tuple(float(x) for x in frame_row[6:9]))
    if kind == 0:
        return M.matte(fr, tuple(p[0:3]), p[3], multi)
    if kind == 1:
        return M.mirror(fr, tuple(p[0:3]), multi)
    if kind == 2:
        return M.glass(fr, tuple(p[0:3]), tuple(p[3:6]), p[6], p[7], p[8], p[9] != 0.0, multi)
    return M.plastic(fr, tuple(p[0:3]), tuple(p[3:6]), p[6], p[7] != 0.0, multi)


def close(a, b, rtol, atol):
    return abs(a - b) <= atol + rtol * max(abs(a), abs(b))


ULP = 2.0 ** -23
_PERT = [(1 + ULP, 1 - ULP, 1 + ULP), (1 - ULP, 1 + ULP, 1 + ULP), (1 + ULP, 1 + ULP, 1 - ULP), (1 - ULP, 1 - ULP, 1 - ULP)]


def sensitivity(fn, *vecs):
    """How far the model's outputs move when its inputs move by one Float32 ulp: the condition of the expression at this sample.  The oracle rounds every
    intermediate to Float32 — dozens of perturbations of that size along the way, in places amplified again (z = sqrt(1 - x² - y²) at grazing angles) — so a value is allowed SENS_K x this beside the flat tolerance."""
    base = fn(*vecs)
    dev = [0.0] * len(base)
    for k, p in enumerate(_PERT):
        moved = [tuple(c * p[(j + k) % 3] for j, c in enumerate(v)) for v in vecs]
        out = fn(*moved)
        for j in range(len(base)):
            if math.isfinite(out[j]) and math.isfinite(base[j]):
                dev[j] = max(dev[j], abs(out[j] - base[j]))
    return dev


SENS_K = 256.0
# A wrong formula moves EVERY sample by percents; Float32 cancellation inside the slope sampler or under a root moves a few samples per thousand by more than the
# input sensitivity predicts.  So: 99.5 % of the compared numbers within the tolerance, none beyond OUTLIER times it.
OUTLIER = 64.0


# Float32 evaluation of these formulas against an exact one: a few ulps each step; where the expression is ill-conditioned (half vectors of nearly opposite
# directions, grazing angles, 1 - cos² under a root) the error is the condition number times that — measured per sample by `sensitivity`.
RTOL = {"lambert": 4e-6, "default": 2e-5}  # flat relative tolerance (a few dozen Float32 roundings); the condition of the sample comes on top (sensitivity)


def rtol_for(name):
    return RTOL["lambert"] if name == "matte" else RTOL["default"]


@pytest.mark.parametrize("name", list(MATERIALS))
@pytest.mark.parametrize("multi", [False, True])
def test_f_and_pdf_agree_with_the_float64_model(name, multi, ob):
    rng = np.random.default_rng(zlib.crc32(f"f {name} {multi}".encode()))
    kind, params, _ = MATERIALS[name]
    osc = ob.OracleScene()
    osc.add_material(kind, params)
    fr = frames(rng, N)
    wo, wi = dirs(rng, N), dirs(rng, N)
    ratios, edge, checked = [], 0, 0
    for flags in FLAG_SETS:
        got = osc.bsdf_query(0, multi, 0, flags, fr, np.concatenate([wo, wi], axis=1))
        for i in range(0, N, 4 if flags != M.BSDF_ALL else 1):  # every sample under BSDF_ALL, a quarter under the other flag sets
            b = build_model(name, fr[i], multi)
            M.E.notes.clear()
            wo_i, wi_i = tuple(float(x) for x in wo[i]), tuple(float(x) for x in wi[i])
            f = b.f(wo_i, wi_i, flags)
            p = b.pdf(wo_i, wi_i, flags)
            if M.E.notes:
                edge += 1
                continue
            notes_before = len(M.E.notes)
            dev = sensitivity(lambda a, c: (*b.f(a, c, flags), b.pdf(a, c, flags)), wo_i, wi_i)
            del M.E.notes[notes_before:]
            rt = rtol_for(name)
            want = (*f, p)
            for c in range(4):
                tol = 1e-7 * max(1.0, abs(want[c])) + rt * abs(want[c]) + SENS_K * dev[c]
                ratio = abs(float(got[i, c]) - want[c]) / tol
                assert ratio <= OUTLIER, (name, multi, flags, i, "f f f pdf"[2 * c], c, got[i], want, dev, M.E.notes)
                ratios.append(ratio)
            checked += 1
    assert edge <= 0.03 * (checked + edge), f"{edge} of {checked + edge} samples sit on decision thresholds or at ill-conditioned azimuths"
    assert np.percentile(ratios, 99.5) <= 1.0, (np.percentile(ratios, [50, 99, 99.5, 100]))


@pytest.mark.parametrize("name", list(MATERIALS))
@pytest.mark.parametrize("multi", [False, True])
def test_sample_f_agrees_with_the_float64_model(name, multi, ob):
    rng = np.random.default_rng(zlib.crc32(f"sample_f {name} {multi}".encode()))
    kind, params, _ = MATERIALS[name]
    osc = ob.OracleScene()
    osc.add_material(kind, params)
    fr = frames(rng, N)
    wo = dirs(rng, N)
    u = rng.uniform(0, 1, size=(N, 2)).astype(f32)
    u = np.minimum(u, np.nextafter(f32(1), f32(0)))
    pad = np.zeros((N, 1), f32)
    edge, checked, kinds, ratios = 0, 0, {}, []
    for flags in FLAG_SETS:
        got = osc.bsdf_query(0, multi, 1, flags, fr, np.concatenate([wo, u, pad], axis=1))
        for i in range(0, N, 4 if flags != M.BSDF_ALL else 1):
            b = build_model(name, fr[i], multi)
            M.E.notes.clear()
            wo_i = tuple(float(x) for x in wo[i])
            wi, f, pdf, st = b.sample_f(wo_i, (float(u[i, 0]), float(u[i, 1])), flags)
            if M.E.notes:
                edge += 1
                continue
            g = [float(x) for x in got[i]]
            assert int(g[7]) == st, (name, multi, flags, i, "sampled type", g, wi, f, pdf, st)
            kinds[st] = kinds.get(st, 0) + 1
            if st == M.BSDF_NONE:
                assert g[0:7] == [0.0] * 7, (name, multi, flags, i, g)
                checked += 1
                continue
            ui = (float(u[i, 0]), float(u[i, 1]), 0.0)

            def sf(a, uu):
                w_, f_, p_, t_ = b.sample_f(a, (uu[0], uu[1]), flags)
                return (*w_, *f_, p_) if t_ == st else (math.nan,) * 7  # (a perturbation that changes the sampled lobe says "edge", not "sensitive")

            notes_before = len(M.E.notes)
            dev = sensitivity(sf, wo_i, ui)
            del M.E.notes[notes_before:]
            rt = rtol_for(name)
            want = (*wi, *f, pdf)
            for c in range(7):
                flat = 4e-6 if c < 3 else (1e-6 * max(1.0, abs(want[c])) + 4 * rt * abs(want[c]))  # the direction is a unit vector: absolute
                ratio = abs(g[c] - want[c]) / (flat + SENS_K * dev[c])
                assert ratio <= OUTLIER, (name, multi, flags, i, "wi wi wi f f f pdf".split()[c], c, g, want, dev, st, b.to_local(wi), b.to_local(wo_i))
                ratios.append(ratio)
            checked += 1
    assert edge <= 0.12 * (checked + edge), f"{edge} of {checked + edge} samples sit on decision thresholds or at ill-conditioned azimuths"  # (a lobe of α = 0.05 stretches 8 % of all directions to within 0.02 of the pole)
    assert checked > 0.9 * N
    if ratios:
        assert np.percentile(ratios, 99.5) <= 1.0, (np.percentile(ratios, [50, 99, 99.5, 100]))


def test_model_reproduces_the_reference_tests_own_literals():
    """test/test_materials.jl:27-68 — the vectors the reference itself holds — through the MODEL (so that the model is anchored to the reference and not only to
    the reading): FresnelDielectric, SpecularReflection / SpecularTransmission / FresnelSpecular at normal incidence."""
    assert M.fresnel_dielectric(1.0, 1.0, 1.0) == 0.0           # "Fresnel Dielectric": vacuum-gas
    assert abs(M.fresnel_dielectric(1.0, 1.0, 1.5) - 0.04) < 1e-7    # vacuum-glass
    assert abs(M.fresnel_dielectric(1.0, 1.0, 2.0) - (1.0 / 9.0)) < 1e-7
    assert M.fresnel_dielectric(math.cos(math.radians(60.0)), 2.0, 1.0) == 1.0  # total internal reflection beyond the critical angle
    sr = M.SpecularReflection((1.0, 1.0, 1.0), M.FresnelNoOp())
    wi, pdf, f, _ = sr.sample_f((0.0, 0.0, 1.0), (0.0, 0.0))
    assert wi == (-0.0, -0.0, 1.0) and pdf == 1.0 and f == (1.0, 1.0, 1.0)
    st = M.SpecularTransmission((1.0, 1.0, 1.0), 1.0, 1.0)
    wi, pdf, f, _ = st.sample_f((0.0, 0.0, 1.0), (0.0, 0.0))
    assert tuple(abs(x) for x in wi[:2]) == (0.0, 0.0) and wi[2] == -1.0 and pdf == 1.0 and f == (1.0, 1.0, 1.0)
    fs = M.FresnelSpecular((1.0, 1.0, 1.0), (1.0, 1.0, 1.0), 1.0, 1.0)
    wi, pdf, f, t = fs.sample_f((0.0, 0.0, 1.0), (0.0, 0.0))
    assert wi[2] == -1.0 and pdf == 1.0 and f == (1.0, 1.0, 1.0) and t == (M.BSDF_SPECULAR | M.BSDF_TRANSMISSION)
