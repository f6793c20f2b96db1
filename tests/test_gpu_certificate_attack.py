"""Run with -m gpu.  The hybrid mode's certificate (csrc/th_trace3c.h) under attack — tests/attack_scenes.py says where and why.

Per family: the scene is committed with default options (two trees), and
  * every attack ray is traced by the certified walk (+ fallback) and, with option "hybrid" = 0, by k_trace3 on the canonical tree alone: primitive, t and both
    barycentrics must be the same BITS, occlusion the same booleans;
  * a subsample goes through the oracle walking the reference tree it built itself (hits and occlusion, bit for bit);
  * a small PathIntegrator frame from a camera inside the box: per-sample radiance and film equal with hybrid on / off (every bounce and shadow ray of the frame
    starts on the attacked geometry).
The share of rays the certificate hands to the reference-order walk is printed per family and ray kind (tools/soak_attack.py writes the long form to profiles/); a family
whose kernel-level rays fall back by more than 20 % is a performance cliff, not an error — DESIGN.md names them.
"""
import numpy as np
import pytest

import attack_scenes as A

pytestmark = pytest.mark.gpu


def u32(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.fixture
def hyb_ctx(ctx):
    def reset():
        ctx.set_option("bvh_builder", -1)
        ctx.set_option("hybrid", 1)
        ctx.set_option("traversal", 3)
        ctx.set_option("count_visits", 0)
    reset()
    yield ctx
    reset()


@pytest.mark.parametrize("family", [f[0] for f in A.FAMILIES])
def test_certified_walk_equals_the_reference_order_walk(family, T, ob, hyb_ctx):
    ctx = hyb_ctx
    scene, tri, lo, hi, spheres = A.make_family(T, family, seed=11)
    flat = scene.flatten(ctx)
    mode = flat.bvh_mode()[0]
    assert mode == 2, f"{family}: the default commit holds one tree only (mode {mode}: {flat.bvh_note()}) — nothing is under test"
    rng = np.random.default_rng(5)
    parts = A.attack_rays(rng, 120_000, lo, hi, tri, spheres)
    shares = {}
    for kind, rays in parts.items():
        got = flat.trace_closest(rays)
        n_rays, n_fb = flat.last_fallback()
        shares[kind] = n_fb / max(1, n_rays)
        occ = flat.trace_any(rays)
        ctx.set_option("hybrid", 0)
        ref = flat.trace_closest(rays)
        occ_ref = flat.trace_any(rays)
        ctx.set_option("hybrid", 1)
        for k in ("prim", "t", "b1", "b2"):
            bad = u32(got[k]) != u32(ref[k])
            assert not bad.any(), f"{family} / {kind}: {k} differs in {int(bad.sum())} of {rays.shape[0]} rays, first ray {rays[np.argmax(bad)].tolist()}: got {got[np.argmax(bad)]}, canonical walk {ref[np.argmax(bad)]}"
        assert np.array_equal(occ, occ_ref), f"{family} / {kind}: occlusion differs in {int((occ != occ_ref).sum())} rays"
    print(f"\n[certificate attack] {family}: fallback share per ray kind: " + ", ".join(f"{k} {v:.4f}" for k, v in shares.items()))
    # the oracle, on the reference tree it builds itself, for a subsample of every kind
    osc = ob.OracleScene.from_scene(scene)
    sub = np.concatenate([r[rng.choice(r.shape[0], min(600, r.shape[0]), replace=False)] for r in parts.values()])
    t_ref, prim_ref, _, _ = osc.trace_closest(sub)
    occ_o, _ = osc.trace_any(sub)
    h = flat.trace_closest(sub)
    assert np.array_equal(h["prim"], prim_ref), f"{family}: primitives differ from the oracle in {int((h['prim'] != prim_ref).sum())} rays"
    assert np.array_equal(u32(h["t"]), u32(t_ref)), f"{family}: t differs from the oracle"
    assert np.array_equal(flat.trace_any(sub), occ_o), f"{family}: occlusion differs from the oracle"
    # a frame from inside the box
    cam = A.attack_camera(T, lo, hi, 40)
    integ = T.PathIntegrator(cam, T.SeededSampler(4, seed=21), 8)
    film = integ.render(scene, ctx).copy()
    L = integ.sample_radiance(scene).copy()
    st = integ.stats
    assert int(st.traversal) == 9, f"{family}: the frame did not run the hybrid walk"
    # the cliff is SAID: a frame that hands more than a fifth of its rays back leaves a note for the caller (trhip_accelerator_note), an ordinary one leaves none
    share = st.fallback_rays / max(1, st.closest_rays)
    note = flat.accelerator_note()
    assert ("reference-order walk" in note) == (share > 0.2), (family, share, note)
    ctx.set_option("hybrid", 0)
    integ0 = T.PathIntegrator(cam, T.SeededSampler(4, seed=21), 8)
    film0 = integ0.render(scene, ctx)
    L0 = integ0.sample_radiance(scene)
    ctx.set_option("hybrid", 1)
    same = (u32(L) == u32(L0)) | (np.isnan(L) & np.isnan(L0))
    assert same.all(), f"{family}: per-sample radiance differs in {int((~same).sum())} values with hybrid on / off"
    samef = (u32(film) == u32(film0)) | (np.isnan(film) & np.isnan(film0))
    assert samef.all(), f"{family}: film differs"
    print(f"[certificate attack] {family}: frame fallback share {st.fallback_rays / max(1, st.closest_rays):.4f} of {st.closest_rays} closest-hit rays")
    scene._flat = None
    flat.free()
