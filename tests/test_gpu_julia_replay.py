"""The Julia shim's marshalling, replayed (-m gpu): tests/julia_replay.ShimReplay walks each scene the way TraceHIP.flatten
does — one GeometricPrimitive per Triangle as the reference holds them, nested BVHAccel primitives spliced, ONE
trhip_scene_add_triangles per run of a mesh's triangles, *_fields entry points for spheres and spot lights — and calls
libtracehip.so through ctypes signatures built from the shim's own `ccall` type tuples.  The film must equal the Python
host's (trace.jl_amd/api.py, the tested path) BIT FOR BIT, and the call sequence must be the committed manifest's."""
import json
import os

import numpy as np
import pytest

import julia_replay as jr

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def manifest():
    return json.load(open(os.path.join(GOLDEN, "julia_shim_calls.json")))


def nested_cornell(T):
    prims = T.scenes.cornell_primitives()[0]
    return T.Scene(T.scenes.cornell_lights(), T.BVHAccel([T.BVHAccel(prims[:6], 1)] + prims[6:], 1))


@pytest.mark.parametrize("name", ["shadows", "caustic_glass_ply", "nested_bvh_cornell", "tangent_uv_mesh"])
def test_replayed_shim_equals_python_host(T, ctx, manifest, name):
    if name == "shadows":
        scene, cam, entry, host = T.scenes.shadows_scene(), T.scenes.shadows_camera(64), "trhip_render_whitted", T.WhittedIntegrator
    elif name == "caustic_glass_ply":  # the reference's 88 064-triangle mesh: the case the old per-triangle marshalling could not carry
        scene, cam, entry, host = T.scenes.caustic_scene(os.path.join(GOLDEN, "caustic-glass.ply")), T.scenes.caustic_camera(64), "trhip_render_path", T.PathIntegrator
    elif name == "tangent_uv_mesh":  # a mesh with tangents and (u, v)s: trhip_scene_add_triangles_ex
        scene, cam, entry, host = jr.tangent_uv_scene(T), T.scenes.cornell_camera(64), "trhip_render_path", T.PathIntegrator
    else:  # a BVHAccel as a primitive of another (test/test_intersection.jl:137-138)
        scene, cam, entry, host = nested_cornell(T), T.scenes.cornell_camera(64), "trhip_render_path", T.PathIntegrator
    r = jr.ShimReplay(T, T._ffi.LIB_PATH, ctx._h)
    got, st = r.render(entry, scene, cam, 4, 5, seed=0x5EED0001, offset=0)
    seq = r.summary()
    flat_end = next(i for i, c in enumerate(seq) if c[0] in ("trhip_scene_commit", "trhip_scene_set_bvh")) + 1
    # a scene without nested BVHAccel primitives goes over with Trace.jl's own tree (EXACT_TREE): its last flattening call is trhip_scene_set_bvh
    assert seq[flat_end - 1][0] == ("trhip_scene_commit" if name == "nested_bvh_cornell" else "trhip_scene_set_bvh")
    assert seq[:flat_end] == manifest["sequences"][name], "the call sequence differs from tests/golden/julia_shim_calls.json"
    assert [c[0] for c in seq[flat_end:]] == [entry, "trhip_scene_free"]
    if name == "tangent_uv_mesh":
        assert sum(c[0] == "trhip_scene_add_triangles_ex" for c in seq) >= 1  # one per run of the mesh's triangles in the order of Trace.jl's tree
    if name == "caustic_glass_ply":
        tri_calls = [c for c in seq if c[0] == "trhip_scene_add_triangles"]
        assert any(c[1] == 1 and "float32[132102]" in c[2] and "uint32[264192]" in c[2] for c in tri_calls)  # the whole mesh in ONE call
    # the Python host on the SAME tree: option "bvh_builder" = 2 builds the reference's topology inside the library (th_bvh_ref.h)
    ctx.set_option("bvh_builder", -1 if name == "nested_bvh_cornell" else 2)
    try:
        scene._flat = None
        ref = host(cam, T.SeededSampler(4, seed=0x5EED0001), 5).render(scene, ctx)
    finally:
        ctx.set_option("bvh_builder", -1)
        scene._flat = None
    assert ref[..., :3].max() > 0 and st.camera_samples > 0
    assert np.array_equal(bits(got), bits(ref)), f"{int((bits(got) != bits(ref)).sum())} film values differ between the replayed shim and the Python host"


def test_replayed_sppm(T, ctx):
    scene, cam = T.scenes.cornell_scene(), T.scenes.cornell_camera(40)
    integ = T.SPPMIntegrator(cam, 0.08, 5, 2, 20000, seed=11)
    r = jr.ShimReplay(T, T._ffi.LIB_PATH, ctx._h)
    got, _ = r.render_sppm(scene, integ)
    ctx.set_option("bvh_builder", 2)  # the shim hands Trace.jl's own tree over (EXACT_TREE): the host side must walk the same one
    try:
        scene._flat = None
        ref = integ.render(scene, ctx)
    finally:
        ctx.set_option("bvh_builder", -1)
        scene._flat = None
    np.testing.assert_allclose(got, ref, rtol=1e-4, atol=1e-4 * np.abs(ref).max())  # Float32 flux sums are not ordered from run to run
    assert np.all(got[..., 3] == 1.0)
