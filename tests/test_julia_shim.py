"""trace.jl_amd/julia/TraceHIP.jl cannot be executed here (no Julia runtime); what CAN be checked without one:
every ccall of the shim binds a prototype of include/tracehip.h with the right argument count, integer widths and pointer
element types, and the manifest tests/golden/julia_shim_calls.json (the call sequences the GPU replay test issues) is the one
the current shim source produces."""
import json
import os

import julia_replay as jr


def test_every_ccall_binds_a_header_prototype():
    calls, protos = jr.parse_ccalls(), jr.parse_header()
    assert len(calls) >= 14, sorted(calls)
    for fn, sigs in calls.items():
        assert fn in protos, f"TraceHIP.jl calls {fn}, which include/tracehip.h does not declare"
        for sig in sigs:
            assert jr.compatible(sig, protos[fn]), f"{fn}: ccall {sig} does not match the C prototype {protos[fn]}"
    # the entry points a scene script needs are all bound
    for need in ("trhip_init", "trhip_scene_new", "trhip_scene_add_material", "trhip_scene_add_triangles", "trhip_scene_add_sphere_fields", "trhip_scene_add_point_light",
                 "trhip_scene_add_spot_light_fields", "trhip_scene_commit", "trhip_scene_free", "trhip_render_path", "trhip_render_whitted", "trhip_render_sppm_ex",
                 "trhip_comm_unique_id", "trhip_comm_init", "trhip_film_reduce", "trhip_last_error"):
        assert need in calls, need


def test_structs_mirror_the_header():
    """TrhipSensor / TrhipStats field order and widths against the ctypes mirrors (which tests/test_abi.py ties to the library)."""
    import re
    import __graft_entry__ as graft
    T = graft.load_package()
    src = open(jr.SHIM, encoding="utf-8").read()

    def fields(name):
        body = re.search(r"struct " + name + r"\n(.*?)\n(?:    " + name + r"\(|end)", src, re.S).group(1)
        return [(m.group(1), m.group(2)) for m in re.finditer(r"^\s+(\w+)::([\w{},]+)", body, re.M)]
    width = {"Float32": 4, "UInt64": 8, "Float64": 8, "UInt32": 4}

    def size(t):
        m = re.match(r"NTuple\{(\d+),(\w+)\}", t)
        return int(m.group(1)) * width[m.group(2)] if m else width[t]
    import ctypes as C
    for jl_name, ct in (("TrhipSensor", T.Sensor), ("TrhipStats", T.Stats)):
        total = sum(size(t) for _, t in fields(jl_name))
        total = (total + 7) // 8 * 8 if jl_name == "TrhipStats" else total  # both languages pad the struct to its 8-byte alignment
        assert total == C.sizeof(ct), f"{jl_name}: {total} bytes in the shim, {C.sizeof(ct)} in the C struct"
    jl_stats = [n for n, _ in fields("TrhipStats")]
    c_stats = [n for n, _ in T.Stats._fields_]
    assert jl_stats[:13] == c_stats[:13] and jl_stats[-9:] == c_stats[-9:]


def test_manifest_matches_the_shim_source():
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "julia_shim_calls.json")
    manifest = json.load(open(path))
    calls = jr.parse_ccalls()
    assert manifest["ccalls"] == {fn: [[ret, args] for ret, args in sigs] for fn, sigs in sorted(calls.items())}, \
        "TraceHIP.jl changed: regenerate with python tests/golden/make_julia_shim_manifest.py"


def test_julia_cpu_bench_scene_equals_the_python_shadows_scene():
    """bench/trace_jl_cpu.jl re-types the shadows scene (docs/src/shadows.md:8-94) for the real Trace.jl; trace.jl_amd/scenes.py holds the same scene for the
    GPU path and the oracle.  The two must not drift (round 3: the Julia script had white floor triangles where the docs — and Python — have mirrors): spheres,
    triangle mesh, per-primitive materials and the light are read out of the Julia source and compared with the Python scene object."""
    import re
    import numpy as np
    import __graft_entry__ as graft
    T = graft.load_package()
    src = open(os.path.join(os.path.dirname(jr.SHIM), "..", "..", "bench", "trace_jl_cpu.jl"), encoding="utf-8").read()
    body = src[src.index("function shadows_scene()"):src.index("function camera(")]
    f = lambda t: float(t.replace("f0", ""))
    rgb = {}
    for name, kind, first_rgb in re.findall(r"(\w+) = Trace\.(\w+Material)\(Trace\.ConstantTexture\(Trace\.RGBSpectrum\(([^)]*)\)", body):
        nums = [f(x) for x in first_rgb.split(",")]
        rgb[name] = (kind, nums if len(nums) == 3 else nums * 3)
    scene = T.scenes.shadows_scene()
    prims = scene.aggregate.primitives

    def py_material(m):
        tex = getattr(m, "Kd", None) or getattr(m, "Kr", None)
        v = tex.value
        c = [float(x) for x in (v.c if hasattr(v, "c") else v)]
        return type(m).__name__, c
    jl_spheres = re.findall(r"sphere\(\(([^)]*)\), ([\d.]+)f0, (\w+)\)", body)
    assert len(jl_spheres) == 4
    for (pos, radius, mat), prim in zip(jl_spheres, prims[:4]):
        centre = np.asarray(prim.shape.core.object_to_world.point([0, 0, 0]), np.float32)
        assert np.array_equal(centre, np.float32([f(x) for x in pos.split(",")])) and np.float32(prim.shape.radius) == np.float32(f(radius))
        kind, c = py_material(prim.material)
        assert kind == rgb[mat][0] and np.array_equal(np.float32(c), np.float32(rgb[mat][1])), (mat, kind, c, rgb[mat])
    idx = [int(x) for x in re.search(r"UInt32\[([^\]]*)\]", body).group(1).split(",")]
    verts = [[f(x) for x in v.split(",")] for v in re.findall(r"Point3f\(([^)]*)\)", body)]
    offset = [f(x) for x in re.search(r"Trace\.translate\(Vec3f\(([^)]*)\)\), false\), 4,", body).group(1).split(",")]
    tri_mats = [x.strip() for x in re.search(r"tri_materials = \[([^\]]*)\]", body).group(1).split(",")]
    assert len(tri_mats) == 4 and len(idx) == 12
    for k, prim in enumerate(prims[4:8]):
        want = np.float32([verts[i - 1] for i in idx[3 * k:3 * k + 3]]) + np.float32(offset)
        tri = prim.shape
        got = tri.mesh.vertices[tri.mesh.indices[3 * tri.k:3 * tri.k + 3].astype(np.int64) - 1]
        assert np.array_equal(got, want), (k, got, want)
        kind, c = py_material(prim.material)
        assert kind == rgb[tri_mats[k]][0] and np.array_equal(np.float32(c), np.float32(rgb[tri_mats[k]][1])), (k, tri_mats[k], kind)
    lt = re.search(r"PointLight\(Trace\.translate\(Vec3f\(([^)]*)\)\), Trace\.RGBSpectrum\(([\d.]+)f0\)\)", body)
    light = scene.lights[0]
    assert np.array_equal(np.asarray(light.light_to_world.point([0, 0, 0]), np.float32), np.float32([f(x) for x in lt.group(1).split(",")]))
    assert np.array_equal(np.float32([float(x) for x in (light.i.c if hasattr(light.i, "c") else light.i)]), np.float32([f(lt.group(2))] * 3))
