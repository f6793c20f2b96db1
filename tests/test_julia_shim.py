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
                 "trhip_scene_add_spot_light_fields", "trhip_scene_commit", "trhip_scene_free", "trhip_render_path", "trhip_render_whitted", "trhip_render_sppm",
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
