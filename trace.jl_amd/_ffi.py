"""ctypes binding of libtracehip.so (include/tracehip.h).

The library is built in-tree by ``__graft_entry__.build()`` (hipcc, gfx950).  There is no CPU fallback: if the shared
object is missing or no MI355X is visible, the entry points raise ``TraceHipError`` — they never route through oracle/.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TRHIP_LIB", os.path.join(_HERE, "libtracehip.so"))  # TRHIP_LIB: A/B builds for tuning (tools/)


class TraceHipError(RuntimeError):
    pass


class Sensor(C.Structure):
    """trhip_sensor"""
    _fields_ = [
        ("raster_to_camera", C.c_float * 16),
        ("camera_to_world", C.c_float * 16),
        ("lens_radius", C.c_float),
        ("focal_distance", C.c_float),
        ("shutter_open", C.c_float),
        ("shutter_close", C.c_float),
        ("crop_min", C.c_float * 2),
        ("crop_max", C.c_float * 2),
        ("filter_radius", C.c_float * 2),
        ("filter_table", C.c_float * 256),
        ("scale", C.c_float),
    ]


class Stats(C.Structure):
    """trhip_stats"""
    _fields_ = [
        ("camera_samples", C.c_uint64),
        ("closest_rays", C.c_uint64),
        ("shadow_rays", C.c_uint64),
        ("nodes_visited", C.c_uint64),
        ("prims_tested", C.c_uint64),
        ("nodes_visited_shadow", C.c_uint64),
        ("prims_tested_shadow", C.c_uint64),
        ("ms_total", C.c_double),
        ("ms_raygen", C.c_double),
        ("ms_trace_closest", C.c_double),
        ("ms_shade", C.c_double),
        ("ms_trace_any", C.c_double),
        ("ms_film", C.c_double),
        ("launches_raygen", C.c_uint32),
        ("launches_trace_closest", C.c_uint32),
        ("launches_shade", C.c_uint32),
        ("launches_trace_any", C.c_uint32),
        ("launches_film", C.c_uint32),
        ("n_batches", C.c_uint32),
        ("max_depth_reached", C.c_uint32),
        ("traversal", C.c_uint32),
        ("node_bytes", C.c_uint32),
        ("replicated_rays", C.c_uint64),
        ("fallback_rays", C.c_uint64),
        ("ms_sub", C.c_double * 4),
        ("launches_sub", C.c_uint32 * 4),
        ("count_sub", C.c_uint64 * 4),
        ("ms_fallback", C.c_double),
        ("launches_fallback", C.c_uint32),
        ("reserved0", C.c_uint32),
        ("nodes_visited_fallback", C.c_uint64),
        ("prims_tested_fallback", C.c_uint64),
    ]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


UNIQUE_ID_BYTES = 128  # TRHIP_UNIQUE_ID_BYTES


def comm_unique_id() -> bytes:
    """ncclGetUniqueId through the library: rank 0 calls this and hands the 128 bytes to the other processes."""
    buf = (C.c_uint8 * UNIQUE_ID_BYTES)()
    rc = lib().trhip_comm_unique_id(buf)
    if rc:
        raise TraceHipError(f"trhip_comm_unique_id failed ({rc}): {lib().trhip_last_error(None).decode()}")
    return bytes(buf)


HIT_DTYPE = np.dtype([("t", np.float32), ("prim", np.int32), ("b1", np.float32), ("b2", np.float32)])

_F = C.POINTER(C.c_float)
_U32 = C.POINTER(C.c_uint32)
_VP = C.c_void_p
SPPM_WRITE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint32, C.POINTER(C.c_float))  # trhip_sppm_write_fn

# name -> (restype, argtypes); every symbol include/tracehip.h declares
SIGNATURES = {
    "trhip_version": (C.c_int, []),
    "trhip_init": (C.c_int, [C.POINTER(_VP), C.c_int]),
    "trhip_shutdown": (None, [_VP]),
    "trhip_last_error": (C.c_char_p, [_VP]),
    "trhip_scene_new": (C.c_int, [_VP, C.POINTER(_VP)]),
    "trhip_scene_free": (None, [_VP]),
    "trhip_scene_add_material": (C.c_int, [_VP, C.c_int, _F, C.c_int, _U32]),
    "trhip_scene_add_triangles": (C.c_int, [_VP, _F, C.c_uint32, _U32, C.c_uint32, _F, _U32, C.c_int, _U32]),
    "trhip_scene_add_triangles_ex": (C.c_int, [_VP, _F, C.c_uint32, _U32, C.c_uint32, _F, _F, _F, _U32, C.c_int, _U32]),
    "trhip_scene_add_sphere": (C.c_int, [_VP, _F, _F, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_uint32, _U32]),
    "trhip_scene_add_sphere_fields": (C.c_int, [_VP, _F, _F, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_uint32, _U32]),
    "trhip_scene_add_spot_light_fields": (C.c_int, [_VP, _F, _F, _F, C.c_float, C.c_float]),
    "trhip_scene_add_point_light": (C.c_int, [_VP, _F, _F, _F]),
    "trhip_scene_add_spot_light": (C.c_int, [_VP, _F, _F, _F, C.c_float, C.c_float]),
    "trhip_scene_commit": (C.c_int, [_VP, C.c_int]),
    "trhip_build_bvh_host": (C.c_int, [C.c_int, _F, C.c_uint32, C.c_int, _F, _U32, _U32, _U32, _U32, _U32]),
    "trhip_scene_bvh_size": (C.c_int, [_VP, _U32, _U32]),
    "trhip_scene_get_bvh": (C.c_int, [_VP, _F, _U32, _U32, _U32]),
    "trhip_scene_set_bvh": (C.c_int, [_VP, _F, _U32, _U32, C.c_uint32, _U32, C.c_uint32]),
    "trhip_plan_bands": (C.c_int, [C.POINTER(Sensor), C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint32, _U32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "trhip_scene_bvh_mode": (C.c_int, [_VP, C.POINTER(C.c_int), _U32, _U32]),
    "trhip_scene_bvh_note": (C.c_int, [_VP, C.c_char_p, C.c_size_t]),
    "trhip_scene_get_accelerator": (C.c_int, [_VP, _F, _U32, _U32, _U32]),
    "trhip_render_whitted": (C.c_int, [_VP, _VP, C.POINTER(Sensor), C.c_uint32, C.c_int, C.c_uint64, C.c_uint32, _F, C.POINTER(Stats)]),
    "trhip_render_path": (C.c_int, [_VP, _VP, C.POINTER(Sensor), C.c_uint32, C.c_int, C.c_uint64, C.c_uint32, _F, C.POINTER(Stats)]),
    "trhip_render_path_device": (C.c_int, [_VP, _VP, C.POINTER(Sensor), C.c_uint32, C.c_int, C.c_uint64, C.c_uint32, _VP, C.POINTER(Stats)]),
    "trhip_render_whitted_device": (C.c_int, [_VP, _VP, C.POINTER(Sensor), C.c_uint32, C.c_int, C.c_uint64, C.c_uint32, _VP, C.POINTER(Stats)]),
    "trhip_last_sample_radiance": (C.c_int, [_VP, _F, C.c_uint64]),
    "trhip_render_sppm": (C.c_int, [_VP, _VP, C.POINTER(Sensor), C.c_float, C.c_int, C.c_uint32, C.c_int64, C.c_uint64, _F, C.POINTER(Stats)]),
    "trhip_render_sppm_ex": (C.c_int, [_VP, _VP, C.POINTER(Sensor), C.c_float, C.c_int, C.c_uint32, C.c_int64, C.c_uint64, _F, C.POINTER(Stats), C.c_uint32, SPPM_WRITE_FN, _VP]),
    "trhip_sppm_state": (C.c_int, [_VP, _F, _F, _F, C.POINTER(C.c_double), C.POINTER(C.c_int64), _F, _F, _F, C.POINTER(C.c_int64)]),
    "trhip_film_to_rgb": (C.c_int, [_VP, _F, C.c_uint32, C.c_uint32, C.c_float, _F]),
    "trhip_trace_closest": (C.c_int, [_VP, _VP, _F, C.c_uint64, _VP]),
    "trhip_trace_any": (C.c_int, [_VP, _VP, _F, C.c_uint64, C.POINTER(C.c_uint8)]),
    "trhip_trace_closest_device": (C.c_int, [_VP, _VP, _VP, C.c_uint64, _VP, C.c_int, C.POINTER(C.c_double)]),
    "trhip_trace_any_device": (C.c_int, [_VP, _VP, _VP, C.c_uint64, _VP, C.c_int, C.POINTER(C.c_double)]),
    "trhip_accelerator_note": (C.c_int, [_VP, _VP, C.c_char_p, C.c_size_t]),
    "trhip_last_visit_counts": (C.c_int, [_VP, C.POINTER(C.c_uint64)]),
    "trhip_last_fallback_counts": (C.c_int, [_VP, C.POINTER(C.c_uint64)]),
    "trhip_last_bvh_build_ms": (C.c_int, [_VP, C.POINTER(C.c_double)]),
    "trhip_hit_geometry": (C.c_int, [_VP, _VP, _F, C.c_uint64, _F]),
    "trhip_generate_rays": (C.c_int, [_VP, C.POINTER(Sensor), _F, C.c_uint64, _F]),
    "trhip_bsdf_query": (C.c_int, [_VP, _VP, C.c_uint32, C.c_int, C.c_int, C.c_int, _F, _F, C.c_uint64, _F]),
    "trhip_film_accumulate": (C.c_int, [_VP, C.POINTER(Sensor), C.c_uint32, C.c_uint64, C.c_uint32, _F, _F]),
    "trhip_set_option": (C.c_int, [_VP, C.c_char_p, C.c_int64]),
    "trhip_option_in_build": (C.c_int, [C.c_char_p, C.c_int64]),
    "trhip_closest_kernel_name": (C.c_int, [_VP, _VP, C.c_char_p, C.c_size_t]),
    "trhip_comm_unique_id": (C.c_int, [C.POINTER(C.c_uint8)]),
    "trhip_comm_init": (C.c_int, [_VP, C.POINTER(C.c_uint8), C.c_int, C.c_int]),
    "trhip_comm_destroy": (C.c_int, [_VP]),
    "trhip_comm_rank": (C.c_int, [_VP, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "trhip_film_reduce": (C.c_int, [_VP, _VP, C.c_uint64, C.c_int]),
    "trhip_film_allreduce": (C.c_int, [_VP, _VP, C.c_uint64]),
    "trhip_detmath_f32": (C.c_int, [C.c_int, _F, _F, C.c_uint64, _F]),
    "trhip_detmath_f32_device": (C.c_int, [_VP, C.c_int, _F, _F, C.c_uint64, _F]),
}

_lib = None


def lib():
    """Load libtracehip.so (once) and attach the prototypes.  Loading needs no GPU; computing does."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise TraceHipError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                                "(hipcc --offload-arch=gfx950); there is no CPU fallback")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def fptr(a):
    return a.ctypes.data_as(_F)


def u32ptr(a):
    return a.ctypes.data_as(_U32)


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def build_bvh_host(prim_bounds, max_node_primitives: int = 1, builder: int = 2):
    """BVHAccel construction alone, on the host (trhip_build_bvh_host; no GPU needed): builder 2 = the reference's own construction node for
    node (accel/bvh.jl:87-206), 0 = the library's binned SAH.  prim_bounds: (n, 6) world bounds.  Returns (bounds (m, 6), a, flags, order, max_depth)
    in the layout of FlatScene.bvh() / set_bvh()."""
    import numpy as np
    pb = f32(prim_bounds).reshape(-1, 6)
    n_nodes, depth = C.c_uint32(0), C.c_uint32(0)

    def check(rc):
        if rc:
            raise TraceHipError(f"trhip_build_bvh_host failed ({rc}): {lib().trhip_last_error(None).decode()}")
    check(lib().trhip_build_bvh_host(builder, fptr(pb), pb.shape[0], max_node_primitives, None, None, None, C.byref(n_nodes), None, C.byref(depth)))
    m = n_nodes.value
    bounds, a, flags, order = np.empty((m, 6), np.float32), np.empty(m, np.uint32), np.empty(m, np.uint32), np.empty(pb.shape[0], np.uint32)
    check(lib().trhip_build_bvh_host(builder, fptr(pb), pb.shape[0], max_node_primitives, fptr(bounds), u32ptr(a), u32ptr(flags), C.byref(n_nodes), u32ptr(order), C.byref(depth)))
    return bounds, a, flags, order, depth.value


def detmath(fn: int, x, y=None):
    """Deterministic Float32 elementary functions (include/trace_detmath.h), evaluated by the library on the host."""
    x = f32(np.atleast_1d(x))
    out = np.empty_like(x)
    yy = f32(np.atleast_1d(y)) if y is not None else None
    rc = lib().trhip_detmath_f32(fn, fptr(x), fptr(yy) if yy is not None else None, x.size, fptr(out))
    if rc:
        raise TraceHipError("trhip_detmath_f32 failed")
    return out


class Context:
    """One GPU (trhip_ctx)."""

    def detmath(self, fn: int, x, y=None):
        """include/trace_detmath.h evaluated by a kernel on this GPU (trhip_detmath_f32_device)."""
        x = f32(np.atleast_1d(x))
        out = np.empty_like(x)
        yy = f32(np.atleast_1d(y)) if y is not None else None
        self.check(lib().trhip_detmath_f32_device(self._h, fn, fptr(x), fptr(yy) if yy is not None else None, x.size, fptr(out)))
        return out

    def __init__(self, device: int = 0):
        self._h = _VP()
        rc = lib().trhip_init(C.byref(self._h), device)
        if rc:
            msg = lib().trhip_last_error(None).decode()
            self._h = None
            raise TraceHipError(f"trhip_init failed ({rc}): {msg}")

    def check(self, rc):
        if rc:
            raise TraceHipError(f"libtracehip error {rc}: {lib().trhip_last_error(self._h).decode()}")

    def set_option(self, name: str, value: int):
        self.check(lib().trhip_set_option(self._h, name.encode(), int(value)))

    def has_option(self, name: str, value: int) -> bool:
        """Whether this build of the library honours the option value (the kernel families that lost their measurements — traversals 4 / 6 / 7, leaf_queue, leaf_sorted,
        the linear BVH builder — exist only in the EXPERIMENTS build: __graft_entry__.build_library(extra_flags=["-DTRHIP_EXPERIMENTS"], …)).  The option is left as it was
        when the answer is no, and SET when it is yes."""
        return lib().trhip_set_option(self._h, name.encode(), int(value)) != -3

    # ---- multi-GPU job (include/tracehip.h "multi-GPU"): one process per GPU, RCCL behind the C ABI -------------------
    def comm_init(self, unique_id: bytes, rank: int, n_ranks: int):
        buf = (C.c_uint8 * UNIQUE_ID_BYTES).from_buffer_copy(bytes(unique_id))
        self.check(lib().trhip_comm_init(self._h, buf, int(rank), int(n_ranks)))

    def comm_destroy(self):
        self.check(lib().trhip_comm_destroy(self._h))

    def comm_rank(self):
        r, n = C.c_int(), C.c_int()
        self.check(lib().trhip_comm_rank(self._h, C.byref(r), C.byref(n)))
        return r.value, n.value

    def film_reduce(self, device_ptr: int, n_pixels: int, root: int = 0):
        """In-place sum of the (H, W, 4) film accumulators of all ranks onto `root` (ncclReduce inside the library)."""
        self.check(lib().trhip_film_reduce(self._h, C.c_void_p(int(device_ptr)), int(n_pixels), int(root)))

    def film_allreduce(self, device_ptr: int, n_pixels: int):
        self.check(lib().trhip_film_allreduce(self._h, C.c_void_p(int(device_ptr)), int(n_pixels)))

    def close(self):
        if self._h:
            lib().trhip_shutdown(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_ctx = None


def default_context() -> Context:
    global _default_ctx
    if _default_ctx is None:
        dev = int(os.environ.get("LOCAL_RANK", "0")) if os.environ.get("TRHIP_USE_LOCAL_RANK", "1") == "1" else 0
        _default_ctx = Context(dev)
    return _default_ctx
