"""ctypes binding of oracle/liboracle.so plus an adapter that feeds a trace_jl_amd Scene to the CPU oracle.

Test infrastructure: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(ROOT, "oracle", "liboracle.so")

_F = C.POINTER(C.c_float)
_U32 = C.POINTER(C.c_uint32)
_I32 = C.POINTER(C.c_int32)
_U64 = C.POINTER(C.c_uint64)
_VP = C.c_void_p


class OrcSensor(C.Structure):
    _fields_ = [("camera_to_world", C.c_float * 16), ("camera_to_world_inv", C.c_float * 16), ("screen_window", C.c_float * 4), ("shutter_open", C.c_float),
                ("shutter_close", C.c_float), ("lens_radius", C.c_float), ("focal_distance", C.c_float), ("fov_deg", C.c_float), ("resolution", C.c_float * 2),
                ("crop", C.c_float * 4), ("filter_radius", C.c_float * 2), ("filter_tau", C.c_float), ("film_scale", C.c_float)]


class OrcStats(C.Structure):
    _fields_ = [("camera_samples", C.c_uint64), ("closest_rays", C.c_uint64), ("shadow_rays", C.c_uint64), ("nodes_visited", C.c_uint64), ("prims_tested", C.c_uint64)]


_lib = None


# exchange(user, phi3, M, n_pixels): sum the per-pixel ϕ / M of a sharded SPPM photon pass over the processes, in place
EXCHANGE_FN = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int64), C.c_uint64)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            import subprocess
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s"])
        l = C.CDLL(LIB_PATH)
        sig = {
            "orc_last_error": (C.c_char_p, []),
            "orc_scene_new": (_VP, []),
            "orc_scene_free": (None, [_VP]),
            "orc_scene_add_material": (C.c_int, [_VP, C.c_int, _F, C.c_int]),
            "orc_scene_add_triangle_mesh": (C.c_int, [_VP, _F, _F, C.c_int, _F, C.c_uint32, _U32, C.c_uint32, _F, _I32]),
            "orc_scene_add_triangle_mesh_ex": (C.c_int, [_VP, _F, _F, C.c_int, _F, C.c_uint32, _U32, C.c_uint32, _F, _F, _F, _I32]),
            "orc_scene_add_sphere": (C.c_int, [_VP, _F, _F, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int]),
            "orc_scene_add_point_light": (C.c_int, [_VP, _F, _F, _F]),
            "orc_scene_add_spot_light": (C.c_int, [_VP, _F, _F, _F, C.c_float, C.c_float]),
            "orc_scene_get_light": (C.c_int, [_VP, C.c_int, _F, _F]),
            "orc_scene_commit_reference_bvh": (C.c_int, [_VP, C.c_int]),
            "orc_scene_commit_external_bvh": (C.c_int, [_VP, _F, _U32, _U32, C.c_uint32, _U32, C.c_uint32]),
            "orc_scene_bvh_node_count": (C.c_uint32, [_VP]),
            "orc_scene_prim_count": (C.c_uint32, [_VP]),
            "orc_scene_get_bvh": (C.c_int, [_VP, _F, _U32, _U32, _U32]),
            "orc_scene_world_bound": (None, [_VP, _F]),
            "orc_trace_closest": (C.c_int, [_VP, _F, C.c_uint64, _F, _I32, _F, _U64]),
            "orc_trace_any": (C.c_int, [_VP, _F, C.c_uint64, C.POINTER(C.c_uint8), _U64]),
            "orc_sensor_derived": (C.c_int, [C.POINTER(OrcSensor), _I32, _F, _F, _F]),
            "orc_generate_rays": (C.c_int, [C.POINTER(OrcSensor), _F, C.c_uint64, _F]),
            "orc_render": (C.c_int, [_VP, C.POINTER(OrcSensor), C.c_int, C.c_int64, C.c_int, C.c_uint64, C.c_uint32, C.c_int, _F, _F, C.POINTER(OrcStats)]),
            "orc_sppm": (C.c_int, [_VP, C.POINTER(OrcSensor), C.c_float, C.c_int, C.c_int64, C.c_int64, C.c_uint64, _F, _F, _F, _F, C.POINTER(C.c_double),
                                   C.POINTER(C.c_int64), _F, _F, _F, C.POINTER(C.c_int64), C.POINTER(OrcStats)]),
            "orc_sppm_ex": (C.c_int, [_VP, C.POINTER(OrcSensor), C.c_float, C.c_int, C.c_int64, C.c_int64, C.c_uint64, C.c_int, C.c_int64, C.c_int64, EXCHANGE_FN, _VP, _F, _F, _F, _F,
                                      C.POINTER(C.c_double), C.POINTER(C.c_int64), _F, _F, _F, C.POINTER(C.c_int64), C.POINTER(OrcStats)]),
            "orc_radical_inverse": (C.c_float, [C.c_int64, C.c_uint64]),
            "orc_grid_hash": (C.c_uint64, [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64]),
            "orc_distribution1d": (C.c_float, [_F, C.c_int, _F]),
            "orc_sample_discrete": (None, [_F, C.c_int, C.c_float, _F]),
            "orc_to_grid": (None, [_F, _F, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
            "orc_sample_le": (C.c_int, [_VP, C.c_int, _F, _F]),
            "orc_film_to_rgb": (C.c_int, [_F, C.c_int, C.c_int, C.c_float, _F]),
            "orc_num_threads": (C.c_int, []),
            "orc_translate": (None, [_F, _F]),
            "orc_scale": (None, [C.c_float, C.c_float, C.c_float, _F]),
            "orc_look_at": (None, [_F, _F, _F, _F]),
            "orc_perspective": (None, [C.c_float, C.c_float, C.c_float, _F]),
            "orc_transform_from_matrix": (None, [_F, _F]),
            "orc_transform_mul": (None, [_F, _F, _F]),
            "orc_transform_point": (None, [_F, _F, _F]),
            "orc_coordinate_system": (None, [_F, _F]),
            "orc_bounds_intersect": (C.c_int, [_F, _F, _F]),
            "orc_bounds_intersect_p": (C.c_int, [_F, _F]),
            "orc_prim_intersect": (C.c_int, [_VP, C.c_int, _F, _F, _F]),
            "orc_prim_intersect_p": (C.c_int, [_VP, C.c_int, _F]),
            "orc_prim_bounds": (None, [_VP, C.c_int, _F, _F]),
            "orc_triangle_area": (C.c_float, [_VP, C.c_int]),
            "orc_scene_add_nested_bvh": (C.c_int, [_VP, _VP]),
            "orc_fresnel_dielectric": (C.c_float, [C.c_float, C.c_float, C.c_float]),
            "orc_fresnel_conductor": (None, [C.c_float, _F, _F, _F, _F]),
            "orc_roughness_to_alpha": (C.c_float, [C.c_float]),
            "orc_filter_eval": (C.c_float, [C.c_float, C.c_float, C.c_float, C.c_float, C.c_float]),
            "orc_filmtile_new": (_VP, [C.POINTER(OrcSensor), _F, _F, _I32]),
            "orc_filmtile_add_sample": (None, [_VP, C.c_float, C.c_float, _F, C.c_float]),
            "orc_filmtile_read": (None, [_VP, _F]),
            "orc_filmtile_merge": (None, [_VP, _F]),
            "orc_filmtile_free": (None, [_VP]),
            "orc_bxdf_type": (C.c_int, [C.c_int, _F]),
            "orc_bxdf_sample_f": (None, [C.c_int, _F, _F, _F, _F]),
            "orc_bxdf_f_pdf": (None, [C.c_int, _F, _F, _F, _F]),
            "orc_bsdf_query": (C.c_int, [_VP, C.c_int, C.c_int, C.c_int, C.c_int, _F, _F, C.c_uint64, _F]),
            "orc_light_query": (C.c_int, [_VP, C.c_int, _F, C.c_uint64, _F]),
            "orc_sampler_u": (C.c_float, [C.c_uint64, C.c_int32, C.c_int32, C.c_uint32, C.c_uint32]),
            "orc_detmath": (None, [C.c_int, _F, _F, C.c_uint64, _F]),
            "orc_detmath_f64": (None, [C.c_int, C.POINTER(C.c_double), C.c_uint64, C.POINTER(C.c_double)]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def fp(a):
    return a.ctypes.data_as(_F)


def f32a(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def transform32(t) -> np.ndarray:
    """m then inv_m, row-major, as the oracle API passes transformations."""
    return np.concatenate([np.asarray(t.m, np.float32).reshape(-1), np.asarray(t.inv_m, np.float32).reshape(-1)])


def make_sensor(cam, screen_window=(-1.0, -1.0, 1.0, 1.0), fov=90.0, crop=None) -> OrcSensor:
    """Raw constructor arguments: the oracle re-derives film geometry / matrices with its own restated constructors.
    crop: the Film's fractional crop window (default: the one the film was built with)."""
    if crop is None:
        crop = getattr(cam.film, "crop_window", (0.0, 0.0, 1.0, 1.0))
    s = OrcSensor()
    s.camera_to_world[:] = np.asarray(cam.camera_to_world.m, np.float32).reshape(-1).tolist()
    s.camera_to_world_inv[:] = np.asarray(cam.camera_to_world.inv_m, np.float32).reshape(-1).tolist()
    s.screen_window[:] = list(screen_window)
    s.shutter_open, s.shutter_close = float(cam.shutter_open), float(cam.shutter_close)
    s.lens_radius, s.focal_distance, s.fov_deg = float(cam.lens_radius), float(cam.focal_distance), float(fov)
    s.resolution[:] = [float(x) for x in cam.film.resolution]
    s.crop[:] = list(crop)
    s.filter_radius[:] = [float(x) for x in cam.film.filter.radius]
    s.filter_tau = float(cam.film.filter.tau)
    s.film_scale = float(cam.film.scale)
    return s


class OracleScene:
    def __init__(self):
        self.h = lib().orc_scene_new()
        self.n_prims = 0

    def __del__(self):
        try:
            if self.h:
                lib().orc_scene_free(self.h)
                self.h = None
        except Exception:
            pass

    # ---- construction -------------------------------------------------------------------------------------------------
    def add_material(self, kind: int, params) -> int:
        p = f32a(params)
        r = lib().orc_scene_add_material(self.h, kind, fp(p), p.size)
        assert r >= 0
        return r

    def add_sphere(self, o2w, reverse, radius, z_min, z_max, phi_max_deg, material=-1) -> int:
        m, im = f32a(o2w.m), f32a(o2w.inv_m)
        self.n_prims += 1
        return lib().orc_scene_add_sphere(self.h, fp(m), fp(im), int(reverse), radius, z_min, z_max, phi_max_deg, material)

    def add_triangle_mesh(self, o2w, reverse, verts_obj, indices_1based, normals=None, materials=None, tangents=None, uv_corners=None) -> int:
        m, im = f32a(o2w.m), f32a(o2w.inv_m)
        v = f32a(verts_obj).reshape(-1, 3)
        idx = np.ascontiguousarray(indices_1based, dtype=np.uint32).reshape(-1)
        n = None if normals is None else f32a(normals).reshape(-1, 3)
        mats = None if materials is None else np.ascontiguousarray(materials, dtype=np.int32)
        self.n_prims += idx.size // 3
        tg = None if tangents is None else f32a(tangents).reshape(-1, 3)
        uvc = None if uv_corners is None else f32a(uv_corners).reshape(-1, 2)
        return lib().orc_scene_add_triangle_mesh_ex(self.h, fp(m), fp(im), int(reverse), fp(v), v.shape[0], idx.ctypes.data_as(_U32), idx.size // 3,
                                                    fp(n) if n is not None else None, fp(tg) if tg is not None else None, fp(uvc) if uvc is not None else None,
                                                    mats.ctypes.data_as(_I32) if mats is not None else None)

    def add_point_light(self, l2w, I):
        m, im, i = f32a(l2w.m), f32a(l2w.inv_m), f32a(I)
        return lib().orc_scene_add_point_light(self.h, fp(m), fp(im), fp(i))

    def add_spot_light(self, l2w, I, total, falloff):
        m, im, i = f32a(l2w.m), f32a(l2w.inv_m), f32a(I)
        return lib().orc_scene_add_spot_light(self.h, fp(m), fp(im), fp(i), total, falloff)

    def commit_reference(self, max_node_primitives: int = 1):
        rc = lib().orc_scene_commit_reference_bvh(self.h, max_node_primitives)
        if rc:
            raise RuntimeError(lib().orc_last_error().decode())

    def commit_external(self, bounds, a, flags, order):
        bounds, a, flags, order = f32a(bounds), np.ascontiguousarray(a, np.uint32), np.ascontiguousarray(flags, np.uint32), np.ascontiguousarray(order, np.uint32)
        rc = lib().orc_scene_commit_external_bvh(self.h, fp(bounds), a.ctypes.data_as(_U32), flags.ctypes.data_as(_U32), a.size, order.ctypes.data_as(_U32), order.size)
        if rc:
            raise RuntimeError(lib().orc_last_error().decode())

    def get_bvh(self):
        nn, npr = lib().orc_scene_bvh_node_count(self.h), lib().orc_scene_prim_count(self.h)
        bounds = np.empty((nn, 6), np.float32)
        a, flags, order = np.empty(nn, np.uint32), np.empty(nn, np.uint32), np.empty(npr, np.uint32)
        lib().orc_scene_get_bvh(self.h, fp(bounds), a.ctypes.data_as(_U32), flags.ctypes.data_as(_U32), order.ctypes.data_as(_U32))
        return bounds, a, flags, order

    def world_bound(self):
        out = np.empty(6, np.float32)
        lib().orc_scene_world_bound(self.h, fp(out))
        return out

    @classmethod
    def from_scene(cls, scene, bvh=None, max_node_primitives: int = 1) -> "OracleScene":
        """Walk a trace_jl_amd Scene (api.py objects) and rebuild it inside the oracle from the RAW constructor arguments
        (object-space vertices are recovered by storing them on the mesh; see _object_vertices)."""
        import sys
        T = sys.modules["trace_jl_amd"]
        s = cls()
        mat_ids = {}

        def mid(m):
            if m is None:
                return -1
            if id(m) not in mat_ids:
                kind, params = m._flat()
                mat_ids[id(m)] = s.add_material(kind, params)
            return mat_ids[id(m)]

        prims = T.api.splice_nested(scene.aggregate.primitives)
        i = 0
        while i < len(prims):
            p = prims[i]
            if isinstance(p, T.MeshPrimitives):
                mesh = p.mesh
                idx = mesh.indices.reshape(-1, 3)
                uvc = None if getattr(mesh, "uv", None) is None else mesh.uv[:3 * mesh.n_triangles]
                s.add_triangle_mesh(mesh.core.object_to_world, mesh.core.reverse_orientation, mesh.object_vertices, idx, mesh.normals, np.full(idx.shape[0], mid(p.material), np.int32),
                                    getattr(mesh, "tangents", None), uvc)
                i += 1
            elif isinstance(p.shape, T.Sphere):
                sp = p.shape
                s.add_sphere(sp.core.object_to_world, sp.core.reverse_orientation, float(sp.radius), float(sp.z_min), float(sp.z_max), float(sp.phi_max_deg), mid(p.material))
                i += 1
            else:
                mesh = p.shape.mesh
                j = i
                ks, mats = [], []
                while j < len(prims) and not isinstance(prims[j], T.MeshPrimitives) and isinstance(prims[j].shape, T.Triangle) and prims[j].shape.mesh is mesh:
                    ks.append(prims[j].shape.k)
                    mats.append(mid(prims[j].material))
                    j += 1
                idx = mesh.indices.reshape(-1, 3)[np.array(ks)]
                # the oracle transforms object-space vertices itself (triangle_mesh.jl:23); pure translations are undone
                # exactly only by keeping the original array, so the mirror's already-transformed vertices are passed with
                # an identity core when the mesh core is not stored
                verts = getattr(mesh, "object_vertices", None)
                tg = getattr(mesh, "tangents", None)
                uvc = None if getattr(mesh, "uv", None) is None else mesh.uv[:3 * mesh.n_triangles].reshape(-1, 3, 2)[np.array(ks)]
                if verts is None:
                    s.add_triangle_mesh(_identity_like(mesh.core, T), mesh.core.reverse_orientation != mesh.core.transform_swaps_handedness, mesh.vertices, idx, mesh.normals, mats, tg, uvc)
                else:
                    s.add_triangle_mesh(mesh.core.object_to_world, mesh.core.reverse_orientation, verts, idx, mesh.normals, mats, tg, uvc)
                i = j
        for l in scene.lights:
            if isinstance(l, T.PointLight):
                s.add_point_light(l.light_to_world, l.i.c)
            else:
                s.add_spot_light(l.light_to_world, l.i.c, float(l.total_width), float(l.falloff_start))
        if bvh is None:
            s.commit_reference(max_node_primitives)
        else:
            s.commit_external(*bvh)
        return s

    # ---- queries --------------------------------------------------------------------------------------------------------
    def trace_closest(self, rays, want_geom=False):
        rays = f32a(rays).reshape(-1, 8)
        n = rays.shape[0]
        t, prim = np.empty(n, np.float32), np.empty(n, np.int32)
        geom = np.empty((n, 15), np.float32) if want_geom else None
        counts = np.zeros(2, np.uint64)
        lib().orc_trace_closest(self.h, fp(rays), n, fp(t), prim.ctypes.data_as(_I32), fp(geom) if want_geom else None, counts.ctypes.data_as(_U64))
        return t, prim, geom, counts

    def trace_any(self, rays):
        rays = f32a(rays).reshape(-1, 8)
        occ = np.empty(rays.shape[0], np.uint8)
        counts = np.zeros(2, np.uint64)
        lib().orc_trace_any(self.h, fp(rays), rays.shape[0], occ.ctypes.data_as(C.POINTER(C.c_uint8)), counts.ctypes.data_as(_U64))
        return occ, counts

    def render(self, cam, integrator: str, spp: int, max_depth: int, seed: int, sample_offset: int = 0, threads: int = 1, want_samples: bool = False, sensor=None):
        sn = sensor or make_sensor(cam)
        h, w = cam.film.size
        xyzw = np.empty((h, w, 4), np.float32)
        sb = cam.film.get_sample_bounds()
        sbw, sbh = int(sb.p_max[0] - sb.p_min[0]) + 1, int(sb.p_max[1] - sb.p_min[1]) + 1
        L = np.empty((spp, sbh, sbw, 3), np.float32) if want_samples else None
        st = OrcStats()
        rc = lib().orc_render(self.h, C.byref(sn), {"whitted": 0, "path": 1}[integrator], spp, max_depth, seed, sample_offset, threads, fp(xyzw), fp(L) if want_samples else None,
                              C.byref(st))
        if rc:
            raise RuntimeError(lib().orc_last_error().decode())
        return xyzw, L, st

    def sppm(self, cam, initial_radius: float, max_depth: int, n_iterations: int, photons_per_iteration: int = -1, seed: int = 0, sensor=None, threads: int = 1,
             photon_range=None, exchange=None):
        """SPPMIntegrator (oracle/orc_sppm.h).  Returns a dict: image (h, w, 3), Ld, tau, radius, N, and the last
        iteration's M / phi / vp_p / vp_beta (before _update_pixels!), info, stats.
        threads: OpenMP threads over tiles / photons like Threads.@threads (sppm.jl:184, 334); photon_range = (begin, end): the photon
        indices this process traces in every iteration; exchange(phi (n, 3) float32, M (n,) int64): sums them over the processes in place."""
        sn = sensor or make_sensor(cam)
        h, w = cam.film.size
        out = {"image": np.empty((h, w, 3), np.float32), "Ld": np.empty((h, w, 3), np.float32), "tau": np.empty((h, w, 3), np.float32),
               "radius": np.empty((h, w), np.float32), "N": np.empty((h, w), np.float64), "M": np.empty((h, w), np.int64), "phi": np.empty((h, w, 3), np.float32),
               "vp_p": np.empty((h, w, 3), np.float32), "vp_beta": np.empty((h, w, 3), np.float32)}
        info = np.zeros(6, np.int64)
        st = OrcStats()
        i64 = C.POINTER(C.c_int64)
        cb = EXCHANGE_FN()
        if exchange is not None:
            def _cb(_user, phi_p, m_p, n):
                exchange(np.ctypeslib.as_array(phi_p, shape=(int(n), 3)), np.ctypeslib.as_array(m_p, shape=(int(n),)))
            cb = EXCHANGE_FN(_cb)
        pb, pe = (0, -1) if photon_range is None else (int(photon_range[0]), int(photon_range[1]))
        rc = lib().orc_sppm_ex(self.h, C.byref(sn), float(initial_radius), max_depth, n_iterations, photons_per_iteration, seed, int(threads), pb, pe, cb, None, fp(out["image"]), fp(out["Ld"]), fp(out["tau"]),
                            fp(out["radius"]), out["N"].ctypes.data_as(C.POINTER(C.c_double)), out["M"].ctypes.data_as(i64), fp(out["phi"]), fp(out["vp_p"]), fp(out["vp_beta"]),
                            info.ctypes.data_as(i64), C.byref(st))
        if rc:
            raise RuntimeError(lib().orc_last_error().decode())
        out["info"] = {"grid_res": info[:3].copy(), "grid_entries": int(info[3]), "photon_hits": int(info[4]), "photons_per_iteration": int(info[5])}
        out["stats"] = st
        return out

    def bsdf_query(self, material, allow_multiple_lobes, mode, flags, frame9, dirs6):
        frame9, dirs6 = f32a(frame9).reshape(-1, 9), f32a(dirs6).reshape(-1, 6)
        out = np.empty((frame9.shape[0], 8), np.float32)
        rc = lib().orc_bsdf_query(self.h, material, int(allow_multiple_lobes), mode, flags, fp(frame9), fp(dirs6), frame9.shape[0], fp(out))
        assert rc == 0
        return out


def _identity_like(core, T):
    return T.Transformation()


def generate_rays(cam, samples5, sensor=None):
    sn = sensor or make_sensor(cam)
    s = f32a(samples5).reshape(-1, 5)
    out = np.empty((s.shape[0], 8), np.float32)
    lib().orc_generate_rays(C.byref(sn), fp(s), s.shape[0], fp(out))
    return out


def sensor_derived(cam, sensor=None):
    sn = sensor or make_sensor(cam)
    i6 = np.empty(6, np.int32)
    crop, table, r2c = np.empty(4, np.float32), np.empty(256, np.float32), np.empty(16, np.float32)
    lib().orc_sensor_derived(C.byref(sn), i6.ctypes.data_as(_I32), fp(crop), fp(table), fp(r2c))
    return i6, crop, table.reshape(16, 16), r2c.reshape(4, 4)
