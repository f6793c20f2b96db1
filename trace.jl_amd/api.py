"""Host-side mirror of the part of Trace.jl's API that scene scripts touch, in Python.

The reference is Julia and no Julia runtime exists in this image, so the tested host above the C ABI is this module:
same names, same argument meaning, same (load-bearing) constructor bugs, Float32 arithmetic in the reference's
operation order.  Scene scripts written against ``Trace.X`` translate 1:1 to ``trace_jl_amd.X`` (see scenes.py, which
transcribes docs/src/shadows.md).  Everything numerically heavy happens behind ``libtracehip.so``; this module only
*constructs* (matrices, film geometry, filter table) and *flattens* the object graph through the C ABI.

Citations are file:line under /root/reference/src.
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

from . import _ffi
from ._ffi import TraceHipError

f32 = np.float32
_PI32 = f32(3.14159274101257324219)  # Float32(π)


def _deg2rad(x) -> np.float32:  # deg2rad(x::Float32) = x * (Float32(π) / 180f0)
    return f32(x) * (_PI32 / f32(180.0))


# ---- 4x4 Float32 matrices with StaticArrays' operation order --------------------------------------------------------------
def _mat(rows) -> np.ndarray:
    return np.array(rows, dtype=np.float32).reshape(4, 4)


def _mat_mul(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    c = np.empty((4, 4), dtype=np.float32)
    for i in range(4):
        for j in range(4):
            c[i, j] = ((a[i, 0] * b[0, j] + a[i, 1] * b[1, j]) + a[i, 2] * b[2, j]) + a[i, 3] * b[3, j]
    return c


def _det3(a, b, c, d, e, f, g, h, i):
    return a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g)


def _mat_inv(A: np.ndarray) -> np.ndarray:
    """inv(::Mat4f) as cofactor * (1/det) (StaticArrays' closed form; term order inside a cofactor is a documented
    tolerance source for general matrices)."""
    cof = np.empty((4, 4), dtype=np.float32)
    for r in range(4):
        for c in range(4):
            s = [A[i, j] for i in range(4) if i != r for j in range(4) if j != c]
            d = _det3(*s)
            cof[r, c] = -d if (r + c) & 1 else d
    det = ((A[0, 0] * cof[0, 0] + A[0, 1] * cof[0, 1]) + A[0, 2] * cof[0, 2]) + A[0, 3] * cof[0, 3]
    idet = f32(1.0) / det
    R = np.empty((4, 4), dtype=np.float32)
    for r in range(4):
        for c in range(4):
            R[r, c] = cof[c, r] * idet
    return R


class Transformation:
    """transformations.jl:1-22.  ``*`` multiplies the inverses in the same order (bug A.3, load-bearing)."""

    def __init__(self, m: Optional[np.ndarray] = None, inv_m: Optional[np.ndarray] = None):
        if m is None:
            m = np.eye(4, dtype=np.float32)
            inv_m = np.eye(4, dtype=np.float32)
        m = np.asarray(m, dtype=np.float32).reshape(4, 4)
        self.m = m
        self.inv_m = _mat_inv(m) if inv_m is None else np.asarray(inv_m, dtype=np.float32).reshape(4, 4)

    def __mul__(self, o: "Transformation") -> "Transformation":
        return Transformation(_mat_mul(self.m, o.m), _mat_mul(self.inv_m, o.inv_m))

    def inv(self) -> "Transformation":
        return Transformation(self.inv_m, self.m)

    def point(self, p) -> np.ndarray:  # :132-138
        m = self.m
        p = np.asarray(p, dtype=np.float32)
        one = f32(1.0)
        v = [((m[i, 0] * p[0] + m[i, 1] * p[1]) + m[i, 2] * p[2]) + m[i, 3] * one for i in range(4)]
        if v[3] == 1:
            return np.array(v[:3], dtype=np.float32)
        return np.array([v[0] / v[3], v[1] / v[3], v[2] / v[3]], dtype=np.float32)

    def vector(self, v) -> np.ndarray:  # :139
        m = self.m
        v = np.asarray(v, dtype=np.float32)
        return np.array([(m[i, 0] * v[0] + m[i, 1] * v[1]) + m[i, 2] * v[2] for i in range(3)], dtype=np.float32)

    def swaps_handedness(self) -> bool:  # :161-163
        m = self.m
        return bool(_det3(m[0, 0], m[0, 1], m[0, 2], m[1, 0], m[1, 1], m[1, 2], m[2, 0], m[2, 1], m[2, 2]) < 0)


def inv(t: Transformation) -> Transformation:
    return t.inv()


def translate(d) -> Transformation:  # :24-38
    d = np.asarray(d, dtype=np.float32)
    return Transformation(_mat([[1, 0, 0, d[0]], [0, 1, 0, d[1]], [0, 0, 1, d[2]], [0, 0, 0, 1]]),
                          _mat([[1, 0, 0, -d[0]], [0, 1, 0, -d[1]], [0, 0, 1, -d[2]], [0, 0, 0, 1]]))


def scale(x, y, z) -> Transformation:  # :40-54
    x, y, z = f32(x), f32(y), f32(z)
    one = f32(1.0)
    return Transformation(_mat([[x, 0, 0, 0], [0, y, 0, 0], [0, 0, z, 0], [0, 0, 0, 1]]),
                          _mat([[one / x, 0, 0, 0], [0, one / y, 0, 0], [0, 0, one / z, 0], [0, 0, 0, 1]]))


def _norm3(v):
    return np.sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2])


def _normalize(v):
    v = np.asarray(v, dtype=np.float32)
    return (f32(1.0) / _norm3(v)) * v


def _cross(a, b):
    return np.array([a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]], dtype=np.float32)


def look_at(position, target, up) -> Transformation:  # :105-117
    position = np.asarray(position, dtype=np.float32)
    target = np.asarray(target, dtype=np.float32)
    up = np.asarray(up, dtype=np.float32)
    z = _normalize(position - target)
    x = _normalize(_cross(up, z))
    y = _cross(z, x)
    m = _mat([[x[0], y[0], z[0], 0], [x[1], y[1], z[1], 0], [x[2], y[2], z[2], 0], [0, 0, 0, 1]])
    return translate(position) * Transformation(m, m.T.copy())


def perspective(fov, near, far) -> Transformation:  # :119-130 — Mat4f literal filled column-major, no transpose (A.4)
    fov, near, far = f32(fov), f32(near), f32(far)
    p = np.zeros((4, 4), dtype=np.float32)
    p[0, 0] = 1
    p[1, 1] = 1
    p[2, 2] = far / (far - near)
    p[3, 2] = -far * near / (far - near)
    p[2, 3] = 1
    p[3, 3] = 0
    inv_tan = f32(1.0) / _ffi.detmath(2, _deg2rad(fov) / f32(2.0))[0]
    return scale(inv_tan, inv_tan, f32(1.0)) * Transformation(p)


def coordinate_system(v1):  # Trace.jl:139-146
    v1 = np.asarray(v1, dtype=np.float32)
    if abs(v1[0]) > abs(v1[1]):
        v2 = np.array([-v1[2], 0, v1[0]], dtype=np.float32) / np.sqrt(v1[0] * v1[0] + v1[2] * v1[2])
    else:
        v2 = np.array([0, v1[2], -v1[1]], dtype=np.float32) / np.sqrt(v1[1] * v1[1] + v1[2] * v1[2])
    return v1, v2, _cross(v1, v2)


# ---- spectrum / textures / materials -----------------------------------------------------------------------------------------
class RGBSpectrum:  # spectrum.jl:56-61
    def __init__(self, r=0.0, g=None, b=None):
        self.c = np.array([r, r, r] if g is None else [r, g, b], dtype=np.float32)


class ConstantTexture:  # textures/basic.jl:4-10
    def __init__(self, value):
        self.value = value


def _tex_rgb(t) -> List[float]:
    v = t.value if isinstance(t, ConstantTexture) else t
    return [float(x) for x in (v.c if isinstance(v, RGBSpectrum) else np.full(3, v, dtype=np.float32))]


def _tex_f(t) -> float:
    v = t.value if isinstance(t, ConstantTexture) else t
    return float(f32(v))


@dataclass
class MatteMaterial:  # materials/material.jl:1-31
    Kd: ConstantTexture
    sigma: ConstantTexture

    def _flat(self):
        return 0, _tex_rgb(self.Kd) + [_tex_f(self.sigma)]


@dataclass
class MirrorMaterial:  # :34-46
    Kr: ConstantTexture

    def _flat(self):
        return 1, _tex_rgb(self.Kr)


@dataclass
class GlassMaterial:  # :49-116
    Kr: ConstantTexture
    Kt: ConstantTexture
    u_roughness: ConstantTexture
    v_roughness: ConstantTexture
    index: ConstantTexture
    remap_roughness: bool

    def _flat(self):
        return 2, _tex_rgb(self.Kr) + _tex_rgb(self.Kt) + [_tex_f(self.u_roughness), _tex_f(self.v_roughness), _tex_f(self.index), 1.0 if self.remap_roughness else 0.0]


@dataclass
class PlasticMaterial:  # :119-151
    Kd: ConstantTexture
    Ks: ConstantTexture
    roughness: ConstantTexture
    remap_roughness: bool

    def _flat(self):
        return 3, _tex_rgb(self.Kd) + _tex_rgb(self.Ks) + [_tex_f(self.roughness), 1.0 if self.remap_roughness else 0.0]


# ---- shapes / primitives ---------------------------------------------------------------------------------------------------------
class ShapeCore:  # shapes/Shape.jl:1-15
    def __init__(self, object_to_world: Transformation, reverse_orientation: bool):
        self.object_to_world = object_to_world
        self.world_to_object = object_to_world.inv()
        self.reverse_orientation = bool(reverse_orientation)
        self.transform_swaps_handedness = object_to_world.swaps_handedness()


class Sphere:  # shapes/sphere.jl:1-30 (clamps and angles are derived inside the library, sphere.jl:13-26)
    def __init__(self, core: ShapeCore, radius, *args):
        self.core = core
        self.radius = f32(radius)
        if len(args) == 1:
            self.z_min, self.z_max, self.phi_max_deg = -self.radius, self.radius, f32(args[0])
        else:
            self.z_min, self.z_max, self.phi_max_deg = f32(args[0]), f32(args[1]), f32(args[2])


class TriangleMesh:  # shapes/triangle_mesh.jl:1-30: vertices go to world space here (:23), normals and tangents do not
    def __init__(self, core: ShapeCore, indices, vertices, normals=None, tangents=None, uv=None):
        self.core = core
        self.indices = np.ascontiguousarray(indices, dtype=np.uint32).reshape(-1)
        v = np.ascontiguousarray(vertices, dtype=np.float32).reshape(-1, 3)
        self.object_vertices = v  # as passed by the caller (the oracle bridge re-derives world space from these)
        self.vertices = transform_points(core.object_to_world, v)
        self.normals = None if normals is None else np.ascontiguousarray(normals, dtype=np.float32).reshape(-1, 3)
        self.n_triangles = self.indices.size // 3
        # optional: one tangent per vertex (:11-12); (u, v)s, which the reference reads by CORNER position `mesh.uv[t.i + j]` (:82), i.e. 3 per triangle
        self.tangents = None if tangents is None else np.ascontiguousarray(tangents, dtype=np.float32).reshape(-1, 3)
        self.uv = None if uv is None else np.ascontiguousarray(uv, dtype=np.float32).reshape(-1, 2)
        if self.tangents is not None and self.tangents.shape[0] != v.shape[0]:
            raise ValueError("tangents: one per vertex")
        if self.uv is not None and self.uv.shape[0] < 3 * self.n_triangles:
            raise ValueError("uv: the reference indexes mesh.uv by corner position (triangle_mesh.jl:82): 3 * n_triangles entries are read")


def transform_points(t: Transformation, v: np.ndarray) -> np.ndarray:
    """(t::Transformation)(p::Point3f) for many points, Float32, left-to-right (transformations.jl:132-138)."""
    m = t.m
    one = f32(1.0)
    cols = [((m[i, 0] * v[:, 0] + m[i, 1] * v[:, 1]) + m[i, 2] * v[:, 2]) + m[i, 3] * one for i in range(4)]
    w = cols[3]
    out = np.stack(cols[:3], axis=1).astype(np.float32)
    div = w != 1
    if np.any(div):
        out[div] = out[div] / w[div, None]
    return np.ascontiguousarray(out, dtype=np.float32)


@dataclass
class Triangle:  # :32-43
    mesh: TriangleMesh
    k: int  # 0-based triangle number; the reference stores i = 3k + 1


def create_triangle_mesh(core: ShapeCore, n_triangles: int, indices, n_vertices: int, vertices, normals=None, tangents=None, uv=None) -> List[Triangle]:  # :45-58
    mesh = TriangleMesh(core, indices, vertices, normals, tangents, uv)
    assert mesh.n_triangles == n_triangles and mesh.vertices.shape[0] == n_vertices
    return [Triangle(mesh, k) for k in range(n_triangles)]


@dataclass
class GeometricPrimitive:  # primitive.jl:1-9
    shape: object
    material: object = None


@dataclass
class MeshPrimitives:
    """Bulk form of `[GeometricPrimitive(t, material) for t in create_triangle_mesh(...)]` for million-triangle meshes
    (one Python object instead of one per triangle); expands to exactly that list, in order."""
    mesh: "TriangleMesh"
    material: object = None


def create_mesh_primitives(core: ShapeCore, indices, vertices, normals=None, material=None, tangents=None, uv=None) -> MeshPrimitives:
    return MeshPrimitives(TriangleMesh(core, indices, vertices, normals, tangents, uv), material)


class BVHAccel:  # accel/bvh.jl:50-79: the tree itself is built inside the library at Scene flattening
    def __init__(self, primitives: Sequence[GeometricPrimitive], max_node_primitives: int = 1):
        self.primitives = list(primitives)
        self.max_node_primitives = min(255, int(max_node_primitives))


@dataclass
class PointLight:  # lights/point.jl:19-24
    light_to_world: Transformation
    i: RGBSpectrum


@dataclass
class SpotLight:  # lights/spot.jl:10-19
    light_to_world: Transformation
    i: RGBSpectrum
    total_width: float
    falloff_start: float


class Scene:  # Trace.jl:176-187
    def __init__(self, lights, aggregate: BVHAccel):
        self.lights = list(lights)
        self.aggregate = aggregate
        self._flat = None

    def flatten(self, ctx: Optional[_ffi.Context] = None) -> "FlatScene":
        if self._flat is None or (ctx is not None and self._flat.ctx is not ctx):
            self._flat = FlatScene(self, ctx or _ffi.default_context())
        return self._flat


# ---- sensor ----------------------------------------------------------------------------------------------------------------------
@dataclass
class Bounds2:  # bounds.jl:1-4
    p_min: Sequence[float]
    p_max: Sequence[float]


@dataclass
class LanczosSincFilter:  # filter.jl:3-23
    radius: Sequence[float]
    tau: float

    def __call__(self, p) -> np.float32:
        return self._ws(f32(p[0]), f32(self.radius[0])) * self._ws(f32(p[1]), f32(self.radius[1]))

    def _sinc(self, x):
        x = abs(x)
        if x < f32(1e-5):
            return f32(1.0)
        x = x * _PI32
        return _ffi.detmath(0, x)[0] / x

    def _ws(self, x, r):
        x = abs(x)
        if x > r:
            return f32(0.0)
        return self._sinc(x) * self._sinc(x / f32(self.tau))


class Film:  # film.jl:7-62
    def __init__(self, resolution, crop_bounds: Bounds2, filter: LanczosSincFilter, diagonal, scale, filename: str):
        self.resolution = np.asarray(resolution, dtype=np.float32).reshape(2)
        res = self.resolution
        cmin = np.asarray(crop_bounds.p_min, dtype=np.float32)
        cmax = np.asarray(crop_bounds.p_max, dtype=np.float32)
        self.crop_window = (float(cmin[0]), float(cmin[1]), float(cmax[0]), float(cmax[1]))  # the constructor's fractional window (kept for hosts / tests)
        self.crop_bounds = Bounds2(np.ceil(res * cmin) + f32(1.0), np.ceil(res * cmax))  # :41-44
        self.filter = filter
        self.diagonal = f32(diagonal) * f32(0.001)
        self.scale = f32(scale)
        self.filename = filename
        w = int(abs(self.crop_bounds.p_max[0] - (self.crop_bounds.p_min[0] - f32(1.0))))  # inclusive_sides bounds.jl:100-102
        h = int(abs(self.crop_bounds.p_max[1] - (self.crop_bounds.p_min[1] - f32(1.0))))
        self.xyz = np.zeros((h, w, 3), dtype=np.float32)  # Pixel.xyz, (y, x)
        self.filter_weight_sum = np.zeros((h, w), dtype=np.float32)
        self.splat_xyz = np.zeros((h, w, 3), dtype=np.float32)
        self.filter_table_width = 16
        r = np.asarray(filter.radius, dtype=np.float32) / f32(16)
        self.filter_table = np.empty((16, 16), dtype=np.float32)  # (y, x) :55-59
        for y in range(16):
            for x in range(16):
                self.filter_table[y, x] = filter(((f32(x) + f32(0.5)) * r[0], (f32(y) + f32(0.5)) * r[1]))

    @property
    def size(self):
        return self.xyz.shape[:2]

    def get_sample_bounds(self) -> Bounds2:  # :68-73
        r = np.asarray(self.filter.radius, dtype=np.float32)
        return Bounds2(np.floor(np.asarray(self.crop_bounds.p_min) + f32(0.5) - r), np.ceil(np.asarray(self.crop_bounds.p_max) - f32(0.5) + r))

    def set_xyzw(self, xyzw: np.ndarray):
        self.xyz[...] = xyzw[..., :3]
        self.filter_weight_sum[...] = xyzw[..., 3]

    def to_rgb(self, ctx: Optional[_ffi.Context] = None) -> np.ndarray:
        """save(film) up to the encoder (film.jl:204-222): linear RGB in [0,1], rows not flipped; computed on the GPU."""
        ctx = ctx or _ffi.default_context()
        h, w = self.size
        xyzw = np.ascontiguousarray(np.concatenate([self.xyz, self.filter_weight_sum[..., None]], axis=-1), dtype=np.float32)
        out = np.empty((h, w, 3), dtype=np.float32)
        ctx.check(_ffi.lib().trhip_film_to_rgb(ctx._h, _ffi.fptr(xyzw), w, h, float(self.scale), _ffi.fptr(out)))
        return out


def save(film: Film, ctx: Optional[_ffi.Context] = None) -> str:
    """film.jl:204-222: writes film.filename (8-bit PNG, rows flipped, no gamma)."""
    from PIL import Image
    rgb = film.to_rgb(ctx)
    img = np.clip(np.rint(rgb[::-1] * 255.0), 0, 255).astype(np.uint8)
    Image.fromarray(img, "RGB").save(film.filename)
    return film.filename


class PerspectiveCamera:  # camera/perspective.jl:11-40, 58-80
    def __init__(self, camera_to_world: Transformation, screen_window: Bounds2, shutter_open, shutter_close, lens_radius, focal_distance, fov, film: Film):
        self.camera_to_world = camera_to_world
        self.shutter_open, self.shutter_close = f32(shutter_open), f32(shutter_close)
        self.lens_radius, self.focal_distance = f32(lens_radius), f32(focal_distance)
        self.film = film
        self.camera_to_screen = perspective(fov, 0.01, 1000.0)  # near / far hard-coded at :65
        smin = np.asarray(screen_window.p_min, dtype=np.float32)
        smax = np.asarray(screen_window.p_max, dtype=np.float32)
        one = f32(1.0)
        self.screen_to_raster = (scale(film.resolution[0], film.resolution[1], 1) * scale(one / (smax[0] - smin[0]), one / (smax[1] - smin[1]), 1)
                                 * translate([-smin[0], -smax[1], f32(0.0)]))
        self.raster_to_screen = self.screen_to_raster.inv()
        self.raster_to_camera = self.camera_to_screen.inv() * self.raster_to_screen

    def sensor(self) -> _ffi.Sensor:
        s = _ffi.Sensor()
        s.raster_to_camera[:] = self.raster_to_camera.m.reshape(-1).tolist()
        s.camera_to_world[:] = self.camera_to_world.m.reshape(-1).tolist()
        s.lens_radius, s.focal_distance = float(self.lens_radius), float(self.focal_distance)
        s.shutter_open, s.shutter_close = float(self.shutter_open), float(self.shutter_close)
        f = self.film
        s.crop_min[:] = [float(x) for x in f.crop_bounds.p_min]
        s.crop_max[:] = [float(x) for x in f.crop_bounds.p_max]
        s.filter_radius[:] = [float(x) for x in f.filter.radius]
        s.filter_table[:] = f.filter_table.reshape(-1).tolist()
        s.scale = float(f.scale)
        return s


def get_film(camera: PerspectiveCamera) -> Film:  # perspective.jl:83
    return camera.film


# ---- sampler -------------------------------------------------------------------------------------------------------------------------
class SeededSampler:
    """The build's seeded counter-based sampler (include/trace_sampler.h) behind UniformSampler's protocol
    (sampler/sampler.jl:129-151).  The reference's UniformSampler draws from Julia's unseeded global RNG."""

    def __init__(self, samples_per_pixel: int, seed: int = 0x5EED0001, sample_offset: int = 0):
        self.samples_per_pixel = int(samples_per_pixel)
        self.seed = int(seed)
        self.sample_offset = int(sample_offset)
        self.current_sample = 1
        self._pixel = (0, 0)
        self._dim = 0

    # -- the AbstractSampler protocol the render loops call (sampler/sampler.jl:129-151); values from include/trace_sampler.h.
    #    The kernels address dimensions directly (camera 0-4, path vertex v at 5 + 8 v); a host that walks the protocol
    #    sequentially consumes the same stream as long as it positions itself with start_vertex(v) at every path vertex.
    def _u(self, dim: int) -> np.float32:
        from . import scenes
        key = scenes.ts_stream_key(self.seed, self._pixel[0], self._pixel[1], self.sample_offset + self.current_sample - 1)
        return np.float32(scenes.ts_uniform(key, dim))

    def start_pixel(self, p):  # start_pixel! :147-149
        self.current_sample = 1
        self._pixel = (int(p[0]), int(p[1]))
        self._dim = 0

    def has_next_sample(self) -> bool:  # :141-143
        return self.current_sample <= self.samples_per_pixel

    def start_next_sample(self):  # start_next_sample! :144-146
        self.current_sample += 1
        self._dim = 0

    def start_vertex(self, v: int):
        self._dim = 5 + 8 * int(v)

    def get_1d(self) -> np.float32:  # :131
        self._dim += 1
        return self._u(self._dim - 1)

    def get_2d(self) -> np.ndarray:  # :132-134
        return np.array([self.get_1d(), self.get_1d()], dtype=np.float32)

    def get_camera_sample(self, p_raster):  # :135-139: (p_film, p_lens, time)
        self._dim = 0
        p = np.asarray(p_raster, dtype=np.float32)
        film = p + self.get_2d()
        return film.astype(np.float32), self.get_2d(), self.get_1d()


def UniformSampler(samples_per_pixel: int) -> SeededSampler:
    return SeededSampler(samples_per_pixel)


# ---- flattening ---------------------------------------------------------------------------------------------------------------------
def splice_nested(prims):
    """A BVHAccel may itself be a primitive of another (accel/bvh.jl:50-53, test/test_intersection.jl:137-138): its primitives are
    spliced in place — the library builds ONE BVH over the flat list (any BVH over the same primitives gives the same hits except
    exact-t ties)."""
    out = []
    for p in prims:
        if isinstance(p, BVHAccel):
            out += splice_nested(p.primitives)
        else:
            out.append(p)
    return out


class FlatScene:
    """Walk Scene -> BVHAccel -> GeometricPrimitive -> shape/material and push everything through the C ABI."""

    def __init__(self, scene: Scene, ctx: _ffi.Context):
        L = _ffi.lib()
        self.ctx = ctx
        self._h = C.c_void_p()
        ctx.check(L.trhip_scene_new(ctx._h, C.byref(self._h)))
        mat_ids = {}

        def material_id(m):
            if m is None:
                return 0x00FFFFFF
            if id(m) not in mat_ids:
                kind, params = m._flat()
                p = np.array(params, dtype=np.float32)
                out = C.c_uint32()
                ctx.check(L.trhip_scene_add_material(self._h, kind, _ffi.fptr(p), p.size, C.byref(out)))
                mat_ids[id(m)] = out.value
            return mat_ids[id(m)]

        prims = splice_nested(scene.aggregate.primitives)
        self.n_prims = len(prims)
        i = 0
        while i < len(prims):
            p = prims[i]
            if not isinstance(p, MeshPrimitives) and isinstance(p.shape, Sphere):
                s = p.shape
                o2w = s.core.object_to_world
                m, im = _ffi.f32(o2w.m), _ffi.f32(o2w.inv_m)
                ctx.check(L.trhip_scene_add_sphere(self._h, _ffi.fptr(m), _ffi.fptr(im), int(s.core.reverse_orientation), float(s.radius), float(s.z_min), float(s.z_max),
                                                   float(s.phi_max_deg), material_id(p.material), None))
                i += 1
            elif isinstance(p, MeshPrimitives):
                mesh = p.mesh
                idx = np.ascontiguousarray(mesh.indices.reshape(-1, 3), dtype=np.uint32)
                mats = np.full(idx.shape[0], material_id(p.material), dtype=np.uint32)
                core = mesh.core
                flip = int(core.reverse_orientation != core.transform_swaps_handedness)
                self._add_triangles(ctx, mesh, idx, mats, flip, None)
                i += 1
            elif isinstance(p.shape, Triangle):
                # batch consecutive triangles of the same mesh into one call (caller order is preserved)
                mesh = p.shape.mesh
                j = i
                ks, mats = [], []
                while j < len(prims) and not isinstance(prims[j], MeshPrimitives) and isinstance(prims[j].shape, Triangle) and prims[j].shape.mesh is mesh:
                    ks.append(prims[j].shape.k)
                    mats.append(material_id(prims[j].material))
                    j += 1
                idx = np.ascontiguousarray(mesh.indices.reshape(-1, 3)[np.array(ks)], dtype=np.uint32)
                mats = np.array(mats, dtype=np.uint32)
                core = mesh.core
                flip = int(core.reverse_orientation != core.transform_swaps_handedness)
                self._add_triangles(ctx, mesh, idx, mats, flip, np.array(ks))
                i = j
            else:
                raise TraceHipError(f"unsupported shape {type(p.shape).__name__}")
        for l in scene.lights:
            m, im = _ffi.f32(l.light_to_world.m), _ffi.f32(l.light_to_world.inv_m)
            I = _ffi.f32(l.i.c)
            if isinstance(l, PointLight):
                ctx.check(L.trhip_scene_add_point_light(self._h, _ffi.fptr(m), _ffi.fptr(im), _ffi.fptr(I)))
            elif isinstance(l, SpotLight):
                ctx.check(L.trhip_scene_add_spot_light(self._h, _ffi.fptr(m), _ffi.fptr(im), _ffi.fptr(I), float(l.total_width), float(l.falloff_start)))
            else:
                raise TraceHipError(f"unsupported light {type(l).__name__}")
        ctx.check(L.trhip_scene_commit(self._h, scene.aggregate.max_node_primitives))

    def _add_triangles(self, ctx, mesh, idx, mats, flip, ks):
        """One trhip_scene_add_triangles(_ex) call; ks = the triangle numbers of `idx` inside the mesh (None: all, in order) — the corner uvs follow them."""
        L = _ffi.lib()
        nrm, tan, uv = mesh.normals, getattr(mesh, "tangents", None), getattr(mesh, "uv", None)
        if tan is None and uv is None:
            ctx.check(L.trhip_scene_add_triangles(self._h, _ffi.fptr(mesh.vertices), mesh.vertices.shape[0], _ffi.u32ptr(idx), idx.shape[0],
                                                  _ffi.fptr(nrm) if nrm is not None else None, _ffi.u32ptr(mats), flip, None))
            return
        if uv is not None:
            uv = uv[:3 * mesh.n_triangles].reshape(-1, 3, 2)
            uv = np.ascontiguousarray(uv if ks is None else uv[ks], dtype=np.float32)
        ctx.check(L.trhip_scene_add_triangles_ex(self._h, _ffi.fptr(mesh.vertices), mesh.vertices.shape[0], _ffi.u32ptr(idx), idx.shape[0],
                                                 _ffi.fptr(nrm) if nrm is not None else None, _ffi.fptr(tan) if tan is not None else None,
                                                 _ffi.fptr(uv) if uv is not None else None, _ffi.u32ptr(mats), flip, None))

    def bvh(self):
        L = _ffi.lib()
        nn, npr = C.c_uint32(), C.c_uint32()
        self.ctx.check(L.trhip_scene_bvh_size(self._h, C.byref(nn), C.byref(npr)))
        bounds = np.empty((nn.value, 6), dtype=np.float32)
        a = np.empty(nn.value, dtype=np.uint32)
        flags = np.empty(nn.value, dtype=np.uint32)
        order = np.empty(npr.value, dtype=np.uint32)
        self.ctx.check(L.trhip_scene_get_bvh(self._h, _ffi.fptr(bounds), _ffi.u32ptr(a), _ffi.u32ptr(flags), _ffi.u32ptr(order)))
        return bounds, a, flags, order

    def bvh_note(self) -> str:
        """Why a default commit holds one tree instead of two (trhip_scene_bvh_note); "" when it holds both."""
        buf = C.create_string_buffer(512)
        self.ctx.check(_ffi.lib().trhip_scene_bvh_note(self._h, buf, 512))
        return buf.value.decode()

    def closest_kernel_name(self) -> str:
        """The kernel a closest-hit launch on this scene runs under the context's current options (trhip_closest_kernel_name)."""
        buf = C.create_string_buffer(64)
        self.ctx.check(_ffi.lib().trhip_closest_kernel_name(self.ctx._h, self._h, buf, 64))
        return buf.value.decode()

    def bvh_mode(self):
        """(mode, accelerator nodes, accelerator depth): mode 0 = the library's tree alone, 1 = the canonical (reference / host) tree alone, 2 = hybrid: the canonical
        tree defines the answers, the library's tree accelerates the rays that carry the order-independence certificate (csrc/th_trace3c.h); 3 = the library's tree is the
        canonical one (the reference's construction fails on this scene) and, four children wide, its own accelerator."""
        mode, nn, dep = C.c_int(), C.c_uint32(), C.c_uint32()
        self.ctx.check(_ffi.lib().trhip_scene_bvh_mode(self._h, C.byref(mode), C.byref(nn), C.byref(dep)))
        return mode.value, nn.value, dep.value

    def accelerator(self):
        """The accelerator tree of a hybrid scene, in the layout of bvh() (order[accelerator slot] = caller primitive index)."""
        mode, nn, _ = self.bvh_mode()
        if mode not in (2, 3):
            raise _ffi.TraceHipError("the scene has no accelerator tree")
        L = _ffi.lib()
        npr = C.c_uint32()
        self.ctx.check(L.trhip_scene_bvh_size(self._h, None, C.byref(npr)))
        bounds = np.empty((nn, 6), dtype=np.float32)
        a = np.empty(nn, dtype=np.uint32)
        flags = np.empty(nn, dtype=np.uint32)
        order = np.empty(npr.value, dtype=np.uint32)
        self.ctx.check(L.trhip_scene_get_accelerator(self._h, _ffi.fptr(bounds), _ffi.u32ptr(a), _ffi.u32ptr(flags), _ffi.u32ptr(order)))
        return bounds, a, flags, order

    def set_bvh(self, bounds, a, flags, order):
        bounds, a, flags, order = _ffi.f32(bounds), np.ascontiguousarray(a, np.uint32), np.ascontiguousarray(flags, np.uint32), np.ascontiguousarray(order, np.uint32)
        self.ctx.check(_ffi.lib().trhip_scene_set_bvh(self._h, _ffi.fptr(bounds), _ffi.u32ptr(a), _ffi.u32ptr(flags), a.size, _ffi.u32ptr(order), order.size))

    def trace_closest(self, rays: np.ndarray) -> np.ndarray:
        rays = _ffi.f32(rays).reshape(-1, 8)
        out = np.empty(rays.shape[0], dtype=_ffi.HIT_DTYPE)
        self.ctx.check(_ffi.lib().trhip_trace_closest(self.ctx._h, self._h, _ffi.fptr(rays), rays.shape[0], out.ctypes.data_as(C.c_void_p)))
        return out

    def accelerator_note(self) -> str:
        """Why this scene's accelerator is idle under the context's current options (trhip_accelerator_note); "" when it is used."""
        buf = C.create_string_buffer(512)
        self.ctx.check(_ffi.lib().trhip_accelerator_note(self.ctx._h, self._h, buf, 512))
        return buf.value.decode()

    def last_fallback(self):
        """(rays, rays re-walked on the canonical tree) of the last closest-hit trace call (hybrid mode: th_trace3c.h)."""
        out = np.zeros(2, np.uint64)
        self.ctx.check(_ffi.lib().trhip_last_fallback_counts(self.ctx._h, out.ctypes.data_as(C.POINTER(C.c_uint64))))
        return int(out[0]), int(out[1])

    def trace_any(self, rays: np.ndarray) -> np.ndarray:
        rays = _ffi.f32(rays).reshape(-1, 8)
        out = np.empty(rays.shape[0], dtype=np.uint8)
        self.ctx.check(_ffi.lib().trhip_trace_any(self.ctx._h, self._h, _ffi.fptr(rays), rays.shape[0], out.ctypes.data_as(C.POINTER(C.c_uint8))))
        return out

    def hit_geometry(self, rays: np.ndarray) -> np.ndarray:
        rays = _ffi.f32(rays).reshape(-1, 8)
        out = np.empty((rays.shape[0], 15), dtype=np.float32)
        self.ctx.check(_ffi.lib().trhip_hit_geometry(self.ctx._h, self._h, _ffi.fptr(rays), rays.shape[0], _ffi.fptr(out)))
        return out

    def bsdf_query(self, material: int, allow_multiple_lobes: bool, mode: int, flags: int, frame9, dirs6) -> np.ndarray:
        frame9, dirs6 = _ffi.f32(frame9).reshape(-1, 9), _ffi.f32(dirs6).reshape(-1, 6)
        out = np.empty((frame9.shape[0], 8), dtype=np.float32)
        self.ctx.check(_ffi.lib().trhip_bsdf_query(self.ctx._h, self._h, material, int(allow_multiple_lobes), mode, flags, _ffi.fptr(frame9), _ffi.fptr(dirs6), frame9.shape[0], _ffi.fptr(out)))
        return out

    def free(self):
        if self._h:
            _ffi.lib().trhip_scene_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


# ---- integrators -------------------------------------------------------------------------------------------------------------------
class _SamplerIntegrator:
    _entry = None
    _entry_device = None

    def __init__(self, camera: PerspectiveCamera, sampler: SeededSampler, max_depth: int):
        self.camera, self.sampler, self.max_depth = camera, sampler, int(max_depth)
        self.stats: Optional[_ffi.Stats] = None

    def render(self, scene: Scene, ctx: Optional[_ffi.Context] = None, device_out: Optional[int] = None) -> np.ndarray:
        """Render into camera.film (and return xyzw, H x W x 4).  With ``device_out`` (a device pointer) the film
        accumulators are written there instead and nothing is copied to the host."""
        flat = scene.flatten(ctx)
        ctx = flat.ctx
        sn = self.camera.sensor()
        st = _ffi.Stats()
        film = self.camera.film
        h, w = film.size
        smp = self.sampler
        L = _ffi.lib()
        if device_out is not None:
            ctx.check(getattr(L, self._entry_device)(ctx._h, flat._h, C.byref(sn), smp.samples_per_pixel, self.max_depth, smp.seed, smp.sample_offset, C.c_void_p(device_out), C.byref(st)))
            self.stats = st
            return None
        out = np.empty((h, w, 4), dtype=np.float32)
        ctx.check(getattr(L, self._entry)(ctx._h, flat._h, C.byref(sn), smp.samples_per_pixel, self.max_depth, smp.seed, smp.sample_offset, _ffi.fptr(out), C.byref(st)))
        self.stats = st
        film.set_xyzw(out)
        return out

    def sample_radiance(self, scene: Scene) -> np.ndarray:
        """Per-sample radiance of the last render: (spp, sb_h, sb_w, 3)."""
        flat = scene.flatten()
        sb = self.camera.film.get_sample_bounds()
        sbw = int(sb.p_max[0] - sb.p_min[0]) + 1
        sbh = int(sb.p_max[1] - sb.p_min[1]) + 1
        out = np.empty((self.sampler.samples_per_pixel, sbh, sbw, 3), dtype=np.float32)
        flat.ctx.check(_ffi.lib().trhip_last_sample_radiance(flat.ctx._h, _ffi.fptr(out), out.size))
        return out

    def __call__(self, scene: Scene):
        """`integrator(scene)` / `scene |> integrator` (integrators/sampler.jl:12-56): render, then save(film)."""
        self.render(scene)
        if self.camera.film.filename:
            return save(self.camera.film)
        return None


class WhittedIntegrator(_SamplerIntegrator):  # integrators/sampler.jl:3-7
    _entry = "trhip_render_whitted"
    _entry_device = "trhip_render_whitted_device"


class PathIntegrator(_SamplerIntegrator):
    """Not in the reference; defined in DESIGN.md from integrators/sppm.jl:208-266, 503-554."""
    _entry = "trhip_render_path"
    _entry_device = "trhip_render_path_device"


class SPPMIntegrator:  # integrators/sppm.jl:108-130
    """SPPMIntegrator(camera, initial_search_radius, max_depth, n_iterations, photons_per_iteration = -1, write_frequency = 1).

    ``seed`` selects the seeded sampler stream of the camera pass (the reference draws from the global RNG there).
    ``write_frequency`` (sppm.jl:166-171): ``__call__`` stores and saves the film after every iteration it divides and after the
    last one, as the reference does; ``render(on_write=...)`` hands those intermediate images to a callback instead.  Note the
    reference's default of 1: an image per iteration (the library then runs one iteration per batch, about 3x slower)."""

    def __init__(self, camera: PerspectiveCamera, initial_search_radius, max_depth: int, n_iterations: int, photons_per_iteration: int = -1, write_frequency: int = 1,
                 seed: int = 0x5EED0001):
        self.camera = camera
        self.initial_search_radius = f32(initial_search_radius)
        self.max_depth, self.n_iterations = int(max_depth), int(n_iterations)
        crop = camera.film.crop_bounds
        area = int((f32(crop.p_max[0]) - f32(crop.p_min[0])) * (f32(crop.p_max[1]) - f32(crop.p_min[1])))  # area(crop_bounds) bounds.jl:87-90
        self.photons_per_iteration = int(photons_per_iteration) if photons_per_iteration > 0 else area  # :121-124
        self.write_frequency = int(write_frequency)
        self.seed = int(seed)
        self.stats: Optional[_ffi.Stats] = None
        self._ctx = None

    def render(self, scene: Scene, ctx: Optional[_ffi.Context] = None, on_write=None) -> np.ndarray:
        """The film after ``n_iterations``.  ``on_write(iteration, xyzw)``: called with the image of the first ``iteration`` iterations after every iteration
        below the last that ``write_frequency`` divides (sppm.jl:166-171); the array is only valid during the call."""
        flat = scene.flatten(ctx)
        ctx = flat.ctx
        self._ctx = ctx  # (before the call: the periodic-image callback asks it for this process's rank in the job)
        sn = self.camera.sensor()
        st = _ffi.Stats()
        film = self.camera.film
        h, w = film.size
        out = np.empty((h, w, 4), dtype=np.float32)
        if on_write is not None and self.write_frequency > 0:
            failure = []

            def _cb(_user, iteration, ptr):
                try:
                    on_write(int(iteration), np.ctypeslib.as_array(ptr, shape=(h, w, 4)))
                    return 0
                except Exception as e:  # nothing may propagate through the C frames
                    failure.append(e)
                    return 1
            cb = _ffi.SPPM_WRITE_FN(_cb)
            rc = _ffi.lib().trhip_render_sppm_ex(ctx._h, flat._h, C.byref(sn), float(self.initial_search_radius), self.max_depth, self.n_iterations, self.photons_per_iteration, self.seed,
                                                 _ffi.fptr(out), C.byref(st), self.write_frequency, cb, None)
            if failure:
                raise failure[0]
            ctx.check(rc)
        else:
            ctx.check(_ffi.lib().trhip_render_sppm(ctx._h, flat._h, C.byref(sn), float(self.initial_search_radius), self.max_depth, self.n_iterations, self.photons_per_iteration, self.seed,
                                                   _ffi.fptr(out), C.byref(st)))
        self.stats = st
        self._ctx = ctx
        film.set_xyzw(out)  # set_image!(film, image) film.jl:195-202
        film.splat_xyz[...] = 0
        return out

    def state(self) -> dict:
        """SPPMPixel fields after the last render (see trhip_sppm_state)."""
        ctx = self._ctx
        h, w = self.camera.film.size
        out = {"Ld": np.empty((h, w, 3), np.float32), "tau": np.empty((h, w, 3), np.float32), "radius": np.empty((h, w), np.float32), "N": np.empty((h, w), np.float64),
               "M": np.empty((h, w), np.int64), "phi": np.empty((h, w, 3), np.float32), "vp_p": np.empty((h, w, 3), np.float32), "vp_beta": np.empty((h, w, 3), np.float32)}
        info = np.zeros(6, np.int64)
        i64 = C.POINTER(C.c_int64)
        ctx.check(_ffi.lib().trhip_sppm_state(ctx._h, _ffi.fptr(out["Ld"]), _ffi.fptr(out["tau"]), _ffi.fptr(out["radius"]), out["N"].ctypes.data_as(C.POINTER(C.c_double)),
                                              out["M"].ctypes.data_as(i64), _ffi.fptr(out["phi"]), _ffi.fptr(out["vp_p"]), _ffi.fptr(out["vp_beta"]), info.ctypes.data_as(i64)))
        out["info"] = {"grid_res": info[:3].copy(), "grid_entries": int(info[3]), "photon_hits": int(info[4]), "photons_per_iteration": int(info[5])}
        return out

    def __call__(self, scene: Scene):
        film = self.camera.film

        def periodic(iteration, xyzw):  # sppm.jl:166-171: set_image!(film, image); save(film)
            if self._job_rank() != 0:  # in a multi-GPU job every rank holds the whole image after the iteration's all-reduce: rank 0 writes it (as julia/TraceHIP.jl does)
                return
            film.set_xyzw(xyzw.copy())
            film.splat_xyz[...] = 0
            save(film)
        self.render(scene, on_write=periodic if film.filename and 0 < self.write_frequency < self.n_iterations else None)
        if film.filename and self._job_rank() == 0:
            return save(film)
        return None

    def _job_rank(self) -> int:
        """This process's rank in the library's multi-GPU job (0 without a communicator)."""
        return self._ctx.comm_rank()[0] if self._ctx is not None else 0  # (no render yet on a context: nothing to ask)


# ---- model_loader.jl:1-11 without Assimp: a minimal PLY reader ---------------------------------------------------------------------
_PLY_TYPES = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2", "ushort": "u2", "uint16": "u2", "int": "i4", "int32": "i4",
              "uint": "u4", "uint32": "u4", "float": "f4", "float32": "f4", "double": "f8", "float64": "f8"}


def read_ply(path: str):
    """(vertices (n, 3) Float32, normals (n, 3) Float32 or None, faces (m, 3) 0-based UInt32) of an ascii / binary PLY with
    triangle faces (a list property on the face element) — the subset docs/src/assets/models needs."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, elements = None, []
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: truncated PLY header")
            tok = line.decode("ascii", "replace").split()
            if not tok or tok[0] == "comment" or tok[0] == "obj_info":
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                elements.append({"name": tok[1], "count": int(tok[2]), "props": []})
            elif tok[0] == "property":
                if tok[1] == "list":
                    elements[-1]["props"].append(("list", tok[2], tok[3], tok[4]))
                else:
                    elements[-1]["props"].append(("scalar", tok[1], tok[2]))
            elif tok[0] == "end_header":
                break
        if fmt not in ("ascii", "binary_little_endian", "binary_big_endian"):
            raise ValueError(f"{path}: unsupported PLY format {fmt}")
        end = ">" if fmt == "binary_big_endian" else "<"
        verts = normals = faces = None
        for el in elements:
            n, props = el["count"], el["props"]
            if all(p[0] == "scalar" for p in props):
                names = [p[2] for p in props]
                if fmt == "ascii":
                    rows = np.array([f.readline().split() for _ in range(n)], dtype=np.float64).reshape(n, len(props))
                    cols = {nm: rows[:, i] for i, nm in enumerate(names)}
                else:
                    dt = np.dtype([(p[2], end + _PLY_TYPES[p[1]]) for p in props])
                    rec = np.frombuffer(f.read(dt.itemsize * n), dtype=dt, count=n)
                    cols = {nm: rec[nm] for nm in names}
                if el["name"] == "vertex":
                    verts = np.stack([cols["x"], cols["y"], cols["z"]], axis=1).astype(np.float32)
                    if all(k in cols for k in ("nx", "ny", "nz")):
                        normals = np.stack([cols["nx"], cols["ny"], cols["nz"]], axis=1).astype(np.float32)
            else:
                if len(props) != 1:
                    raise ValueError(f"{path}: element {el['name']}: only a single list property is supported")
                _, ct, it, _ = props[0]
                if fmt == "ascii":
                    rows = [f.readline().split() for _ in range(n)]
                    if any(int(r[0]) != 3 for r in rows):
                        raise ValueError("Only triangles supported.")  # model_loader.jl:29
                    idx = np.array([r[1:4] for r in rows], dtype=np.int64).reshape(n, 3)
                else:
                    dt = np.dtype([("n", end + _PLY_TYPES[ct]), ("i", end + _PLY_TYPES[it], (3,))])
                    rec = np.frombuffer(f.read(dt.itemsize * n), dtype=dt, count=n)
                    if n and np.any(rec["n"] != 3):
                        raise ValueError("Only triangles supported.")
                    idx = rec["i"].astype(np.int64)
                if el["name"] == "face":
                    faces = idx.astype(np.uint32)
        if verts is None or faces is None:
            raise ValueError(f"{path}: no vertex / face element")
        if faces.size and int(faces.max()) >= verts.shape[0]:
            raise ValueError(f"{path}: face index out of range")
        return verts, normals, faces


def load_triangle_mesh(model_file: str, core: Optional[ShapeCore] = None):
    """model_loader.jl:1-11: (triangle_meshes, triangles).  The reference goes through Assimp; PLY is read directly here.
    Like the reference it requires one normal per vertex (model_loader.jl:30)."""
    core = core or ShapeCore(Transformation(), False)
    verts, normals, faces = read_ply(model_file)
    if normals is None or normals.shape[0] != verts.shape[0]:
        raise ValueError("Number of normals is different from the number of vertices")
    indices = (faces.reshape(-1) + 1).astype(np.uint32)  # 1-based (model_loader.jl:36)
    triangles = create_triangle_mesh(core, faces.shape[0], indices, verts.shape[0], verts, normals)
    return [triangles[0].mesh] if triangles else [], triangles
