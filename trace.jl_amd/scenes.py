"""Workload definitions: the reference's *shadows* scene transcribed as data, the synthetic Cornell box,
synthetic N-triangle height-field meshes (the Stanford Dragon is not available offline), and random-ray generators.
Geometry comes from integer hashes so that every host reproduces it bit-for-bit.
"""
from __future__ import annotations

import numpy as np

from . import api as T

f32 = np.float32
_GOLDEN = np.uint64(0x9E3779B97F4A7C15)


# ---- the seeded sampler of include/trace_sampler.h, vectorised (host-side input generation only) --------------------------
def ts_mix64(z: np.ndarray) -> np.ndarray:
    z = np.asarray(z, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = z ^ (z >> np.uint64(30))
        z = z * np.uint64(0xBF58476D1CE4E5B9)
        z = z ^ (z >> np.uint64(27))
        z = z * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def ts_stream_key(seed: int, px, py, sample) -> np.ndarray:
    px = np.asarray(px, dtype=np.int64).astype(np.uint64) & np.uint64(0xFFFFFFFF)
    py = np.asarray(py, dtype=np.int64).astype(np.uint64) & np.uint64(0xFFFFFFFF)
    pix = px | (py << np.uint64(32))
    with np.errstate(over="ignore"):
        return ts_mix64(ts_mix64(np.uint64(seed) ^ pix) + np.asarray(sample, dtype=np.uint64) * _GOLDEN)


def ts_uniform(key: np.ndarray, dim: int) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = ts_mix64(np.asarray(key, dtype=np.uint64) + np.uint64(dim + 1) * _GOLDEN)
    return (z >> np.uint64(40)).astype(np.float32) * f32(2.0 ** -24)


# ---- docs/src/shadows.md:8-107 ---------------------------------------------------------------------------------------------------
def shadows_scene():
    """Materials, 4 spheres, 4 triangles, 1 point light — literal transcription of docs/src/shadows.md:10-94."""
    red = T.MatteMaterial(T.ConstantTexture(T.RGBSpectrum(0.796, 0.235, 0.2)), T.ConstantTexture(0.0))
    blue = T.MatteMaterial(T.ConstantTexture(T.RGBSpectrum(0.251, 0.388, 0.847)), T.ConstantTexture(0.0))
    white = T.MatteMaterial(T.ConstantTexture(T.RGBSpectrum(1.0)), T.ConstantTexture(0.0))
    mirror = T.MirrorMaterial(T.ConstantTexture(T.RGBSpectrum(1.0)))
    glass = T.GlassMaterial(T.ConstantTexture(T.RGBSpectrum(1.0)), T.ConstantTexture(T.RGBSpectrum(1.0)), T.ConstantTexture(0.0), T.ConstantTexture(0.0),
                            T.ConstantTexture(1.5), True)
    p1 = T.GeometricPrimitive(T.Sphere(T.ShapeCore(T.translate([0.3, 0.11, -2.2]), False), 0.1, 360.0), glass)
    p2 = T.GeometricPrimitive(T.Sphere(T.ShapeCore(T.translate([0.2, 0.11, -2.6]), False), 0.1, 360.0), blue)
    p3 = T.GeometricPrimitive(T.Sphere(T.ShapeCore(T.translate([0.7, 0.31, -2.8]), False), 0.3, 360.0), mirror)
    p4 = T.GeometricPrimitive(T.Sphere(T.ShapeCore(T.translate([0.7, 0.11, -2.3]), False), 0.1, 360.0), red)
    tris = T.create_triangle_mesh(
        T.ShapeCore(T.translate([0, 0, -2]), False), 4, np.array([1, 2, 3, 1, 4, 3, 2, 3, 5, 6, 5, 3], dtype=np.uint32), 6,
        [[0, 0, 0], [0, 0, -1], [1, 0, -1], [1, 0, 0], [0, 1, -1], [1, 1, -1]],
        [[0, 1, 0], [0, 1, 0], [0, 1, 0], [0, 1, 0], [0, 0, 1], [0, 0, 1]])
    t1, t2 = T.GeometricPrimitive(tris[0], mirror), T.GeometricPrimitive(tris[1], mirror)
    t3, t4 = T.GeometricPrimitive(tris[2], white), T.GeometricPrimitive(tris[3], white)
    bvh = T.BVHAccel([p1, p2, p3, p4, t1, t2, t3, t4], 1)
    lights = [T.PointLight(T.translate([-1, 1, 0]), T.RGBSpectrum(25.0))]
    return T.Scene(lights, bvh)


def shadows_camera(resolution: int = 341, filename: str = ""):
    """docs/src/shadows.md:96-104 (there: resolution 1024 ÷ 3 = 341)."""
    flt = T.LanczosSincFilter([1.0, 1.0], 3.0)
    film = T.Film([resolution, resolution], T.Bounds2([0.0, 0.0], [1.0, 1.0]), flt, 1.0, 1.0, filename)
    screen = T.Bounds2([-1.0, -1.0], [1.0, 1.0])
    return T.PerspectiveCamera(T.look_at([0, 15, 50], [0, 0, -2], [0, 1, 0]), screen, 0.0, 1.0, 0.0, 1e6, 90.0, film)


# ---- S-cornell ---------------------------------------------------------------------------------------------------
def _quad(core, p0, p1, p2, p3, normal):
    return T.create_triangle_mesh(core, 2, np.array([1, 2, 3, 1, 3, 4], dtype=np.uint32), 4, [p0, p1, p2, p3], [normal] * 4)


def cornell_primitives(spheres: bool = True):
    red = T.MatteMaterial(T.ConstantTexture(T.RGBSpectrum(0.796, 0.235, 0.2)), T.ConstantTexture(0.0))
    blue = T.MatteMaterial(T.ConstantTexture(T.RGBSpectrum(0.251, 0.388, 0.847)), T.ConstantTexture(0.0))
    white = T.MatteMaterial(T.ConstantTexture(T.RGBSpectrum(1.0)), T.ConstantTexture(0.0))
    mirror = T.MirrorMaterial(T.ConstantTexture(T.RGBSpectrum(1.0)))
    glass = T.GlassMaterial(T.ConstantTexture(T.RGBSpectrum(1.0)), T.ConstantTexture(T.RGBSpectrum(1.0)), T.ConstantTexture(0.0), T.ConstantTexture(0.0),
                            T.ConstantTexture(1.5), True)
    core = T.ShapeCore(T.translate([0, 0, 0]), False)
    prims = []
    walls = [
        (([0, 0, -2], [1, 0, -2], [1, 0, -3], [0, 0, -3]), [0, 1, 0], white),   # floor
        (([0, 1, -2], [0, 1, -3], [1, 1, -3], [1, 1, -2]), [0, -1, 0], white),  # ceiling
        (([0, 0, -3], [1, 0, -3], [1, 1, -3], [0, 1, -3]), [0, 0, 1], white),   # back wall
        (([0, 0, -2], [0, 0, -3], [0, 1, -3], [0, 1, -2]), [1, 0, 0], red),     # left wall
        (([1, 0, -2], [1, 1, -2], [1, 1, -3], [1, 0, -3]), [-1, 0, 0], blue),   # right wall
    ]
    for (p0, p1, p2, p3), n, m in walls:
        for t in _quad(core, p0, p1, p2, p3, n):
            prims.append(T.GeometricPrimitive(t, m))
    if spheres:
        prims.append(T.GeometricPrimitive(T.Sphere(T.ShapeCore(T.translate([0.3, 0.25, -2.7]), False), 0.25, 360.0), mirror))
        prims.append(T.GeometricPrimitive(T.Sphere(T.ShapeCore(T.translate([0.7, 0.2, -2.35]), False), 0.2, 360.0), glass))
    return prims, white


def cornell_lights():
    return [T.PointLight(T.translate([0.5, 0.9, -2.5]), T.RGBSpectrum(2.5))]


def cornell_scene(spheres: bool = True):
    prims, _ = cornell_primitives(spheres)
    return T.Scene(cornell_lights(), T.BVHAccel(prims, 1))


def cornell_camera(resolution: int = 1024, filename: str = ""):
    """Same camera convention as the shadows scene (far camera + the reference's load-bearing projection matrices):
    the raster maps to the window x in [0, 1.10], y in [-0.29, 0.85] at the back wall for 1024²."""
    return shadows_camera(resolution, filename)


# ---- S-mesh-N: height field of n x n quads over the Cornell floor ----------------------------------------------------------------
def _hash32(x: np.ndarray) -> np.ndarray:
    x = np.asarray(x, dtype=np.uint32)
    with np.errstate(over="ignore"):
        x = (x ^ (x >> np.uint32(16))) * np.uint32(0x7FEB352D)
        x = (x ^ (x >> np.uint32(15))) * np.uint32(0x846CA68B)
        x = x ^ (x >> np.uint32(16))
    return x


def value_noise(x: np.ndarray, y: np.ndarray, seed: int = 1234, lattice: int = 64) -> np.ndarray:
    """Bilinear value noise on a lattice x lattice grid of 32-bit-hash values in [0, 1); x, y in [0, 1]."""
    gx, gy = x.astype(np.float64) * lattice, y.astype(np.float64) * lattice
    ix, iy = np.floor(gx).astype(np.int64), np.floor(gy).astype(np.int64)
    fx, fy = gx - ix, gy - iy

    def lat(i, j):
        h = _hash32((np.uint32(seed) + (i % (lattice + 1)).astype(np.uint32) * np.uint32(0x9E3779B1) + (j % (lattice + 1)).astype(np.uint32) * np.uint32(0x85EBCA77)).astype(np.uint32))
        return h.astype(np.float64) / 4294967296.0

    v = (lat(ix, iy) * (1 - fx) + lat(ix + 1, iy) * fx) * (1 - fy) + (lat(ix, iy + 1) * (1 - fx) + lat(ix + 1, iy + 1) * fx) * fy
    return v.astype(np.float32)


def heightfield_mesh(n: int, seed: int = 1234, amplitude: float = 0.15):
    """2 n² triangles over x in [0,1], z in [-3,-2]; per-vertex normals by central differences.  Returns (vertices, indices(1-based), normals)."""
    g = np.linspace(0.0, 1.0, n + 1, dtype=np.float64)
    X, Z = np.meshgrid(g, g, indexing="xy")  # X varies along columns
    H = amplitude * value_noise(X.ravel().astype(np.float32), Z.ravel().astype(np.float32), seed).reshape(n + 1, n + 1).astype(np.float64)
    verts = np.stack([X, H, -2.0 - Z], axis=-1).reshape(-1, 3).astype(np.float32)
    step = 1.0 / n
    dhdx = np.zeros_like(H)
    dhdz = np.zeros_like(H)
    dhdx[:, 1:-1] = (H[:, 2:] - H[:, :-2]) / (2 * step)
    dhdx[:, 0] = (H[:, 1] - H[:, 0]) / step
    dhdx[:, -1] = (H[:, -1] - H[:, -2]) / step
    dhdz[1:-1, :] = (H[2:, :] - H[:-2, :]) / (2 * step)
    dhdz[0, :] = (H[1, :] - H[0, :]) / step
    dhdz[-1, :] = (H[-1, :] - H[-2, :]) / step
    # surface y = h(x, g) with world z = -2 - g: normal ∝ (-dh/dx, 1, +dh/dg)
    nrm = np.stack([-dhdx, np.ones_like(H), dhdz], axis=-1)
    nrm /= np.linalg.norm(nrm, axis=-1, keepdims=True)
    nrm = nrm.reshape(-1, 3).astype(np.float32)
    i, j = np.meshgrid(np.arange(n), np.arange(n), indexing="xy")
    v00 = (j * (n + 1) + i).ravel()
    v10, v01, v11 = v00 + 1, v00 + (n + 1), v00 + (n + 2)
    # winding chosen so that (v1 - v3) x (v2 - v3) (triangle_mesh.jl:230) points up (+y)
    tris = np.concatenate([np.stack([v00, v10, v11], axis=1), np.stack([v00, v11, v01], axis=1)], axis=0)
    return verts, (tris + 1).astype(np.uint32).reshape(-1), nrm


def mesh_scene(n: int, seed: int = 1234):
    """S-mesh-N: the Cornell box with an n x n-quad height field on the floor (2 n² + 12 primitives); matte 0.8."""
    prims, _ = cornell_primitives()
    grey = T.MatteMaterial(T.ConstantTexture(T.RGBSpectrum(0.8)), T.ConstantTexture(0.0))
    verts, idx, nrm = heightfield_mesh(n, seed)
    prims = prims + [T.create_mesh_primitives(T.ShapeCore(T.translate([0, 0, 0]), False), idx, verts, nrm, grey)]
    return T.Scene(cornell_lights(), T.BVHAccel(prims, 1))


MESH_N = {"mesh_tiny": 16, "mesh_64k": 181, "mesh_870k": 660, "mesh_1m": 724, "mesh_10m": 2290}


# ---- ray generators (SURVEY.md §8d "random-ray micro-benchmarks") ---------------------------------------------------------------
def incoherent_rays(n: int, bound_min, bound_max, seed: int = 0x5EED0002) -> np.ndarray:
    """Origins uniform in the bound, directions uniform on the sphere, t_max = Inf; n x 8 Float32."""
    idx = np.arange(n, dtype=np.uint64)
    key = ts_stream_key(seed, (idx & np.uint64(0xFFFF)).astype(np.int64), (idx >> np.uint64(16)).astype(np.int64), 0)
    u = [ts_uniform(key, d) for d in range(5)]
    lo, hi = np.asarray(bound_min, dtype=np.float32), np.asarray(bound_max, dtype=np.float32)
    o = np.stack([lo[k] + (hi[k] - lo[k]) * u[k] for k in range(3)], axis=1)
    z = f32(1.0) - f32(2.0) * u[3]
    r = np.sqrt(np.maximum(f32(0.0), f32(1.0) - z * z))
    phi = f32(2.0 * np.pi) * u[4]
    d = np.stack([r * np.cos(phi), r * np.sin(phi), z], axis=1).astype(np.float32)
    rays = np.empty((n, 8), dtype=np.float32)
    rays[:, 0:3] = o
    rays[:, 3] = np.inf
    rays[:, 4:7] = d
    rays[:, 7] = 0.0
    return rays


def camera_sample_grid(camera, spp: int = 1, seed: int = 0x5EED0001) -> np.ndarray:
    """Camera samples (film.x, film.y, lens.x, lens.y, time) of every sample-pixel, sample-major — what k_raygen draws."""
    sb = camera.film.get_sample_bounds()
    xs = np.arange(int(sb.p_min[0]), int(sb.p_max[0]) + 1)
    ys = np.arange(int(sb.p_min[1]), int(sb.p_max[1]) + 1)
    X, Y = np.meshgrid(xs, ys, indexing="xy")
    out = []
    for s in range(spp):
        key = ts_stream_key(seed, X.ravel(), Y.ravel(), s)
        c = np.stack([X.ravel().astype(np.float32) + ts_uniform(key, 0), Y.ravel().astype(np.float32) + ts_uniform(key, 1), ts_uniform(key, 2), ts_uniform(key, 3),
                      ts_uniform(key, 4)], axis=1)
        out.append(c)
    return np.concatenate(out, axis=0).astype(np.float32)


# ---- S-caustic: docs/code/caustic_glass.jl with a procedural glass (the reference's PLY cannot travel) ------------------------------
def goblet_mesh(n_theta: int = 256, n_profile: int = 172):
    """A glass goblet as a surface of revolution (outer wall up to the rim, inner wall back down): ~2 * n_theta * n_profile
    triangles with smooth vertex normals, inside the bounding box of docs/src/assets/models/caustic-glass.ply after its
    translate (x 0.2..2.3, y 0..2, z -98.6..-96.5).  Returns (vertices (n, 3), 1-based indices, normals)."""
    prof = np.array([[0.0, 0.0], [0.62, 0.0], [0.62, 0.05], [0.09, 0.13], [0.07, 0.85], [0.45, 1.05], [0.88, 1.55], [0.84, 2.0],   # outer, bottom to rim
                     [0.79, 2.0], [0.83, 1.56], [0.42, 1.1], [0.0, 0.98]], dtype=np.float64)                                       # inner, rim to bottom
    seg = np.sqrt(((prof[1:] - prof[:-1]) ** 2).sum(1))
    s = np.concatenate([[0.0], np.cumsum(seg)])
    t = np.linspace(0.0, s[-1], n_profile + 1)
    r = np.interp(t, s, prof[:, 0])
    y = np.interp(t, s, prof[:, 1])
    r[0] = r[-1] = 0.0
    th = np.arange(n_theta, dtype=np.float64) * (2.0 * np.pi / n_theta)
    cx, cz = 1.27, -97.57
    vx = cx + r[:, None] * np.cos(th)[None, :]
    vz = cz + r[:, None] * np.sin(th)[None, :]
    vy = 0.01 + np.repeat(y[:, None], n_theta, axis=1)
    verts = np.stack([vx, vy, vz], axis=-1).reshape(-1, 3).astype(np.float32)
    i, j = np.meshgrid(np.arange(n_profile), np.arange(n_theta), indexing="ij")
    a = i * n_theta + j
    b = i * n_theta + (j + 1) % n_theta
    c = (i + 1) * n_theta + (j + 1) % n_theta
    d = (i + 1) * n_theta + j
    t1 = np.stack([a, d, c], axis=-1).reshape(-1, 3)
    t2 = np.stack([a, c, b], axis=-1).reshape(-1, 3)
    tris = np.concatenate([t1, t2], axis=0)
    p = verts.astype(np.float64)
    fn = np.cross(p[tris[:, 1]] - p[tris[:, 0]], p[tris[:, 2]] - p[tris[:, 0]])
    keep = np.sqrt((fn ** 2).sum(1)) > 1e-12  # the rings on the axis collapse one triangle of every quad
    tris, fn = tris[keep], fn[keep]
    nrm = np.zeros_like(p)
    for k in range(3):
        np.add.at(nrm, tris[:, k], fn)
    ln = np.sqrt((nrm ** 2).sum(1, keepdims=True))
    nrm = np.where(ln > 0, nrm / np.maximum(ln, 1e-30), np.array([0.0, 1.0, 0.0]))
    return verts, (tris + 1).astype(np.uint32).reshape(-1), nrm.astype(np.float32)


def caustic_scene(model: str = "", n_theta: int = 256, n_profile: int = 172):
    """docs/code/caustic_glass.jl:5-79: glass object on a plastic floor under a SpotLight.  ``model``: path of a PLY to load
    with load_triangle_mesh (the reference's caustic-glass.ply); empty = the procedural goblet of the same size."""
    glass = T.GlassMaterial(T.ConstantTexture(T.RGBSpectrum(1.0)), T.ConstantTexture(T.RGBSpectrum(1.0)), T.ConstantTexture(0.0), T.ConstantTexture(0.0),
                            T.ConstantTexture(1.25), True)
    plastic = T.PlasticMaterial(T.ConstantTexture(T.RGBSpectrum(0.6399999857, 0.6399999857, 0.6399999857)),
                                T.ConstantTexture(T.RGBSpectrum(0.1000000015, 0.1000000015, 0.1000000015)), T.ConstantTexture(0.010408001), True)
    prims = []
    if model:
        _, triangles = T.load_triangle_mesh(model, T.ShapeCore(T.translate([5, -1.49, -100]), False))
        mesh = triangles[0].mesh
        prims.append(T.create_mesh_primitives(T.ShapeCore(T.translate([0, 0, 0]), False), mesh.indices, mesh.vertices, mesh.normals, glass))
    else:
        verts, idx, nrm = goblet_mesh(n_theta, n_profile)
        prims.append(T.create_mesh_primitives(T.ShapeCore(T.translate([0, 0, 0]), False), idx, verts, nrm, glass))
    floor = T.create_triangle_mesh(T.ShapeCore(T.translate([-10, 0, -87]), False), 2, np.array([1, 2, 3, 1, 4, 3], dtype=np.uint32), 4,
                                   [[0, 0, 0], [0, 0, -30], [30, 0, -30], [30, 0, 0]], [[0, 1, 0]] * 4)
    prims += [T.GeometricPrimitive(t, plastic) for t in floor]
    frm, to = np.float32([0, 2, 0]), np.float32([-5, 0, 5])
    d = (to - frm) / np.float32(np.sqrt(np.float32(((to - frm) ** 2).sum())))
    d, du, dv = T.coordinate_system(d)
    m = np.eye(4, dtype=np.float32)
    m[0, :3], m[1, :3], m[2, :3] = du, dv, d
    light_to_world = T.translate([4.5, 0, -101]) * T.translate(frm) * T.inv(T.Transformation(m))
    lights = [T.SpotLight(light_to_world, T.RGBSpectrum(60.0), 30.0, 20.0)]
    return T.Scene(lights, T.BVHAccel(prims, 1))


def caustic_camera(resolution: int = 1024, filename: str = ""):
    """docs/code/caustic_glass.jl:81-100."""
    film = T.Film([resolution, resolution], T.Bounds2(np.float32([0, 0]), np.float32([1, 1])), T.LanczosSincFilter([1.0, 1.0], 3.0), 1.0, 1.0, filename)
    return T.PerspectiveCamera(T.look_at([0, 150, 150], [-3, 0, -91], [0, 1, 0]), T.Bounds2(np.float32([-1, -1]), np.float32([1, 1])), 0.0, 1.0, 0.0, 1e6, 90.0, film)


# ---- S-blob: a closed, bumpy object in the Cornell box (a stand-in for BASELINE.json configs[2], "Stanford Dragon in Cornell box") ----
def blob_mesh(n: int, seed: int = 4321, amplitude: float = 0.05):
    """A cube-sphere (6 faces of n x n quads, so the triangles are near-uniform) of radius 0.28 at (0.5, 0.4, -2.5), displaced
    along the normal by a smooth function of the direction: 12 n^2 triangles, smooth vertex normals, shared vertices along the
    cube edges (closed surface).  Returns (vertices (m, 3), 1-based indices, normals)."""
    t = np.linspace(-1.0, 1.0, n + 1)
    a, b = np.meshgrid(t, t, indexing="ij")
    one = np.ones_like(a)
    faces = [np.stack(c, -1) for c in ((one, a, b), (-one, b, a), (b, one, a), (a, -one, b), (a, b, one), (b, a, -one))]
    pts = np.concatenate([f.reshape(-1, 3) for f in faces], axis=0)
    # weld the duplicated edge / corner vertices: key on the exactly representable lattice coordinates
    key = np.round((pts + 1.0) * (n / 2.0)).astype(np.int64)
    uniq, inv = np.unique(key[:, 0] * (n + 1) ** 2 + key[:, 1] * (n + 1) + key[:, 2], return_inverse=True)
    first = np.zeros(uniq.size, dtype=np.int64)
    first[inv[::-1]] = np.arange(pts.shape[0])[::-1]
    cube = pts[first]
    dirs = cube / np.sqrt((cube ** 2).sum(1, keepdims=True))
    x, y, z = dirs[:, 0], dirs[:, 1], dirs[:, 2]
    rng = np.random.default_rng(seed)
    bump = np.zeros_like(x)
    for k in range(24):  # a few dozen smooth lobes: folds and creases like a scanned figure, no slivers
        w = rng.normal(size=3)
        w /= np.linalg.norm(w)
        f = rng.uniform(3.0, 14.0)
        bump += rng.uniform(0.3, 1.0) * np.sin(f * (x * w[0] + y * w[1] + z * w[2]) + rng.uniform(0, 6.28))
    r = 0.28 + amplitude * bump / 6.0
    verts = (np.array([0.5, 0.4, -2.5]) + r[:, None] * dirs).astype(np.float32)
    i, j = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    tris = []
    for f in range(6):
        base = f * (n + 1) ** 2
        v00 = inv[base + i * (n + 1) + j]
        v10 = inv[base + (i + 1) * (n + 1) + j]
        v11 = inv[base + (i + 1) * (n + 1) + j + 1]
        v01 = inv[base + i * (n + 1) + j + 1]
        tris.append(np.stack([v00, v10, v11], -1).reshape(-1, 3))
        tris.append(np.stack([v00, v11, v01], -1).reshape(-1, 3))
    tris = np.concatenate(tris, axis=0)
    pd = verts.astype(np.float64)
    fn = np.cross(pd[tris[:, 1]] - pd[tris[:, 0]], pd[tris[:, 2]] - pd[tris[:, 0]])
    out = (fn * (pd[tris[:, 0]] - np.array([0.5, 0.4, -2.5]))).sum(1) < 0  # orient every triangle outwards
    tris[out] = tris[out][:, [0, 2, 1]]
    fn[out] = -fn[out]
    nrm = np.zeros_like(pd)
    for k in range(3):
        np.add.at(nrm, tris[:, k], fn)
    nrm /= np.maximum(np.sqrt((nrm ** 2).sum(1, keepdims=True)), 1e-30)
    return verts, (tris + 1).astype(np.uint32).reshape(-1), nrm.astype(np.float32)


def blob_scene(n_lat: int = 270):
    """S-blob: the Cornell walls + one closed bumpy matte object of 12 n² triangles (270 -> 874 800)."""
    prims, _ = cornell_primitives(spheres=False)
    grey = T.MatteMaterial(T.ConstantTexture(T.RGBSpectrum(0.75, 0.7, 0.6)), T.ConstantTexture(0.0))
    verts, idx, nrm = blob_mesh(n_lat)
    prims = prims + [T.create_mesh_primitives(T.ShapeCore(T.translate([0, 0, 0]), False), idx, verts, nrm, grey)]
    return T.Scene(cornell_lights(), T.BVHAccel(prims, 1))
