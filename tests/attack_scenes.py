"""Scene and ray families that ATTACK the hybrid mode's order-independence certificate (csrc/th_trace3c.h) — VERDICT r4 weak #5 / next #4.

The certificate is a floating-point argument with margins that scale with D (the ray's reach: largest coordinate offset between its origin and the scene bound), with the
largest leaf extent along the dominant axis, and with L³ / 2A of flat triangles.  Round 4's soak ran at Cornell scale around the origin only.  A margin that is too small
gives a silent wrong hit, not a crash — so the families below push where the margins are thinnest:

  scaled / moved   the same scene at 1e-3, 1e2, 1e4 times its size and 1e3 … 1e5 away from the origin (coordinates lose their low bits; reach and extent terms move apart)
  coplanar         duplicate and mirrored triangles in one plane, rays through their shared edges and vertices (ties at every hit)
  slivers          long NON-flat needles whose extent along the dominant axis is the scene's size (the mle_small term), plus flat ones (the sq_flat term)
  tiny directions  direction components of 1e-30 … 1e-6 (|1 / d| up to 1e30: the growth term against kCertCap)
  far origins      rays that start 1e6 scene sizes away
  tiny scenes      <= tiny_scene_prims primitives: the one-leaf accelerator (k_trace_leaf_c), the same stresses

Used by tests/test_gpu_certificate_attack.py (-m gpu) and tools/soak_attack.py (the long run, logs under profiles/).
"""
import numpy as np

f32 = np.float32


def _white(T):
    return T.MatteMaterial(T.ConstantTexture(T.RGBSpectrum(0.8)), T.ConstantTexture(0.0))


def mesh_prims(T, verts, mat=None):
    """One mesh from an (n, 3, 3) vertex array."""
    v = np.ascontiguousarray(verts, f32).reshape(-1, 3)
    core = T.ShapeCore(T.translate([0, 0, 0]), False)
    return T.create_mesh_primitives(core, np.arange(v.shape[0], dtype=np.uint32) + 1, v, None, mat or _white(T))


def box_tris(lo, hi):
    """12 triangles of the axis-aligned box [lo, hi] (flat leaves of any size: walls)."""
    lo, hi = np.asarray(lo, np.float64), np.asarray(hi, np.float64)
    c = np.array([[lo[0] if (k & 1) == 0 else hi[0], lo[1] if (k & 2) == 0 else hi[1], lo[2] if (k & 4) == 0 else hi[2]] for k in range(8)])
    quads = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]
    t = []
    for a, b, c_, d in quads:
        t.append([c[a], c[b], c[c_]])
        t.append([c[a], c[c_], c[d]])
    return np.asarray(t)


def random_tris(rng, n, lo, hi, size):
    c = lo + (hi - lo) * rng.random((n, 1, 3))
    e = rng.standard_normal((n, 3, 3)) * (size * rng.random((n, 1, 1)) ** 2 + 1e-3 * size)
    return c + e


def base_geometry(rng, n_tris, kind):
    """Unit-scale geometry in [0, 1]^3: a closed box + content.  Returns (triangles (n, 3, 3) float64, spheres [(centre, radius)])."""
    tris = [box_tris([0, 0, 0], [1, 1, 1])]
    spheres = []
    if kind == "mixed":
        tris.append(random_tris(rng, n_tris, 0.05, 0.95, 0.08))
        spheres = [((0.3, 0.3, 0.35), 0.18), ((0.7, 0.25, 0.6), 0.12)]
    elif kind == "coplanar":
        # stacks of duplicate / mirrored triangles in shared planes: a fan around shared vertices, each triangle present twice (once mirrored)
        for plane in range(6):
            z = 0.15 + 0.12 * plane
            ctr = np.array([0.5, 0.5, z])
            m = max(3, n_tris // 24)
            ang = np.sort(rng.random(m) * 2 * np.pi)
            ring = ctr + 0.4 * np.stack([np.cos(ang), np.sin(ang), np.zeros(m)], axis=1)
            fan = np.stack([np.tile(ctr, (m, 1)), ring, np.roll(ring, -1, axis=0)], axis=1)
            tris += [fan, fan[:, [0, 2, 1]], fan.copy()]  # the fan, its mirror image (reversed winding), a duplicate
        # … and tilted shared planes (not axis-aligned: non-flat leaves with ties)
        nrm = np.array([0.3, 0.5, 0.81])
        u = np.cross(nrm, [1, 0, 0])
        u /= np.linalg.norm(u)
        v = np.cross(nrm, u)
        q = rng.random((n_tris // 4 + 3, 3, 2)) * 0.5 - 0.25
        pl = np.array([0.5, 0.5, 0.5]) + q[..., :1] * u + q[..., 1:] * v
        tris += [pl, pl[:, [1, 0, 2]]]
    elif kind == "slivers":
        n = max(8, n_tris // 2)
        a = 0.02 + 0.96 * rng.random((n, 3))
        b = 0.02 + 0.96 * rng.random((n, 3))  # the far end: anywhere in the box — extents of the order of the scene along every axis
        w = rng.standard_normal((n, 3)) * (10.0 ** rng.uniform(-6, -2.5, (n, 1)))  # width 1e-6 … 3e-3
        tris.append(np.stack([a, b, a + w], axis=1))
        # flat needles in axis-aligned planes (the sq_flat term: L³ / 2A large)
        a2 = 0.02 + 0.96 * rng.random((n, 3))
        b2 = 0.02 + 0.96 * rng.random((n, 3))
        ax = rng.integers(0, 3, n)
        b2[np.arange(n), ax] = a2[np.arange(n), ax]
        w2 = rng.standard_normal((n, 3)) * (10.0 ** rng.uniform(-5, -2.5, (n, 1)))
        w2[np.arange(n), ax] = 0.0
        tris.append(np.stack([a2, b2, a2 + w2], axis=1))
        tris.append(random_tris(rng, n_tris // 4 + 1, 0.05, 0.95, 0.05))
        spheres = [((0.5, 0.5, 0.5), 0.1)]
    else:
        raise ValueError(kind)
    return np.concatenate(tris), spheres


def build_scene(T, tris, spheres, scale=1.0, shift=(0.0, 0.0, 0.0), glass=True):
    """The geometry scaled and moved (in Float64, then rounded once to Float32: the scene IS what Float32 holds), spheres through their own transforms."""
    shift = np.asarray(shift, np.float64)
    v = (np.asarray(tris, np.float64) * scale + shift).astype(f32)
    prims = [mesh_prims(T, v)]
    mats = [T.GlassMaterial(T.ConstantTexture(T.RGBSpectrum(1.0)), T.ConstantTexture(T.RGBSpectrum(1.0)), T.ConstantTexture(0.0), T.ConstantTexture(0.0), T.ConstantTexture(1.5), True),
            T.MirrorMaterial(T.ConstantTexture(T.RGBSpectrum(0.9)))]
    for k, (c, r) in enumerate(spheres):
        cc = np.asarray(c, np.float64) * scale + shift
        core = T.ShapeCore(T.translate([float(cc[0]), float(cc[1]), float(cc[2])]), False)
        prims.append(T.GeometricPrimitive(T.Sphere(core, float(r * scale), 360.0), mats[k % 2] if glass else _white(T)))
    lp = np.asarray([0.5, 0.9, 0.5]) * scale + shift
    lights = [T.PointLight(T.translate([float(lp[0]), float(lp[1]), float(lp[2])]), T.RGBSpectrum(float(2.0 * scale * scale)))]
    lo = (np.zeros(3) * scale + shift).astype(f32)
    hi = (np.ones(3) * scale + shift).astype(f32)
    return T.Scene(lights, T.BVHAccel(prims, 1)), v.reshape(-1, 3, 3), lo, hi


def rays8(o, d, tmax=np.inf):
    r = np.empty((o.shape[0], 8), f32)
    r[:, 0:3], r[:, 3], r[:, 4:7], r[:, 7] = o, tmax, d, 0.0
    return r


def attack_rays(rng, n, lo, hi, tri, spheres_world):
    """The ray families, about n rays in total.  `tri`: the scene's triangles (n, 3, 3) Float32, world space."""
    lo64, hi64 = np.asarray(lo, np.float64), np.asarray(hi, np.float64)
    size = float(np.max(hi64 - lo64))
    m = max(64, n // 10)

    def inside(k):
        return (lo64 + (hi64 - lo64) * rng.random((k, 3))).astype(f32)

    def unit(k):
        d = rng.standard_normal((k, 3))
        return (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(f32)

    parts = {}
    parts["uniform"] = rays8(inside(2 * m), unit(2 * m))
    # through shared edges and vertices: aim at a vertex / a point on an edge from a random origin, and start ON surfaces (points of triangles) in random directions
    k = rng.integers(0, tri.shape[0], m)
    vtx = tri[k, rng.integers(0, 3, m)].astype(np.float64)
    o = inside(m)
    parts["through vertices"] = rays8(o, (vtx - o).astype(f32))
    a, b = tri[k, 0].astype(np.float64), tri[k, 1].astype(np.float64)
    s = rng.random((m, 1))
    edge = a + s * (b - a)
    o = inside(m)
    parts["through edges"] = rays8(o, (edge - o).astype(f32))
    bc = rng.dirichlet([1, 1, 1], m)
    surf = (tri[k].astype(np.float64) * bc[:, :, None]).sum(axis=1)
    parts["from surfaces"] = rays8(surf.astype(f32), unit(m))
    # in-plane rays: along a triangle's own plane (grazing its box and its neighbours')
    e1 = (tri[k, 1] - tri[k, 0]).astype(np.float64)
    e2 = (tri[k, 2] - tri[k, 0]).astype(np.float64)
    w = rng.standard_normal((m, 2))
    dpl = w[:, :1] * e1 + w[:, 1:] * e2
    parts["in-plane"] = rays8((surf - 0.3 * size * dpl / (np.linalg.norm(dpl, axis=1, keepdims=True) + 1e-300)).astype(f32), dpl.astype(f32))
    # tiny direction components: 1e-30 … 1e-6 of the largest, one or two of them
    d = unit(m).astype(np.float64)
    tiny = 10.0 ** rng.uniform(-30, -6, (m, 3)) * np.sign(rng.standard_normal((m, 3)))
    which = rng.integers(0, 3, m)
    d[np.arange(m), which] = tiny[np.arange(m), which]
    two = rng.random(m) < 0.3
    w2 = (which + 1) % 3
    d[two, w2[two]] = tiny[two, w2[two]]
    parts["tiny direction components"] = rays8(inside(m), d.astype(f32))
    # far origins: 1e2, 1e4, 1e6 scene sizes away, aimed into the scene
    tgt = inside(m).astype(np.float64)
    dist = size * 10.0 ** rng.choice([2.0, 4.0, 6.0], (m, 1))
    o = tgt - unit(m).astype(np.float64) * dist
    parts["far origins"] = rays8(o.astype(f32), (tgt - o).astype(f32))
    # inside / on the spheres
    if spheres_world:
        q = []
        for c, r in spheres_world:
            u = unit(m // 2).astype(np.float64)
            q.append(rays8((np.asarray(c) + u * (r * rng.uniform(0.0, 1.02, (m // 2, 1)))).astype(f32), unit(m // 2)))
        parts["inside spheres"] = np.concatenate(q)
    # a finite t_max on a third of everything, scaled to the scene
    for name, r in parts.items():
        fin = rng.random(r.shape[0]) < 0.3
        r[fin, 3] = (rng.random(int(fin.sum())) * 1.5 * size / np.maximum(np.linalg.norm(r[fin, 4:7], axis=1), 1e-30)).astype(f32)
    return parts


FAMILIES = [
    # name, geometry kind, triangles, scale, shift
    ("cornell-scale", "mixed", 6000, 1.0, (0, 0, -3)),
    ("scaled 1e-3", "mixed", 6000, 1e-3, (0, 0, 0)),
    ("scaled 1e2", "mixed", 6000, 1e2, (0, 0, 0)),
    ("scaled 1e4", "mixed", 6000, 1e4, (0, 0, 0)),
    ("moved 1e3", "mixed", 6000, 1.0, (1e3, -1e3, 5e2)),
    ("moved 1e5", "mixed", 6000, 1.0, (1e5, 3e4, -7e4)),
    ("scaled 1e2, moved 1e5", "mixed", 6000, 1e2, (-1e5, 1e5, 1e5)),
    ("coplanar", "coplanar", 4000, 1.0, (0, 0, -3)),
    ("coplanar, moved 1e3", "coplanar", 4000, 1.0, (1e3, 1e3, 1e3)),
    ("slivers", "slivers", 4000, 1.0, (0, 0, -3)),
    ("slivers, scaled 1e2", "slivers", 4000, 1e2, (0, 0, 0)),
    ("tiny scene", "mixed", 6, 1.0, (0, 0, -3)),
    ("tiny scene, moved 1e3", "mixed", 6, 1.0, (1e3, 1e3, -1e3)),
    ("tiny coplanar", "coplanar", 12, 1.0, (0, 0, -3)),
]


def make_family(T, name, seed=1):
    for fam in FAMILIES:
        if fam[0] == name:
            _, kind, n_tris, scale, shift = fam
            rng = np.random.default_rng(seed)
            tris, spheres = base_geometry(rng, n_tris, kind)
            if name.startswith("tiny"):  # few enough primitives for the one-leaf accelerator: the box alone would be 12
                tris = np.concatenate([tris[:4], tris[12:]]) if kind == "mixed" else np.concatenate([tris[:2], tris[12:12 + n_tris]])
            scene, tri32, lo, hi = build_scene(T, tris, spheres, scale, shift)
            sw = [(np.asarray(c, np.float64) * scale + np.asarray(shift, np.float64), r * scale) for c, r in spheres]
            return scene, tri32, lo, hi, sw
    raise KeyError(name)


def attack_camera(T, lo, hi, resolution=48):
    """A camera INSIDE the box, looking across it (every path vertex then lies on the attacked geometry)."""
    lo, hi = np.asarray(lo, np.float64), np.asarray(hi, np.float64)
    pos = lo + (hi - lo) * np.array([0.5, 0.55, 0.93])
    tgt = lo + (hi - lo) * np.array([0.45, 0.4, 0.2])
    film = T.Film([resolution, resolution], T.Bounds2([0.0, 0.0], [1.0, 1.0]), T.LanczosSincFilter([1.0, 1.0], 3.0), 1.0, 1.0, "")
    return T.PerspectiveCamera(T.look_at([float(x) for x in pos], [float(x) for x in tgt], [0, 1, 0]), T.Bounds2([-1.0, -1.0], [1.0, 1.0]), 0.0, 1.0, 0.0, 1e6, 75.0, film)
