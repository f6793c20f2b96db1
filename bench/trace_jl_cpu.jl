# bench/trace_jl_cpu.jl — times the REAL Trace.jl (the reference itself, CPU, `julia -t N`) on the workloads of BASELINE.md §2,
# through its public API only, if a Julia runtime and a Trace.jl checkout happen to exist on the box (neither ships with this
# repository: SURVEY.md F5 — this script has never been executed here).  Prints one JSON line per workload:
#
#     julia -t $(nproc) --project=/path/to/Trace.jl bench/trace_jl_cpu.jl [shadows|caustic] [resolution] [spp_or_iterations]
#
# Rays are not counted by the reference; the line reports wall seconds and camera samples / s (Msample/s), the quantity
# bench.py prints as `Msample_per_s`.  The scenes are the reference's own scripts (docs/src/shadows.md:8-107 with the
# WhittedIntegrator of :105 enabled, docs/code/caustic_glass.jl with its SPPMIntegrator); caustic_glass.jl needs
# Trace.load_triangle_mesh, which is dead code at the 2024_10_08 snapshot (SURVEY.md F4): the mesh is read here with a small
# binary-PLY reader instead.
using Trace, GeometryBasics, LinearAlgebra, Printf

function read_ply(path)
    open(path) do io
        readline(io) == "ply" || error("not a PLY file")
        nv = nf = 0
        while (l = readline(io)) != "end_header"
            t = split(l)
            length(t) == 3 && t[1] == "element" && t[2] == "vertex" && (nv = parse(Int, t[3]))
            length(t) == 3 && t[1] == "element" && t[2] == "face" && (nf = parse(Int, t[3]))
        end
        raw = Vector{Float32}(undef, 6nv)
        read!(io, raw)
        v = [Point3f(raw[6i+1], raw[6i+2], raw[6i+3]) for i in 0:nv-1]
        n = [Trace.Normal3f(raw[6i+4], raw[6i+5], raw[6i+6]) for i in 0:nv-1]
        idx = Vector{UInt32}(undef, 3nf)
        for f in 0:nf-1
            read(io, UInt8) == 3 || error("Only triangles supported.")
            for j in 1:3
                idx[3f+j] = UInt32(read(io, Int32) + 1)
            end
        end
        v, n, idx
    end
end

function shadows_scene()   # docs/src/shadows.md:8-64
    material_red = Trace.MatteMaterial(Trace.ConstantTexture(Trace.RGBSpectrum(0.796f0, 0.235f0, 0.2f0)), Trace.ConstantTexture(0f0))
    material_blue = Trace.MatteMaterial(Trace.ConstantTexture(Trace.RGBSpectrum(0.251f0, 0.388f0, 0.847f0)), Trace.ConstantTexture(0f0))
    material_white = Trace.MatteMaterial(Trace.ConstantTexture(Trace.RGBSpectrum(1f0)), Trace.ConstantTexture(0f0))
    mirror = Trace.MirrorMaterial(Trace.ConstantTexture(Trace.RGBSpectrum(1f0)))
    glass = Trace.GlassMaterial(Trace.ConstantTexture(Trace.RGBSpectrum(1f0)), Trace.ConstantTexture(Trace.RGBSpectrum(1f0)),
        Trace.ConstantTexture(0f0), Trace.ConstantTexture(0f0), Trace.ConstantTexture(1.5f0), true)
    sphere(p, r, m) = Trace.GeometricPrimitive(Trace.Sphere(Trace.ShapeCore(Trace.translate(Vec3f(p...)), false), r, 360f0), m)
    tris = Trace.create_triangle_mesh(Trace.ShapeCore(Trace.translate(Vec3f(0, 0, -2)), false), 4,
        UInt32[1, 2, 3, 1, 4, 3, 2, 3, 5, 6, 5, 3], 6,
        [Point3f(0, 0, 0), Point3f(0, 0, -1), Point3f(1, 0, -1), Point3f(1, 0, 0), Point3f(0, 1, -1), Point3f(1, 1, -1)],
        [Trace.Normal3f(0, 1, 0), Trace.Normal3f(0, 1, 0), Trace.Normal3f(0, 1, 0), Trace.Normal3f(0, 1, 0), Trace.Normal3f(0, 0, 1), Trace.Normal3f(0, 0, 1)])
    prims = Trace.GeometricPrimitive[
        sphere((0.3, 0.11, -2.2), 0.1f0, glass), sphere((0.2, 0.11, -2.6), 0.1f0, material_blue), sphere((0.7, 0.31, -2.8), 0.3f0, mirror),
        sphere((0.7, 0.11, -2.3), 0.1f0, material_red)]
    tri_materials = [mirror, mirror, material_white, material_white]   # docs/src/shadows.md:77-80: the two floor triangles are mirrors
    append!(prims, [Trace.GeometricPrimitive(t, m) for (t, m) in zip(tris, tri_materials)])
    lights = [Trace.PointLight(Trace.translate(Vec3f(-1, 1, 0)), Trace.RGBSpectrum(25f0))]
    Trace.Scene(lights, Trace.BVHAccel(prims, 1))
end

function camera(res, from, to; filename = "trace_jl_cpu.png")
    film = Trace.Film(Point2f(res), Trace.Bounds2(Point2f(0), Point2f(1)), Trace.LanczosSincFilter(Point2f(1f0), 3f0), 1f0, 1f0, filename)
    Trace.PerspectiveCamera(Trace.look_at(Point3f(from...), Point3f(to...), Vec3f(0, 1, 0)), Trace.Bounds2(Point2f(-1f0), Point2f(1f0)), 0f0, 1f0, 0f0, 1f6, 90f0, film)
end

function main()
    what = get(ARGS, 1, "shadows")
    res = parse(Int, get(ARGS, 2, "256"))
    n = parse(Int, get(ARGS, 3, what == "shadows" ? "8" : "2"))
    if what == "shadows"     # BASELINE.json configs[0]: 256 x 256, 8 spp, depth 5
        scene = shadows_scene()
        cam = camera(res, (0, 15, 50), (0, 0, -2))
        integ = Trace.WhittedIntegrator(cam, Trace.UniformSampler(n), 5)
        samples = (res + 2)^2 * n
    else                     # BASELINE.json configs[3] scaled: docs/code/caustic_glass.jl, `n` SPPM iterations, depth 8
        v, nrm, idx = read_ply(get(ENV, "CAUSTIC_PLY", joinpath(@__DIR__, "..", "tests", "golden", "caustic-glass.ply")))
        glass = Trace.GlassMaterial(Trace.ConstantTexture(Trace.RGBSpectrum(1f0)), Trace.ConstantTexture(Trace.RGBSpectrum(1f0)),
            Trace.ConstantTexture(0f0), Trace.ConstantTexture(0f0), Trace.ConstantTexture(1.25f0), true)
        plastic = Trace.PlasticMaterial(Trace.ConstantTexture(Trace.RGBSpectrum(0.64f0)), Trace.ConstantTexture(Trace.RGBSpectrum(0.1f0)), Trace.ConstantTexture(0.010408001f0), true)
        tris = Trace.create_triangle_mesh(Trace.ShapeCore(Trace.translate(Vec3f(5, -1.49, -100)), false), length(idx) ÷ 3, idx, length(v), v, nrm)
        floor = Trace.create_triangle_mesh(Trace.ShapeCore(Trace.translate(Vec3f(-10, 0, -87)), false), 2, UInt32[1, 2, 3, 1, 4, 3], 4,
            [Point3f(0, 0, 0), Point3f(0, 0, -30), Point3f(30, 0, -30), Point3f(30, 0, 0)], [Trace.Normal3f(0, 1, 0) for _ in 1:4])
        prims = vcat([Trace.GeometricPrimitive(t, glass) for t in tris], [Trace.GeometricPrimitive(t, plastic) for t in floor])
        from, to = Point3f(0, 2, 0), Point3f(-5, 0, 5)
        dir = normalize(Vec3f(to - from))
        dir, du, dv = Trace.coordinate_system(dir, Vec3f(0f0))
        dir_to_z = Trace.Transformation(transpose(Mat4f(du[1], du[2], du[3], 0, dv[1], dv[2], dv[3], 0, dir[1], dir[2], dir[3], 0, 0, 0, 0, 1)))
        l2w = Trace.translate(Vec3f(4.5, 0, -101)) * Trace.translate(Vec3f(from)) * inv(dir_to_z)
        scene = Trace.Scene([Trace.SpotLight(l2w, Trace.RGBSpectrum(60f0), 30f0, 20f0)], Trace.BVHAccel(prims, 1))
        cam = camera(res, (0, 150, 150), (-3, 0, -91))
        integ = Trace.SPPMIntegrator(cam, 0.075f0, 8, n, -1)
        samples = res^2 * n
    end
    integ(scene)   # warm-up (compilation)
    t = @elapsed integ(scene)
    @printf("{\"workload\": \"%s\", \"resolution\": %d, \"n\": %d, \"threads\": %d, \"seconds\": %.3f, \"Msample_per_s\": %.4f, \"kind\": \"reference\"}\n",
        what, res, n, Threads.nthreads(), t, samples / t / 1e6)
end

main()
