# TraceHIP.jl — the shim a Trace.jl user loads to run the render hot path on an MI355X through libtracehip.so.
#
# STATUS: written against include/tracehip.h, NEVER EXECUTED — the build image has no Julia runtime (SURVEY.md F5).
# The tested host over the same C ABI is the Python mirror (trace.jl_amd/api.py); this file is the reference-side
# binding a maintainer would add (INTEGRATION.md).  What it sends across the boundary — the exact call sequence and argument
# layout of flatten() / sensor() / render!() — is written down in tests/golden/julia_shim_calls.json and REPLAYED through ctypes by
# tests/test_gpu_julia_replay.py against the Python host's film, so the marshalling below is pinned even though Julia is absent.  Scene scripts stay unchanged: they build Trace.Scene / Trace.Film /
# Trace.PerspectiveCamera with Trace.jl's own constructors (so every load-bearing matrix bug is the reference's own) and
# call `integrator(scene)`; the methods below replace the CPU render loops of src/integrators/sampler.jl:12-56.
module TraceHIP

using Trace
using GeometryBasics
using StaticArrays
using Random

const LIB = get(ENV, "TRACEHIP_LIB", joinpath(@__DIR__, "..", "libtracehip.so"))

# ---- mirrors of the C structs (include/tracehip.h) -------------------------------------------------------------------
struct TrhipSensor
    raster_to_camera::NTuple{16,Float32}
    camera_to_world::NTuple{16,Float32}
    lens_radius::Float32
    focal_distance::Float32
    shutter_open::Float32
    shutter_close::Float32
    crop_min::NTuple{2,Float32}
    crop_max::NTuple{2,Float32}
    filter_radius::NTuple{2,Float32}
    filter_table::NTuple{256,Float32}
    scale::Float32
end

mutable struct TrhipStats
    camera_samples::UInt64
    closest_rays::UInt64
    shadow_rays::UInt64
    nodes_visited::UInt64
    prims_tested::UInt64
    nodes_visited_shadow::UInt64
    prims_tested_shadow::UInt64
    ms_total::Float64
    ms_raygen::Float64
    ms_trace_closest::Float64
    ms_shade::Float64
    ms_trace_any::Float64
    ms_film::Float64
    launches::NTuple{5,UInt32}
    n_batches::UInt32
    max_depth_reached::UInt32
    traversal::UInt32
    node_bytes::UInt32
    replicated_rays::UInt64
    fallback_rays::UInt64
    ms_sub::NTuple{4,Float64}
    launches_sub::NTuple{4,UInt32}
    count_sub::NTuple{4,UInt64}
    ms_fallback::Float64
    launches_fallback::UInt32
    reserved0::UInt32
    nodes_visited_fallback::UInt64
    prims_tested_fallback::UInt64
    TrhipStats() = new(0, 0, 0, 0, 0, 0, 0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, (0, 0, 0, 0, 0), 0, 0, 0, 0, 0, 0, (0.0, 0.0, 0.0, 0.0), (0, 0, 0, 0), (0, 0, 0, 0), 0.0, 0, 0, 0, 0)
end

struct TraceHIPError <: Exception
    code::Cint
    msg::String
end

const CTX = Ref{Ptr{Cvoid}}(C_NULL)
# hand Trace.jl's own BVH (scene.aggregate.nodes) to the library instead of letting it build one: same tree, same equal-t tie-breaks as the CPU path
const EXACT_TREE = Ref(true)

function context()
    if CTX[] == C_NULL
        rc = ccall((:trhip_init, LIB), Cint, (Ptr{Ptr{Cvoid}}, Cint), CTX, parse(Cint, get(ENV, "LOCAL_RANK", "0")))
        rc == 0 || throw(TraceHIPError(rc, unsafe_string(ccall((:trhip_last_error, LIB), Cstring, (Ptr{Cvoid},), C_NULL))))
    end
    CTX[]
end

check(rc) = rc == 0 || throw(TraceHIPError(rc, unsafe_string(ccall((:trhip_last_error, LIB), Cstring, (Ptr{Cvoid},), CTX[]))))

# Mat4f is column-major; the ABI wants row-major m[4*row + col].
rowmajor(m::Mat4f) = NTuple{16,Float32}(vec(permutedims(Matrix(m))))
rowmajor_vec(m::Mat4f) = collect(rowmajor(m))

# ---- scene flattening: Scene -> BVHAccel -> GeometricPrimitive -> shape / material --------------------------------------
const MATTE, MIRROR, GLASS, PLASTIC = Cint(0), Cint(1), Cint(2), Cint(3)
const NO_MATERIAL = UInt32(0x00ffffff)

rgb(t::Trace.ConstantTexture) = Float32[t.value.c...]
flt(t::Trace.ConstantTexture) = Float32(t.value)
material_params(m::Trace.MatteMaterial) = (MATTE, vcat(rgb(m.Kd), flt(m.σ)))
material_params(m::Trace.MirrorMaterial) = (MIRROR, rgb(m.Kr))
material_params(m::Trace.GlassMaterial) =
    (GLASS, vcat(rgb(m.Kr), rgb(m.Kt), flt(m.u_roughness), flt(m.v_roughness), flt(m.index), m.remap_roughness ? 1f0 : 0f0))
material_params(m::Trace.PlasticMaterial) = (PLASTIC, vcat(rgb(m.Kd), rgb(m.Ks), flt(m.roughness), m.remap_roughness ? 1f0 : 0f0))

function flatten(scene::Trace.Scene)
    ctx = context()
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:trhip_scene_new, LIB), Cint, (Ptr{Cvoid}, Ptr{Ptr{Cvoid}}), ctx, h))
    s = h[]
    mat_ids = IdDict{Any,UInt32}()
    function material_id(m)
        m === nothing && return NO_MATERIAL
        get!(mat_ids, m) do
            kind, params = material_params(m)
            id = Ref{UInt32}(0)
            check(ccall((:trhip_scene_add_material, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{Float32}, Cint, Ptr{UInt32}), s, kind, params, length(params), id))
            id[]
        end
    end
    bvh = scene.aggregate::Trace.BVHAccel
    # Every GeometricPrimitive under the aggregate, in order; a BVHAccel may itself be a primitive of another
    # (primitive.jl / accel/bvh.jl:50-53, test/test_intersection.jl:137-138): its primitives are spliced in place.
    # `bvh.primitives` is the ORDERED list the constructor left behind (bvh.jl:66-78): primitive k of the flat list is ordered slot k of
    # `bvh.nodes`, so Trace.jl's own tree goes to the library as it is (EXACT_TREE, below) and equal-t ties resolve as they do on the CPU.
    prims = Trace.GeometricPrimitive[]
    function collect_prims!(list)
        for p in list
            if p isa Trace.BVHAccel
                collect_prims!(p.primitives)
            elseif p isa Trace.GeometricPrimitive
                push!(prims, p)
            else
                error("TraceHIP: unsupported primitive $(typeof(p))")
            end
        end
    end
    collect_prims!(bvh.primitives)
    i = 1
    while i <= length(prims)
        p = prims[i]
        shape = p.shape
        if shape isa Trace.Sphere
            o2w = shape.core.object_to_world
            check(ccall((:trhip_scene_add_sphere_fields, LIB), Cint,
                (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}, Cint, Float32, Float32, Float32, Float32, Float32, Float32, UInt32, Ptr{UInt32}),
                s, rowmajor_vec(o2w.m), rowmajor_vec(o2w.inv_m), shape.core.reverse_orientation, shape.radius, shape.z_min, shape.z_max,
                shape.θ_min, shape.θ_max, shape.ϕ_max, material_id(p.material), C_NULL))
            i += 1
        elseif shape isa Trace.Triangle
            # ONE call per run of consecutive triangles of the same TriangleMesh (create_triangle_mesh returns them in order,
            # triangle_mesh.jl:45-58): the mesh's vertex / normal arrays cross the boundary once, with the run's index triples and
            # per-triangle materials — what trace.jl_amd/api.py does for the same object graph.
            mesh = shape.mesh
            flip = shape.core.reverse_orientation ⊻ shape.core.transform_swaps_handedness
            j = i
            idx = UInt32[]
            mats = UInt32[]
            uvc = Float32[]   # the run's corner (u, v)s: Trace.jl reads mesh.uv[t.i + j], by corner position (triangle_mesh.jl:82)
            while j <= length(prims) && prims[j].shape isa Trace.Triangle && prims[j].shape.mesh === mesh &&
                  (prims[j].shape.core.reverse_orientation ⊻ prims[j].shape.core.transform_swaps_handedness) == flip
                t = prims[j].shape
                append!(idx, (mesh.indices[t.i], mesh.indices[t.i+1], mesh.indices[t.i+2]))   # 1-based, as the ABI wants them
                push!(mats, material_id(prims[j].material))
                if mesh.uv !== nothing
                    for c in 0:2
                        append!(uvc, (mesh.uv[t.i+c][1], mesh.uv[t.i+c][2]))
                    end
                end
                j += 1
            end
            verts = collect(reinterpret(Float32, mesh.vertices))          # already world space (triangle_mesh.jl:23)
            nrm = mesh.normals === nothing ? C_NULL : collect(reinterpret(Float32, mesh.normals))
            if mesh.uv === nothing && mesh.tangents === nothing
                check(ccall((:trhip_scene_add_triangles, LIB), Cint,
                    (Ptr{Cvoid}, Ptr{Float32}, UInt32, Ptr{UInt32}, UInt32, Ptr{Float32}, Ptr{UInt32}, Cint, Ptr{UInt32}),
                    s, verts, mesh.n_vertices, idx, length(mats), nrm, mats, flip, C_NULL))
            else
                tang = mesh.tangents === nothing ? C_NULL : collect(reinterpret(Float32, mesh.tangents))   # per vertex, untransformed (:27)
                check(ccall((:trhip_scene_add_triangles_ex, LIB), Cint,
                    (Ptr{Cvoid}, Ptr{Float32}, UInt32, Ptr{UInt32}, UInt32, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Ptr{UInt32}, Cint, Ptr{UInt32}),
                    s, verts, mesh.n_vertices, idx, length(mats), nrm, tang, mesh.uv === nothing ? C_NULL : uvc, mats, flip, C_NULL))
            end
            i = j
        else
            error("TraceHIP: unsupported shape $(typeof(shape))")
        end
    end
    for l in scene.lights
        I = Float32[l.i.c...]
        m, im = rowmajor_vec(l.light_to_world.m), rowmajor_vec(l.light_to_world.inv_m)
        if l isa Trace.PointLight
            check(ccall((:trhip_scene_add_point_light, LIB), Cint, (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}), s, m, im, I))
        elseif l isa Trace.SpotLight
            check(ccall((:trhip_scene_add_spot_light_fields, LIB), Cint, (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Float32, Float32),
                s, m, im, I, l.cos_total_width, l.cos_falloff_start))
        else
            error("TraceHIP: unsupported light $(typeof(l))")
        end
    end
    # Trace.jl's own BVH topology (accel/bvh.jl:38-48 nodes, depth-first, first child = i + 1) instead of a tree the library builds: where two
    # primitives are accepted at the same t the later visited one wins (bvh.jl:229-237), so only the SAME tree gives the CPU's answer on those
    # rays.  Not possible with a BVHAccel nested as a primitive (its own walk inside the outer one has no flat equivalent) or a tree deeper than
    # the 64-entry stack (where the reference itself throws, bvh.jl:222): those scenes get the library's tree.
    if EXACT_TREE[] && !any(p -> p isa Trace.BVHAccel, bvh.primitives) && !isempty(bvh.nodes)
        n = length(bvh.nodes)
        bounds = Vector{Float32}(undef, 6n)
        a = Vector{UInt32}(undef, n)
        flags = Vector{UInt32}(undef, n)
        for (k, nd) in enumerate(bvh.nodes)
            bounds[6k-5:6k-3] .= nd.bounds.p_min
            bounds[6k-2:6k] .= nd.bounds.p_max
            if nd isa Trace.LinearBVHLeaf
                a[k] = nd.primitives_offset - 1                       # first ordered slot, 0-based
                flags[k] = (UInt32(nd.n_primitives) << 2) | UInt32(3)
            else
                a[k] = nd.second_child_offset - 1                     # 0-based index of the second child
                flags[k] = UInt32(nd.split_axis - 1)
            end
        end
        order = collect(UInt32(0):UInt32(length(prims) - 1))          # flat primitive k IS ordered slot k
        rc = ccall((:trhip_scene_set_bvh, LIB), Cint, (Ptr{Cvoid}, Ptr{Float32}, Ptr{UInt32}, Ptr{UInt32}, UInt32, Ptr{UInt32}, UInt32),
            s, bounds, a, flags, n, order, length(prims))
        rc == 0 && return s
        rc == -3 || check(rc)                                          # TRHIP_ERR_UNSUPPORTED (deeper than 64): fall through
    end
    check(ccall((:trhip_scene_commit, LIB), Cint, (Ptr{Cvoid}, Cint), s, bvh.max_node_primitives))
    s
end

function sensor(camera::Trace.PerspectiveCamera)
    film = Trace.get_film(camera)
    pc = camera.core
    TrhipSensor(rowmajor(pc.raster_to_camera.m), rowmajor(pc.core.camera_to_world.m), pc.lens_radius, pc.focal_distance,
        pc.core.shutter_open, pc.core.shutter_close, Tuple(film.crop_bounds.p_min), Tuple(film.crop_bounds.p_max), Tuple(film.filter.radius),
        NTuple{256,Float32}(vec(permutedims(film.filter_table))),   # (y, x) matrix -> table[16*y + x]
        film.scale)
end

# ---- the seeded sampler (include/trace_sampler.h) behind UniformSampler's protocol ---------------------------------------
mutable struct SeededSampler <: Trace.AbstractSampler
    current_sample::Int64
    samples_per_pixel::Int64
    seed::UInt64
    sample_offset::UInt32
    SeededSampler(spp::Integer; seed::Integer = 0x5EED0001, sample_offset::Integer = 0) = new(1, spp, seed, sample_offset)
end
seed_of(s::SeededSampler) = (s.seed, s.sample_offset)
seed_of(::Trace.UniformSampler) = (UInt64(0x5EED0001), UInt32(0))   # the reference's sampler has no seed (F7)

struct PathIntegrator <: Trace.SamplerIntegrator
    camera::Trace.Camera
    sampler::Trace.AbstractSampler
    max_depth::Int64
end

# ---- multi-GPU jobs (include/tracehip.h "multi-GPU"): one Julia process per GPU ----------------------------------------------
# RANK / WORLD_SIZE / LOCAL_RANK in the environment (any launcher); rank 0 writes the 128-byte RCCL id to TRACEHIP_ID_FILE, the
# others read it: trhip_comm_init.  Then every rank renders its share of the samples (`sample_offset`) and trhip_film_reduce sums
# the film accumulators onto rank 0, which alone writes the image; SPPM shards its photons inside trhip_render_sppm.
const JOB = Ref{Tuple{Int,Int}}((0, 1))
const JOB_ID_PATH = Ref{String}("")
# Optional suffix of the id file's name (two jobs sharing TRACEHIP_ID_FILE): the same on every rank under any launcher — no pid, no ppid.  What makes a file a crashed
# job left behind harmless is the token handshake below, not the name.
function job_suffix()
    for name in ("TRACEHIP_JOB_ID", "SLURM_JOB_ID", "PBS_JOBID", "LSB_JOBID")
        haskey(ENV, name) && !isempty(ENV[name]) && return string(".", ENV[name])
    end
    ""
end
write_atomically(path, data) = (tmp = string(path, ".tmp", getpid()); write(tmp, data); mv(tmp, path; force = true))
read_or_empty(path) = try read(path) catch; UInt8[] end
random_token() = bytes2hex(rand(Random.RandomDevice(), UInt8, 16))
# The RCCL id from rank 0 to the others through a shared directory — the protocol of trace.jl_amd/parallel.py file_rendezvous, byte for byte (a job may mix hosts):
# reader r writes <path>.hello<r> = a fresh token, polls <path> for the line "r:<token>", answers with <path>.ack<r> = "<token>:<rank 0's token>"; rank 0 rewrites
# <path> = id, "0:<its token>", one line per reader seen, until every ack matches.  Both sides give up after `timeout_s` (TRACEHIP_RENDEZVOUS_TIMEOUT, default 120).
function file_rendezvous(path, rank, world; timeout_s = parse(Float64, get(ENV, "TRACEHIP_RENDEZVOUS_TIMEOUT", "120")))
    t0 = time()
    if rank == 0
        id = Vector{UInt8}(undef, 128)
        check(ccall((:trhip_comm_unique_id, LIB), Cint, (Ptr{UInt8},), id))
        mine = random_token()
        written = nothing
        while true
            tokens = Dict{Int,String}()
            for r in 1:world-1
                tok = strip(String(read_or_empty(string(path, ".hello", r))))
                isempty(tok) || (tokens[r] = tok)
            end
            if tokens != written
                io = IOBuffer()
                write(io, id)
                write(io, "0:", mine, "\n")
                for r in sort(collect(keys(tokens)))
                    write(io, string(r), ":", tokens[r], "\n")
                end
                write_atomically(path, take!(io))
                written = copy(tokens)
            end
            if length(tokens) == world - 1 && all(strip(String(read_or_empty(string(path, ".ack", r)))) == string(tokens[r], ":", mine) for r in 1:world-1)
                return id
            end
            time() - t0 > timeout_s && error("TraceHIP: RCCL id rendezvous at $path: not every rank showed up and acknowledged within $timeout_s s")
            sleep(0.05)
        end
    end
    token = random_token()
    write_atomically(string(path, ".hello", rank), token)
    want = string(rank, ":", token)
    while true
        data = read_or_empty(path)
        if length(data) > 128
            lines = split(String(data[129:end]), "\n")
            if want in lines && startswith(lines[1], "0:")
                write_atomically(string(path, ".ack", rank), string(token, ":", lines[1][3:end]))
                return data[1:128]
            end
        end
        time() - t0 > timeout_s && error("TraceHIP: no RCCL id for rank $rank at $path after $timeout_s s (is rank 0 running, and is the directory shared?)")
        sleep(0.05)
    end
end
function init_job!(; rank = parse(Int, get(ENV, "RANK", "0")), world = parse(Int, get(ENV, "WORLD_SIZE", "1")), id_file = get(ENV, "TRACEHIP_ID_FILE", ""))
    world <= 1 && return JOB[]
    isempty(id_file) && error("TraceHIP: set TRACEHIP_ID_FILE to a path all ranks can reach")
    path = string(id_file, job_suffix())
    id = file_rendezvous(path, rank, world)
    rank == 0 && (JOB_ID_PATH[] = path)
    check(ccall((:trhip_comm_init, LIB), Cint, (Ptr{Cvoid}, Ptr{UInt8}, Cint, Cint), context(), id, rank, world))
    JOB[] = (rank, world)
end
# end of the job: the communicator goes, rank 0 removes the id file (every rank has read it: trhip_comm_init returned everywhere)
function close_job!()
    JOB[][2] > 1 && ccall((:trhip_comm_destroy, LIB), Cint, (Ptr{Cvoid},), context())
    if !isempty(JOB_ID_PATH[])
        rm(JOB_ID_PATH[]; force = true)
        for kind in ("hello", "ack"), r in 1:JOB[][2]-1
            rm(string(JOB_ID_PATH[], ".", kind, r); force = true)
        end
    end
    JOB_ID_PATH[] = ""
    JOB[] = (0, 1)
end
# the `spp` samples of one frame split over the ranks: (spp of this rank, first global sample index)
function shard_samples(spp::Integer)
    rank, world = JOB[]
    base, rem = divrem(spp, world)
    (base + (rank < rem ? 1 : 0), rank * base + min(rank, rem))
end

function write_film!(film, out, h, w; clear_splat = false)
    @inbounds for y in 1:h, x in 1:w            # film.pixels is (y, x); out is row-major over (y, x)
        k = 4 * ((y - 1) * w + (x - 1))
        px = film.pixels[y, x]
        px.xyz = Point3f(out[k+1], out[k+2], out[k+3])
        px.filter_weight_sum = out[k+4]
        clear_splat && (px.splat_xyz = Point3f(0f0))
    end
end

function render!(entry::Symbol, i, scene::Trace.Scene)
    film = Trace.get_film(i.camera)
    s = flatten(scene)
    sn = Ref(sensor(i.camera))
    h, w = size(film.pixels)
    out = Vector{Float32}(undef, 4 * h * w)
    stats = TrhipStats()
    seed, offset = seed_of(i.sampler)
    rank, world = JOB[]
    spp, first = world > 1 ? shard_samples(i.sampler.samples_per_pixel) : (i.sampler.samples_per_pixel, 0)
    if world > 1
        # device-resident film, summed over the ranks by the library (ncclReduce), then copied out on rank 0
        d_film = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:hipMalloc, "libamdhip64"), Cint, (Ptr{Ptr{Cvoid}}, Csize_t), d_film, sizeof(out)))
        rc = ccall((Symbol(entry, :_device), LIB), Cint,
            (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{TrhipSensor}, UInt32, Cint, UInt64, UInt32, Ptr{Cvoid}, Ptr{TrhipStats}),
            context(), s, sn, max(spp, 1), i.max_depth, seed, offset + first, d_film[], Ref(stats))
        rc == 0 && spp == 0 && ccall((:hipMemset, "libamdhip64"), Cint, (Ptr{Cvoid}, Cint, Csize_t), d_film[], 0, sizeof(out))
        # a rank whose render failed still ENTERS the collective (the others would block in it for ever) — with a film of NaNs (0xff bytes), so
        # that the sum on rank 0 is NaN everywhere and rank 0 fails too instead of writing an image that misses this rank's samples
        rc == 0 || ccall((:hipMemset, "libamdhip64"), Cint, (Ptr{Cvoid}, Cint, Csize_t), d_film[], 0xff, sizeof(out))
        rc_red = ccall((:trhip_film_reduce, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, UInt64, Cint), context(), d_film[], h * w, 0)
        rc == 0 && (rc = rc_red)
        rc == 0 && rank == 0 && ccall((:hipMemcpy, "libamdhip64"), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Csize_t, Cint), out, d_film[], sizeof(out), 2)
        ccall((:hipFree, "libamdhip64"), Cint, (Ptr{Cvoid},), d_film[])
    else
        rc = ccall((entry, LIB), Cint,
            (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{TrhipSensor}, UInt32, Cint, UInt64, UInt32, Ptr{Float32}, Ptr{TrhipStats}),
            context(), s, sn, spp, i.max_depth, seed, offset, out, Ref(stats))
    end
    ccall((:trhip_scene_free, LIB), Cvoid, (Ptr{Cvoid},), s)
    check(rc)
    rank == 0 || return nothing
    world > 1 && isnan(out[4]) && error("TraceHIP: a rank of the job failed to render its samples (the reduced film is NaN)")
    write_film!(film, out, h, w)
    Trace.save(film)
end

(i::PathIntegrator)(scene::Trace.Scene) = render!(:trhip_render_path, i, scene)

# SPPMIntegrator (integrators/sppm.jl:132-173) on the device: trhip_render_sppm returns the film after set_image!
# (film.jl:195-202).  `seed` selects the seeded stream of the camera pass (the reference draws from the global RNG there).
# The reference's periodic image (sppm.jl:166-171: every iteration that write_frequency divides is stored in the film and saved): the library calls back with the
# image of the first k iterations; the last iteration's image is the call's result.
mutable struct SppmWriteCtx  # (mutable: pointer_from_objref needs an object with an address)
    film::Trace.Film
    h::Int
    w::Int
end
function sppm_write_cb(user::Ptr{Cvoid}, iteration::UInt32, xyzw::Ptr{Float32})::Cint
    c = unsafe_pointer_to_objref(user)::SppmWriteCtx
    JOB[][1] == 0 || return Cint(0)
    write_film!(c.film, unsafe_wrap(Array, xyzw, 4 * c.h * c.w), c.h, c.w; clear_splat = true)
    Trace.save(c.film)
    Cint(0)
end
# periodic_images: the reference stores and saves the image after every iteration `write_frequency` divides (sppm.jl:166-171; its default, 1, is EVERY iteration) — faithful, and
# about 3x the time of a call without them (one iteration per batch of traversal launches + an image copy and a PNG per iteration).  `periodic_images = false` renders the
# last image only; a default-constructed integrator with periodic images on gets a one-time note.
const SPPM_NOTED = Ref(false)
function render_sppm!(i::Trace.SPPMIntegrator, scene::Trace.Scene; seed::Integer = 0x5EED0001, periodic_images::Bool = true)
    film = Trace.get_film(i.camera)
    s = flatten(scene)
    sn = Ref(sensor(i.camera))
    h, w = size(film.pixels)
    out = Vector{Float32}(undef, 4 * h * w)
    stats = TrhipStats()
    wf = periodic_images ? UInt32(clamp(i.write_frequency, 0, typemax(UInt32))) : UInt32(0)
    if wf == 1 && i.n_iterations > 1 && !SPPM_NOTED[]
        SPPM_NOTED[] = true
        @info "TraceHIP: write_frequency = 1 (Trace.jl's default) saves the image after every SPPM iteration: about 3x the time of render_sppm!(…; periodic_images = false)"
    end
    user = SppmWriteCtx(film, Int(h), Int(w))
    cb = @cfunction(sppm_write_cb, Cint, (Ptr{Cvoid}, UInt32, Ptr{Float32}))
    rc = GC.@preserve user ccall((:trhip_render_sppm_ex, LIB), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{TrhipSensor}, Cfloat, Cint, UInt32, Int64, UInt64, Ptr{Float32}, Ptr{TrhipStats}, UInt32, Ptr{Cvoid}, Ptr{Cvoid}),
        context(), s, sn, i.initial_search_radius, i.max_depth, i.n_iterations, i.photons_per_iteration, UInt64(seed), out, Ref(stats),
        wf < i.n_iterations ? wf : UInt32(0), cb, pointer_from_objref(user))
    ccall((:trhip_scene_free, LIB), Cvoid, (Ptr{Cvoid},), s)
    check(rc)
    JOB[][1] == 0 || return nothing                 # with a communicator every rank holds the whole image; rank 0 writes it
    write_film!(film, out, h, w; clear_splat = true)  # filter_weight_sum = 1 after set_image!
    Trace.save(film)
end
# Opt-in replacement of the CPU loop for SPPMIntegrator (shadows integrators/sppm.jl:132):
accelerate_sppm!() = @eval (i::Trace.SPPMIntegrator)(scene::Trace.Scene) = render_sppm!(i, scene)
# Opt-in replacement of the CPU loop for WhittedIntegrator (shadows integrators/sampler.jl:12):
accelerate_whitted!() = @eval (i::Trace.WhittedIntegrator)(scene::Trace.Scene) = render!(:trhip_render_whitted, i, scene)

end # module
