// th_scene.h — flat, HBM-resident scene representation shared by the host flattener and the kernels.
//
// Layout (DESIGN.md "data layout in HBM"):
//   nodes      float4[2*n_nodes]   node i = {min.xyz, a}, {max.xyz, flags}; 32 B like LinearBVHLeaf/Interior (bvh.jl:38-48).
//                                  leaf: (flags & 3) == 3, a = first ordered-primitive slot, n = flags >> 2
//                                  interior: a = second child, flags = split axis 0..2, first child = i + 1 (bvh.jl:187-206)
//   prims      float4[3*n_prims]   ordered primitive slot k (BVH leaf order = BVHAccel.primitives):
//                                  triangle: v0 v1 v2 (world space); v0.w = as_float(meta), meta = material | flags << 24
//                                  sphere:   {as_float(sphere index), 0, 0, as_float(meta)}, unused, unused
//   tri_nrm    float4[3*n_prims]   vertex normals of slot k (zeros when the mesh has none / slot is a sphere)
//   tri_tan    float4[3*n_prims]   vertex tangents of slot k (only when some mesh has them; PRIM_HAS_TANGENTS)
//   shade      float4[8*n_prims]   prims and tri_nrm of slot k side by side, 128-byte aligned: what rebuild_shading reads
//   spheres    SphereRec[n_spheres]
//   materials  MaterialRec[n_materials]: the ≤2 BxDF lobes each material adds, for allow_multiple_lobes = false / true
//   lights     LightRec[n_lights]
#pragma once
#include "th_math.h"

namespace th {

enum : uint32_t {
    PRIM_SPHERE = 1u << 24,       // slot holds a sphere reference
    PRIM_HAS_NORMALS = 1u << 25,  // triangle mesh has vertex normals
    PRIM_FLIP = 1u << 26,         // reverse_orientation XOR transform_swaps_handedness
    PRIM_DEGENERATE = 1u << 27,
    PRIM_FAST = 1u << 28,         // triangle whose material (compute_scattering! with multiple lobes) is ONE LambertianReflection lobe: its
                                  // reflectance rides in the .w lanes of the three normal records (upload_scene), the shading kernel
                                  // classifies and shades it without touching the material table   // is_degenerate(triangle) (triangle_mesh.jl:65-68), evaluated once at scene commit
    PRIM_HAS_TANGENTS = 1u << 29, // the mesh has vertex tangents: slot k's three are in DeviceScene::tri_tan (triangle_mesh.jl:172-176)
    PRIM_MATERIAL_MASK = 0x00ffffffu,
    PRIM_NO_MATERIAL = 0x00ffffffu
};

struct SphereRec {      // shapes/sphere.jl:1-30
    float o2w[16];      // core.object_to_world.m
    float o2w_inv[16];  // core.object_to_world.inv_m  (== world_to_object.m)
    float radius, z_min, z_max, theta_min, theta_max, phi_max;
    uint32_t flip;  // reverse_orientation XOR transform_swaps_handedness
    uint32_t never_clipped;  // !(z_min > -r) && !(z_max < r) && phi_max >= Float32(2π): test_clipping (sphere.jl:65-69) is constantly false
};

enum LobeKind : int32_t { LOBE_LAMBERT_R = 0, LOBE_OREN_NAYAR = 1, LOBE_SPECULAR_R = 2, LOBE_SPECULAR_T = 3, LOBE_FRESNEL_SPECULAR = 4, LOBE_MICROFACET_R = 5, LOBE_MICROFACET_T = 6, LOBE_LAMBERT_T = 7 };
enum : int32_t { BSDF_NONE = 0, BSDF_REFLECTION = 1, BSDF_TRANSMISSION = 2, BSDF_DIFFUSE = 4, BSDF_GLOSSY = 8, BSDF_SPECULAR = 16, BSDF_ALL = 31 };  // bxdf.jl:1-7
enum : int32_t { FRESNEL_NOOP = 0, FRESNEL_DIELECTRIC = 1 };

struct Lobe {  // one BxDF (reflection/*.jl); 64 B
    int32_t kind, type, fresnel, pad;
    float r[3];            // r (reflection spectrum) or t for pure transmission lobes
    float t[3];            // FresnelSpecular's t
    float a, b;            // OrenNayar a,b | TrowbridgeReitz α_x, α_y
    float eta_a, eta_b;    // SpecularTransmission / FresnelSpecular / MicrofacetTransmission
    float fr_eta_i, fr_eta_t;  // FresnelDielectric(ηi, ηt)
};
struct LobeSet {
    int32_t n;
    float eta;  // BSDF.η (bsdf.jl:11)
    int32_t pad[2];
    Lobe lobe[2];
};
struct MaterialRec {
    LobeSet set[2];  // [allow_multiple_lobes]
};

struct LightRec {  // lights/point.jl:1-24, lights/spot.jl:1-19
    int32_t kind;  // 0 point, 1 spot
    float position[3];
    float I[3];
    float cos_total_width, cos_falloff_start;
    float w2l[9];  // world_to_light.m[1:3,1:3] (= light_to_world.inv_m), row-major, for falloff (spot.jl:32-34)
    float l2w[9];  // light_to_world.m[1:3,1:3], row-major, for sample_le (spot.jl:49)
    float pad;
};

struct DeviceScene {
    const float4* nodes;
    const float4* prims;
    const float4* tri_nrm;
    const float4* tri_tan;  // float4[3*n_prims] vertex tangents of slot k, or null when no mesh of the scene has tangents
    const float4* shade;  // the shading kernels' view of slot k: {v0 | meta, v1, v2, n0 | r, n1 | g, n2 | b, geometric normal, unit dpdu (k_shade_constants)} in ONE 128-byte line (prims + tri_nrm
                          // interleaved): a path vertex gathers one line instead of ~2.75
    const SphereRec* spheres;
    const MaterialRec* materials;
    const LightRec* lights;
    uint32_t n_nodes, n_prims, n_spheres, n_materials, n_lights;
};

struct DeviceSensor {
    float raster_to_camera[16];
    float camera_to_world[16];
    float lens_radius, focal_distance, shutter_open, shutter_close;
    float crop_min[2], crop_max[2];
    float filter_radius[2];
    float scale;
    int32_t sb_min[2], sb_max[2];  // get_sample_bounds(film) film.jl:68-73
    int32_t sb_w, sb_h;            // sample-pixel grid
    int32_t film_w, film_h;        // size(film.pixels) = (film_h, film_w)
    int32_t tiles_x, tiles_y;      // 16x16 sample tiles, integrators/sampler.jl:15-20
    // A frame too large for HBM is rendered in BANDS of whole tile rows (tu_path.hip, render_impl): the wavefront and the film gather of one
    // launch cover sample rows band_y0 .. band_y0 + band_rows - 1 = tile rows band_ty0 .. band_ty1; with `accumulate` the gather adds the band's
    // tiles, in k order, onto what the film pixel already holds — merge_film_tile! (film.jl:182-193) adds tile after tile the same way.
    int32_t band_y0, band_rows, band_ty0, band_ty1, accumulate;
};

}  // namespace th
