#!/usr/bin/env python
"""DESIGN.md and README.md from docs/templates/*.in.md: every RN_* name is replaced by a figure read from the round's committed bench lines (profiles/<round>/<tag>_bench_*.json), so the
prose never quotes a number no file under profiles/ holds.

    python tools/fill_docs.py [--round r6] [--tag r6z] [--so-mb 5.0] [--gpu-suite-s 140]
"""
import argparse
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


ROUND = "r6"


def line(tag, name):
    with open(os.path.join(ROOT, "profiles", ROUND, f"{tag}_bench_{name}.json")) as f:
        return json.loads(f.read())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--round", default="r6")
    ap.add_argument("--tag", default="r6z")
    ap.add_argument("--so-mb", type=float, default=os.path.getsize(os.path.join(ROOT, "trace.jl_amd", "libtracehip.so")) / 1e6)
    ap.add_argument("--gpu-suite-s", type=float, default=140.0)
    a = ap.parse_args()
    global ROUND
    ROUND = a.round
    m, c, b, m10, s, c5 = (line(a.tag, n) for n in ("mesh1m", "cornell", "blob_870k", "mesh_10m", "caustic_sppm", "c5_share"))
    r = m["roofline"]
    k = r["kernel_ms_per_step"]
    fb = r["hybrid"]["fallback_walk_ms_per_step"]
    frame = m["ms_per_step"]
    bytes_ray = 48 + 32 * r["visits_per_ray"]["closest_nodes"] + 48 * r["visits_per_ray"]["closest_prims"]
    pct = lambda x: f"{100.0 * x / frame:.1f} %"
    hr = m.get("hbm_resident") or {}
    rows = []
    for name, d in (("S-mesh, 1 M triangles (the default bench line)", m), ("S-cornell (`configs[1]`)", c), ("S-blob, 875 k triangles", b), ("S-mesh-10M, 10.5 M triangles", m10)):
        rr = d["roofline"]
        kk = rr["kernel_ms_per_step"]
        modes = d.get("bvh_modes") or {}
        lib = modes.get("library_tree_alone", {}).get("ms_per_step")
        ref = modes.get("reference_tree_alone", {}).get("ms_per_step")
        v = rr.get("valu") or {}
        rows.append(f"| {name} | **{d['ms_per_step']:.1f} ms** | {d['value']:.0f} | {('%.1f' % lib) if lib else '—'} / {('%.1f' % ref) if ref else '—'} | closest-hit {kk['trace_closest']:.1f}"
                    f" (fallback walk {rr['hybrid']['fallback_walk_ms_per_step']:.1f}; {100 * rr['fallback_fraction_of_closest_rays']:.3f} % of the rays), shade {kk['shade']:.1f}, any-hit {kk['trace_any']:.1f},"
                    f" film {kk['film']:.1f}, raygen {kk['raygen']:.1f} | `{rr['kernel']}`: {rr['avg_launch_ms']:.2f} ms per launch, {rr['frac_requests']:.2f} by requests, {rr['frac_counters']:.2f} by counters,"
                    f" {v.get('lanes_per_valu_inst', 0):.1f} of 64 lanes, VALU busy {100 * v.get('valu_busy', 0):.0f} %, `valu_frac` {rr.get('valu_frac')} | {d['cpu_baseline']['value']:.2f} |")
    sk = s["roofline"]["kernel_ms_per_step"]
    block = ("Round " + a.round[1:] + ", one MI355X (`profiles/" + a.round + "/" + a.tag + "_bench_*.json`; the driver's command for the first line; default configuration = exact):\n\n"
             "| workload | frame | Mray/s | library tree alone (not exact) / reference tree alone | kernel classes, ms per frame | dominant kernel | oracle, 128 cores, Mray/s |\n|---|---|---|---|---|---|---|\n" + "\n".join(rows) + "\n"
             f"| S-caustic SPPM (`caustic-glass.ply`, 100 iterations; `configs[3]`) | **{s['ms_per_step']:.1f} ms** | {s['value']:.0f} | — | closest-hit {sk['trace_closest']:.1f}, photon gather {sk['photon_gather']:.1f},"
             f" camera + photon shading {sk['camera+photon_shading']:.1f}, grid {sk['grid+bin+scan']:.1f}, any-hit {sk['trace_any']:.1f} | | {s['cpu_baseline']['value']:.2f} |\n"
             f"| C5 share per GPU (10.5 M triangles, 4096², 128 spp, depth 16) | **{c5['ms_per_step'] / 1e3:.2f} s** | {c5['value']:.0f} | — | closest-hit {c5['roofline']['kernel_ms_per_step']['trace_closest']:.0f} | | |\n\n"
             f"`hbm_resident` (inside the default command; child run on S-mesh-10M): {hr.get('ms_per_step')} ms per frame, `{hr.get('kernel')}` {hr.get('avg_launch_ms')} ms per launch, {hr.get('frac_requests')} by requests,"
             f" **{hr.get('frac_counters')} by counters** ({hr.get('achieved_counters_GBps')} GB/s left L2), {(hr.get('valu') or {}).get('lanes_per_valu_inst')} of 64 lanes at VALU busy"
             f" {(hr.get('valu') or {}).get('valu_busy')}: `valu_frac` {hr.get('valu_frac')} — VALU issue bounds the walk on the scene that does not fit the caches as well.")
    readme = (f"S-mesh (1 M triangles) **{m['value']:.0f} Mray/s** ({frame:.1f} ms per frame; the library's tree alone, not exact: {m['bvh_modes']['library_tree_alone']['ms_per_step']:.1f} ms; the reference's tree alone:"
              f" {m['bvh_modes']['reference_tree_alone']['ms_per_step']:.1f} ms), S-cornell {c['value']:.0f} ({c['ms_per_step']:.1f} ms), S-blob {b['value']:.0f} ({b['ms_per_step']:.1f} ms), 10.5 M triangles {m10['value']:.0f}"
              f" ({m10['ms_per_step']:.1f} ms), SPPM on the reference's `caustic-glass.ply` {s['ms_per_step']:.1f} ms per 100 iterations; the oracle on 128 host cores: {m['cpu_baseline']['value']:.2f} Mray/s on S-mesh.")
    rep = {
        "RN_MESH_MS": f"{frame:.1f}", "RN_LIB_MS": f"{m['bvh_modes']['library_tree_alone']['ms_per_step']:.1f}", "RN_CORNELL_CLOSEST": f"{c['roofline']['kernel_ms_per_step']['trace_closest']:.1f} ms",
        "RN_FB_SHARE": f"{100 * fb / k['trace_closest']:.1f} %", "RN_C4_MS": f"{s['ms_per_step']:.1f}", "RN_SO_MB": f"{a.so_mb:.2f}", "RN_GPU_S": f"{a.gpu_suite_s:.0f}",
        "RN_10M_COMMIT": f"{m10['config']['bvh_build_upload_s']:.1f}", "RN_BYTES_RAY": f"{bytes_ray:.0f}", "RN_SHARE_TRACE": pct(k["trace_closest"] - fb), "RN_SHARE_FB": pct(fb),
        "RN_SHARE_SHADE": pct(k["shade"]), "RN_SHARE_ANY": f"overlapped: {k['trace_any']:.0f} ms of wall time on the second stream, beside the closest-hit rays of the next depth", "RN_SHARE_FILM": pct(k["film"]), "RN_SHARE_RAYGEN": pct(k["raygen"]), "RN_LANES": f"{r['valu']['lanes_per_valu_inst']:.1f}",
        "RN_VALU_FRAC": f"{r['valu_frac']}", "RN_VALU_BUSY": f"{100 * r['valu']['valu_busy']:.0f} %", "RN_WAVE_INSTS": f"{r['valu']['wave_valu_insts_per_launch'] / 1e9:.1f}e9", "RN_VALU_MS": f"{r['valu']['wave_valu_insts_per_launch'] * 4 / 1024 / 2.4e9 * 1e3:.1f}",
        "RN_LAUNCH_MS": f"{r['avg_launch_ms']:.1f}", "RN_MEASURED_BLOCK": block, "RN_README_NUMBERS": readme,
        "RN_HBM_MS": f"{hr.get('ms_per_step', 0):.1f}", "RN_CORNELL_MS": f"{c['ms_per_step']:.1f}", "RN_FULL_FRAME": str(m["parity"].get("full_frame_equal")),
    }
    for src, dst in (("docs/templates/DESIGN.in.md", "DESIGN.md"), ("docs/templates/README.in.md", "README.md")):
        t = open(os.path.join(ROOT, src)).read()
        for key in sorted(rep, key=len, reverse=True):
            t = t.replace(key, rep[key])
        left = [w for w in t.split() if w.startswith("RN_")]
        assert not left, left
        open(os.path.join(ROOT, dst), "w").write(t)
        print(dst, len(t))


if __name__ == "__main__":
    main()
