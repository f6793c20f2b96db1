#!/bin/bash
# tools/sweep_libs.sh <tag> <workload> <spp> [lib names...]: tools/hybrid_probe.py (hybrid vs the reference's tree alone: times, fallback, visits, bit equality) for the in-tree
# library and every named _diag/lib_<name>.so; one JSON per library under gpurun_out/<tag>_<name>.json and a one-line summary each.  Run on the GPU box.
TAG=$1; W=$2; S=$3; shift 3
O=gpurun_out; mkdir -p $O
run() {  # name, lib path or ""
  if [ -n "$2" ]; then export TRHIP_LIB=$PWD/$2; else unset TRHIP_LIB; fi
  timeout 600 python tools/hybrid_probe.py --workload $W --spp $S --check-spp 4 --count --skip-library > $O/${TAG}_$1.json 2> $O/${TAG}_$1.err < /dev/null
  python - "$O/${TAG}_$1.json" "$1" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    h = d["modes"]["hybrid (default)"]; r = d["modes"]["reference tree alone (hybrid 0)"]; e = d["hybrid_equals_reference_tree"]
    print(f"{sys.argv[2]:>18}: frame {h['frame_ms']:8.2f} closest {h['closest_ms']:8.2f} any {h['any_ms']:6.2f} shade {h['shade_ms']:6.2f} fb {h['fallback_fraction']:.5f} boxes {h.get('boxes_per_closest_ray')} prims {h.get('prims_per_closest_ray')} why {h.get('fallback_why (direction, sphere, tie/guard, unknown entry)')} | ref-tree closest {r['closest_ms']:8.2f} | differ film {e['film_values_differing']} samples {e['sample_values_differing']}")
except Exception as ex:
    print(f"{sys.argv[2]:>18}: FAILED {ex}")
PY
}
run intree ""
for n in "$@"; do run $n _diag/lib_$n.so; done
