#!/bin/bash
# tools/ab_libs.sh <tag> <workload> <spp> "<hybrid_probe extra args>" [lib names...]: as sweep_libs.sh, with extra arguments for tools/hybrid_probe.py (e.g. "--opt overlap=0":
# kernel classes one after the other, so that a class time is the kernel's own) and WITHOUT the in-tree run unless "intree" is named.
TAG=$1; W=$2; S=$3; X=$4; shift 4
O=gpurun_out; mkdir -p $O
for n in "$@"; do
  if [ "$n" = intree ]; then unset TRHIP_LIB; else export TRHIP_LIB=$PWD/_diag/lib_$n.so; fi
  timeout 600 python tools/hybrid_probe.py --workload $W --spp $S --check-spp 4 --count --skip-library $X > $O/${TAG}_$n.json 2> $O/${TAG}_$n.err < /dev/null
  python - "$O/${TAG}_$n.json" "$n" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    h = d["modes"]["hybrid (default)"]; r = d["modes"]["reference tree alone (hybrid 0)"]; e = d["hybrid_equals_reference_tree"]
    print(f"{sys.argv[2]:>18}: frame {h['frame_ms']:8.2f} closest {h['closest_ms']:8.2f} any {h['any_ms']:6.2f} shade {h['shade_ms']:6.2f} fb {h['fallback_fraction']:.5f} boxes {h.get('boxes_per_closest_ray')} prims {h.get('prims_per_closest_ray')} | ref-tree closest {r['closest_ms']:8.2f} | differ film {e['film_values_differing']} samples {e['sample_values_differing']}")
except Exception as ex:
    print(f"{sys.argv[2]:>18}: FAILED {ex}")
PY
done
