#!/bin/bash
# tools/matrix.sh <outdir> — frame-level A/B of traversal kernels / BVH topologies (bench.py lines without the CPU / PMC legs)
OUT=$1; mkdir -p $OUT
run() { tag=$1; shift; timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-micro --no-traffic "$@" > $OUT/$tag.json 2>$OUT/$tag.err; python - "$OUT/$tag.json" "$tag" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d['roofline']
    print(sys.argv[2], d['ms_per_step'], r['kernel_ms_per_step'], r['visits_per_ray'])
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
}
run mesh1m_c0_t3 --opt compose_spheres=0 --traversal 3
run mesh1m_c1_t3 --traversal 3
run mesh1m_c1_t4 --traversal 4
run blob_t3 --workload blob_870k --traversal 3
run blob_t4 --workload blob_870k --traversal 4
