#!/bin/bash
O=gpurun_out/r4z; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_hybrid.py tests/test_abi.py -x -q 2>&1 | tail -2
timeout 1500 python tools/soak_hybrid.py --scenes 40 --rays 300000 --frames 40 --seed 11 > $O/soak2.txt 2>$O/soak2.err < /dev/null; tail -1 $O/soak2.txt
timeout 300 python - <<'PY' 2>/dev/null
import __graft_entry__ as g, bench
T=g.load_package(); ctx=T.default_context()
scene,cam,desc=bench.build_workload(T,"mesh_10m",1024)
flat=scene.flatten(ctx); print("mesh_10m mode", flat.bvh_mode(), "note:", flat.bvh_note())
PY
