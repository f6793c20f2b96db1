#!/bin/bash
O=gpurun_out/r4z; mkdir -p $O
for nl in 0 1; do
for b in 0 -1; do
timeout 600 python bench.py --workload mesh_10m --steps 3 --warmup 1 --no-cpu-baseline --no-micro --no-modes --opt node_layout=$nl --opt bvh_builder=$b > $O/nl${nl}_b${b}.json 2>$O/nl.err < /dev/null
python - $O/nl${nl}_b${b}.json $nl $b <<'PY'
import json,sys
for line in open(sys.argv[1]):
    if line.startswith("{"):
        d=json.loads(line); r=d["roofline"]
        print("node_layout",sys.argv[2],"bvh_builder",sys.argv[3],"ms",d["ms_per_step"],"closest",d["roofline"]["kernel_ms_per_step"]["trace_closest"],"traffic",r.get("traffic"),"frac_counters",r.get("frac_counters"))
PY
done
done
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-micro --no-modes --no-traffic --opt node_layout=1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('mesh_1m node_layout 1', d['ms_per_step'], d['roofline']['kernel_ms_per_step'])"
timeout 300 python -m pytest tests/test_gpu_hybrid.py -x -q -m gpu 2>&1 | tail -2
