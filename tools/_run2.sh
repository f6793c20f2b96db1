#!/bin/bash
O=gpurun_out/r4z; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_sppm.py tests/test_gpu_comm.py -x -q -m gpu 2>&1 | tail -2
M=tests/golden/caustic-glass.ply
for v in libtracehip lib_a lib_b lib_c; do
TRHIP_LIB=$PWD/trace.jl_amd/$v.so timeout 300 python tools/sppm_bench.py --model $M > $O/sppm_$v.json 2>/dev/null < /dev/null; python - $O/sppm_$v.json $v <<'PY'
import json,sys
for line in open(sys.argv[1]):
    if line.startswith("{"):
        d=json.loads(line); print(sys.argv[2], d['ms_total'], d['kernel_ms'])
PY
done
timeout 900 python tools/soak_sppm.py --scenes 40 2>&1 | tail -1
