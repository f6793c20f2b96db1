#!/bin/bash
O=gpurun_out/r4z; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_hybrid.py tests/test_gpu_sppm.py tests/test_gpu_scale.py -x -q -m gpu 2>&1 | tail -2
timeout 600 python tools/soak_sppm.py --scenes 30 2>&1 | tail -1
timeout 600 python bench.py --workload caustic_sppm --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['roofline']['kernel_ms_per_step'])"
