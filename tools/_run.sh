#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu > $O/r4g_pytest_gpu_full_suite.txt 2>&1 < /dev/null; grep -E "passed|failed" $O/r4g_pytest_gpu_full_suite.txt | tail -1
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r4g_bench_mesh1m.json 2>/dev/null < /dev/null; cut -c1-330 $O/r4g_bench_mesh1m.json
timeout 900 python bench.py --workload caustic_sppm --steps 5 --warmup 2 > $O/r4g_bench_caustic_sppm.json 2>/dev/null < /dev/null; cut -c1-330 $O/r4g_bench_caustic_sppm.json
