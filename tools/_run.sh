#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu > $O/r4d_pytest_gpu_full_suite.txt 2>&1 < /dev/null; grep -E "passed|failed" $O/r4d_pytest_gpu_full_suite.txt | tail -1
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 900 python bench.py > $O/r4d_bench_default.json 2>$O/r4d_bench_default.err < /dev/null; cut -c1-400 $O/r4d_bench_default.json
