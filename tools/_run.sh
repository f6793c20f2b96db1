#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu > $O/r4e_pytest_gpu_full_suite.txt 2>&1 < /dev/null; grep -E "passed|failed" $O/r4e_pytest_gpu_full_suite.txt | tail -1
timeout 1500 python tools/soak_hybrid.py --scenes 120 --rays 400000 --frames 48 --seed 31 > $O/r4e_soak_hybrid.txt 2>/dev/null < /dev/null; tail -1 $O/r4e_soak_hybrid.txt
bash tools/final_runs.sh r4e < /dev/null
