#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu > $O/r4a_pytest_gpu_full_suite.txt 2>&1 < /dev/null; tail -3 $O/r4a_pytest_gpu_full_suite.txt
timeout 1500 python tools/soak_hybrid.py --scenes 60 --rays 400000 --frames 48 --seed 3 > $O/r4a_soak_hybrid.txt 2>$O/r4a_soak_hybrid.err < /dev/null; tail -1 $O/r4a_soak_hybrid.txt
bash tools/final_runs.sh r4a < /dev/null
