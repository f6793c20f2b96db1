#!/bin/bash
O=gpurun_out/r4z; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_baseline_configs.py tests/test_gpu_hybrid.py -x -q -m gpu --durations=5 > $O/pytest3.log 2>&1 < /dev/null; tail -15 $O/pytest3.log
