#!/bin/bash
O=gpurun_out/r4z; mkdir -p $O
timeout 1700 python -m pytest tests/test_gpu_baseline_configs.py tests/test_gpu_hybrid.py tests/test_gpu_device_sah.py -x -q -m gpu --durations=6 > $O/pytest3.log 2>&1 < /dev/null; tail -15 $O/pytest3.log
