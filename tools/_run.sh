#!/bin/bash
O=gpurun_out/r4l; mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -q --deselect tests/test_gpu_soak.py 2>&1 | tail -150 > $O/pytest_gpu.txt; grep -E "^FAILED|^ERROR|passed|failed" $O/pytest_gpu.txt
