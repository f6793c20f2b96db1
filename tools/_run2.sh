#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_hybrid.py -x -q -m gpu 2>&1 | tail -3
