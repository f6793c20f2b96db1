#!/bin/bash
for w in mesh64 blob24 cornell shadows; do echo "== $w"; timeout 600 python tools/option_exactness.py --workload $w 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"; done > gpurun_out/r4f_option_exactness.txt 2>&1
grep -E "DIFFERENT|refused|total:|==|not compared" gpurun_out/r4f_option_exactness.txt
