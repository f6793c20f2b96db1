#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout 2400 python tools/soak_hybrid.py --scenes 240 --rays 400000 --frames 48 --seed 21 > $O/r4c_soak_hybrid_240.txt 2>$O/r4c_soak_hybrid.err < /dev/null; tail -1 $O/r4c_soak_hybrid_240.txt
timeout 1500 python tools/soak_parity.py --scenes 120 > $O/r4c_soak_parity.txt 2>&1 < /dev/null; tail -1 $O/r4c_soak_parity.txt
