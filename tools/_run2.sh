#!/bin/bash
O=gpurun_out/r4z; mkdir -p $O
for v in libtracehip lib_a lib_b lib_c lib_d; do
TRHIP_LIB=$PWD/trace.jl_amd/$v.so timeout 300 python tools/hybrid_probe.py --workload mesh_1m --spp 64 --check-spp 1 --skip-library > $O/probe_$v.json 2>/dev/null < /dev/null; echo $v $(grep -E "closest_ms|differing" $O/probe_$v.json | head -1)
done
