#!/bin/bash
O=gpurun_out/r4z; mkdir -p $O
for rep in 1 2; do
for v in libtracehip lib_a lib_b; do
TRHIP_LIB=$PWD/trace.jl_amd/$v.so timeout 300 python tools/hybrid_probe.py --workload mesh_1m --spp 128 --check-spp 1 --skip-library > $O/probe_$v.json 2>/dev/null < /dev/null; echo $rep $v $(grep -E "closest_ms|frame_ms|differing" $O/probe_$v.json | head -2)
done
done
timeout 300 python tools/hybrid_probe.py --workload cornell --spp 64 --check-spp 2 --skip-library 2>/dev/null | grep -E "closest_ms|differing"
