#!/bin/bash
O=gpurun_out/r4z; mkdir -p $O
for cfg in "libtracehip 0" "libtracehip 1" "lib_a 1"; do set -- $cfg
TRHIP_LIB=$PWD/trace.jl_amd/$1.so timeout 300 python tools/hybrid_probe.py --workload mesh_1m --spp 128 --check-spp 1 --skip-library --opt leaf_queue=$2 > $O/probe_q.json 2>/dev/null < /dev/null; echo $1 leaf_queue=$2 $(grep -E "closest_ms|differing" $O/probe_q.json | head -1) $(grep differing $O/probe_q.json | head -1)
done
