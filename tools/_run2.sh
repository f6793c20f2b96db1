#!/bin/bash
O=gpurun_out/r4z; mkdir -p $O
for v in libtracehip lib_a; do
TRHIP_LIB=$PWD/trace.jl_amd/$v.so timeout 300 python tools/hybrid_probe.py --workload cornell --spp 64 --check-spp 2 --skip-library > $O/probe_c_$v.json 2>/dev/null < /dev/null; echo $v $(grep -E "closest_ms|fallback_fraction|differing" $O/probe_c_$v.json)
TRHIP_LIB=$PWD/trace.jl_amd/$v.so timeout 300 python tools/hybrid_probe.py --workload mesh_1m --spp 64 --check-spp 1 --skip-library > $O/probe_m_$v.json 2>/dev/null < /dev/null; echo $v $(grep -E "closest_ms" $O/probe_m_$v.json)
done
