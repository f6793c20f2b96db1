#!/bin/bash
O=gpurun_out/r4z; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_hybrid.py tests/test_gpu_edge_cases.py -x -q -m gpu 2>&1 | tail -3
for v in libtracehip lib_a; do
TRHIP_LIB=$PWD/trace.jl_amd/$v.so timeout 300 python tools/hybrid_probe.py --workload cornell --spp 64 --check-spp 2 --skip-library > $O/probe_c_$v.json 2>/dev/null < /dev/null; echo $v $(grep -E "closest_ms|fallback_fraction|differing" $O/probe_c_$v.json)
done
timeout 600 python tools/hybrid_probe.py --workload blob_870k --spp 64 --check-spp 2 > $O/probe_blob.json 2>$O/probe_blob.err < /dev/null; grep -E "bvh_mode|frame_ms|closest_ms|fallback_fraction|differing|accelerator_nodes" $O/probe_blob.json
