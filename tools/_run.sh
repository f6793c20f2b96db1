#!/bin/bash
O=gpurun_out/r4z; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_hybrid.py tests/test_gpu_edge_cases.py tests/test_gpu_reference_tree.py -x -q -m gpu > $O/pytest.log 2>&1 < /dev/null; tail -3 $O/pytest.log
timeout 900 python tools/soak_hybrid.py --scenes 24 --rays 300000 --frames 48 > $O/soak.txt 2>$O/soak.err < /dev/null; tail -2 $O/soak.txt
timeout 300 python tools/hybrid_probe.py --workload mesh_1m --spp 256 --check-spp 2 > $O/probe256.json 2>/dev/null < /dev/null; grep -E "frame_ms|closest_ms|differing" $O/probe256.json
