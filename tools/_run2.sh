#!/bin/bash
O=gpurun_out/r4z; mkdir -p $O
for t in 16 0; do
timeout 300 python tools/hybrid_probe.py --workload cornell --spp 64 --check-spp 2 --opt tiny_scene_prims=$t > $O/probe_tiny$t.json 2>/dev/null < /dev/null; echo tiny $t $(grep -E "closest_ms|accelerator_nodes|differing" $O/probe_tiny$t.json)
done
