#!/bin/bash
O=gpurun_out/r4z; mkdir -p $O
timeout 600 python tools/tie_exposure.py --workload mesh_1m --res 1024 --spp 16 --depth 8 --time-spp 256 > $O/tie_mesh1m.json 2>$O/tie.err < /dev/null; cut -c1-2500 $O/tie_mesh1m.json
timeout 600 python tools/tie_exposure.py --workload cornell --res 1024 --spp 16 --depth 8 --time-spp 256 > $O/tie_cornell.json 2>>$O/tie.err < /dev/null; cut -c1-1800 $O/tie_cornell.json
timeout 600 python tools/tie_exposure.py --workload caustic --sppm --iterations 20 > $O/tie_caustic.json 2>>$O/tie.err < /dev/null; cut -c1-1500 $O/tie_caustic.json
tail -3 $O/tie.err
