#!/bin/bash
# tools/final_runs.sh <tag>  — on the GPU box: the bench lines and the rocprof summaries a round's profiles/ directory keeps.
set -u
TAG=$1
O=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
timeout -k 5 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/${TAG}_bench_mesh1m.json 2> $O/${TAG}_bench_mesh1m.err
for wl in cornell blob_870k mesh_10m caustic_sppm; do
  timeout -k 5 600 python bench.py --workload $wl --steps 5 --warmup 2 --no-hbm-resident > $O/${TAG}_bench_$wl.json 2> $O/${TAG}_bench_$wl.err
done
timeout -k 5 900 bash tools/profile.sh $TAG --steps 2 --warmup 1 > $O/${TAG}_profile.log 2>&1
cd $GRAFT_REPO_ROOT
timeout -k 5 900 python bench.py --workload mesh_10m --res 4096 --spp 128 --depth 16 --steps 2 --warmup 1 --no-hbm-resident > $O/${TAG}_bench_c5_share.json 2> $O/${TAG}_bench_c5_share.err
timeout -k 5 600 python tools/soak_attack.py > $O/${TAG}_soak_attack.txt 2>&1
tail -c 600 $O/${TAG}_bench_mesh1m.json
