#!/bin/bash
# tools/kstats.sh <tag> <workload> <spp>: rocprofv3 --kernel-trace --stats of one frame (tools/option_sweep.py); prints the per-kernel table.  Run on the GPU box.
TAG=$1; W=$2; S=$3
ROOT=$PWD; D=$ROOT/gpurun_out/kstats_$TAG; mkdir -p $D
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats -- python3 $ROOT/tools/option_sweep.py --workload $W --spp $S --repeat 2 > $D/log.txt 2>&1
python3 $ROOT/tools/summarize_pmc.py $D | head -24 | cut -c1-140
rm -rf $D/stats
