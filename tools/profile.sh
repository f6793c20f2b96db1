#!/bin/bash
# tools/profile.sh <tag> <bench args...>   — run on the GPU box (via gpurun): kernel-trace stats + PMC passes for bench.py.
# Writes gpurun_out/prof_<tag>/{stats,pmc_*}; copy the summaries you want judged into profiles/.
set -u
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $GRAFT_REPO_ROOT/bench.py "$@" > $OUT/bench_stats.log 2>&1
# PMC passes (no trace domains next to --pmc): HBM traffic and VALU / wait breakdown
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python $GRAFT_REPO_ROOT/bench.py "$@" > $OUT/bench_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python $GRAFT_REPO_ROOT/bench.py "$@" > $OUT/bench_pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_WAVES --output-format csv -d $OUT/pmc_sq -- python $GRAFT_REPO_ROOT/bench.py "$@" > $OUT/bench_pmc_sq.log 2>&1
ls -R $OUT | head -40
