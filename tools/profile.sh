#!/bin/bash
# tools/profile.sh <tag> <bench args...>   — run on the GPU box (via gpurun): kernel-trace stats + PMC passes for bench.py.
# Writes gpurun_out/prof_<tag>/{stats,pmc_*}; copy the summaries you want judged into profiles/.
# PMC passes never share a run with trace domains; FETCH_SIZE and WRITE_SIZE need separate passes (TCC slots).
set -u
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="$GRAFT_REPO_ROOT/bench.py --no-traffic --no-cpu-baseline --no-micro --no-hbm-resident --no-modes"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $B "$@" > $OUT/bench_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $B "$@" > $OUT/bench_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $B "$@" > $OUT/bench_pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES --output-format csv -d $OUT/pmc_sq -- python3 $B "$@" > $OUT/bench_pmc_sq.log 2>&1
rocprofv3 --pmc VALUBusy MemUnitStalled SALUBusy TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_busy -- python3 $B "$@" > $OUT/bench_pmc_busy.log 2>&1
cd $GRAFT_REPO_ROOT && python tools/summarize_pmc.py gpurun_out/prof_$TAG > $OUT/summary.txt 2>&1
grep "^{\"metric\"" $OUT/bench_stats.log | tail -1 > $OUT/bench_line.json
ls $OUT
