#!/bin/bash
# tools/prof_trace.sh <tag> <trace_bench args...> — run on the GPU box: kernel stats + SQ / cache counters of the traversal kernels alone
# (tools/trace_bench.py).  Writes gpurun_out/prof_<tag>/summary.txt.  PMC passes never share a run with trace domains.
set -u
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="$GRAFT_REPO_ROOT/tools/trace_bench.py"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $B "$@" > $OUT/stats.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD --output-format csv -d $OUT/pmc_sq -- python3 $B "$@" > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAVES TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_mem -- python3 $B "$@" > $OUT/pmc_mem.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $B "$@" > $OUT/pmc_fetch.log 2>&1
cd $GRAFT_REPO_ROOT && python tools/summarize_pmc.py gpurun_out/prof_$TAG > $OUT/summary.txt 2>&1
grep -E "k_trace|counters|kernel stats" $OUT/summary.txt | head -60
