#!/bin/bash
# tools/pmc_libs.sh <tag> <workload> <spp> [lib names...]: SQ counters of the in-tree library and of every named _diag/lib_<name>.so, one frame of the workload each
# (tools/option_sweep.py under rocprofv3 --pmc, two passes).  Writes gpurun_out/pmc_<tag>_<name>/ and prints the k_trace3c / k_trace3 rows.  Run on the GPU box.
TAG=$1; W=$2; S=$3; shift 3
ROOT=$PWD; O=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
run() {
  if [ -n "$2" ]; then export TRHIP_LIB=$ROOT/$2; else unset TRHIP_LIB; fi
  D=$O/pmc_${TAG}_$1; mkdir -p $D
  rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $D/pmc_sq -- python3 $ROOT/tools/option_sweep.py --workload $W --spp $S --repeat 1 > $D/log_sq.txt 2>&1
  rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY SQ_INSTS_FLAT SQ_ACTIVE_INST_SCA --output-format csv -d $D/pmc_wait -- python3 $ROOT/tools/option_sweep.py --workload $W --spp $S --repeat 1 > $D/log_wait.txt 2>&1
  # (round 6) the L1 side: address / tag processing of the texture path; a pass that names a counter this rocprofv3 does not know fails alone
  rocprofv3 --pmc TA_BUSY_avr TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum --output-format csv -d $D/pmc_l1 -- python3 $ROOT/tools/option_sweep.py --workload $W --spp $S --repeat 1 > $D/log_l1.txt 2>&1
  echo "== $1"; python3 $ROOT/tools/summarize_pmc.py $D | grep -E "${PMC_PAT:-k_trace3c<false|k_trace3<false}" | cut -c1-600
  rm -rf $D/pmc_sq $D/pmc_wait $D/pmc_l1
}
run intree ""
for n in "$@"; do run $n _diag/lib_$n.so; done
