#!/bin/bash
O=$PWD/gpurun_out/r4z; mkdir -p $O
R=$PWD
cd /tmp && export TMPDIR=/tmp
for lq in 0 1; do
rm -rf $O/pmc_lq$lq
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d $O/pmc_lq$lq -- python3 $R/tools/hybrid_probe.py --workload mesh_1m --spp 16 --check-spp 1 --skip-library --opt leaf_queue=$lq > $O/pmc_lq$lq.log 2>&1 < /dev/null
done
python3 - $O <<'PY'
import csv,sys,collections,glob
for lq in (0,1):
    f=glob.glob(sys.argv[1]+"/pmc_lq%d/*/*counter_collection.csv"%lq)[0]
    acc=collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        if 'k_trace3c<false' in k or 'k_trace3d<false' in k: acc[k[:44]][r['Counter_Name']]+=float(r['Counter_Value'])
    for k,v in acc.items():
        lanes=v['SQ_THREAD_CYCLES_VALU']/max(1,v['SQ_ACTIVE_INST_VALU'])
        print(lq,k,{a:f"{b:.3e}" for a,b in v.items()},"lanes/VALU instr %.1f"%lanes)
PY
