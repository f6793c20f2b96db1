#!/bin/bash
O=$PWD/gpurun_out/r4z; mkdir -p $O
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf $O/pmc_cornell
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $O/pmc_cornell -- python3 $R/tools/hybrid_probe.py --workload cornell --spp 16 --check-spp 1 > $O/pmc_cornell.log 2>&1 < /dev/null
f=$(find $O/pmc_cornell -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && python3 - "$f" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k=r['Kernel_Name'][:40]
    acc[k][r['Counter_Name']]+=float(r['Counter_Value'])
for k,v in acc.items():
    if 'leaf' in k: print(k, {a:int(b) for a,b in v.items()})
PY
