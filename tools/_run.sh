#!/bin/bash
O=$PWD/gpurun_out/r4z; mkdir -p $O
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf $O/pmc_film2
for sw in 0 1; do
cat > /tmp/fp.py <<PY
import sys; sys.path.insert(0, "$R")
import __graft_entry__ as g
T = g.load_package()
scene, cam = T.scenes.cornell_scene(), T.scenes.cornell_camera(1024)
ctx = T.default_context()
ctx.set_option("film_swizzle", $sw)
integ = T.PathIntegrator(cam, T.SeededSampler(256, seed=1), 8)
integ.render(scene, ctx)
integ.render(scene, ctx)
print("swizzle", $sw, "ms_film", integ.stats.ms_film)
PY
python3 /tmp/fp.py 2>/dev/null | tail -1
timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCP_TCC_READ_REQ_sum --output-format csv -d $O/pmc_film2/c$sw -- python3 /tmp/fp.py > $O/pmc_film2_$sw.log 2>&1 < /dev/null
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_film2/f$sw -- python3 /tmp/fp.py >> $O/pmc_film2_$sw.log 2>&1 < /dev/null
done
python3 - $O/pmc_film2 <<'PY'
import csv,sys,collections,glob
for f in sorted(glob.glob(sys.argv[1]+"/*/*/*counter_collection.csv")):
    acc=collections.defaultdict(float); n=0
    for r in csv.DictReader(open(f)):
        if 'k_film_gather_packed' in r['Kernel_Name']: acc[r['Counter_Name']]+=float(r['Counter_Value']); n+=1
    print(f.split('/')[-3], n, {a:int(b) for a,b in acc.items()})
PY
