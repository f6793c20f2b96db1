#!/bin/bash
# tools/calib/run_fetch_calib.sh — on the GPU box (gpurun): build fetch_calib, run it alone and under rocprofv3 --pmc FETCH_SIZE (its own pass, no trace
# domains), print per kernel: known bytes, FETCH_SIZE x 1024, and the factor bytes / FETCH_SIZE.  Output: gpurun_out/fetch_calib/summary.txt
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/fetch_calib
mkdir -p $OUT
cd $GRAFT_REPO_ROOT/tools/calib && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o fetch_calib fetch_calib.hip || exit 1
cd /tmp && export TMPDIR=/tmp
$GRAFT_REPO_ROOT/tools/calib/fetch_calib > $OUT/plain.jsonl 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $GRAFT_REPO_ROOT/tools/calib/fetch_calib > $OUT/under_pmc.jsonl 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $OUT/pmc_tcc -- $GRAFT_REPO_ROOT/tools/calib/fetch_calib > $OUT/under_pmc_tcc.jsonl 2>&1
cd $GRAFT_REPO_ROOT && python3 - <<'PY' > $OUT/summary.txt 2>&1
import csv, glob, json, os
out = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "fetch_calib")
known = {}
for l in open(os.path.join(out, "plain.jsonl")):
    if l.startswith("{"):
        r = json.loads(l)
        known[r["kernel"].split(" ")[0].split("<")[0] + (r["kernel"].split(">")[0].split("<")[1] if "<" in r["kernel"] else "")] = r
        print(l.strip())
for sub in ("pmc_fetch", "pmc_tcc"):
    for f in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
        acc = {}
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            acc.setdefault((k, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
        for (k, c), v in sorted(acc.items()):
            key = k.split("<")[0] + (k.split("<")[1].split(">")[0] if "<" in k else "")
            b = known.get(key, {}).get("bytes")
            line = f"{k:24s} {c:24s} per dispatch {v}"
            if c == "FETCH_SIZE" and b:
                line += f"   known bytes {b}   FETCH_SIZE x 1024 = {v[-1] * 1024:.0f}   factor bytes / FETCH_SIZE = {b / (v[-1] * 1024):.3f}"
            print(line)
PY
cat $OUT/summary.txt
