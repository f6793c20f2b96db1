#!/usr/bin/env python
"""Gaps in a rocprofv3 --kernel-trace timeline: per queue, how long the GPU waited between consecutive kernels, and what ran.
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python3 bench.py --steps 1 --warmup 1 --no-traffic --no-cpu-baseline --no-micro --no-visits
    python tools/timeline.py gpurun_out/tl [last_n_kernels]"""
import csv, glob, sys, collections
d = sys.argv[1]
last = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rows = list(csv.DictReader(open(glob.glob(d + "/*/*kernel_trace.csv")[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-last:]
t0 = int(rows[0]["Start_Timestamp"])
prev_end = {}
busy_end = t0
idle = 0
for r in rows:
    s, e, q = int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"]
    gap_q = (s - prev_end[q]) / 1e3 if q in prev_end else 0.0
    if s > busy_end:
        idle += s - busy_end
    busy_end = max(busy_end, e)
    prev_end[q] = e
    print(f"{(s - t0) / 1e6:9.3f} ms  q{q}  {(e - s) / 1e3:9.1f} us  gap on its queue {gap_q:8.1f} us  {r['Kernel_Name'][:70]}")
print(f"window {(busy_end - t0) / 1e6:.3f} ms, GPU idle (no kernel on any queue) {idle / 1e6:.3f} ms")
