#!/usr/bin/env python
"""Durations of one kernel over a rocprofv3 --kernel-trace run, in dispatch order:   python tools/trace_series.py <kernel_trace.csv> <name substring> [every]
(how a kernel's cost develops over the iterations of an SPPM render), and the per-kernel totals of the run."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
pat = sys.argv[2]
every = int(sys.argv[3]) if len(sys.argv) > 3 else 10
tot = defaultdict(lambda: [0, 0.0])
series = []
for r in rows:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    name = r["Kernel_Name"].replace("void th::", "").replace("th::", "").split("(")[0]
    tot[name][0] += 1
    tot[name][1] += d
    if pat in name:
        series.append(d)
print("calls", len(series), "of", pat)
print(" ".join(f"{x:.0f}" for x in series[::every]))
span = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e6
print(f"trace span {span:.1f} ms")
for name, (n, ms) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"{name[:44]:44s} calls {n:6d} total {ms / 1e3:9.2f} ms avg {ms / n:9.1f} us")
