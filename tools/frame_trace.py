#!/usr/bin/env python
"""Kernel timeline of the last frame of a rocprofv3 --kernel-trace run:   python tools/frame_trace.py <kernel_trace.csv> [n_kernels]
Prints every dispatch with its duration and the idle gap since the previous dispatch ended, and the totals (busy / idle)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 80
last = rows[-n:]
prev = None
busy = idle = 0.0
for r in last:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev) / 1e3 if prev else 0.0
    name = r["Kernel_Name"].replace("void th::", "").replace("th::", "")[:48]
    print(f"{name:48s} dur {(e - s) / 1e3:9.1f} us  gap {gap:8.1f} us")
    busy += (e - s) / 1e3
    idle += max(gap, 0.0)
    prev = max(prev or 0, e)
print(f"total busy {busy / 1e3:.2f} ms, idle between dispatches {idle / 1e3:.2f} ms")
