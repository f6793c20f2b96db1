#!/usr/bin/env python
"""Summarise rocprofv3 CSV output (kernel stats + PMC counter passes) per kernel.   python tools/summarize_pmc.py gpurun_out/prof_<tag>"""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    n = name.split("(")[0].replace("void ", "").replace("th::", "")
    return n[:40]


def main(root):
    for f in glob.glob(os.path.join(root, "stats", "**", "*kernel_stats.csv"), recursive=True):
        print("== kernel stats", f)
        for r in csv.DictReader(open(f)):
            print(f"{short(r['Name']):40s} calls {int(r['Calls']):6d}  total_ms {int(r['TotalDurationNs'])/1e6:10.3f}  avg_us {float(r['AverageNs'])/1e3:10.2f}  {float(r['Percentage']):6.2f}%")
    for sub in sorted(glob.glob(os.path.join(root, "pmc_*"))):
        for f in glob.glob(os.path.join(sub, "**", "*counter_collection.csv"), recursive=True):
            acc = defaultdict(lambda: defaultdict(float))
            calls = defaultdict(int)
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
                calls[(k, r["Counter_Name"])] += 1
            print("== counters", f)
            for k in acc:
                print(f"{k:40s} " + "  ".join(f"{c}={v:.4g} (n={calls[(k, c)]})" for c, v in acc[k].items()))


if __name__ == "__main__":
    main(sys.argv[1])
