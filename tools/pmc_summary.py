#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc counter_collection CSV per kernel (mean per dispatch).
usage: tools/pmc_summary.py <counter_collection.csv> [name-filter]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
agg = defaultdict(lambda: defaultdict(list))
for r in rows:
    n = r["Kernel_Name"]
    if flt and flt not in n:
        continue
    short = n.split("::")[1].split("(")[0] if "anonymous" in n else n[:40]
    key = (short, r.get("Grid_Size", r.get("Grid_Size_X", "")))
    agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for key, cs in sorted(agg.items(), key=lambda kv: -sum(kv[1].get("SQ_WAVE_CYCLES", [0]))):
    n = len(next(iter(cs.values())))
    print("%-40s grid=%-9s n=%d" % (key[0], key[1], n))
    for c, v in sorted(cs.items()):
        print("      %-28s %16.0f" % (c, sum(v) / len(v)))
