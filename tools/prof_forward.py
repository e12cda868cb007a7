#!/usr/bin/env python3
"""Per-dispatch breakdown of ONE forward pass from a rocprofv3 --kernel-trace CSV (eager bench run).
usage: tools/prof_forward.py <kernel_trace.csv> [--all]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "head_act_mean" in r["Kernel_Name"]]
a, b = idx[-2] + 1, idx[-1] + 1
tot = 0.0
agg = {}
for r in rows[a:b]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    n = r["Kernel_Name"]
    short = n.split("::")[1].split("(")[0] if "anonymous" in n else n[:50]
    tot += d
    k = agg.setdefault(short, [0, 0.0])
    k[0] += 1
    k[1] += d
    if "--all" in sys.argv:
        print("%-40s grid=%-7d vgpr=%-4s lds=%-6s %9.1f us" % (short, int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]),
                                                           r["VGPR_Count"], r["LDS_Block_Size"], d))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-44s calls=%-4d %9.1f us  %5.1f%%" % (k, v[0], v[1], 100 * v[1] / tot))
print("sum of kernels %.1f us; span %.1f us" % (tot, (int(rows[b - 1]["End_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e3))
