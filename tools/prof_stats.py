#!/usr/bin/env python3
"""Per-step kernel-time table from a rocprofv3 kernel_stats CSV.  usage: prof_stats.py <stats.csv> <steps> [top]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]); top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:top]:
    n = r["Name"]
    short = n.split("::")[1].split("(")[0] if "anonymous" in n else n[:70]
    print("%-60s calls/step=%-7.1f %8.2f ms/step %5.1f%%" % (short, float(r["Calls"]) / steps, float(r["TotalDurationNs"]) / 1e6 / steps, 100 * float(r["TotalDurationNs"]) / tot))
print("total kernel time per step: %.2f ms" % (tot / 1e6 / steps))
