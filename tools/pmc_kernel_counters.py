#!/usr/bin/env python3
"""Mean per-dispatch PMC counters of kernels longer than 1 ms from any number of rocprofv3 counter_collection CSVs
(one --pmc pass each), merged by kernel name.  usage: tools/pmc_kernel_counters.py <csv> [<csv> ...] [name-filter]"""
import csv
import sys
from collections import defaultdict

files = [a for a in sys.argv[1:] if a.endswith(".csv")]
flt = next((a for a in sys.argv[1:] if not a.endswith(".csv")), "")
per = defaultdict(lambda: defaultdict(list))
for path in files:
    for r in csv.DictReader(open(path)):
        n = r["Kernel_Name"]
        if flt and flt not in n:
            continue
        if (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) < 1e6:
            continue
        short = n.split("::")[1].split("(")[0] if "anonymous" in n else n[:50]
        per[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in per.items():
    print(k)
    wc = None
    for c, v in sorted(cs.items()):
        m = sum(v) / len(v)
        if c == "SQ_WAVE_CYCLES":
            wc = m
        print("    %-30s %18.0f" % (c, m))
    if wc:
        for c in ("SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_ANY",
                  "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_MISC"):
            if c in cs:
                print("    %-30s %17.1f%% of SQ_WAVE_CYCLES" % (c, 100.0 * sum(cs[c]) / len(cs[c]) / wc))
