#!/usr/bin/env python3
"""Effective shader clock and MFMA-pipe occupancy of the long kernels from two rocprofv3 PMC passes
(GRBM_GUI_ACTIVE; SQ_VALU_MFMA_BUSY_CYCLES + SQ_BUSY_CU_CYCLES), per kernel name (mean over dispatches >= 1 ms):
  clock  = GRBM_GUI_ACTIVE / 8 XCDs / duration           (MI355X_MICROARCH.md, "DVFS give-back")
  mfma % = SQ_VALU_MFMA_BUSY_CYCLES / (clock cycles of the dispatch x 4 SIMDs x 256 CUs)
usage: tools/pmc_clock_mfma.py <grbm_counter_collection.csv> <sq_counter_collection.csv> [min dispatch ms]"""
import csv
import sys
from collections import defaultdict


MIN_S = float(sys.argv[3]) * 1e-3 if len(sys.argv) > 3 else 1e-3  # optional 3rd argument: minimum dispatch ms


def load(path):
    per = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(path)):
        n = r["Kernel_Name"]
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
        if dur < MIN_S:
            continue
        short = n.split("::")[1].split("(")[0] if "anonymous" in n else n[:50]
        per[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
        per[short]["_dur"].append(dur)
    return per


g, q = load(sys.argv[1]), load(sys.argv[2])
print("%-40s %6s %8s %10s %10s" % ("kernel (dispatches >= %.2f ms)" % (MIN_S * 1e3), "n", "ms", "clock GHz", "MFMA busy"))
for k in sorted(g, key=lambda k: -sum(g[k]["_dur"])):
    if "GRBM_GUI_ACTIVE" not in g[k] or k not in q or "SQ_VALU_MFMA_BUSY_CYCLES" not in q[k]:
        continue
    dur = sum(g[k]["_dur"]) / len(g[k]["_dur"])
    clk = sum(g[k]["GRBM_GUI_ACTIVE"]) / len(g[k]["GRBM_GUI_ACTIVE"]) / 8.0 / dur
    durq = sum(q[k]["_dur"]) / len(q[k]["_dur"])
    busy = sum(q[k]["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(q[k]["SQ_VALU_MFMA_BUSY_CYCLES"])
    print("%-40s %6d %8.3f %10.2f %9.1f%%" % (k, len(g[k]["_dur"]), dur * 1e3, clk / 1e9,
                                              100.0 * busy / (clk * durq * 4 * 256)))
