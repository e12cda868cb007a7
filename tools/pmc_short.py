#!/usr/bin/env python3
"""Per (kernel, grid) summary of rocprofv3 PMC passes for SHORT dispatches (no 1 ms floor): duration, effective clock
(GRBM_GUI_ACTIVE / 8 / duration — reads high below ~0.3 ms), MFMA pipe busy and the wavefront-cycle split.
usage: tools/pmc_short.py <counter_collection.csv> [...]   (one csv per --pmc pass, same command)"""
import csv
import re
import sys
from collections import defaultdict

per = defaultdict(lambda: defaultdict(list))
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        n = r["Kernel_Name"]
        if "conv" not in n and "attn" not in n:
            continue
        m = re.search(r"(\w+<[^>]*>|\w+)\(", n.replace("(anonymous namespace)::", ""))
        short = m.group(1) if m else n[:44]
        key = (short, r.get("Grid_Size", "?"))
        per[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        per[key]["_dur"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9)
print("%-44s %9s %4s %8s %6s %7s | %6s %6s %6s" % ("kernel", "grid", "n", "us", "GHz", "MFMA%", "waitI%", "wait%", "act%"))
for key in sorted(per, key=lambda k: (k[0], int(k[1]) if k[1].isdigit() else 0)):
    cs = per[key]
    mean = lambda c: sum(cs[c]) / len(cs[c]) if c in cs else None
    dur = mean("_dur")
    g = mean("GRBM_GUI_ACTIVE")
    clk = g / 8.0 / dur if g else None
    mf = mean("SQ_VALU_MFMA_BUSY_CYCLES")
    busy = 100.0 * mf / ((clk or 2.1e9) * dur * 4 * 256) if mf else None
    wc = mean("SQ_WAVE_CYCLES")
    f = lambda c: ("%6.1f" % (100.0 * mean(c) / wc)) if (wc and mean(c) is not None) else "     -"
    print("%-44s %9s %4d %8.1f %6s %7s | %s %s %s" % (key[0][:44], key[1], len(cs["_dur"]), dur * 1e6,
          ("%.2f" % (clk / 1e9)) if clk else "-", ("%.1f" % busy) if busy else "-",
          f("SQ_WAIT_INST_ANY"), f("SQ_WAIT_ANY"), f("SQ_ACTIVE_INST_ANY")))
