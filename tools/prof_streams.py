#!/usr/bin/env python3
"""Per-stream picture of one training step in a rocprofv3 kernel-trace CSV (steps are delimited by the optimizer's
multi-tensor kernels): busy time and the heaviest kernels of every HIP stream — which stream is the critical path.
usage: tools/prof_streams.py <kernel_trace.csv>"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:48]


idx = [i for i, r in enumerate(rows) if "multi_tensor_apply" in r["Kernel_Name"]]
bursts = []
for i in idx:
    if not bursts or i - bursts[-1][-1] > 50:
        bursts.append([i])
    else:
        bursts[-1].append(i)
a, b = bursts[-3][-1] + 1, bursts[-2][-1] + 1  # from the end of one optimizer step to the end of the next
ev = rows[a:b]
t0 = int(ev[0]["Start_Timestamp"])
t1 = max(int(r["End_Timestamp"]) for r in ev)
print("step span %.2f ms, %d kernels" % ((t1 - t0) / 1e6, len(ev)))
per = collections.defaultdict(list)
for r in ev:
    per[r["Stream_Id"]].append(r)
dur = lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for s, rs in sorted(per.items(), key=lambda kv: -sum(dur(r) for r in kv[1])):
    busy = sum(dur(r) for r in rs) / 1e6
    gaps = sum(max(0, int(rs[i + 1]["Start_Timestamp"]) - int(rs[i]["End_Timestamp"])) for i in range(len(rs) - 1))
    print("stream %s: %d kernels, busy %.2f ms, window %.2f..%.2f ms, gaps inside %.2f ms" % (
        s, len(rs), busy, (int(rs[0]["Start_Timestamp"]) - t0) / 1e6, (int(rs[-1]["End_Timestamp"]) - t0) / 1e6,
        gaps / 1e6))
    by = collections.Counter()
    for r in rs:
        by[short(r["Kernel_Name"])] += dur(r) / 1e6
    print("    " + ", ".join("%s %.2f" % (k, v) for k, v in by.most_common(10)))
