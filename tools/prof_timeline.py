#!/usr/bin/env python3
"""Concurrency picture of the last training step in a rocprofv3 kernel-trace CSV: span, GPU-busy time, how much of
the span has 1 / 2 / 3+ kernels in flight, and which kernels own the solo (critical-path) time.
usage: tools/prof_timeline.py <kernel_trace.csv> [marker-kernel-substring]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
marker = sys.argv[2] if len(sys.argv) > 2 else "multi_tensor_apply"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# steps are delimited by the optimizer's multi-tensor kernels: take the region between the last two bursts
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
bursts = []
for i in idx:
    if not bursts or i - bursts[-1][-1] > 50:
        bursts.append([i])
    else:
        bursts[-1].append(i)
a, b = bursts[-2][-1] + 1, bursts[-1][0]
ev = rows[a:b]
t0 = int(ev[0]["Start_Timestamp"])
t1 = max(int(r["End_Timestamp"]) for r in ev)
points = []
for r in ev:
    n = r["Kernel_Name"]
    short = n.split("::")[1].split("(")[0] if "anonymous" in n else n[:40]
    points.append((int(r["Start_Timestamp"]), 1, short))
    points.append((int(r["End_Timestamp"]), -1, short))
points.sort(key=lambda p: (p[0], p[1]))
active = defaultdict(int)
depth_time = defaultdict(float)
solo = defaultdict(float)
prev = t0
for ts, d, name in points:
    k = sum(active.values())
    dt = ts - prev
    if dt > 0:
        depth_time[min(k, 3)] += dt
        if k == 1:
            solo[next(n for n, c in active.items() if c > 0)] += dt
    active[name] += d
    prev = ts
span = (t1 - t0) / 1e6
print("step span %.2f ms, kernels %d" % (span, len(ev)))
for k in sorted(depth_time):
    print("  %s kernels in flight: %7.2f ms (%4.1f%%)" % ("3+" if k == 3 else k, depth_time[k] / 1e6, 100 * depth_time[k] / (t1 - t0)))
print("solo time by kernel:")
for n, v in sorted(solo.items(), key=lambda kv: -kv[1])[:18]:
    print("  %-44s %7.2f ms" % (n, v / 1e6))

# ---- idle gaps (no kernel in flight): the longest ones with the kernels either side, and totals by the kernel that
#      FOLLOWS the gap (what the GPU was waiting for: a host-side launch, or a cross-stream event)
gaps = []
end = None
last = None
for r in sorted(ev, key=lambda r: int(r["Start_Timestamp"])):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = r["Kernel_Name"]
    short = n.split("::")[1].split("(")[0] if "anonymous" in n else n[:40]
    if end is not None and s > end:
        gaps.append((s - end, last, short, (end - t0) / 1e6))
    if end is None or e > end:
        end, last = e, short
tot = sum(g[0] for g in gaps)
print("idle gaps: %d, total %.2f ms; > 20 us: %d (%.2f ms)" % (
    len(gaps), tot / 1e6, sum(1 for g in gaps if g[0] > 20000), sum(g[0] for g in gaps if g[0] > 20000) / 1e6))
by_next = defaultdict(float)
for g in gaps:
    by_next[g[2]] += g[0]
print("idle time by the kernel that follows the gap:")
for n, v in sorted(by_next.items(), key=lambda kv: -kv[1])[:14]:
    print("  %-44s %7.2f ms" % (n, v / 1e6))
print("longest gaps (us, at ms into the step, before -> after):")
for g in sorted(gaps, key=lambda g: -g[0])[:16]:
    print("  %7.1f  @%6.2f  %s -> %s" % (g[0] / 1e3, g[3], g[1], g[2]))
# idle time per 5-ms slice of the step
sl = defaultdict(float)
for g in gaps:
    sl[int(g[3] // 5)] += g[0]
print("idle per 5 ms slice:", " ".join("%d:%.1f" % (k * 5, v / 1e6) for k, v in sorted(sl.items())))

# ---- time with NO MFMA-bound kernel in flight (convolutions, attention sweeps): what runs then is what the two-stream
#      schedule fails to hide behind the matrix pipes
MFMA_KEYS = ("conv_wave", "conv_wgrad", "conv_stem", "conv_igemm", "conv_kernel", "conv_bx", "conv_pw_bx", "conv_small",
             "attn_fwd", "attn_bwd", "attn_small", "attn_lane")


def is_mfma(name):
    return any(k in name for k in MFMA_KEYS) and "finish" not in name and "reduce" not in name and "merge" not in name


active = defaultdict(int)
prev = t0
no_mfma = defaultdict(float)
no_mfma_total = 0.0
for ts, d, name in points:
    dt = ts - prev
    if dt > 0:
        live = [n for n, c in active.items() if c > 0]
        if not any(is_mfma(n) for n in live):
            no_mfma_total += dt
            if live:
                for n in live:
                    no_mfma[n] += dt / len(live)
            else:
                no_mfma["(idle)"] += dt
    active[name] += d
    prev = ts
print("no MFMA-bound kernel in flight: %.2f ms of %.2f; by what runs then:" % (no_mfma_total / 1e6, span))
for n, v in sorted(no_mfma.items(), key=lambda kv: -kv[1])[:22]:
    print("  %-44s %7.2f ms" % (n, v / 1e6))
