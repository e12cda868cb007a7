#!/usr/bin/env python3
"""HBM bytes per launch of the attention kernels from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs
of the same command).  hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (KiB counters; gfx950 FETCH_SIZE counts half of
wide coalesced reads — MI355X_MICROARCH.md).  Kernels are keyed by (name, grid size in threads): the two streams of a
step interleave differently from run to run, so dispatch order is not comparable between the passes.  The plane / part
reductions that follow a sweep kernel (backward: dK parts, dV parts, dQ planes -> 3 launches of attn_dq_reduce_kernel;
forward: 1 launch of attn_fwd_merge_kernel) run one thread per (row, 4 channels): grid = B*N*CP/4 rounded up to 256.
usage: tools/pmc_traffic_json.py <fetch_counter_collection.csv> <write_counter_collection.csv> > out.json"""
import csv
import json
import sys
from collections import OrderedDict, defaultdict


def load(path, counter):
    agg = defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        n = r["Kernel_Name"]
        if "attn" not in n:
            continue
        short = n.split("::")[1].split("(")[0] if "anonymous" in n else n[:60]
        agg[(short, int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    return agg


fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
assert set(fetch) == set(write), (sorted(set(fetch) ^ set(write)))
kern = OrderedDict()
mean = lambda v: sum(v) / len(v)  # noqa: E731
for key in fetch:
    name, grid = key
    f, w = mean(fetch[key]), mean(write[key])
    kern["%s grid=%d" % (name, grid)] = {"launches": len(fetch[key]), "FETCH_SIZE_KiB": round(f),
                                         "WRITE_SIZE_KiB": round(w), "hbm_bytes_per_launch": int((2 * f + w) * 1024)}
json.dump({"note": __doc__.split("usage:")[0].strip(), "kernels": kern}, sys.stdout, indent=1)
print()
