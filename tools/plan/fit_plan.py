#!/usr/bin/env python3
"""Calibrate the conv_wave planner's time model against gpurun_out/conv_wave_bench.txt."""
import math, re, sys
sys.path.insert(0, __import__("os").path.dirname(__file__))
CFG = {"13x2k4": (13, 2, 4), "13x2k1": (13, 2, 1), "7x4k4": (7, 4, 4), "7x4k1": (7, 4, 1), "7x2k4": (7, 2, 4),
       "7x2k1": (7, 2, 1), "13x1k4": (13, 1, 4), "13x1k1": (13, 1, 1)}
OCC = {(13, 2): 2, (7, 4): 2, (7, 2): 4, (13, 1): 3}
B = 8


def parse(path):
    rows = []
    for line in open(path):
        m = re.match(r"^(.{26})\s+(\d+) \|\s+([\d.]+)\s+([\d.]+) \| (.*)$", line)
        if not m:
            continue
        name, M, t_old = m.group(1).strip(), int(m.group(2)), float(m.group(3))
        cells = {}
        for c in re.finditer(r"(\*|\w+) ([\d.]+) (\d+) \(", m.group(5)):
            cells[c.group(1)] = float(c.group(2))
        rows.append((name, M, t_old, cells))
    return rows


def shape_of(name):
    m = re.search(r"(\d)x(\d)x(\d) (\d+)->(\d+)", name) or re.search(r"()()1x1 (\d+)->(\d+)", name)
    if m.lastindex == 5:
        taps = int(m.group(1)) * int(m.group(2)) * int(m.group(3)); cin = int(m.group(4)); cout = int(m.group(5))
    else:
        taps = 1; cin = int(m.group(3)); cout = int(m.group(4))
    return taps, cin, cout


def model(M, taps, cin, cout, tm, tn, ks, P, has_res=False):
    cin_pad = (cin + 15) // 16 * 16
    nk = taps * cin_pad // 16
    nbn = math.ceil(cout / (tn * 16))
    nbm0 = math.ceil(M / (tm * 16))
    best = None
    for nbm in range(nbm0, nbm0 + 48):
        rows = math.ceil(M / nbm)
        tiles = nbm * nbn
        wgs = math.ceil(tiles / (4 // ks))
        steps = math.ceil(nk / ks)
        occ = OCC[(tm, tn)]
        per_cu = wgs / 256.0
        rounds = math.ceil(wgs / 256)          # workgroups the busiest CU runs
        mfma = steps * tm * tn * 128.0           # cycles of one wavefront's MFMAs
        load = steps * (tm + tn) * P["ld"] * 4   # cycles of the CU's load path for one workgroup
        # co-resident workgroups share the MFMA pipe and the load path; beyond the occupancy they queue
        body = rounds * max(mfma, load)
        fixed = math.ceil(rounds / occ) * (P["f0"] + P["f1"] * tm * tn)
        out_bytes = M * cout * 4.0 * (2 if has_res else 1)
        in_bytes = M * cin * 4.0 * (1 if taps == 1 else 1.3)
        hbm = (out_bytes + in_bytes) / P["bw"] * 2.1e3   # cycles at 2.1 GHz, bw in bytes/us... see below
        t = max(body + fixed, hbm) + out_bytes / P["bw"] * 2.1e3 * P["ep"]
        if best is None or t < best[0]:
            best = (t, rows, wgs)
    return best[0] / 2.1e3, best[1], best[2]   # us


if __name__ == "__main__":
    rows = parse(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/conv_wave_bench.txt")
    P = {"ld": 22.0, "f0": 6000.0, "f1": 150.0, "bw": 4.5e6, "ep": 0.5}
    for a in sys.argv[2:]:
        k, v = a.split("="); P[k] = float(v)
    loss = 0; tot_pick = tot_best = tot_old = 0
    for name, M, t_old, cells in rows:
        taps, cin, cout = shape_of(name)
        pred = {c: model(M, taps, cin, cout, *CFG[c], P, "+res" in name)[0] for c in CFG}
        pick = min(pred, key=pred.get)
        bestc = min(CFG, key=lambda c: cells[c])
        tot_pick += cells[pick]; tot_best += cells[bestc]; tot_old += t_old
        loss += sum((math.log(pred[c] / (cells[c] * 1e3))) ** 2 for c in CFG)
        print("%-26s pick %-7s %.3f  best %-7s %.3f old %.3f | " % (name, pick, cells[pick], bestc, cells[bestc], t_old) +
              " ".join("%s %.0f/%.0f" % (c, pred[c], cells[c] * 1e3) for c in CFG))
    print("loss %.2f  picked %.3f ms  best %.3f ms  igemm %.3f ms" % (loss, tot_pick, tot_best, tot_old))
