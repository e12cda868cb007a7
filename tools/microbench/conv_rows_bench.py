#!/usr/bin/env python3
"""Small-channel stride-1 layers of cfg #3 at 8 clips (the Fast pathway, lateral / q|k|v projections), forward and data
gradient: conv_rows.hip (sf_conv_tune(22, 2)) against the kernels it replaces (sf_conv_tune(22, 0): conv_wave /
conv_wave_p / conv_small), us per call with cold operands (three 640 MiB read passes in front of every timed call),
beside the layer's operands at 5.5 TB/s and its FLOPs at 157 TFLOP/s.  usage: tools/microbench/conv_rows_bench.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "efficient-slowfast_amd")]
import torch  # noqa: E402
import sfhip  # noqa: E402
from sfhip import Act  # noqa: E402

dev = torch.device("cuda:0")
L = sfhip.lib()
B = int(os.environ.get("B", "8"))
# (name, T, H, W, Cin, Cout, kernel, forward launches per step, data-gradient launches per step)
SHAPES = [
    ("s2a 32->8 t3", 32, 56, 56, 32, 8, (3, 1, 1), 2, 2),
    ("s2a 16->8 t3", 32, 56, 56, 16, 8, (3, 1, 1), 1, 1),
    ("s2b 8->8 s3", 32, 56, 56, 8, 8, (1, 3, 3), 3, 3),
    ("s2c 8->32 p", 32, 56, 56, 8, 32, (1, 1, 1), 3, 3),
    ("s2 16->32 p", 32, 56, 56, 16, 32, (1, 1, 1), 1, 1),
    ("s3a 64->16 t3 @56", 32, 56, 56, 64, 16, (3, 1, 1), 1, 1),
    ("s3a 64->16 t3", 32, 28, 28, 64, 16, (3, 1, 1), 3, 3),
    ("s3b 16->16 s3", 32, 28, 28, 16, 16, (1, 3, 3), 3, 3),
    ("s3c 16->64 p", 32, 28, 28, 16, 64, (1, 1, 1), 4, 4),
    ("s4a 128->32 t3 @28", 32, 28, 28, 128, 32, (3, 1, 1), 1, 1),
    ("s4a 128->32 t3", 32, 14, 14, 128, 32, (3, 1, 1), 5, 5),
    ("s4b 32->32 s3", 32, 14, 14, 32, 32, (1, 3, 3), 5, 5),
    ("s4c 32->128 p", 32, 14, 14, 32, 128, (1, 1, 1), 6, 6),
    ("slow 256->32 p @56", 8, 56, 56, 256, 32, (1, 1, 1), 0, 0),
    ("slow 32->256 p @56", 8, 56, 56, 32, 256, (1, 1, 1), 0, 1),
]
FLUSH = torch.empty(640 * 1024 * 1024 // 4, device=dev)


def timeit(fn, iters=8):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    pairs = []
    for _ in range(iters):
        for _ in range(3):
            FLUSH.sum()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        pairs.append((e0, e1))
        torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in pairs)
    return t[len(t) // 2] * 1e3


print("%-22s %8s | %8s %8s %6s | %7s %7s | %s" % ("layer", "M", "old us", "rows us", "x", "hbm us", "mfma us", "max|d|/max"))
tot = [0.0, 0.0, 0.0]
for name, T, H, W, cin, cout, k, nf, nd in SHAPES:
    g = torch.Generator(device="cpu").manual_seed(len(name))
    p = tuple(kk // 2 for kk in k)
    x = Act(torch.randn((B, T, H, W, cin), generator=g).to(dev))
    w = (torch.randn((cout, cin) + k, generator=g) / (cin * k[0] * k[1] * k[2]) ** 0.5).to(dev)
    wp, wtp = sfhip.pack_conv_weight_pair(w)
    dz = Act(torch.randn((B, T, H, W, cout), generator=g).to(dev))
    for kind, n in (("fwd", nf), ("dgrad", nd)):
        if kind == "fwd":
            run = lambda: sfhip.conv(x, wp, k, padding=p, stats=True)[0]  # noqa: E731
        else:
            run = lambda: sfhip.conv_dgrad(dz, wtp, x, k, padding=p)  # noqa: E731
        res = []
        outs = []
        for mode in (0, 2):
            L.sf_conv_tune(22, mode)
            outs.append(run().buf.clone())
            res.append(timeit(run))
        L.sf_conv_tune(22, 1)
        M = x.rows
        taps = k[0] * k[1] * k[2]
        hbm = 4.0 * M * (cin + cout) / 5.5e6
        mf = 2.0 * M * taps * cin * cout / 157.3e6
        err = float((outs[0] - outs[1]).abs().max() / outs[0].abs().max())
        print("%-22s %8d | %8.1f %8.1f %6.2f | %7.1f %7.1f | %.1e" % (name + " " + kind, M, res[0], res[1], res[0] / res[1],
                                                                     hbm, mf, err))
        tot[0] += n * res[0]
        tot[1] += n * res[1]
        tot[2] += n * max(hbm, mf)
print("per step (launch counts of cfg #3): old %.0f us, conv_rows %.0f us, floor %.0f us" % tuple(tot))
