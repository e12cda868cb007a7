#!/usr/bin/env python3
"""Where does a short-K 1x1x1 layer lose its time?  The same launch with the persistent kernel's ablation switches
(sf_conv_tune(5, mask): 1 = stores dropped, 2 = every A row reads rows 0..15 (L1-resident), 3 = both), on every tile
configuration, next to the one-pass kernel.  usage: tools/microbench/conv_pw_probe.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "efficient-slowfast_amd"))
import torch  # noqa: E402
import sfhip  # noqa: E402

L = sfhip.lib()
dev = torch.device("cuda")
CFG = {1: "13x2", 3: "7x4", 5: "7x2", 7: "13x1"}


def bench(x, wp, out, res, iters=30):
    for _ in range(3):
        sfhip.conv(x, wp, (1, 1, 1), res=res, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        sfhip.conv(x, wp, (1, 1, 1), res=res, out=out)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def run(name, rows, cin, cout, res=False):
    x = sfhip.Act(torch.randn(8, 1, rows // 8, 1, cin, device=dev))
    wp = sfhip.pack_conv_weight(torch.randn(cout, cin, 1, 1, 1, device=dev) * 0.05)
    out = sfhip.Act(torch.empty(8, 1, rows // 8, 1, cout, device=dev))
    r = sfhip.Act(torch.randn(8, 1, rows // 8, 1, cout, device=dev)) if res else None
    fl = 2.0 * rows * cin * cout
    by = 4.0 * rows * (cin + cout * (2 if res else 1))
    print("%s: rows %d, %d -> %d%s; MFMA floor %.1f us, HBM floor (6.3 TB/s) %.1f us" % (
        name, rows, cin, cout, " +res" if res else "", fl / 157.3e6, by / 6.3e6))
    for cfg in (3, 5, 1, 7):
        if cfg in (1, 3) and cout < 32:
            continue
        L.sf_conv_tune(1, cfg)
        L.sf_conv_tune(4, 0)
        t_old = bench(x, wp, out, r)
        L.sf_conv_tune(4, 1)
        ts = []
        for mask in (0, 1, 2, 3):
            L.sf_conv_tune(5, mask)
            ts.append(bench(x, wp, out, r))
        L.sf_conv_tune(5, 0)
        print("  %-5s one-pass %6.1f us | persistent %6.1f  no-stores %6.1f  A-from-L1 %6.1f  both %6.1f   (%.1f TF/s -> %.1f)" % (
            CFG[cfg], t_old, ts[0], ts[1], ts[2], ts[3], fl / t_old / 1e6, fl / ts[0] / 1e6))
    L.sf_conv_tune(1, -1)


run("res2 c", 200704, 64, 256)
run("res2 c dgrad-like", 200704, 256, 64, res=True)
run("res3 c", 50176, 128, 512)
run("res3 a", 50176, 512, 128)
run("res4 c", 12544, 256, 1024)
run("fast res2 c", 802816, 8, 32)
run("fast res3 c", 200704, 16, 64)
