#!/usr/bin/env python3
"""conv_small.hip against the matrix-core kernels on the Fast pathway's shapes, with the kernel's ablation switches
(sf_conv_tune(6, 1 | mask << 4): 1 = no staging loads, 2 = no FMA loop, 4 = no stores)."""
import os
import sys

os.environ["SF_CONV_SMALL"] = "2"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "efficient-slowfast_amd"))
import torch  # noqa: E402
import sfhip  # noqa: E402

L = sfhip.lib()
dev = torch.device("cuda")


def bench(fn, iters=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def run(name, shp, cin, cout, k, stats=False):
    n, t, h, w = shp
    x = sfhip.Act(torch.randn(n, t, h, w, cin, device=dev))
    wp = sfhip.pack_conv_weight(torch.randn(cout, cin, *k, device=dev) * 0.05)
    p = tuple(kk // 2 for kk in k)
    out = sfhip.Act(torch.empty(n, t, h, w, cout, device=dev))
    rows = n * t * h * w
    by = 4.0 * rows * (cin + cout)

    def f():
        return sfhip.conv(x, wp, k, (1, 1, 1), p, out=out, stats=stats)

    L.sf_conv_tune(6, 0)
    L.sf_conv_tune(7, 0)
    t0 = bench(f)
    L.sf_conv_tune(7, 1)
    th = bench(f)
    L.sf_conv_tune(7, 0)
    res = []
    for mask in (0, 1, 2, 4, 7):
        L.sf_conv_tune(6, 1 | (mask << 4))
        res.append(bench(f))
    L.sf_conv_tune(6, 1)
    L.sf_conv_tune(7, 1)
    print("%-22s rows %7d  HBM floor %5.1f us | wave %6.1f | HALO %6.1f | small %6.1f  no-loads %6.1f  no-fma %6.1f  no-stores %6.1f  none %6.1f" % (
        name + (" +stats" if stats else ""), rows, by / 6.3e6, t0, th, *res))


run("8->8 1x3x3", (8, 32, 56, 56), 8, 8, (1, 3, 3))
run("8->8 1x3x3", (8, 32, 56, 56), 8, 8, (1, 3, 3), stats=True)
run("32->8 3x1x1", (8, 32, 56, 56), 32, 8, (3, 1, 1))
run("8->32 1x1x1", (8, 32, 56, 56), 8, 32, (1, 1, 1))
run("8->32 1x1x1", (8, 32, 56, 56), 8, 32, (1, 1, 1), stats=True)
run("16->16 1x3x3", (8, 32, 28, 28), 16, 16, (1, 3, 3))
run("64->16 3x1x1", (8, 32, 28, 28), 64, 16, (3, 1, 1))
run("16->64 1x1x1", (8, 32, 28, 28), 16, 64, (1, 1, 1))
run("32->32 1x3x3", (8, 32, 14, 14), 32, 32, (1, 3, 3))
run("128->32 3x1x1", (8, 32, 14, 14), 128, 32, (3, 1, 1))
run("32->128 1x1x1", (8, 32, 14, 14), 32, 128, (1, 1, 1))
run("16->8 3x1x1", (8, 32, 56, 56), 16, 8, (3, 1, 1))
run("32->32 1x3x3", (8, 32, 14, 14), 32, 32, (1, 3, 3), stats=True)
