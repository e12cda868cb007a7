#!/usr/bin/env python3
"""A/B of the weight-gradient kernels per layer shape of cfg #3 (B = 8): conv_wgrad.hip (LDS-tiled,
sf_conv_tune(10, 0)) vs conv_wgrad_wave.hip with each blocks-per-wavefront shape (knob 11) and workgroup target
(knob 12).  Times include the partial-tile sum (torch) — the same for both — so compare columns, not absolutes.
usage: tools/microbench/wgrad_wave_bench.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "efficient-slowfast_amd")]
import torch  # noqa: E402
import sfhip  # noqa: E402
from sfhip import Act  # noqa: E402

dev = torch.device("cuda:0")
L = sfhip.lib()
B = 8
# (name, T, H, W, Cin, Cout, kernel, stride, pad)
SHAPES = [
    ("s5 3x1x1 2048->512", 8, 7, 7, 2048, 512, (3, 1, 1), (1, 1, 1), (1, 0, 0)),
    ("s5 1x3x3 512->512", 8, 7, 7, 512, 512, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    ("s5 1x1 512->2048", 8, 7, 7, 512, 2048, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ("s4 3x1x1 1024->256", 8, 14, 14, 1024, 256, (3, 1, 1), (1, 1, 1), (1, 0, 0)),
    ("s4 1x3x3 256->256", 8, 14, 14, 256, 256, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    ("s4 1x1 256->1024", 8, 14, 14, 256, 1024, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ("s4 3x1x1 1152->512", 8, 14, 14, 1152, 512, (3, 1, 1), (1, 1, 1), (1, 0, 0)),
    ("s3 1x3x3 128->128", 8, 28, 28, 128, 128, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    ("s3 1x1 128->512", 8, 28, 28, 128, 512, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ("s3 1x1 512->128", 8, 28, 28, 512, 128, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ("s3 3x1x1 576->256", 8, 28, 28, 576, 256, (3, 1, 1), (1, 1, 1), (1, 0, 0)),
    ("s3 1x3x3 256->256 s2", 8, 28, 28, 256, 256, (1, 3, 3), (1, 2, 2), (0, 1, 1)),
    ("s2 1x3x3 64->64", 8, 56, 56, 64, 64, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    ("s2 1x1 64->256", 8, 56, 56, 64, 256, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ("s2 1x1 256->64", 8, 56, 56, 256, 64, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ("s2 1x1 288->128", 8, 56, 56, 288, 128, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ("s2 1x1 288->512 s2", 8, 56, 56, 288, 512, (1, 1, 1), (1, 2, 2), (0, 0, 0)),
]


def odim(i, k, s, p):
    return (i + 2 * p - k) // s + 1


def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


print("%-22s %7s | %-13s | %s" % ("layer", "M", "old ms TF/s", "wave: shape/target ms TF/s (maxrel) S"))
tot_old = tot_new = 0.0
for name, T, H, W, cin, cout, k, s, p in SHAPES:
    g = torch.Generator(device="cpu").manual_seed(len(name))
    x = Act(torch.randn((B, T, H, W, cin), generator=g).to(dev))
    To, Ho, Wo = odim(T, k[0], s[0], p[0]), odim(H, k[1], s[1], p[1]), odim(W, k[2], s[2], p[2])
    dz = Act(torch.randn((B, To, Ho, Wo, cout), generator=g).to(dev))
    M = dz.rows
    flops = 2.0 * M * cin * cout * k[0] * k[1] * k[2]
    fn = lambda: sfhip.conv_wgrad(x, dz, cout, k, s, p)
    L.sf_conv_tune(10, 0)
    ref = fn()
    t_old = timeit(fn)
    L.sf_conv_tune(10, 1)
    cells = []
    best = 1e9
    for shape, target in ((-1, 0), (0, 512), (1, 512), (2, 512), (1, 256), (1, 1024), (2, 256), (2, 1024)):
        L.sf_conv_tune(11, shape)
        L.sf_conv_tune(12, target)
        y = fn()
        err = float((y - ref).abs().max() / ref.abs().max())
        t = timeit(fn)
        tag = "*" if shape < 0 else "%s/%d" % (("1x1", "2x1", "1x2", "2x2")[shape], target)
        cells.append("%s %.3f %.0f (%.0e)" % (tag, t, flops / t / 1e9, err))
        if shape < 0:
            t_plan = t
        best = min(best, t)
    L.sf_conv_tune(11, -1)
    L.sf_conv_tune(12, 0)
    tot_old += t_old
    tot_new += t_plan
    print("%-22s %7d | %6.3f %6.1f | %s" % (name, M, t_old, flops / t_old / 1e9, "  ".join(cells)))
print("sum: old %.3f ms, wave(planner) %.3f ms" % (tot_old, tot_new))
