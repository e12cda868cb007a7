#!/usr/bin/env python3
"""A/B of the dense-conv kernels per layer shape of cfg #3 (SlowFastDualAttention 8x8 R50, 224^2, B = 8):
conv_igemm.hip (LDS-tiled, sf_conv_tune(0, 0)) vs every tile configuration of conv_wave.hip (sf_conv_tune(1, c)) and
the planner's own choice.  Prints ms, algorithmic TFLOP/s (output positions x real Cin) and the max relative
difference of each variant against the LDS-tiled kernel's output.
usage: tools/microbench/conv_wave_bench.py [quick]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "efficient-slowfast_amd")]
import torch  # noqa: E402
import sfhip  # noqa: E402
from sfhip import Act  # noqa: E402

dev = torch.device("cuda:0")
L = sfhip.lib()
CFG_NAMES = ["13x2k4", "13x2k1", "7x4k4", "7x4k1", "7x2k4", "7x2k1", "13x1k4", "13x1k1"]
B = 8
# (name, T, H, W, Cin, Cout, kernel, stride, pad, transposed, residual+relu)
SHAPES = [
    ("s5 3x1x1 2048->512", 8, 7, 7, 2048, 512, (3, 1, 1), (1, 1, 1), (1, 0, 0), 0, 0),
    ("s5 1x3x3 512->512", 8, 7, 7, 512, 512, (1, 3, 3), (1, 1, 1), (0, 1, 1), 0, 0),
    ("s5 1x1 512->2048 +res", 8, 7, 7, 512, 2048, (1, 1, 1), (1, 1, 1), (0, 0, 0), 0, 1),
    ("s4 3x1x1 1024->256", 8, 14, 14, 1024, 256, (3, 1, 1), (1, 1, 1), (1, 0, 0), 0, 0),
    ("s4 1x3x3 256->256", 8, 14, 14, 256, 256, (1, 3, 3), (1, 1, 1), (0, 1, 1), 0, 0),
    ("s4 1x1 256->1024 +res", 8, 14, 14, 256, 1024, (1, 1, 1), (1, 1, 1), (0, 0, 0), 0, 1),
    ("s4 dgrad 1x1 1024->256", 8, 14, 14, 1024, 256, (1, 1, 1), (1, 1, 1), (0, 0, 0), 1, 0),
    ("s3 1x3x3 128->128", 8, 28, 28, 128, 128, (1, 3, 3), (1, 1, 1), (0, 1, 1), 0, 0),
    ("s3 dgrad 1x3x3 128->128", 8, 28, 28, 128, 128, (1, 3, 3), (1, 1, 1), (0, 1, 1), 1, 0),
    ("s3 1x1 128->512 +res", 8, 28, 28, 128, 512, (1, 1, 1), (1, 1, 1), (0, 0, 0), 0, 1),
    ("s3 1x1 512->128", 8, 28, 28, 512, 128, (1, 1, 1), (1, 1, 1), (0, 0, 0), 0, 0),
    ("s3 3x1x1 576->256 (s4a)", 8, 28, 28, 576, 256, (3, 1, 1), (1, 1, 1), (1, 0, 0), 0, 0),
    ("s3 1x3x3 256->256 s2", 8, 28, 28, 256, 256, (1, 3, 3), (1, 2, 2), (0, 1, 1), 0, 0),
    ("s2 1x3x3 64->64", 8, 56, 56, 64, 64, (1, 3, 3), (1, 1, 1), (0, 1, 1), 0, 0),
    ("s2 1x1 64->256 +res", 8, 56, 56, 64, 256, (1, 1, 1), (1, 1, 1), (0, 0, 0), 0, 1),
    ("s2 1x1 256->64", 8, 56, 56, 256, 64, (1, 1, 1), (1, 1, 1), (0, 0, 0), 0, 0),
    ("s2 1x1 288->128 (s3a)", 8, 56, 56, 288, 128, (1, 1, 1), (1, 1, 1), (0, 0, 0), 0, 0),
    ("f2 1x3x3 8->8", 32, 56, 56, 8, 8, (1, 3, 3), (1, 1, 1), (0, 1, 1), 0, 0),
    ("f2 1x1 8->32 +res", 32, 56, 56, 8, 32, (1, 1, 1), (1, 1, 1), (0, 0, 0), 0, 1),
    ("f2 3x1x1 32->8", 32, 56, 56, 32, 8, (3, 1, 1), (1, 1, 1), (1, 0, 0), 0, 0),
    ("f3 1x3x3 16->16", 32, 28, 28, 16, 16, (1, 3, 3), (1, 1, 1), (0, 1, 1), 0, 0),
    ("f3 3x1x1 64->16", 32, 28, 28, 64, 16, (3, 1, 1), (1, 1, 1), (1, 0, 0), 0, 0),
    ("f4 1x3x3 32->32", 32, 14, 14, 32, 32, (1, 3, 3), (1, 1, 1), (0, 1, 1), 0, 0),
    ("f4 1x1 32->128 +res", 32, 14, 14, 32, 128, (1, 1, 1), (1, 1, 1), (0, 0, 0), 0, 1),
    ("f5 1x3x3 64->64", 32, 7, 7, 64, 64, (1, 3, 3), (1, 1, 1), (0, 1, 1), 0, 0),
    ("f2s 7x1x1 32->64 s4", 32, 56, 56, 32, 64, (7, 1, 1), (4, 1, 1), (3, 0, 0), 0, 0),
]
if "quick" in sys.argv[1:]:
    SHAPES = SHAPES[::4]
L.sf_conv_tune(3, 1 if "k32" in sys.argv[1:] else 0)


def run(x, wp, k, s, p, tr, res, relu):
    if tr:
        xl = Act(torch.empty((x.N, x.T, x.H, x.W, wp.shape[0]), device=dev))
        return sfhip.conv_dgrad(x, wp, xl, k, s, p)
    return sfhip.conv(x, wp, k, s, p, res=res, relu=relu)


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


print("%-26s %9s | %-14s | %s" % ("layer", "M", "igemm ms TF/s", "wave: cfg ms TF/s (maxrel)  [* = planner]"))
tot_old = tot_new = tot_best = 0.0
for name, T, H, W, cin, cout, k, s, p, tr, rr in SHAPES:
    g = torch.Generator(device="cpu").manual_seed(hash(name) % 1000)
    x = Act(torch.randn((B, T, H, W, cin), generator=g).to(dev))
    w = torch.randn((cout, cin) + k, generator=g).to(dev) * (1.0 / (cin * k[0] * k[1] * k[2]) ** 0.5)
    wp = sfhip.pack_conv_weight(w)
    L.sf_conv_tune(0, 0)
    y0 = run(x, wp, k, s, p, tr, None, False)
    res = Act(torch.randn(y0.buf.shape, generator=g).to(dev)) if rr else None
    relu = bool(rr)
    y0 = run(x, wp, k, s, p, tr, res, relu)
    M = y0.rows
    flops = 2.0 * M * cin * cout * k[0] * k[1] * k[2]
    t_old = timeit(lambda: run(x, wp, k, s, p, tr, res, relu))
    L.sf_conv_tune(0, 1)
    cells = []
    best = 1e9
    for c in [-1] + list(range(len(CFG_NAMES))):
        if c >= 0 and cout <= 16 and CFG_NAMES[c][:2] != "13" and "x1" not in CFG_NAMES[c]:
            pass
        L.sf_conv_tune(1, c)
        y = run(x, wp, k, s, p, tr, res, relu)
        err = float((y.buf - y0.buf).abs().max() / y0.buf.abs().max().clamp_min(1e-20))
        t = timeit(lambda: run(x, wp, k, s, p, tr, res, relu))
        tag = "*" if c < 0 else CFG_NAMES[c]
        cells.append("%s %.3f %.0f (%.0e)" % (tag, t, flops / t / 1e9, err))
        if c < 0:
            t_plan = t
        else:
            best = min(best, t)
    L.sf_conv_tune(1, -1)
    tot_old += t_old
    tot_new += t_plan
    tot_best += best
    print("%-26s %9d | %6.3f %6.1f | %s" % (name, M, t_old, flops / t_old / 1e9, "  ".join(cells)))
print("sum: igemm %.3f ms, wave(planner) %.3f ms, wave(best cfg) %.3f ms" % (tot_old, tot_new, tot_best))
