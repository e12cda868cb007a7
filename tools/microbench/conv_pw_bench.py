#!/usr/bin/env python3
"""Pointwise (1x1x1) layers of cfg #3 at 8 clips, forward and data gradient: conv_pw_bx_kernel (bf16 pieces, activations
split in registers; sf_conv_tune(21, 2)) against the f32 kernels (sf_conv_tune(21, 0)), us per call with cold operands
(three 640 MiB read passes in front of every timed call, which also let the host enqueue the call ahead of the GPU), both
against an fp64 GEMM.  usage: tools/microbench/conv_pw_bench.py [noref]"""
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
# (name, T, H, W, Cin, Cout, launches per step forward, data gradient)
SHAPES = [
    ("s2 64->256", 8, 56, 56, 64, 256, 3, 2),
    ("s2 256->64", 8, 56, 56, 256, 64, 2, 3),
    ("s3 128->512", 8, 28, 28, 128, 512, 4, 3),
    ("s3 512->128", 8, 28, 28, 512, 128, 3, 4),
    ("s4 256->1024", 8, 14, 14, 256, 1024, 6, 0),
    ("s4 1024->256", 8, 14, 14, 1024, 256, 0, 6),
    ("s5 512->2048", 8, 7, 7, 512, 2048, 3, 0),
    ("s5 2048->512", 8, 7, 7, 2048, 512, 0, 3),
    ("f2s 288->128 @56", 8, 56, 56, 288, 128, 1, 0),
    ("s3a 512->64 @28", 8, 28, 28, 512, 64, 1, 0),
]
FLUSH = torch.empty((32 if "hot" in sys.argv[1:] else 640) * 1024 * 1024 // 4, device=dev)


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


print("%-20s %7s | %8s %6s %9s | %8s %6s %9s | %5s | %s" % ("layer", "M", "f32 us", "TF/s", "err", "pw us", "TF/s", "err",
                                                            "x", "model's pick"))
tot = [0.0, 0.0]
for name, T, H, W, cin, cout, nf, nd in SHAPES:
    g = torch.Generator(device="cpu").manual_seed(len(name))
    x = Act(torch.randn((B, T, H, W, cin), generator=g).to(dev))
    w = (torch.randn((cout, cin, 1, 1, 1), generator=g) / cin ** 0.5).to(dev)
    wp, wtp = sfhip.pack_conv_weight_pair(w)
    dz = Act(torch.randn((B, T, H, W, cout), generator=g).to(dev))
    like = Act(torch.empty((B, T, H, W, cin), device=dev))
    for kind, n in (("fwd", nf), ("dgrad", nd)):
        if n == 0:
            continue
        if kind == "fwd":
            run = lambda: sfhip.conv(x, wp, (1, 1, 1))  # noqa: E731
            ref = None if "noref" in sys.argv[1:] else (x.buf.double().view(-1, cin) @ w.double().view(cout, cin).t())
        else:
            run = lambda: sfhip.conv_dgrad(dz, wtp, like, (1, 1, 1))  # noqa: E731
            ref = None if "noref" in sys.argv[1:] else (dz.buf.double().view(-1, cout) @ w.double().view(cout, cin))
        res = []
        for mode in (0, 2, 1):
            L.sf_conv_tune(21, mode)
            y = run()
            err = float((y.buf.double().view(ref.shape) - ref).abs().max() / ref.abs().max()) if ref is not None else float("nan")
            res.append((timeit(run), err))
        L.sf_conv_tune(21, 1)
        M = x.rows
        fl = 2.0 * M * cin * cout
        pick = "pw" if abs(res[2][0] - res[1][0]) < abs(res[2][0] - res[0][0]) else "f32"
        tot[0] += n * res[0][0]
        tot[1] += n * res[1][0]
        print("%-20s %7d | %8.1f %6.1f %9.2e | %8.1f %6.1f %9.2e | %5.2f | %s" % (
            name + " " + kind, M, res[0][0], fl / res[0][0] / 1e6, res[0][1], res[1][0], fl / res[1][0] / 1e6, res[1][1],
            res[0][0] / res[1][0], pick))
print("per step (launch counts of cfg #3): f32 %.0f us, pw %.0f us" % tuple(tot))
