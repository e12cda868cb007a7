#!/usr/bin/env python3
"""A/B of conv_bx.hip (fp32 products as six bf16 MFMAs on operand piece planes, 256-row LDS tiles, direct-to-LDS loads)
against conv_wave.hip (v_mfma_f32_*_f32 per-wavefront kernel) on the long-reduction layer shapes of cfg #3
(SlowFastDualAttention 8x8 R50, 224^2, B = 8): forward and stride-1 data gradient, ms per call (all launches of the
call: operand splits + GEMM + split-K finish), algorithmic TFLOP/s, and both paths' max error against an fp64
convolution of the same inputs (relative to the output's max).
usage: tools/microbench/conv_bx_bench.py [noref]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "efficient-slowfast_amd")]
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
import sfhip  # noqa: E402
from sfhip import Act  # noqa: E402

dev = torch.device("cuda:0")
L = sfhip.lib()
B = int(os.environ.get("B", "8"))
# (name, T, H, W, Cin, Cout, kernel, stride, pad, transposed)
SHAPES = [
    ("s3 1x3x3 128->128", 8, 28, 28, 128, 128, (1, 3, 3), (1, 1, 1), (0, 1, 1), 0),
    ("s3 dgrad 1x3x3 128->128", 8, 28, 28, 128, 128, (1, 3, 3), (1, 1, 1), (0, 1, 1), 1),
    ("s3 3x1x1 576->256 (s4a)", 8, 28, 28, 576, 256, (3, 1, 1), (1, 1, 1), (1, 0, 0), 0),
    ("s3 1x3x3 256->256 s2", 8, 28, 28, 256, 256, (1, 3, 3), (1, 2, 2), (0, 1, 1), 0),
    ("s3 1x1 512->128", 8, 28, 28, 512, 128, (1, 1, 1), (1, 1, 1), (0, 0, 0), 0),
    ("s4 3x1x1 1024->256", 8, 14, 14, 1024, 256, (3, 1, 1), (1, 1, 1), (1, 0, 0), 0),
    ("s4 dgrad 3x1x1 256->1024", 8, 14, 14, 256, 1024, (3, 1, 1), (1, 1, 1), (1, 0, 0), 1),
    ("s4 1x3x3 256->256", 8, 14, 14, 256, 256, (1, 3, 3), (1, 1, 1), (0, 1, 1), 0),
    ("s4 dgrad 1x3x3 256->256", 8, 14, 14, 256, 256, (1, 3, 3), (1, 1, 1), (0, 1, 1), 1),
    ("s4 1x1 1024->256", 8, 14, 14, 1024, 256, (1, 1, 1), (1, 1, 1), (0, 0, 0), 0),
    ("s4 3x1x1 1152->512 (s5a)", 8, 14, 14, 1152, 512, (3, 1, 1), (1, 1, 1), (1, 0, 0), 0),
    ("s5 3x1x1 2048->512", 8, 7, 7, 2048, 512, (3, 1, 1), (1, 1, 1), (1, 0, 0), 0),
    ("s5 dgrad 3x1x1 512->2048", 8, 7, 7, 512, 2048, (3, 1, 1), (1, 1, 1), (1, 0, 0), 1),
    ("s5 1x3x3 512->512", 8, 7, 7, 512, 512, (1, 3, 3), (1, 1, 1), (0, 1, 1), 0),
    ("s5 1x1 512->2048", 8, 7, 7, 512, 2048, (1, 1, 1), (1, 1, 1), (0, 0, 0), 0),
    ("s5 dgrad 1x1 2048->512", 8, 7, 7, 2048, 512, (1, 1, 1), (1, 1, 1), (0, 0, 0), 1),
]


FLUSH = None if "hot" in sys.argv[1:] else torch.empty(640 * 1024 * 1024 // 4, device=dev)  # > the 256 MiB Infinity Cache


def timeit(fn, iters=10):
    """ms per call.  Default: COLD — a 640 MiB fill runs in front of every timed call, so operands come from HBM as
    they do inside a training step (re-running one call back to back leaves its operands in the 256 MiB Infinity
    Cache and flatters the bandwidth-sensitive kernels: the bf16-piece convs read 1.3x faster that way than in the
    model); `hot` on the command line times back-to-back calls."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    if FLUSH is None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters
    pairs = []
    for _ in range(iters):
        FLUSH.fill_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        pairs.append((e0, e1))
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in pairs)
    return t[len(t) // 2]


print("%-28s %7s %6s | %9s %6s %9s | %9s %6s %9s | %5s" % (
    "layer", "M", "K", "wave ms", "TF/s", "err", "bx ms", "TF/s", "err", "x"))
tot = [0.0, 0.0]
for name, T, H, W, cin, cout, k, s, p, tr in SHAPES:
    g = torch.Generator(device="cpu").manual_seed(len(name))
    if tr:  # data gradient: "input" is dL/dz over the forward OUTPUT dims (stride 1: the same), channels = forward Cout
        x = Act(torch.randn((B, T, H, W, cin), generator=g).to(dev))
        w = (torch.randn((cin, cout) + k, generator=g) / (cin * k[0] * k[1] * k[2]) ** 0.5).to(dev)  # forward [Cout_f = cin][Cin_f = cout]
        wtp = sfhip.pack_conv_weight(w.transpose(0, 1).contiguous())
        like = Act(torch.empty((B, T, H, W, cout), device=dev))

        def run():
            return sfhip.conv_dgrad(x, wtp, like, k, s, p)
    else:
        x = Act(torch.randn((B, T, H, W, cin), generator=g).to(dev))
        w = (torch.randn((cout, cin) + k, generator=g) / (cin * k[0] * k[1] * k[2]) ** 0.5).to(dev)
        wp = sfhip.pack_conv_weight(w)

        def run():
            return sfhip.conv(x, wp, k, s, p)
    ref = None
    if "noref" not in sys.argv[1:]:
        xd = x.buf.permute(0, 4, 1, 2, 3).double()
        if tr:
            ref = F.conv_transpose3d(xd, w.double(), None, s, p).permute(0, 2, 3, 4, 1)
        else:
            ref = F.conv3d(xd, w.double(), None, s, p).permute(0, 2, 3, 4, 1)
    res = []
    for on in (0, 2):  # 2: every shape the kernel covers, also where the planner's gate would leave it to conv_wave
        L.sf_conv_tune(7, on)
        y = run()
        err = float((y.buf.double() - ref).abs().max() / ref.abs().max()) if ref is not None else float("nan")
        ms = timeit(run)
        res.append((ms, err))
    L.sf_conv_tune(7, 1)
    xd_ = sfhip.ConvDesc
    gate = "*" if True else " "
    L.sf_conv_tune(7, 2)
    # breakdown of the bx call: operand splits / GEMM / finish (sf_conv_tune(8, mask) skips launches)
    parts = []
    for mask in (6, 3, 5):  # only splits, only GEMM, only finish
        L.sf_conv_tune(8, mask)
        parts.append(timeit(run))
    L.sf_conv_tune(8, 0)
    L.sf_conv_tune(7, 1)
    gated = timeit(run)  # default gate: did the planner take it?
    M = y.rows
    K = cin * k[0] * k[1] * k[2]
    fl = 2.0 * M * K * cout
    tot[0] += res[0][0]
    tot[1] += res[1][0]
    print("%-28s %7d %6d | %9.4f %6.1f %9.2e | %9.4f %6.1f %9.2e | %5.2f | split %.4f gemm %.4f (%5.1f TF/s) finish %.4f" % (
        name, M, K, res[0][0], fl / res[0][0] / 1e9, res[0][1], res[1][0], fl / res[1][0] / 1e9, res[1][1],
        res[0][0] / res[1][0], parts[0], parts[1], fl / parts[1] / 1e9, parts[2]) +
          ("  gate: %s" % ("bx" if abs(gated - res[1][0]) < abs(gated - res[0][0]) else "wave")))
print("total: wave %.3f ms, bx %.3f ms" % (tot[0], tot[1]))
