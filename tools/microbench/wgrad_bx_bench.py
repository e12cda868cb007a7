#!/usr/bin/env python3
"""A/B of the bf16-piece weight-gradient kernel (conv_bx.hip, sf_conv_wgrad_bx) against conv_wgrad_wave.hip on the
long-reduction layer shapes of cfg #3 (B = 8): ms per call (operand splits + GEMM; the partial sum is done by torch in
both arms here), algorithmic TFLOP/s, and both paths' max error against an fp64 weight gradient.
usage: tools/microbench/wgrad_bx_bench.py [noref]"""
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
SHAPES = [
    ("s2 1x3x3 64->64", 8, 56, 56, 64, 64, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    ("s3 1x3x3 128->128", 8, 28, 28, 128, 128, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    ("s3 3x1x1 576->256 (s4a)", 8, 28, 28, 576, 256, (3, 1, 1), (1, 1, 1), (1, 0, 0)),
    ("s3 1x3x3 256->256 s2", 8, 28, 28, 256, 256, (1, 3, 3), (1, 2, 2), (0, 1, 1)),
    ("s3 1x1 512->128", 8, 28, 28, 512, 128, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ("s3 1x1 128->512", 8, 28, 28, 128, 512, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ("s4 3x1x1 1024->256", 8, 14, 14, 1024, 256, (3, 1, 1), (1, 1, 1), (1, 0, 0)),
    ("s4 1x3x3 256->256", 8, 14, 14, 256, 256, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    ("s4 1x1 256->1024", 8, 14, 14, 256, 1024, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ("s4 3x1x1 1152->512 (s5a)", 8, 14, 14, 1152, 512, (3, 1, 1), (1, 1, 1), (1, 0, 0)),
    ("s4 1x3x3 512->512 s2", 8, 14, 14, 512, 512, (1, 3, 3), (1, 2, 2), (0, 1, 1)),
    ("s5 3x1x1 2048->512", 8, 7, 7, 2048, 512, (3, 1, 1), (1, 1, 1), (1, 0, 0)),
    ("s5 1x3x3 512->512", 8, 7, 7, 512, 512, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    ("s5 1x1 512->2048", 8, 7, 7, 512, 2048, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ("s4 1x1 1152->2048 s2", 8, 14, 14, 1152, 2048, (1, 1, 1), (1, 2, 2), (0, 0, 0)),
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


print("%-26s %7s %6s | %9s %6s %9s | %9s %6s %9s | %5s" % (
    "layer", "M", "K", "wave ms", "TF/s", "err", "bx ms", "TF/s", "err", "x"))
tot = [0.0, 0.0]
for name, T, H, W, cin, cout, k, s, p in SHAPES:
    g = torch.Generator(device="cpu").manual_seed(len(name))
    x = Act(torch.randn((B, T, H, W, cin), generator=g).to(dev))
    od = [(i + 2 * pp - kk) // ss + 1 for i, pp, kk, ss in zip((T, H, W), p, k, s)]
    dz = Act(torch.randn((B, od[0], od[1], od[2], cout), generator=g).to(dev))

    def run():
        return sfhip.conv_wgrad(x, dz, cout, k, s, p)
    ref = None
    if "noref" not in sys.argv[1:]:
        wd = torch.zeros((cout, cin) + k, dtype=torch.float64, device=dev, requires_grad=True)
        F.conv3d(x.buf.permute(0, 4, 1, 2, 3).double(), wd, None, s, p).backward(dz.buf.permute(0, 4, 1, 2, 3).double())
        ref = wd.grad
    res = []
    for on in (0, 2):
        L.sf_conv_tune(9, on)
        dwp = run()
        err = float("nan")
        if ref is not None:
            dw = sfhip.unpack_conv_weight_grad(dwp, ref.shape)
            err = float((dw.double() - ref).abs().max() / ref.abs().max())
        res.append((timeit(run), err))
    L.sf_conv_tune(9, 1)
    M = dz.rows
    K = cin * k[0] * k[1] * k[2]
    fl = 2.0 * M * K * cout
    tot[0] += res[0][0]
    tot[1] += res[1][0]
    print("%-26s %7d %6d | %9.4f %6.1f %9.2e | %9.4f %6.1f %9.2e | %5.2f" % (
        name, M, K, res[0][0], fl / res[0][0] / 1e9, res[0][1], res[1][0], fl / res[1][0] / 1e9, res[1][1],
        res[0][0] / res[1][0]))
print("total: wave %.3f ms, bx %.3f ms" % (tot[0], tot[1]))
