#!/usr/bin/env python3
"""Weight gradients of the small-channel layers (Cin or Cout <= 32: the Fast pathway and the lateral convs) of cfg #3 at
8 clips: us per call of the partial kernel alone and with the finish (sum of the split partials + un-packing into the
nn.Conv3d layout), against the time the operands take at 5 TB/s.  Cold operands (640 MiB fill in front of every call).
usage: tools/microbench/wgrad_small_bench.py [hot]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "efficient-slowfast_amd")]
import ctypes  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
import sfhip  # noqa: E402
from sfhip import Act  # noqa: E402

dev = torch.device("cuda:0")
L_ = sfhip.lib()
B = int(os.environ.get("B", "8"))
# (name, T, H, W, Cin, Cout, kernel, stride, pad, launches per step)
SHAPES = [
    ("f2 1x3x3 8->8", 32, 56, 56, 8, 8, (1, 3, 3), (1, 1, 1), (0, 1, 1), 3),
    ("f2 3x1x1 32->8", 32, 56, 56, 32, 8, (3, 1, 1), (1, 1, 1), (1, 0, 0), 2),
    ("f2 1x1 8->32", 32, 56, 56, 8, 32, (1, 1, 1), (1, 1, 1), (0, 0, 0), 3),
    ("f2 3x1x1 16->8", 32, 56, 56, 16, 8, (3, 1, 1), (1, 1, 1), (1, 0, 0), 1),
    ("f3 1x3x3 16->16", 32, 28, 28, 16, 16, (1, 3, 3), (1, 1, 1), (0, 1, 1), 3),
    ("f3 1x3x3 16->16 s2", 32, 56, 56, 16, 16, (1, 3, 3), (1, 2, 2), (0, 1, 1), 1),
    ("f3 1x1 16->64", 32, 28, 28, 16, 64, (1, 1, 1), (1, 1, 1), (0, 0, 0), 4),
    ("f3 3x1x1 64->16", 32, 28, 28, 64, 16, (3, 1, 1), (1, 1, 1), (1, 0, 0), 3),
    ("f4 1x3x3 32->32", 32, 14, 14, 32, 32, (1, 3, 3), (1, 1, 1), (0, 1, 1), 5),
    ("f4 1x1 32->128", 32, 14, 14, 32, 128, (1, 1, 1), (1, 1, 1), (0, 0, 0), 6),
    ("f4 3x1x1 128->32", 32, 14, 14, 128, 32, (3, 1, 1), (1, 1, 1), (1, 0, 0), 5),
    ("lat 5x1x1 8->16 s4", 32, 56, 56, 8, 16, (5, 1, 1), (4, 1, 1), (2, 0, 0), 1),
    ("lat 5x1x1 32->64 s4", 32, 56, 56, 32, 64, (5, 1, 1), (4, 1, 1), (2, 0, 0), 1),
]
FLUSH = torch.empty((64 if "hot" in sys.argv[1:] else 640) * 1024 * 1024 // 4, device=dev)


def timeit(fn, iters=10):
    """GPU time of fn's launches: three 640 MiB read passes run in front of every timed call — they evict the caches
    with CLEAN lines (a fill would leave 256 MiB of dirty lines whose write-back competes with the timed kernel) and
    keep the GPU busy long enough for the host to enqueue the whole call behind them, so the events do not measure
    the Python wrapper (tens of us: longer than these kernels)."""
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


print("%-22s %8s | %9s %9s %8s | %7s %5s | %9s" % ("layer", "M", "kernel us", "+finish", "at 5TB/s", "splits", "n", "err"))
tot = [0.0, 0.0, 0.0]
for name, T, H, W, cin, cout, k, s, p, n in SHAPES:
    g = torch.Generator(device="cpu").manual_seed(len(name))
    x = Act(torch.randn((B, T, H, W, cin), generator=g).to(dev))
    od = [(i + 2 * pp - kk) // ss + 1 for i, pp, kk, ss in zip((T, H, W), p, k, s)]
    dz = Act(torch.randn((B, od[0], od[1], od[2], cout), generator=g).to(dev))
    dst = torch.zeros((cout, cin) + k, device=dev)

    def kern():  # the partial kernel alone, through the C ABI (no allocation, no sum)
        L_.sf_conv_wgrad(ctypes.byref(d), x.ptr(), dz.ptr(), dz.cs, dz.coff, part.data_ptr(), stream)

    def full():
        return sfhip.conv_wgrad(x, dz, cout, k, s, p, finish_into=(dst, cin, 0))

    dst.zero_()
    full()
    ref = torch.nn.grad.conv3d_weight(x.buf.permute(0, 4, 1, 2, 3).double(), (cout, cin) + k,
                                      dz.buf.permute(0, 4, 1, 2, 3).double(), s, p)
    err = float((dst.double() - ref).abs().max() / ref.abs().max())
    kT, kH, kW = k
    d = sfhip.ConvDesc(x.N, x.T, x.H, x.W, cin, x.cs, x.coff, dz.T, dz.H, dz.W, cout, 0, 0, 1, kT, kH, kW, s[0], s[1], s[2],
                       p[0], p[1], p[2], 1, 1, 1, (cin + 15) // 16 * 16, 0, 0, 0, 0)
    S = L_.sf_conv_wgrad_splits(ctypes.byref(d))
    part = torch.empty((S, cout, kT * kH * kW, (cin + 15) // 16 * 16), device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    tk, tf = timeit(kern), timeit(full)
    floor = (x.buf.numel() + dz.buf.numel()) * 4 / 5e12 * 1e6
    tot[0] += n * tk
    tot[1] += n * tf
    tot[2] += n * floor
    print("%-22s %8d | %9.1f %9.1f %8.1f | %7d %5d | %9.2e" % (name, dz.rows, tk, tf, floor, S, n, err))
print("per step (n launches each): kernel %.0f us, with finish %.0f us, operands at 5 TB/s %.0f us" % tuple(tot))
