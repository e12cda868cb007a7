#!/usr/bin/env python3
"""Forward / backward time of the S->F spatial attention at the production shapes of cfg #3 (B = 8): d = 32 N = 25 088
(s2_fuse), d = 64 N = 6 272 (s3_fuse), d = 8 N = 25 088 (s1_fuse).  ATTN_SHAPES=32,64,8 selects; ATTN_ITERS launches
each.  Also the driver for rocprofv3 --pmc passes (tools/attn_pmc.sh).
usage: tools/microbench/attn_bench.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "efficient-slowfast_amd")]
import torch  # noqa: E402
import sfhip  # noqa: E402

dev = torch.device("cuda:0")
SHAPES = {32: (8, (8, 56, 56)), 64: (8, (8, 28, 28)), 8: (8, (8, 56, 56)), 128: (8, (8, 14, 14))}
ITERS = int(os.environ.get("ATTN_ITERS", "5"))


def timeit(fn, iters=ITERS):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for c in [int(v) for v in os.environ.get("ATTN_SHAPES", "32,64,8").split(",")]:
    B, (t, h, w) = SHAPES[c]
    n = t * h * w
    g = torch.Generator(device="cpu").manual_seed(c)
    qkv = sfhip.Act((torch.randn(B, t, h, w, 3 * c, generator=g) * (0.6 if c <= 8 else 0.3)).to(dev))
    x = sfhip.Act(torch.randn(B, t, h, w, c, generator=g).to(dev))
    dz = sfhip.Act(torch.randn(B, t, h, w, c, generator=g).to(dev))
    gamma = torch.tensor([0.7], device=dev)
    save = {}
    fwd = lambda: sfhip.attention(qkv.slice(0, c), qkv.slice(c, c), qkv.slice(2 * c, c), x, gamma, save=save)
    tf = timeit(fwd)
    d = sfhip.Act(torch.zeros(B, t, h, w, 3 * c, device=dev))
    bwd = lambda: sfhip.attention_bwd(qkv.slice(0, c), qkv.slice(c, c), qkv.slice(2 * c, c), dz, save["o"], save["lse"],
                                      gamma, d.slice(0, c), d.slice(c, c), d.slice(2 * c, c))
    tb = timeit(bwd)
    fl = 2.0 * B * n * n * c
    print("d=%d N=%d B=%d: forward %.3f ms (%.1f TFLOP/s)   backward %.3f ms (%.1f TFLOP/s)" % (
        c, n, B, tf, 2 * fl / tf / 1e9, tb, 5 * fl / tb / 1e9))
