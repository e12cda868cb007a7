#!/usr/bin/env python3
"""Forward / backward time of the small-head-dim attention at the two production shapes: d = 8, N = 25 088, B = 8 (cfg #3
s1_fuse) and d = 4, N = 100 352, B = 2 (cfg #5 s1_fuse).  Run with SF_ATTN_LANE=0 / 1 (and SF_ATTN_LANE_BWD) to A/B.
usage: tools/microbench/attn_small_bench.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "efficient-slowfast_amd")]
import torch  # noqa: E402
import sfhip  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, iters=5):
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


for B, thw, c in ((8, (8, 56, 56), 8), (2, (8, 112, 112), 4)):
    t, h, w = thw
    n = t * h * w
    g = torch.Generator(device="cpu").manual_seed(c)
    qkv = sfhip.Act((torch.randn(B, t, h, w, 3 * c, generator=g) * 0.6).to(dev))
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
