#!/usr/bin/env python3
"""Micro-benchmark of sf_conv_fwd shapes: TFLOP/s per call (HIP events on the launch stream)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "efficient-slowfast_amd"))
import torch, sfhip

def run(name, n, t, h, w, cin, cout, k=(1, 1, 1), s=(1, 1, 1), p=(0, 0, 0), res=False, iters=20):
    dev = torch.device("cuda")
    x = sfhip.Act(torch.randn(n, t, h, w, cin, device=dev))
    wt = torch.randn(cout, cin, *k, device=dev) * 0.05
    wp = sfhip.pack_conv_weight(wt)
    sc, bi = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev)
    out = sfhip.conv(x, wp, k, s, p, scale=sc, bias=bi, relu=True)
    r = sfhip.Act(torch.randn_like(out.buf)) if res else None
    for _ in range(3):
        sfhip.conv(x, wp, k, s, p, scale=sc, bias=bi, relu=True, res=r, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        sfhip.conv(x, wp, k, s, p, scale=sc, bias=bi, relu=True, res=r, out=out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    flops = 2.0 * out.rows * cout * cin * k[0] * k[1] * k[2]
    gb = (x.buf.numel() + out.buf.numel() * (2 if res else 1)) * 4 / 1e9
    print("%-34s M=%-7d K=%-5d N=%-5d %8.1f us  %6.1f TF/s  %5.2f TB/s" % (name, out.rows, cin * k[0] * k[1] * k[2], cout, ms * 1e3, flops / ms / 1e9, gb / ms))

if __name__ == "__main__":
    run("bigK 1x1 1024->256 M=100352", 8, 8, 56, 28, 1024, 256)
    run("bigK 1x1 4096->256 M=100352", 8, 8, 56, 28, 4096, 256)
    run("bigK 1x1 1024->512 M=196608 (6/CU)", 8, 8, 64, 48, 1024, 512)
    run("s3.a 288->128 @56", 8, 8, 56, 56, 288, 128)
    run("s3.b 3x3 s2 128->128", 8, 8, 56, 56, 128, 128, (1, 3, 3), (1, 2, 2), (0, 1, 1))
    run("s3.c 128->512 +res", 8, 8, 28, 28, 128, 512, res=True)
    run("s4.a t3 1024->256", 8, 8, 14, 14, 1024, 256, (3, 1, 1), (1, 1, 1), (1, 0, 0))
    run("s4.b 3x3 256->256", 8, 8, 14, 14, 256, 256, (1, 3, 3), (1, 1, 1), (0, 1, 1))
    run("s4.c 256->1024 +res", 8, 8, 14, 14, 256, 1024, res=True)
    run("s5.a t3 2048->512", 8, 8, 7, 7, 2048, 512, (3, 1, 1), (1, 1, 1), (1, 0, 0))
    run("s5.b 3x3 512->512", 8, 8, 7, 7, 512, 512, (1, 3, 3), (1, 1, 1), (0, 1, 1))
    run("s5.c 512->2048 +res", 8, 8, 7, 7, 512, 2048, res=True)
    run("s2.c 64->256 +res", 8, 8, 56, 56, 64, 256, res=True)
    run("fast s2.a t3 16->8", 8, 32, 56, 56, 16, 8, (3, 1, 1), (1, 1, 1), (1, 0, 0))
    run("fast s2.c 8->32 +res", 8, 32, 56, 56, 8, 32, res=True)
