#!/usr/bin/env python3
"""Do the bf16-pipe attention kernels and the f32-MFMA conv kernels share a CU?  The d = 32 attention backward
(v_mfma_f32_32x32x16_bf16: the matrix pipe) and a train of conv weight-gradient / forward launches
(v_mfma_f32_*_f32: the vector pipe) are timed alone on one stream each, then together on two streams.  If the second
number is near max(a, b) the two pipes overlap; near a + b they time-slice like two f32-MFMA streams do.
SF_ATTN_BX_NW=4|8 picks the attention workgroup width (8: one 512-thread workgroup per CU, registers full).
usage: tools/microbench/coexec_attn_conv.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "efficient-slowfast_amd")]
import torch  # noqa: E402
import sfhip  # noqa: E402

dev = torch.device("cuda:0")
B, (t, h, w), c = 8, (8, 56, 56), 32
g = torch.Generator(device="cpu").manual_seed(3)
qkv = sfhip.Act((torch.randn(B, t, h, w, 3 * c, generator=g) * 0.3).to(dev))
x = sfhip.Act(torch.randn(B, t, h, w, c, generator=g).to(dev))
dz = sfhip.Act(torch.randn(B, t, h, w, c, generator=g).to(dev))
gamma = torch.tensor([0.7], device=dev)
save = {}
sfhip.attention(qkv.slice(0, c), qkv.slice(c, c), qkv.slice(2 * c, c), x, gamma, save=save)
d = sfhip.Act(torch.zeros(B, t, h, w, 3 * c, device=dev))


def attn():
    sfhip.attention_bwd(qkv.slice(0, c), qkv.slice(c, c), qkv.slice(2 * c, c), dz, save["o"], save["lse"], gamma,
                        d.slice(0, c), d.slice(c, c), d.slice(2 * c, c))


# res4 3x1x1 1024 -> 256 at 8x14x14 (weight gradient and forward: the two heaviest conv classes of the step)
cx = sfhip.Act(torch.randn(8, 8, 14, 14, 1024, generator=g).to(dev))
cdz = sfhip.Act(torch.randn(8, 8, 14, 14, 256, generator=g).to(dev))
wp = sfhip.pack_conv_weight((torch.randn(256, 1024, 3, 1, 1, generator=g) * 0.02).to(dev))
NCONV = int(os.environ.get("NCONV", "24"))


def convs():
    for _ in range(NCONV):
        sfhip.conv_wgrad(cx, cdz, 256, (3, 1, 1), padding=(1, 0, 0))
        sfhip.conv(cx, wp, (3, 1, 1), padding=(1, 0, 0))


def timed(fa, fb, iters=4):
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    for _ in range(2):
        if fa:
            with torch.cuda.stream(sa):
                fa()
        if fb:
            with torch.cuda.stream(sb):
                fb()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    sa.wait_event(e0)
    sb.wait_event(e0)
    for _ in range(iters):
        if fa:
            with torch.cuda.stream(sa):
                fa()
        if fb:
            with torch.cuda.stream(sb):
                fb()
    torch.cuda.current_stream().wait_stream(sa)
    torch.cuda.current_stream().wait_stream(sb)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


ta, tb, tab = timed(attn, None), timed(None, convs), timed(attn, convs)
print("attention backward alone %.2f ms | %d x (wgrad + fwd) alone %.2f ms | together %.2f ms  (sum %.2f, max %.2f)" % (
    ta, NCONV, tb, tab, ta + tb, max(ta, tb)))
