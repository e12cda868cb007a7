#!/usr/bin/env python3
"""attn_bwd_bxpp_kernel (ping-pong schedule) against attn_bwd_bx_kernel<., 8> (free-running sweep): the same products in
the same order — bit-identical while both started S' / dP from -LSE / -D; since the ping-pong kernel adds them in its
vector segment the two differ by fp32 rounding of that one addition (bound here: 5e-6 of the tensor's max) — then the
time of both at the production shape.
usage: tools/microbench/attn_pp_check.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "efficient-slowfast_amd")]
import torch  # noqa: E402
import sfhip  # noqa: E402

dev = torch.device("cuda:0")
L = sfhip.lib()


def run(B, thw, c, pp, iters=0):
    t, h, w = thw
    g = torch.Generator(device="cpu").manual_seed(c + B)
    qkv = sfhip.Act((torch.randn(B, t, h, w, 3 * c, generator=g) * 0.3).to(dev))
    x = sfhip.Act(torch.randn(B, t, h, w, c, generator=g).to(dev))
    dz = sfhip.Act(torch.randn(B, t, h, w, c, generator=g).to(dev))
    gamma = torch.tensor([0.7], device=dev)
    save = {}
    sfhip.attention(qkv.slice(0, c), qkv.slice(c, c), qkv.slice(2 * c, c), x, gamma, save=save)
    d = sfhip.Act(torch.zeros(B, t, h, w, 3 * c, device=dev))
    assert L.sf_attn_tune(0, 8) == 0 and L.sf_attn_tune(2, pp) == 0
    fn = lambda: sfhip.attention_bwd(qkv.slice(0, c), qkv.slice(c, c), qkv.slice(2 * c, c), dz, save["o"], save["lse"],
                                     gamma, d.slice(0, c), d.slice(c, c), d.slice(2 * c, c))
    fn()
    torch.cuda.synchronize()
    out = d.buf.clone()
    ms = None
    if iters:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fn()
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
    L.sf_attn_tune(0, 0)
    L.sf_attn_tune(2, 0)
    return out, ms


ok = True
for B, thw, c in [(2, (1, 8, 8), 32), (3, (3, 25, 33), 32), (2, (2, 17, 19), 20), (1, (4, 28, 28), 32), (2, (8, 28, 28), 24),
                  (8, (8, 56, 56), 32)]:
    big = thw == (8, 56, 56)
    a, ta = run(B, thw, c, 0, 5 if big else 0)
    b, tb = run(B, thw, c, 1, 5 if big else 0)
    same = torch.equal(a, b)
    rel = float((a - b).abs().max() / a.abs().max())
    ok &= rel < 5e-6
    n = thw[0] * thw[1] * thw[2]
    print("B=%d N=%d d=%d: ping-pong vs free-running: bit-identical %s, max |diff| / max |ref| %.2e (max |diff| %.3e, |ref| %.3e)%s" % (
        B, n, c, same, rel, float((a - b).abs().max()), float(a.abs().max()),
        "   free-running %.3f ms, ping-pong %.3f ms" % (ta, tb) if big else ""))
print("ALL WITHIN 5e-6" if ok else "MISMATCH")
