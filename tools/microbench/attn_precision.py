#!/usr/bin/env python3
"""Error of the attention kernels against an fp64 softmax attention on the same inputs, for the product path the
environment selects (SF_ATTN_BX=0: v_mfma_f32_32x32x2_f32; 1: three-way bf16 split, six products).  Reports the
max / rms error of O = softmax(q k^T) v relative to max|O|, and of the log-sum-exp, at a few score scales (the larger
the scores, the more one rounding of s moves exp(s)).  usage: [SF_ATTN_BX=1] tools/microbench/attn_precision.py [d ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "efficient-slowfast_amd")]
import torch  # noqa: E402
import sfhip  # noqa: E402

dev = torch.device("cuda:0")
dims = [int(a) for a in sys.argv[1:]] or [32]
print("SF_ATTN_BX=%s" % os.environ.get("SF_ATTN_BX", "0"))
for c in dims:
    for scale in (0.3, 1.0, 2.5):
        B, (t, h, w) = 2, (3, 27, 31)  # N = 2511: ragged last tile
        n = t * h * w
        g = torch.Generator(device="cpu").manual_seed(7 * c + int(10 * scale))
        qkv = (torch.randn(B, t, h, w, 3 * c, generator=g) * scale).to(dev)
        qa = sfhip.Act(qkv)
        x = sfhip.Act(torch.zeros(B, t, h, w, c, device=dev))
        save = {}
        sfhip.attention(qa.slice(0, c), qa.slice(c, c), qa.slice(2 * c, c), x, torch.ones(1, device=dev), save=save)
        q, k, v = [qkv.view(B, n, 3 * c)[..., i * c:(i + 1) * c].double() for i in range(3)]
        s = q @ k.transpose(1, 2)
        ref = torch.softmax(s, dim=-1) @ v
        lse = torch.logsumexp(s, dim=-1) * 1.4426950408889634
        # backward: dL/dz = dz, L = sum(dz * (gamma * softmax(q k^T) v + x)), gamma = 0.7
        dz = torch.randn(B, t, h, w, c, generator=g).to(dev)
        gamma = torch.tensor([0.7], device=dev)
        qd, kd, vd = [u.clone().requires_grad_(True) for u in (q, k, v)]
        (0.7 * (torch.softmax(qd @ kd.transpose(1, 2), dim=-1) @ vd) * dz.view(B, n, c).double()).sum().backward()
        d = sfhip.Act(torch.zeros(B, t, h, w, 3 * c, device=dev))
        sfhip.attention_bwd(qa.slice(0, c), qa.slice(c, c), qa.slice(2 * c, c), sfhip.Act(dz), save["o"], save["lse"],
                            gamma, d.slice(0, c), d.slice(c, c), d.slice(2 * c, c))
        got = d.buf.view(B, n, 3 * c).double()
        bw = []
        for i, rg in enumerate((qd.grad, kd.grad, vd.grad)):
            e = (got[..., i * c:(i + 1) * c] - rg).abs()
            bw.append("%.2e/%.2e" % ((e.max() / rg.abs().max()).item(), (e.pow(2).mean().sqrt() / rg.abs().max()).item()))
        err = (save["o"].double() - ref).abs()
        print("d=%-3d scale %.1f  backward max/rms rel. max|.|:  dq %s  dk %s  dv %s" % (c, scale, *bw))
        print("d=%-3d scale %.1f  max|s| %6.1f   O: max %.2e rms %.2e (rel. max|O|)   lse: max abs %.2e" % (
            c, scale, s.abs().max().item(), (err.max() / ref.abs().max()).item(),
            (err.pow(2).mean().sqrt() / ref.abs().max()).item(), (save["lse"].double() - lse).abs().max().item()))
