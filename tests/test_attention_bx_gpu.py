"""The d = 32, d = 8 (packed planes) and d <= 64 (two channel blocks) attention kernels on the bf16 matrix pipe (attn_bx.h: fp32 operands as three
bf16 pieces, six products per fp32 product) against an fp64 softmax attention and its autograd gradients on the same inputs — the bounds are
those the f32-MFMA kernels meet (tools/microbench/attn_precision.py prints both paths side by side: at unit-scale
scores O 1.2e-6 / 1.3e-6, gradients 3.7-5.0e-6 / 3.2-3.9e-6 of the tensor's max, split / f32 path)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture
def attn_knobs():
    """sf_attn_tune is process-wide: every test that forces a variant hands the planner back."""
    import sfhip
    L = sfhip.lib()
    yield L
    L.sf_attn_tune(0, 0)
    L.sf_attn_tune(1, 0)


# nw: wavefronts per workgroup of the backward sweep (sf_attn_tune knob 0; 0 = the launcher's own choice, which is 8
# only when B * ceil(N / 256) >= 256 — at 8 clips of N = 25 088, never at these sizes); parts: sweep parts (knob 1)
@pytest.mark.parametrize("nw,parts", [(0, 0), (8, 0), (8, 3), (4, 5)], ids=["auto", "nw8", "nw8_z3", "nw4_z5"])
@pytest.mark.parametrize("c", [32, 8, 64, 48])
@pytest.mark.parametrize("scale,thw", [(0.3, (3, 27, 31)), (1.0, (3, 27, 31)), (1.0, (2, 16, 16)), (1.0, (1, 5, 7))])
def test_split_product_attention_matches_fp64(scale, thw, c, nw, parts, attn_knobs):
    import sfhip
    if os.environ.get("SF_ATTN_BX", "1") == "0":
        pytest.skip("SF_ATTN_BX=0: the f32-MFMA kernels are selected")
    assert sfhip.lib().sf_attn_products_per_fp32(c) == 6  # d = 33..64: two 32-channel blocks
    if nw and c > 32:
        pytest.skip("the two-block kernels have one workgroup shape")
    if (nw or parts) and (scale != 1.0 or thw[0] * thw[1] * thw[2] < 512):
        pytest.skip("forced variants run at unit scale, on inputs with at least 8 tiles per sweep part")
    dev = torch.device("cuda:0")
    B = 2
    t, h, w = thw
    assert attn_knobs.sf_attn_tune(0, nw) == 0 and attn_knobs.sf_attn_tune(1, parts) == 0
    variant = attn_knobs.sf_attn_bwd_variant(B, t * h * w, c)
    assert variant == {32: 30, 8: 20, 64: 40, 48: 40}[c] + (nw or 4), variant
    n = t * h * w  # 2511: ragged last tile, partial last key block; 35: a single partial tile
    g = torch.Generator(device="cpu").manual_seed(11 + n)
    qkv = (torch.randn(B, t, h, w, 3 * c, generator=g) * scale).to(dev)
    x = torch.randn(B, t, h, w, c, generator=g).to(dev)
    dz = torch.randn(B, t, h, w, c, generator=g).to(dev)
    gamma = torch.tensor([0.7], device=dev)
    qa = sfhip.Act(qkv)
    save = {}
    out = sfhip.attention(qa.slice(0, c), qa.slice(c, c), qa.slice(2 * c, c), sfhip.Act(x), gamma, save=save)
    q, k, v = [qkv.view(B, n, 3 * c)[..., i * c:(i + 1) * c].double().requires_grad_(True) for i in range(3)]
    o = torch.softmax(q @ k.transpose(1, 2), dim=-1) @ v
    z = 0.7 * o + x.view(B, n, c).double()
    (z * dz.view(B, n, c).double()).sum().backward()

    def rel(got, ref):
        e = (got.double() - ref).abs()
        return (e.max() / ref.abs().max()).item(), (e.pow(2).mean().sqrt() / ref.abs().max()).item()

    mx, rms = rel(save["o"], o.detach())
    assert mx <= 5e-6 and rms <= 6e-7, (mx, rms)
    mx, _ = rel(out.buf.view(B, n, c), z.detach())
    assert mx <= 5e-6, mx
    d = sfhip.Act(torch.zeros(B, t, h, w, 3 * c, device=dev))
    sfhip.attention_bwd(qa.slice(0, c), qa.slice(c, c), qa.slice(2 * c, c), sfhip.Act(dz), save["o"], save["lse"],
                        gamma, d.slice(0, c), d.slice(c, c), d.slice(2 * c, c))
    got = d.buf.view(B, n, 3 * c)
    for i, ref in enumerate((q.grad, k.grad, v.grad)):
        mx, rms = rel(got[..., i * c:(i + 1) * c], ref)
        assert mx <= 2e-5 and rms <= 1e-6, ("dq dk dv".split()[i], mx, rms)
