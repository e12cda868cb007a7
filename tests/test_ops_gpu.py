"""GPU parity tests of every libsfhip kernel (through the C ABI via the ctypes binding) against the
oracle's primitives / plain torch fp32 CPU ops on the same seeded inputs.

Tolerance: 1e-3 max-norm relative is the north_star budget for the whole network; single ops are held to
2e-4 (fp32 MFMA is a k-ordered fmaf chain, error ~1e-7 * sum|a*b|; softmax adds __expf's ~2 ulp).
"""
import os
import zlib

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = 2e-4
REPORT = []


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _rel(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def _report(name, err):
    REPORT.append("%-60s %.3e" % (name, err))
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "ops_report.txt"), "a") as f:
            f.write(REPORT[-1] + "\n")
    except OSError:
        pass


def _ndhwc(x):  # NCTHW cpu tensor -> Act on gpu
    import sfhip
    return sfhip.Act(x.permute(0, 2, 3, 4, 1).contiguous().to(_dev()))


def _back(a):  # Act -> NCTHW cpu
    return a.buf[..., a.coff:a.coff + a.C].permute(0, 4, 1, 2, 3).contiguous().cpu()


def test_layout_roundtrip():
    import sfhip
    dev = _dev()
    x = torch.randn(2, 3, 4, 10, 12)
    a = sfhip.from_ncthw(x.to(dev), cpad=4, ph=3, pw=3, wp=20)
    assert tuple(a.buf.shape) == (2, 4, 16, 20, 4)
    ref = torch.zeros(2, 4, 16, 20, 4)
    ref[:, :, 3:13, 3:15, :3] = x.permute(0, 2, 3, 4, 1)
    assert torch.equal(a.buf.cpu(), ref)
    b = sfhip.from_ncthw(x.to(dev))
    assert torch.equal(sfhip.to_ncthw(b).cpu(), x)


CONV_CASES = [
    # name, Cin, Cout, kernel, stride, pad, dil, (N,T,H,W), relu, res, bn
    ("pw_64_256_res", 64, 256, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1), (2, 2, 14, 14), True, True, True),
    ("pw_72_64", 72, 64, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1), (1, 2, 13, 11), True, False, True),
    ("pw_stride2_288_512", 288, 512, (1, 1, 1), (1, 2, 2), (0, 0, 0), (1, 1, 1), (1, 2, 14, 14), False, False, True),
    ("t3_1152_512", 1152, 512, (3, 1, 1), (1, 1, 1), (1, 0, 0), (1, 1, 1), (1, 4, 7, 7), True, False, True),
    ("t3_16_8", 16, 8, (3, 1, 1), (1, 1, 1), (1, 0, 0), (1, 1, 1), (2, 8, 9, 9), True, False, True),
    ("s3_64_64", 64, 64, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 1, 1), (1, 2, 20, 20), True, False, True),
    ("s3_128_128_s2", 128, 128, (1, 3, 3), (1, 2, 2), (0, 1, 1), (1, 1, 1), (2, 2, 14, 14), True, False, True),
    ("s3_dil2_32_32", 32, 32, (1, 3, 3), (1, 1, 1), (0, 2, 2), (1, 2, 2), (1, 2, 12, 12), False, False, False),
    ("f2s_k7_32_64", 32, 64, (7, 1, 1), (4, 1, 1), (3, 0, 0), (1, 1, 1), (1, 16, 6, 6), True, False, True),
    ("pw_8_24_bias", 8, 24, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1), (2, 4, 9, 9), False, False, False),
    ("pw_32_96", 32, 96, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1), (1, 4, 10, 10), False, False, False),
    ("odd_27_16", 27, 16, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1), (2, 4, 5, 5), True, False, True),
    ("odd_3_24_k333", 3, 24, (3, 3, 3), (1, 2, 2), (1, 1, 1), (1, 1, 1), (2, 4, 16, 16), True, False, True),
    ("fc_2304_400", 2304, 400, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1), (8, 1, 1, 1), False, False, False),
    # small channel counts at >= 4096 positions: conv_small.hip (LDS-staged input, scalar-register weights)
    ("sm_8_8_s3", 8, 8, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 1, 1), (2, 4, 20, 28), True, True, True),
    ("sm_16_16_s3_ragged", 16, 16, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 1, 1), (2, 3, 28, 28), True, False, True),
    ("sm_32_32_s3_frame", 32, 32, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 1, 1), (4, 6, 14, 14), False, False, True),
    ("sm_32_8_t3", 32, 8, (3, 1, 1), (1, 1, 1), (1, 0, 0), (1, 1, 1), (2, 6, 20, 20), True, False, True),
    ("sm_128_32_t3", 128, 32, (3, 1, 1), (1, 1, 1), (1, 0, 0), (1, 1, 1), (2, 4, 24, 24), True, True, True),
    ("sm_8_32_pw", 8, 32, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1), (2, 4, 24, 25), False, True, True),
    ("sm_16_64_pw", 16, 64, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1), (2, 4, 24, 24), True, False, True),
    ("sm_32_128_pw", 32, 128, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1), (2, 4, 23, 24), False, False, False),
    ("sm_64_16_t3", 64, 16, (3, 1, 1), (1, 1, 1), (1, 0, 0), (1, 1, 1), (1, 5, 30, 30), True, False, True),
]


@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv(case):
    import sfhip
    name, cin, cout, k, s, p, d, shp, relu, use_res, use_bn = case
    dev = _dev()
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()) % 10000)
    n, t, h, w = shp
    x = torch.randn(n, cin, t, h, w, generator=g)
    wt = torch.randn(cout, cin, *k, generator=g) / np.sqrt(cin * k[0] * k[1] * k[2])
    scale = torch.rand(cout, generator=g) + 0.5 if use_bn else None
    bias = torch.randn(cout, generator=g) * 0.1
    y = F.conv3d(x, wt, None, s, p, d)
    y = y * scale.view(1, -1, 1, 1, 1) + bias.view(1, -1, 1, 1, 1) if use_bn else y + bias.view(1, -1, 1, 1, 1)
    res = torch.randn(y.shape, generator=g) if use_res else None
    if use_res:
        y = y + res
    if relu:
        y = F.relu(y)
    # input lives in a slice of a wider buffer, output goes into a slice of a wider buffer
    xa_full = sfhip.Act(torch.randn(n, t, h, w, cin + 8, generator=g).to(dev))
    xa = xa_full.slice(4 if cin % 4 == 0 else 3, cin)
    xa.buf[..., xa.coff:xa.coff + cin] = x.permute(0, 2, 3, 4, 1).to(dev)
    wp = sfhip.pack_conv_weight(wt.to(dev))
    sc = scale.to(dev) if use_bn else None
    out = sfhip.conv(xa, wp, k, s, p, d, scale=sc, bias=bias.to(dev), relu=relu,
                     res=_ndhwc(res) if use_res else None, out_reserve=(4, 8))
    torch.cuda.synchronize()
    err = _rel(_back(out), y)
    _report("conv/" + name, err)
    assert err < TOL, (name, err)


@pytest.mark.parametrize("cin,cout,k,shp,res", [(512, 256, (3, 1, 1), (2, 4, 7, 7), True),
                                                (256, 128, (1, 3, 3), (2, 4, 14, 14), False),
                                                (1024, 256, (1, 1, 1), (2, 4, 14, 14), True)])
def test_conv_split_k_schedule(cin, cout, k, shp, res):
    """Short-M / long-K layers on the LDS-tiled kernels (conv_igemm.hip, the fallback of conv_wave.hip: selected here
    with sf_conv_tune(0, 0)) take the split-K schedule (partials in a workspace + finish kernel with the full
    epilogue): same result as torch, and within fp32 re-association of the single-pass schedule and of the
    per-wavefront kernel's in-workgroup K split."""
    import ctypes
    import sfhip
    dev = _dev()
    sfhip.lib().sf_conv_tune(0, 0)
    try:
        _split_k_case(cin, cout, k, shp, res, dev)
    finally:
        sfhip.lib().sf_conv_tune(0, 1)


def _split_k_case(cin, cout, k, shp, res, dev):
    import ctypes
    import sfhip
    g = torch.Generator().manual_seed(cin + cout)
    n, t, h, w = shp
    x = torch.randn(n, cin, t, h, w, generator=g)
    wt = torch.randn(cout, cin, *k, generator=g) / np.sqrt(cin * k[0] * k[1] * k[2])
    scale = torch.rand(cout, generator=g) + 0.5
    bias = torch.randn(cout, generator=g) * 0.1
    pad = tuple(kk // 2 for kk in k)
    ref = F.conv3d(x, wt, None, 1, pad) * scale.view(-1, 1, 1, 1) + bias.view(-1, 1, 1, 1)
    r = torch.randn(ref.shape, generator=g) if res else None
    if res:
        ref = ref + r
    ref = F.relu(ref)
    xa = _ndhwc(x)
    wp = sfhip.pack_conv_weight(wt.to(dev))
    kw = dict(scale=scale.to(dev), bias=bias.to(dev), relu=True, res=_ndhwc(r) if res else None)
    y_split = sfhip.conv(xa, wp, k, (1, 1, 1), pad, **kw)
    d = sfhip.ConvDesc(n, t, h, w, cin, cin, 0, t, h, w, cout, cout, 0, 1, k[0], k[1], k[2], 1, 1, 1, pad[0], pad[1], pad[2],
                       1, 1, 1, wp.shape[2], 1, 0, 0, 0)
    assert sfhip.lib().sf_conv_fwd_ws_floats(ctypes.byref(d)) > 0, "this shape is meant to exercise split-K"
    saved = sfhip.SPLIT_K
    sfhip.SPLIT_K = False
    try:
        y_single = sfhip.conv(xa, wp, k, (1, 1, 1), pad, **kw)
    finally:
        sfhip.SPLIT_K = saved
    torch.cuda.synchronize()
    e = _rel(_back(y_split), ref)
    _report("conv split-K %d->%d k%s M=%d" % (cin, cout, k, n * t * h * w), e)
    assert e < TOL
    assert _rel(_back(y_split), _back(y_single)) < 1e-5
    sfhip.lib().sf_conv_tune(0, 1)  # the default path: per-wavefront kernel, K split across the workgroup's wavefronts
    assert sfhip.lib().sf_conv_fwd_ws_floats(ctypes.byref(d)) == 0
    y_wave = sfhip.conv(xa, wp, k, (1, 1, 1), pad, **kw)
    torch.cuda.synchronize()
    sfhip.lib().sf_conv_tune(0, 0)
    assert _rel(_back(y_wave), ref) < TOL
    assert _rel(_back(y_wave), _back(y_single)) < 1e-5


def test_conv_stem_trick():
    """7x7 stem on a border-padded NDHWC4 input expressed as kW=1 / 'Cin'=28 contiguous floats."""
    import sfhip
    dev = _dev()
    g = torch.Generator().manual_seed(5)
    for kt, cout in ((1, 64), (5, 8)):
        x = torch.randn(2, 3, 6, 32, 32, generator=g)
        wt = torch.randn(cout, 3, kt, 7, 7, generator=g) / np.sqrt(147 * kt)
        y = F.conv3d(x, wt, None, (1, 2, 2), (kt // 2, 3, 3))
        xa = sfhip.from_ncthw(x.to(dev), cpad=4, ph=3, pw=3, wp=38)  # Wp even
        w4 = torch.zeros(cout, 4, kt, 7, 7)
        w4[:, :3] = wt
        # packed [Cout][kt*kh][kw*4 + c] -> pad 28 -> 32
        wp = torch.zeros(cout, kt * 7, 32)
        wp[:, :, :28] = w4.permute(0, 2, 3, 4, 1).reshape(cout, kt * 7, 28)
        view = sfhip.Act(xa.buf.view(2, 6, 38, 19, 8))
        out = sfhip.conv(view, wp.to(dev).contiguous(), (kt, 7, 1), (1, 2, 1), (kt // 2, 0, 0), cin=28,
                         out_thw=(6, 16, 16))
        torch.cuda.synchronize()
        err = _rel(_back(out), y)
        _report("conv/stem_trick_kt%d" % kt, err)
        assert err < TOL


@pytest.mark.parametrize("shape", [(3, 9, 20, 40), (1, 12, 18, 70)])
def test_conv_stem_ring_forward(shape):
    """The Fast pathway's 5x7x7 stem through conv_stem_fwd_kernel (LDS ring): zero frames outside the clip in T,
    ragged last 16-position block, the t range cut into parts, scale / bias / ReLU epilogue, channel-slice store."""
    import sfhip
    dev = _dev()
    g = torch.Generator().manual_seed(11 + shape[0])
    kt, cout = 5, 8
    n, t, h, w = shape
    x = torch.randn(n, 3, t, h, w, generator=g)
    wt = torch.randn(cout, 3, kt, 7, 7, generator=g) / np.sqrt(147 * kt)
    scale = torch.rand(cout, generator=g) + 0.5
    bias = torch.randn(cout, generator=g) * 0.2
    y = F.relu(F.conv3d(x, wt, None, (1, 2, 2), (kt // 2, 3, 3)) * scale.view(-1, 1, 1, 1) + bias.view(-1, 1, 1, 1))
    wpx = (w + 6 + 1) // 2 * 2
    xa = sfhip.from_ncthw(x.to(dev), cpad=4, ph=3, pw=3, wp=wpx)
    w4 = torch.zeros(cout, 4, kt, 7, 7)
    w4[:, :3] = wt
    wp = torch.zeros(cout, kt * 7, 32)
    wp[:, :, :28] = w4.permute(0, 2, 3, 4, 1).reshape(cout, kt * 7, 28)
    view = sfhip.Act(xa.buf.view(n, t, h + 6, wpx // 2, 8))
    ho, wo = y.shape[3], y.shape[4]
    wide = sfhip.Act(torch.full((n, t, ho, wo, cout + 4), 7.0, device=dev)).slice(4, cout)  # store into a channel slice
    out = sfhip.conv(view, wp.to(dev).contiguous(), (kt, 7, 1), (1, 2, 1), (kt // 2, 0, 0), cin=28,
                     out_thw=(t, ho, wo), scale=scale.to(dev), bias=bias.to(dev), relu=True, out=wide)
    torch.cuda.synchronize()
    err = _rel(_back(out), y)
    _report("conv/stem_ring %s" % (shape,), err)
    assert err < TOL
    assert float((wide.buf[..., :4] - 7.0).abs().max()) == 0.0, "neighbouring channels untouched"


def test_conv_out_cmul():
    import sfhip
    dev = _dev()
    g = torch.Generator().manual_seed(9)
    x = torch.randn(1, 16, 2, 6, 6, generator=g)
    wt = torch.randn(16, 16, 1, 1, 1, generator=g) / 4
    y = F.conv3d(x, wt)
    out_full = sfhip.Act(torch.zeros(1, 2, 6, 6, 32, device=dev))
    sfhip.conv(_ndhwc(x), sfhip.pack_conv_weight(wt.to(dev)), (1, 1, 1), out=sfhip.Act(out_full.buf, 1, 31),
               out_cmul=2)
    torch.cuda.synchronize()
    got = out_full.buf[..., 1::2].permute(0, 4, 1, 2, 3).cpu()
    assert _rel(got, y) < TOL
    assert float(out_full.buf[..., 0::2].abs().max()) == 0.0


@pytest.mark.parametrize("c", [8, 27, 64])
def test_pool(c):
    import sfhip
    g = torch.Generator().manual_seed(c)
    x = torch.randn(2, c, 4, 13, 12, generator=g)
    for k, s, p in (((1, 3, 3), (1, 2, 2), (0, 1, 1)), ((3, 3, 3), (1, 2, 2), (1, 1, 1)), ((4, 1, 1), (4, 1, 1), (0, 0, 0))):
        out = sfhip.pool(_ndhwc(x), k, s, p)
        torch.cuda.synchronize()
        assert torch.equal(_back(out), F.max_pool3d(x, k, s, p))
    out = sfhip.pool(_ndhwc(x), (4, 3, 3), (1, 1, 1), avg=True)
    torch.cuda.synchronize()
    err = _rel(_back(out), F.avg_pool3d(x, (4, 3, 3), 1))
    _report("pool/avg_c%d" % c, err)
    assert err < 1e-5


@pytest.mark.parametrize("c,alpha", [(8, 4), (32, 4), (3, 8), (128, 4), (300, 1)])
def test_eca_gate(c, alpha):
    import sfhip
    from oracle import slowfast_oracle as oracle
    dev = _dev()
    g = torch.Generator().manual_seed(c)
    x = torch.randn(2, c, 8, 7, 9, generator=g)
    w3 = torch.randn(1, 1, 3, generator=g)
    scale, bias = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.1
    ref = F.max_pool3d(x, (alpha, 1, 1), (alpha, 1, 1))
    pooled_ref = ref.mean((2, 3, 4))
    ref = oracle.eca({"m.conv.weight": w3}, "m", ref)
    ref = F.relu(ref * scale.view(1, -1, 1, 1, 1) + bias.view(1, -1, 1, 1, 1))
    xa = _ndhwc(x)
    pooled = sfhip.tmax_mean(xa, alpha)
    out = sfhip.gate_apply(xa, alpha, pooled, w3=w3.to(dev), scale=scale.to(dev), bias=bias.to(dev), relu=True)
    torch.cuda.synchronize()
    e1, e2 = _rel(pooled, pooled_ref), _rel(_back(out), ref)
    _report("eca/c%d_a%d pooled" % (c, alpha), e1)
    _report("eca/c%d_a%d out" % (c, alpha), e2)
    assert e1 < 1e-5 and e2 < TOL


ATTN_CASES = [(8, (2, 12, 12), 4), (32, (2, 14, 14), 4), (64, (4, 7, 7), 4), (128, (2, 7, 7), 2), (3, (4, 8, 8), 8),
              (28, (2, 9, 9), 4), (16, (1, 5, 5), 1), (32, (1, 2, 2), 1), (32, (8, 28, 28), 1)]


@pytest.mark.parametrize("c,thw,alpha", ATTN_CASES, ids=["c%d_n%d" % (c, t * h * w) for c, (t, h, w), a in ATTN_CASES])
def test_attention(c, thw, alpha):
    """Flash kernel == dense softmax attention of the oracle (wdf_attention_helper.py:41-54), incl. the
    fused gamma-residual, BN affine, ReLU and nearest T-upsample."""
    import sfhip
    from oracle import slowfast_oracle as oracle
    dev = _dev()
    g = torch.Generator().manual_seed(c * 1000 + thw[0])
    t, h, w = thw
    B = 2
    x = torch.randn(B, c, t, h, w, generator=g)
    sd = {}
    for nm in ("query_conv", "key_conv", "value_conv"):
        std = (0.7 if nm != "value_conv" else 1.0) / np.sqrt(c)
        sd["m.%s.weight" % nm] = torch.randn(c, c, 1, 1, 1, generator=g) * std
        sd["m.%s.bias" % nm] = torch.randn(c, generator=g) * 0.1
    sd["m.gamma"] = torch.tensor([0.7])
    scale, bias = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.1
    ref = oracle.spatial_attention(sd, "m", x)
    ref = F.relu(ref * scale.view(1, -1, 1, 1, 1) + bias.view(1, -1, 1, 1, 1)).repeat_interleave(alpha, dim=2)
    # product path: one pointwise GEMM for q|k|v, then the flash kernel writing into a wider buffer
    xa = _ndhwc(x)
    wqkv = torch.cat([sd["m.query_conv.weight"], sd["m.key_conv.weight"], sd["m.value_conv.weight"]], 0)
    bqkv = torch.cat([sd["m.query_conv.bias"], sd["m.key_conv.bias"], sd["m.value_conv.bias"]], 0)
    qkv = sfhip.conv(xa, sfhip.pack_conv_weight(wqkv.to(dev)), (1, 1, 1), bias=bqkv.to(dev))
    out = sfhip.new_act(dev, B, t * alpha, h, w, c, 0, 8)
    sfhip.attention(qkv.slice(0, c), qkv.slice(c, c), qkv.slice(2 * c, c), xa, sd["m.gamma"].to(dev),
                    scale.to(dev), bias.to(dev), relu=True, alpha=alpha, out=out)
    torch.cuda.synchronize()
    err = _rel(_back(out), ref)
    _report("attn/c%d_n%d_a%d" % (c, t * h * w, alpha), err)
    assert err < TOL, err


@pytest.mark.parametrize("c,jump", [(32, 277.0), (32, 45.0), (8, 120.0), (8, 40.0), (64, 150.0), (64, 50.0), (16, 90.0)])
def test_attention_online_softmax_rescale_branch(c, jump):
    """The softmax reference maximum of the forward kernels is STALE by design (folded into the score MFMAs' C operand,
    refreshed only when a tile exceeds it by 2^64): one key far above the rest, placed in a LATE tile, (a) beyond the
    headroom — the refresh path: scores recomputed from zero, O and the denominator rescaled — and (b) inside it — the
    reference stays put and 2^(s - m_ref) reaches 2^40 .. 2^50.  `jump` = the spike's score above the rest in log2
    units (guide rule 26)."""
    import sfhip
    dev = _dev()
    g = torch.Generator().manual_seed(1)
    n = 640
    q = torch.randn(1, n, c, generator=g)
    k = torch.randn(1, n, c, generator=g)
    v = torch.randn(1, n, c, generator=g)
    LOG2E = 1.4426950408889634
    k[0, 517] = q[0, 100] * (jump / LOG2E / float(q[0, 100].pow(2).sum()))   # query 100 spikes at key 517 (tile 8)
    k[0, 3] = q[0, 200] * (0.8 * jump / LOG2E / float(q[0, 200].pow(2).sum()))  # early spike then nothing larger
    x = torch.zeros(1, n, c)
    ref = torch.softmax(q.double() @ k.double().transpose(1, 2), -1) @ v.double()
    qa, ka, va, xa = [sfhip.Act(z.view(1, 1, 1, n, c).contiguous().to(dev)) for z in (q, k, v, x)]
    out = sfhip.attention(qa, ka, va, xa, torch.ones(1, device=dev))
    torch.cuda.synchronize()
    err = _rel(out.buf.view(1, n, c), ref)
    _report("attn/rescale_branch c%d jump%d" % (c, jump), err)
    assert err < TOL


@pytest.mark.parametrize("c", [8, 32, 64])
def test_attention_workspace_free_entry_point(c):
    """sf_attn_fwd (no workspace: one launch, never cut into key parts) — the entry the Python binding no longer
    takes — called straight through the C ABI, against the fp64 softmax attention and against sf_attn_fwd_ws."""
    import ctypes
    import sfhip
    dev = _dev()
    g = torch.Generator().manual_seed(40 + c)
    n = 700
    q, k, v, x = [torch.randn(2, n, c, generator=g) * (0.6 if i < 2 else 1.0) for i in range(4)]
    gamma = torch.tensor([0.7])
    ref = gamma.double() * (torch.softmax(q.double() @ k.double().transpose(1, 2), -1) @ v.double()) + x.double()
    qa, ka, va, xa = [sfhip.Act(z.view(2, 1, 1, n, c).contiguous().to(dev)) for z in (q, k, v, x)]
    gam = gamma.to(dev)
    via_ws = sfhip.attention(qa, ka, va, xa, gam)
    out = sfhip.Act(torch.empty(2, 1, 1, n, c, device=dev))
    o_save = torch.empty(2, n, c, device=dev)
    lse = torch.empty(2, n, device=dev)
    vp = ctypes.c_void_p
    rc = sfhip.lib().sf_attn_fwd(vp(qa.buf.data_ptr()), c, vp(ka.buf.data_ptr()), c, vp(va.buf.data_ptr()), c,
                                 vp(xa.buf.data_ptr()), c, vp(gam.data_ptr()), None, None, 0, vp(out.buf.data_ptr()),
                                 c, 0, 2, 1, 1, n, c, 1, vp(o_save.data_ptr()), vp(lse.data_ptr()),
                                 vp(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    assert rc == 0
    assert _rel(out.buf.view(2, n, c), ref) < TOL
    assert _rel(out.buf, via_ws.buf) < 1e-5
    lse_ref = torch.logsumexp(q.double() @ k.double().transpose(1, 2), -1) * 1.4426950408889634
    assert float((lse.double().cpu() - lse_ref).abs().max()) < 1e-3


@pytest.mark.parametrize("c,k,s,cout", [(32, (3, 3, 3), (1, 1, 1), 32), (12, (1, 5, 5), (1, 2, 2), 12),
                                        (27, (3, 3, 3), (1, 2, 2), 27), (16, (3, 3, 3), (1, 1, 1), 13)])
def test_dwconv(c, k, s, cout):
    import sfhip
    dev = _dev()
    g = torch.Generator().manual_seed(c)
    x = torch.randn(2, c, 4, 10, 10, generator=g)
    wt = torch.randn(c, 1, *k, generator=g) / np.sqrt(k[0] * k[1] * k[2])
    p = tuple(kk // 2 for kk in k)
    scale, bias = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.1
    y = F.conv3d(x, wt, None, s, p, 1, c) * scale.view(1, -1, 1, 1, 1) + bias.view(1, -1, 1, 1, 1)
    y = F.relu(y)[:, :cout]
    out = sfhip.dwconv(_ndhwc(x), sfhip.pack_dw_weight(wt.to(dev)), k, s, p, scale=scale.to(dev), bias=bias.to(dev),
                       relu=True, cout=cout)
    torch.cuda.synchronize()
    err = _rel(_back(out), y)
    _report("dwconv/c%d" % c, err)
    assert err < TOL


def test_head_act_mean():
    import sfhip
    dev = _dev()
    g = torch.Generator().manual_seed(2)
    lg = torch.randn(3, 2, 2, 2, 400, generator=g) * 3
    a = sfhip.Act(lg.to(dev))
    for act, ref in ((sfhip.ACT_SOFTMAX, torch.softmax(lg, 4).mean((1, 2, 3))),
                     (sfhip.ACT_RELU, F.relu(lg).mean((1, 2, 3))),
                     (sfhip.ACT_SIGMOID, torch.sigmoid(lg).mean((1, 2, 3)))):
        out = sfhip.head_act_mean(a, act)
        torch.cuda.synchronize()
        assert _rel(out, ref) < 1e-5


@pytest.mark.parametrize("c,shape", [(16, (2, 4, 6, 6)), (7, (2, 4, 5, 3)), (320, (2, 4, 2, 2)), (64, (4, 8, 28, 28))])
def test_channel_stats_are_robust_to_large_mean(c, shape):
    """Train-mode BN statistics on channels with |mean| >> std: the variance must not lose digits to
    E[x^2] - E[x]^2 cancellation (torch's CPU kernel is two-pass)."""
    import sfhip
    g = torch.Generator().manual_seed(c)
    n, t, h, w = shape
    mu = torch.randn(c, generator=g) * 30.0
    sd = torch.rand(c, generator=g) * 0.05 + 0.005
    x = (torch.randn(n, t, h, w, c, generator=g) * sd + mu).float()
    mean, var = sfhip.channel_stats(sfhip.Act(x.to(_dev())))
    torch.cuda.synchronize()
    xd = x.double().reshape(-1, c)
    e1 = float(((mean.double().cpu() - xd.mean(0)).abs() / xd.std(0, unbiased=False)).max())
    e2 = float(((var.double().cpu() - xd.var(0, unbiased=False)).abs() / xd.var(0, unbiased=False)).max())
    _report("channel_stats large-mean c%d rows%d" % (c, n * t * h * w), max(e1, e2))
    assert e1 < 1e-3 and e2 < 1e-4, (e1, e2)


STAT_CASES = [
    # cin, cout, kernel, stride, pad, (N,T,H,W), conv bias (large: |mean| >> std per channel)
    (64, 64, (1, 3, 3), (1, 1, 1), (0, 1, 1), (2, 4, 28, 28), False),
    (72, 24, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 3, 13, 11), True),
    (256, 512, (1, 1, 1), (1, 2, 2), (0, 0, 0), (2, 4, 14, 14), False),
    (512, 128, (3, 1, 1), (1, 1, 1), (1, 0, 0), (2, 4, 7, 7), True),
    (16, 8, (3, 1, 1), (1, 1, 1), (1, 0, 0), (2, 8, 9, 9), False),
    (32, 64, (7, 1, 1), (4, 1, 1), (3, 0, 0), (1, 16, 6, 6), True),
    # conv_small.hip: one record per workgroup
    (8, 8, (1, 3, 3), (1, 1, 1), (0, 1, 1), (2, 4, 20, 28), False),
    (32, 8, (3, 1, 1), (1, 1, 1), (1, 0, 0), (2, 6, 20, 20), True),
    (16, 64, (1, 1, 1), (1, 1, 1), (0, 0, 0), (2, 4, 24, 25), True),
    (32, 128, (1, 1, 1), (1, 1, 1), (0, 0, 0), (2, 4, 23, 24), False),
]


@pytest.mark.parametrize("case", STAT_CASES, ids=["%dx%d_k%d%d%d" % (c[0], c[1], *c[2]) for c in STAT_CASES])
def test_conv_epilogue_statistics(case):
    """sf_conv_fwd_stats + sf_bn_train_stats_merge (batch statistics taken where the conv stores its outputs) against
    fp64 statistics of the conv's own output and against sf_bn_train_stats on it: same mean / var / scale / shift and
    the same running-stat update, also with a conv bias that puts |mean| at 30x the standard deviation."""
    import sfhip
    cin, cout, k, s, p, shp, big_bias = case
    dev = _dev()
    g = torch.Generator().manual_seed(cin * 7 + cout)
    n, t, h, w = shp
    x = torch.randn(n, cin, t, h, w, generator=g)
    wt = torch.randn(cout, cin, *k, generator=g) / np.sqrt(cin * k[0] * k[1] * k[2])
    bias = (torch.randn(cout, generator=g) * 30.0) if big_bias else None
    wp = sfhip.pack_conv_weight(wt.to(dev))
    xa = _ndhwc(x)
    z, parts = sfhip.conv(xa, wp, k, s, p, bias=bias.to(dev) if big_bias else None, stats=True, out_reserve=(4, 4))
    assert parts is not None, "the per-wavefront conv kernels take this shape and must produce statistics"
    z0 = sfhip.conv(xa, wp, k, s, p, bias=bias.to(dev) if big_bias else None, out_reserve=(4, 4))
    torch.cuda.synchronize()
    assert torch.equal(_back(z), _back(z0)), "the statistics must not change what the conv stores"
    gamma = (torch.rand(cout, generator=g) + 0.5).to(dev)
    beta = torch.randn(cout, generator=g).to(dev)
    rm0, rv0 = torch.randn(cout, generator=g).to(dev), (torch.rand(cout, generator=g) + 0.5).to(dev)
    rm1, rv1 = rm0.clone(), rv0.clone()
    m1, i1, sc1, sh1 = sfhip.bn_train_stats_merge(parts, cout, gamma, beta, 1e-5, 0.1, rm1, rv1)
    rm2, rv2 = rm0.clone(), rv0.clone()
    m2, i2, sc2, sh2 = sfhip.bn_train_stats(z, gamma, beta, 1e-5, 0.1, rm2, rv2)
    torch.cuda.synchronize()
    zd = _back(z).double().permute(0, 2, 3, 4, 1).reshape(-1, cout)
    mean, var = zd.mean(0), zd.var(0, unbiased=False)
    e_mean = float(((m1.double().cpu() - mean).abs() / var.sqrt()).max())
    e_is = _rel(i1, torch.rsqrt(var + 1e-5))
    _report("conv epilogue stats %s" % (case[:3],), max(e_mean, e_is))
    assert e_mean < 1e-4 and e_is < 1e-4, (e_mean, e_is)
    # the two kernels sum in different orders: means agree to 1e-5 of a standard deviation, the rest to 5e-5
    assert float(((m1 - m2).double().cpu().abs() / var.sqrt()).max()) < 1e-5
    for a, b in ((i1, i2), (sc1, sc2), (sh1, sh2), (rm1, rm2), (rv1, rv2)):
        assert _rel(a, b) < 5e-5
    nrows = zd.shape[0]
    assert _rel(rv1, 0.9 * rv0.cpu().double() + 0.1 * var * nrows / (nrows - 1)) < 1e-5


def test_copy_channels_shuffle():
    import sfhip
    dev = _dev()
    x = torch.randn(1, 2, 3, 3, 10)
    out = sfhip.Act(torch.zeros(1, 2, 3, 3, 20, device=dev))
    sfhip.copy_channels(sfhip.Act(x.to(dev)), sfhip.Act(out.buf, 0, 20), out_cmul=2)
    torch.cuda.synchronize()
    assert torch.equal(out.buf[..., 0::2].cpu(), x)


def test_ops_reject_cpu_tensors():
    import sfhip
    with pytest.raises(sfhip.SfhipError):
        sfhip.from_ncthw(torch.randn(1, 3, 2, 4, 4))
