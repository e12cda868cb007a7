"""conv_rows.hip — forward and data gradient of the small-channel stride-1 "same" convolutions (the Fast pathway's
1x1x1 / 3x1x1 / 1x3x3 layers, resnet_helper.py:182-223 at dim_inner 8 .. 64; the lateral and q | k | v projections)
forced onto every shape the kernel covers (sf_conv_tune(22, 2)) against fp64 convolutions: ragged row counts (partial
last stage, partial last workgroup), clip / frame / row borders inside a stage, channel-slice inputs and outputs,
training-mode statistics from the epilogue, the eval epilogue (scale, bias, residual, ReLU), the data gradient written
and accumulated; and the route itself (sf_conv_rows_parts > 0 exactly when switched on)."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
TOL = 2e-5  # f32 MFMA, K <= 1152: fp32 accumulation noise only

# name, Cin, Cout, kernel, (N, T, H, W)
SHAPES = [
    ("p_8_32", 8, 32, (1, 1, 1), (3, 5, 23, 21)),
    ("p_32_8", 32, 8, (1, 1, 1), (3, 5, 23, 21)),
    ("p_16_64", 16, 64, (1, 1, 1), (2, 6, 19, 17)),
    ("p_128_32", 128, 32, (1, 1, 1), (2, 4, 15, 14)),
    ("p_32_256", 32, 256, (1, 1, 1), (2, 3, 14, 15)),
    ("p_24_40", 24, 40, (1, 1, 1), (2, 5, 17, 13)),     # rows that fill their LDS width / column tiles partly
    ("t_32_8", 32, 8, (3, 1, 1), (3, 6, 23, 21)),
    ("t_64_16", 64, 16, (3, 1, 1), (2, 7, 19, 17)),
    ("t_128_32", 128, 32, (3, 1, 1), (2, 5, 14, 13)),   # two channel blocks per position stage
    ("t_16_64", 16, 64, (3, 1, 1), (2, 5, 15, 14)),
    ("t_8_16_T1", 8, 16, (3, 1, 1), (5, 1, 33, 31)),    # T = 1: both temporal neighbours outside (window form)
    ("t_32_8_ring", 32, 8, (3, 1, 1), (2, 8, 40, 40)),  # ring over t: several t segments, 25 blocks of 64 per frame
    ("t_8_32_ring", 8, 32, (3, 1, 1), (1, 5, 56, 56)),
    ("t_64_16_ring", 64, 16, (3, 1, 1), (2, 6, 28, 28)),  # 64-float rows
    ("t_16_64_T2", 16, 64, (3, 1, 1), (3, 2, 23, 21)),    # T = 2: every frame is a border frame
    ("s_16_16", 16, 16, (1, 3, 3), (3, 4, 23, 21)),
    ("s_32_32", 32, 32, (1, 3, 3), (2, 5, 14, 14)),
    ("s_8_8", 8, 8, (1, 3, 3), (2, 6, 29, 27)),
    ("s_8_32_w56", 8, 32, (1, 3, 3), (2, 2, 56, 56)),   # the halo of a 56-wide frame
]


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


@pytest.fixture
def tune():
    import sfhip
    L = sfhip.lib()
    yield L
    L.sf_conv_tune(22, 1)


def _rel(a, b):
    return float((a.double() - b).abs().max() / b.abs().max())


def _act(x):
    import sfhip
    return sfhip.Act(x.permute(0, 2, 3, 4, 1).contiguous())


def _ncthw(a):
    return a.buf[..., a.coff:a.coff + a.C].permute(0, 4, 1, 2, 3)


@pytest.mark.parametrize("shape", SHAPES, ids=[s[0] for s in SHAPES])
def test_conv_rows_forward_backward(shape, tune):
    import sfhip
    name, cin, cout, k, (n, t, h, w) = shape
    dev = _dev()
    p = tuple(kk // 2 for kk in k)
    g = torch.Generator().manual_seed(cin * 7 + cout + k[0] + k[1])
    x = torch.randn(n, cin, t, h, w, generator=g).to(dev)
    wt = (torch.randn(cout, cin, *k, generator=g) / np.sqrt(cin * k[0] * k[1] * k[2])).to(dev)
    bias = (torch.randn(cout, generator=g) * 0.1 + 3.0).to(dev)   # a mean far from zero: the statistics' shift matters
    scale = (torch.rand(cout, generator=g) + 0.5).to(dev)
    res = torch.randn(n, cout, t, h, w, generator=g).to(dev)
    dy = torch.randn(n, cout, t, h, w, generator=g).to(dev)
    assert tune.sf_conv_tune(22, 2) == 0
    wp, wtp = sfhip.pack_conv_weight_pair(wt)
    xa = _act(x)
    d = sfhip.ConvDesc(xa.N, xa.T, xa.H, xa.W, cin, xa.cs, xa.coff, xa.T, xa.H, xa.W, cout, cout, 0, 1, k[0], k[1], k[2],
                       1, 1, 1, p[0], p[1], p[2], 1, 1, 1, wp.shape[2], 0, 0, 0, 0)
    assert tune.sf_conv_rows_parts(ctypes.byref(d)) > 0, name
    wide = sfhip.Act(torch.randn(xa.buf.shape[:-1] + (cin + 12,), device=dev), 8, cin)  # a slice of a wider buffer
    wide.buf[..., 8:8 + cin] = xa.buf
    ref = F.conv3d(x.double(), wt.double(), bias.double(), padding=p)
    flat = ref.transpose(0, 1).reshape(cout, -1)
    for a in (xa, wide):
        z, st = sfhip.conv(a, wp, k, padding=p, bias=bias, stats=True)
        assert _rel(_ncthw(z), ref) < TOL, name
        assert st is not None, "the epilogue leaves the training-mode statistics"
        ones = torch.ones(cout, device=dev)
        mean, invstd, _, _ = sfhip.bn_train_stats_merge(st, cout, ones, ones, 1e-5, 0.1, None, None)
        assert float((mean.double() - flat.mean(1)).abs().max()) < 1e-5 * float(flat.std() + flat.mean(1).abs().max())
        assert _rel(invstd, torch.rsqrt(flat.var(1, unbiased=False) + 1e-5)) < 1e-4
    # eval epilogue into a channel slice of a wider output
    outw = sfhip.Act(torch.zeros(xa.buf.shape[:-1] + (cout + 8,), device=dev), 4, cout)
    y = sfhip.conv(xa, wp, k, padding=p, scale=scale, bias=bias, relu=True, res=_act(res), out=outw)
    ref2 = torch.relu(F.conv3d(x.double(), wt.double(), padding=p) * scale.double().view(1, -1, 1, 1, 1) +
                      bias.double().view(1, -1, 1, 1, 1) + res.double())
    assert _rel(_ncthw(y), ref2) < TOL, name
    assert float(outw.buf[..., :4].abs().max()) == 0.0 and float(outw.buf[..., 4 + cout:].abs().max()) == 0.0
    # data gradient (the transposed pack, mirrored taps): written, then accumulated
    xd = x.double().requires_grad_(True)
    F.conv3d(xd, wt.double(), padding=p).backward(dy.double())
    dxa = sfhip.conv_dgrad(_act(dy), wtp, xa, k, padding=p)
    assert _rel(_ncthw(dxa), xd.grad) < TOL, name
    sfhip.conv_dgrad(_act(dy), wtp, xa, k, padding=p, out=dxa, accumulate=True)
    assert _rel(_ncthw(dxa), 2 * xd.grad) < TOL, name
    # off: the other kernels serve the shape, same results at their own tolerance
    assert tune.sf_conv_tune(22, 0) == 0
    assert tune.sf_conv_rows_parts(ctypes.byref(d)) == 0
    z0 = sfhip.conv(xa, wp, k, padding=p, bias=bias)
    assert _rel(_ncthw(z0), ref) < 2e-4, name


def test_conv_rows_is_bit_reproducible(tune):
    import sfhip
    dev = _dev()
    g = torch.Generator().manual_seed(5)
    x = _act(torch.randn(3, 64, 6, 28, 28, generator=g).to(dev))
    wt = (torch.randn(16, 64, 3, 1, 1, generator=g) / 14.0).to(dev)
    wp, _ = sfhip.pack_conv_weight_pair(wt)
    tune.sf_conv_tune(22, 2)
    a, sa = sfhip.conv(x, wp, (3, 1, 1), padding=(1, 0, 0), stats=True)
    b, sb = sfhip.conv(x, wp, (3, 1, 1), padding=(1, 0, 0), stats=True)
    assert torch.equal(a.buf, b.buf) and sa[1] == sb[1] and torch.equal(sa[0][:sa[1] * 4 * 16], sb[0][:sb[1] * 4 * 16])
