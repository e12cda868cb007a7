"""Row-march depthwise kernels (dwconv_march.hip: kT x 3 x 3, stride 1, "same"; round 6: 1 x 3 x 3 / 1 x 5 x 5 at
stride (1, 2, 2)) through the C ABI (sf_dwconv_fwd /
sf_dwconv_dgrad / sf_dwconv_wgrad) against fp64 torch and against the position-per-thread kernels they replace
(sf_conv_tune(30, 0)): channel counts that are no multiples of 4 (the Fast pathway's 2 / 6 / 10), channel SLICES of wider
buffers (GhostModule: primary conv -> channels [0, init), cheap operation -> [init, oup), ghostnet_helper.py:72-100),
Cout < Cin (the [:oup] cut), the whole conv epilogue, heights that the march length does not divide, one-row and
one-frame inputs."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
TOL = 2e-5

CASES = [
    # name, N, C, T, H, W, kT, in (pitch, offset), out (pitch, offset), Cout
    ("fast_c6_slice", 2, 6, 4, 23, 20, 3, (12, 0), (12, 6), 6),
    ("fast_c2", 1, 2, 5, 17, 16, 3, (2, 0), (2, 0), 2),
    ("fast_c10_cut", 2, 10, 3, 9, 14, 3, (20, 0), (19, 10), 9),
    ("slow_c24_vec4_slice", 2, 24, 3, 14, 14, 3, (48, 0), (48, 24), 24),
    ("slow_c32_vec4_cut", 1, 32, 2, 7, 7, 3, (32, 0), (60, 32), 28),
    ("c16_kt1_vec4", 2, 16, 3, 12, 10, 1, (16, 0), (16, 0), 16),
    ("c5_kt1_odd_offsets", 1, 5, 2, 8, 6, 1, (9, 3), (7, 1), 5),
    ("one_row_one_frame", 1, 8, 1, 1, 9, 3, (8, 0), (8, 0), 8),
    ("tall_march_c4", 1, 4, 2, 113, 6, 3, (4, 0), (4, 0), 4),
]


def _rel(a, b):
    return float((a.double().cpu() - b.double().cpu()).abs().max() / max(float(b.double().abs().max()), 1e-30))


def _slice_act(sfhip, dense, pitch, off, fill=0.0):
    """dense [N, C, T, H, W] -> an Act that is channels [off, off + C) of a [N, T, H, W, pitch] buffer"""
    n, c, t, h, w = dense.shape
    buf = torch.full((n, t, h, w, pitch), fill, device=dense.device)
    buf[..., off:off + c] = dense.permute(0, 2, 3, 4, 1)
    return sfhip.Act(buf, off, c)


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_march_forward_with_the_conv_epilogue(case):
    import sfhip
    name, n, c, t, h, w, kt, (ipitch, ioff), (opitch, ooff), cout = case
    dev = torch.device("cuda:0")
    torch.manual_seed(len(name))
    k, p = (kt, 3, 3), (kt // 2, 1, 1)
    x = torch.randn(n, c, t, h, w, device=dev)
    wt = torch.randn(c, 1, *k, device=dev) / np.sqrt(kt * 9)
    scale, bias = torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev) * 0.1
    res = torch.randn(n, cout, t, h, w, device=dev)
    ref = F.conv3d(x.double(), wt.double(), None, 1, p, 1, c)
    ref = ref * scale.double().view(1, -1, 1, 1, 1) + bias.double().view(1, -1, 1, 1, 1)
    ref = F.relu(ref[:, :cout] + res.double())
    xa = _slice_act(sfhip, x, ipitch, ioff, fill=7.0)           # the channels beside the slice must not leak in
    ra = _slice_act(sfhip, res, cout, 0)
    wp = sfhip.pack_dw_weight(wt)
    outs = []
    for on in (1, 0):
        assert sfhip.lib().sf_conv_tune(30, on) == 0
        obuf = torch.full((n, t, h, w, opitch), -3.0, device=dev)
        out = sfhip.Act(obuf, ooff, cout)
        sfhip.dwconv(xa, wp, k, (1, 1, 1), p, scale=scale, bias=bias, relu=True, res=ra, out=out, cout=cout)
        torch.cuda.synchronize()
        outs.append(obuf)
    sfhip.lib().sf_conv_tune(30, 1)
    got = outs[0][..., ooff:ooff + cout].permute(0, 4, 1, 2, 3)
    assert _rel(got, ref) < TOL, name
    # nothing outside the output slice was written
    assert float((outs[0][..., :ooff] + 3.0).abs().sum()) == 0 and float((outs[0][..., ooff + cout:] + 3.0).abs().sum()) == 0
    # and the generic kernel agrees to rounding (same taps, another summation order)
    assert _rel(outs[0], outs[1]) < 1e-5, name


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_march_data_and_weight_gradients(case):
    import sfhip
    name, n, c, t, h, w, kt, (ipitch, ioff), (opitch, ooff), cout = case
    dev = torch.device("cuda:0")
    torch.manual_seed(50 + len(name))
    k, p = (kt, 3, 3), (kt // 2, 1, 1)
    x = torch.randn(n, c, t, h, w, device=dev)
    wt = torch.randn(c, 1, *k, device=dev) / np.sqrt(kt * 9)
    dz = torch.randn(n, c, t, h, w, device=dev)
    dx0 = torch.randn(n, c, t, h, w, device=dev)               # the data gradient ACCUMULATES into dx
    xd = x.double().requires_grad_(True)
    wd = wt.double().requires_grad_(True)
    (F.conv3d(xd, wd, None, 1, p, 1, c) * dz.double()).sum().backward()
    xa = _slice_act(sfhip, x, ipitch, ioff, fill=5.0)
    dza = _slice_act(sfhip, dz, opitch if opitch >= ooff + c else ooff + c, ooff, fill=9.0)
    wp = sfhip.pack_dw_weight(wt)
    res = []
    for on in (1, 0):
        assert sfhip.lib().sf_conv_tune(30, on) == 0
        dxa = _slice_act(sfhip, dx0, ipitch, ioff, fill=-1.0)
        dw = sfhip.dwconv_bwd(xa, dza, wp, k, (1, 1, 1), p, dx=dxa)
        torch.cuda.synchronize()
        res.append((dxa.buf.clone(), dw.clone()))
    sfhip.lib().sf_conv_tune(30, 1)
    dxb, dw = res[0]
    got_dx = dxb[..., ioff:ioff + c].permute(0, 4, 1, 2, 3) - dx0
    assert _rel(got_dx, xd.grad) < TOL, name
    assert float((dxb[..., :ioff] + 1.0).abs().sum()) == 0 and float((dxb[..., ioff + c:] + 1.0).abs().sum()) == 0
    assert _rel(dw.t().reshape(wt.shape), wd.grad) < TOL, name
    assert _rel(res[0][0], res[1][0]) < 1e-5 and _rel(res[0][1], res[1][1]) < 1e-5, name
    # bit-reproducible: a second run gives the same bits
    dxa = _slice_act(sfhip, dx0, ipitch, ioff, fill=-1.0)
    dw2 = sfhip.dwconv_bwd(xa, dza, wp, k, (1, 1, 1), p, dx=dxa)
    torch.cuda.synchronize()
    assert torch.equal(dw2, dw) and torch.equal(dxa.buf, dxb)


S2_CASES = [
    # name, N, C, T, H, W, K, in (pitch, offset), out (pitch, offset), Cout  — 1 x K x K, stride (1, 2, 2), padding K / 2
    ("s2_k3_c96_vec4", 2, 96, 2, 28, 28, 3, (96, 0), (96, 0), 96),
    ("s2_k3_c12_odd_hw", 2, 12, 3, 23, 21, 3, (12, 0), (12, 0), 12),
    ("s2_k3_c6_slice", 2, 6, 4, 17, 20, 3, (12, 6), (15, 4), 6),
    ("s2_k5_c20_vec4", 2, 20, 3, 28, 28, 5, (20, 0), (20, 0), 20),
    ("s2_k5_c14_novec", 1, 14, 4, 19, 22, 5, (14, 0), (14, 0), 14),
    ("s2_k5_c144_slice", 1, 144, 2, 14, 14, 5, (160, 8), (144, 0), 144),
    ("s2_k5_c8_cut", 2, 8, 2, 9, 9, 5, (8, 0), (12, 4), 5),
    ("s2_k3_one_row", 1, 8, 2, 1, 9, 3, (8, 0), (8, 0), 8),
    ("s2_k5_two_rows", 1, 4, 1, 2, 3, 5, (4, 0), (4, 0), 4),
    ("s2_k3_tall_march", 1, 4, 1, 225, 6, 3, (4, 0), (4, 0), 4),
    ("s2_k5_tall_march", 1, 4, 1, 226, 7, 5, (4, 0), (4, 0), 4),
]


def _s2_dims(h, w, K):
    p = K // 2
    return (h + 2 * p - K) // 2 + 1, (w + 2 * p - K) // 2 + 1


@pytest.mark.parametrize("case", S2_CASES, ids=[c[0] for c in S2_CASES])
def test_stride2_march_forward_with_the_conv_epilogue(case):
    """The down-sampling depthwise layers (ghostnet_helper.py:114-120: conv_dw, kernel 3 | 5, stride 2; ShuffleNetV2's
    stride-2 branches) as marches with a doubled row step: against fp64 torch and the position-per-thread kernel
    (sf_conv_tune(31, 0))."""
    import sfhip
    name, n, c, t, h, w, K, (ipitch, ioff), (opitch, ooff), cout = case
    dev = torch.device("cuda:0")
    torch.manual_seed(len(name))
    k, s, p = (1, K, K), (1, 2, 2), (0, K // 2, K // 2)
    ho, wo = _s2_dims(h, w, K)
    x = torch.randn(n, c, t, h, w, device=dev)
    wt = torch.randn(c, 1, *k, device=dev) / K
    scale, bias = torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev) * 0.1
    res = torch.randn(n, cout, t, ho, wo, device=dev)
    ref = F.conv3d(x.double(), wt.double(), None, s, p, 1, c)
    assert ref.shape[-2:] == (ho, wo)
    ref = ref * scale.double().view(1, -1, 1, 1, 1) + bias.double().view(1, -1, 1, 1, 1)
    ref = F.relu(ref[:, :cout] + res.double())
    xa = _slice_act(sfhip, x, ipitch, ioff, fill=7.0)
    ra = _slice_act(sfhip, res, cout, 0)
    wp = sfhip.pack_dw_weight(wt)
    outs = []
    for on in (1, 0):
        assert sfhip.lib().sf_conv_tune(31, on) == 0
        obuf = torch.full((n, t, ho, wo, opitch), -3.0, device=dev)
        out = sfhip.Act(obuf, ooff, cout)
        sfhip.dwconv(xa, wp, k, s, p, scale=scale, bias=bias, relu=True, res=ra, out=out, cout=cout)
        torch.cuda.synchronize()
        outs.append(obuf)
    sfhip.lib().sf_conv_tune(31, 1)
    got = outs[0][..., ooff:ooff + cout].permute(0, 4, 1, 2, 3)
    assert _rel(got, ref) < TOL, name
    assert float((outs[0][..., :ooff] + 3.0).abs().sum()) == 0 and float((outs[0][..., ooff + cout:] + 3.0).abs().sum()) == 0
    assert _rel(outs[0], outs[1]) < 1e-5, name


@pytest.mark.parametrize("case", S2_CASES, ids=[c[0] for c in S2_CASES])
def test_stride2_march_data_and_weight_gradients(case):
    import sfhip
    name, n, c, t, h, w, K, (ipitch, ioff), (opitch, ooff), cout = case
    dev = torch.device("cuda:0")
    torch.manual_seed(70 + len(name))
    k, s, p = (1, K, K), (1, 2, 2), (0, K // 2, K // 2)
    ho, wo = _s2_dims(h, w, K)
    x = torch.randn(n, c, t, h, w, device=dev)
    wt = torch.randn(c, 1, *k, device=dev) / K
    dz = torch.randn(n, c, t, ho, wo, device=dev)
    dx0 = torch.randn(n, c, t, h, w, device=dev)               # the data gradient ACCUMULATES into dx
    xd = x.double().requires_grad_(True)
    wd = wt.double().requires_grad_(True)
    (F.conv3d(xd, wd, None, s, p, 1, c) * dz.double()).sum().backward()
    xa = _slice_act(sfhip, x, ipitch, ioff, fill=5.0)
    dza = _slice_act(sfhip, dz, opitch if opitch >= ooff + c else ooff + c, ooff, fill=9.0)
    wp = sfhip.pack_dw_weight(wt)
    res = []
    for on in (1, 0):
        assert sfhip.lib().sf_conv_tune(31, on) == 0
        dxa = _slice_act(sfhip, dx0, ipitch, ioff, fill=-1.0)
        dw = sfhip.dwconv_bwd(xa, dza, wp, k, s, p, dx=dxa)
        torch.cuda.synchronize()
        res.append((dxa.buf.clone(), dw.clone()))
    sfhip.lib().sf_conv_tune(31, 1)
    dxb, dw = res[0]
    got_dx = dxb[..., ioff:ioff + c].permute(0, 4, 1, 2, 3) - dx0
    assert _rel(got_dx, xd.grad) < TOL, name
    assert float((dxb[..., :ioff] + 1.0).abs().sum()) == 0 and float((dxb[..., ioff + c:] + 1.0).abs().sum()) == 0
    assert _rel(dw.t().reshape(wt.shape), wd.grad) < TOL, name
    assert _rel(res[0][0], res[1][0]) < 1e-5 and _rel(res[0][1], res[1][1]) < 1e-5, name
    dxa = _slice_act(sfhip, dx0, ipitch, ioff, fill=-1.0)
    dw2 = sfhip.dwconv_bwd(xa, dza, wp, k, s, p, dx=dxa)
    torch.cuda.synchronize()
    assert torch.equal(dw2, dw) and torch.equal(dxa.buf, dxb)


def test_shapes_outside_the_march_take_the_generic_kernels():
    """temporal stride, kT = 3 with a spatial stride, 7 x 7 and dilated layers are not the marches': same results with
    the knob on and off, bit for bit"""
    import sfhip
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    for k, s in (((3, 3, 3), (1, 2, 2)), ((1, 7, 7), (1, 2, 2)), ((1, 3, 3), (2, 2, 2)), ((1, 5, 5), (1, 2, 1))):
        p = tuple(kk // 2 for kk in k)
        x = torch.randn(2, 12, 4, 10, 10, device=dev)
        wt = torch.randn(12, 1, *k, device=dev)
        xa = sfhip.from_ncthw(x)
        outs = []
        for on in (1, 0):
            sfhip.lib().sf_conv_tune(30, on)
            outs.append(sfhip.dwconv(xa, sfhip.pack_dw_weight(wt), k, s, p).buf.clone())
        sfhip.lib().sf_conv_tune(30, 1)
        assert torch.equal(outs[0], outs[1])
        ref = F.conv3d(x.double(), wt.double(), None, s, p, 1, 12)
        assert _rel(sfhip.to_ncthw(sfhip.Act(outs[0])), ref) < TOL
