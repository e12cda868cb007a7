"""Grouped convolutions (1 < groups < channels) on the HIP path — engine.grouped_conv: ONE launch per layer and kind
(sf_conv_fwd_grouped / sf_conv_wgrad_grouped: the block-diagonal GEMM's group is the grid's z index) on channel windows
of the same buffers — against torch's nn.Conv3d(groups=G) on the same parameters: ShuffleNet-v1's grouped
1x1x1 convs with channel_shuffle folded into the stores (shufflenet_helper.py:22-34, 48-63, 73) and ResNeXt's grouped
1x3x3 (resnet_helper.py:196-205, RESNET.NUM_GROUPS > 1).  Eval (folded BN epilogue) and the taped training path
(batch-statistics BN, gradients of input / weight / BN)."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

CASES = [
    # name, Cin, Cout, groups, kernel, stride, pad, (N, T, H, W), shuffle
    ("shuffle_v1_54_240_g3", 54, 240, 3, (1, 1, 1), (1, 1, 1), (0, 0, 0), (2, 3, 9, 11), True),
    ("shuffle_v1_12_30_g3", 12, 30, 3, (1, 1, 1), (1, 1, 1), (0, 0, 0), (2, 8, 9, 11), True),
    ("shuffle_v1_240_240_g3_noshuffle", 240, 240, 3, (1, 1, 1), (1, 1, 1), (0, 0, 0), (2, 3, 9, 11), False),
    ("resnext_64_64_g4_s3", 64, 64, 4, (1, 3, 3), (1, 1, 1), (0, 1, 1), (2, 2, 12, 12), False),
    ("resnext_128_128_g32_s3_stride2", 128, 128, 32, (1, 3, 3), (1, 2, 2), (0, 1, 1), (1, 2, 14, 14), False),
]


def _shuffle(x, g):
    b, c = x.shape[:2]
    return x.view(b, g, c // g, *x.shape[2:]).transpose(1, 2).reshape(x.shape)


def _rel(a, b):
    return float((a.double().cpu() - b.double().cpu()).abs().max() / b.double().abs().max().cpu())


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_grouped_conv_eval_and_train(case):
    import sfhip
    from slowfast.models import engine
    name, cin, cout, G, k, s, p, (n, t, h, w), shuffle = case
    dev = torch.device("cuda:0")
    torch.manual_seed(len(name))
    conv = nn.Conv3d(cin, cout, k, s, p, groups=G, bias=False).to(dev)
    bn = nn.BatchNorm3d(cout).to(dev)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.normal_()
        bn.running_mean.normal_()
        bn.running_var.uniform_(0.5, 1.5)
    x = torch.randn(n, cin, t, h, w, device=dev)
    xa = sfhip.from_ncthw(x)
    # ---- eval: conv + folded BN + ReLU (+ shuffle as index math of the stores)
    bn.eval()
    with torch.no_grad():
        ref = F.relu(bn.double()(F.conv3d(x.double(), conv.weight.double(), None, s, p, 1, G)))
        ref = _shuffle(ref, G) if shuffle else ref
        bn.float()
        y = engine.conv_bn_act(xa, conv, bn, relu=True, shuffle=shuffle)
    assert _rel(sfhip.to_ncthw(y), ref) < 2e-4, name
    # ---- training: taped, batch statistics; gradients against autograd in fp64
    bn.train()
    up = torch.randn_like(ref, dtype=torch.float32)
    t_ = engine.Tape()
    with engine.taping(t_):
        y = engine.conv_bn_act(xa, conv, bn, relu=True, shuffle=shuffle)
    g = t_.grad_of(y)
    g.buf[..., g.coff:g.coff + g.C].copy_(up.permute(0, 2, 3, 4, 1))
    dxa = t_.grad_of(xa)  # the (zero-initialised) buffer the data gradients accumulate into
    with torch.no_grad():
        t_.backward()
    xd = x.double().requires_grad_(True)
    wd = conv.weight.detach().double().requires_grad_(True)
    gam, bet = bn.weight.detach().double().requires_grad_(True), bn.bias.detach().double().requires_grad_(True)
    z = F.conv3d(xd, wd, None, s, p, 1, G)
    yr = F.relu(F.batch_norm(z, None, None, gam, bet, True, 0.1, bn.eps))
    yr = _shuffle(yr, G) if shuffle else yr
    assert _rel(sfhip.to_ncthw(y), yr.detach()) < 2e-4, name
    (yr * up.double()).sum().backward()
    dx = sfhip.to_ncthw(dxa)
    assert _rel(t_.pgrads[conv.weight], wd.grad) < 5e-4, name
    assert _rel(t_.pgrads[bn.weight], gam.grad) < 5e-4 and _rel(t_.pgrads[bn.bias], bet.grad) < 5e-4, name
    assert _rel(dx, xd.grad) < 5e-4, name


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_grouped_entry_points_through_the_c_abi(case):
    """sf_conv_fwd_grouped with the full epilogue (scale, bias, residual, ReLU; shuffled stores), its transposed form
    accumulating into an existing gradient, and sf_conv_wgrad_grouped + sf_conv_wgrad_finish accumulating into a tensor
    in nn.Conv3d's grouped layout — each ONE C-ABI call — against fp64 torch."""
    import sfhip
    name, cin, cout, G, k, s, p, (n, t, h, w), shuffle = case
    dev = torch.device("cuda:0")
    torch.manual_seed(100 + len(name))
    wt = torch.randn(cout, cin // G, *k, device=dev) * 0.2
    x = torch.randn(n, cin, t, h, w, device=dev)
    scale, bias = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev)
    wp, wtp = sfhip.pack_grouped_weight_pair(wt, G)
    assert tuple(wp.shape) == (cout, k[0] * k[1] * k[2], (cin // G + 15) // 16 * 16)
    assert tuple(wtp.shape) == (cin, k[0] * k[1] * k[2], (cout // G + 15) // 16 * 16)
    z = F.conv3d(x.double(), wt.double(), None, s, p, 1, G)
    res = torch.randn(z.shape, device=dev)
    xa = sfhip.from_ncthw(x)
    sfhip.EVENT_TRACE = []
    try:
        # ---- forward, un-shuffled, whole epilogue
        y = sfhip.conv_grouped(xa, wp, G, k, s, p, scale=scale, bias=bias, relu=True, res=sfhip.from_ncthw(res))
        ref = F.relu(z * scale.double().view(1, -1, 1, 1, 1) + bias.double().view(1, -1, 1, 1, 1) + res.double())
        assert _rel(sfhip.to_ncthw(y), ref) < 2e-5, name
        # ---- forward, shuffled stores (no residual: shufflenet_helper.py:73 shuffles conv1's BN-ReLU output)
        y = sfhip.conv_grouped(xa, wp, G, k, s, p, scale=scale, bias=bias, relu=True, shuffle=True)
        ref = _shuffle(F.relu(z * scale.double().view(1, -1, 1, 1, 1) + bias.double().view(1, -1, 1, 1, 1)), G)
        assert _rel(sfhip.to_ncthw(y), ref) < 2e-5, name
        # ---- data gradient accumulated over an existing gradient, weight gradient accumulated into the parameter's
        dz = torch.randn(z.shape, device=dev)
        dza = sfhip.from_ncthw(dz)
        dx0 = torch.randn_like(x)
        dxa = sfhip.from_ncthw(dx0)
        sfhip.conv_dgrad_grouped(dza, wtp, G, xa, k, s, p, out=dxa, accumulate=True)
        dw0 = torch.randn_like(wt)
        dw = dw0.clone()
        sfhip.conv_wgrad_grouped(xa, dza, G, k, s, p, cin_pad=wp.shape[2], finish_into=dw)
        calls = len(sfhip.EVENT_TRACE)
    finally:
        sfhip.EVENT_TRACE = None
    assert calls == 4, calls   # one C-ABI call per conv launch (the finish is not traced)
    xd = x.double().requires_grad_(True)
    wd = wt.double().requires_grad_(True)
    (F.conv3d(xd, wd, None, s, p, 1, G) * dz.double()).sum().backward()
    assert _rel(sfhip.to_ncthw(dxa) - dx0, xd.grad) < 2e-5, name
    assert _rel(dw - dw0, wd.grad) < 2e-5, name
    # packed return form == the finished one
    dwp = sfhip.conv_wgrad_grouped(xa, dza, G, k, s, p, cin_pad=wp.shape[2])
    assert _rel(sfhip.unpack_conv_weight_grad(dwp, tuple(wt.shape)), wd.grad) < 2e-5, name


def test_grouped_entry_points_reject_what_they_cannot_run():
    import ctypes
    import sfhip
    dev = torch.device("cuda:0")
    x = sfhip.from_ncthw(torch.randn(1, 12, 2, 4, 4, device=dev))
    wp, _ = sfhip.pack_grouped_weight_pair(torch.randn(30, 4, 1, 1, 1, device=dev), 3)
    with pytest.raises(sfhip.SfhipError):     # 12 input channels are not divisible by 5 groups
        _raw_grouped(sfhip, x, wp, 5)
    _raw_grouped(sfhip, x, wp, 3)             # the same call with a divisor is accepted
    d = sfhip.ConvDesc(1, 2, 4, 4, 12, 12, 0, 2, 4, 4, 30, 30, 0, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 1, 1, 1, 16, 0, 0, 0, 0)
    assert sfhip.lib().sf_conv_wgrad_grouped_splits(ctypes.byref(d), 5) == 0
    assert sfhip.lib().sf_conv_wgrad_grouped_splits(ctypes.byref(d), 3) > 0


def _raw_grouped(sfhip, x, wp, groups):
    import ctypes
    out = sfhip.new_act(x, x.N, x.T, x.H, x.W, wp.shape[0])
    d = sfhip.ConvDesc(x.N, x.T, x.H, x.W, x.C, x.cs, x.coff, x.T, x.H, x.W, wp.shape[0], out.cs, out.coff, 1,
                       1, 1, 1, 1, 1, 1, 0, 0, 0, 1, 1, 1, wp.shape[2], 0, 0, 0, 0)
    sfhip._check(sfhip.lib().sf_conv_fwd_grouped(ctypes.byref(d), groups, 0, x.ptr(), sfhip._ptr(wp), None, None, None,
                                                 out.ptr(), sfhip._stream()), "sf_conv_fwd_grouped")


@pytest.mark.parametrize("c,g,coff", [(240, 3, 0), (24, 4, 0), (30, 3, 6)])
def test_channel_shuffle_one_launch_and_its_inverse(c, g, coff):
    """sf_channel_shuffle == shufflenet_helper.py:22-34's view / transpose / flatten; groups = C/G undoes it (the
    backward's gather, accumulating)."""
    import sfhip
    dev = torch.device("cuda:0")
    torch.manual_seed(c)
    x = torch.randn(2, c, 3, 5, 7, device=dev)
    xa = sfhip.from_ncthw(x)
    wide = torch.zeros(2, 3, 5, 7, c + coff + 2, device=dev)
    out = sfhip.Act(wide, coff, c)
    sfhip.channel_shuffle(xa, out, g)
    got = wide[..., coff:coff + c].permute(0, 4, 1, 2, 3)
    assert torch.equal(got, _shuffle(x, g))
    assert float(wide[..., :coff].abs().sum()) == 0 and float(wide[..., coff + c:].abs().sum()) == 0
    back0 = torch.randn_like(x)
    back = sfhip.from_ncthw(back0)
    sfhip.channel_shuffle(out, back, c // g, accumulate=True)
    assert torch.equal(sfhip.to_ncthw(back), back0 + x)


@pytest.mark.parametrize("c,red,thw", [(240, 1, (4, 2, 2)), (144, 1, (2, 5, 7)), (256, 8, (2, 4, 4))])
def test_wide_head_spatial_attention(c, red, thw):
    """SpatialAttention heads wider than the flash kernels' 128 channels (SlowFastShuffleNet w2.0 / g3: C = 240 at
    s4_fuse) run on materialised scores (wdf_attention_helper.SpatialAttention._run_wide): module forward against an
    fp64 restatement of wdf_attention_helper.py:41-54, eval and taped training (all parameter and input gradients)."""
    import sfhip
    from slowfast.models import engine
    from slowfast.models.wdf_attention_helper import SpatialAttention
    dev = torch.device("cuda:0")
    torch.manual_seed(c)
    m = SpatialAttention(c, reduction=red).to(dev)
    with torch.no_grad():
        m.gamma.fill_(0.6)
        for cv in (m.query_conv, m.key_conv):
            cv.weight.mul_(0.5)
    b, (t, h, w) = 2, thw
    x = torch.randn(b, c, t, h, w, device=dev)
    up = torch.randn(b, c, t, h, w, device=dev)

    def ref(xd, params):
        wq, bq, wk, bk, wv, bv, gam = params
        n = t * h * w
        q = F.conv3d(xd, wq, bq).view(b, -1, n).permute(0, 2, 1)
        k = F.conv3d(xd, wk, bk).view(b, -1, n)
        v = F.conv3d(xd, wv, bv).view(b, -1, n)
        p = torch.softmax(torch.bmm(q, k), dim=-1)
        return gam * torch.bmm(v, p.permute(0, 2, 1)).view(b, c, t, h, w) + xd

    names = ("query_conv.weight", "query_conv.bias", "key_conv.weight", "key_conv.bias", "value_conv.weight",
             "value_conv.bias", "gamma")
    pd = dict(m.named_parameters())
    params = [pd[k].detach().double().requires_grad_(True) for k in names]
    xd = x.double().requires_grad_(True)
    yr = ref(xd, params)
    with torch.no_grad():
        y = m(x)
    assert _rel(y, yr.detach()) < 2e-5
    tape = engine.Tape()
    xa = sfhip.from_ncthw(x)
    with engine.taping(tape):
        z = m.run(xa)
    z.buf.copy_(up.permute(0, 2, 3, 4, 1))  # dL/dz arrives IN PLACE over z (the BN backward's convention)
    dxa = tape.grad_of(xa)
    with torch.no_grad():
        tape.backward()
    (yr * up.double()).sum().backward()
    assert _rel(sfhip.to_ncthw(dxa), xd.grad) < 2e-4
    gmax = max(float(pr.grad.abs().max()) for pr in params)
    for k, pr in zip(names, params):
        got = tape.pgrads[pd[k]].reshape(pr.shape)
        if k == "key_conv.bias":  # zero in exact arithmetic (q.b_k shifts a whole softmax row): rounding noise on both sides
            assert float(got.abs().max()) < 1e-5 * gmax, k
        else:
            assert _rel(got, pr.grad) < 2e-4, k


def test_grouped_weights_ride_in_the_batched_repack():
    """After an optimizer step engine.repack_all re-packs a grouped conv's G per-group packs as G records of its ONE
    launch, in place in the two tensors of the previous version (round 5 advice: it used to be G launches and two
    allocations per grouped conv per step): same bytes as sfhip.pack_grouped_weight_pair, same storage, one C-ABI call
    for two grouped convs and a dense one together."""
    import sfhip
    from slowfast.models import engine
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    model = nn.Sequential(nn.Conv3d(24, 36, (1, 3, 3), padding=(0, 1, 1), groups=3, bias=False),
                          nn.Conv3d(36, 40, 1, groups=4, bias=False), nn.Conv3d(40, 16, 1, bias=False)).to(dev)
    g1, g2, dense = model[0], model[1], model[2]
    before = [engine._group_pairs(g1), engine._group_pairs(g2), engine._packed_pair(dense.weight)]
    ptrs = [(a.data_ptr(), b.data_ptr()) for a, b in before]
    with torch.no_grad():
        for p in model.parameters():
            p.mul_(1.25).add_(0.01)  # what an optimizer step does: new values, same storage, new version
    engine.parameters_changed()
    calls = sfhip.CALLS
    engine.repack_all(model)
    assert sfhip.CALLS - calls == 1, "grouped + dense re-packs are one launch"
    calls = sfhip.CALLS
    after = [engine._group_pairs(g1), engine._group_pairs(g2), engine._packed_pair(dense.weight)]
    assert sfhip.CALLS == calls, "the cache entries are fresh after repack_all: nothing re-packs"
    assert [(a.data_ptr(), b.data_ptr()) for a, b in after] == ptrs, "in place"
    for conv, (wp, wtp) in ((g1, after[0]), (g2, after[1])):
        rp, rt = sfhip.pack_grouped_weight_pair(conv.weight, conv.groups)
        assert torch.equal(rp, wp) and torch.equal(rt, wtp)
    rp, rt = sfhip.pack_conv_weight_pair(dense.weight)
    assert torch.equal(rp, after[2][0]) and torch.equal(rt, after[2][1])
