"""GPU parity of the backward kernels (through the C ABI) against torch autograd on the CPU reference ops."""
import os
import zlib

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
TOL = 2e-4


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _rel(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def _act(x):
    import sfhip
    return sfhip.Act(x.detach().permute(0, 2, 3, 4, 1).contiguous().to(_dev()))


def _back(a):
    return a.buf[..., a.coff:a.coff + a.C].permute(0, 4, 1, 2, 3).contiguous().cpu()


def _report(name, err):
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "bwd_report.txt"), "a") as f:
            f.write("%-50s %.3e\n" % (name, err))
    except OSError:
        pass


CONV_CASES = [
    ("pw_64_256", 64, 256, (1, 1, 1), (1, 1, 1), (0, 0, 0), (2, 2, 12, 12)),
    ("pw_s2_288_512", 288, 512, (1, 1, 1), (1, 2, 2), (0, 0, 0), (1, 2, 14, 14)),
    ("t3_128_64", 128, 64, (3, 1, 1), (1, 1, 1), (1, 0, 0), (2, 4, 7, 7)),
    ("s3_64_64", 64, 64, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 2, 14, 14)),
    ("s3_s2_128_128", 128, 128, (1, 3, 3), (1, 2, 2), (0, 1, 1), (2, 2, 14, 14)),
    ("f2s_k7_32_64", 32, 64, (7, 1, 1), (4, 1, 1), (3, 0, 0), (1, 16, 6, 6)),
    ("pw_8_24", 8, 24, (1, 1, 1), (1, 1, 1), (0, 0, 0), (2, 4, 9, 9)),
    ("odd_27_16", 27, 16, (1, 1, 1), (1, 1, 1), (0, 0, 0), (2, 4, 5, 5)),
    ("fc_2304_400", 2304, 400, (1, 1, 1), (1, 1, 1), (0, 0, 0), (8, 1, 1, 1)),
    ("s5_k5_s2_48_32", 48, 32, (1, 5, 5), (1, 2, 2), (0, 2, 2), (2, 2, 13, 11)),
    ("t3_s2t_odd_24_40", 24, 40, (3, 3, 1), (2, 2, 1), (1, 1, 0), (1, 7, 9, 5)),
    ("head_320_1280_m32", 320, 1280, (1, 1, 1), (1, 1, 1), (0, 0, 0), (2, 4, 2, 2)),
    ("pw_96_16_m32", 96, 16, (1, 1, 1), (1, 1, 1), (0, 0, 0), (2, 4, 2, 2)),
    # small channel counts at >= 4096 positions: the data gradients run on conv_small.hip (mirrored taps)
    ("sm_8_8_s3", 8, 8, (1, 3, 3), (1, 1, 1), (0, 1, 1), (2, 4, 20, 28)),
    ("sm_16_16_s3", 16, 16, (1, 3, 3), (1, 1, 1), (0, 1, 1), (2, 3, 28, 28)),
    ("sm_32_8_t3", 32, 8, (3, 1, 1), (1, 1, 1), (1, 0, 0), (2, 6, 20, 20)),
    ("sm_128_32_t3", 128, 32, (3, 1, 1), (1, 1, 1), (1, 0, 0), (2, 4, 24, 24)),
    ("sm_8_32_pw", 8, 32, (1, 1, 1), (1, 1, 1), (0, 0, 0), (2, 4, 24, 25)),
    ("sm_32_128_pw", 32, 128, (1, 1, 1), (1, 1, 1), (0, 0, 0), (2, 4, 23, 24)),
]


@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv_dgrad_wgrad(case):
    import sfhip
    name, cin, cout, k, s, p, shp = case
    dev = _dev()
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()) % 10000)
    n, t, h, w = shp
    x = torch.randn(n, cin, t, h, w, generator=g, requires_grad=True)
    wt = (torch.randn(cout, cin, *k, generator=g) / np.sqrt(cin * k[0] * k[1] * k[2])).requires_grad_(True)
    y = F.conv3d(x, wt, None, s, p)
    dy = torch.randn(y.shape, generator=g)
    dx_ref, dw_ref = torch.autograd.grad(y, (x, wt), dy)
    xa, dza = _act(x), _act(dy)
    wtp = sfhip.pack_conv_weight(wt.detach().transpose(0, 1).contiguous().to(dev))
    base = torch.randn(n, t, h, w, cin, generator=g)
    dxa = sfhip.Act(base.clone().to(dev))
    sfhip.conv_dgrad(dza, wtp, xa, k, s, p, out=dxa, accumulate=True)
    # first-writer form into an uninitialised (NaN-filled) buffer: strided layers write every residue class (or zero
    # the buffer first when a class gets no tap at all), dense layers overwrite
    fresh = sfhip.Act(torch.full((n, t, h, w, cin), float("nan"), device=dev))
    sfhip.conv_dgrad(dza, wtp, xa, k, s, p, out=fresh, accumulate=False)
    dwp = sfhip.conv_wgrad(xa, dza, cout, k, s, p)
    torch.cuda.synchronize()
    assert _rel(_back(fresh), dx_ref) < TOL, name
    e1 = _rel(_back(dxa) - base.permute(0, 4, 1, 2, 3), dx_ref)
    e2 = _rel(sfhip.unpack_conv_weight_grad(dwp, wt.shape), dw_ref)
    _report("conv/%s dgrad" % name, e1)
    _report("conv/%s wgrad" % name, e2)
    assert e1 < TOL and e2 < TOL, (e1, e2)


@pytest.mark.parametrize("shape", [(2, 6, 32, 32), (3, 9, 20, 40), (1, 12, 18, 70)])
def test_stem_trick_wgrad(shape):
    """Weight gradient of the 5x7x7 Fast-pathway stem computed directly in the stem-trick layout (LDS-ring kernel
    conv_wgrad_stem_kernel: zero frames outside the clip in T, ragged last 16-position block, t range cut in parts)."""
    import sfhip
    dev = _dev()
    g = torch.Generator().manual_seed(4 + shape[0])
    kt, cout = 5, 8
    n, t, h, w = shape
    x = torch.randn(n, 3, t, h, w, generator=g)
    wt = (torch.randn(cout, 3, kt, 7, 7, generator=g) / np.sqrt(147 * kt)).requires_grad_(True)
    y = F.conv3d(x, wt, None, (1, 2, 2), (kt // 2, 3, 3))
    dy = torch.randn(y.shape, generator=g)
    (dw_ref,) = torch.autograd.grad(y, (wt,), dy)
    wp = (w + 6 + 1) // 2 * 2
    xa = sfhip.from_ncthw(x.to(dev), cpad=4, ph=3, pw=3, wp=wp)
    view = sfhip.Act(xa.buf.view(n, t, h + 6, wp // 2, 8))
    dwp = sfhip.conv_wgrad(view, _act(dy), cout, (kt, 7, 1), (1, 2, 1), (kt // 2, 0, 0), cin=28, cin_pad=32)
    again = sfhip.conv_wgrad(view, _act(dy), cout, (kt, 7, 1), (1, 2, 1), (kt // 2, 0, 0), cin=28, cin_pad=32)
    torch.cuda.synchronize()
    dw = dwp[:, :, :28].reshape(cout, kt, 7, 7, 4)[..., :3].permute(0, 4, 1, 2, 3)
    e = _rel(dw, dw_ref)
    _report("stem wgrad %s" % (shape,), e)
    assert e < TOL
    assert torch.equal(dwp, again), "bit-reproducible"
    assert float(dwp[:, :, 28:].abs().max()) == 0.0, "packed padding channels must come back zero"


@pytest.mark.parametrize("c,relu,use_res,rep,hw", [
    (64, True, True, 1, (6, 5)), (8, True, False, 1, (6, 5)), (27, False, False, 1, (6, 5)),
    (32, True, False, 4, (6, 5)), (300, True, True, 1, (6, 5)), (48, 6, False, 1, (6, 5)), (1280, 6, False, 1, (2, 2)),
    (512, True, True, 1, (14, 14)), (256, True, False, 1, (9, 7)),
    (18, 6, True, 1, (3, 3))])
def test_bn_backward(c, relu, use_res, rep, hw):
    import sfhip
    dev = _dev()
    g = torch.Generator().manual_seed(c + rep)
    shp = (2, c, 4) + hw
    z = (torch.randn(shp, generator=g) * 1.5 + 0.3).requires_grad_(True)
    gamma = (torch.rand(c, generator=g) + (2.5 if relu == 6 else 0.5)).requires_grad_(True)  # relu6: saturate some
    beta = (torch.randn(c, generator=g) * 0.1 + (2.0 if relu == 6 else 0.0)).requires_grad_(True)
    res = torch.randn(shp, generator=g).requires_grad_(True) if use_res else None
    y = F.batch_norm(z, None, None, gamma, beta, True, 0.0, 1e-5)
    if use_res:
        y = y + res
    if relu == 6:
        y = F.relu6(y)
    elif relu:
        y = F.relu(y)
    y = y.repeat_interleave(rep, dim=2)
    dy = torch.randn(y.shape, generator=g)
    outs = torch.autograd.grad(y, [z, gamma, beta] + ([res] if use_res else []), dy)
    za = _act(z)
    mean, var = sfhip.channel_stats(za)
    invstd = torch.rsqrt(var + 1e-5)
    dres = sfhip.Act(torch.zeros((2, 4) + hw + (c,), device=dev)) if use_res else None
    dz, dgamma, dbeta = sfhip.bn_bwd(_act(dy), _act(y), za, mean, invstd, gamma.detach().to(dev), relu, rep=rep,
                                     dres=dres)
    # the reduction is bit-reproducible (c >= 256: partial sums and final step in ONE launch, last-workgroup tickets)
    dres2 = sfhip.Act(torch.zeros((2, 4) + hw + (c,), device=dev)) if use_res else None
    dz2, dgamma2, dbeta2 = sfhip.bn_bwd(_act(dy), _act(y), _act(z), mean, invstd, gamma.detach().to(dev), relu, rep=rep,
                                        dres=dres2)
    torch.cuda.synchronize()
    assert torch.equal(dgamma, dgamma2) and torch.equal(dbeta, dbeta2) and torch.equal(dz.buf, dz2.buf)
    errs = [_rel(_back(dz), outs[0]), _rel(dgamma, outs[1]), _rel(dbeta, outs[2])]
    if use_res:
        errs.append(_rel(_back(dres), outs[3]))
    _report("bn_bwd c%d relu%d res%d rep%d hw%s" % (c, relu, use_res, rep, hw), max(errs))
    assert max(errs) < TOL, errs


@pytest.mark.parametrize("c,nsplit,n,relu,use_res,rep", [(16, 2, 4, True, True, 1), (7, 4, 8, False, False, 1),
                                                        (32, 2, 6, True, False, 4)])
def test_sub_batchnorm_forward_backward(c, nsplit, n, relu, use_res, rep):
    """SubBatchNorm3d training pass (batchnorm_helper.py:96-109): per-split statistics, shared affine, running
    statistics of the S*C split_bn, and the backward through the split statistics — vs torch autograd."""
    import sfhip
    from slowfast.models import engine
    from slowfast.models.batchnorm_helper import SubBatchNorm3d
    dev = _dev()
    g = torch.Generator().manual_seed(c * 7 + nsplit)
    x = (torch.randn(n, c, 2, 5, 3, generator=g) * 2.0 + 0.5).requires_grad_(True)
    res = torch.randn(n, c, 2, 5, 3, generator=g).requires_grad_(True) if use_res else None
    ref_bn = SubBatchNorm3d(nsplit, num_features=c, eps=1e-5, momentum=0.1)
    ref_bn.weight.data = torch.rand(c, generator=g) + 0.5
    ref_bn.bias.data = torch.randn(c, generator=g) * 0.2
    ref_bn.split_bn.running_mean.copy_(torch.randn(nsplit * c, generator=g))
    ref_bn.split_bn.running_var.copy_(torch.rand(nsplit * c, generator=g) + 0.5)
    mine = SubBatchNorm3d(nsplit, num_features=c, eps=1e-5, momentum=0.1).to(dev)
    mine.load_state_dict(ref_bn.state_dict())
    # torch reference: exactly the reference module's training forward
    y = ref_bn.split_bn(x.reshape(n // nsplit, c * nsplit, 2, 5, 3)).reshape(n, c, 2, 5, 3)
    y = y * ref_bn.weight.view(-1, 1, 1, 1) + ref_bn.bias.view(-1, 1, 1, 1)
    if use_res:
        y = y + res
    if relu:
        y = F.relu(y)
    y = y.repeat_interleave(rep, dim=2)
    dy = torch.randn(y.shape, generator=g)
    outs = torch.autograd.grad(y, [x, ref_bn.weight, ref_bn.bias] + ([res] if use_res else []), dy)
    t = engine.Tape()
    xa = _act(x)
    ra = _act(res) if use_res else None
    with torch.no_grad(), engine.taping(t):
        ya = engine.bn_train_apply(mine, xa, res=ra, relu=relu, rep=rep)
        yv = _back(ya)
        t.grad_of(ya).buf.copy_(dy.permute(0, 2, 3, 4, 1).to(dev))
        if use_res:
            dra = t.grad_of(ra)
        t.backward()
    torch.cuda.synchronize()
    errs = [_rel(yv, y), _rel(_back(xa), outs[0]), _rel(t.pgrads[mine.weight], outs[1]),
            _rel(t.pgrads[mine.bias], outs[2]),
            _rel(mine.split_bn.running_mean, ref_bn.split_bn.running_mean),
            _rel(mine.split_bn.running_var, ref_bn.split_bn.running_var)]
    if use_res:
        errs.append(_rel(_back(dra), outs[3]))
    _report("sub_bn c%d S%d n%d relu%d res%d rep%d" % (c, nsplit, n, relu, use_res, rep), max(errs))
    assert max(errs) < TOL, errs
    assert int(mine.split_bn.num_batches_tracked) == 1 and int(mine.bn.num_batches_tracked) == 0


@pytest.mark.parametrize("c", [12, 7])  # float4 kernel / scalar kernel
@pytest.mark.parametrize("k,s,p", [((1, 3, 3), (1, 2, 2), (0, 1, 1)), ((3, 3, 3), (1, 2, 2), (1, 1, 1))])
def test_maxpool_backward(k, s, p, c):
    import sfhip
    dev = _dev()
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, c, 4, 13, 12, generator=g).requires_grad_(True)
    y = F.max_pool3d(x, k, s, p)
    dy = torch.randn(y.shape, generator=g)
    (dx_ref,) = torch.autograd.grad(y, (x,), dy)
    dx = sfhip.Act(torch.zeros(2, 4, 13, 12, c, device=dev))
    sfhip.maxpool_bwd(_act(x), _act(y), _act(dy), dx, k, s, p)
    first = sfhip.Act(torch.full((2, 4, 13, 12, c), float("nan"), device=dev))  # first-writer form: no zero fill
    sfhip.maxpool_bwd(_act(x), _act(y), _act(dy), first, k, s, p, overwrite=True)
    torch.cuda.synchronize()
    assert _rel(_back(dx), dx_ref) < 1e-6
    assert torch.equal(dx.buf, first.buf)


@pytest.mark.parametrize("ties", [False, True])
@pytest.mark.parametrize("k,s,p", [((1, 3, 3), (1, 2, 2), (0, 1, 1)), ((3, 3, 3), (1, 2, 2), (1, 1, 1)),
                                   ((2, 2, 2), (2, 2, 2), (0, 0, 0)), ((4, 1, 1), (4, 1, 1), (0, 0, 0))])
def test_maxpool_winner_map_backward(k, s, p, ties):
    """sf_maxpool_fwd_arg / sf_maxpool_bwd_arg (stem pool1, the Nonlocal key pooling): the forward records the first
    maximum of every window, the backward gathers dL/dy through that map alone — equal to autograd through
    F.max_pool3d, also where windows overlap and where elements TIE (ReLU outputs: runs of zeros; ties = quantised
    inputs), in both the accumulating and the first-writer form, on channel-slice views."""
    import sfhip
    dev = _dev()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 8, 4, 13, 12, generator=g)
    if ties:
        x = torch.relu((x * 2).round() / 2)
    x.requires_grad_(True)
    y = F.max_pool3d(x, k, s, p)
    dy = torch.randn(y.shape, generator=g)
    (dx_ref,) = torch.autograd.grad(y, (x,), dy)
    xa = sfhip.Act(torch.randn(2, 4, 13, 12, 20, device=dev), 8, 8)  # a channel slice of a wider buffer
    xa.buf[..., 8:16] = x.detach().permute(0, 2, 3, 4, 1).to(dev)
    ya, arg = sfhip.pool(xa, k, s, p, want_arg=True, out_reserve=(4, 0))
    assert arg is not None and arg.dtype == torch.uint8
    assert torch.equal(_back(ya), y.detach())
    dya = sfhip.Act(torch.zeros(ya.buf.shape, device=dev), ya.coff, ya.C)
    dya.buf[..., ya.coff:ya.coff + 8] = dy.permute(0, 2, 3, 4, 1).to(dev)
    dx = sfhip.Act(torch.ones(2, 4, 13, 12, 8, device=dev))
    assert sfhip.maxpool_bwd_arg(xa, arg, dya, dx, k, s, p)
    first = sfhip.Act(torch.full((2, 4, 13, 12, 8), float("nan"), device=dev))
    assert sfhip.maxpool_bwd_arg(xa, arg, dya, first, k, s, p, overwrite=True)
    torch.cuda.synchronize()
    assert _rel(_back(first), dx_ref) < 1e-6
    assert torch.equal(dx.buf - 1.0, first.buf) or _rel(_back(dx) - 1.0, dx_ref) < 1e-6
    # the search form (no map) agrees bit for bit
    ref2 = sfhip.Act(torch.zeros(2, 4, 13, 12, 8, device=dev))
    sfhip.maxpool_bwd(xa, ya, dya, ref2, k, s, p)
    assert torch.equal(ref2.buf, first.buf)


@pytest.mark.parametrize("c,reduction", [(32, 8), (32, 1), (64, 4)])
def test_spatial_attention_module_with_reduction(c, reduction):
    """SpatialAttention(channel, reduction) as the reference class builds it (wdf_attention_helper.py:17-31: q / k with
    channel // reduction outputs, default reduction 8): forward and every gradient (x, q/k/v weights and biases, gamma)
    of the HIP module against autograd through the oracle.  r > 1 runs on zero-padded q / k rows of the merged
    projection (models/wdf_attention_helper.py::SpatialAttention.qkv)."""
    import sfhip
    from oracle import slowfast_oracle as oracle
    from slowfast.models import engine
    from slowfast.models.wdf_attention_helper import SpatialAttention
    dev = _dev()
    g = torch.Generator().manual_seed(c + reduction)
    x = torch.randn(2, c, 2, 6, 7, generator=g).requires_grad_(True)
    m = SpatialAttention(c, reduction=reduction)
    with torch.no_grad():
        for nm, cv in (("query_conv", m.query_conv), ("key_conv", m.key_conv), ("value_conv", m.value_conv)):
            cv.weight.copy_(torch.randn(cv.weight.shape, generator=g) * (0.7 / np.sqrt(c)))
            cv.bias.copy_(torch.randn(cv.bias.shape, generator=g) * 0.1)
        m.gamma.fill_(0.6)
    sd = {"m." + k: v.detach().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    ref = oracle.spatial_attention(sd, "m", x)
    dy = torch.randn(ref.shape, generator=g)
    names = sorted(sd)
    grads = torch.autograd.grad(ref, [x] + [sd[k] for k in names], dy)
    m = m.to(dev)
    t = engine.Tape()
    with torch.no_grad(), engine.taping(t):
        xa = sfhip.from_ncthw(x.detach().to(dev))
        z = m.run(xa)
        torch.cuda.synchronize()
        e_fwd = _rel(_back(z), ref)
        # the taped attention reads dL/dz from z's own buffer (where the BN backward leaves it in a model)
        z.buf[..., z.coff:z.coff + c] = dy.permute(0, 2, 3, 4, 1).to(dev)
        for fn, side in reversed(t.ops):
            fn()
    torch.cuda.synchronize()
    e_dx = _rel(_back(t.grad_of(xa)), grads[0])
    errs = {"fwd": e_fwd, "dx": e_dx}
    for k, gref in zip(names, grads[1:]):
        got = t.pgrads[getattr(m, k.split(".")[1]) if k.count(".") == 1 else getattr(getattr(m, k.split(".")[1]), k.split(".")[2])]
        if k.endswith("key_conv.bias"):  # shift-invariance of the softmax: analytically zero, both sides hold noise
            assert float(got.abs().max()) < 1e-3 * float(grads[1 + names.index(k.replace("bias", "weight"))].abs().max())
            continue
        errs[k] = _rel(got.reshape(gref.shape), gref)
    _report("SpatialAttention c%d r%d" % (c, reduction), max(errs.values()))
    assert max(errs.values()) < TOL, errs


@pytest.mark.parametrize("c,alpha", [(8, 4), (32, 4), (3, 8)])
def test_eca_backward(c, alpha):
    """d/dx and d/dw3 of  max-pool_alpha -> ECA gate  against autograd of the oracle's functions."""
    import sfhip
    from oracle import slowfast_oracle as oracle
    dev = _dev()
    g = torch.Generator().manual_seed(c)
    x = torch.randn(2, c, 8, 5, 7, generator=g).requires_grad_(True)
    w3 = torch.randn(1, 1, 3, generator=g).requires_grad_(True)
    mx = F.max_pool3d(x, (alpha, 1, 1), (alpha, 1, 1))
    y = oracle.eca({"m.conv.weight": w3}, "m", mx)
    dy = torch.randn(y.shape, generator=g)
    dx_ref, dw_ref = torch.autograd.grad(y, (x, w3), dy)
    # HIP path: reductions on the GPU and the [N, C]-sized gate algebra in ONE small HIP launch (sf_eca_gate_bwd)
    xa, dza = _act(x), _act(dy)
    pooled = sfhip.tmax_mean(xa, alpha)
    dg = sfhip.tmax_dot(xa, alpha, dza)
    w = w3.detach().to(dev)
    count = (8 // alpha) * 5 * 7
    dw = torch.full((3,), 0.5, device=dev)  # accumulated in place on top of what is there
    gate, dpool = sfhip.eca_gate_bwd(dg, pooled, w.contiguous(), 1.0 / count, dw)
    assert _rel(gate, torch.sigmoid(F.conv1d(pooled.unsqueeze(1), w, None, 1, 1).squeeze(1))) < 1e-6
    dx = sfhip.Act(torch.zeros(2, 8, 5, 7, c, device=dev))
    sfhip.eca_bwd_apply(xa, alpha, dza, gate, dpool, dx)
    torch.cuda.synchronize()
    e1, e2 = _rel(_back(dx), dx_ref), _rel((dw - 0.5).view(1, 1, 3), dw_ref)
    _report("eca_bwd c%d a%d" % (c, alpha), max(e1, e2))
    assert e1 < TOL and e2 < TOL, (e1, e2)


def test_small_helpers():
    import sfhip
    dev = _dev()
    a = torch.randn(1, 2, 3, 4, 10)
    b = torch.randn(1, 2, 3, 4, 10)
    d = sfhip.rowdot(sfhip.Act(a.to(dev)), sfhip.Act(b.to(dev)), 0.5)
    gbuf = sfhip.Act(torch.ones(2, 2, 3, 4, 6, device=dev))
    v = torch.randn(2, 6)
    sfhip.bcast_add(gbuf, v.to(dev), 0.25)
    o = sfhip.Act(torch.ones(1, 2, 3, 4, 10, device=dev))
    sfhip.axpy(sfhip.Act(a.to(dev)), o, alpha=2.0, accumulate=True)
    torch.cuda.synchronize()
    assert _rel(d.view(2, 3, 4), (a * b).sum(-1)[0] * 0.5) < 1e-5
    assert _rel(gbuf.buf, 1 + 0.25 * v.view(2, 1, 1, 1, 6).expand(2, 2, 3, 4, 6)) < 1e-6
    assert _rel(o.buf, 1 + 2 * a) < 1e-6


ATTN_BWD = [(8, (2, 12, 12)), (32, (2, 14, 14)), (64, (4, 7, 7)), (128, (2, 7, 7)), (3, (4, 8, 8)), (28, (2, 9, 9)),
            (32, (1, 3, 3)),
            # d <= 8 off the packed-bf16 path: the 4x4x1-MFMA fused kernel (cfg #5's d = 4 / 6), ragged N, one quad and two
            (4, (2, 9, 9)), (6, (3, 7, 9)), (7, (2, 8, 8)), (5, (1, 5, 13)), (4, (1, 4, 4))]


@pytest.mark.parametrize("c,thw", ATTN_BWD, ids=["c%d_n%d" % (c, t * h * w) for c, (t, h, w) in ATTN_BWD])
def test_attention_backward(c, thw):
    """dq, dk, dv, dgamma of the flash attention vs autograd through the oracle's dense softmax attention."""
    import sfhip
    dev = _dev()
    g = torch.Generator().manual_seed(c * 7 + thw[1])
    t, h, w = thw
    B, n = 2, t * h * w
    q = (torch.randn(B, n, c, generator=g) * 0.7).requires_grad_(True)
    k = (torch.randn(B, n, c, generator=g) * 0.7).requires_grad_(True)
    v = torch.randn(B, n, c, generator=g).requires_grad_(True)
    x = torch.randn(B, n, c, generator=g)
    gamma = torch.tensor([0.6], requires_grad=True)
    z = gamma * (torch.softmax(q @ k.transpose(1, 2), -1) @ v) + x
    dz = torch.randn(z.shape, generator=g)
    dq_r, dk_r, dv_r, dg_r = torch.autograd.grad(z, (q, k, v, gamma), dz)
    qkv = sfhip.Act(torch.cat([q, k, v], -1).detach().view(B, t, h, w, 3 * c).contiguous().to(dev))
    xa = sfhip.Act(x.view(B, t, h, w, c).contiguous().to(dev))
    gam = gamma.detach().to(dev)
    save = {}
    out = sfhip.attention(qkv.slice(0, c), qkv.slice(c, c), qkv.slice(2 * c, c), xa, gam, save=save)
    dqkv = sfhip.Act(torch.zeros(B, t, h, w, 3 * c, device=dev))
    dza = sfhip.Act(dz.view(B, t, h, w, c).contiguous().to(dev))
    dvec = sfhip.attention_bwd(qkv.slice(0, c), qkv.slice(c, c), qkv.slice(2 * c, c), dza, save["o"], save["lse"],
                               gam, dqkv.slice(0, c), dqkv.slice(c, c), dqkv.slice(2 * c, c))
    torch.cuda.synchronize()
    assert _rel(out.buf.view(B, n, c), z) < TOL
    got = dqkv.buf.view(B, n, 3 * c)
    errs = [_rel(got[..., :c], dq_r), _rel(got[..., c:2 * c], dk_r), _rel(got[..., 2 * c:], dv_r),
            _rel(dvec.sum().view(1), dg_r)]
    _report("attn_bwd c%d n%d" % (c, n), max(errs))
    assert max(errs) < TOL, errs


@pytest.mark.parametrize("parts", [3, 8])
def test_attention_backward_query_parts(parts):
    """The fused backward cuts the query sweep into parts to fill the chip's tail (sf_sweep_parts); the shapes above
    are too small to be cut, so force the cut (SF_SWEEP_PARTS is read once per process -> child process).  8 parts of
    a 5..7-tile sweep also leave parts EMPTY, which must contribute zero dK / dV."""
    import subprocess
    import sys
    if os.environ.get("SF_SWEEP_PARTS"):
        pytest.skip("already inside the forced-parts child")
    env = dict(os.environ, SF_SWEEP_PARTS=str(parts))
    if parts == 3:  # also take the 16x16x4 forward kernel's d = 32 instantiation (off by default) through the suite
        env["SF_ATTN_FWD32"] = "small"
    else:           # ... and the 8-wavefront (128 keys per workgroup) form of the d <= 4 backward, which only large
        env["SF_ATTN_SMALL_NW"] = "8"   # problems take by themselves
    fwd = os.path.join(os.path.dirname(__file__), "test_ops_gpu.py")  # its forward tests cut the KEY sweep alike
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", __file__, fwd, "-k",
                        "(test_attention_backward and not query_parts) or test_attention"], env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "passed" in r.stdout


@pytest.mark.parametrize("c,relu,use_res,wide", [(64, True, True, False), (8, True, False, True), (48, 6, False, False),
                                                 (256, 6, True, True)])
def test_bn_backward_byte_mask(c, relu, use_res, wide):
    """The normalise pass can leave a byte per 4 channels (which of them pass a gradient through ReLU / ReLU6) and
    the BN backward reads that instead of the activation: bit-identical dz / dres / dgamma / dbeta to the
    activation-reading form, for dense and channel-slice views."""
    import sfhip
    dev = _dev()
    g = torch.Generator().manual_seed(c + (7 if use_res else 0))
    shp = (2, 3, 5, 6, c)
    z = torch.randn(shp, generator=g) * 1.5 + 0.3
    gamma = (torch.rand(c, generator=g) + (2.5 if relu == 6 else 0.5)).to(dev)
    beta = (torch.randn(c, generator=g) * 0.1 + (2.0 if relu == 6 else 0.0)).to(dev)
    res = sfhip.Act(torch.randn(shp, generator=g).to(dev)) if use_res else None
    dy = sfhip.Act(torch.randn(shp, generator=g).to(dev))

    def run(with_mask):
        za = sfhip.Act(z.to(dev).clone())
        mean, invstd, scale, shift = sfhip.bn_train_stats(za, gamma, beta, 1e-5, 0.1, None, None)
        if wide:  # the activation lives in a channel slice of a wider buffer
            out = sfhip.Act(torch.zeros(shp[:4] + (c + 8,), device=dev)).slice(4, c)
        else:
            out = None
        mk = {} if with_mask else None
        y = sfhip.affine(za, scale, shift, res=res, relu=relu, out=out, mask=mk)
        if with_mask:
            assert "bytes" in mk and mk["bytes"].dtype == torch.uint8 and mk["bytes"].numel() == za.rows * (c // 4)
        dres = sfhip.Act(torch.ones(shp, device=dev)) if use_res else None
        dz, dg, db = sfhip.bn_bwd(dy, y, za, mean, invstd, gamma, relu, dres=dres,
                                  mask=mk["bytes"] if with_mask else None)
        torch.cuda.synchronize()
        return y.buf.clone(), dz.buf.clone(), dg.clone(), db.clone(), (dres.buf.clone() if use_res else None)

    a, b = run(True), run(False)
    for u, v in zip(a, b):
        if u is not None:
            assert torch.equal(u, v)
    passed = float((a[0][..., -c:] > 0).float().mean())
    assert 0.05 < passed < 0.98, passed  # the mask is neither all ones nor all zeros


@pytest.mark.parametrize("c", [64, 7])
def test_bn_backward_first_writer_of_residual_gradient(c):
    """sf_bn_bwd_apply_first writes the residual branch's gradient (dres = g) where sf_bn_bwd_apply accumulates: an
    uninitialised (here NaN-filled) buffer comes back equal to the accumulate-into-zeros result, bit for bit."""
    import sfhip
    dev = _dev()
    g = torch.Generator().manual_seed(c)
    shp = (2, 3, 5, 6, c)
    z = torch.randn(shp, generator=g) * 1.5 + 0.3
    gamma, beta = (torch.rand(c, generator=g) + 0.5).to(dev), torch.randn(c, generator=g).to(dev)
    res = sfhip.Act(torch.randn(shp, generator=g).to(dev))
    dy = sfhip.Act(torch.randn(shp, generator=g).to(dev))
    outs = []
    for first in (False, True):
        za = sfhip.Act(z.to(dev).clone())
        mean, invstd, scale, shift = sfhip.bn_train_stats(za, gamma, beta, 1e-5, 0.1, None, None)
        y = sfhip.affine(za, scale, shift, res=res, relu=True)
        dres = sfhip.Act(torch.full(shp, float("nan"), device=dev) if first else torch.zeros(shp, device=dev))
        dz, dg, db = sfhip.bn_bwd(dy, y, za, mean, invstd, gamma, True, dres=dres, dres_overwrite=first)
        torch.cuda.synchronize()
        outs.append((dz.buf.clone(), dres.buf.clone(), dg.clone(), db.clone()))
    for u, v in zip(*outs):
        assert torch.equal(u, v)
    assert bool(torch.isfinite(outs[1][1]).all())


@pytest.mark.parametrize("c,k,s", [(32, (3, 3, 3), (1, 1, 1)), (12, (1, 5, 5), (1, 2, 2)), (27, (3, 3, 3), (1, 2, 2))])
def test_dwconv_backward(c, k, s):
    import sfhip
    dev = _dev()
    g = torch.Generator().manual_seed(c)
    x = torch.randn(2, c, 4, 10, 10, generator=g).requires_grad_(True)
    wt = (torch.randn(c, 1, *k, generator=g) / np.sqrt(k[0] * k[1] * k[2])).requires_grad_(True)
    p = tuple(kk // 2 for kk in k)
    y = F.conv3d(x, wt, None, s, p, 1, c)
    dy = torch.randn(y.shape, generator=g)
    dx_ref, dw_ref = torch.autograd.grad(y, (x, wt), dy)
    dx = sfhip.Act(torch.zeros(2, 4, 10, 10, c, device=dev))
    dw = sfhip.dwconv_bwd(_act(x), _act(dy), sfhip.pack_dw_weight(wt.detach().to(dev)), k, s, p, dx=dx)
    torch.cuda.synchronize()
    e1 = _rel(_back(dx), dx_ref)
    e2 = _rel(dw.t().reshape(wt.shape), dw_ref)
    _report("dwconv_bwd c%d" % c, max(e1, e2))
    assert e1 < TOL and e2 < TOL, (e1, e2)


@pytest.mark.parametrize("relu", [True, 6])
def test_bare_activation_backward(relu):
    import sfhip
    dev = _dev()
    g = torch.Generator().manual_seed(11)
    y = torch.randn(2, 10, 3, 5, 4, generator=g) * 4.0
    y = F.relu6(y) if relu == 6 else F.relu(y)
    dy = torch.randn(y.shape, generator=g)
    base = torch.randn(2, 3, 5, 4, 10, generator=g)
    ref = dy * ((y > 0) & ((y < 6) if relu == 6 else (y == y))).float()
    dx = sfhip.Act(base.clone().to(dev))
    sfhip.act_bwd(_act(dy), _act(y), relu, dx, accumulate=True)
    dx2 = sfhip.Act(torch.full((2, 3, 5, 4, 10), 7.0, device=dev))
    sfhip.act_bwd(_act(dy), _act(y), relu, dx2, accumulate=False)
    torch.cuda.synchronize()
    assert torch.equal(_back(dx2), ref)
    assert _rel(_back(dx) - base.permute(0, 4, 1, 2, 3), ref) < 1e-6


def test_avgpool_shortcut_forward_backward():
    """ShuffleNet v1 shortcut: conv1x1 -> AvgPool3d((1,3,3),(1,2,2),(0,1,1)) -> ReLU evaluated as pool -> conv
    -> ReLU through the taped engine ops (shufflenet_helper.py:61-68)."""
    import sfhip
    from slowfast.models import engine
    dev = _dev()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 27, 3, 9, 7, generator=g, requires_grad=True)
    conv = torch.nn.Conv3d(27, 20, 1, bias=False)
    ref = F.relu(F.avg_pool3d(conv(x), (1, 3, 3), (1, 2, 2), (0, 1, 1)))
    dy = torch.randn(ref.shape, generator=g)
    dx_ref, dw_ref = torch.autograd.grad(ref, (x, conv.weight), dy)
    conv_d = torch.nn.Conv3d(27, 20, 1, bias=False).to(dev)
    conv_d.weight.data.copy_(conv.weight.data)
    t = engine.Tape()
    xa = _act(x)
    with torch.no_grad(), engine.taping(t):
        pooled = engine.avgpool(xa, (1, 3, 3), (1, 2, 2), (0, 1, 1))
        y = engine.conv_bn_act(pooled, conv_d, None, relu=True)
        t.grad_of(y).buf.copy_(dy.permute(0, 2, 3, 4, 1).to(dev))
        dxa = t.grad_of(xa)
        t.backward()
    torch.cuda.synchronize()
    e = [_rel(_back(y), ref), _rel(_back(dxa), dx_ref), _rel(t.pgrads[conv_d.weight], dw_ref)]
    _report("avgpool shortcut fwd/dx/dw", max(e))
    assert max(e) < TOL, e


@pytest.mark.parametrize("inst,pool,dim,thw", [("softmax", (1, 2, 2), 64, (2, 6, 6)), ("dot_product", (1, 2, 2), 32, (3, 4, 4)),
                                                ("softmax", None, 32, (2, 3, 5)), ("dot_product", (2, 2, 2), 96, (4, 6, 6))])
def test_nonlocal_block_forward_backward(inst, pool, dim, thw):
    """Nonlocal (nonlocal_helper.py:105-148), train mode, against the oracle's restatement under fp64 autograd:
    output, input gradient and every parameter gradient."""
    from oracle import slowfast_oracle as oracle
    from slowfast.models import engine
    from slowfast.models.nonlocal_helper import Nonlocal
    dev = _dev()
    torch.manual_seed(dim + len(inst))
    blk = Nonlocal(dim, dim // 2, pool, instantiation=inst).to(dev).train()
    with torch.no_grad():
        for k, v in blk.named_parameters():
            v.copy_(torch.randn_like(v) * (0.3 if v.dim() > 1 else 0.2) + (1.0 if k == "bn.weight" else 0.0))
    x = torch.randn((2, dim) + thw)
    dy = torch.randn((2, dim) + thw)
    sd = {"m." + k: v.detach().double().cpu().requires_grad_(v.dtype.is_floating_point and "running" not in k)
          for k, v in blk.state_dict().items() if "num_batches" not in k}
    xr = x.double().requires_grad_(True)
    ref = oracle.nonlocal_block(sd, "m", xr, pool, inst, True)
    ref.backward(dy.double())
    t = engine.Tape()
    xa = _act(x)
    with torch.no_grad(), engine.taping(t):
        ya = blk.run(xa)
        out = _back(ya)
        t.grad_of(ya).buf.copy_(dy.permute(0, 2, 3, 4, 1).to(dev))
        dxa = t.grad_of(xa)
        t.backward()
    torch.cuda.synchronize()
    errs = {"y": _rel(out, ref), "dx": _rel(_back(dxa), xr.grad)}
    for k, v in blk.named_parameters():
        a, b = t.pgrads[v].double().cpu(), sd["m." + k].grad
        # biases whose true gradient is exactly 0 (a constant shift removed by the train-mode BN / the softmax's
        # shift invariance) only carry cancellation noise: measure them ...
        scale = b.abs().max()
        if k.endswith(".bias"):  # ... relative to the same layer's weight gradient
            scale = torch.maximum(scale, sd["m." + k[:-4] + "weight"].grad.abs().max())
        errs[k] = float((a.reshape(b.shape) - b).abs().max() / scale.clamp_min(1e-30))
    _report("nonlocal %s pool=%s dim=%d" % (inst, pool, dim), max(errs.values()))
    assert max(errs.values()) < TOL, errs


def test_gather_add_is_the_adjoint_of_the_shuffled_store():
    import sfhip
    dev = _dev()
    x = torch.randn(1, 2, 3, 3, 10)
    wide = sfhip.Act(torch.zeros(1, 2, 3, 3, 24, device=dev))
    sfhip.copy_channels(sfhip.Act(x.to(dev)), sfhip.Act(wide.buf, 3, 21), out_cmul=2)
    back = sfhip.Act(torch.ones(1, 2, 3, 3, 10, device=dev))
    sfhip.gather_add(sfhip.Act(wide.buf, 3, 21), 2, back, accumulate=True)
    torch.cuda.synchronize()
    assert torch.equal(back.buf.cpu(), x + 1)


@pytest.mark.gpu
def test_batched_weight_repack_matches_per_weight_pack():
    """engine.repack_all's single launch (sf_pack_conv_weights over a device table) == sf_pack_conv_weight per weight,
    bit for bit, incl. in-place reuse of the previous packed buffers and odd channel counts (zero padding)."""
    import sfhip
    g = torch.Generator().manual_seed(5)
    shapes = [(64, 24, 1, 3, 3), (8, 32, 3, 1, 1), (256, 64, 1, 1, 1), (20, 12, 5, 7, 1), (512, 128, 3, 1, 1)]
    ws = [torch.randn(*s, generator=g).to(_dev()) for s in shapes]
    ref = [sfhip.pack_conv_weight_pair(w) for w in ws]
    got = sfhip.pack_conv_weight_pairs(ws, [None] * len(ws))
    for (rp, rt), (gp, gt) in zip(ref, got):
        assert torch.equal(rp, gp) and torch.equal(rt, gt)
    ws2 = [w * 1.5 + 0.25 for w in ws]  # new values, same buffers: the cached pointer table is keyed on the pointers
    for w, w2 in zip(ws, ws2):
        w.copy_(w2)
    again = sfhip.pack_conv_weight_pairs(ws, got)
    for w, (gp, gt), (ap, at) in zip(ws, got, again):
        assert ap.data_ptr() == gp.data_ptr() and at.data_ptr() == gt.data_ptr()
        rp, rt = sfhip.pack_conv_weight_pair(w)
        assert torch.equal(rp, ap) and torch.equal(rt, at)
