"""Forced-configuration sweep of the dense-conv kernels: the planner of conv_wave.hip picks a
(TM, TN, KS) tile, the persistent swapped-operand form, conv_small.hip or — for the weight gradient — a block shape and
a position-split count FROM THE BATCH SIZE, so the picks the benchmark makes at 8 clips are not the picks the small
parity cases make.  Here every configuration is forced with sf_conv_tune on shapes each of them covers and compared
with an fp64 convolution of the same inputs (forward with bias, data gradient = transposed descriptor, conv-epilogue
BN statistics, weight gradient), at the per-op tolerance of tests/test_ops_gpu.py.

Reference ops being matched: nn.Conv3d of resnet_helper.py:182-223, 326-335 and its autograd gradients."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
TOL = 2e-4

# name, Cin, Cout, kernel, pad, (N, T, H, W): ragged row counts (not multiples of 16 / 112 / 196) on purpose
SHAPES = [
    ("plain_64_256", 64, 256, (1, 1, 1), (0, 0, 0), (3, 2, 23, 21)),        # 4 K steps: the persistent form's home
    ("s3_128_128", 128, 128, (1, 3, 3), (0, 1, 1), (3, 2, 19, 17)),         # K = 1152, border taps
    ("t3_256_128", 256, 128, (3, 1, 1), (1, 0, 0), (2, 5, 13, 11)),         # K = 768, temporal borders
    ("plain_320_64", 320, 64, (1, 1, 1), (0, 0, 0), (2, 3, 17, 19)),        # 20 K steps, narrow output
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
    for knob, v in ((0, 1), (1, -1), (2, 0), (3, 0), (4, 1), (5, 0), (6, 1), (7, 1), (8, 0), (9, 1), (10, 1), (11, -1),
                    (12, 0), (20, 1), (21, 1), (22, 1), (23, -1)):
        L.sf_conv_tune(knob, v)


def _case(shape, seed=0):
    name, cin, cout, k, p, (n, t, h, w) = shape
    dev = _dev()
    g = torch.Generator().manual_seed(seed + cin * 7 + cout)
    x = torch.randn(n, cin, t, h, w, generator=g).to(dev)
    wt = (torch.randn(cout, cin, *k, generator=g) / np.sqrt(cin * k[0] * k[1] * k[2])).to(dev)
    bias = (torch.randn(cout, generator=g) * 0.1).to(dev)
    dy = torch.randn(n, cout, t, h, w, generator=g).to(dev)
    return x, wt, bias, dy


def _rel(a, b):
    return float((a.double() - b).abs().max() / b.abs().max())


def _act(x):
    import sfhip
    return sfhip.Act(x.permute(0, 2, 3, 4, 1).contiguous())


def _ncthw(a):
    return a.buf[..., a.coff:a.coff + a.C].permute(0, 4, 1, 2, 3)


# cfg: index into conv_wave.hip's CFGS = (13x2, 7x4, 7x2, 13x1) x (KS 4, KS 1); persist: sf_conv_tune(4, .)
# (0 = one-pass kernels only, 1 = default level, 11 = the persistent form for every KS == 1 layer it covers)
@pytest.mark.parametrize("persist", [0, 1, 11])
@pytest.mark.parametrize("cfg", list(range(8)))
@pytest.mark.parametrize("shape", SHAPES, ids=[s[0] for s in SHAPES])
def test_forced_conv_wave_configuration(shape, cfg, persist, tune):
    import sfhip
    name, cin, cout, k, p, dims = shape
    if persist and cfg % 2 == 0:
        pytest.skip("KS = 4 configurations have no persistent form")
    x, wt, bias, dy = _case(shape)
    assert tune.sf_conv_tune(1, cfg) == 0 and tune.sf_conv_tune(4, persist) == 0
    tune.sf_conv_tune(6, 0)  # conv_small and conv_bx have their own tests
    tune.sf_conv_tune(7, 0)
    xa = _act(x)
    wp, wtp = sfhip.pack_conv_weight_pair(wt)
    # forward with the BN statistics taken in the epilogue
    z, st = sfhip.conv(xa, wp, k, (1, 1, 1), p, bias=bias, stats=True)
    ref = F.conv3d(x.double(), wt.double(), bias.double(), 1, p)
    assert _rel(_ncthw(z), ref) < TOL, (name, cfg, persist)
    if st is not None:
        g = torch.rand(cout, device=x.device) + 0.5
        b = torch.randn(cout, device=x.device)
        mean, invstd, scale, shift = sfhip.bn_train_stats_merge(st, cout, g, b, 1e-5, 0.1, None, None)
        flat = ref.transpose(0, 1).reshape(cout, -1)
        rm, rv = flat.mean(1), flat.var(1, unbiased=False)
        assert float((mean.double() - rm).abs().max()) < 1e-5 * float(rm.abs().max() + flat.std())
        assert _rel(invstd, torch.rsqrt(rv + 1e-5)) < 1e-4
    # epilogue variants: scale + residual + ReLU (the eval-mode form)
    res = torch.randn_like(ref, dtype=torch.float32)
    sc = torch.rand(cout, device=x.device) + 0.5
    y = sfhip.conv(xa, wp, k, (1, 1, 1), p, scale=sc, bias=bias, relu=True, res=_act(res))
    ref2 = F.relu(F.conv3d(x.double(), wt.double(), None, 1, p) * sc.double().view(1, -1, 1, 1, 1) +
                  bias.double().view(1, -1, 1, 1, 1) + res.double())
    assert _rel(_ncthw(y), ref2) < TOL, (name, cfg, persist)
    # data gradient: written, then accumulated on top of itself
    xd = x.double().requires_grad_(True)
    F.conv3d(xd, wt.double(), None, 1, p).backward(dy.double())
    dxa = sfhip.conv_dgrad(_act(dy), wtp, xa, k, (1, 1, 1), p)
    assert _rel(_ncthw(dxa), xd.grad) < TOL, (name, cfg, persist)
    sfhip.conv_dgrad(_act(dy), wtp, xa, k, (1, 1, 1), p, out=dxa, accumulate=True)
    assert _rel(_ncthw(dxa), 2 * xd.grad) < TOL, (name, cfg, persist)
    torch.cuda.synchronize()


@pytest.mark.parametrize("rows", [16, 48, 100, 112])
def test_forced_rows_per_tile(rows, tune):
    """sf_conv_tune(2, rows): tiles with a runtime row count below TM * 16 (the planner picks 196 / 112 / ... rows
    from M = 49 * 2^k; other batch sizes give other remainders)."""
    import sfhip
    shape = SHAPES[1]
    name, cin, cout, k, p, dims = shape
    x, wt, bias, dy = _case(shape, seed=rows)
    tune.sf_conv_tune(6, 0)
    tune.sf_conv_tune(7, 0)
    wp, _ = sfhip.pack_conv_weight_pair(wt)
    ref = F.conv3d(x.double(), wt.double(), bias.double(), 1, p)
    for cfg in (0, 3, 5, 6):
        tune.sf_conv_tune(1, cfg)
        tune.sf_conv_tune(2, rows)
        z = sfhip.conv(_act(x), wp, k, (1, 1, 1), p, bias=bias)
        assert _rel(_ncthw(z), ref) < TOL, (cfg, rows)


SMALL = [
    ("sm_8_8_s3", 8, 8, (1, 3, 3), (0, 1, 1), (2, 4, 24, 28)),
    ("sm_16_32_s3", 16, 32, (1, 3, 3), (0, 1, 1), (2, 3, 28, 28)),
    ("sm_32_8_t3", 32, 8, (3, 1, 1), (1, 0, 0), (2, 6, 20, 20)),
    ("sm_128_32_t3", 128, 32, (3, 1, 1), (1, 0, 0), (2, 4, 24, 24)),
    ("sm_16_64_pw", 16, 64, (1, 1, 1), (0, 0, 0), (2, 4, 24, 24)),
]


@pytest.mark.parametrize("shape", SMALL, ids=[s[0] for s in SMALL])
def test_conv_small_every_shape_it_covers(shape, tune):
    """conv_small.hip forced onto every shape its instantiations cover (sf_conv_tune(6, 3) = SF_CONV_SMALL=2): forward
    with epilogue statistics and the data gradient (flipped taps) against fp64."""
    import sfhip
    name, cin, cout, k, p, dims = shape
    x, wt, bias, dy = _case(shape)
    assert tune.sf_conv_tune(6, 3) == 0
    wp, wtp = sfhip.pack_conv_weight_pair(wt)
    z, st = sfhip.conv(_act(x), wp, k, (1, 1, 1), p, bias=bias, stats=True)
    ref = F.conv3d(x.double(), wt.double(), bias.double(), 1, p)
    assert _rel(_ncthw(z), ref) < TOL, name
    if st is not None:
        ones = torch.ones(cout, device=x.device)
        mean, invstd, _, _ = sfhip.bn_train_stats_merge(st, cout, ones, ones, 1e-5, 0.1, None, None)
        flat = ref.transpose(0, 1).reshape(cout, -1)
        assert float((mean.double() - flat.mean(1)).abs().max()) < 1e-5 * float(flat.std() + flat.mean(1).abs().max())
        assert _rel(invstd, torch.rsqrt(flat.var(1, unbiased=False) + 1e-5)) < 1e-4
    xd = x.double().requires_grad_(True)
    F.conv3d(xd, wt.double(), None, 1, p).backward(dy.double())
    dxa = sfhip.conv_dgrad(_act(dy), wtp, _act(x), k, (1, 1, 1), p)
    assert _rel(_ncthw(dxa), xd.grad) < TOL, name


WG_SHAPES = [
    ("s3_128_128", 128, 128, (1, 3, 3), (0, 1, 1), (3, 2, 19, 17)),
    ("t3_256_192", 256, 192, (3, 1, 1), (1, 0, 0), (2, 5, 13, 12)),
    ("plain_64_256", 64, 256, (1, 1, 1), (0, 0, 0), (3, 2, 23, 21)),
    ("plain_320_72", 320, 72, (1, 1, 1), (0, 0, 0), (2, 3, 17, 19)),   # ragged channel blocks on both sides
]


@pytest.mark.parametrize("target", [0, 40, 200, 1536])
@pytest.mark.parametrize("blocks", [-1, 0, 1, 2, 3])
@pytest.mark.parametrize("shape", WG_SHAPES, ids=[s[0] for s in WG_SHAPES])
def test_forced_wgrad_wave_configuration(shape, blocks, target, tune):
    """conv_wgrad_wave.hip: every blocks-per-wavefront shape (sf_conv_tune(11, .): 1x1, 2x1, 1x2, 2x2) and several
    workgroup targets = position-split counts (sf_conv_tune(12, .)) against the fp64 weight gradient, both through the
    returned packed gradient and through the finish kernel that accumulates into a .grad-shaped tensor."""
    import sfhip
    name, cin, cout, k, p, dims = shape
    if blocks == -1 and target not in (0, 200):
        pytest.skip("planner's block shape: two split counts are enough")
    x, wt, bias, dy = _case(shape)
    assert tune.sf_conv_tune(11, blocks) == 0 and tune.sf_conv_tune(12, target) == 0 and tune.sf_conv_tune(9, 0) == 0
    wd = wt.double().requires_grad_(True)
    F.conv3d(x.double(), wd, None, 1, p).backward(dy.double())
    dwp = sfhip.conv_wgrad(_act(x), _act(dy), cout, k, (1, 1, 1), p)
    dw = sfhip.unpack_conv_weight_grad(dwp, wt.shape)
    assert _rel(dw, wd.grad) < TOL, (name, blocks, target)
    acc = torch.ones_like(wt)
    sfhip.conv_wgrad(_act(x), _act(dy), cout, k, (1, 1, 1), p, finish_into=(acc, cin, 0))
    assert _rel(acc - 1.0, wd.grad) < TOL, (name, blocks, target)


# ---- conv_bx.hip: fp32 products as six bf16 MFMAs on operand piece planes (forward, data gradient, weight gradient),
# forced onto every shape it covers (sf_conv_tune(7, 2) / (9, 2)) and held to an fp32-LEVEL bound against fp64 — the
# three-way split is exact and the dropped product terms are below 2^-24, so the results are as close as the f32 MFMA's
BX_TOL = 5e-6
BX_SHAPES = [
    # name, Cin, Cout, kernel, stride, pad, dil, (N, T, H, W)
    ("direct_128_128_s3", 128, 128, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 1, 1), (2, 4, 23, 21)),       # S = 1, ragged M
    ("splitk_512_512_s3", 512, 512, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 1, 1), (2, 8, 14, 14)),       # 26 tiles: S > 1
    ("t3_1024_256", 1024, 256, (3, 1, 1), (1, 1, 1), (1, 0, 0), (1, 1, 1), (3, 5, 13, 11)),            # temporal borders
    ("stride2_256_256", 256, 256, (1, 3, 3), (1, 2, 2), (0, 1, 1), (1, 1, 1), (3, 4, 28, 28)),          # tap-major K order
    ("ragged_144_192_dil2", 144, 192, (1, 3, 3), (1, 1, 1), (0, 2, 2), (1, 2, 2), (2, 3, 19, 23)),      # ragged N tile
    ("plain_1152_512", 1152, 512, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1), (2, 2, 25, 27)),
]


@pytest.mark.parametrize("shape", BX_SHAPES, ids=[s[0] for s in BX_SHAPES])
def test_bf16_piece_conv_forward_dgrad_wgrad(shape, tune):
    import ctypes
    import sfhip
    name, cin, cout, k, s, p, dl, (n, t, h, w) = shape
    dev = _dev()
    g = torch.Generator().manual_seed(len(name))
    x = torch.randn(n, cin, t, h, w, generator=g).to(dev)
    wt = (torch.randn(cout, cin, *k, generator=g) / np.sqrt(cin * k[0] * k[1] * k[2])).to(dev)
    bias = (torch.randn(cout, generator=g) * 0.1).to(dev)
    assert tune.sf_conv_tune(7, 2) == 0 and tune.sf_conv_tune(9, 2) == 0
    try:
        # the input is a channel slice of a wider buffer, the output lands in a slice of a wider buffer
        xfull = sfhip.Act(torch.randn(n, t, h, w, cin + 32, generator=g).to(dev))
        xa = xfull.slice(16, cin)
        xa.buf[..., 16:16 + cin] = x.permute(0, 2, 3, 4, 1)
        wp, wtp = sfhip.pack_conv_weight_pair(wt)
        ref = F.conv3d(x.double(), wt.double(), bias.double(), s, p, dl)
        od = ref.shape[2:]
        d = sfhip.ConvDesc(n, t, h, w, cin, xa.cs, xa.coff, od[0], od[1], od[2], cout, cout, 0, 1, k[0], k[1], k[2],
                           s[0], s[1], s[2], p[0], p[1], p[2], dl[0], dl[1], dl[2], cin, 0, 0, 0, 0)
        assert tune.sf_conv_bx_ws_floats(ctypes.byref(d), 0, 1) > 0, "the forced kernel must take this shape"
        z = sfhip.conv(xa, wp, k, s, p, dl, bias=bias, out_reserve=(8, 4))
        assert _rel(_ncthw(z), ref) < BX_TOL, name
        # scale + residual + ReLU epilogue, with the input's planes handed in (what the training path does)
        res = torch.randn_like(ref, dtype=torch.float32)
        sc = torch.rand(cout, device=dev) + 0.5
        keep = {}
        y = sfhip.conv(xa, wp, k, s, p, dl, scale=sc, bias=bias, relu=True, res=_act(res), keep=keep)
        assert sfhip.BX_AF32 or name.startswith("plain") or "x" in keep  # planes are kept only in the plane-fed mode
        ref2 = F.relu(F.conv3d(x.double(), wt.double(), None, s, p, dl) * sc.double().view(1, -1, 1, 1, 1) +
                      bias.double().view(1, -1, 1, 1, 1) + res.double())
        assert _rel(_ncthw(y), ref2) < BX_TOL, name
        # gradients
        dy = torch.randn(ref.shape, generator=g).to(dev)
        xd, wd = x.double().requires_grad_(True), wt.double().requires_grad_(True)
        F.conv3d(xd, wd, None, s, p, dl).backward(dy.double())
        dya = _act(dy)
        zp = sfhip.act_planes(dya)
        dxa = sfhip.conv_dgrad(dya, wtp, xa, k, s, p, dl, dz_planes=zp)
        assert _rel(_ncthw(dxa), xd.grad) < BX_TOL, name
        sfhip.conv_dgrad(dya, wtp, xa, k, s, p, dl, out=dxa, accumulate=True)     # planes made inside this time
        assert _rel(_ncthw(dxa), 2 * xd.grad) < BX_TOL, name
        dwp = sfhip.conv_wgrad(xa, dya, cout, k, s, p, dl, x_planes=keep.get("x"), dz_planes=zp)
        assert _rel(sfhip.unpack_conv_weight_grad(dwp, wt.shape), wd.grad) < BX_TOL, name
        acc = torch.ones_like(wt)
        sfhip.conv_wgrad(xa, dya, cout, k, s, p, dl, finish_into=(acc, cin, 0))  # planes made inside
        assert _rel(acc - 1.0, wd.grad) < BX_TOL, name
    finally:
        tune.sf_conv_tune(7, 1)
        tune.sf_conv_tune(9, 1)


# ---- conv_wgrad_rows.hip: the small-channel stride-1 "same" layers (all taps of a run of positions from one LDS image,
# direct-to-LDS loads, tap validity masks).  Ragged position counts (tails of the last stage and of the last split),
# clips whose borders fall inside a stage, channel counts that fill a 16- / 32-wide block partly, wide other sides.
ROWS_SHAPES = [
    ("s3_8_8", 8, 8, (1, 3, 3), (0, 1, 1), (2, 3, 19, 17)),
    ("s3_16_16", 16, 16, (1, 3, 3), (0, 1, 1), (3, 2, 13, 29)),
    ("s3_32_32", 32, 32, (1, 3, 3), (0, 1, 1), (2, 5, 14, 14)),
    ("s3_24_8", 24, 8, (1, 3, 3), (0, 1, 1), (1, 2, 56, 56)),
    ("t3_32_8", 32, 8, (3, 1, 1), (1, 0, 0), (2, 5, 13, 11)),
    ("t3_16_8", 16, 8, (3, 1, 1), (1, 0, 0), (3, 1, 9, 10)),      # T = 1: only the centre tap is ever inside
    ("t3_128_32", 128, 32, (3, 1, 1), (1, 0, 0), (2, 4, 7, 9)),
    ("t3_8_32", 8, 32, (3, 1, 1), (1, 0, 0), (2, 6, 11, 10)),
    ("plain_8_32", 8, 32, (1, 1, 1), (0, 0, 0), (2, 3, 23, 21)),
    ("plain_16_64", 16, 64, (1, 1, 1), (0, 0, 0), (3, 2, 17, 19)),
    ("plain_32_128", 32, 128, (1, 1, 1), (0, 0, 0), (2, 7, 14, 14)),
    ("plain_32_96", 32, 96, (1, 1, 1), (0, 0, 0), (1, 2, 9, 33)),
    ("plain_8_24", 8, 24, (1, 1, 1), (0, 0, 0), (2, 2, 15, 15)),
    ("plain_256_32", 256, 32, (1, 1, 1), (0, 0, 0), (1, 3, 11, 13)),
]


@pytest.mark.parametrize("sliced", [False, True])
@pytest.mark.parametrize("shape", ROWS_SHAPES, ids=[s[0] for s in ROWS_SHAPES])
def test_small_channel_wgrad_rows_kernel(shape, sliced, tune):
    """sf_conv_wgrad on the shapes conv_wgrad_rows.hip takes, against the fp64 weight gradient — through the packed
    gradient and through the finish kernel — and against the kernel it replaces (sf_conv_tune(20, 0)) as a cross-check
    that the launcher really switched paths (the split counts differ).  `sliced`: both operands are channel slices of
    wider buffers (pitch > channels, offset > 0), as the concatenating layers hand them over."""
    import ctypes
    import sfhip
    name, cin, cout, k, p, dims = shape
    x, wt, bias, dy = _case(shape)
    xa, dya = _act(x), _act(dy)
    if sliced:
        def widen(a, before, after):
            buf = torch.randn(a.buf.shape[:-1] + (before + a.C + after,), device=a.buf.device)
            buf[..., before:before + a.C] = a.buf
            return sfhip.Act(buf, before, a.C)
        xa, dya = widen(xa, 8, 4), widen(dya, 4, 12)
    wd = wt.double().requires_grad_(True)
    F.conv3d(x.double(), wd, None, 1, p).backward(dy.double())
    d = sfhip.ConvDesc(xa.N, xa.T, xa.H, xa.W, cin, xa.cs, xa.coff, dya.T, dya.H, dya.W, cout, 0, 0, 1, k[0], k[1], k[2],
                       1, 1, 1, p[0], p[1], p[2], 1, 1, 1, (cin + 15) // 16 * 16, 0, 0, 0, 0)
    assert tune.sf_conv_tune(20, 1) == 0
    s_on = tune.sf_conv_wgrad_splits(ctypes.byref(d))
    dwp = sfhip.conv_wgrad(xa, dya, cout, k, (1, 1, 1), p)
    assert _rel(sfhip.unpack_conv_weight_grad(dwp, wt.shape), wd.grad) < 2e-6, name
    acc = torch.ones_like(wt)
    sfhip.conv_wgrad(xa, dya, cout, k, (1, 1, 1), p, finish_into=(acc, cin, 0))
    assert _rel(acc - 1.0, wd.grad) < 2e-6, name
    assert tune.sf_conv_tune(20, 0) == 0
    s_off = tune.sf_conv_wgrad_splits(ctypes.byref(d))
    dwp0 = sfhip.conv_wgrad(xa, dya, cout, k, (1, 1, 1), p)
    assert _rel(sfhip.unpack_conv_weight_grad(dwp0, wt.shape), wd.grad) < 2e-6, name
    assert (s_on, s_off) != (0, 0)


# ---- the ring over t of the 3x1x1 weight gradients (conv_wgrad_tring_kernel, round 6): several t segments per column,
# 128- and 64-position blocks with a partly filled last block, frames at both ends of the clip, prefetch distance 1 / 2
TRING_SHAPES = [
    ("tr_32_8_hw1600", 32, 8, (3, 1, 1), (1, 0, 0), (2, 8, 40, 40)),
    ("tr_64_16_hw784", 64, 16, (3, 1, 1), (1, 0, 0), (2, 6, 28, 28)),
    ("tr_8_32_hw3136", 8, 32, (3, 1, 1), (1, 0, 0), (1, 4, 56, 56)),
    ("tr_128_32_hw196", 128, 32, (3, 1, 1), (1, 0, 0), (3, 5, 14, 14)),
    ("tr_16_8_T2", 16, 8, (3, 1, 1), (1, 0, 0), (4, 2, 23, 21)),
]


@pytest.mark.parametrize("shape", TRING_SHAPES, ids=[s[0] for s in TRING_SHAPES])
def test_wgrad_ring_over_t(shape, tune):
    import ctypes
    import sfhip
    name, cin, cout, k, p, dims = shape
    x, wt, bias, dy = _case(shape)
    xa, dya = _act(x), _act(dy)
    wd = wt.double().requires_grad_(True)
    F.conv3d(x.double(), wd, None, 1, p).backward(dy.double())
    d = sfhip.ConvDesc(xa.N, xa.T, xa.H, xa.W, cin, xa.cs, xa.coff, dya.T, dya.H, dya.W, cout, 0, 0, 1, k[0], k[1], k[2],
                       1, 1, 1, p[0], p[1], p[2], 1, 1, 1, (cin + 15) // 16 * 16, 0, 0, 0, 0)
    assert tune.sf_conv_tune(23, 1) == 0
    s_ring = tune.sf_conv_wgrad_splits(ctypes.byref(d))
    a = sfhip.conv_wgrad(xa, dya, cout, k, (1, 1, 1), p)
    b = sfhip.conv_wgrad(xa, dya, cout, k, (1, 1, 1), p)
    assert torch.equal(a, b), "bit-reproducible"
    assert _rel(sfhip.unpack_conv_weight_grad(a, wt.shape), wd.grad) < 2e-6, name
    acc = torch.ones_like(wt)
    sfhip.conv_wgrad(xa, dya, cout, k, (1, 1, 1), p, finish_into=(acc, cin, 0))
    assert _rel(acc - 1.0, wd.grad) < 2e-6, name
    assert tune.sf_conv_tune(23, 0) == 0
    s_rows = tune.sf_conv_wgrad_splits(ctypes.byref(d))
    c = sfhip.conv_wgrad(xa, dya, cout, k, (1, 1, 1), p)
    assert _rel(sfhip.unpack_conv_weight_grad(c, wt.shape), wd.grad) < 2e-6, name
    assert s_ring > 0 and s_rows > 0, (s_ring, s_rows)   # (both forms serve the shape; their split counts may coincide)


# ---- conv_pw_bx_kernel: pointwise layers on the bf16 pipe with the activations split in registers (sf_conv_tune(21, 2)
# forces it onto every shape it covers).  Ragged row counts (M tiles with rows past M), channel counts that fill a
# 256- / 128-wide column block partly, K from 4 to 64 steps, a channel-slice input.
PW_SHAPES = [
    ("pw_64_256", 64, 256, (3, 2, 23, 21)),
    ("pw_256_64", 256, 64, (2, 4, 19, 17)),
    ("pw_128_192", 128, 192, (3, 5, 13, 11)),
    ("pw_512_320", 512, 320, (3, 4, 14, 14)),
    ("pw_1024_256", 1024, 256, (3, 4, 14, 13)),
    ("pw_80_72", 80, 72, (3, 3, 17, 19)),
    ("pw_320_64", 320, 64, (2, 3, 21, 19)),   # 256 x 64 tile (round 6) under a 20-step reduction, ragged rows
]


@pytest.mark.parametrize("shape", PW_SHAPES, ids=[s[0] for s in PW_SHAPES])
def test_pointwise_bf16_piece_conv(shape, tune):
    """Forward (bias; training-mode statistics from the epilogue; eval epilogue with scale / bias / residual / ReLU),
    data gradient written and accumulated, all against fp64 at the fp32-level bound of the other bf16-piece kernels;
    and the launcher's route: sf_conv_pw_ws_floats > 0 exactly when forced on / 0 when switched off."""
    import ctypes
    import sfhip
    name, cin, cout, (n, t, h, w) = shape
    dev = _dev()
    k, p1 = (1, 1, 1), (0, 0, 0)
    g = torch.Generator().manual_seed(cin + 3 * cout)
    x = torch.randn(n, cin, t, h, w, generator=g).to(dev)
    wt = (torch.randn(cout, cin, 1, 1, 1, generator=g) / np.sqrt(cin)).to(dev)
    bias = (torch.randn(cout, generator=g) * 0.1).to(dev)
    scale = (torch.rand(cout, generator=g) + 0.5).to(dev)
    res = torch.randn(n, cout, t, h, w, generator=g).to(dev)
    dy = torch.randn(n, cout, t, h, w, generator=g).to(dev)
    assert tune.sf_conv_tune(21, 2) == 0
    wp, wtp = sfhip.pack_conv_weight_pair(wt)
    xa = _act(x)
    wide = sfhip.Act(torch.randn(xa.buf.shape[:-1] + (cin + 24,), device=dev), 16, cin)  # slice of a wider buffer
    wide.buf[..., 16:16 + cin] = xa.buf
    d = sfhip.ConvDesc(xa.N, xa.T, xa.H, xa.W, cin, xa.cs, xa.coff, xa.T, xa.H, xa.W, cout, cout, 0, 1, 1, 1, 1, 1, 1, 1,
                       0, 0, 0, 1, 1, 1, cin, 0, 0, 0, 0)
    assert tune.sf_conv_pw_ws_floats(ctypes.byref(d), 1) > 0
    ref = F.conv3d(x.double(), wt.double(), bias.double())
    for a in (xa, wide):
        z, st = sfhip.conv(a, wp, k, bias=bias, stats=True)
        assert _rel(_ncthw(z), ref) < BX_TOL, name
        assert st is not None, "pointwise layers leave their statistics from the epilogue"
        ones = torch.ones(cout, device=dev)
        mean, invstd, _, _ = sfhip.bn_train_stats_merge(st, cout, ones, ones, 1e-5, 0.1, None, None)
        flat = ref.transpose(0, 1).reshape(cout, -1)
        assert float((mean.double() - flat.mean(1)).abs().max()) < 1e-5 * float(flat.std() + flat.mean(1).abs().max())
        assert _rel(invstd, torch.rsqrt(flat.var(1, unbiased=False) + 1e-5)) < 1e-4
    y = sfhip.conv(xa, wp, k, scale=scale, bias=bias, relu=True, res=_act(res))
    ref2 = torch.relu(F.conv3d(x.double(), wt.double()) * scale.double().view(1, -1, 1, 1, 1) +
                      bias.double().view(1, -1, 1, 1, 1) + res.double())
    assert _rel(_ncthw(y), ref2) < BX_TOL, name
    xd = x.double().requires_grad_(True)
    F.conv3d(xd, wt.double()).backward(dy.double())
    dxa = sfhip.conv_dgrad(_act(dy), wtp, xa, k)
    assert _rel(_ncthw(dxa), xd.grad) < BX_TOL, name
    sfhip.conv_dgrad(_act(dy), wtp, xa, k, out=dxa, accumulate=True)
    assert _rel(_ncthw(dxa), 2 * xd.grad) < BX_TOL, name
    assert tune.sf_conv_tune(21, 0) == 0
    assert tune.sf_conv_pw_ws_floats(ctypes.byref(d), 1) == 0
    z0 = sfhip.conv(xa, wp, k, bias=bias)
    assert _rel(_ncthw(z0), ref) < TOL, name


# ---- the C-ABI's own workspace rule with operands the bf16-piece paths refuse at run time (round-4 advisor finding):
# sf_conv_fwd_ws_floats sizes the workspace from the SHAPE (weight planes for pointwise layers, planes + partial tiles
# for the long-reduction kernel); an input that is not 16-byte aligned is refused by those paths at run time and the
# call falls through to the generic implicit GEMM, which must then not run split-K into a workspace sized for planes.
@pytest.mark.parametrize("shape", [
    ("pw_256_1024", 256, 1024, (1, 1, 1), (0, 0, 0), (2, 4, 16, 17)),     # pointwise layer: ws = weight planes only
    ("bx_512_512_s3", 512, 512, (1, 3, 3), (0, 1, 1), (2, 8, 14, 14)),    # long reduction: ws = planes + partial tiles
], ids=["pw", "bx"])
def test_cabi_workspace_rule_with_misaligned_input(shape, tune):
    import ctypes
    import sfhip
    name, cin, cout, k, p, (n, t, h, w) = shape
    dev = _dev()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(n, t, h, w, cin, generator=g).to(dev)
    wt = (torch.randn(cout, cin, *k, generator=g) / np.sqrt(cin * k[0] * k[1] * k[2])).to(dev)
    wp, _ = sfhip.pack_conv_weight_pair(wt)
    tune.sf_conv_tune(7, 2)   # every shape the bf16-piece kernels cover: the sizing rule answers for them
    tune.sf_conv_tune(21, 2)
    d = sfhip.ConvDesc(n, t, h, w, cin, cin, 0, t, h, w, cout, cout, 0, 1, k[0], k[1], k[2], 1, 1, 1, p[0], p[1], p[2],
                       1, 1, 1, cin, 0, 0, 0, 0)
    nws = tune.sf_conv_fwd_ws_floats(ctypes.byref(d))
    assert nws > 0
    CANARY = 4096
    ws = torch.full((nws + CANARY,), 7.25, device=dev)
    # the same rows 4 bytes further: x.data_ptr() + 4 is not 16-byte aligned
    shifted = torch.empty(x.numel() + 1, device=dev)
    shifted[1:] = x.reshape(-1)
    out = torch.empty(n, t, h, w, cout, device=dev)
    rc = tune.sf_conv_fwd_ws(ctypes.byref(d), ctypes.c_void_p(shifted.data_ptr() + 4), ctypes.c_void_p(wp.data_ptr()),
                             None, None, None, ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(ws.data_ptr()),
                             ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    assert rc == 0, rc
    assert bool((ws[nws:] == 7.25).all()), "write past the workspace the header's sizing rule asked for"
    ref = F.conv3d(x.permute(0, 4, 1, 2, 3).double(), wt.double(), None, 1, p)
    assert _rel(out.permute(0, 4, 1, 2, 3), ref) < TOL
    # and through the Python binding: the run-time refusal falls back to the f32 kernels instead of raising
    xa = sfhip.Act(shifted[1:].view(n, t, h, w, cin))
    z = sfhip.conv(xa, wp, k, (1, 1, 1), p)
    assert _rel(_ncthw(z), ref) < TOL
