"""ORACLE — test infrastructure, NOT product code.

CPU restatement (torch functional ops, fp32 or fp64, NCTHW) of the reference's SlowFast hot
path, written from SURVEY.md §8(a) and the reference lines cited on each function.  It is
driven by a plain ``state_dict`` (the reference's key names) plus a small hyper-parameter dict,
so it shares no code with either the reference or the HIP product path.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module, and only as the checker.  The product (``efficient-slowfast_amd/``) never does.

Parity pin: the reference ships no tests for this path, so the oracle is pinned
against golden vectors generated in the build container by importing the reference itself
(``tests/golden/make_golden.py`` → ``tests/golden/*.npz``); ``tests/test_oracle_golden.py``
checks every one of them.

The arithmetic itself lives in PyTorch (third-party; the reference pins only "PyTorch 1.3" in
prose, SlowFast/INSTALL.md:6): conv3d / batch_norm / max_pool3d / bmm / softmax here are the
same ATen CPU kernels the reference's nn.Modules call.
"""
import math

import torch
import torch.nn.functional as F

# Test aid: ACT_HOOK(kind, x) -> tensor or None, kind in {"relu", "relu6"}.  The gradient tests hand the oracle the
# activation masks of the HIP forward (y = x * mask), so that both sides differentiate the SAME piecewise-linear
# function: a pre-activation within fp32 rounding of zero otherwise lands on different sides of the kink in the two
# implementations and moves every upstream gradient (tests/_masks.py).  None (default) = plain F.relu / F.relu6.
ACT_HOOK = None


POOL_HOOK = None  # POOL_HOOK(x, kernel, stride, padding) -> tensor or None: the HIP forward's arg-max decisions (same aid)


def _t3(v):
    return (v, v, v) if isinstance(v, int) else tuple(int(e) for e in v)


def _max_pool3d(x, kernel, stride=None, padding=0):
    kernel, stride, padding = _t3(kernel), _t3(kernel if stride is None else stride), _t3(padding)
    if POOL_HOOK is not None:
        y = POOL_HOOK(x, kernel, stride, padding)
        if y is not None:
            return y
    return F.max_pool3d(x, kernel, stride, padding)


def _relu(x):
    if ACT_HOOK is not None:
        y = ACT_HOOK("relu", x)
        if y is not None:
            return y
    return F.relu(x)


# ----------------------------------------------------------------------------- tables
# video_model_builder.py:16-17 / custom_video_model_builder.py:151-152
MODEL_STAGE_DEPTH = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3), 18: (2, 2, 2, 2), 34: (3, 4, 6, 3)}
# custom_video_model_builder.py:155-163 ("slowfast" arch): [stage][pathway] temporal kernel
TEMPORAL_KERNEL = [[1, 5], [1, 3], [1, 3], [3, 3], [3, 3]]
# video_model_builder.py:20-75: single-pathway bases, [stage] -> repeating pattern of temporal kernels
SINGLE_TEMPORAL_KERNEL = {
    "c2d": [[1], [1], [1], [1], [1]], "c2d_nopool": [[1], [1], [1], [1], [1]],
    "i3d": [[5], [3], [3, 1], [3, 1], [1, 3]], "i3d_nopool": [[5], [3], [3, 1], [3, 1], [1, 3]],
    "slow": [[1], [1], [1], [3], [3]],
}
SINGLE_POOL1 = {"c2d": (2, 1, 1), "c2d_nopool": (1, 1, 1), "i3d": (2, 1, 1), "i3d_nopool": (1, 1, 1),
                "slow": (1, 1, 1)}  # video_model_builder.py:77-84


def default_hparams(**over):
    """Hyper-parameters the reference's ctor reads from cfg."""
    hp = dict(
        alpha=4, beta_inv=8, depth=50, width_per_group=64, num_groups=1,
        fusion_conv_channel_ratio=2, fusion_kernel=7,
        spatial_strides=(1, 2, 2, 2), spatial_dilations=(1, 1, 1, 1),
        num_block_temp_kernel=((3, 3), (4, 4), (6, 6), (3, 3)),
        num_frames=32, crop_size=224, num_classes=400, short_cycle=False,
        head_act="softmax", width_multi=0.25, eps=1e-5,
    )
    hp.update(over)
    return hp


# ----------------------------------------------------------------------------- primitives
def _bn(sd, p, x, training, eps=1e-5):
    """nn.BatchNorm3d forward (batchnorm_helper.py:15-34 → torch).  training=True uses biased
    batch statistics (running buffers are left untouched: the oracle is stateless)."""
    if (p + ".split_bn.running_mean") in sd:
        return _sub_bn(sd, p, x, training, eps)
    w, b = sd[p + ".weight"], sd[p + ".bias"]
    if training:
        rec = sd.get("__bn_batch_stats__")  # optional recorder: {prefix: (batch mean, UNBIASED batch var)}
        if rec is not None:
            rec[p] = (x.mean((0, 2, 3, 4)).detach(), x.transpose(0, 1).reshape(x.shape[1], -1).var(1).detach())
        return F.batch_norm(x, None, None, w, b, True, 0.0, eps)
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], w, b, False, 0.0, eps)


def _sub_bn(sd, p, x, training, eps=1e-5):
    """SubBatchNorm3d (batchnorm_helper.py:37-109): training = affine-free BN over the batch viewed as
    [N/S, S*C, T, H, W] (sample n -> split n % S), eval = affine-free BN with the aggregated `bn` running
    statistics; then the single shared weight / bias."""
    n, c, t, h, w_ = x.shape
    if training:
        S = sd[p + ".split_bn.running_mean"].numel() // c
        y = F.batch_norm(x.reshape(n // S, c * S, t, h, w_), None, None, None, None, True, 0.0, eps)
        y = y.reshape(n, c, t, h, w_)
    else:
        y = F.batch_norm(x, sd[p + ".bn.running_mean"], sd[p + ".bn.running_var"], None, None, False, 0.0, eps)
    if (p + ".weight") in sd:
        y = y * sd[p + ".weight"].view(-1, 1, 1, 1) + sd[p + ".bias"].view(-1, 1, 1, 1)
    return y


def sub_bn_aggregate(means, variances, n):
    """SubBatchNorm3d._get_aggregated_mean_std (batchnorm_helper.py:66-79): mean of the split means, mean of the
    split variances plus the variance of the split means."""
    m = means.view(n, -1)
    mean = m.sum(0) / n
    return mean, variances.view(n, -1).sum(0) / n + ((m - mean) ** 2).sum(0) / n


def _conv(sd, p, x, stride=1, padding=0, dilation=1, groups=1):
    return F.conv3d(x, sd[p + ".weight"], sd.get(p + ".bias"), stride, padding, dilation, groups)


# ----------------------------------------------------------------------------- stem / res blocks
def resnet_basic_stem(sd, p, x, kt, training):
    """stem_helper.py:153-178: conv [kt,7,7]/s(1,2,2)/p(kt//2,3,3) → BN → ReLU → maxpool 1x3x3/2."""
    x = _conv(sd, p + ".conv", x, (1, 2, 2), (kt // 2, 3, 3))
    x = _relu(_bn(sd, p + ".bn", x, training))
    return _max_pool3d(x, (1, 3, 3), (1, 2, 2), (0, 1, 1))


def basic_transform(sd, p, x, kt, stride, training):
    """resnet_helper.py:25-107 (BasicTransform._construct :71-96, forward :98-106): [kt,3,3] conv / s(1,s,s) /
    p(kt//2,1,1) -> BN -> ReLU -> [1,3,3] conv / p(0,1,1) -> BN (no final ReLU: ResBlock adds the shortcut first)."""
    x = _conv(sd, p + ".a", x, (1, stride, stride), (kt // 2, 1, 1))
    x = _relu(_bn(sd, p + ".a_bn", x, training))
    x = _conv(sd, p + ".b", x, 1, (0, 1, 1))
    return _bn(sd, p + ".b_bn", x, training)


def bottleneck(sd, p, x, kt, stride, groups, dilation, training):
    """resnet_helper.py:169-240.  SlowFast* never pass stride_1x1 → stride sits on the 3x3."""
    x = _conv(sd, p + ".a", x, 1, (kt // 2, 0, 0))
    x = _relu(_bn(sd, p + ".a_bn", x, training))
    x = _conv(sd, p + ".b", x, (1, stride, stride), (0, dilation, dilation),
              (1, dilation, dilation), groups)
    x = _relu(_bn(sd, p + ".b_bn", x, training))
    x = _conv(sd, p + ".c", x)
    return _bn(sd, p + ".c_bn", x, training)


def res_block(sd, p, x, kt, stride, groups, dilation, training):
    """resnet_helper.py:310-358: projection shortcut iff branch1 exists in the state_dict
    (dim_in != dim_out or stride != 1)."""
    f = bottleneck(sd, p + ".branch2", x, kt, stride, groups, dilation, training)
    if (p + ".branch1.weight") in sd:
        sc = _bn(sd, p + ".branch1_bn", _conv(sd, p + ".branch1", x, (1, stride, stride)), training)
    else:
        sc = x
    return _relu(sc + f)


def nonlocal_block(sd, p, x, pool_size, instantiation, training):
    """nonlocal_helper.py:105-148: theta from x, phi / g from the (optionally max-pooled) x, softmax(theta^T phi /
    sqrt(d)) or theta^T phi / N_k ("dot_product"), times g, 1x1x1 conv_out + BN, added to x."""
    n, c, t, h, w = x.shape
    theta = _conv(sd, p + ".conv_theta", x)
    d = theta.shape[1]
    xp = x
    if pool_size is not None and any(k > 1 for k in pool_size):
        xp = _max_pool3d(x, tuple(pool_size), tuple(pool_size))
    phi = _conv(sd, p + ".conv_phi", xp).reshape(n, d, -1)
    g = _conv(sd, p + ".conv_g", xp).reshape(n, d, -1)
    tp = torch.einsum("nct,ncp->ntp", theta.reshape(n, d, -1), phi)
    if instantiation == "softmax":
        tp = torch.softmax(tp * (d ** -0.5), dim=2)
    elif instantiation == "dot_product":
        tp = tp / tp.shape[2]
    else:
        raise NotImplementedError(instantiation)
    y = torch.einsum("ntg,ncg->nct", tp, g).reshape(n, d, t, h, w)
    return x + _bn(sd, p + ".bn", _conv(sd, p + ".conv_out", y), training)


def res_stage(sd, p, xs, stage_idx, hp, training):
    """resnet_helper.py:444-448, 530-561: per pathway, block i gets temporal kernel
    (k*n)[:num_block_temp_kernel] + [1]*rest and stride only on block 0."""
    depth = MODEL_STAGE_DEPTH[hp["depth"]][stage_idx]
    out = []
    for pw, x in enumerate(xs):
        if len(xs) == 1:
            basis = list(SINGLE_TEMPORAL_KERNEL[hp["arch"]][stage_idx + 1])
        else:
            basis = [TEMPORAL_KERNEL[stage_idx + 1][pw]]
        nbtk = hp["num_block_temp_kernel"][stage_idx][pw]
        kts = (basis * depth)[:nbtk] + [1] * (depth - nbtk)
        for i in range(depth):
            x = res_block(sd, "%s.pathway%d_res%d" % (p, pw, i), x, kts[i],
                          hp["spatial_strides"][stage_idx] if i == 0 else 1,
                          hp["num_groups"], hp["spatial_dilations"][stage_idx], training)
            nl = "%s.pathway%d_nonlocal%d" % (p, pw, i)
            if (nl + ".conv_theta.weight") in sd:  # resnet_helper.py:519-528, 541-559
                grp = hp["nonlocal_group"][stage_idx][pw]
                b, c, t, h, w = x.shape
                if grp > 1:  # fold groups of frames into the batch
                    x = x.permute(0, 2, 1, 3, 4).reshape(b * grp, t // grp, c, h, w).permute(0, 2, 1, 3, 4)
                x = nonlocal_block(sd, nl, x, hp["nonlocal_pool"][stage_idx][pw], hp["nonlocal_instantiation"],
                                   training)
                if grp > 1:
                    x = x.permute(0, 2, 1, 3, 4).reshape(b, t, c, h, w).permute(0, 2, 1, 3, 4)
        out.append(x)
    return out


# ----------------------------------------------------------------------------- lateral fusions
def fuse_fast_to_slow(sd, p, xs, hp, training):
    """video_model_builder.py:128-150: conv [K,1,1]/s(alpha,1,1)/p(K//2) → BN → ReLU → cat on slow."""
    k = hp["fusion_kernel"]
    f = _conv(sd, p + ".conv_f2s", xs[1], (hp["alpha"], 1, 1), (k // 2, 0, 0))
    f = _relu(_bn(sd, p + ".bn", f, training))
    return [torch.cat([xs[0], f], 1), xs[1]]


def eca(sd, p, x):
    """wdf_attention_helper.py:77-91: global avg-pool → Conv1d(1,1,3,pad=1) ALONG CHANNELS →
    sigmoid → broadcast multiply."""
    y = x.mean((2, 3, 4))  # [B, C]
    y = F.conv1d(y.unsqueeze(1), sd[p + ".conv.weight"], None, 1, 1).squeeze(1)
    return x * torch.sigmoid(y)[:, :, None, None, None]


def spatial_attention(sd, p, x):
    """wdf_attention_helper.py:33-54: full N x N softmax self-attention over N = T*H*W,
    NO 1/sqrt(d) scaling, q/k/v 1x1x1 convs with bias, out = gamma * (V attn^T) + x."""
    b, c, t, h, w = x.shape
    n = t * h * w
    q = _conv(sd, p + ".query_conv", x).reshape(b, -1, n)
    k = _conv(sd, p + ".key_conv", x).reshape(b, -1, n)
    v = _conv(sd, p + ".value_conv", x).reshape(b, -1, n)
    attn = torch.softmax(torch.bmm(q.transpose(1, 2), k), dim=-1)  # [B, N(query), N(key)]
    out = torch.bmm(v, attn.transpose(1, 2)).reshape(b, c, t, h, w)
    return sd[p + ".gamma"] * out + x


def fuse_fast_and_slow(sd, p, xs, hp, training):
    """CMDA, custom_video_model_builder.py:123-148."""
    a = hp["alpha"]
    x_s, x_f = xs
    f2s = _max_pool3d(x_f, (a, 1, 1), (a, 1, 1))
    f2s = eca(sd, p + ".attention_channel_f2s", f2s)
    f2s = _relu(_bn(sd, p + ".bn_f2s", f2s, training))
    s_out = torch.cat([x_s, f2s], 1)
    s2f = _conv(sd, p + ".downsample_c_of_slow", x_s)
    s2f = spatial_attention(sd, p + ".attention_spatial_s2f", s2f)
    s2f = _relu(_bn(sd, p + ".bn_s2f", s2f, training))
    s2f = s2f.repeat_interleave(a, dim=2)  # nn.Upsample(scale=(a,1,1), nearest)
    f_out = torch.cat([s2f, x_f], 1)  # slow-derived channels FIRST (:146)
    return [s_out, f_out]


# ----------------------------------------------------------------------------- heads
def _head_tail(x, training, act):
    """eval only: activation over the class dim (dim=4 of N,T,H,W,C) then mean over T,H,W."""
    if not training:
        if act == "softmax":
            x = torch.softmax(x, dim=4)
        elif act == "sigmoid":
            x = torch.sigmoid(x)
        elif act == "relu":
            x = _relu(x)
        x = x.mean((1, 2, 3))
    return x.reshape(x.shape[0], -1)


def resnet_basic_head(sd, p, xs, hp, training):
    """head_helper.py:198-223 (dropout omitted: identity in eval, fixtures use rate 0 in train).
    pool_size = [T/alpha, CROP//32, CROP//32] / [T, ...] from cfg.DATA.CROP_SIZE
    (custom_video_model_builder.py:408-423); None → global pool when MULTIGRID.SHORT_CYCLE."""
    pooled = []
    for pw, x in enumerate(xs):
        if hp["short_cycle"]:
            pooled.append(x.mean((2, 3, 4), keepdim=True))
        else:
            if len(xs) == 1:  # ResNet: NUM_FRAMES // pool1_T (video_model_builder.py:587-593)
                t = hp["num_frames"] // SINGLE_POOL1[hp["arch"]][0]
            else:
                t = hp["num_frames"] // hp["alpha"] if pw == 0 else hp["num_frames"]
            s = hp["crop_size"] // 32
            pooled.append(F.avg_pool3d(x, (t, s, s), 1))
    x = torch.cat(pooled, 1).permute(0, 2, 3, 4, 1)
    logits = F.linear(x, sd[p + ".projection.weight"], sd[p + ".projection.bias"])
    return logits, _head_tail(logits, training, hp["head_act"])


class StopForward(Exception):
    """Raised by _mark when sd["__stop__"] names the boundary just recorded (test aid: child-level evaluation)."""


def _mark(sd, acts, name, x):
    """Record the output of top-level child `name`.  Test aids (tests/test_stage_grads*.py): sd["__override__"] =
    {name: tensors} substitutes the child's output — the NEXT child is then evaluated on exactly those tensors (e.g.
    the HIP path's own activations, as autograd leaves); sd["__stop__"] = name ends the forward here."""
    ov = sd.get("__override__")
    if ov is not None and name in ov:
        x = list(ov[name])
    acts[name] = x
    if sd.get("__stop__") == name:
        raise StopForward(acts)
    return x


# ----------------------------------------------------------------------------- R50 models
def slowfast_forward(sd, inputs, hp, dual, training=False):
    """SlowFast (video_model_builder.py:399-416) when dual=False,
    SlowFastDualAttention (custom_video_model_builder.py:428-445) when dual=True.
    Returns dict of every top-level child's output + 'logits' (pre-activation) + 'out'."""
    fuse = fuse_fast_and_slow if dual else fuse_fast_to_slow
    acts = {}
    x = [resnet_basic_stem(sd, "s1.pathway%d_stem" % i, inputs[i], TEMPORAL_KERNEL[0][i], training)
         for i in range(2)]
    x = _mark(sd, acts, "s1", x)
    for si in range(4):
        if si > 0:
            x = res_stage(sd, "s%d" % (si + 1), x, si - 1, hp, training)
            x = _mark(sd, acts, "s%d" % (si + 1), x)
        x = fuse(sd, "s%d_fuse" % (si + 1), x, hp, training)
        x = _mark(sd, acts, "s%d_fuse" % (si + 1), x)
        # pathway{0,1}_pool = MaxPool3d(k=s=[1,1,1]) → identity (:278-284, :433-435)
    x = res_stage(sd, "s5", x, 3, hp, training)
    x = _mark(sd, acts, "s5", x)
    acts["logits"], acts["out"] = resnet_basic_head(sd, "head", x, hp, training)
    return acts


def resnet_forward(sd, inputs, hp, training=False):
    """Single-pathway ResNet — C2D / I3D / Slow (video_model_builder.py:420-616): s1, s2, pathway0_pool
    (MaxPool3d(k = s = _POOL1[arch]), temporal for c2d / i3d), s3, s4, s5, head."""
    arch = hp["arch"]
    acts = {}
    x = [resnet_basic_stem(sd, "s1.pathway0_stem", inputs[0], SINGLE_TEMPORAL_KERNEL[arch][0][0], training)]
    x = _mark(sd, acts, "s1", x)
    for si in range(4):
        x = res_stage(sd, "s%d" % (si + 2), x, si, hp, training)
        x = _mark(sd, acts, "s%d" % (si + 2), x)
        if si == 0 and SINGLE_POOL1[arch] != (1, 1, 1):
            x = _mark(sd, acts, "pathway0_pool", [_max_pool3d(x[0], SINGLE_POOL1[arch], SINGLE_POOL1[arch])])
    acts["logits"], acts["out"] = resnet_basic_head(sd, "head", x, hp, training)
    return acts


# ----------------------------------------------------------------------------- ShuffleNetV2 (cfg #1)
SHUFFLENETV2_CHANNELS = {  # custom_video_model_builder.py:471-480
    0.25: [-1, 24, 32, 64, 128, 1024], 0.5: [-1, 24, 48, 96, 192, 1024],
    1.0: [-1, 24, 116, 240, 464, 1024], 1.5: [-1, 24, 176, 352, 704, 1024],
    2.0: [-1, 24, 224, 496, 976, 2048],
}
SHUFFLENETV2_REPEATS = [4, 8, 4]


def channel_shuffle(x, groups):
    """shufflenetv2_helper.py:32-43."""
    b, c = x.shape[:2]
    return x.reshape(b, groups, c // groups, *x.shape[2:]).transpose(1, 2).reshape(x.shape)


def _seq_conv_bn(sd, p, i, x, training, relu, stride=1, padding=0, groups=1):
    """conv at Sequential index i, BN at i+1 (hard-coded nn.BatchNorm3d, eps 1e-5)."""
    x = _bn(sd, "%s.%d" % (p, i + 1), _conv(sd, "%s.%d" % (p, i), x, stride, padding, 1, groups), training)
    return _relu(x) if relu else x


def shufflev2_block(sd, p, x, stride, training):
    """InvertedResidual, shufflenetv2_helper.py:47-112 (attribute names 'banch1/2' are the
    reference's spelling and therefore the state_dict's)."""
    def branch2(z):
        z = _seq_conv_bn(sd, p + ".banch2", 0, z, training, True)
        c = z.shape[1]
        z = _seq_conv_bn(sd, p + ".banch2", 3, z, training, False, (1, stride, stride), 1, c)
        return _seq_conv_bn(sd, p + ".banch2", 5, z, training, True)

    if stride == 1:
        half = x.shape[1] // 2
        out = torch.cat([x[:, :half], branch2(x[:, half:])], 1)
    else:
        b1 = _seq_conv_bn(sd, p + ".banch1", 0, x, training, False, (1, stride, stride), 1, x.shape[1])
        b1 = _seq_conv_bn(sd, p + ".banch1", 2, b1, training, True)
        out = torch.cat([b1, branch2(x)], 1)
    return channel_shuffle(out, 2)


def shufflenetv2_forward(sd, inputs, hp, training=False):
    """SlowFastShuffleNetV2, custom_video_model_builder.py:599-617."""
    chans = SHUFFLENETV2_CHANNELS[hp["width_multi"]]
    fchans = [c // hp["beta_inv"] for c in chans]
    acts = {}
    x = []
    for pw in range(2):  # stem_helper.py:237-270: conv3x3x3/s(1,2,2) BN ReLU MaxPool3d(3,(1,2,2),1)
        z = _seq_conv_bn(sd, "s1.pathway%d_stem" % pw, 0, inputs[pw], training, True, (1, 2, 2), 1)
        x.append(_max_pool3d(z, 3, (1, 2, 2), 1))
    x = _mark(sd, acts, "s1", x)
    x = fuse_fast_and_slow(sd, "s1_fuse", x, hp, training)
    x = _mark(sd, acts, "s1_fuse", x)
    for st in range(3):
        nxt = []
        for pw in range(2):
            cout = (chans if pw == 0 else fchans)[st + 2]
            p = "s%d.pathway%d_channel_%d.features" % (st + 2, pw, cout)
            z = x[pw]
            for i in range(SHUFFLENETV2_REPEATS[st]):
                z = shufflev2_block(sd, "%s.%d" % (p, i), z, 2 if i == 0 else 1, training)
            nxt.append(z)
        x = nxt
        x = _mark(sd, acts, "s%d" % (st + 2), x)
        x = fuse_fast_and_slow(sd, "s%d_fuse" % (st + 2), x, hp, training)
        x = _mark(sd, acts, "s%d_fuse" % (st + 2), x)
    pooled = []  # head_helper.py:499-557
    for pw in range(2):
        z = _seq_conv_bn(sd, "head.pathway%d_conv1x1x1.0" % pw, 0, x[pw], training, True)
        pooled.append(z.mean((2, 3, 4), keepdim=True))
    z = torch.cat(pooled, 1).permute(0, 2, 3, 4, 1)
    logits = F.linear(z, sd["head.classifier.1.weight"], sd["head.classifier.1.bias"])
    acts["logits"], acts["out"] = logits, _head_tail(logits, training, hp["head_act"])
    return acts


# ----------------------------------------------------------------------------- GhostNet (cfg #5)
GHOST_STAGE_CFGS = [  # custom_video_model_builder.py:816-843: k, exp, c, se_ratio, stride
    [[3, 16, 16, 0, 1]],
    [[3, 48, 24, 0, 2], [3, 72, 24, 0, 1]],
    [[5, 72, 40, 0.25, 2], [5, 120, 40, 0.25, 1]],
    [[3, 240, 80, 0, 2], [3, 200, 80, 0, 1], [3, 184, 80, 0, 1], [3, 184, 80, 0, 1],
     [3, 480, 112, 0.25, 1], [3, 672, 112, 0.25, 1]],
    [[5, 672, 160, 0.25, 2], [5, 960, 160, 0, 1], [5, 960, 160, 0.25, 1], [5, 960, 160, 0, 1],
     [5, 960, 160, 0.25, 1]],
]


def make_divisible(v, divisor, min_value=None):
    """ghostnet_helper.py:11-24."""
    if min_value is None:
        min_value = divisor
    new_v = max(min_value, int(v + divisor / 2) // divisor * divisor)
    if new_v < 0.9 * v:
        new_v += divisor
    return new_v


def ghost_cfgs(hp):
    """custom_video_model_builder.py:845-863 (note the float floor-division on the fast path)."""
    wm, bi = hp["width_multi"], hp["beta_inv"]
    slow, fast = [], []
    for st in GHOST_STAGE_CFGS:
        slow.append([[c[0], make_divisible(c[1] * wm, 4), make_divisible(c[2] * wm, 4), c[3], c[4]] for c in st])
        fast.append([[c[0], make_divisible(c[1] * wm // bi, 4), make_divisible(c[2] * wm // bi, 4), c[3], c[4]]
                     for c in st])
    return slow, fast


def ghost_module(sd, p, x, oup, relu, training):
    """ghostnet_helper.py:72-99 (kernel_size=1, ratio=2, dw_size=3, stride=1 as instantiated)."""
    x1 = _seq_conv_bn(sd, p + ".primary_conv", 0, x, training, relu)
    x2 = _seq_conv_bn(sd, p + ".cheap_operation", 0, x1, training, relu, 1, 1, x1.shape[1])
    return torch.cat([x1, x2], 1)[:, :oup]


def ghost_bottleneck(sd, p, x, k, mid, out, se_ratio, stride, training):
    """ghostnet_helper.py:105-163."""
    res = x
    pad = (k - 1) // 2
    y = ghost_module(sd, p + ".ghost1", x, mid, True, training)
    if stride > 1:
        y = _bn(sd, p + ".bn_dw", _conv(sd, p + ".conv_dw", y, (1, stride, stride), (0, pad, pad), 1, mid), training)
    if se_ratio is not None and se_ratio > 0:  # SqueezeExcite :34-52, hard_sigmoid gate :27-31
        s = y.mean((2, 3, 4), keepdim=True)
        s = _relu(_conv(sd, p + ".se.conv_reduce", s))
        s = _conv(sd, p + ".se.conv_expand", s)
        y = y * (F.relu6(s + 3.0) / 6.0)
    y = ghost_module(sd, p + ".ghost2", y, out, False, training)
    if (p + ".shortcut.0.weight") in sd:
        cin = res.shape[1]
        r = _bn(sd, p + ".shortcut.1",
                _conv(sd, p + ".shortcut.0", res, (1, stride, stride), (0, pad, pad), 1, cin), training)
        r = _bn(sd, p + ".shortcut.3", _conv(sd, p + ".shortcut.2", r), training)
    else:
        r = res
    return y + r


def ghostnet_forward(sd, inputs, hp, training=False):
    """SlowFastGhostNet, custom_video_model_builder.py:1007-1026."""
    slow_cfg, fast_cfg = ghost_cfgs(hp)
    acts = {}
    # stem_helper.py:310-336: conv3x3x3/s(1,2,2)/p1 BN ReLU, NO max-pool
    x = [_seq_conv_bn(sd, "s0.pathway%d_stem" % pw, 0, inputs[pw], training, True, (1, 2, 2), 1) for pw in range(2)]
    x = _mark(sd, acts, "s0", x)
    for st in range(5):
        nxt = []
        for pw in range(2):
            cfgs = (slow_cfg if pw == 0 else fast_cfg)[st]
            p = "s%d.pathway%d_channel_%d.features" % (st + 1, pw, cfgs[-1][2])
            z = x[pw]
            for i, (k, exp, c, se, s) in enumerate(cfgs):
                # GhostNet_Inverted_Residual_Block re-rounds with divisor 2 (ghostnet_helper.py:273-274)
                z = ghost_bottleneck(sd, "%s.%d" % (p, i), z, k, make_divisible(exp, 2), make_divisible(c, 2),
                                     se, s, training)
            nxt.append(z)
        x = nxt
        x = _mark(sd, acts, "s%d" % (st + 1), x)
        if st < 4:
            x = fuse_fast_and_slow(sd, "s%d_fuse" % (st + 1), x, hp, training)
            x = _mark(sd, acts, "s%d_fuse" % (st + 1), x)
    pooled = []  # head_helper.py:630-700
    for pw, nm in enumerate(("slow", "fast")):
        z = _relu(_bn(sd, "head.stage5_conv_%s.bn1" % nm, _conv(sd, "head.stage5_conv_%s.conv" % nm, x[pw]), training))
        z = z.mean((2, 3, 4), keepdim=True)
        pooled.append(_relu(_conv(sd, "head.conv_head_%s" % nm, z)))
    z = torch.cat(pooled, 1).permute(0, 2, 3, 4, 1)
    logits = F.linear(z, sd["head.classifier.1.weight"], sd["head.classifier.1.bias"])
    # head_helper.py:640-643 vs :653 — self.act (softmax) is overwritten by nn.ReLU: eval output is
    # relu(logits) averaged, not probabilities (bug-compatible, SURVEY.md §7 hard part 5).
    acts["logits"], acts["out"] = logits, _head_tail(logits, training, "relu")
    return acts


# ----------------------------------------------------------------------------- MobileNetV2
MOBILENETV2_SETTINGS = [  # custom_video_model_builder.py:1028-1047: t, c, n, stride
    [1, 16, 1, (1, 1, 1)], [6, 24, 2, (1, 2, 2)], [6, 32, 3, (1, 2, 2)], [6, 64, 4, (1, 2, 2)],
    [6, 96, 3, (1, 1, 1)], [6, 160, 3, (1, 2, 2)], [6, 320, 1, (1, 1, 1)],
]
MOBILENETV2_STAGES = [("s2", 0, 2), ("s4", 2, 3), ("s5", 3, 4), ("s6", 4, 5), ("s7", 5, 6), ("s8", 6, 7)]
MOBILENETV2_FUSE_AFTER = {"s2": "s3_fuse", "s4": "s4_fuse", "s5": "s5_fuse", "s7": "s7_fuse"}


def _relu6(x):
    if ACT_HOOK is not None:
        y = ACT_HOOK("relu6", x)
        if y is not None:
            return y
    return F.relu6(x)


def mbv2_block(sd, p, x, inp, oup, stride, t, training):
    """mobilenetv2_helper.py:30-68 InvertedResidual."""
    hidden = int(round(inp * t))
    q = p + ".conv"
    y = x
    i = 0
    if t != 1:
        y = _relu6(_bn(sd, "%s.1" % q, _conv(sd, "%s.0" % q, y), training))
        i = 3
    y = _relu6(_bn(sd, "%s.%d" % (q, i + 1), _conv(sd, "%s.%d" % (q, i), y, stride, 1, 1, hidden), training))
    y = _bn(sd, "%s.%d" % (q, i + 4), _conv(sd, "%s.%d" % (q, i + 3), y), training)
    if tuple(stride) == (1, 1, 1) and inp == oup:
        y = x + y
    return y


def mobilenetv2_forward(sd, inputs, hp, training=False):
    """SlowFastMoibleNetV2, custom_video_model_builder.py:1058-1285."""
    wm, bi = hp["width_multi"], hp["beta_inv"]
    acts = {}
    x = [_relu6(_bn(sd, "s1.pathway%d_stem.features.1" % pw,
                    _conv(sd, "s1.pathway%d_stem.features.0" % pw, inputs[pw], (1, 2, 2), 1), training))
         for pw in range(2)]
    x = _mark(sd, acts, "s1", x)
    for name, a, b in MOBILENETV2_STAGES:
        nxt = []
        for pw in range(2):
            z = x[pw]
            inp = z.shape[1]
            q = "%s.pathway%d_channel_%d.features" % (name, pw, MOBILENETV2_SETTINGS[a][1])
            idx = 0
            for t, c, n, s in MOBILENETV2_SETTINGS[a:b]:
                oup = int(c * wm) if pw == 0 else int(c * wm // bi)
                for i in range(n):
                    z = mbv2_block(sd, "%s.%d" % (q, idx), z, inp, oup, s if i == 0 else (1, 1, 1), t, training)
                    inp = oup
                    idx += 1
            nxt.append(z)
        x = nxt
        x = _mark(sd, acts, name, x)
        if name in MOBILENETV2_FUSE_AFTER:
            f = MOBILENETV2_FUSE_AFTER[name]
            x = fuse_fast_and_slow(sd, f, x, hp, training)
            x = _mark(sd, acts, f, x)
    pooled = []  # head_helper.py:436-486
    for pw in range(2):
        z = _relu6(_bn(sd, "head.pathway%d_conv1x1x1.1" % pw, _conv(sd, "head.pathway%d_conv1x1x1.0" % pw, x[pw]),
                       training))
        pooled.append(z.mean((2, 3, 4), keepdim=True))
    z = torch.cat(pooled, 1).permute(0, 2, 3, 4, 1)
    logits = F.linear(z, sd["head.classifier.1.weight"], sd["head.classifier.1.bias"])
    acts["logits"], acts["out"] = logits, _head_tail(logits, training, hp["head_act"])
    return acts


# ----------------------------------------------------------------------------- ShuffleNet v1
SHUFFLENET_OUT_PLANES = {1: [24, 144, 288, 567], 2: [24, 200, 400, 800], 3: [24, 240, 480, 960],
                         4: [24, 272, 544, 1088], 8: [24, 384, 768, 1536]}  # custom_video_model_builder.py:646-659


def _channel_shuffle(x, groups):
    """shufflenet_helper.py:22-29."""
    b, c, t, h, w = x.shape
    return x.view(b, groups, c // groups, t, h, w).permute(0, 2, 1, 3, 4, 5).reshape(b, c, t, h, w)


def shufflenet_bottleneck(sd, p, x, stride, groups, training):
    """shufflenet_helper.py:32-79: grouped 1x1 -> shuffle -> dw 3x3x3 -> grouped 1x1; stride 2 concatenates a
    1x1-conv + AvgPool3d((1,3,3),(1,2,2),(0,1,1)) shortcut, stride 1 adds the identity; ReLU last."""
    g1 = 1 if x.shape[1] == 24 else groups
    mid = sd[p + ".conv1.weight"].shape[0]
    out = _relu(_bn(sd, p + ".bn1", _conv(sd, p + ".conv1", x, groups=g1), training))
    out = _channel_shuffle(out, groups)
    out = _bn(sd, p + ".bn2", _conv(sd, p + ".conv2", out, (1, stride, stride), 1, 1, mid), training)
    out = _bn(sd, p + ".bn3", _conv(sd, p + ".conv3", out, groups=groups), training)
    if stride == 2:
        sc = F.avg_pool3d(_conv(sd, p + ".shortcut.0", x), (1, 3, 3), (1, 2, 2), (0, 1, 1))
        return _relu(torch.cat([out, sc], 1))
    return _relu(out + x)


def shufflenet_forward(sd, inputs, hp, training=False):
    """SlowFastShuffleNet, custom_video_model_builder.py:620-789."""
    groups = hp.get("groups", 1)
    planes = [int(c * hp["width_multi"]) for c in SHUFFLENET_OUT_PLANES[groups]]
    fast = [c // hp["beta_inv"] for c in planes]
    acts = {}
    x = []
    for pw in range(2):  # stem_helper.py:274-306
        q = "s1.pathway%d_stem" % pw
        z = _relu(_bn(sd, q + ".1", _conv(sd, q + ".0", inputs[pw], (1, 2, 2), 1), training))
        x.append(_max_pool3d(z, 3, (1, 2, 2), 1))
    x = _mark(sd, acts, "s1", x)
    x = fuse_fast_and_slow(sd, "s1_fuse", x, hp, training)
    x = _mark(sd, acts, "s1_fuse", x)
    for si, nb in enumerate((4, 8, 4)):
        name = "s%d" % (si + 2)
        nxt = []
        for pw in range(2):
            z = x[pw]
            q = "%s.pathway%d_channel_%d.features" % (name, pw, (planes if pw == 0 else fast)[si + 1])
            for i in range(nb):
                z = shufflenet_bottleneck(sd, "%s.%d" % (q, i), z, 2 if i == 0 else 1, groups, training)
            nxt.append(z)
        x = nxt
        x = _mark(sd, acts, name, x)
        x = fuse_fast_and_slow(sd, name + "_fuse", x, hp, training)
        x = _mark(sd, acts, name + "_fuse", x)
    z = torch.cat([t.mean((2, 3, 4), keepdim=True) for t in x], 1).permute(0, 2, 3, 4, 1)  # head_helper.py:562-609
    logits = F.linear(z, sd["head.classifier.1.weight"], sd["head.classifier.1.bias"])
    acts["logits"], acts["out"] = logits, _head_tail(logits, training, hp["head_act"])
    return acts


FORWARDS = {
    "SlowFast": lambda sd, x, hp, training=False: slowfast_forward(sd, x, hp, False, training),
    "SlowFastDualAttention": lambda sd, x, hp, training=False: slowfast_forward(sd, x, hp, True, training),
    "SlowFastShuffleNetV2": shufflenetv2_forward,
    "SlowFastGhostNet": ghostnet_forward,
    "SlowFastMoibleNetV2": mobilenetv2_forward,
    "SlowFastShuffleNet": shufflenet_forward,
    "ResNet": resnet_forward,
}


def forward(model_name, sd, inputs, hp, training=False):
    """Entry point: returns dict {stage: [slow, fast] | tensor, 'logits', 'out'}."""
    with torch.no_grad():
        return FORWARDS[model_name](sd, list(inputs), hp, training)


def pack_pathway_indices(t, alpha):
    """datasets/utils.py:93-104: slow frame indices = linspace(0, T-1, T//alpha).long()."""
    return torch.linspace(0, t - 1, t // alpha).long().tolist()
