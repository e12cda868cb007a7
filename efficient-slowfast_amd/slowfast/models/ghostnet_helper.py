"""GhostNet stages (reference ghostnet_helper.py).  A GhostModule is two launches writing the two halves
of one buffer (primary 1x1 GEMM -> channels [0, init), depthwise 3x3x3 -> channels [init, oup)), the
`[:, :oup]` slice is a channel count, the bottleneck's residual add runs in those two epilogues, and
SqueezeExcite is pool -> two tiny GEMMs -> one gate pass."""
import math

import torch
import torch.nn as nn

import sfhip
from . import engine
from .shufflenetv2_helper import _efficient_init


def _make_divisible(v, divisor, min_value=None):
    """Round channel counts the way the TF-slim mobilenet code does (ghostnet_helper.py:11-24)."""
    if min_value is None:
        min_value = divisor
    new_v = max(min_value, int(v + divisor / 2) // divisor * divisor)
    if new_v < 0.9 * v:
        new_v += divisor
    return new_v


class SqueezeExcite(nn.Module):
    """x * hard_sigmoid(conv_expand(relu(conv_reduce(avgpool(x))))) (ghostnet_helper.py:34-52)."""

    def __init__(self, in_chs, se_ratio=0.25, reduced_base_chs=None, divisor=4):
        super(SqueezeExcite, self).__init__()
        reduced_chs = _make_divisible((reduced_base_chs or in_chs) * se_ratio, divisor)
        self.avg_pool = nn.AdaptiveAvgPool3d(1)
        self.conv_reduce = nn.Conv3d(in_chs, reduced_chs, 1, bias=True)
        self.act1 = nn.ReLU(inplace=True)
        self.conv_expand = nn.Conv3d(reduced_chs, in_chs, 1, bias=True)

    def forward(self, x):
        pooled = engine.global_mean(x)                                   # [N,1,1,1,C] (HIP reduction, taped)
        # the two FCs act on [N, C] vectors: 1x1x1 convs over N rows on the library's own GEMM kernels (forward, data
        # and weight gradients) — no vendor BLAS launch in the step; the gate is applied by the HIP kernel below
        e = engine.conv_bn_act(engine.conv_bn_act(pooled, self.conv_reduce, relu=True), self.conv_expand)
        y = sfhip.gate_apply(x, 1, e.buf.view(x.N, x.C), w3=None)       # x * hard_sigmoid(e)
        t = engine.tape()
        if t is not None:
            def bwd():
                ev = e.buf.view(x.N, x.C)
                gate = (torch.clamp(ev + 3.0, 0.0, 6.0) / 6.0).contiguous()
                dy = t.grad_of(y)
                dgate = sfhip.tmax_dot(x, 1, dy)                         # sum_pos dy * x
                de = dgate * ((ev > -3.0) & (ev < 3.0)).to(torch.float32) / 6.0
                sfhip.axpy(sfhip.Act(de.contiguous().view(x.N, 1, 1, 1, x.C)), t.grad_of(e), 1.0, accumulate=True)
                zero = torch.zeros_like(gate)
                sfhip.eca_bwd_apply(x, 1, dy, gate, zero, t.grad_of(x))  # dx += dy * gate
            t.record(bwd)
        return y


class GhostModule(nn.Module):
    """primary 1xkxk conv+BN(+ReLU) -> cheap depthwise 3x3x3 conv+BN(+ReLU) -> cat -> [:oup]
    (ghostnet_helper.py:71-99)."""

    def __init__(self, inp, oup, kernel_size=1, ratio=2, dw_size=3, stride=1, relu=True):
        super(GhostModule, self).__init__()
        self.oup = oup
        self.relu = relu
        init_channels = math.ceil(oup / ratio)
        new_channels = init_channels * (ratio - 1)
        self.primary_conv = nn.Sequential(
            nn.Conv3d(inp, init_channels, kernel_size=(1, kernel_size, kernel_size), stride=(1, stride, stride),
                      padding=(0, kernel_size // 2, kernel_size // 2), bias=False),
            nn.BatchNorm3d(init_channels),
            nn.ReLU(inplace=True) if relu else nn.Sequential(),
        )
        self.cheap_operation = nn.Sequential(
            nn.Conv3d(init_channels, new_channels, kernel_size=dw_size, stride=1, padding=dw_size // 2,
                      groups=init_channels, bias=False),
            nn.BatchNorm3d(new_channels),
            nn.ReLU(inplace=True) if relu else nn.Sequential(),
        )

    def forward(self, x, res=None, reserve=(0, 0)):
        pc, co = self.primary_conv, self.cheap_operation
        init = pc[0].out_channels
        keep = self.oup - init  # channels of the cheap half that survive the [:oup] slice
        out = sfhip.new_act(x, x.N, x.T, x.H, x.W, self.oup, reserve[0], reserve[1])
        training = pc[1].training
        if res is None:
            x1 = engine.conv_bn_act(x, pc[0], pc[1], relu=self.relu, out=out.slice(0, init))
            engine.conv_bn_act(x1, co[0], co[1], relu=self.relu, out=out.slice(init, keep), cout=keep)
        elif training:
            # x += shortcut(residual): the cheap half must see x1 BEFORE the add
            x1 = engine.conv_bn_act(x, pc[0], pc[1], relu=self.relu)
            engine.conv_bn_act(x1, co[0], co[1], relu=self.relu, res=res.slice(init, keep),
                               out=out.slice(init, keep), cout=keep)
            engine.add_into(x1, res.slice(0, init), out.slice(0, init))
        else:
            # x += shortcut(residual) (ghostnet_helper.py:162) folded in: the cheap half must see x1 BEFORE
            # the add, so x1 goes to scratch, the depthwise epilogue adds its residual slice, and a 1-tap
            # identity depthwise pass writes x1 + residual into the first half.
            x1 = engine.conv_bn_act(x, pc[0], pc[1], relu=self.relu)
            engine.conv_bn_act(x1, co[0], co[1], relu=self.relu, res=res.slice(init, keep),
                               out=out.slice(init, keep), cout=keep)
            ones = engine._cached(self, "_sf_ones", (init, str(x.buf.device)),
                                  lambda: torch.ones((1, init), dtype=torch.float32, device=x.buf.device))
            sfhip.dwconv(x1, ones, (1, 1, 1), res=res.slice(0, init), out=out.slice(0, init))
        return out


class GhostBottleneck(nn.Module):
    """ghost1 -> [dw stride conv + BN] -> [SE] -> ghost2 (+ shortcut) (ghostnet_helper.py:102-163)."""

    def __init__(self, in_chs, mid_chs, out_chs, dw_kernel_size=3, stride=1, se_ratio=0.):
        super(GhostBottleneck, self).__init__()
        has_se = se_ratio is not None and se_ratio > 0.
        self.stride = stride
        self.ghost1 = GhostModule(in_chs, mid_chs, relu=True)
        if self.stride > 1:
            self.conv_dw = nn.Conv3d(mid_chs, mid_chs, kernel_size=(1, dw_kernel_size, dw_kernel_size),
                                     stride=(1, stride, stride),
                                     padding=(0, (dw_kernel_size - 1) // 2, (dw_kernel_size - 1) // 2),
                                     groups=mid_chs, bias=False)
            self.bn_dw = nn.BatchNorm3d(mid_chs)
        self.se = SqueezeExcite(mid_chs, se_ratio=se_ratio) if has_se else None
        self.ghost2 = GhostModule(mid_chs, out_chs, relu=False)
        if in_chs == out_chs and self.stride == 1:
            self.shortcut = nn.Sequential()
        else:
            self.shortcut = nn.Sequential(
                nn.Conv3d(in_chs, in_chs, kernel_size=(1, dw_kernel_size, dw_kernel_size), stride=(1, stride, stride),
                          padding=(0, (dw_kernel_size - 1) // 2, (dw_kernel_size - 1) // 2), groups=in_chs,
                          bias=False),
                nn.BatchNorm3d(in_chs),
                nn.Conv3d(in_chs, out_chs, 1, stride=1, padding=0, bias=False),
                nn.BatchNorm3d(out_chs),
            )

    def forward(self, x, reserve=(0, 0)):
        y = self.ghost1(x)
        if self.stride > 1:
            y = engine.conv_bn_act(y, self.conv_dw, self.bn_dw, relu=False)
        if self.se is not None:
            y = self.se(y)
        if len(self.shortcut) == 0:
            res = x
        else:
            sc = self.shortcut
            r = engine.conv_bn_act(x, sc[0], sc[1], relu=False)
            res = engine.conv_bn_act(r, sc[2], sc[3], relu=False)
        return self.ghost2(y, res=res, reserve=reserve)


class GhostNet_Inverted_Residual_Block(nn.Module):
    """A list of GhostBottlenecks from [k, exp, c, se, s] rows; channel counts re-rounded with divisor 2
    (ghostnet_helper.py:269-312)."""

    def __init__(self, input_channel, cfg):
        super(GhostNet_Inverted_Residual_Block, self).__init__()
        layers = []
        for k, exp_size, c, se_ratio, s in cfg:
            output_channel = _make_divisible(c, 2)
            hidden_channel = _make_divisible(exp_size, 2)
            layers.append(GhostBottleneck(input_channel, hidden_channel, output_channel, dw_kernel_size=k,
                                          stride=s, se_ratio=se_ratio))
            input_channel = output_channel
        self.features = nn.Sequential(*layers)
        _efficient_init(self)

    def forward(self, x, reserve=(0, 0)):
        n = len(self.features)
        for i, blk in enumerate(self.features):
            x = blk(x, reserve if i == n - 1 else (0, 0))
        return x


class GhostNet_Stage(nn.Module):
    """children pathway{p}_channel_{C_out} (ghostnet_helper.py:315-380)."""

    def __init__(self, input_channel, slow_cfg, fast_cfg):
        super(GhostNet_Stage, self).__init__()
        self.slow_cfg, self.fast_cfg = slow_cfg, fast_cfg
        self.num_pathways = len(input_channel)
        self._names = []
        for pathway in range(self.num_pathways):
            cfg = slow_cfg if pathway == 0 else fast_cfg
            block = GhostNet_Inverted_Residual_Block(input_channel=input_channel[pathway], cfg=cfg)
            name = "pathway{}_channel_{}".format(pathway, cfg[-1][2])
            self.add_module(name, block)
            self._names.append(name)
            _efficient_init(self)

    def forward(self, inputs, reserve=None):
        xs = engine.enter(inputs)
        with engine.internal():
            out = engine.run_paths(
                [lambda p=p: getattr(self, self._names[p])(xs[p], reserve[p] if reserve else (0, 0))
                 for p in range(self.num_pathways)], xs[0].buf.device)
        return engine.leave(out)
