"""MobileNetV2 stages (reference mobilenetv2_helper.py).  An InvertedResidual is three launches in eval
mode -- 1x1x1 expand GEMM (+BN+ReLU6 in the epilogue), depthwise 3x3x3 (+BN+ReLU6), 1x1x1 project GEMM
(+BN, + the identity shortcut in the epilogue) -- and the last block of a stage stores straight into the
channel slice the following CMDA fusion reserved."""
import torch.nn as nn

from . import engine
from .shufflenetv2_helper import _efficient_init


class InvertedResidual(nn.Module):
    """expand_ratio 1: dw -> BN -> ReLU6 -> pw-linear -> BN; otherwise pw -> BN -> ReLU6 first; identity
    shortcut iff stride (1,1,1) and inp == oup (mobilenetv2_helper.py:30-68)."""

    def __init__(self, inp, oup, stride, expand_ratio):
        super(InvertedResidual, self).__init__()
        self.stride = stride
        hidden_dim = round(inp * expand_ratio)
        self.use_res_connect = self.stride == (1, 1, 1) and inp == oup
        if expand_ratio == 1:
            self.conv = nn.Sequential(
                nn.Conv3d(hidden_dim, hidden_dim, 3, stride, 1, groups=hidden_dim, bias=False),
                nn.BatchNorm3d(hidden_dim),
                nn.ReLU6(inplace=True),
                nn.Conv3d(hidden_dim, oup, 1, 1, 0, bias=False),
                nn.BatchNorm3d(oup),
            )
        else:
            self.conv = nn.Sequential(
                nn.Conv3d(inp, hidden_dim, 1, 1, 0, bias=False),
                nn.BatchNorm3d(hidden_dim),
                nn.ReLU6(inplace=True),
                nn.Conv3d(hidden_dim, hidden_dim, 3, stride, 1, groups=hidden_dim, bias=False),
                nn.BatchNorm3d(hidden_dim),
                nn.ReLU6(inplace=True),
                nn.Conv3d(hidden_dim, oup, 1, 1, 0, bias=False),
                nn.BatchNorm3d(oup),
            )

    def forward(self, x, reserve=(0, 0)):
        c = self.conv
        y, i = x, 0
        if len(c) == 8:
            y, i = engine.conv_bn_act(x, c[0], c[1], relu=6), 3
        y = engine.conv_bn_act(y, c[i], c[i + 1], relu=6)
        return engine.conv_bn_act(y, c[i + 3], c[i + 4], relu=False, res=x if self.use_res_connect else None,
                                  out_reserve=reserve)


class MobileV2_Inverted_Residual_Block(nn.Module):
    """The InvertedResiduals of one or more [t, c, n, s] rows; widths int(c*w) on the slow pathway and
    int(c*w // beta_inv) on the fast one (mobilenetv2_helper.py:70-103)."""

    def __init__(self, input_channel, interverted_residual_setting, width_mult, beta_inv=None):
        super(MobileV2_Inverted_Residual_Block, self).__init__()
        rows = interverted_residual_setting
        if not isinstance(rows[0], list):
            rows = [rows]
        feats = []
        for t, c, n, s in rows:
            output_channel = int(c * width_mult) if beta_inv is None else int(c * width_mult // beta_inv)
            for i in range(n):
                feats.append(InvertedResidual(input_channel, output_channel, tuple(s) if i == 0 else (1, 1, 1),
                                              expand_ratio=t))
                input_channel = output_channel
        self.features = nn.Sequential(*feats)

    def forward(self, x, reserve=(0, 0)):
        n = len(self.features)
        for i, blk in enumerate(self.features):
            x = blk(x, reserve if i == n - 1 else (0, 0))
        return x


class MobileNetV2_Stage(nn.Module):
    """children pathway{p}_channel_{c of the first row} (mobilenetv2_helper.py:257-328)."""

    def __init__(self, input_channel, slow_residual_setting, fast_residual_setting=None, width_mult=1.,
                 beta_inv=4):
        super(MobileNetV2_Stage, self).__init__()
        assert isinstance(slow_residual_setting, list) and isinstance(fast_residual_setting, list)
        self.slow_residual_setting = slow_residual_setting
        self.fast_residual_setting = fast_residual_setting
        self.width_mult = width_mult
        self.num_pathways = len(input_channel)
        if self.num_pathways > 2:
            raise Exception("Only support 1 or 2 pathways")
        self._names = []
        for pathway in range(self.num_pathways):
            rows = slow_residual_setting if pathway == 0 else fast_residual_setting
            block = MobileV2_Inverted_Residual_Block(input_channel[pathway], rows, width_mult,
                                                     beta_inv=None if pathway == 0 else beta_inv)
            name = "pathway{}_channel_{}".format(pathway, rows[0][1])
            self.add_module(name, block)
            self._names.append(name)
            _efficient_init(self)  # the reference re-initialises the stage after adding each pathway (RNG parity)

    def forward(self, inputs, reserve=None):
        xs = engine.enter(inputs)
        with engine.internal():
            out = engine.run_paths(
                [lambda p=p: getattr(self, self._names[p])(xs[p], reserve[p] if reserve else (0, 0))
                 for p in range(self.num_pathways)], xs[0].buf.device)
        return engine.leave(out)
