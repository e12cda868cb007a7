"""Stems (reference stem_helper.py).  Parameters live in plain nn.Conv3d / nn.BatchNorm3d children with
the reference's names; forward is conv+BN+ReLU in one MFMA kernel plus one pooling kernel."""
import torch.nn as nn

import sfhip
from . import engine


class ResNetBasicStem(nn.Module):
    """conv [kT,7,7]/s[1,2,2] -> BN -> ReLU -> MaxPool [1,3,3]/s[1,2,2]/p[0,1,1] (stem_helper.py:102-178)."""

    def __init__(self, dim_in, dim_out, kernel, stride, padding, inplace_relu=True, eps=1e-5, bn_mmt=0.1,
                 norm_module=nn.BatchNorm3d):
        super(ResNetBasicStem, self).__init__()
        self.kernel, self.stride, self.padding = kernel, stride, padding
        self.inplace_relu, self.eps, self.bn_mmt = inplace_relu, eps, bn_mmt
        self.conv = nn.Conv3d(dim_in, dim_out, self.kernel, stride=self.stride, padding=self.padding, bias=False)
        self.bn = norm_module(num_features=dim_out, eps=self.eps, momentum=self.bn_mmt)
        self.relu = nn.ReLU(self.inplace_relu)
        self.pool_layer = nn.MaxPool3d(kernel_size=[1, 3, 3], stride=[1, 2, 2], padding=[0, 1, 1])

    def forward(self, x, reserve=(0, 0)):
        y = engine.stem_conv_bn_relu(x, self.conv, self.bn, relu=True)
        return engine.maxpool(y, (1, 3, 3), (1, 2, 2), (0, 1, 1), out_reserve=reserve)


class VideoModelStem(nn.Module):
    """One ResNetBasicStem per pathway, children named pathway{p}_stem (stem_helper.py:9-99)."""

    def __init__(self, dim_in, dim_out, kernel, stride, padding, inplace_relu=True, eps=1e-5, bn_mmt=0.1,
                 norm_module=nn.BatchNorm3d):
        super(VideoModelStem, self).__init__()
        assert len({len(dim_in), len(dim_out), len(kernel), len(stride), len(padding)}) == 1, \
            "Input pathway dimensions are not consistent."
        self.num_pathways = len(dim_in)
        self.kernel, self.stride, self.padding = kernel, stride, padding
        self.inplace_relu, self.eps, self.bn_mmt = inplace_relu, eps, bn_mmt
        for pathway in range(len(dim_in)):
            stem = ResNetBasicStem(dim_in[pathway], dim_out[pathway], self.kernel[pathway], self.stride[pathway],
                                   self.padding[pathway], self.inplace_relu, self.eps, self.bn_mmt, norm_module)
            self.add_module("pathway{}_stem".format(pathway), stem)

    def forward(self, x, reserve=None):
        assert len(x) == self.num_pathways, "Input tensor does not contain {} pathway".format(self.num_pathways)
        dev = x[0].buf.device if hasattr(x[0], "buf") else x[0].device
        acts = engine.run_paths(
            [lambda p=p: getattr(self, "pathway{}_stem".format(p))(x[p], reserve[p] if reserve else (0, 0))
             for p in range(len(x))], dev)
        out = engine.leave(acts)
        for pathway in range(len(x)):  # the reference mutates the caller's list in place (:96-98)
            x[pathway] = out[pathway]
        return x


class SimpleStem(nn.Module):
    """Efficient-backbone stems held in an nn.Sequential-compatible layout: children '0' conv, '1' BN,
    '2' ReLU (, '3' MaxPool3d(3,(1,2,2),1)) — shufflenetv2_stem (stem_helper.py:237-245) and the GhostNet
    stem (stem_helper.py:318-327) share it."""

    def __init__(self, img_dim, dim_out, with_pool, relu6=False):
        super(SimpleStem, self).__init__()
        self.act = 6 if relu6 else True
        self.add_module("0", nn.Conv3d(img_dim, dim_out, kernel_size=3, stride=(1, 2, 2), padding=(1, 1, 1),
                                       bias=False))
        self.add_module("1", nn.BatchNorm3d(dim_out))
        self.add_module("2", nn.ReLU6(inplace=True) if relu6 else nn.ReLU(inplace=True))
        if with_pool:
            self.add_module("3", nn.MaxPool3d(kernel_size=3, stride=(1, 2, 2), padding=1))
        self.with_pool = with_pool

    def forward(self, x, reserve=(0, 0)):
        conv, bn = self._modules["0"], self._modules["1"]
        if not self.with_pool:
            y = engine.stem_conv_bn_relu(x, conv, bn, relu=self.act)
            if reserve != (0, 0):
                wide = sfhip.new_act(y, y.N, y.T, y.H, y.W, y.C, reserve[0], reserve[1])
                return engine.copy_channels(y, wide)
            return y
        y = engine.stem_conv_bn_relu(x, conv, bn, relu=self.act)
        return engine.maxpool(y, (3, 3, 3), (1, 2, 2), (1, 1, 1), out_reserve=reserve)


class _EfficientStem(nn.Module):
    def __init__(self, input_channels, img_dim, with_pool):
        super(_EfficientStem, self).__init__()
        self.num_pathways = len(input_channels)
        for pathway in range(self.num_pathways):
            self.add_module("pathway{}_stem".format(pathway), SimpleStem(img_dim, input_channels[pathway], with_pool))

    def forward(self, x, reserve=None):
        assert len(x) == self.num_pathways, "Input tensor does not contain {} pathway".format(self.num_pathways)
        dev = x[0].buf.device if hasattr(x[0], "buf") else x[0].device
        acts = engine.run_paths(
            [lambda p=p: getattr(self, "pathway{}_stem".format(p))(x[p], reserve[p] if reserve else (0, 0))
             for p in range(len(x))], dev)
        out = engine.leave(acts)
        for pathway in range(len(x)):
            x[pathway] = out[pathway]
        return x


class ShuffleNetV2_Model_Stem(_EfficientStem):
    """stem_helper.py:248-270."""

    def __init__(self, input_channels=[32], sample_size=224, width_mult=1., img_dim=3):
        super(ShuffleNetV2_Model_Stem, self).__init__(input_channels, img_dim, True)


class ShuffleNet_Model_Stem(_EfficientStem):
    """stem_helper.py:274-306 (conv 3x3x3 /(1,2,2) + BN + ReLU + MaxPool3d 3 /(1,2,2); prints its arguments as the
    reference does)."""

    def __init__(self, input_channels=[32], sample_size=224, img_dim=3):
        for c in input_channels:
            print(img_dim, c, (1, 2, 2))
        super(ShuffleNet_Model_Stem, self).__init__(input_channels, img_dim, True)


class GhostNet_Model_Stem(_EfficientStem):
    """stem_helper.py:310-336 (no max-pool: s1_fuse sees S/2)."""

    def __init__(self, input_channels=[32, ], sample_size=224, img_dim=3):
        super(GhostNet_Model_Stem, self).__init__(input_channels, img_dim, False)


class MobilenetV2_Basic_Stem(nn.Module):
    """features = conv 3x3x3 /(1,2,2) -> BN -> ReLU6 with int(input_channel * width_mult) filters
    (stem_helper.py:181-200)."""

    def __init__(self, input_channel=32, sample_size=224, width_mult=1., img_dim=3):
        super(MobilenetV2_Basic_Stem, self).__init__()
        assert sample_size % 16 == 0.
        self.features = SimpleStem(img_dim, int(input_channel * width_mult), False, relu6=True)

    def forward(self, x, reserve=(0, 0)):
        return self.features(x, reserve)


class MobilenetV2_Model_Stem(nn.Module):
    """stem_helper.py:202-232."""

    def __init__(self, input_channels=[32], sample_size=224, width_mult=[1.], img_dim=3):
        super(MobilenetV2_Model_Stem, self).__init__()
        if len(input_channels) != len(width_mult):
            width_mult = width_mult * len(input_channels)
        self.num_pathways = len(input_channels)
        for pathway in range(self.num_pathways):
            self.add_module("pathway{}_stem".format(pathway),
                            MobilenetV2_Basic_Stem(input_channels[pathway], sample_size, width_mult[pathway], img_dim))

    forward = _EfficientStem.forward
