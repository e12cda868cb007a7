"""ShuffleNet v1 stages (reference shufflenet_helper.py).  SLOWFAST.GROUPS = 1 (the shipped YAML): the 1x1 convs are
plain GEMMs and channel_shuffle(x, 1) is the identity.  GROUPS > 1 (the published w2.0 / g3 model, README.md:260,
wdf_all_run_scripts/run_shufflenet_w2_g3.sh): the 1x1 convs are block-diagonal GEMMs — ONE launch per layer, the group on
the grid's z index, on channel windows of the same buffers (engine.grouped_conv) — and channel_shuffle is index math of
conv1's stores in eval mode, one sf_channel_shuffle launch behind the batch-statistics BN in training mode.  The stride-2 shortcut
conv1x1 -> AvgPool3d((1,3,3),(1,2,2),(0,1,1)) is evaluated as pool -> conv1x1 (both linear, no bias: same
result, a quarter of the GEMM), and its ReLU and the concat are the GEMM's epilogue and store slice."""
import torch.nn as nn

import sfhip
from . import engine
from .shufflenetv2_helper import _efficient_init


class Bottleneck(nn.Module):
    """grouped 1x1 + BN + ReLU -> shuffle -> dw 3x3x3 + BN -> grouped 1x1 + BN; stride 2: relu(cat[out,
    shortcut(x)]), stride 1: relu(out + x) (shufflenet_helper.py:32-79)."""

    def __init__(self, in_planes, out_planes, stride, groups):
        super(Bottleneck, self).__init__()
        self.stride = stride
        self.groups = groups
        mid_planes = out_planes // 4
        if self.stride == 2:
            mid_planes = out_planes // 2
            out_planes = out_planes - out_planes // 2
        g = 1 if in_planes == 24 else groups
        self.conv1 = nn.Conv3d(in_planes, mid_planes, kernel_size=1, groups=g, bias=False)
        self.bn1 = nn.BatchNorm3d(mid_planes)
        self.conv2 = nn.Conv3d(mid_planes, mid_planes, kernel_size=(3, 3, 3), stride=(1, stride, stride), padding=1,
                               groups=mid_planes, bias=False)
        self.bn2 = nn.BatchNorm3d(mid_planes)
        self.conv3 = nn.Conv3d(mid_planes, out_planes, kernel_size=1, groups=groups, bias=False)
        self.bn3 = nn.BatchNorm3d(out_planes)
        self.relu = nn.ReLU(inplace=True)
        if stride == 2:
            self.shortcut = nn.Sequential(
                nn.Conv3d(in_planes, mid_planes, kernel_size=1, bias=False),
                nn.AvgPool3d(kernel_size=(1, 3, 3), stride=(1, 2, 2), padding=(0, 1, 1)),
            )

    def forward(self, x, reserve=(0, 0)):
        # conv1 runs with g = 1 behind a 24-channel input but the shuffle always uses self.groups
        # (shufflenet_helper.py:47-51, 73): a grouped conv1 stores shuffled, a dense one is shuffled by copies
        if self.conv1.groups == self.groups or self.groups == 1:
            y = engine.conv_bn_act(x, self.conv1, self.bn1, relu=True, shuffle=self.groups > 1)
        else:
            y = engine.channel_shuffle(engine.conv_bn_act(x, self.conv1, self.bn1, relu=True), self.groups)
        y = engine.conv_bn_act(y, self.conv2, self.bn2, relu=False)
        if self.stride != 2:
            return engine.conv_bn_act(y, self.conv3, self.bn3, relu=True, res=x, out_reserve=reserve)
        c3, cs = self.conv3.out_channels, self.shortcut[0].out_channels
        out = sfhip.new_act(y, y.N, y.T, y.H, y.W, c3 + cs, reserve[0], reserve[1])
        engine.conv_bn_act(y, self.conv3, self.bn3, relu=True, out=out.slice(0, c3))
        pool = self.shortcut[1]
        pooled = engine.avgpool(x, tuple(pool.kernel_size), tuple(pool.stride), tuple(pool.padding))
        engine.conv_bn_act(pooled, self.shortcut[0], None, relu=True, out=out.slice(c3, cs))
        return out


class ShuffleNet_Residual_Block(nn.Module):
    """num_block Bottlenecks, the first with stride 2 (shufflenet_helper.py:171-212)."""

    def __init__(self, in_plane, out_plane, num_block, group):
        super(ShuffleNet_Residual_Block, self).__init__()
        self.in_planes = in_plane
        layers = []
        for i in range(num_block):
            layers.append(Bottleneck(self.in_planes, out_plane, stride=2 if i == 0 else 1, groups=group))
            self.in_planes = out_plane
        self.features = nn.Sequential(*layers)
        _efficient_init(self)

    def forward(self, x, reserve=(0, 0)):
        n = len(self.features)
        for i, blk in enumerate(self.features):
            x = blk(x, reserve if i == n - 1 else (0, 0))
        return x


class ShuffleNet_Stage(nn.Module):
    """children pathway{p}_channel_{C_out} (shufflenet_helper.py:214-270)."""

    def __init__(self, input_channel, slow_stage_out_channels, fast_stage_out_channels, num_block, group):
        super(ShuffleNet_Stage, self).__init__()
        self.slow_stage_out_channels = slow_stage_out_channels
        self.fast_stage_out_channels = fast_stage_out_channels
        self.num_pathways = len(input_channel)
        self._names = []
        for pathway in range(self.num_pathways):
            out_plane = slow_stage_out_channels if pathway == 0 else fast_stage_out_channels
            block = ShuffleNet_Residual_Block(in_plane=input_channel[pathway], out_plane=out_plane,
                                              num_block=num_block, group=group)
            name = "pathway{}_channel_{}".format(pathway, out_plane)
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
