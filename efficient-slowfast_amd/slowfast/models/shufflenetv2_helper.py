"""ShuffleNetV2 stages (reference shufflenetv2_helper.py).  The channel shuffle (g=2) is never executed as
a pass: every producer of a block's output writes channel c of branch g to position 2*c + g directly
(out_cmul = 2), which is exactly view(N,2,C/2,..).permute(0,2,1,..) (shufflenetv2_helper.py:32-43)."""
import math

import torch.nn as nn

import sfhip
from . import engine


class InvertedResidual(nn.Module):
    """stride 1: split | pw-dw-pw, cat, shuffle.  stride 2: (dw-pw) || (pw-dw-pw), cat, shuffle
    (shufflenetv2_helper.py:46-112; 'banch' is the reference's attribute spelling = state_dict keys)."""

    def __init__(self, inp, oup, stride):
        super(InvertedResidual, self).__init__()
        self.stride = stride
        assert stride in [1, 2]
        oup_inc = oup // 2
        if self.stride == 1:
            self.banch2 = nn.Sequential(
                nn.Conv3d(oup_inc, oup_inc, 1, 1, 0, bias=False), nn.BatchNorm3d(oup_inc), nn.ReLU(inplace=True),
                nn.Conv3d(oup_inc, oup_inc, 3, (1, stride, stride), 1, groups=oup_inc, bias=False),
                nn.BatchNorm3d(oup_inc),
                nn.Conv3d(oup_inc, oup_inc, 1, 1, 0, bias=False), nn.BatchNorm3d(oup_inc), nn.ReLU(inplace=True),
            )
        else:
            self.banch1 = nn.Sequential(
                nn.Conv3d(inp, inp, 3, (1, stride, stride), 1, groups=inp, bias=False), nn.BatchNorm3d(inp),
                nn.Conv3d(inp, oup_inc, 1, 1, 0, bias=False), nn.BatchNorm3d(oup_inc), nn.ReLU(inplace=True),
            )
            self.banch2 = nn.Sequential(
                nn.Conv3d(inp, oup_inc, 1, 1, 0, bias=False), nn.BatchNorm3d(oup_inc), nn.ReLU(inplace=True),
                nn.Conv3d(oup_inc, oup_inc, 3, (1, stride, stride), 1, groups=oup_inc, bias=False),
                nn.BatchNorm3d(oup_inc),
                nn.Conv3d(oup_inc, oup_inc, 1, 1, 0, bias=False), nn.BatchNorm3d(oup_inc), nn.ReLU(inplace=True),
            )
        self.oup = oup

    def forward(self, x, reserve=(0, 0)):
        b2 = self.banch2
        if self.stride == 1:
            half = x.C // 2
            x1, x2 = x.slice(0, half), x.slice(half, x.C - half)
        else:
            x2 = x
        y = engine.conv_bn_act(x2, b2[0], b2[1], relu=True)
        y = engine.conv_bn_act(y, b2[3], b2[4], relu=False)
        c2 = b2[5].out_channels
        c1 = x1.C if self.stride == 1 else self.banch1[2].out_channels
        assert c1 == c2, "channel shuffle needs equal halves"
        out = sfhip.new_act(y, y.N, y.T, y.H, y.W, c1 + c2, reserve[0], reserve[1])
        # shuffled positions: first half (g=0) -> even channels, second half (g=1) -> odd channels
        even = sfhip.Act(out.buf, out.coff, out.C)
        odd = sfhip.Act(out.buf, out.coff + 1, out.C - 1)
        engine.conv_bn_act(y, b2[5], b2[6], relu=True, out=odd, out_cmul=2)
        if self.stride == 1:
            engine.copy_channels(x1, even, out_cmul=2)
        else:
            b1 = self.banch1
            z = engine.conv_bn_act(x, b1[0], b1[1], relu=False)
            engine.conv_bn_act(z, b1[2], b1[3], relu=True, out=even, out_cmul=2)
        return out


def _efficient_init(module):
    """The efficient stages' own init (shufflenetv2_helper.py:204-219 / ghostnet_helper.py:298-312); it is
    overwritten by init_weights afterwards but consumes RNG, so it is reproduced for seed parity."""
    for m in module.modules():
        if isinstance(m, nn.Conv3d):
            n = m.kernel_size[0] * m.kernel_size[1] * m.kernel_size[2] * m.out_channels
            m.weight.data.normal_(0, math.sqrt(2. / n))
            if m.bias is not None:
                m.bias.data.zero_()
        elif isinstance(m, nn.BatchNorm3d):
            m.weight.data.fill_(1)
            m.bias.data.zero_()
        elif isinstance(m, nn.Linear):
            m.weight.data.normal_(0, 0.01)
            m.bias.data.zero_()


class ShuffleNetV2_Inverted_Residual_Block(nn.Module):
    """stage_repeats [4, 8, 4] InvertedResiduals, first one stride 2 (shufflenetv2_helper.py:184-219)."""

    def __init__(self, input_channel, idxstage, stage_out_channels):
        super(ShuffleNetV2_Inverted_Residual_Block, self).__init__()
        self.stage_repeats = [4, 8, 4]
        feats = []
        output_channel = stage_out_channels[idxstage + 2]
        for i in range(self.stage_repeats[idxstage]):
            feats.append(InvertedResidual(input_channel, output_channel, 2 if i == 0 else 1))
            input_channel = output_channel
        self.features = nn.Sequential(*feats)
        _efficient_init(self)

    def forward(self, x, reserve=(0, 0)):
        n = len(self.features)
        for i, blk in enumerate(self.features):
            x = blk(x, reserve if i == n - 1 else (0, 0))
        return x


class ShuffleNetV2_Stage(nn.Module):
    """children pathway{p}_channel_{C} (shufflenetv2_helper.py:222-297)."""

    def __init__(self, input_channel, idxstage, slow_stage_out_channels, fast_stage_out_channels):
        super(ShuffleNetV2_Stage, self).__init__()
        self.slow_stage_out_channels = slow_stage_out_channels
        self.fast_stage_out_channels = fast_stage_out_channels
        self.idxstage = idxstage
        self.num_pathways = len(input_channel)
        self._names = []
        for pathway in range(self.num_pathways):
            chans = slow_stage_out_channels if pathway == 0 else fast_stage_out_channels
            block = ShuffleNetV2_Inverted_Residual_Block(input_channel[pathway], idxstage=idxstage,
                                                         stage_out_channels=chans)
            name = "pathway{}_channel_{}".format(pathway, chans[idxstage + 2])
            self.add_module(name, block)
            self._names.append(name)
            _efficient_init(self)  # the reference re-initialises the whole stage after adding each pathway

    def forward(self, inputs, reserve=None):
        xs = engine.enter(inputs)
        with engine.internal():
            out = engine.run_paths(
                [lambda p=p: getattr(self, self._names[p])(xs[p], reserve[p] if reserve else (0, 0))
                 for p in range(self.num_pathways)], xs[0].buf.device)
        return engine.leave(out)
