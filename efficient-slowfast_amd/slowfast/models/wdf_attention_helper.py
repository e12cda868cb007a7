"""CMDA's two attention halves as parameter containers (reference wdf_attention_helper.py:13-91).
They are executed by FuseFastAndSlow as fused kernels; called on their own (NCTHW tensor in/out) they run
the same kernels un-fused."""
import torch
import torch.nn as nn

import sfhip
from . import engine


class SpatialAttention(nn.Module):
    """Full softmax self-attention over N = T*H*W positions, no 1/sqrt(d) scaling, q/k/v 1x1x1 convs with
    bias, out = gamma * attn(x) + x with gamma initialised to 0 (wdf_attention_helper.py:17-54)."""

    def __init__(self, channel, reduction=8):
        super(SpatialAttention, self).__init__()
        self.input_channel = channel
        self.query_conv = nn.Conv3d(in_channels=channel, out_channels=channel // reduction, kernel_size=1)
        self.key_conv = nn.Conv3d(in_channels=channel, out_channels=channel // reduction, kernel_size=1)
        self.value_conv = nn.Conv3d(in_channels=channel, out_channels=channel, kernel_size=1)
        self.gamma = nn.Parameter(torch.zeros(1))
        self.softmax = nn.Softmax(dim=-1)

    def qkv(self, x):
        """One pointwise GEMM producing [q | k | v] rows: [N,T,H,W, 2*C/r + C]."""
        if self.query_conv.out_channels != self.value_conv.out_channels:
            raise NotImplementedError("SpatialAttention(reduction != 1) is never instantiated by the reference "
                                      "models and is not on the HIP path")
        convs = (self.query_conv, self.key_conv, self.value_conv)

        def make():
            w = torch.cat([c.weight for c in convs], 0)
            b = torch.cat([c.bias for c in convs], 0)
            return sfhip.pack_conv_weight(w), b.contiguous()

        wp, b = engine._cached(self, "_sf_qkv", engine._key(*[t for c in convs for t in (c.weight, c.bias)]), make)
        return sfhip.conv(x, wp, (1, 1, 1), bias=b)

    def run(self, x, scale=None, bias=None, relu=False, alpha=1, out=None):
        c = self.input_channel
        qkv = self.qkv(x)
        return sfhip.attention(qkv.slice(0, c), qkv.slice(c, c), qkv.slice(2 * c, c), x, self.gamma,
                               scale=scale, bias=bias, relu=relu, alpha=alpha, out=out)

    def forward(self, x):
        plain = not isinstance(x, engine.Act)
        a = engine.enter([x])[0]
        y = self.run(a)
        return sfhip.to_ncthw(y) if plain else y


class ECA(nn.Module):
    """Efficient channel attention: global avg-pool -> Conv1d(1,1,k=3,pad=1,bias=False) ALONG THE CHANNEL
    AXIS -> sigmoid -> broadcast multiply (wdf_attention_helper.py:57-91)."""

    def __init__(self, channel, k_size=3):
        super(ECA, self).__init__()
        if k_size != 3:
            raise NotImplementedError("ECA k_size != 3 is never used by the reference models")
        self.avg_pool = nn.AdaptiveAvgPool3d(1)
        self.conv = nn.Conv1d(1, 1, kernel_size=k_size, padding=(k_size - 1) // 2, bias=False)
        self.sigmoid = nn.Sigmoid()

    def run(self, x, alpha=1, scale=None, bias=None, relu=False, out=None):
        """gate(max-pool_alpha(x)) [* BN affine, ReLU], written into `out`."""
        pooled = sfhip.tmax_mean(x, alpha)
        return sfhip.gate_apply(x, alpha, pooled, w3=self.conv.weight, scale=scale, bias=bias, relu=relu, out=out)

    def forward(self, x):
        plain = not isinstance(x, engine.Act)
        a = engine.enter([x])[0]
        y = self.run(a)
        return sfhip.to_ncthw(y) if plain else y
