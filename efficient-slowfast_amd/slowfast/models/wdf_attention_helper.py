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
        """One pointwise GEMM producing [q | k | v] rows: [N,T,H,W, 3*C].  With reduction r > 1 (the reference class's
        default is 8; no shipped model uses it) q and k have C/r channels: their rows of the merged weight are padded
        with ZERO rows up to C, so the padded channels of q / k are exactly zero, q k^T is unchanged and the flash
        kernels run as for r = 1 (correct, not tuned: the score product is taken over C instead of C/r channels)."""
        c = self.input_channel
        cr = self.query_conv.out_channels
        convs = (self.query_conv, self.key_conv, self.value_conv)

        def padded(t, rows):
            if t.shape[0] == rows:
                return t
            return torch.cat([t, t.new_zeros((rows - t.shape[0],) + tuple(t.shape[1:]))], 0)

        def make():  # (forward pack, bias, data-gradient pack) of the merged projection, once per parameter version
            w = torch.cat([padded(cv.weight, c) for cv in convs], 0)
            b = torch.cat([padded(cv.bias, c) for cv in convs], 0)
            if w.is_cuda:  # both layouts by one launch
                wp_, wtp_ = sfhip.pack_conv_weight_pair(w.detach())
                return wp_, b.contiguous(), wtp_
            return sfhip.pack_conv_weight(w), b.contiguous(), None

        wp, b, wtp = engine._cached(self, "_sf_qkv", engine._key(*[t for cv in convs for t in (cv.weight, cv.bias)]),
                                    make)
        qkv = sfhip.conv(x, wp, (1, 1, 1), bias=b)
        t = engine.tape()
        if t is not None:
            nout = (cr, cr, c)

            def bwd():  # merged q|k|v projection: one wgrad / dgrad, then split per conv
                g = t.grad_of(qkv)
                dwp = sfhip.conv_wgrad(x, g, 3 * c, (1, 1, 1), cin_pad=wp.shape[2])
                dw = sfhip.unpack_conv_weight_grad(dwp, (3 * c, c, 1, 1, 1))
                db = engine._colsum(g)
                t.add_pgrads([cv.weight for cv in convs] + [cv.bias for cv in convs],
                             [dw[i * c:i * c + nout[i]] for i in range(3)] + [db[i * c:i * c + nout[i]] for i in range(3)])
                sfhip.conv_dgrad(g, wtp, x, (1, 1, 1), out=t.grad_of(x), accumulate=True)

            t.record(bwd)
        return qkv

    def _run_wide(self, x, scale, bias, relu, alpha, out):
        """Heads wider than the flash kernels' 128 channels (SlowFastShuffleNet w2.0 / g3: C = 240 at s4_fuse, where
        N = T*H*W <= 64): three dense 1x1x1 projections and the materialised-score attention of the Nonlocal block
        (nonlocal_helper.dense_attention, no 1/sqrt(d)), then z = gamma * O + x (and the eval-mode BN affine / ReLU /
        nearest T-upsample as one more small pass)."""
        from .nonlocal_helper import dense_attention
        c = self.input_channel
        q = engine.conv_bn_act(x, self.query_conv)
        k = engine.conv_bn_act(x, self.key_conv)
        v = engine.conv_bn_act(x, self.value_conv)
        o = dense_attention(q, k, v, softmax=True, sm_scale=1.0)
        gvec = self.gamma.detach().expand(c).contiguous()
        zero = torch.zeros_like(gvec)  # sf_affine_fwd takes scale and bias together
        t = engine.tape()
        if t is None:
            plain = scale is None and bias is None and not relu and alpha == 1
            z = sfhip.affine(o, scale=gvec, bias=zero, res=x, out=out if plain else None)
            return z if plain else sfhip.affine(z, scale=scale, bias=bias, relu=relu, rep=alpha, out=out)
        assert scale is None and not relu and alpha == 1 and out is None, "taped attention runs un-fused"
        z = sfhip.affine(o, scale=gvec, bias=zero, res=x)

        def bwd():  # z's buffer holds dL/dz (the BN backward wrote it in place), as in run()
            gz = z if getattr(bwd, "grad_in_place", True) else t.grad_of(z)
            t.add_pgrad(self.gamma, sfhip.rowdot(gz, o).sum().reshape(1))
            sfhip.affine(gz, scale=self.gamma.detach().expand(c).contiguous(), bias=zero, out=t.grad_of(o))  # dO = gamma * dz
            sfhip.axpy(gz, t.grad_of(x), 1.0, accumulate=True)                                     # residual

        t.record(bwd)
        return z

    def run(self, x, scale=None, bias=None, relu=False, alpha=1, out=None):
        c = self.input_channel
        if c > 128:
            return self._run_wide(x, scale, bias, relu, alpha, out)
        qkv = self.qkv(x)
        t = engine.tape()
        save = {} if t is not None else None
        q, k, v = qkv.slice(0, c), qkv.slice(c, c), qkv.slice(2 * c, c)
        z = sfhip.attention(q, k, v, x, self.gamma, scale=scale, bias=bias, relu=relu, alpha=alpha, out=out,
                            save=save)
        if t is not None:
            assert scale is None and not relu and alpha == 1, "taped attention runs un-fused (training-mode BN)"

            def bwd():  # z's grad buffer = the BN backward's in-place dz when z fed a BN, else grad_of(z)
                gz = z if getattr(bwd, "grad_in_place", True) else t.grad_of(z)
                dq = t.grad_of(qkv)
                dvec = sfhip.attention_bwd(q, k, v, gz, save["o"], save["lse"], self.gamma, dq.slice(0, c),
                                           dq.slice(c, c), dq.slice(2 * c, c))
                t.add_pgrad(self.gamma, dvec.sum().reshape(1))
                sfhip.axpy(gz, t.grad_of(x), 1.0, accumulate=True)  # residual: z = gamma*O + x

            t.record(bwd)
        return z

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
        z = sfhip.gate_apply(x, alpha, pooled, w3=self.conv.weight, scale=scale, bias=bias, relu=relu, out=out)
        t = engine.tape()
        if t is not None:
            assert scale is None and not relu, "taped ECA runs un-fused (training-mode BN follows)"
            w3 = self.conv.weight

            def bwd():  # z's buffer holds dL/dz (BN backward wrote it in place)
                dg = sfhip.tmax_dot(x, alpha, z)                       # [N, C] = sum dz * max_r x
                count = float((x.T // alpha) * x.H * x.W)
                tgt = t.pgrad_target(w3)
                dw = tgt if tgt is not None else torch.zeros(3, dtype=torch.float32, device=dg.device)
                # the 3-tap gate's derivative on the [N, C] vectors: one small HIP launch (no vendor conv kernels)
                gate, dpool = sfhip.eca_gate_bwd(dg, pooled, w3.detach().contiguous(), 1.0 / count, dw)
                if tgt is None:
                    t.add_pgrad(w3, dw)
                sfhip.eca_bwd_apply(x, alpha, z, gate, dpool, t.grad_of(x))

            t.record(bwd)
        return z

    def forward(self, x):
        plain = not isinstance(x, engine.Act)
        a = engine.enter([x])[0]
        y = self.run(a)
        return sfhip.to_ncthw(y) if plain else y
