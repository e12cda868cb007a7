"""Non-local block (reference nonlocal_helper.py:8-148) on the HIP path.

The block's attention runs over N_q <= 6272 positions against N_k <= 1568 max-pooled positions with d = 256 / 512
channels — small enough to materialise one sample's score matrix, far outside the flash kernels' d <= 128 register
tiling.  So the core is three launches per sample on the existing implicit-GEMM kernel, with ACTIVATIONS in the weight
slot:  S = theta . phi^T (phi's NDHWC rows [N_k][d] already are the packed-weight layout), the normalisation
(`sf_row_softmax_fwd`, or 1/N_k folded into the GEMM epilogue for "dot_product"), and Y = P . g (weights = g^T from
the NDHWC->NCTHW kernel).  The backward is the same GEMMs transposed (data-gradient convs and weight-gradient
reductions over the query positions)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

import sfhip
from . import engine


class Nonlocal(nn.Module):
    """theta/phi/g/out 1x1x1 convs (+bias), optional max-pool of the key side, softmax or dot-product
    normalisation, final BN (zero-initialised when zero_init_final_norm), residual add."""

    def __init__(self, dim, dim_inner, pool_size=None, instantiation="softmax", zero_init_final_conv=False,
                 zero_init_final_norm=True, norm_eps=1e-5, norm_momentum=0.1, norm_module=nn.BatchNorm3d):
        super(Nonlocal, self).__init__()
        self.dim = dim
        self.dim_inner = dim_inner
        self.pool_size = pool_size
        self.instantiation = instantiation
        self.use_pool = False if pool_size is None else any((size > 1 for size in pool_size))
        self.norm_eps = norm_eps
        self.norm_momentum = norm_momentum
        self.conv_theta = nn.Conv3d(dim, dim_inner, kernel_size=1, stride=1, padding=0)
        self.conv_phi = nn.Conv3d(dim, dim_inner, kernel_size=1, stride=1, padding=0)
        self.conv_g = nn.Conv3d(dim, dim_inner, kernel_size=1, stride=1, padding=0)
        self.conv_out = nn.Conv3d(dim_inner, dim, kernel_size=1, stride=1, padding=0)
        self.conv_out.zero_init = zero_init_final_conv
        self.bn = norm_module(num_features=dim, eps=norm_eps, momentum=norm_momentum)
        self.bn.transform_final_bn = zero_init_final_norm
        if self.use_pool:
            self.pool = nn.MaxPool3d(kernel_size=self.pool_size, stride=self.pool_size, padding=[0, 0, 0])
        if instantiation not in ("softmax", "dot_product"):
            raise NotImplementedError("Unknown norm type {}".format(instantiation))

    def forward(self, x, reserve=(0, 0)):
        (xa,) = engine.enter([x])
        with engine.internal():
            y = self.run(xa, reserve)
        return engine.leave([y])[0]

    # ------------------------------------------------------------------------------------------------
    def run(self, x, reserve=(0, 0)):
        theta = engine.conv_bn_act(x, self.conv_theta)
        xp = x
        if self.use_pool:
            ks = tuple(self.pool_size)
            xp = engine.maxpool(x, ks, ks)
        phi = engine.conv_bn_act(xp, self.conv_phi)
        g = engine.conv_bn_act(xp, self.conv_g)
        y = dense_attention(theta, phi, g, self.instantiation == "softmax", float(self.dim_inner) ** -0.5)
        return engine.conv_bn_act(y, self.conv_out, self.bn, relu=False, res=x, out_reserve=reserve)


def dense_attention(theta, phi, g, softmax=True, sm_scale=1.0):
    """Y = normalise(theta phi^T) g per sample with the score matrix MATERIALISED: softmax(sm_scale * S) over the keys,
    or S / N_k (softmax=False: Nonlocal's "dot_product").  theta [N, ., d] queries, phi [N, ., d] keys, g [N, ., dv]
    values, dense NDHWC buffers.  Taped.  Serves Nonlocal (d = 256 / 512) and SpatialAttention heads wider than the
    flash kernels' 128 channels (SlowFastShuffleNet w2.0 / g3: d = 240 at s4_fuse, N <= 64 positions)."""
    N, d, dv = theta.N, theta.C, g.C
    nq, nk = theta.T * theta.H * theta.W, phi.T * phi.H * phi.W
    if d % 16 != 0:
        raise NotImplementedError("materialised attention needs a key width that is a multiple of 16 on the HIP "
                                  "path (got %d)" % d)
    assert theta.cs == d and phi.cs == d and g.cs == dv and theta.coff == phi.coff == g.coff == 0
    assert phi.C == d and (g.T, g.H, g.W) == (phi.T, phi.H, phi.W)
    dev = theta.buf.device
    nk_pad = (nk + 15) // 16 * 16
    inv_nk = None if softmax else torch.full((max(nk, d),), 1.0 / nk, dtype=torch.float32, device=dev)

    def transposed(a):  # [N, Tp, Hp, Wp, c] -> [N][c][nk_pad]: the packed weights of "multiply by a"
        t = sfhip.to_ncthw(a).view(N, a.C, nk)
        return t if nk_pad == nk else F.pad(t, (0, nk_pad - nk)).contiguous()

    def sample(a, n):
        return sfhip.Act(a.buf[n:n + 1], a.coff, a.C)

    one = (1, 1, 1)
    P = sfhip.new_act(theta, N, theta.T, theta.H, theta.W, nk)
    g_t = transposed(g)
    y = sfhip.new_act(theta, N, theta.T, theta.H, theta.W, dv)
    for n in range(N):  # S_n = theta_n phi_n^T (x 1/N_k for dot_product)
        sfhip.conv(sample(theta, n), phi.buf[n].view(nk, 1, d), one, scale=None if softmax else inv_nk[:nk],
                   out=sample(P, n))
    if softmax:
        sfhip.row_softmax(P, scale=sm_scale)
    for n in range(N):  # Y_n = P_n g_n
        sfhip.conv(sample(P, n), g_t[n].view(dv, 1, nk_pad), one, out=sample(y, n))
    t = engine.tape()
    if t is not None:
        def bwd():
            dy = t.grad_of(y)
            dth, dph, dg = t.grad_of(theta), t.grad_of(phi), t.grad_of(g)
            phi_t = transposed(phi)
            dP = sfhip.new_act(P, N, P.T, P.H, P.W, nk)
            for n in range(N):  # dP_n = dY_n g_n^T ;  dg_n = P_n^T dY_n
                sfhip.conv(sample(dy, n), g.buf[n].view(nk, 1, dv), one, scale=None if softmax else inv_nk[:nk],
                           out=sample(dP, n))
                dgn = sfhip.conv_wgrad(sample(dy, n), sample(P, n), nk, one, cin_pad=dv)
                dg.buf[n].view(nk, dv).add_(dgn.view(nk, dv))
            if softmax:
                sfhip.row_softmax_bwd(P, dP, scale=sm_scale)  # dP now holds dL/dS
            for n in range(N):  # dtheta_n += dS_n phi_n ;  dphi_n = dS_n^T theta_n
                dthn = sample(dth, n)
                sfhip.conv(sample(dP, n), phi_t[n].view(d, 1, nk_pad), one, res=dthn, out=dthn)
                dpn = sfhip.conv_wgrad(sample(theta, n), sample(dP, n), nk, one, cin_pad=d)
                dph.buf[n].view(nk, d).add_(dpn.view(nk, d))
        t.record(bwd)
    return y
