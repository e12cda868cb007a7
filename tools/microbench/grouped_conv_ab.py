"""One-launch grouped conv (sf_conv_fwd_grouped / sf_conv_wgrad_grouped, group = grid z) against the per-group form it
replaced (G launches of the dense kernels on channel slices) on SlowFastShuffleNet-g3-sized layers: forward, data
gradient, weight gradient (+ finish).  usage: python tools/microbench/grouped_conv_ab.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "efficient-slowfast_amd"))
import sfhip  # noqa: E402

dev = torch.device("cuda:0")
SHAPES = [  # (name, N, T, H, W, Cin, Cout, G, kernel, pad)
    ("g3 s2 conv1 240->60 1x1", 8, 8, 28, 28, 240, 60, 3, (1, 1, 1), (0, 0, 0)),
    ("g3 s2 conv3 60->240 1x1", 8, 8, 28, 28, 60, 240, 3, (1, 1, 1), (0, 0, 0)),
    ("g3 s3 conv1 480->120 1x1", 8, 8, 14, 14, 480, 120, 3, (1, 1, 1), (0, 0, 0)),
    ("g3 s4 conv3 240->960 1x1", 8, 8, 7, 7, 240, 960, 3, (1, 1, 1), (0, 0, 0)),
    ("resnext 128->128 g32 1x3x3", 8, 8, 28, 28, 128, 128, 32, (1, 3, 3), (0, 1, 1)),
    ("resnext 256->256 g4 1x3x3", 8, 8, 14, 14, 256, 256, 4, (1, 3, 3), (0, 1, 1)),
]


def timed(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print("%-28s %5s | %-23s | %-23s | %-23s   (us: one launch / G launches)" % ("layer", "G", "forward", "data grad", "weight grad"))
for name, n, t, h, w, cin, cout, G, k, p in SHAPES:
    torch.manual_seed(0)
    wt = torch.randn(cout, cin // G, *k, device=dev) * 0.1
    x = sfhip.from_ncthw(torch.randn(n, cin, t, h, w, device=dev))
    dz = sfhip.from_ncthw(torch.randn(n, cout, t, h, w, device=dev))
    wp, wtp = sfhip.pack_grouped_weight_pair(wt, G)
    cg_i, cg_o = cin // G, cout // G
    pairs = [sfhip.pack_conv_weight_pair(wt[g * cg_o:(g + 1) * cg_o]) for g in range(G)]
    out = sfhip.new_act(x, n, t, h, w, cout)
    dx = sfhip.new_act(x, n, t, h, w, cin)
    dw = torch.zeros_like(wt)
    s1 = (1, 1, 1)

    def f_one():
        sfhip.conv_grouped(x, wp, G, k, s1, p, out=out)

    def f_per():
        for g in range(G):
            sfhip.conv(x.slice(g * cg_i, cg_i), pairs[g][0], k, s1, p, out=out.slice(g * cg_o, cg_o))

    def d_one():
        sfhip.conv_dgrad_grouped(dz, wtp, G, x, k, s1, p, out=dx, accumulate=True)

    def d_per():
        for g in range(G):
            sfhip.conv_dgrad(dz.slice(g * cg_o, cg_o), pairs[g][1], x.slice(g * cg_i, cg_i), k, s1, p,
                             out=dx.slice(g * cg_i, cg_i), accumulate=True)

    def w_one():
        sfhip.conv_wgrad_grouped(x, dz, G, k, s1, p, cin_pad=wp.shape[2], finish_into=dw)

    def w_per():
        for g in range(G):
            sfhip.conv_wgrad(x.slice(g * cg_i, cg_i), dz.slice(g * cg_o, cg_o), cg_o, k, s1, p,
                             cin_pad=pairs[g][0].shape[2], finish_into=(dw[g * cg_o:(g + 1) * cg_o], cg_i, 0))

    row = []
    for one, per in ((f_one, f_per), (d_one, d_per), (w_one, w_per)):
        a, b = timed(one), timed(per)
        row.append("%8.1f / %8.1f %4.2fx" % (a, b, b / a))
    print("%-28s %5d | %s | %s | %s" % (name, G, row[0], row[1], row[2]))
