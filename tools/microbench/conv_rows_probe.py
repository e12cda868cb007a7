#!/usr/bin/env python3
"""One conv_rows shape, kernel time by HIP events over 20 back-to-back launches (hot operands) — for A/B runs with
SF_CONV_ROWS_NB / SF_CONV_ROWS_WGS.  usage: conv_rows_probe.py T H W Cin Cout kT kH kW"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "efficient-slowfast_amd")]
import torch  # noqa: E402
import sfhip  # noqa: E402
from sfhip import Act  # noqa: E402

T, H, W, cin, cout, kt, kh, kw = [int(v) for v in sys.argv[1:9]]
dev = torch.device("cuda:0")
L = sfhip.lib()
k = (kt, kh, kw)
p = tuple(kk // 2 for kk in k)
x = Act(torch.randn((8, T, H, W, cin), device=dev))
w = torch.randn((cout, cin) + k, device=dev) / (cin * kt * kh * kw) ** 0.5
wp, wtp = sfhip.pack_conv_weight_pair(w)
out = sfhip.new_act(x, 8, T, H, W, cout)
for mode in (0, 2):
    L.sf_conv_tune(22, mode)
    for _ in range(3):
        sfhip.conv(x, wp, k, padding=p, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        sfhip.conv(x, wp, k, padding=p, out=out)
    e1.record()
    torch.cuda.synchronize()
    print("mode %d: %.1f us per launch (hot, 20 back to back)" % (mode, e0.elapsed_time(e1) * 1e3 / 20))
