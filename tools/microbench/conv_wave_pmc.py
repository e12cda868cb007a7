#!/usr/bin/env python3
"""Driver for rocprofv3 PMC passes over the conv_wave kernels: a few long-K layer shapes of cfg #3, each with a few
forced tile configurations, 10 launches each (kernel names carry <TM, TN, KS, KV>; grids identify the shape).
usage: rocprofv3 --kernel-trace --pmc <counters> -d <dir> --output-format csv -- python3 tools/microbench/conv_wave_pmc.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "efficient-slowfast_amd")]
import torch  # noqa: E402
import sfhip  # noqa: E402
from sfhip import Act  # noqa: E402

dev = torch.device("cuda:0")
L = sfhip.lib()
SHAPES = [
    ("s5 3x1x1 2048->512", 8, 7, 7, 2048, 512, (3, 1, 1), (1, 0, 0)),
    ("s4 3x1x1 1024->256", 8, 14, 14, 1024, 256, (3, 1, 1), (1, 0, 0)),
    ("s3 3x1x1 576->256", 8, 28, 28, 576, 256, (3, 1, 1), (1, 0, 0)),
    ("s3 1x3x3 128->128", 8, 28, 28, 128, 128, (1, 3, 3), (0, 1, 1)),
    ("s2 1x3x3 64->64", 8, 56, 56, 64, 64, (1, 3, 3), (0, 1, 1)),
]
CFGS = [int(c) for c in os.environ.get("PMC_CFGS", "0,2,3").split(",")]
for name, T, H, W, cin, cout, k, p in SHAPES:
    x = Act(torch.randn((8, T, H, W, cin), device=dev))
    wp = sfhip.pack_conv_weight(torch.randn((cout, cin) + k, device=dev) * 0.02)
    for c in CFGS:
        L.sf_conv_tune(1, c)
        for _ in range(10):
            sfhip.conv(x, wp, k, (1, 1, 1), p)
        torch.cuda.synchronize()
