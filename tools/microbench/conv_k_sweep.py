#!/usr/bin/env python3
"""Fixed cost vs per-K-step cost of the implicit-GEMM kernel: the same 128-wide 1x1x1 conv at M = 65 536 (two full
rounds of 128x128 tiles) for growing Cin.  Run with SF_CONV_CFG=4 SF_SPLIT_K=0 to pin the 128x128 tiling."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "tools")]
from microbench_conv import run  # noqa: E402

for cin in (64, 128, 256, 512, 1024, 2048):
    run("1x1 %4d->128 M=65536" % cin, 8, 8, 32, 32, cin, 128)
for cin in (64, 128, 256, 512, 1024, 2048):
    run("1x1 %4d->256 M=65536" % cin, 8, 8, 32, 32, cin, 256)
