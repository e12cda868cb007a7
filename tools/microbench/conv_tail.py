#!/usr/bin/env python3
"""How much do the conv layers lose to the tail of their last round of workgroups?  The same layer at its real size
(49*2^k rows: 1568 / 784 / 392 tiles on 256 CUs) and at a size that fills whole rounds."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "tools")]
from microbench_conv import run  # noqa: E402

K3 = ((1, 3, 3), (1, 1, 1), (0, 1, 1))
T3 = ((3, 1, 1), (1, 1, 1), (1, 0, 0))
run("s2.b 3x3 64->64   M=200704 (1568 t)", 8, 8, 56, 56, 64, 64, *K3)
run("s2.b 3x3 64->64   M=196608 (1536 t)", 8, 8, 64, 48, 64, 64, *K3)
run("s2.c 64->256 +res M=200704", 8, 8, 56, 56, 64, 256, res=True)
run("s2.c 64->256 +res M=196608", 8, 8, 64, 48, 64, 256, res=True)
run("s2.a 256->64      M=200704", 8, 8, 56, 56, 256, 64)
run("s2.a 256->64      M=196608", 8, 8, 64, 48, 256, 64)
run("s3.b 3x3 128->128 M=50176 (392 t)", 8, 8, 28, 28, 128, 128, *K3)
run("s3.b 3x3 128->128 M=49152 (384 t)", 8, 8, 32, 24, 128, 128, *K3)
run("s3.b 3x3 128->128 M=65536 (512 t)", 8, 8, 32, 32, 128, 128, *K3)
run("s3.c 128->512 +res M=50176", 8, 8, 28, 28, 128, 512, res=True)
run("s3.c 128->512 +res M=49152", 8, 8, 32, 24, 128, 512, res=True)
run("s4.a t3 1024->256 M=12544", 8, 8, 14, 14, 1024, 256, *T3)
run("s4.a t3 1024->256 M=12288", 8, 8, 16, 12, 1024, 256, *T3)
run("s4.b 3x3 256->256 M=12544", 8, 8, 14, 14, 256, 256, *K3)
run("s4.b 3x3 256->256 M=12288", 8, 8, 16, 12, 256, 256, *K3)
run("s4.c 256->1024 +res M=12544", 8, 8, 14, 14, 256, 1024, res=True)
run("s4.c 256->1024 +res M=12288", 8, 8, 16, 12, 256, 1024, res=True)
