#!/usr/bin/env python3
"""What the vendor fp32 GEMM (hipBLASLt via torch.mm) reaches on the conv-as-GEMM shapes of the R50 layers — a
practical ceiling to compare conv_igemm against.  usage: tools/microbench/gemm_ref.py"""
import torch

torch.backends.cuda.matmul.allow_tf32 = False
dev = torch.device("cuda:0")
shapes = [  # (M, N, K) = (positions, Cout, Cin*taps)
    (200704, 64, 576), (200704, 256, 64), (200704, 64, 256), (50176, 128, 1152), (50176, 512, 128),
    (12544, 256, 3072), (12544, 256, 2304), (12544, 1024, 256), (3136, 512, 6144), (3136, 2048, 512),
]
for m, n, k in shapes:
    a = torch.randn(m, k, device=dev)
    b = torch.randn(k, n, device=dev)
    for _ in range(3):
        torch.mm(a, b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        torch.mm(a, b)
    e1.record()
    e1.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print("M=%6d N=%4d K=%4d  %.3f ms  %.1f TFLOP/s" % (m, n, k, ms, 2.0 * m * n * k / ms / 1e9))
