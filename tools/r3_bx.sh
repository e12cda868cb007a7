#!/bin/bash
cd /root/repo
for v in 0 1; do SF_ATTN_BX64=$v timeout 600 python tools/microbench/attn_precision.py 64 2>&1 | grep "backward"; SF_ATTN_BX64=$v ATTN_SHAPES=64 ATTN_ITERS=8 timeout 300 python tools/microbench/attn_bench.py 2>&1 | grep "d=64"; done
timeout 900 python -m pytest tests/test_backward_ops_gpu.py tests/test_ops_gpu.py -q -m gpu -k "attention or attn" 2>&1 | tail -3
