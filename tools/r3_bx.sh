#!/bin/bash
cd /root/repo
timeout 600 python tools/microbench/attn_precision.py 32 2>&1 | grep "O:"
for i in 1 2 3; do ATTN_SHAPES=32 ATTN_ITERS=8 timeout 300 python tools/microbench/attn_bench.py 2>&1 | grep "d=32"; done
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_attention_bx_gpu.py -q -m gpu -k "attention or split" 2>&1 | tail -2
