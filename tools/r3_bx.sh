#!/bin/bash
cd /root/repo
for nw in 4 8; do SF_ATTN_BX_NW=$nw timeout 600 python tools/microbench/attn_precision.py 8 2>&1 | grep "backward"; SF_ATTN_BX_NW=$nw ATTN_SHAPES=8 ATTN_ITERS=6 timeout 300 python tools/microbench/attn_bench.py 2>&1 | grep "d=8"; done
SF_ATTN_BX=0 timeout 600 python tools/microbench/attn_precision.py 8 2>&1 | grep "backward"
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_backward_ops_gpu.py -q -m gpu -k "attention or attn" 2>&1 | tail -3
