#!/bin/bash
cd /root/repo
for dbg in 0 16 32 48 2; do echo "dbg $dbg"; SF_ATTN_BX_DBG=$dbg ATTN_SHAPES=32 ATTN_ITERS=6 timeout 300 python tools/microbench/attn_bench.py 2>&1 | grep "d=32"; done
