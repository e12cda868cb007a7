#!/bin/bash
cd /root/repo
for pad in 0 20000; do echo "pad $pad"; SF_SWEEP_PARTS=1 SF_ATTN_BX_DBG=64 SF_ATTN_BX_PADLDS=$pad ATTN_SHAPES=32 ATTN_ITERS=1 timeout 300 python tools/microbench/attn_bench.py 2>&1 | grep "fwd_bx\|d=32" | sort | uniq -c | head -5; done
