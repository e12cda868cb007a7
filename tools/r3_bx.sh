#!/bin/bash
# bf16-piece attention kernels: precision against fp64 and times on both product paths, the ablations of DESIGN 6a-4,
# and the tests that cover them.  Run on the GPU box: gpurun -- bash tools/r3_bx.sh
cd /root/repo
for bx in 0 1; do
  SF_ATTN_BX=$bx timeout 600 python tools/microbench/attn_precision.py 32 8 64 2>&1 | grep -v amdgpu
  SF_ATTN_BX=$bx ATTN_ITERS=8 timeout 300 python tools/microbench/attn_bench.py 2>&1 | grep "d="
done
for dbg in 1 2 4 7; do echo "SF_ATTN_BX_DBG=$dbg (NW = 4)"; SF_ATTN_BX_NW=4 SF_ATTN_BX_DBG=$dbg ATTN_SHAPES=32 ATTN_ITERS=6 timeout 300 python tools/microbench/attn_bench.py 2>&1 | grep "d=32"; done
SF_SWEEP_PARTS=1 SF_ATTN_BX_DBG=64 SF_ATTN_BX_PADLDS=20000 ATTN_SHAPES=32 ATTN_ITERS=1 timeout 300 python tools/microbench/attn_bench.py 2>&1 | grep "fwd_bx" | sort | uniq | head -4
timeout 900 python -m pytest tests/test_attention_bx_gpu.py tests/test_ops_gpu.py tests/test_backward_ops_gpu.py -q -m gpu -k "attention or attn or split" 2>&1 | tail -2
