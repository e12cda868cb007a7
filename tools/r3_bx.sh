#!/bin/bash
mkdir -p gpurun_out/r3bx; cd /root/repo; rm -f gpurun_out/r3bx/*.txt
for bx in 0 1; do
SF_ATTN_BX=$bx timeout 300 python tools/microbench/attn_precision.py 32 >> gpurun_out/r3bx/precision.txt 2>&1
SF_ATTN_BX=$bx ATTN_SHAPES=32 ATTN_ITERS=6 timeout 300 python tools/microbench/attn_bench.py 2>&1 | grep "d=32" >> gpurun_out/r3bx/abl.txt
done
SF_ATTN_BX=1 timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_backward_ops_gpu.py -q -m gpu -k "attention or attn" -x 2>&1 | tail -5 > gpurun_out/r3bx/tests.txt
cat gpurun_out/r3bx/precision.txt gpurun_out/r3bx/abl.txt gpurun_out/r3bx/tests.txt | grep -v amdgpu.ids
