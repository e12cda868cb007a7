#!/bin/bash
mkdir -p gpurun_out/r3bx; cd /root/repo
timeout 3000 python -m pytest tests -q -m gpu -x 2>&1 | tail -8 > gpurun_out/r3bx/tests_all.txt
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r3bx/bench.json 2> gpurun_out/r3bx/bench.err
cat gpurun_out/r3bx/tests_all.txt; tail -3 gpurun_out/r3bx/bench.err; cat gpurun_out/r3bx/bench.json
