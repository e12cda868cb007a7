#!/bin/bash
mkdir -p gpurun_out/r3bx; cd /root/repo
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3bx/bench.json 2> gpurun_out/r3bx/bench.err
tail -2 gpurun_out/r3bx/bench.err; cat gpurun_out/r3bx/bench.json | cut -c1-1200
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-graph | cut -c1-400
