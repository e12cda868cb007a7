#!/bin/bash
mkdir -p gpurun_out/r3bx; cd /root/repo
timeout 3000 python -m pytest tests -q -m gpu -x 2>&1 | tail -3
timeout 900 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-extras --no-graph 2>&1 | grep '^{' | cut -c1-330
