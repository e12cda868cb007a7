#!/bin/bash
mkdir -p gpurun_out/r3bx; cd /root/repo
timeout 3000 python -m pytest tests -q -m gpu -x 2>&1 | tail -3
for rp in 0 1 0 1; do SF_BATCH_REPACK=$rp timeout 900 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-extras --no-graph 2>&1 | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('repack batched=$rp', d['ms_per_step'])"; done
timeout 600 python tools/host_lead.py 2>&1 | grep -v amdgpu | tail -3
