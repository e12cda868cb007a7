#!/bin/bash
cd /root/repo
for c in 1 0 1 0; do SF_WGRAD_COMPANION=$c timeout 900 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-extras --no-graph 2>&1 | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('wgrad companion=$c', d['ms_per_step'])"; done
