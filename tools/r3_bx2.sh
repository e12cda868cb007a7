#!/bin/bash
cd /root/repo
timeout 900 python -m pytest tests/test_backward_ops_gpu.py -q -m gpu -k "repack" 2>&1 | tail -3
