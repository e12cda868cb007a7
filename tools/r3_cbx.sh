#!/bin/bash
# bf16-split conv kernel: correctness on every shape it can take (level 2), then per-shape timing at levels 0 / 2
O=gpurun_out/r3cbx; mkdir -p $O; cd /root/repo; rm -f $O/*.txt
SF_CONV_BX=2 timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_backward_ops_gpu.py -q -m gpu -k "conv" 2>&1 | tail -12 > $O/tests.txt
for lv in 0 2; do SF_CONV_BX=$lv timeout 400 python tools/prof_convs.py dual > $O/conv_bx$lv.txt 2>&1; done
cat $O/tests.txt; grep "total" $O/conv_bx0.txt $O/conv_bx2.txt
