#!/bin/bash
# bf16-split conv kernel: correctness on every shape it can take (level 2), then per-shape timing at levels 0 / 2 with both tile heights
O=gpurun_out/r3cbx; mkdir -p $O; cd /root/repo; rm -f $O/*.txt
SF_CONV_BX=2 timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_backward_ops_gpu.py -q -m gpu -k "conv" 2>&1 | tail -4 > $O/tests.txt
SF_CONV_BX=0 timeout 400 python tools/prof_convs.py dual > $O/conv_bx0.txt 2>&1
SF_CONV_BX=2 SF_CONV_BX_TILE=1 timeout 400 python tools/prof_convs.py dual > $O/conv_bx2_t1.txt 2>&1
SF_CONV_BX=2 timeout 400 python tools/prof_convs.py dual > $O/conv_bx2.txt 2>&1
cat $O/tests.txt; grep "total" $O/conv_bx0.txt $O/conv_bx2_t1.txt $O/conv_bx2.txt
