#!/bin/bash
cd /root/repo
for nw in 8 4; do SF_ATTN_BX_NW=$nw timeout 300 python tools/microbench/coexec_attn_conv.py 2>&1 | grep alone; done
SF_ATTN_BX=0 timeout 300 python tools/microbench/coexec_attn_conv.py 2>&1 | grep alone
