cd $GRAFT_REPO_ROOT
for m in 0 1 2 3 4 5 7; do echo "SF_ATTN_DBG=$m"; SF_ATTN_DBG=$m ATTN_SHAPES=32 ATTN_ITERS=6 timeout 120 python tools/microbench/attn_bench.py 2>&1 | grep "d=32"; done
