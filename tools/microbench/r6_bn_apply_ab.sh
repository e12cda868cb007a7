# bn_bwd_apply with four elements per thread (SF_BN_APPLY_UNROLL=1, default) against one (=0): per-shape pass rates
# (tools/microbench/bn_passes.py) and the cfg #3 step, alternating on one box
cd $GRAFT_REPO_ROOT
for v in 0 1; do echo "== SF_BN_APPLY_UNROLL=$v"; SF_BN_APPLY_UNROLL=$v timeout 300 python tools/microbench/bn_passes.py 2>/dev/null | tail -8; done
timeout 600 python -m pytest tests/test_backward_ops_gpu.py tests/test_ops_gpu.py tests/test_stage_grads_gpu.py -x -q -m gpu 2>&1 | tail -2
for rep in 1 2 3; do for v in 0 1; do echo -n "[step SF_BN_APPLY_UNROLL=$v] "; SF_BN_APPLY_UNROLL=$v timeout 200 python bench.py --steps 20 --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1; done; done
