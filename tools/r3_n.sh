cd $GRAFT_REPO_ROOT; O=gpurun_out/r3n; mkdir -p $O
SF_BN_TICKET=2 timeout 600 python -m pytest tests/test_backward_ops_gpu.py -q -k "bn_backward" > $O/pytest_bn.log 2>&1; tail -2 $O/pytest_bn.log
for lv in 0 2 0 2; do SF_BN_TICKET=$lv timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 30 > $O/bench_t$lv.json 2>/dev/null; echo "TICKET=$lv $(grep -o '"ms_per_step": [0-9.]*' $O/bench_t$lv.json | head -1)"; done
