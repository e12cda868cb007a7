cd $GRAFT_REPO_ROOT; O=gpurun_out/r3i; mkdir -p $O
timeout 900 python -m pytest tests/test_backward_ops_gpu.py tests/test_ops_gpu.py -q > $O/pytest_ops.log 2>&1; tail -3 $O/pytest_ops.log
timeout 300 python bench.py --workload ghostnet --batch 8 --no-cpu-baseline --no-extras --steps 10 --warmup 5 > $O/bench_gb8.json 2>/dev/null; grep -o '"ms_per_step": [0-9.]*' $O/bench_gb8.json | head -1
for lv in 0 2; do SF_CONV_SMALL=$lv timeout 400 python tools/prof_convs.py dual > $O/conv_s$lv.txt 2>&1; tail -6 $O/conv_s$lv.txt | head -4; done
