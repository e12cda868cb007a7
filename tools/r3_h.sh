cd $GRAFT_REPO_ROOT; O=gpurun_out/r3h; mkdir -p $O
timeout 900 python -m pytest tests/test_backward_ops_gpu.py -q -k "dwconv" > $O/pytest_dw.log 2>&1; tail -3 $O/pytest_dw.log
timeout 1500 python -m pytest tests/test_stage_grads_gpu.py tests/test_models_gpu.py -q -k "ghostnet or shufflenetv2 or mobilenet" > $O/pytest_g.log 2>&1; tail -4 $O/pytest_g.log
timeout 300 python bench.py --workload ghostnet --batch 8 --no-cpu-baseline --steps 10 --warmup 5 > $O/bench_gb8.json 2>/dev/null; grep -o '"ms_per_step": [0-9.]*' $O/bench_gb8.json | head -1
timeout 300 python bench.py --workload ghostnet --no-cpu-baseline --steps 10 --warmup 5 > $O/bench_gb2.json 2>/dev/null; grep -o '"ms_per_step": [0-9.]*' $O/bench_gb2.json | head -1
timeout 300 python bench.py --workload shufflenetv2 --no-cpu-baseline --steps 20 --warmup 5 > $O/bench_sh.json 2>/dev/null; grep -o '"ms_per_step": [0-9.]*' $O/bench_sh.json | head -1
