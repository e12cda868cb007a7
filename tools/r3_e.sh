cd $GRAFT_REPO_ROOT; O=gpurun_out/r3e; mkdir -p $O
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_backward_ops_gpu.py -x -q > $O/pytest_ops.log 2>&1; tail -2 $O/pytest_ops.log
timeout 300 python tools/count_copies.py > $O/copies.txt 2>&1; tail -32 $O/copies.txt | grep -v "^{" 
timeout 300 python tools/prof_torch_ops.py > $O/torch_ops.txt 2>&1; tail -70 $O/torch_ops.txt
for lv in 0 3 0 3; do SF_CONV_WAVE_P=$lv timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 30 > $O/bench_p$lv.json 2>/dev/null; echo "P=$lv $(grep -o '"ms_per_step": [0-9.]*' $O/bench_p$lv.json | head -1)"; done
