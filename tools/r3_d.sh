cd $GRAFT_REPO_ROOT; O=gpurun_out/r3d; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_backward_ops_gpu.py -x -q > $O/pytest_ops.log 2>&1; tail -2 $O/pytest_ops.log
timeout 900 python -m pytest tests/test_models_gpu.py -x -q > $O/pytest_models.log 2>&1; tail -2 $O/pytest_models.log
for lv in 0 1 2; do SF_CONV_WAVE_P=$lv timeout 400 python tools/prof_convs.py dual > $O/conv_p$lv.txt 2>&1; tail -6 $O/conv_p$lv.txt | head -3; done
for lv in 0 1 2 0 1 2; do SF_CONV_WAVE_P=$lv timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 30 > $O/bench_p$lv.json 2>/dev/null; echo "P=$lv $(grep -o '"ms_per_step": [0-9.]*' $O/bench_p$lv.json | head -1)"; done
