cd $GRAFT_REPO_ROOT; O=gpurun_out/r3k; mkdir -p $O
timeout 900 python -m pytest tests/test_backward_ops_gpu.py -q -k "dwconv" > $O/pytest_dw.log 2>&1; tail -2 $O/pytest_dw.log
timeout 900 python -m pytest tests/test_models_gpu.py tests/test_stage_grads_gpu.py -q -k "ghostnet_w2_s64 or shufflenetv2 or mobilenet" > $O/pytest_g.log 2>&1; tail -2 $O/pytest_g.log
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3k
SF_OVERLAP_PATHS=0 timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_gb8 --output-format csv -- python3 $R/bench.py --workload ghostnet --batch 8 --steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-graph > $O/prof_gb8.log 2>&1
cp $(find $O/prof_gb8 -name "*kernel_stats.csv" | head -1) $O/ghostnet_b8_serial_kernel_stats.csv
python3 $R/tools/prof_stats.py $O/ghostnet_b8_serial_kernel_stats.csv 8 12 2>&1 | head -14
rm -rf $O/prof_gb8
cd $R; timeout 300 python bench.py --workload ghostnet --batch 8 --no-cpu-baseline --no-extras --steps 10 --warmup 5 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1
