cd $GRAFT_REPO_ROOT; O=gpurun_out/r3m; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_backward_ops_gpu.py -q > $O/pytest_ops.log 2>&1; tail -12 $O/pytest_ops.log
timeout 300 python tools/microbench/conv_small_probe.py 2>&1 | grep -v amdgpu.ids > $O/probe.txt; cat $O/probe.txt
