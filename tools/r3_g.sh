cd $GRAFT_REPO_ROOT; O=gpurun_out/r3g; mkdir -p $O
timeout 1500 python -m pytest tests/test_models_gpu.py tests/test_stage_grads_gpu.py -q -k "s112" > $O/pytest_s112.log 2>&1; tail -6 $O/pytest_s112.log
grep "s112" gpurun_out/stage_grads_report.txt | tail -3
bash tools/r3_workloads.sh > $O/workloads.log 2>&1; tail -40 $O/workloads.log
