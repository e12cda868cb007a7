cd $GRAFT_REPO_ROOT; O=gpurun_out/r3p; mkdir -p $O
timeout 900 python -m pytest tests/test_fullsize_gpu.py -q -k "train_step" > $O/pytest_full.log 2>&1; tail -4 $O/pytest_full.log
timeout 1500 python -m pytest tests/test_stage_grads_gpu.py -q > $O/pytest_stage.log 2>&1; tail -4 $O/pytest_stage.log
tail -16 gpurun_out/fullsize_report.txt; grep "stage-wise" gpurun_out/stage_grads_report.txt | tail -12
