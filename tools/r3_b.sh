cd $GRAFT_REPO_ROOT; O=gpurun_out/r3b; mkdir -p $O
timeout 300 python tools/microbench/stream_latency.py > $O/stream_latency.txt 2>&1; cat $O/stream_latency.txt
timeout 900 python -m pytest tests/test_fullsize_gpu.py -x -q -k "train_step" > $O/pytest_full.log 2>&1; tail -5 $O/pytest_full.log
timeout 900 python -m pytest tests/test_stage_grads_gpu.py tests/test_backward_ops_gpu.py tests/test_models_gpu.py -q > $O/pytest_stage.log 2>&1; tail -15 $O/pytest_stage.log
tail -20 gpurun_out/fullsize_report.txt
grep "stage-wise" gpurun_out/stage_grads_report.txt | tail -12
bash tools/timeline_prof.sh > $O/timeline_tail.txt 2>&1; cp gpurun_out/timeline/timeline.txt $O/timeline.txt; head -60 $O/timeline.txt
