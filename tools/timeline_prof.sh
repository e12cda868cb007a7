# two-stream kernel timeline of the train step -> gpurun_out/timeline/timeline.txt (run on the GPU box via gpurun)
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/timeline; rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace -d $O/raw --output-format csv -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras > $O/run.log 2>&1
cd $R
python3 tools/prof_timeline.py $(find gpurun_out/timeline/raw -name "*kernel_trace.csv") > gpurun_out/timeline/timeline.txt 2>&1
rm -rf gpurun_out/timeline/raw
cat gpurun_out/timeline/timeline.txt | tail -40
