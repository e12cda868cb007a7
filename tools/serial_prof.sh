# one-stream kernel profile of the train step -> gpurun_out/serial/per_step.txt (run on the GPU box via gpurun)
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/serial; rm -rf $O; mkdir -p $O
SF_OVERLAP_PATHS=0 timeout 300 rocprofv3 --kernel-trace --stats -d $O/raw --output-format csv -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras > $O/serial.log 2>&1
cd $R
cp $(find gpurun_out/serial/raw -name "*kernel_stats.csv") gpurun_out/serial/kernel_stats.csv
python3 tools/prof_stats.py gpurun_out/serial/kernel_stats.csv 8 60 > gpurun_out/serial/per_step.txt
rm -rf gpurun_out/serial/raw
tail -2 gpurun_out/serial/per_step.txt
