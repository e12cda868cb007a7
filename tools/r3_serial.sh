# serial per-step kernel table of the current build -> gpurun_out/r3s/serial_per_step.txt
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3s; mkdir -p $O
SF_OVERLAP_PATHS=0 timeout 300 rocprofv3 --kernel-trace --stats -d $O/serial --output-format csv -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-graph > $O/serial.log 2>&1
cp $(find $O/serial -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv; rm -rf $O/serial
python3 $R/tools/prof_stats.py $O/kernel_stats.csv 8 70 > $O/serial_per_step.txt; head -45 $O/serial_per_step.txt
