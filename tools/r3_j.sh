cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3j; mkdir -p $O
SF_OVERLAP_PATHS=0 timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_gb8 --output-format csv -- python3 $R/bench.py --workload ghostnet --batch 8 --steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-graph > $O/prof_gb8.log 2>&1
cp $(find $O/prof_gb8 -name "*kernel_stats.csv" | head -1) $O/ghostnet_b8_serial_kernel_stats.csv
python3 $R/tools/prof_stats.py $O/ghostnet_b8_serial_kernel_stats.csv 8 14 2>&1 | head -16
rm -rf $O/prof_gb8
