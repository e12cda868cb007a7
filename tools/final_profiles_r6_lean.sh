# round 6, lean evidence set (after a lost box: short, bounded commands only) -> gpurun_out/final_r6
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final_r6; mkdir -p $O
cd $R
timeout 200 python -m pytest tests/test_conv_configs_gpu.py -q -k pointwise > $O/pw_tests.log 2>&1; tail -2 $O/pw_tests.log
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $O/default --output-format csv -- python3 $R/bench.py --no-cpu-baseline > $O/default.log 2>&1
cp $(find $O/default -name "*kernel_stats.csv" | head -1) $O/bench_default_kernel_stats.csv; rm -rf $O/default
SF_OVERLAP_PATHS=0 timeout 300 rocprofv3 --kernel-trace --stats -d $O/serial --output-format csv -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-graph > $O/serial.log 2>&1
cp $(find $O/serial -name "*kernel_stats.csv" | head -1) $O/bench_train_only_serial_kernel_stats.csv; rm -rf $O/serial
cd $R
NSTEPS=$(python3 -c "
import json
j = [json.loads(l) for l in open('$O/serial.log') if l.startswith('{')][-1]
print(2 + j['config']['launch_probe']['steady_state_steps'] + 1 + 6 + 1)")
python3 tools/prof_stats.py $O/bench_train_only_serial_kernel_stats.csv $NSTEPS 70 > $O/serial_per_step.txt
for w in slowfast ghostnet shufflenetv2; do timeout 300 python bench.py --workload $w --no-cpu-baseline >> $O/bench_lines_workloads.jsonl 2>> $O/bench_workloads.err; done
timeout 300 python bench.py --workload ghostnet --batch 8 --no-cpu-baseline > $O/bench_line_ghostnet_b8.jsonl 2> $O/bench_ghostnet_b8.err
tail -3 $O/serial_per_step.txt; grep -o '"value": [0-9.]*, "unit": "clips/s", "n_gpus"' $O/default.log $O/bench_lines_workloads.jsonl $O/bench_line_ghostnet_b8.jsonl
