# round 6 evidence set -> gpurun_out/final_r6 (copy what is to be judged into profiles/ as r06_*).
# NOTE (round 6): this full list lost its GPU box six minutes in; the round's evidence came from tools/final_profiles_r6_lean.sh
# (short, bounded commands only).  Kept for the command list; prefer the lean script.
# gpurun -- bash tools/final_profiles_r6.sh     (the FIRST command is the driver's, on the fresh box)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final_r6; mkdir -p $O
cd $R
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_cmd.jsonl 2> $O/driver_cmd.err
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --stats -d $O/default --output-format csv -- python3 $R/bench.py --no-cpu-baseline > $O/default.log 2>&1
cp $(find $O/default -name "*kernel_stats.csv" | head -1) $O/bench_default_kernel_stats.csv; rm -rf $O/default
SF_OVERLAP_PATHS=0 timeout 300 rocprofv3 --kernel-trace --stats -d $O/serial --output-format csv -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-graph > $O/serial.log 2>&1
cp $(find $O/serial -name "*kernel_stats.csv" | head -1) $O/bench_train_only_serial_kernel_stats.csv; rm -rf $O/serial
cd $R
NSTEPS=$(python3 -c "
import json
j = [json.loads(l) for l in open('$O/serial.log') if l.startswith('{')][-1]
print(2 + j['config']['launch_probe']['steady_state_steps'] + 1 + 6 + 1)")
python3 tools/prof_stats.py $O/bench_train_only_serial_kernel_stats.csv $NSTEPS 70 > $O/serial_per_step.txt
timeout 400 python tools/prof_convs.py dual > $O/conv_per_shape.txt 2>&1
timeout 200 python tools/microbench/bn_passes.py 2>&1 | grep -v amdgpu > $O/bn_passes.txt
timeout 300 python tools/microbench/conv_pw_bench.py 2>&1 | grep -v amdgpu > $O/conv_pw64_ab.txt
for w in slowfast ghostnet shufflenetv2; do timeout 900 python bench.py --workload $w >> $O/bench_lines_workloads.jsonl 2>> $O/bench_workloads.err; done
timeout 600 python bench.py --workload ghostnet --batch 8 --no-cpu-baseline > $O/bench_line_ghostnet_b8.jsonl 2> $O/bench_ghostnet_b8.err
bash tools/timeline_prof.sh > /dev/null 2>&1; cp gpurun_out/timeline/timeline.txt $O/timeline.txt
tail -3 $O/serial_per_step.txt; tail -c 1200 $O/driver_cmd.jsonl
