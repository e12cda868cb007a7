# round 4 evidence set -> gpurun_out/final_r4 (copy what is to be judged into profiles/ as r04_*).
# gpurun -- bash tools/final_profiles_r4.sh
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final_r4; mkdir -p $O
cd /tmp
timeout 500 rocprofv3 --kernel-trace --stats -d $O/default --output-format csv -- python3 $R/bench.py --no-cpu-baseline > $O/default.log 2>&1
cp $(find $O/default -name "*kernel_stats.csv" | head -1) $O/bench_default_kernel_stats.csv; rm -rf $O/default
SF_OVERLAP_PATHS=0 timeout 300 rocprofv3 --kernel-trace --stats -d $O/serial --output-format csv -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-graph > $O/serial.log 2>&1
cp $(find $O/serial -name "*kernel_stats.csv" | head -1) $O/bench_train_only_serial_kernel_stats.csv; rm -rf $O/serial
# --no-graph: no host-bound probe steps; warm-up 2 + timed 6 = 8 identical steps
python3 $R/tools/prof_stats.py $O/bench_train_only_serial_kernel_stats.csv 8 70 > $O/serial_per_step.txt
timeout 300 rocprofv3 --kernel-trace -d $O/tl --output-format csv -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras --no-graph > $O/tl.log 2>&1
python3 $R/tools/prof_timeline.py $(find $O/tl -name "*kernel_trace.csv") > $O/timeline.txt 2>&1; rm -rf $O/tl
cd $R
bash tools/attn_traffic.sh > /dev/null 2>&1; cp gpurun_out/attn_traffic/traffic.json $O/attention_hbm_traffic.json
timeout 400 python tools/prof_convs.py dual > $O/conv_per_shape.txt 2>&1
timeout 300 python tools/microbench/conv_bx_bench.py > $O/conv_bx_ab.txt 2>&1
timeout 300 python tools/microbench/wgrad_bx_bench.py > $O/wgrad_bx_ab.txt 2>&1
(echo "conv_wgrad_rows.hip + flat finish (default)"; timeout 200 python tools/microbench/wgrad_small_bench.py; echo; echo "SF_WGRAD_ROWS=0 (conv_wgrad_small_kernel)"; SF_WGRAD_ROWS=0 timeout 200 python tools/microbench/wgrad_small_bench.py) 2>&1 | grep -v "amdgpu.ids" > $O/wgrad_small_ab.txt
timeout 300 python tools/microbench/conv_pw_bench.py > $O/conv_pw_ab.txt 2>&1
timeout 600 python tools/whatif_skip.py > $O/whatif_skip.txt 2>&1
timeout 300 python tools/host_lead.py > $O/host_lead.txt 2>&1
timeout 700 python bench.py > $O/bench_dual.json 2> $O/bench_dual.err
for w in slowfast ghostnet shufflenetv2; do timeout 900 python bench.py --workload $w >> $O/bench_lines_workloads.jsonl 2>> $O/bench_workloads.err; done
tail -3 $O/serial_per_step.txt; tail -5 $O/conv_per_shape.txt; grep -v amdgpu $O/whatif_skip.txt | tail -9; tail -c 2500 $O/bench_dual.json
