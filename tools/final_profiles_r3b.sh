# round 3 (second half: bf16-piece attention) final evidence set -> gpurun_out/final_r3b (copy what is to be judged into
# profiles/ as r03b_*).  gpurun -- bash tools/final_profiles_r3b.sh
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final_r3b; mkdir -p $O
cd $R; timeout 1800 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $O/default --output-format csv -- python3 $R/bench.py --no-cpu-baseline > $O/default.log 2>&1
cp $(find $O/default -name "*kernel_stats.csv" | head -1) $O/bench_default_kernel_stats.csv; rm -rf $O/default
SF_OVERLAP_PATHS=0 timeout 300 rocprofv3 --kernel-trace --stats -d $O/serial --output-format csv -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-graph > $O/serial.log 2>&1
cp $(find $O/serial -name "*kernel_stats.csv" | head -1) $O/bench_train_only_serial_kernel_stats.csv; rm -rf $O/serial
python3 $R/tools/prof_stats.py $O/bench_train_only_serial_kernel_stats.csv 8 60 > $O/serial_per_step.txt
timeout 300 rocprofv3 --kernel-trace -d $O/tl --output-format csv -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras --no-graph > $O/tl.log 2>&1
python3 $R/tools/prof_timeline.py $(find $O/tl -name "*kernel_trace.csv") > $O/timeline.txt 2>&1; rm -rf $O/tl
cd $R
bash tools/attn_traffic.sh > /dev/null 2>&1; cp gpurun_out/attn_traffic/traffic.json $O/attention_hbm_traffic.json
timeout 400 python tools/prof_convs.py dual > $O/conv_per_shape.txt 2>&1
timeout 300 python tools/whatif_skip.py > $O/whatif_skip.txt 2>&1
timeout 300 python tools/host_lead.py > $O/host_lead.txt 2>&1
for bx in 0 1; do SF_ATTN_BX=$bx timeout 300 python tools/microbench/attn_precision.py 32 8 >> $O/attn_precision.txt 2>&1; SF_ATTN_BX=$bx ATTN_ITERS=8 timeout 300 python tools/microbench/attn_bench.py >> $O/attn_bench.txt 2>&1; done
timeout 300 python tools/microbench/coexec_attn_conv.py > $O/coexec_attn_conv.txt 2>&1
timeout 700 python bench.py > $O/bench_dual.json 2> $O/bench_dual.err
tail -3 $O/serial_per_step.txt; tail -5 $O/conv_per_shape.txt; grep -v amdgpu $O/whatif_skip.txt | tail -8; grep -v amdgpu $O/attn_bench.txt; tail -c 3000 $O/bench_dual.json
