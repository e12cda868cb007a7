cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final; mkdir -p $O
timeout 400 rocprofv3 --kernel-trace --stats -d $O/default --output-format csv -- python3 $R/bench.py --no-cpu-baseline > $O/default.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/train --output-format csv -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras > $O/train.log 2>&1
SF_OVERLAP_PATHS=0 timeout 300 rocprofv3 --kernel-trace --stats -d $O/serial --output-format csv -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras > $O/serial.log 2>&1
for p in "GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do n=$(echo $p | cut -c1-12 | tr " " _); PMC_CFGS=0,2,3 timeout 250 rocprofv3 --kernel-trace --pmc $p -d $O/pmcw_$n --output-format csv -- python3 $R/tools/microbench/conv_wave_pmc.py > $O/pmcw_$n.log 2>&1; done
cd $R
python3 tools/pmc_short.py $(find gpurun_out/final/pmcw_* -name "*counter_collection.csv") > gpurun_out/final/pmc_conv_wave.txt
timeout 300 python tools/prof_convs.py dual > gpurun_out/final/conv_per_shape.txt 2>&1
timeout 400 python tools/microbench/conv_wave_bench.py > gpurun_out/final/conv_wave_ab.txt 2>&1
timeout 400 python tools/microbench/wgrad_wave_bench.py > gpurun_out/final/wgrad_wave_ab.txt 2>&1
for w in dual slowfast ghostnet shufflenetv2; do timeout 600 python bench.py --workload $w > gpurun_out/final/bench_$w.json 2> gpurun_out/final/bench_$w.err; done
for d in default train serial; do cp $(find gpurun_out/final/$d -name "*kernel_stats.csv") gpurun_out/final/${d}_kernel_stats.csv; done
python3 tools/prof_stats.py gpurun_out/final/serial_kernel_stats.csv 8 50 > gpurun_out/final/serial_per_step.txt
tail -3 gpurun_out/final/serial_per_step.txt; cat gpurun_out/final/pmc_conv_wave.txt | head -20; tail -4 gpurun_out/final/conv_per_shape.txt
# round-2 additions: attention PMC view, two-stream timeline, fp32 MFMA / VALU co-execution and dependent-chain probes
bash tools/attn_pmc.sh > /dev/null 2>&1; cp gpurun_out/attn_pmc/summary.txt gpurun_out/final/pmc_attention.txt
bash tools/timeline_prof.sh > /dev/null 2>&1; cp gpurun_out/timeline/timeline.txt gpurun_out/final/timeline.txt
bash tools/pmc_step.sh > /dev/null 2>&1; cp gpurun_out/pmc_step/summary.txt gpurun_out/final/pmc_step.txt
python tools/microbench/attn_bench.py > gpurun_out/final/attn_bench.txt 2>&1
(cd /tmp && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $GRAFT_REPO_ROOT/tools/microbench/mfma_coexec.hip -o /tmp/mfma_coexec 2>/dev/null && timeout 100 /tmp/mfma_coexec 4000 > $GRAFT_REPO_ROOT/gpurun_out/final/mfma_coexec.txt)
(cd /tmp && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $GRAFT_REPO_ROOT/tools/microbench/mfma_chain.hip -o /tmp/mfma_chain 2>/dev/null && timeout 100 /tmp/mfma_chain 4000 > $GRAFT_REPO_ROOT/gpurun_out/final/mfma_chain.txt)
bash tools/attn_traffic.sh > /dev/null 2>&1; cp gpurun_out/attn_traffic/traffic.json gpurun_out/final/attention_hbm_traffic.json
ls gpurun_out/final | head -40
