# round 5 evidence set -> gpurun_out/final_r5 (copy what is to be judged into profiles/ as r05_*).
# gpurun -- bash tools/final_profiles_r5.sh     (the FIRST command is the driver's, on the fresh box)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final_r5; mkdir -p $O
cd $R
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_cmd.jsonl 2> $O/driver_cmd.err
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --stats -d $O/default --output-format csv -- python3 $R/bench.py --no-cpu-baseline > $O/default.log 2>&1
cp $(find $O/default -name "*kernel_stats.csv" | head -1) $O/bench_default_kernel_stats.csv; rm -rf $O/default
SF_OVERLAP_PATHS=0 timeout 300 rocprofv3 --kernel-trace --stats -d $O/serial --output-format csv -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-graph > $O/serial.log 2>&1
cp $(find $O/serial -name "*kernel_stats.csv" | head -1) $O/bench_train_only_serial_kernel_stats.csv; rm -rf $O/serial
cd $R
# steps in that trace: warm-up 2 + steady-state groups + the step that re-binds `out` + timed 6 + the canonical
# gradient-hash step; the group count is in the JSON line
NSTEPS=$(python3 -c "
import json
j = [json.loads(l) for l in open('$O/serial.log') if l.startswith('{')][-1]
print(2 + j['config']['launch_probe']['steady_state_steps'] + 1 + 6 + 1)")
python3 tools/prof_stats.py $O/bench_train_only_serial_kernel_stats.csv $NSTEPS 70 > $O/serial_per_step.txt
bash tools/attn_traffic.sh > /dev/null 2>&1; cp gpurun_out/attn_traffic/traffic.json $O/attention_hbm_traffic.json
timeout 400 python tools/prof_convs.py dual > $O/conv_per_shape.txt 2>&1
timeout 300 python tools/microbench/conv_bx_bench.py > $O/conv_bx_ab.txt 2>&1
timeout 300 python tools/microbench/wgrad_bx_bench.py > $O/wgrad_bx_ab.txt 2>&1
(SF_ATTN_BX_DBG=8 ATTN_SHAPES=32 ATTN_ITERS=1 timeout 100 python tools/microbench/attn_bench.py; SF_ATTN_BX_PP=1 SF_ATTN_BX_DBG=8 ATTN_SHAPES=32 ATTN_ITERS=1 timeout 100 python tools/microbench/attn_bench.py) 2>&1 | grep -v amdgpu | sort | uniq > $O/attn_bwd_stamps.txt
(timeout 100 tools/microbench/build/mfma_pingpong 2000; timeout 100 tools/microbench/build/mfma_pingpong2 2000) > $O/mfma_pingpong.txt 2>&1
(timeout 300 python tools/microbench/attn_pp_check.py; for d in 0 256 128 512 640 2048 4096 32 64; do echo -n "ping-pong, SF_ATTN_BX_DBG=$d: "; SF_ATTN_BX_PP=1 SF_ATTN_BX_DBG=$d ATTN_SHAPES=32 ATTN_ITERS=3 timeout 100 python tools/microbench/attn_bench.py 2>&1 | grep "d=32" | sed "s/.*backward//"; done) 2>&1 | grep -v amdgpu > $O/attn_pingpong_ablations.txt
timeout 200 python tools/microbench/bn_passes.py 2>&1 | grep -v amdgpu > $O/bn_passes.txt
timeout 300 python tools/host_lead.py > $O/host_lead.txt 2>&1
timeout 700 python bench.py > $O/bench_dual.json 2> $O/bench_dual.err
for w in slowfast ghostnet shufflenetv2; do timeout 900 python bench.py --workload $w >> $O/bench_lines_workloads.jsonl 2>> $O/bench_workloads.err; done
# second half of the round: cfg #5 at 8 clips, the depthwise marches, the 4x4x1 attention backward, grouped convs
timeout 600 python bench.py --workload ghostnet --batch 8 --no-cpu-baseline > $O/bench_line_ghostnet_b8.jsonl 2> $O/bench_ghostnet_b8.err
(for on in 1 0; do echo "SF_ATTN_SMALL_44=$on"; SF_ATTN_SMALL_44=$on timeout 200 python tools/microbench/attn_small_bench.py; done) 2>&1 | grep -v amdgpu > $O/attn_small_44_ab.txt
timeout 100 ./tools/microbench/build/mfma4x4_probe > $O/mfma4x4_probe.txt 2>&1
timeout 200 python tools/microbench/grouped_conv_ab.py 2>&1 | grep -v amdgpu > $O/grouped_conv_ab.txt
(echo "# SF_DW_MARCH=0 python tools/prof_dwconvs.py ghostnet 8"; SF_DW_MARCH=0 timeout 300 python tools/prof_dwconvs.py ghostnet 8) 2>&1 | grep -v amdgpu > $O/dwconv_per_shape_before.txt
(echo "# python tools/prof_dwconvs.py ghostnet 8"; timeout 300 python tools/prof_dwconvs.py ghostnet 8) 2>&1 | grep -v amdgpu > $O/dwconv_per_shape_after.txt
(for w in "ghostnet --batch 8" "ghostnet" "shufflenetv2"; do for v in "SF_DW_MARCH=0 SF_ATTN_SMALL_44=0 SF_GRAD_ARENA_MB=0 SF_BN_TICKET=0" "SF_DW_MARCH=1"; do echo -n "$w [$v]: "; env $v timeout 400 python bench.py --workload $w --steps 10 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['ms_per_step'], 'ms', d['value'], 'clips/s', d['config'].get('launch'))"; done; done) > $O/small_configs_ab.txt 2>&1
tail -3 $O/serial_per_step.txt; tail -c 1500 $O/driver_cmd.jsonl
