# v_dot2c_f32_bf16 residuals in the three-way operand split (bx.h): exactness + rate, attention kernels, the step.
# tools/microbench/build/libsfhip_shift.so = the same sources with -DSF_BX_DOT2=0; split_dot2 = split_dot2.hip
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
echo "== multiplier from an SGPR"; timeout 120 tools/microbench/build/split_dot2
echo "== multiplier as the compiler folds it (inline constant -1.0)"; timeout 120 tools/microbench/build/split_dot2_inline | head -3
for rep in 1 2; do
  echo "== attn_bench, dot2 residuals"; ATTN_ITERS=10 timeout 200 python tools/microbench/attn_bench.py
  echo "== attn_bench, shift/mask residuals"; SF_LIB=$PWD/tools/microbench/build/libsfhip_shift.so ATTN_ITERS=10 timeout 200 python tools/microbench/attn_bench.py
done
for rep in 1 2 3; do
 for v in "SF_NONE=1" "SF_LIB=$PWD/tools/microbench/build/libsfhip_shift.so"; do
  echo -n "[step, ${v%%=*}] "; env $v timeout 200 python bench.py --steps 20 --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['ms_per_step'], 'ms', d['value'], 'clips/s', 'fwd_err', d.get('fwd_max_rel_err'), 'bwd_med', d.get('bwd_median_rel_err'))"
 done
done
echo '== driver command, dot2 residuals'; timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 | cut -c1-1500
} 2>&1 | tee gpurun_out/r06_dot2_ab.txt
