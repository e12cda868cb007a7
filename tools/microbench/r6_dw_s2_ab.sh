# stride-2 depthwise marches (dwconv_march.hip, round 6): per-shape launches of cfg #5 at 8 clips and the bench lines,
# SF_DW_MARCH_S2=0 (position-per-thread kernels) against the default -> gpurun_out/r06_dw_s2
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06_dw_s2; mkdir -p $O
SF_DW_MARCH_S2=0 timeout 300 python tools/prof_dwconvs.py ghostnet 8 > $O/dwconv_per_shape_before.txt 2>/dev/null
timeout 300 python tools/prof_dwconvs.py ghostnet 8 > $O/dwconv_per_shape_after.txt 2>/dev/null
for f in before after; do echo "== $f"; grep "1x2x2" $O/dwconv_per_shape_$f.txt | awk '{k=$1; ms[k]+=$8} END {for (k in ms) print k, ms[k]}'; done
for rep in 1 2; do for v in "SF_DW_MARCH_S2=0" "SF_DW_MARCH_S2=1"; do
  echo -n "[ghostnet b8 $v] "; env $v timeout 300 python bench.py --workload ghostnet --batch 8 --no-cpu-baseline --no-extras 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1
done; done
for v in "SF_DW_MARCH_S2=0" "SF_DW_MARCH_S2=1"; do
  echo -n "[ghostnet b2 $v] "; env $v timeout 300 python bench.py --workload ghostnet --no-cpu-baseline --no-extras 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1
  echo -n "[shufflenetv2 $v] "; env $v timeout 300 python bench.py --workload shufflenetv2 --no-cpu-baseline --no-extras 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1
done
timeout 900 python -m pytest tests/test_models_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu -k "ghost or shuffle or mobilenet" 2>&1 | tail -3
