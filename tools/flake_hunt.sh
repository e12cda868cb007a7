# tests/test_multigpu_gpu.py::test_flat_gradients_two_ranks... failed in ~3 % of runs (round 6): two processes on one GPU
# computing the same shard's gradient disagree grossly.  Raise the rate with a background load, then toggle switches.
cd $GRAFT_REPO_ROOT
( while true; do ATTN_SHAPES=8 ATTN_ITERS=200 timeout 120 python tools/microbench/attn_bench.py > /dev/null 2>&1; done ) &
BG=$!
sleep 15
runs=${RUNS:-15}
for v in "$@"; do
  f=0
  for i in $(seq 1 $runs); do
    r=$(env $v timeout 300 python -m pytest tests/test_multigpu_gpu.py -x -q -m gpu -k "flat_gradients_two_ranks" 2>&1 | grep -E "passed|failed" | tail -1)
    case "$r" in *failed*) f=$((f+1));; esac
  done
  echo "[$v] $f failures of $runs"
done
kill $BG 2>/dev/null; sleep 1
ls gpurun_out | grep flat_fail; cat gpurun_out/flat_fail_rank*.json 2>/dev/null | cut -c1-1200
