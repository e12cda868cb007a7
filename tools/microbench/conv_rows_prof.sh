# true kernel durations (rocprofv3 --kernel-trace --stats) of conv_rows_probe.py per ablation; run on the GPU box
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for shape in "32 56 56 32 8 3 1 1" "32 28 28 16 16 1 3 3" "32 56 56 8 32 1 1 1" "32 28 28 64 16 3 1 1"; do
 for dbg in 0 7 2 1 4; do
  rm -rf /tmp/pr; SF_CONV_ROWS_DBG=$dbg timeout 120 rocprofv3 --kernel-trace --stats -d /tmp/pr --output-format csv -- python3 $R/tools/microbench/conv_rows_probe.py $shape > /dev/null 2>&1
  f=$(find /tmp/pr -name "*kernel_stats.csv" | head -1)
  python3 - "$f" "$shape" "$dbg" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
out = []
for r in rows:
    n = r["Name"]
    if "conv_rows" in n or "conv_wave" in n or "conv_small" in n:
        out.append("%s x%s avg %.1f us" % (n.split("::")[-1][:44], r["Calls"], float(r["AverageNs"]) / 1e3))
print("shape %s dbg %s: %s" % (sys.argv[2], sys.argv[3], "; ".join(out)))
PY
 done
done
