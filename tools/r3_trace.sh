cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3t; mkdir -p $O; rm -rf $O/tl
SF_OVERLAP_PATHS=0 timeout 300 rocprofv3 --kernel-trace -d $O/tl --output-format csv -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extras --no-graph > $O/tl.log 2>&1
f=$(find $O/tl -name "*kernel_trace.csv" | head -1)
python3 - "$f" > $O/copy_neighbors.txt <<'P'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
n = len(rows)
last = rows[int(n * 0.7):]  # the last steps
prev = collections.Counter(); nxt = collections.Counter(); sizes = collections.Counter()
for i, r in enumerate(last):
    if "copyBuffer" in r["Kernel_Name"]:
        p = last[i - 1]["Kernel_Name"][:60] if i else ""
        q = last[i + 1]["Kernel_Name"][:60] if i + 1 < len(last) else ""
        prev[p] += 1; nxt[q] += 1
        sizes[(r.get("Grid_Size", r.get("Grid_Size_X", "")), r.get("Workgroup_Size", r.get("Workgroup_Size_X", "")))] += 1
print("after:"); [print("%5d  %s" % (c, k)) for k, c in prev.most_common(12)]
print("before:"); [print("%5d  %s" % (c, k)) for k, c in nxt.most_common(12)]
print("grid/wg:"); [print("%5d  %s" % (c, k)) for k, c in sizes.most_common(8)]
P
cat $O/copy_neighbors.txt; rm -rf $O/tl
