# PMC view of the attention kernels at production shapes -> gpurun_out/attn_pmc/summary.txt (run on the GPU box)
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/attn_pmc; rm -rf $O; mkdir -p $O
export ATTN_ITERS=3
i=0
for p in "GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES"; do
  i=$((i+1)); timeout 250 rocprofv3 --kernel-trace --pmc $p -d $O/p$i --output-format csv -- python3 $R/tools/microbench/attn_bench.py > $O/p$i.log 2>&1
done
cd $R
python3 tools/pmc_short.py $(find gpurun_out/attn_pmc/p* -name "*counter_collection.csv") > gpurun_out/attn_pmc/summary.txt 2>&1
python3 - <<'PY' >> gpurun_out/attn_pmc/summary.txt
import csv, glob, collections
per = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/attn_pmc/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "attn" not in n: continue
        k = n.replace("(anonymous namespace)::", "").split("(")[0][-40:] + " g" + r["Grid_Size"]
        per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(per):
    print(k, " ".join("%s=%.3g" % (c, sum(v) / len(v)) for c, v in sorted(per[k].items())))
PY
rm -rf gpurun_out/attn_pmc/p1 gpurun_out/attn_pmc/p2 gpurun_out/attn_pmc/p3
cat gpurun_out/attn_pmc/summary.txt | grep -v reduce | grep -v merge
