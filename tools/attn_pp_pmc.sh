# PMC comparison of the free-running and the ping-pong attention backward (d = 32, N = 25 088, B = 8):
# LDS bank conflicts, LDS / VALU / MFMA busy -> gpurun_out/attn_pp_pmc.txt (run on the GPU box)
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/attn_pp_pmc; rm -rf $O; mkdir -p $O
export ATTN_ITERS=2 ATTN_SHAPES=32
for pp in 0 1; do
  i=0
  for p in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM"; do
    i=$((i+1)); SF_ATTN_BX_PP=$pp timeout 250 rocprofv3 --kernel-trace --pmc $p -d $O/pp${pp}_p$i --output-format csv -- python3 $R/tools/microbench/attn_bench.py > $O/pp${pp}_p$i.log 2>&1
  done
done
cd $R
python3 - <<'PY' > gpurun_out/attn_pp_pmc.txt
import csv, glob, collections
per = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/attn_pp_pmc/pp*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "attn_bwd_bx" not in n: continue
        k = n.replace("(anonymous namespace)::", "").split("(")[0][-30:]
        per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(per):
    print(k)
    for c, v in sorted(per[k].items()):
        print("    %-28s %16.0f" % (c, sum(v) / len(v)))
PY
rm -rf $O
cat gpurun_out/attn_pp_pmc.txt
