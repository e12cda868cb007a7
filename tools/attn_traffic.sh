# HBM traffic of the attention kernels (two PMC passes: FETCH_SIZE, WRITE_SIZE) -> gpurun_out/attn_traffic/traffic.json
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/attn_traffic; rm -rf $O; mkdir -p $O
export ATTN_ITERS=2
timeout 250 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/f --output-format csv -- python3 $R/tools/microbench/attn_bench.py > $O/f.log 2>&1
timeout 250 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/w --output-format csv -- python3 $R/tools/microbench/attn_bench.py > $O/w.log 2>&1
cd $R
python3 tools/pmc_traffic_json.py $(find gpurun_out/attn_traffic/f -name "*counter_collection.csv") $(find gpurun_out/attn_traffic/w -name "*counter_collection.csv") > gpurun_out/attn_traffic/traffic.json
rm -rf gpurun_out/attn_traffic/f gpurun_out/attn_traffic/w
grep -E "grid=|hbm_bytes" gpurun_out/attn_traffic/traffic.json | paste - - | head -20
