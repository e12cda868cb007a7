# effective clock + MFMA pipe busy per conv / attention kernel over real train steps -> gpurun_out/pmc_step/summary.txt
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_step; rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d $O/p1 --output-format csv -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extras > $O/p1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES -d $O/p2 --output-format csv -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extras > $O/p2.log 2>&1
cd $R
python3 tools/pmc_short.py $(find gpurun_out/pmc_step/p1 gpurun_out/pmc_step/p2 -name "*counter_collection.csv") > gpurun_out/pmc_step/summary.txt 2>&1
rm -rf gpurun_out/pmc_step/p1 gpurun_out/pmc_step/p2
grep -E "attn|stem" gpurun_out/pmc_step/summary.txt | head -40
