# round 3, first GPU call: whole GPU suite, default bench line, launcher path on one GPU with a real RCCL group, conv table
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3a; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 3000 $O/bench_default.json
SF_BENCH_FORCE_SPAWN=1 SF_FORCE_ALLREDUCE=1 timeout 600 python bench.py --no-cpu-baseline > $O/bench_spawn1.json 2> $O/bench_spawn1.err; echo "spawn rc=$?"; tail -c 1500 $O/bench_spawn1.json
timeout 400 python tools/prof_convs.py dual > $O/conv_per_shape.txt 2>&1; tail -8 $O/conv_per_shape.txt
cat gpurun_out/fullsize_report.txt | tail -12
