# round 6 closing check on a fresh box, in the driver's order: the driver's bench command first (first GPU process of the
# box), then the GPU suite, then smoke -> gpurun_out/final_r6b
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final_r6b; mkdir -p $O
cd $R
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_cmd.jsonl 2> $O/driver_cmd.err; echo "bench rc=$?"
timeout 900 python -m pytest tests -m gpu -x -q > $O/gputest_full.log 2>&1; echo "pytest rc=$?"; tail -2 $O/gputest_full.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
python3 -c "
import json
d = json.loads([l for l in open('$O/driver_cmd.jsonl') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d.get('parity_gate'), d['roofline']['frac'], d['step_roofline']['frac'])"
