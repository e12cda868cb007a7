# Is the train step of cfg #3 clock / power limited?  rocm-smi power + sclk sampled every ~0.25 s while bench.py runs 300
# timed steps (eager), and while the box idles.  -> gpurun_out/r06_power_during_step.txt
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
O=gpurun_out/r06_power_during_step.txt
sample() { rocm-smi --showpower --showclocks --showuse --showtemp 2>/dev/null | grep -E "Power|sclk|fclk|mclk|GPU use|Temperature \(Sensor (junction|edge)" | tr -s ' ' | cut -c1-90 | tr '\n' ';'; echo; }
{
echo "== idle"; sample; rocm-smi --showmaxpower 2>/dev/null | grep -i "max"
echo "== during bench.py --steps 800 --warmup 10 --no-cpu-baseline --no-extras"
python bench.py --steps 800 --warmup 10 --no-cpu-baseline --no-extras > gpurun_out/power_bench.log 2>/dev/null &
BP=$!
sleep 6
while kill -0 $BP 2>/dev/null; do sample; sleep 0.25; done
grep -o '"ms_per_step": [0-9.]*' gpurun_out/power_bench.log
echo "== idle again"; sample
} > $O 2>&1
tail -40 $O
