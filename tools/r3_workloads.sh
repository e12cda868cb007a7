# round 3: cfg #1 / #2 / #5 as first-class measurements (bench lines, kernel tables, timelines) and the graph-vs-eager
# timelines of the cfg #3 train step.  Run on the GPU box: gpurun -- bash tools/r3_workloads.sh
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3w; mkdir -p $O; cd $R
for w in slowfast ghostnet shufflenetv2; do
  timeout 600 python bench.py --workload $w > $O/bench_$w.json 2> $O/bench_$w.err
done
timeout 600 python bench.py --workload ghostnet --batch 8 > $O/bench_ghostnet_b8.json 2> $O/bench_ghostnet_b8.err
cd /tmp
for w in slowfast ghostnet shufflenetv2; do
  SF_OVERLAP_PATHS=0 timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_$w --output-format csv -- python3 $R/bench.py --workload $w --steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-graph > $O/prof_$w.log 2>&1
  cp $(find $O/prof_$w -name "*kernel_stats.csv" | head -1) $O/${w}_serial_kernel_stats.csv
  python3 $R/tools/prof_stats.py $O/${w}_serial_kernel_stats.csv 8 40 > $O/${w}_serial_per_step.txt 2>&1
  rm -rf $O/prof_$w
done
SF_OVERLAP_PATHS=0 timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_gb8 --output-format csv -- python3 $R/bench.py --workload ghostnet --batch 8 --steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-graph > $O/prof_gb8.log 2>&1
cp $(find $O/prof_gb8 -name "*kernel_stats.csv" | head -1) $O/ghostnet_b8_serial_kernel_stats.csv
python3 $R/tools/prof_stats.py $O/ghostnet_b8_serial_kernel_stats.csv 8 40 > $O/ghostnet_b8_serial_per_step.txt 2>&1
rm -rf $O/prof_gb8
# vendor kernels on the training path?
grep -il "miopen\|naive_conv\|Cijk" $O/*_serial_kernel_stats.csv > $O/vendor_kernels.txt; echo "files with vendor kernels:"; cat $O/vendor_kernels.txt
# cfg #3: eager vs hipGraph replay of the train step, both under the kernel trace
timeout 300 rocprofv3 --kernel-trace -d $O/tl_eager --output-format csv -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras > $O/tl_eager.log 2>&1
python3 $R/tools/prof_timeline.py $(find $O/tl_eager -name "*kernel_trace.csv") > $O/timeline_eager.txt 2>&1; rm -rf $O/tl_eager
timeout 300 rocprofv3 --kernel-trace -d $O/tl_graph --output-format csv -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras --graph-train > $O/tl_graph.log 2>&1
python3 $R/tools/prof_timeline.py $(find $O/tl_graph -name "*kernel_trace.csv") > $O/timeline_graph.txt 2>&1; rm -rf $O/tl_graph
grep -o '"ms_per_step": [0-9.]*' $O/tl_eager.log $O/tl_graph.log
cd $R
for f in bench_slowfast bench_ghostnet bench_ghostnet_b8 bench_shufflenetv2; do python3 -c "
import json,sys
d=json.loads([l for l in open('$O/$f.json') if l.startswith('{')][-1])
print('$f', d['value'], d['unit'], d['ms_per_step'], 'ms', d['config']['launch'][:90], d.get('roofline',{}).get('frac'), d.get('eval_forward',{}).get('value'))
"; done
head -12 $O/timeline_eager.txt; head -12 $O/timeline_graph.txt
