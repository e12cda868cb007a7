# in-step A/B of the round-6 rings on ONE box (alternating): eager train step of cfg #3 at 8 clips
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
 for v in "SF_NONE=1" "SF_CONV_ROWS=0 SF_WGRAD_TRING=0" "SF_CONV_ROWS=0" "SF_WGRAD_TRING=0"; do
  echo -n "[$v] "; env $v timeout 200 python bench.py --steps 20 --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['ms_per_step'], 'ms', d['value'], 'clips/s')"
 done
done
