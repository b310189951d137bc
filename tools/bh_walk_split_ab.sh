#!/bin/bash
# Round 5, VERDICT r04 next #3: the costliest p % of the walks of the previous step run as two halves of 32 bodies (Morton order kept).
# A/B of the traversal at 262 144 and 1 048 576 bodies, theta 0.5, p = 0 / 10 / 25 / 50 / 100; one line per run.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for n in 1048576 262144; do
  for p in 0 10 25 50 100; do
    NBX_WALK_SPLIT_PCT=$p python bench.py --workload bh --bodies $n --no-cpu-baseline --no-traffic --steps 40 --warmup 5 --steady-seconds 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'bodies': $n, 'split_pct': $p, 'ms_per_step': round(d['ms_per_step'],4), 'traversal_ms': round(d['ms_split']['bh_eval_kernel'],4), 'build_ms': round(d['ms_split']['tree_build'],4)}))"
  done
done
