#!/bin/bash
# Barnes-Hut step (device tree, reference fold) with the root fold on a side stream vs inline, by body count
cd "${GRAFT_REPO_ROOT:-/root/repo}"
echo "# bench.py --workload bh --theta 0.85 --bodies N --steps 60 --warmup 10 (ms per step); side = NBX_SIDE_STREAMS_FROM=0, inline = 100000000"
for n in 600 1000 2000 3000 4096 6000 8000 10000 20000; do
  for mode in 0 100000000; do
    NBX_SIDE_STREAMS_FROM=$mode python bench.py --workload bh --theta 0.85 --bodies $n --steps 60 --warmup 10 --no-cpu-baseline --no-traffic --steady-seconds 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('n', $n, 'side' if $mode == 0 else 'inline', 'ms/step %.4f'%d['ms_per_step'])"
  done
done
