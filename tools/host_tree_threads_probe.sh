cd $GRAFT_REPO_ROOT
for t in 12 16 24 32; do
  for rep in 1 2; do
    NBX_HOST_THREADS=$t python bench.py --workload bh --bh-tree host --no-cpu-baseline --no-traffic --steady-seconds 0 --steps 40 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('threads $t', 'ms/step %.2f'%d['ms_per_step'], {k: round(v,2) for k,v in d['ms_split'].items() if k!='tree_nodes'})"
  done
done
cat /sys/fs/cgroup/cpu.max 2>/dev/null; nproc
