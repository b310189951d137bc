#!/bin/bash
# A/B of the wave walk's node records (VERDICT r02 next #5c): 32-byte records vs the compact 16-byte copy, with in-run traffic
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for n in 262144 1048576; do for rec in 32 16; do
  python bench.py --workload bh --bodies $n --bh-walk-records $rec --no-cpu-baseline --steady-seconds 0 --steps 30 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; ic=r.get('issue_counters') or {}
print(json.dumps({'bodies': $n, 'records': $rec, 'ms_per_step': round(d['ms_per_step'],4), 'eval_ms': round(d['ms_split']['bh_eval_kernel'],4), 'build_ms': round(d['ms_split']['tree_build'],4), 'traffic_MB': None if r['traffic'] is None else round(r['traffic']/1e6,1), 'valu_busy': ic.get('valu_busy_frac'), 'valu_per_visit': ic.get('valu_insts_per_wave_visit'), 'salu_per_visit': ic.get('salu_insts_per_wave_visit')}))"
done; done
