cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
NBX_BH_WALK_PIPE=1 timeout 1500 python -m pytest tests/test_gpu_bh_group_walk.py tests/test_gpu_bh.py tests/test_gpu_bh_device_tree.py -q -k "not resources" 2>&1 | tail -3 | cut -c1-300
for r in 1 2 3; do for pipe in 0 1; do
  NBX_BH_WALK_PIPE=$pipe timeout 300 python bench.py --workload bh --no-cpu-baseline --no-traffic --steps 40 --warmup 5 --steady-seconds 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('pipe $pipe', 'ms/step %.4f' % d['ms_per_step'], 'traversal_ms', r.get('kernel_avg_ms'), 'build', r.get('build_ms'))"
done; done | tee $O/r06_bh_walk_pipelined_ab.txt
for pipe in 0 1; do NBX_BH_WALK_PIPE=$pipe timeout 300 python bench.py --workload bh --bodies 10000 --theta 0.85 --no-cpu-baseline --no-traffic 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('10k pipe $pipe', 'ms/step %.4f' % d['ms_per_step'], 'traversal_ms', r.get('kernel_avg_ms'))"; done | tee -a $O/r06_bh_walk_pipelined_ab.txt
timeout 3000 python tests/fuzz_fast.py 60000 1500 > $O/r06_fuzz_fast.txt 2>&1; tail -4 $O/r06_fuzz_fast.txt | cut -c1-600
