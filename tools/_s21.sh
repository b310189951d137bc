cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 2500 python -m pytest tests/test_gpu_bh_device_tree.py tests/test_gpu_bh_chains.py tests/test_gpu_bh_warm_sort.py tests/test_gpu_bh.py tests/test_gpu_full_size_configs.py tests/test_gpu_randomized.py -q 2>&1 | tail -4 | cut -c1-300
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/st1 -o p --output-format csv -- python $OLDPWD/bench.py --workload bh --no-cpu-baseline --no-traffic --steps 40 --warmup 5 --steady-seconds 0 > $OLDPWD/$O/s21_bh1m.json 2>/dev/null); find /tmp/st1 -name '*kernel_stats.csv' -exec cp {} $O/s21_bh_kernel_stats_1m.csv \;
python tools/kstats.py $O/s21_bh_kernel_stats_1m.csv | head -16; python -c "
import json; d=json.load(open('$O/s21_bh1m.json')); print('ms/step', d['ms_per_step'])"
timeout 300 python bench.py --workload bh --bodies 10000 --theta 0.85 --no-cpu-baseline --no-traffic 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('10k ms/step %.4f' % d['ms_per_step'])"
