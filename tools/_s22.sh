cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 2800 python -m pytest tests -q -m gpu > $O/r06_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/r06_pytest_gpu.log | cut -c1-300
timeout 600 python __graft_entry__.py smoke > $O/r06_smoke.log 2>&1; tail -2 $O/r06_smoke.log
timeout 900 python bench.py > $O/r06_bench_n1.json 2> $O/r06_bench_n1.err; echo "bench rc=$?"; python -c "
import json; d=json.load(open('$O/r06_bench_n1.json')); r=d['roofline']; print(d['value'], r['frac'], {k:v for k,v in r.items() if k.startswith(('c0_','c4_'))})" | cut -c1-1500
