cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 3000 python tests/fuzz_fast.py 60000 3000 > $O/r06_fuzz_fast.txt 2>&1; tail -3 $O/r06_fuzz_fast.txt | cut -c1-600
timeout 3000 python tests/fuzz_strict.py 60000 300 > $O/r06_fuzz_strict.txt 2>&1; tail -2 $O/r06_fuzz_strict.txt | cut -c1-400
timeout 3000 python tests/fuzz_api.py 60000 100 > $O/r06_fuzz_api.txt 2>&1; tail -2 $O/r06_fuzz_api.txt | cut -c1-400
timeout 3000 python tests/fuzz_group.py 60000 500 > $O/r06_fuzz_group.txt 2>&1; tail -2 $O/r06_fuzz_group.txt | cut -c1-400
