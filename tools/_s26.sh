cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out; mkdir -p $O
timeout 3000 python tests/fuzz_fast.py 400000 12000 > $O/r06_fuzz_long.txt 2>&1; grep FAIL $O/r06_fuzz_long.txt | cut -c1-300; tail -1 $O/r06_fuzz_long.txt | cut -c1-500
timeout 1200 python tests/fuzz_fast.py 60000 3000 > $O/r06_fuzz_fast.txt 2>&1; tail -1 $O/r06_fuzz_fast.txt | cut -c1-400
