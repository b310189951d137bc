cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python tools/bh_reference_fold_async_probe.py > $O/r06_bh_reference_fold_async_probe.txt 2>&1; tail -2 $O/r06_bh_reference_fold_async_probe.txt | cut -c1-600
timeout 1500 python -m pytest tests/test_gpu_bh_device_tree.py tests/test_gpu_bh_warm_sort.py tests/test_gpu_bh_chains.py -q 2>&1 | tail -3 | cut -c1-300
timeout 900 python tools/bh_sizes.py > $O/r06_bh_sizes.jsonl 2> $O/r06_bh_sizes.err; echo "sizes rc=$?"
timeout 1500 python tools/bh_dense_probe.py --accuracy 1048576 2097152 4194304 > $O/r06_bh_dense_probe.jsonl 2> $O/r06_bh_dense_probe.err; echo "dense rc=$?"; cut -c1-1200 $O/r06_bh_dense_probe.jsonl
timeout 3000 python tests/fuzz_fast.py 60000 1500 > $O/r06_fuzz_fast.txt 2>&1; tail -4 $O/r06_fuzz_fast.txt | cut -c1-600
