cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
NBX_LONG_STEPS=1000 timeout 1500 python tools/bh_warm_long_run.py > $O/r06_bh_warm_long_run_all_scenes.jsonl 2> $O/r06_bh_warm_long_run_all_scenes.err; echo "long rc=$?"; cut -c1-330 $O/r06_bh_warm_long_run_all_scenes.jsonl
timeout 3000 python tests/fuzz_fast.py 400000 12000 > $O/r06_fuzz_long.txt 2>&1; tail -2 $O/r06_fuzz_long.txt | cut -c1-500
timeout 1200 python tests/fuzz_strict.py 40000 400 >> $O/r06_fuzz_long.txt 2>&1; tail -1 $O/r06_fuzz_long.txt
timeout 600 python tests/fuzz_group.py 40000 2000 >> $O/r06_fuzz_long.txt 2>&1; tail -1 $O/r06_fuzz_long.txt
NBX_GROUP_EXCHANGE=copy timeout 600 python bench.py --gpus 8 --dry-run --no-cpu-baseline > $O/r06_bench_group8_dry_run.json 2> /dev/null; echo "dry rc=$?"; cut -c1-300 $O/r06_bench_group8_dry_run.json
NBX_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --dry-run --bodies 65536 > $O/r06_bench_torch2_gloo_one_gpu_dry_run.json 2> /dev/null; echo "gloo dry rc=$?"
NBX_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 5 --warmup 1 --bodies 65536 --no-cpu-baseline > $O/r06_bench_torch2_gloo_one_gpu.json 2> /dev/null; echo "gloo bench rc=$?"; cut -c1-300 $O/r06_bench_torch2_gloo_one_gpu.json
NBX_GROUP_EXCHANGE=copy timeout 900 python bench.py --gpus 8 --verify --no-cpu-baseline > $O/r06_bench_group8_verify.json 2> /dev/null; echo "verify8 rc=$?"; cut -c1-300 $O/r06_bench_group8_verify.json
timeout 900 python tools/soak.py 30000 > $O/r06_soak.txt 2>&1; tail -3 $O/r06_soak.txt | cut -c1-300
