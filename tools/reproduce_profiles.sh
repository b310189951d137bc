#!/bin/bash
# Regenerates what profiles/ holds for one round on an MI355X box (run through gpurun from the repo root):
#   gpurun --timeout 5000 -- 'TAG=r02 bash tools/reproduce_profiles.sh'   ;   then, locally:  python tools/pmc_summary.py r02
# and copy the gpurun_out/<TAG>_* files named in profiles/README.md into profiles/.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
T=${TAG:-r02}
bash tools/gpu_session.sh smoke tests bench bh prof pmc power k1ab k1sweep cfgs shapes ubench strict fuzz
# round 3: the self-validating 8-engine line, the exchange floor + scaling bound, the reference-fold device tree (probe, kernel
# stats), the level-1 frame loops, the hand-over rates of long runs; then the soak and the crossover tables
bash tools/gpu_session.sh verify8 xlat fold frames fallback
python tools/soak.py 100000 > gpurun_out/${T}_soak.txt 2>&1
bash tools/bh_side_stream_crossover.sh > gpurun_out/${T}_bh_side_stream_crossover.txt 2>&1
bash tools/bh_walk_pmc.sh > /dev/null
(cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats -d /tmp/nbx_bh_stats -o p --output-format csv -- python $OLDPWD/bench.py --workload bh --no-cpu-baseline --no-traffic --steps 10 > /dev/null 2>&1); find /tmp/nbx_bh_stats -name '*kernel_stats.csv' -exec cp {} gpurun_out/${T}_bh_kernel_stats.csv \;
python tools/bh_device_tree_probe.py > gpurun_out/${T}_bh_device_tree_probe.log 2>&1; cp gpurun_out/bh_device_tree_probe.json gpurun_out/${T}_bh_device_tree_probe.json
python tools/scale_model.py > gpurun_out/${T}_scaling_expectation.json
python tools/bench_bh.py > gpurun_out/${T}_bench_bh_tool.json 2> gpurun_out/bench_bh.err
python tools/frame_loop.py > gpurun_out/${T}_frame_loop_level1.txt 2>&1
NB_BH_TREE=host NB_DRAW=host python tools/frame_loop.py > gpurun_out/${T}_frame_loop_level1_host_tree_host_draw.txt 2>&1
python tools/pcie_inclusive.py > gpurun_out/${T}_pcie_inclusive.json 2>&1
# round 4: the three fast walks side by side (A/B table, kernel trace, PMC), the reference's own scene kernel by kernel, the
# first-contact dry runs, the walk kernel's timeline, the sort shapes, the torch host with two gloo ranks on one GPU
bash tools/gpu_session.sh walk small dry
python tools/bh_walk_trace.py > gpurun_out/${T}_bh_walk_trace_1m.json 2> gpurun_out/${T}_bh_walk_trace.err
[ -x tools/ubench_sort_cfg ] || hipcc --offload-arch=gfx950 -O3 tools/ubench_sort_cfg.hip -o tools/ubench_sort_cfg
tools/ubench_sort_cfg > gpurun_out/${T}_ubench_sort_cfg.txt 2>&1
NBX_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --dry-run --bodies 65536 > gpurun_out/${T}_bench_torch2_gloo_one_gpu_dry_run.json 2> /dev/null
NBX_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 5 --warmup 1 --bodies 65536 --no-cpu-baseline > gpurun_out/${T}_bench_torch2_gloo_one_gpu.json 2> /dev/null
python tools/bh_union_model.py 1048576 > gpurun_out/${T}_bh_walk_union_model_n1048576.json 2> /dev/null   # (CPU: needs no GPU)
# round 5: the driver's line with its companions (configs #2, #4, #5), the Barnes-Hut line on the host tree (traffic of the wave walk
# alone), kernel stats of the 1 M-body step, the bucket-sort shapes, long runs of the warm sort (fallback rates), the frame loops.
# (The walk-split A/B -- profiles/r05_bh_walk_split_ab.jsonl, r05_bh_walk_trace_1m_split*.json -- was measured on a tree that still
#  had the experiment: commit cc66460, `NBX_WALK_SPLIT_PCT=25 python tools/bh_walk_trace.py`, `bash tools/bh_walk_split_ab.sh`.)
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${T}_bench_n1_companions.json 2> gpurun_out/${T}_bench.err
python bench.py --workload bh --bh-tree host --steps 10 --warmup 2 --steady-seconds 0 --no-accuracy > gpurun_out/${T}_bench_bh_host.json 2> /dev/null
(cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats -d /tmp/nbx_bh_stats5 -o p --output-format csv -- python $OLDPWD/bench.py --workload bh --no-cpu-baseline --no-traffic --steps 40 --warmup 5 --steady-seconds 0 > /dev/null 2>&1); find /tmp/nbx_bh_stats5 -name '*kernel_stats.csv' -exec cp {} gpurun_out/${T}_bh_kernel_stats_1m.csv \;
[ -x tools/ubench_bucket_sort ] || hipcc --offload-arch=gfx950 -O3 tools/ubench_bucket_sort.hip -o tools/ubench_bucket_sort
(for a in "1048576 800 0" "1048576 640 4" "1048576 640 4 10" "262144 640 4"; do tools/ubench_bucket_sort $a; done) > gpurun_out/${T}_ubench_bucket_sort.txt 2>&1
python tools/bh_warm_long_run.py > gpurun_out/${T}_bh_warm_long_run.jsonl 2>&1
NBX_INC_SORT=0 python tools/bh_warm_long_run.py plummer:1048576 random_disk:262144 > gpurun_out/${T}_bh_cold_long_run.jsonl 2>&1
NB_BH_FOLD=exact python tools/frame_loop.py > gpurun_out/${T}_frame_loop_level1_exact_fold.txt 2>&1
NBX_GROUP_EXCHANGE=copy python bench.py --gpus 8 --dry-run > gpurun_out/${T}_bench_group8_dry_run.json 2> /dev/null
bash tools/pmc_host_tree_walk.sh > gpurun_out/${T}_pmc_host_tree_walk.txt 2>&1
python tools/bh_dense_probe.py > gpurun_out/${T}_bh_dense_handover.jsonl 2>&1
bash tools/bh_build_valu.sh > gpurun_out/${T}_bh_step_issue_counters.json 2> /dev/null
(timeout 1500 python tests/fuzz_fast.py 700000 20000 2>&1 | tail -1; timeout 700 python tests/fuzz_strict.py 70000 400 2>&1 | tail -1; timeout 300 python tests/fuzz_group.py 70000 3000 2>&1 | tail -1; timeout 600 python tests/fuzz_api.py 70000 150 2>&1 | tail -1) > gpurun_out/${T}_fuzz_long2.txt 2>&1
# round 6: the default tree class by cost and its done-criteria (sizes, long runs, frame loop), the chain replay (dense models with the
# force error after 35 steps, the reference-fold class with pipelined steps), K1's counters of the round, the walk loop A/B, the fuzz
# campaigns (chains injected), the first-contact rehearsals once more
python tools/bh_sizes.py > gpurun_out/${T}_bh_sizes.jsonl 2> /dev/null
NBX_LONG_STEPS=1000 python tools/bh_warm_long_run.py random_disk:65536 stable_orbits:10000 random_disk:10000 > gpurun_out/${T}_bh_warm_long_run.jsonl 2> /dev/null
python tools/bh_dense_probe.py --accuracy 1048576 2097152 4194304 > gpurun_out/${T}_bh_dense_probe.jsonl 2> /dev/null
python tools/bh_reference_fold_async_probe.py > gpurun_out/${T}_bh_reference_fold_async_probe.txt 2>&1
bash tools/gpu_session.sh k1pmc
for r in 1 2 3; do for pipe in 0 1; do echo -n "pipe $pipe "; NBX_BH_WALK_PIPE=$pipe python bench.py --workload bh --no-cpu-baseline --no-traffic --steps 40 --warmup 5 --steady-seconds 0 2> /dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms/step %.4f traversal_ms %.4f' % (d['ms_per_step'], d['roofline']['kernel_avg_ms']))"; done; done > gpurun_out/${T}_bh_walk_pipelined_ab.txt
python tests/fuzz_fast.py 60000 3000 > gpurun_out/${T}_fuzz_fast.txt 2>&1; python tests/fuzz_strict.py 60000 300 > gpurun_out/${T}_fuzz_strict.txt 2>&1
python tests/fuzz_api.py 60000 100 > gpurun_out/${T}_fuzz_api.txt 2>&1; python tests/fuzz_group.py 60000 500 > gpurun_out/${T}_fuzz_group.txt 2>&1
