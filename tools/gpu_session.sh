#!/bin/bash
# One parametrised GPU session (replaces the round-1 gpu_session*.sh scratch scripts).  Run through gpurun from the repo root:
#   gpurun --timeout 1800 -- 'bash tools/gpu_session.sh smoke tests bench prof pmc power'
# Every stage writes under gpurun_out/<TAG>_* (TAG env, default r02); copy what is to be judged into profiles/.
#   smoke   __graft_entry__.smoke()
#   tests   pytest -m gpu (TESTS env narrows the selection, e.g. TESTS='tests/test_gpu_group.py')
#   bench   bench.py default line (N=1) + --gpus 2 through the group host (copy exchange when the box has one GPU)
#   bh      bench.py --workload bh (host tree, device tree)
#   prof    rocprofv3 --kernel-trace --stats of the bench command
#   pmc     six rocprofv3 --pmc passes of the bench command (summarise locally with tools/pmc_summary.py TAG)
#   power   tools/power_probe.py: board power during a >= 6 s K1 loop
#   shapes  launch-shape sweeps (sweep_shapes.py, shard-of-8 shape)
#   k1ab    K1 kernel variants 1/6/7 A/B (bench lines + board power)
#   cfgs    bench.py on BASELINE configs #2 and #5 (one GPU), the 2-D kernel, the bit-exact mode
#   fuzz    the four hand-run fuzz campaigns (tests/fuzz_*.py)
#   ubench  instruction-issue microbenchmarks
#   strict  bit-exact all-pairs kernels: kernel sweep by size + PMC summaries
#   xlat / verify8 / fold / frames / fallback   exchange floor + scaling bound; the self-validating 8-engine line; device tree with the
#           reference fold (bit-equality, ms per step, kernel trace); the level-1 frame loop; hand-over rate of long runs
#   walk    round 4: the three fast Barnes-Hut walks (nodes / child groups compiled / child groups hand-scheduled) -- A/B table,
#           kernel trace + PMC counters at 1 M and 10 000 bodies; small: per-kernel trace of the reference's own scene;
#           dry: bench.py --gpus 8 --dry-run (first contact with a multi-GPU node)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG="${TAG:-r04}"
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
for stage in "$@"; do
  echo "== $stage $(date +%T)"
  case "$stage" in
    smoke) timeout 600 python __graft_entry__.py smoke > $O/${TAG}_smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/${TAG}_smoke.log ;;
    tests) timeout ${TEST_TIMEOUT:-3000} python -m pytest ${TESTS:-tests} -m gpu -q -x > $O/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/${TAG}_pytest_gpu.log; tail -5 $O/${TAG}_pytest_gpu.log ;;
    bench)
      timeout 900 python bench.py > $O/${TAG}_bench_n1.json 2> $O/${TAG}_bench_n1.err; echo "bench rc=$?"; cat $O/${TAG}_bench_n1.json
      if [ "$(python -c 'import rust_exp_amd as r; print(r.device_count())' 2>/dev/null)" -ge 2 ]; then
        timeout 900 python bench.py --gpus 2 > $O/${TAG}_bench_group2.json 2> $O/${TAG}_bench_group2.err
      else
        NBX_GROUP_EXCHANGE=copy timeout 900 python bench.py --gpus 2 --no-cpu-baseline > $O/${TAG}_bench_group2_one_gpu_copy.json 2> $O/${TAG}_bench_group2.err
      fi
      echo "group bench rc=$?" ;;
    bh)
      timeout 900 python bench.py --workload bh > $O/${TAG}_bench_bh_default.json 2> $O/${TAG}_bench_bh_default.err; echo "bh rc=$?"; cat $O/${TAG}_bench_bh_default.json
      timeout 900 python bench.py --workload bh --bh-tree host --no-cpu-baseline > $O/${TAG}_bench_bh_host.json 2> $O/${TAG}_bench_bh_host.err
      timeout 900 python bench.py --workload bh --bh-tree device --no-cpu-baseline > $O/${TAG}_bench_bh_device.json 2> $O/${TAG}_bench_bh_device.err
      timeout 900 python bench.py --workload bh --bodies 10000 --theta 0.85 --no-cpu-baseline > $O/${TAG}_bench_bh_10000.json 2> $O/${TAG}_bench_bh_10000.err ;;
    prof)
      (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OLDPWD/$O/${TAG}_prof -o p --output-format csv -- python $OLDPWD/bench.py --no-cpu-baseline --no-traffic > $OLDPWD/$O/${TAG}_prof_bench.json 2> $OLDPWD/$O/${TAG}_prof.err)
      find $O/${TAG}_prof -name '*kernel_stats.csv' -exec cp {} $O/${TAG}_bench_kernel_stats.csv \; ; head -5 $O/${TAG}_bench_kernel_stats.csv ;;
    pmc)
      i=0
      for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "GRBM_GUI_ACTIVE GRBM_COUNT" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
        name=$(echo fetch write sq1 sq2 grbm tcc | cut -d' ' -f$((i+1))); i=$((i+1))
        (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $grp -d $OLDPWD/$O/pmc_$name -o p --output-format csv -- python $OLDPWD/bench.py --no-cpu-baseline --no-traffic --steps 5 --warmup 1 > /dev/null 2> $OLDPWD/$O/pmc_$name.err)
        f=$(find $O/pmc_$name -name '*counter_collection.csv' | head -1); [ -n "$f" ] && cp "$f" $O/pmc_$name/p_counter_collection.csv
      done ;;
    k1pmc)  # round 6: K1's counters in the round they are quoted -- both sweeps of the default bench line
      # (k_force_smem_pkw<3,8,true> = the headline, <3,8,false> = general masses), one counter group per pass, no child runs
      i=0
      for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
        name=$(echo fetch write sq1 grbm | cut -d' ' -f$((i+1))); i=$((i+1))
        rm -rf $O/pmc_$name
        (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $grp -d $OLDPWD/$O/pmc_$name -o p --output-format csv -- python $OLDPWD/bench.py --no-cpu-baseline --no-traffic --no-companions --steady-seconds 0 --steps 5 --warmup 1 > /dev/null 2> $OLDPWD/$O/pmc_$name.err)
        f=$(find $O/pmc_$name -name '*counter_collection.csv' | head -1); [ -n "$f" ] && { cp -f "$f" $O/pmc_$name/p_counter_collection.csv 2>/dev/null; cp -f "$f" $O/${TAG}_pmc_${name}_counter_collection.csv; }
      done
      rm -rf $O/pmc_sq2 $O/pmc_tcc
      python tools/pmc_summary.py $TAG > $O/${TAG}_pmc_summary.log 2>&1; cp profiles/${TAG}_pmc_summary.json $O/ 2>/dev/null; tail -5 $O/${TAG}_pmc_summary.log ;;
    power)
      timeout 300 python tools/power_probe.py 6 > $O/${TAG}_power_k1.json 2> $O/${TAG}_power_k1.err; echo "power rc=$?"; cut -c1-1500 $O/${TAG}_power_k1.json
      timeout 300 python tools/power_probe.py 6 --variant 1 > $O/${TAG}_power_k1_variant1.json 2>> $O/${TAG}_power_k1.err
      ls /sys/class/drm/card*/device/hwmon/hwmon*/ > $O/${TAG}_hwmon_ls.txt 2>&1
      rocm-smi --showpower --showclocks --showmaxpower > $O/${TAG}_rocm_smi.txt 2>&1 ;;
    shapes)
      timeout 1200 python tools/sweep_shapes.py > $O/${TAG}_shapes.log 2>&1
      for v in 6 ; do for s in 0 32 64 128; do   # (round 1 swept variant 5 here; removed in round 5)
        timeout 300 python bench.py --shard-of 8 --variant $v --jsplit $s --no-cpu-baseline --no-traffic >> $O/${TAG}_shard_of_8.jsonl 2>> $O/${TAG}_shard.err
      done; done ;;
    k1ab)   # K1 variants A/B: bench line + board power per variant, at the headline size, config #2's size and the 8-way shard shape
      for v in 1 6 7; do   # (variant 5 of the round-2 A/B was removed in round 5)
        timeout 300 python bench.py --variant $v --no-cpu-baseline --no-traffic >> $O/${TAG}_k1ab_n262144.jsonl 2>> $O/${TAG}_k1ab.err
        timeout 300 python bench.py --variant $v --n 65536 --no-cpu-baseline --no-traffic >> $O/${TAG}_k1ab_n65536.jsonl 2>> $O/${TAG}_k1ab.err
        timeout 300 python bench.py --variant $v --shard-of 8 --no-cpu-baseline --no-traffic >> $O/${TAG}_k1ab_shard_of_8.jsonl 2>> $O/${TAG}_k1ab.err
        timeout 300 python tools/power_probe.py 6 --variant $v > $O/${TAG}_power_k1_variant$v.json 2>> $O/${TAG}_k1ab.err
      done
      python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/*_k1ab_*.jsonl")):
    for ln in open(f):
        d=json.loads(ln); print(f.split("/")[-1], d["config"]["launch"], "%.3e"%d["value"], "frac %.4f"%d["roofline"]["frac"], "k_ms %.3f"%d["roofline"]["kernel_avg_ms"], "step %.3f"%d["ms_per_step"])
for f in sorted(glob.glob("gpurun_out/*_power_k1_variant*.json")):
    d=json.load(open(f)); r=d.get("rocm-smi",{}); print(f.split("/")[-1], "%.3e"%d["interactions_per_s"], r.get("avg_w"), r.get("joules_per_interaction"))
PY
      ;;
    k1sweep)   # source-split sweep of the wave-split kernels at the three shapes that matter
      for v in 6 7; do
        for sp in 2 4 8 16 32; do timeout 300 python bench.py --variant $v --jsplit $sp --no-cpu-baseline --no-traffic >> $O/${TAG}_k1sweep_n262144.jsonl 2>> $O/${TAG}_k1sweep.err; done
        for sp in 2 4 8 16 32 64; do timeout 300 python bench.py --variant $v --jsplit $sp --n 65536 --no-cpu-baseline --no-traffic >> $O/${TAG}_k1sweep_n65536.jsonl 2>> $O/${TAG}_k1sweep.err; done
        for sp in 8 16 32 64 128; do timeout 300 python bench.py --variant $v --jsplit $sp --shard-of 8 --no-cpu-baseline --no-traffic >> $O/${TAG}_k1sweep_shard_of_8.jsonl 2>> $O/${TAG}_k1sweep.err; done
      done
      python - <<'PY' | tee gpurun_out/${TAG}_k1_wave_split_sweep.txt
import json,glob
for f in sorted(glob.glob("gpurun_out/*_k1sweep_*.jsonl")):
    for ln in open(f):
        d=json.loads(ln); print(f.split("/")[-1], d["config"]["launch"], "%.3e"%d["value"], "frac %.4f"%d["roofline"]["frac"], "k_ms %.3f"%d["roofline"]["kernel_avg_ms"], "step %.3f"%d["ms_per_step"])
PY
      ;;
    cfgs)   # the other BASELINE configs on one GPU + the 2-D kernel on the headline size
      timeout 600 python bench.py --bodies 65536 --steps 20 --warmup 3 --no-traffic > $O/${TAG}_bench_cfg2_20_steps.json 2>> $O/${TAG}_cfgs.err
      timeout 600 python bench.py --bodies 65536 --steps 400 --warmup 100 --no-traffic > $O/${TAG}_bench_cfg2.json 2>> $O/${TAG}_cfgs.err
      timeout 600 python bench.py --workload two_galaxies --bodies 524288 --source-bits 16 --cpu-seconds 4 --no-general-masses > $O/${TAG}_bench_cfg5_1gpu.json 2>> $O/${TAG}_cfgs.err
      timeout 600 python bench.py --workload two_galaxies --bodies 524288 --no-traffic --no-cpu-baseline > $O/${TAG}_bench_cfg5_1gpu_fp32.json 2>> $O/${TAG}_cfgs.err
      timeout 600 python bench.py --dim 2 --no-traffic --no-cpu-baseline > $O/${TAG}_bench_n1_dim2.json 2>> $O/${TAG}_cfgs.err
      timeout 600 python bench.py --workload stable_orbits --no-traffic --no-cpu-baseline > $O/${TAG}_bench_stable_orbits_262144.json 2>> $O/${TAG}_cfgs.err
      for m in strict; do for nb in 10000 65536 262144; do timeout 600 python bench.py --mode $m --bodies $nb --no-traffic --no-cpu-baseline >> $O/${TAG}_bench_strict.jsonl 2>> $O/${TAG}_cfgs.err; done; done
      python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/*_bench_cfg*.json")+glob.glob("gpurun_out/*_bench_n1_dim2.json")+glob.glob("gpurun_out/*_bench_stable_orbits_*.json")):
    d=json.load(open(f)); print(f.split("/")[-1], d["config"]["workload"], "%.3e"%d["value"], "frac %.4f"%d["roofline"]["frac"], d["config"]["launch"])
for ln in open(glob.glob("gpurun_out/*_bench_strict.jsonl")[0]):
    d=json.loads(ln); print("strict", d["config"]["bodies"], "%.3e"%d["value"], "frac %.4f"%d["roofline"]["frac"], d["config"]["launch"])
PY
      ;;
    fuzz)   # hand-run campaigns against the oracle / the bit-exact kernels (FUZZ_N cases each, default 200)
      N=${FUZZ_N:-200}
      { timeout 3000 python tests/fuzz_strict.py 0 $N; timeout 3000 python tests/fuzz_fast.py 0 $N; timeout 3000 python tests/fuzz_api.py 0 $((N/4)); timeout 3000 python tests/fuzz_group.py 0 $((N/2)); } > $O/${TAG}_fuzz.txt 2>&1
      tail -30 $O/${TAG}_fuzz.txt ;;
    ubench) for u in valu banks chain sort; do [ -x tools/ubench_$u ] || hipcc --offload-arch=gfx950 -O3 tools/ubench_$u.hip -o tools/ubench_$u; done
      tools/ubench_valu > $O/${TAG}_ubench_valu.txt 2>&1; tools/ubench_banks > $O/${TAG}_ubench_banks.txt 2>&1; tools/ubench_chain > $O/${TAG}_ubench_chain.txt 2>&1; tools/ubench_sort > $O/${TAG}_ubench_sort.txt 2>&1 ;;
    strict) # bit-exact all-pairs kernels: sweep of the three kernels by size, PMC of the default at three sizes
      TAG=$TAG bash tools/sweep_strict_kernels.sh > /dev/null 2>&1; cat $O/${TAG}_strict_kernel_sweep.txt
      for b in 10000 65536 262144; do rm -rf $O/stpmc_*; TAG=$TAG BODIES=$b bash tools/strict_pmc.sh > /dev/null 2>&1; done; rm -rf $O/stpmc_* ;;
    xlat)   # software floor of the per-step exchange on one GPU + the scaling bound built on it
      timeout 1200 python tools/exchange_latency.py > $O/${TAG}_exchange_latency.json 2> $O/${TAG}_exchange_latency.err; echo "xlat rc=$?"; cut -c1-1200 $O/${TAG}_exchange_latency.json
      mkdir -p profiles; cp $O/${TAG}_exchange_latency.json profiles/${TAG}_exchange_latency.json
      timeout 1500 python tools/scale_model.py > $O/${TAG}_scaling_bound.json 2> $O/${TAG}_scaling_bound.err; echo "scale rc=$?"; cut -c1-600 $O/${TAG}_scaling_bound.json ;;
    verify8)   # the self-validating multi-GPU line on whatever this box has (8 engines share one GPU through peer copies otherwise)
      if [ "$(python -c 'import rust_exp_amd as r; print(r.device_count())' 2>/dev/null)" -ge 8 ]; then X=""; else X="NBX_GROUP_EXCHANGE=copy"; fi
      env $X timeout 900 python bench.py --gpus 8 --verify --no-cpu-baseline > $O/${TAG}_bench_group8_verify.json 2> $O/${TAG}_bench_group8_verify.err; echo "verify8 rc=$?"; cut -c1-800 $O/${TAG}_bench_group8_verify.json
      env $X NBX_GROUP_ENQUEUE=threads timeout 900 python bench.py --gpus 8 --verify --no-cpu-baseline > $O/${TAG}_bench_group8_verify_threads.json 2>> $O/${TAG}_bench_group8_verify.err; echo "verify8 threads rc=$?" ;;
    fold)   # device quadtree with the reference fold vs exact sums vs host tree: bit-equality + ms per step by size; kernel trace
      timeout 900 python tools/bh_fold_probe.py 0.85 > $O/${TAG}_bh_fold_probe.jsonl 2> $O/${TAG}_bh_fold_probe.err; cut -c1-60,130-420 $O/${TAG}_bh_fold_probe.jsonl
      bash tools/prof_bh_fold.sh > /dev/null 2>&1
      for n in 10000 65536; do cp $O/prof_fold_$n/p_kernel_stats.csv $O/${TAG}_bh_fold_kernel_stats_$n.csv; done ;;
    frames) # the reference's frame loop through the six nb_* symbols (defaults; host tree + host draw; exact-sum device tree)
      timeout 600 python tools/frame_loop.py > $O/${TAG}_frame_loop_level1.txt 2>&1; cat $O/${TAG}_frame_loop_level1.txt
      NB_BH_TREE=host NB_DRAW=host timeout 600 python tools/frame_loop.py > $O/${TAG}_frame_loop_level1_host_tree_host_draw.txt 2>&1
      NB_BH_FOLD=exact timeout 600 python tools/frame_loop.py > $O/${TAG}_frame_loop_level1_exact_fold.txt 2>&1; cat $O/${TAG}_frame_loop_level1_exact_fold.txt ;;
    fallback) # how often the reference-fold device tree hands a step to the host build over long runs of the reference's own scenes
      timeout 1500 python tools/bh_fallback_rate.py stable_orbits:10000 random_disk:10000 random_disk:2000 random_disk:20000 random_disk:30000 stable_orbits:65536 random_disk:65536 > $O/${TAG}_bh_fallback_rate.jsonl 2> $O/${TAG}_bh_fallback_rate.err
      NBX_BH_BACKOFF_MAX=0 timeout 600 python tools/bh_fallback_rate.py random_disk:65536 | sed 's/^{/{"note": "NBX_BH_BACKOFF_MAX=0: every step tries the device first", /' >> $O/${TAG}_bh_fallback_rate.jsonl
      cut -c1-200 $O/${TAG}_bh_fallback_rate.jsonl ;;
    walk)   # the fast Barnes-Hut walks side by side
      timeout 900 python tools/bh_walk_ab.py > $O/${TAG}_bh_walk_ab.jsonl 2> $O/${TAG}_bh_walk_ab.err; cut -c1-150 $O/${TAG}_bh_walk_ab.jsonl
      bash tools/bh_walk_pmc.sh 1048576 0.5 $TAG > $O/${TAG}_bh_walk_pmc_1m.log 2>&1
      bash tools/bh_walk_pmc.sh 10000 0.85 $TAG > $O/${TAG}_bh_walk_pmc_10k.log 2>&1
      python tools/bh_walk_table.py $TAG | tee $O/${TAG}_bh_walk_table.txt ;;
    small)  # the reference's own scene (10 000 bodies): per-kernel trace of the step, both tree classes; front-end A/B
      bash tools/prof_bh_small.sh 10000 $TAG | tee $O/${TAG}_bh_small_kernels.txt
      for f in 16384 0; do echo "NBX_SMALL_FRONT_MAX=$f"; NBX_SMALL_FRONT_MAX=$f python tools/bh_walk_ab.py 2000:0.85 10000:0.85 16384:0.85 2>&1 | grep '"groups"' | cut -c1-200; done | tee $O/${TAG}_bh_small_front_ab.txt ;;
    dry)    # first contact with a multi-GPU node, rehearsed on whatever this box has
      if [ "$(python -c 'import rust_exp_amd as r; print(r.device_count())' 2>/dev/null)" -ge 8 ]; then X=""; else X="NBX_GROUP_EXCHANGE=copy"; fi
      env $X timeout 600 python bench.py --gpus 8 --dry-run --no-cpu-baseline > $O/${TAG}_bench_group8_dry_run.json 2> $O/${TAG}_bench_group8_dry_run.err; echo "dry rc=$?"; cut -c1-1000 $O/${TAG}_bench_group8_dry_run.json
      NBX_GROUP_RCCL_FAIL=init timeout 600 python bench.py --gpus 2 --dry-run --no-cpu-baseline > $O/${TAG}_bench_group2_dry_run_rccl_failure.json 2> $O/${TAG}_bench_group2_dry_run_rccl_failure.err; echo "dry (simulated RCCL failure) rc=$?"; cut -c1-600 $O/${TAG}_bench_group2_dry_run_rccl_failure.json ;;
    *) echo "unknown stage $stage" ;;
  esac
done
echo "== done $(date +%T)"
