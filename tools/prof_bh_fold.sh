cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cat > /tmp/bhstep.py <<'PY'
import sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import rust_exp_amd as rx
n=int(sys.argv[1]); fold=sys.argv[2]
e=rx.NBodyEngine(); e.seed(5); e.stable_orbits(n,0.5,30.0)
e.set_bh_tree("device"); e.set_bh_fold(fold)
for _ in range(30): e.step_barnes_hut(0.85,0.01,1)
e.synchronize()
PY
for n in 10000 65536; do for fold in reference; do
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_fold_$n -o p --output-format csv -- python /tmp/bhstep.py $n $fold > /dev/null 2>&1
f=$(find $R/gpurun_out/prof_fold_$n -name '*kernel_stats.csv' | head -1); echo "== $n"; cut -d, -f1-4 "$f" | cut -c1-150 | head -24
done; done
