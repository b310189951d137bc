import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import rust_exp_amd as rx
e=rx.NBodyEngine(); e.seed(5); e.random_disk(65536)
for i in range(60): e.step_barnes_hut(0.85,0.01,1)
e.synchronize()
print("fallbacks", e.get_option(rx.engine.NBX_OPT_BH_FALLBACKS))
