import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rust_exp_amd as rx
from rust_exp_amd.engine import NBX_STAT_BH_CLASS_SWITCHES, NBX_STAT_BH_FALLBACKS, NBX_STAT_BH_REFUSAL
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
e = rx.NBodyEngine(mode="fast"); e.set_bh_fold("reference"); e.set_bh_tree("device")
st = rx.plummer_sphere(n, dim=2)
e.set_particles(st["px"], st["py"], st["vx"], st["vy"], st["m"])
T = []
t00 = time.perf_counter()
for s in range(60):
    t0 = time.perf_counter(); e.step_barnes_hut(0.5, 0.01, 1); T.append((time.perf_counter() - t0) * 1e3)
e.synchronize()
print("total %.1f ms for 60 steps; switches %d fallbacks %d" % ((time.perf_counter() - t00) * 1e3, e.get_stat(NBX_STAT_BH_CLASS_SWITCHES), e.get_stat(NBX_STAT_BH_FALLBACKS)))
print(" ".join("%.2f" % t for t in T))
