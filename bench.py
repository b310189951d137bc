#!/usr/bin/env python3
"""bench.py -- body-pair interactions/s of the brute-force N-body step (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One "step" = one nb_step_brute_force (rs-src/nbody.rs:106-162): all-pairs force + kick-drift over
the whole synthetic system.  Workload: N = 262 144-body Plummer sphere (3-D float4 kernel,
17 algorithmic flops / interaction), the configuration the metric is quoted on; it fits one GPU.
With N GPUs the SAME system is sharded as slabs of targets (strong scaling) with one all-gather of
(x,y,z,m) per step.  Inputs are resident in HBM before the timed region.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOPS_PER_INTERACTION = 17   # 3 sub, 3 mul + 2 add, 1 add eps, 1 mul, 1 div, 3 mul, 3 add
DT = 0.01                    # RustNBodyExperiment.hs:45


def effective_cores():
    """Cores this process may actually use: scheduler affinity, capped by a cgroup CPU quota if there is one."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, int(q / p + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


def cpu_baseline(st, seconds):
    """Oracle (CPU restatement of nbody.rs:132-144) on the host cores: a bounded i-slice of the same
    workload (work per target is uniform), threads = all cores with the reference's slab split."""
    from oracle import binding as ob

    n = len(st["px"])
    cores = effective_cores()
    p = ob.particles(st["px"], st["py"], st["vx"], st["vy"], st["m"])   # the reference law is 2-D: z ignored
    # calibrate
    ni = min(n, 16 * cores)
    t0 = time.perf_counter(); ob.brute_forces(p, 0, ni, nthreads=cores); t1 = time.perf_counter()
    rate = ni * (n - 1) / max(t1 - t0, 1e-9)
    ni = int(min(n, max(ni, rate * seconds / (n - 1))))
    ni -= ni % cores or 0
    ni = max(ni, cores)
    t0 = time.perf_counter(); ob.brute_forces(p, 0, ni, nthreads=cores); t1 = time.perf_counter()
    mt = ni * (n - 1) / (t1 - t0)
    n1 = max(1, min(n, int(ni / cores)))
    t0 = time.perf_counter(); ob.brute_forces(p, 0, n1, nthreads=1); t2 = time.perf_counter()
    st1 = n1 * (n - 1) / (t2 - t0)
    return {
        "value": mt, "unit": "interactions/s", "cores": cores, "kind": "port", "logical_cpus": os.cpu_count(),
        "sample": f"first {ni} targets x {n} sources (2-D reference law), {cores} threads, reference slab split",
        "single_thread_value": st1,
        "single_thread_sample": f"first {n1} targets x {n} sources, 1 thread (the reference's brute force is single-threaded)",
    }


def cpu_baseline_barnes_hut(st, theta, dt, threads, reps):
    """The CPU-baseline leg of the Barnes-Hut measurements (tools/bench_bh.py): the oracle's nb_step_barnes_hut
    (nbody.rs:186-480: serial tree build + `threads` traversal workers) timed on the host cores. Median ms per step."""
    from oracle import binding as ob

    p = ob.particles(st["px"], st["py"], st["vx"], st["vy"], st["m"])
    ts, rc = [], 0
    for _ in range(reps):
        t0 = time.perf_counter(); rc = ob.step_barnes_hut(p, theta, dt, threads); ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e3, rc


def _claim_stdout():
    """RCCL (and other C libraries) print banners to the C stdout ("RCCL version : ..." at communicator
    creation, flushed at exit). The contract is ONE JSON line on stdout, so keep a private handle to the real
    stdout for that line and point file descriptor 1 at stderr for everything else in this process."""
    sys.stdout.flush()
    real = os.dup(1)
    os.dup2(2, 1)
    return real


def main():
    real_stdout = _claim_stdout()
    # multi-process GPU work on this stack needs dmabuf IPC (exported by the driver; harmless to restate)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n", type=int, default=262144)
    ap.add_argument("--dim", type=int, default=3)
    ap.add_argument("--mode", default="fast")
    ap.add_argument("--jsplit", type=int, default=0)
    ap.add_argument("--bpt", type=int, default=0)
    ap.add_argument("--variant", type=int, default=-1)
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", default="plummer", choices=["plummer", "two_galaxies"])
    ap.add_argument("--source-bits", type=int, default=32, choices=[16, 32],
                    help="16 = fp16 source copy / fp32 accumulators (BASELINE config #5)")
    ap.add_argument("--torch-path", action="store_true",
                    help="use the multi-GPU code path (torch-owned buffer + torch stream) even on one GPU")
    args = ap.parse_args()

    import rust_exp_amd as rx

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
    n = args.n
    st = rx.plummer_sphere(n, dim=args.dim) if args.workload == "plummer" else rx.two_galaxies(n)

    if world == 1 and not args.torch_path:
        eng = rx.NBodyEngine(device=0, mode=args.mode)
        eng.set_source_precision(args.source_bits)
        eng.set_launch(jsplit=args.jsplit, bodies_per_thread=args.bpt, variant=args.variant)
        eng.set_particles(st["px"], st["py"], st["vx"], st["vy"], st["m"], st["pz"], st["vz"])

        def step():
            eng.step_brute_force(DT)

        def sync():
            eng.synchronize()

        def barrier():
            pass

        engine = eng
    else:
        import torch
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # one rank per GPU. (NBX_DIST_BACKEND=gloo lets several ranks share one GPU: the builder's only way to
        # run this multi-rank path on a single-GPU box; the driver's launch uses the default, nccl = RCCL.)
        backend = os.environ.get("NBX_DIST_BACKEND", "nccl")
        local_rank = local_rank % max(1, torch.cuda.device_count())
        torch.cuda.set_device(local_rank)
        if world > 1:
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            else:
                dist.init_process_group(backend)
        slab = rx.sharded.TorchSlabEngine(local_rank, mode=args.mode, source_half=args.source_bits == 16)
        slab.eng.set_launch(jsplit=args.jsplit, bodies_per_thread=args.bpt, variant=args.variant)
        sim = rx.ShardedNBody(slab)
        sim.set_particles(st)

        def step():
            sim.step_brute_force(DT)

        def sync():
            torch.cuda.synchronize()

        def barrier():
            if world > 1:
                dist.barrier()

        engine = slab.eng

    # inputs resident in HBM and every buffer allocated before anything is timed, whatever --warmup says: a force-only
    # evaluation (no state change) uploads the state and sizes the work buffers; the W warm-up steps follow
    engine.forces(0.0)
    if world > 1:
        sim._exchange()   # communicator set-up outside the timed region too (re-gathers the initial positions: a no-op on the data)
    for _ in range(args.warmup):
        step()
    sync()
    engine.profile(True)
    engine.profile_reset()
    barrier(); sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync(); barrier()
    elapsed = time.perf_counter() - t0
    k_ms, k_cnt = engine.profile_read(rx.NBX_K_FORCE)
    i_ms, _ = engine.profile_read(rx.NBX_K_INTEGRATE)
    engine.profile(False)
    if world > 1:
        import torch
        import torch.distributed as dist

        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        info = rx.device_info(local_rank)
        interactions_per_step = float(n) * float(n - 1)
        value = interactions_per_step * args.steps / elapsed
        lo, hi = engine.slab()
        launch = engine.last_launch()
        # dominant kernel: K1 force tiles. Algorithmic work of ONE launch on this rank:
        inter_per_launch = float(hi - lo) * float(n - 1)
        flops_per_inter = FLOPS_PER_INTERACTION if launch["dim"] == 3 else 12
        k_avg_s = (k_ms / max(k_cnt, 1)) * 1e-3
        achieved = inter_per_launch * flops_per_inter / k_avg_s / 1e12
        peak = info["peak_fp32_flops"] / 1e12
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tfile) and world == 1 and n == 262144:
            try:
                traffic = json.load(open(tfile)).get("k_force_tile_hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "body-pair interactions/s at N=262144 (brute-force O(N^2) step)" if n == 262144
                      else f"body-pair interactions/s at N={n} (brute-force O(N^2) step)",
            "value": value, "unit": "interactions/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32" if args.source_bits == 32 else "f32 (fp16 source copy)", "data": "synthetic",
            "config": {"workload": f"{'plummer_sphere' if args.workload == 'plummer' else 'two_galaxies'}_N{n}_brute_force_dim{launch['dim']}_dt{DT}"
                                   + ("_fp16sources" if args.source_bits == 16 else ""),
                       "bodies": n, "seed": "0x5EED0001", "force_mode": args.mode,
                       "sharding": f"slab x{world}, one all-gather of (x,y,z,m) per step" if world > 1 else "single GPU",
                       "launch": launch},
            "roofline": {"bound": "valu_fp32", "bound_contract_class": "mfma (dense fp32 peak: the f32 MFMA rate equals the fp32 vector rate, "
                                                                  "157.3 TFLOP/s; no MFMA is used)", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved / peak, "traffic": traffic,
                         "kernel": {0: "k_force_tile", 1: "k_force_tile_pk", 2: "k_force_smem", 3: "k_force_tile_pk", 4: "k_force_tile_pkb", 5: "k_force_smem_pk", 16: "k_force_tile_pk_h", -1: "k_force_strict<1>", -2: "k_force_strict<2>", -4: "k_force_strict<4>"}.get(launch["variant"], "k_force"), "kernel_avg_ms": k_avg_s * 1e3, "kernel_launches": k_cnt,
                         "flops_per_interaction": flops_per_inter,
                         "interactions_per_launch": inter_per_launch,
                         "hbm_algorithmic_bytes_per_launch": 16.0 * n + 16.0 * (hi - lo) * launch["jsplit"],
                         "note": "VALU-bound path (arithmetic intensity ~1e5 flop/B): peak = CUs*clock*256 flop/clk "
                                 "(fp32 vector FMA roofline, = the f32 MFMA rate); HBM is not the bound"},
            "integrate_kernel_avg_ms": i_ms / max(k_cnt, 1),
            "device": info["name"], "arch": info["arch"], "compute_units": info["compute_units"],
            "clock_khz": info["clock_khz"],
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(st, args.cpu_seconds)
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if world > 1:
        import torch.distributed as dist

        dist.destroy_process_group()


if __name__ == "__main__":
    main()
