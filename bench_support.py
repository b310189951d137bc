"""bench_support.py -- parts of bench.py that measure or drive but decide nothing about the line: the in-run rocprofv3 counter
passes, the rocm-smi sampler and steady-state loop, and the three hosts (one engine / the library's single-process group /
one process per GPU under torch.distributed) behind one interface.  No oracle here: the CPU-baseline legs stay in bench.py."""
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
DT = 0.01                    # RustNBodyExperiment.hs:45


def measure_traffic(argv_tail, kernel_substr, timeout=120):
    """HBM bytes per launch of the dominant kernel, measured NOW on this box with rocprofv3 PMC counters exactly as
    /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes
    (TCC slots), unit KiB, FETCH_SIZE doubled on gfx950 (it reports 1/2 of the bytes of 16-B/lane coalesced reads),
    WRITE_SIZE as reported. Each pass re-runs this script as a short child (--traffic-child). Returns (dict | None)."""
    exe = shutil.which("rocprofv3")
    if not exe:
        return None
    import csv

    out = {}
    tmp = tempfile.mkdtemp(prefix="nbx_pmc_", dir="/tmp")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, counter)
            cmd = [exe, "--kernel-trace", "--pmc", counter, "-d", d, "-o", "p", "--output-format", "csv", "--",
                   sys.executable, os.path.join(ROOT, "bench.py"), "--traffic-child"] + argv_tail
            env = dict(os.environ, TMPDIR="/tmp")
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
            if r.returncode != 0:
                return None
            subs = (kernel_substr,) if isinstance(kernel_substr, str) else tuple(kernel_substr)
            per = {k: [] for k in subs}     # several kernels (e.g. conversion + walk of one traversal): per-launch means are summed
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if row["Counter_Name"] != counter:
                        continue
                    for k in subs:
                        if k in row["Kernel_Name"]:
                            per[k].append(float(row["Counter_Value"]))
                            break
            if not per[subs[0]]:
                return None
            out[counter] = float(sum(np.mean(v) for v in per.values() if v))
            out[counter + "_launches"] = len(per[subs[0]])
            out[counter + "_by_kernel"] = {k: float(np.mean(v)) for k, v in per.items() if v}
        read_b = 2.0 * out["FETCH_SIZE"] * 1024.0
        write_b = out["WRITE_SIZE"] * 1024.0
        return {"bytes_per_launch": read_b + write_b, "read_bytes": read_b, "write_bytes": write_b,
                "launches_sampled": out["FETCH_SIZE_launches"],
                "read_bytes_by_kernel": {k: 2048.0 * v for k, v in out["FETCH_SIZE_by_kernel"].items()},
                "write_bytes_by_kernel": {k: 1024.0 * v for k, v in out["WRITE_SIZE_by_kernel"].items()},
                "how": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) on a 3-step child run of this "
                       "command, this box, this run; KiB -> B, FETCH_SIZE x2 (gfx950 correction, MI355X_MICROARCH.md HBM)"}
    except (subprocess.TimeoutExpired, OSError, KeyError, ValueError):
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def measure_counters(argv_tail, kernel_substr, groups, timeout=180):
    """Average per-launch value of every counter in `groups` (one rocprofv3 --pmc pass per group: counters of one block share
    its slots) for the kernels whose name contains `kernel_substr`, on a 3-step child run of this command. dict | None."""
    exe = shutil.which("rocprofv3")
    if not exe:
        return None
    import csv

    out = {}
    tmp = tempfile.mkdtemp(prefix="nbx_pmc_", dir="/tmp")
    try:
        for gi, group in enumerate(groups):
            d = os.path.join(tmp, f"g{gi}")
            cmd = [exe, "--kernel-trace", "--pmc"] + group.split() + ["-d", d, "-o", "p", "--output-format", "csv", "--",
                   sys.executable, os.path.join(ROOT, "bench.py"), "--traffic-child"] + argv_tail
            r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                               timeout=timeout)
            if r.returncode != 0:
                return None
            vals = {}
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if kernel_substr in row["Kernel_Name"]:
                        vals.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
            for k, v in vals.items():
                out[k] = float(np.mean(v))
                out[k + "_launches"] = len(v)
        return out or None
    except (subprocess.TimeoutExpired, OSError, KeyError, ValueError):
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


class RocmSmiSampler:
    """sclk and socket power while a loop runs: `rocm-smi --showpower --showclocks --json` polled from a thread (the only
    power/clock source that read plausibly on these boxes: profiles/r02_power_k1.json; the hwmon node reads a flat 248 W)."""

    def __init__(self):
        import threading

        self.exe = shutil.which("rocm-smi")
        self.rows, self._stop = [], threading.Event()
        self._th = threading.Thread(target=self._run, daemon=True) if self.exe else None

    def _run(self):
        while not self._stop.is_set():
            t = time.perf_counter()
            try:
                r = subprocess.run([self.exe, "--showpower", "--showclocks", "--json"], stdout=subprocess.PIPE,
                                   stderr=subprocess.DEVNULL, timeout=20)
                txt = r.stdout.decode(errors="replace")
                js = json.loads(txt[txt.index("{"):])
                card = js.get("card0") or next(iter(js.values()))
                sclk = w = None
                for k, v in card.items():
                    kl = k.lower()
                    if kl.startswith("sclk clock speed"):
                        sclk = float(str(v).strip("()").lower().replace("mhz", ""))
                    elif "power (w)" in kl and "cap" not in kl:
                        w = float(v)
                self.rows.append((t, sclk, w))
            except Exception:   # noqa: BLE001 (a probe: no sample)
                pass
            self._stop.wait(0.05)

    def start(self):
        if self._th:
            self._th.start()

    def stop(self, t_from, t_to):
        if not self._th:
            return {"sclk_mhz": None, "socket_w": None, "samples": 0, "source": "rocm-smi not found"}
        self._stop.set()
        self._th.join(timeout=30)
        rows = [r for r in self.rows if t_from <= r[0] <= t_to]
        ck = [r[1] for r in rows if r[1]]
        pw = [r[2] for r in rows if r[2]]
        return {"sclk_mhz": float(np.mean(ck)) if ck else None, "sclk_mhz_min": float(np.min(ck)) if ck else None,
                "socket_w": float(np.mean(pw)) if pw else None, "socket_w_max": float(np.max(pw)) if pw else None,
                "samples": len(rows), "source": "rocm-smi --showpower --showclocks --json, polled during the loop"}


def steady_state(host, step, seconds):
    """The same step back to back for >= `seconds` s AFTER the official timed window (which, at 0.25 s, measures whatever
    thermal / clock state the box happens to be in): ms per step over the window past its first half second."""
    smp = RocmSmiSampler()
    smp.start()
    host.sync()
    t0 = time.perf_counter()
    marks = []
    while True:
        for _ in range(10):
            step()
        host.sync()
        now = time.perf_counter()
        marks.append(now)
        if now - t0 >= seconds:
            break
    t1 = marks[-1]
    k0 = next((k for k, t in enumerate(marks) if t - t0 >= 0.5), 0)
    if k0 >= len(marks) - 1:
        k0 = 0
    ms = (marks[-1] - marks[k0]) / (10 * (len(marks) - 1 - k0)) * 1e3 if len(marks) - 1 > k0 else (t1 - t0) / (10 * len(marks)) * 1e3
    out = {"ms_per_step": ms, "steps": 10 * len(marks), "window_s": t1 - t0,
           "note": "back-to-back steps after the official window; ms_per_step excludes the first 0.5 s"}
    out.update(smp.stop(t0 + 0.5, t1))
    return out


class SingleHost:
    """One engine on one GPU."""
    kind = "single"

    def __init__(self, args, rx, st, device=0):
        self.rx, self.world, self.rank = rx, 1, 0
        e = rx.NBodyEngine(device=device, mode=args.mode)
        e.set_source_precision(args.source_bits)
        e.set_launch(jsplit=args.jsplit, bodies_per_thread=args.bpt, variant=args.variant)
        e.set_strict_kernel(args.strict_kernel)
        if args.bh_tree != "default":
            e.set_bh_tree(args.bh_tree)
        if args.bh_walk >= 0:
            e.set_option(rx.engine.NBX_OPT_BH_WALK, args.bh_walk)
        if args.shard_of > 1:
            e.set_shard(0, args.shard_of)
        e.set_particles(st["px"], st["py"], st["vx"], st["vy"], st["m"], st["pz"], st["vz"])
        self.eng = e
        self.engines = [e]
        self.local_only = args.shard_of > 1

    def prepare(self, theta=0.0):
        self.eng.forces(theta)   # uploads the state, sizes the work buffers (all-pairs or tree walk); no state change

    def step_brute(self):
        if self.local_only:
            self.eng.step_local(DT)
        else:
            self.eng.step_brute_force(DT)

    def step_bh(self, theta):
        self.eng.step_barnes_hut(theta, DT, 1)

    def get_state(self):
        return self.eng.get_particles()

    def sync(self):
        self.eng.synchronize()

    def barrier(self):
        pass

    def reduce_max(self, x):
        return x

    def close(self):
        self.eng.close()


class GroupHost:
    """ONE process, G GPUs: the library's own group (nbx_group_*). RCCL is issued by the library; no torch anywhere."""
    kind = "group"

    def __init__(self, args, rx, st):
        self.rx, self.world, self.rank = rx, args.gpus, 0
        have = rx.device_count()
        shared_ok = os.environ.get("NBX_GROUP_EXCHANGE") == "copy" or os.environ.get("NBX_GROUP_RCCL_FAIL") == "init"
        if have < args.gpus and not shared_ok:
            sys.exit(f"bench.py --gpus {args.gpus}: only {have} HIP device(s) visible (set NBX_GROUP_EXCHANGE=copy to let "
                     f"the engines of the group share devices: control-flow check only, not a measurement)")
        devices = [i % max(have, 1) for i in range(args.gpus)]
        g = rx.NBodyGroup(devices, mode=args.mode)
        g.set_source_precision(args.source_bits)
        from rust_exp_amd.engine import (NBX_OPT_BH_TREE, NBX_OPT_BODIES_PER_THREAD, NBX_OPT_JSPLIT, NBX_OPT_KERNEL_VARIANT,
                                         NBX_OPT_STRICT_KERNEL)

        g.set_option(NBX_OPT_JSPLIT, args.jsplit)
        g.set_option(NBX_OPT_BODIES_PER_THREAD, args.bpt)
        g.set_option(NBX_OPT_KERNEL_VARIANT, args.variant)
        g.set_option(NBX_OPT_STRICT_KERNEL, args.strict_kernel)
        if args.bh_tree != "default":
            g.set_option(NBX_OPT_BH_TREE, {"host": 0, "device": 1}[args.bh_tree])
        g.set_particles(st["px"], st["py"], st["vx"], st["vy"], st["m"], st["pz"], st["vz"])
        self.group = g
        self.engines = [g.engine(i) for i in range(g.size())]
        self.eng = self.engines[0]
        self.devices = devices

    def prepare(self, theta=0.0):
        for e in self.engines:
            e.forces(theta)
        # communicator set-up (ncclCommInitAll) outside the timed region: a zero-length step gathers the unchanged positions
        self.group.step_brute_force(0.0)
        self.group.synchronize()

    def step_brute(self):
        self.group.step_brute_force(DT)

    def step_bh(self, theta):
        self.group.step_barnes_hut(theta, DT, 1)

    def get_state(self):
        return self.group.get_particles()

    def sync(self):
        self.group.synchronize()

    def barrier(self):
        pass

    def reduce_max(self, x):
        return x

    def close(self):
        self.group.close()


class TorchHost:
    """One process per GPU (torch.distributed.run); torch.distributed's nccl backend (= RCCL) moves the slabs."""
    kind = "torch"

    def __init__(self, args, rx, st):
        import torch
        import torch.distributed as dist

        self.torch, self.dist, self.rx = torch, dist, rx
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # one rank per GPU. (NBX_DIST_BACKEND=gloo lets several ranks share one GPU: the builder's only way to
        # run this multi-rank path on a single-GPU box; the driver's launch uses the default, nccl = RCCL.)
        backend = os.environ.get("NBX_DIST_BACKEND", "nccl")
        self.local_rank = local_rank % max(1, torch.cuda.device_count())
        torch.cuda.set_device(self.local_rank)
        if self.world > 1:
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", self.local_rank))
            else:
                dist.init_process_group(backend)
        slab = rx.sharded.TorchSlabEngine(self.local_rank, mode=args.mode, source_half=args.source_bits == 16)
        slab.eng.set_launch(jsplit=args.jsplit, bodies_per_thread=args.bpt, variant=args.variant)
        slab.eng.set_strict_kernel(args.strict_kernel)
        if args.bh_tree != "default":
            slab.eng.set_bh_tree(args.bh_tree)
        self.sim = rx.ShardedNBody(slab)
        self.sim.set_particles(st)
        self.eng = slab.eng
        self.engines = [slab.eng]

    def prepare(self, theta=0.0):
        self.eng.forces(theta)
        if self.world > 1:
            self.sim._exchange()   # communicator set-up outside the timed region (re-gathers the initial positions: a no-op on the data)

    def step_brute(self):
        self.sim.step_brute_force(DT)

    def step_bh(self, theta):
        self.sim.step_barnes_hut(theta, DT, 1)

    def get_state(self):
        return self.sim.gather_state()   # collective: every rank calls it

    def sync(self):
        self.torch.cuda.synchronize()

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()

    def reduce_max(self, x):
        if self.world == 1:
            return x
        t = self.torch.tensor([x], dtype=self.torch.float64, device="cuda")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def gather_floats(self, xs):
        """list of floats of every rank, on every rank"""
        if self.world == 1:
            return [xs]
        t = self.torch.tensor(xs, dtype=self.torch.float64, device="cuda")
        out = [self.torch.zeros_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t)
        return [o.tolist() for o in out]

    def close(self):
        if self.world > 1:
            self.dist.destroy_process_group()


# ---- who ran the CPU leg, and what each side computes (bench.py's cpu_baseline block) -----------------------------------------
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


def cpu_model():
    """The host CPU's model string (/proc/cpuinfo "model name"; BASELINE.md section 3: "state T and CPU model")."""
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.lower().startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform

    return platform.processor() or platform.machine() or "unknown"


def cpu_facts(cores):
    """Who ran the CPU leg: model, and how many of the host's logical CPUs this process may use (cgroup quota / affinity)."""
    logical = os.cpu_count() or cores
    return {"model": cpu_model(), "logical_cpus": logical,
            "quota": f"{cores} of {logical} logical CPUs (cgroup CPU quota / scheduler affinity)"}


# what the CPU leg computes, beside what the GPU sweep computes: NOT the same work per interaction
CPU_LAW = ("2-D reference law as written (nbody.rs:174-183 + :141-142): 12 flops per interaction, one IEEE divide, f32, "
           "ascending-j sequential sum, self skipped by index")
GPU_LAW_3D = "3-D float4 sweep: 17 algorithmic flops per interaction (z terms added), v_rcp_f32 for the divide, packed FMA, tiled sums"
GPU_LAW_2D = "2-D sweep: 12 algorithmic flops per interaction, v_rcp_f32 for the divide, packed FMA, tiled sums"
