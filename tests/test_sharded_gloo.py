"""CPU, world_size 2 and 3 over gloo: the multi-GPU host logic (rust-exp_amd/sharded.py) -- slab
partition per nbody.rs:426-428, local step, ONE exchange of positions per step -- with the HIP
engine replaced by a test double built on the oracle (tests may use the oracle; the product path
never does).  The sharded result must equal the single-process oracle step bit for bit, for even
and ragged splits."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleSlabEngine:
    """Test double with the interface ShardedNBody expects from a local engine."""

    def __init__(self):
        import torch

        from oracle import binding as ob

        self.torch, self.ob = torch, ob
        self.rank, self.world = 0, 1

    def set_shard(self, rank, world):
        self.rank, self.world = rank, world

    def set_particles(self, st):
        from rust_exp_amd import reference_slab

        n = len(st["px"])
        self.n = n
        self.lo, self.hi = reference_slab(n, self.rank, self.world)
        self.m = np.array(st["m"], np.float32)
        self.vx = np.array(st["vx"], np.float32)
        self.vy = np.array(st["vy"], np.float32)
        pos = np.zeros((n, 4), np.float32)
        pos[:, 0], pos[:, 1], pos[:, 3] = st["px"], st["py"], st["m"]
        self.pos = self.torch.from_numpy(pos)

    def slab(self):
        return self.lo, self.hi

    def step_local(self, dt):
        pos = self.pos.numpy()
        p = self.ob.particles(pos[:, 0], pos[:, 1], self.vx, self.vy, self.m)
        fx, fy = self.ob.brute_forces(p, self.lo, self.hi)
        dt = np.float32(dt)
        sl = slice(self.lo, self.hi)
        self.vx[sl] = self.vx[sl] + (dt * fx) / self.m[sl]      # nbody.rs:155
        self.vy[sl] = self.vy[sl] + (dt * fy) / self.m[sl]
        pos[sl, 0] = pos[sl, 0] + dt * self.vx[sl]              # nbody.rs:158
        pos[sl, 1] = pos[sl, 1] + dt * self.vy[sl]

    def step_local_barnes_hut(self, theta, dt):
        pos = self.pos.numpy()
        p = self.ob.particles(pos[:, 0], pos[:, 1], self.vx, self.vy, self.m)
        rc, fx, fy = self.ob.bh_forces(p, theta)
        assert rc == 0
        dt = np.float32(dt)
        sl = slice(self.lo, self.hi)
        self.vx[sl] = self.vx[sl] + (dt * fx[sl]) / self.m[sl]     # nbody.rs:453
        self.vy[sl] = self.vy[sl] + (dt * fy[sl]) / self.m[sl]
        pos[sl, 0] = pos[sl, 0] + dt * self.vx[sl]                 # nbody.rs:457
        pos[sl, 1] = pos[sl, 1] + dt * self.vy[sl]
        kill = (np.abs(np.float32(0.0) - pos[sl, 0]) > np.float32(100.0) * np.float32(0.55)) | \
               (np.abs(np.float32(0.0) - pos[sl, 1]) > np.float32(100.0) * np.float32(0.55))   # nbody.rs:466-471
        self.vx[sl][kill] = 0.0
        self.vy[sl][kill] = 0.0

    def positions_array(self):
        return self.pos

    def get_particles(self):
        pos = self.pos.numpy()
        z = np.zeros(self.n, np.float32)
        return {"px": pos[:, 0].copy(), "py": pos[:, 1].copy(), "pz": z, "vx": self.vx.copy(), "vy": self.vy.copy(),
                "vz": z.copy(), "m": self.m.copy()}


def _worker(rank, world, port, n, steps, q, theta=0.0):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import rust_exp_amd as rx
        from oracle import binding as ob

        p = ob.random_disk(n, 77)
        st = {k: np.array(p[k]) for k in ("px", "py", "vx", "vy", "m")}
        sim = rx.ShardedNBody(OracleSlabEngine())
        sim.set_particles(st)
        for _ in range(steps):
            sim.step_barnes_hut(theta, 0.01, 1)      # theta == 0 -> brute force (nbody.rs:197-200)
        full = sim.gather_state()
        q.put((rank, sim.lo, sim.hi, {k: np.array(full[k]) for k in ("px", "py", "vx", "vy")}))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("world,n,theta", [(2, 512, 0.0), (2, 301, 0.0), (3, 100, 0.0), (2, 400, 0.85), (3, 333, 0.5)])
def test_sharded_step_equals_single_process_oracle(world, n, theta):
    import torch.multiprocessing as mp

    from oracle import binding as ob

    steps = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, steps, q, theta)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = ob.random_disk(n, 77)
    for _ in range(steps):
        assert ob.step_barnes_hut(ref, theta, 0.01, 1) == 0
    slabs = sorted((lo, hi) for _, lo, hi, _ in results)
    assert slabs[0][0] == 0 and slabs[-1][1] == n and all(a[1] == b[0] for a, b in zip(slabs, slabs[1:]))
    for rank, lo, hi, stt in results:
        for k in ("px", "py", "vx", "vy"):
            assert np.array_equal(stt[k].view(np.uint32), np.array(ref[k]).view(np.uint32)), (rank, k)


def test_sharded_world1_needs_no_process_group():
    import rust_exp_amd as rx
    from oracle import binding as ob

    p = ob.random_disk(64, 5)
    st = {k: np.array(p[k]) for k in ("px", "py", "vx", "vy", "m")}
    sim = rx.ShardedNBody(OracleSlabEngine())
    sim.set_particles(st)
    sim.step_brute_force(0.01)
    ob.step_brute_force(p, 0.01)
    got = sim.gather_state()
    assert np.array_equal(got["px"], p["px"]) and np.array_equal(got["vx"], p["vx"])
