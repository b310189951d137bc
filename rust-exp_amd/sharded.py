"""Multi-GPU host logic: bodies shard across ranks as contiguous slabs of TARGETS, the reference's
own static split (nbody.rs:426-428: range = N / T, the last worker takes the remainder).

One process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm, xGMI inside a node).
Every rank keeps the full (x, y, z, m) source array and the velocities of its own slab. A step is

    local force + kick-drift on the slab            (HIP kernels, nbx_step_local)
    ONE all-gather of the updated (x, y, z, m) slabs (in place when slabs are equal)

which is the only exchange the path has: the force on body i needs every position but only i's
own velocity (nbody.rs:132-160).  The collective runs on the same torch stream the kernels were
enqueued on, so no host synchronisation happens inside a step.

The class is backend-agnostic: `local_engine` is anything with set_shard / set_particles /
step_local / slab / positions_array. The product engine is `TorchSlabEngine` (HIP library on
torch-owned device memory); CPU tests inject a test double (tests/test_sharded_gloo.py).
"""
import numpy as np


def reference_slab(n, rank, world):
    """[lo, hi) of `rank` under the reference split (nbody.rs:426-428)."""
    rng = n // world
    lo = rng * rank
    hi = n if rank == world - 1 else rng * (rank + 1)
    return lo, hi


class TorchSlabEngine:
    """The HIP engine bound to a torch-owned positions buffer on this rank's GPU."""

    def __init__(self, device_index, mode="fast", source_half=False):
        import sys

        from . import engine as _engine

        if _engine._lib is not None and "torch" not in sys.modules:
            # The PyTorch-ROCm wheel bundles its own libamdhip64/libhsa-runtime64. Loaded AFTER the
            # system ROCm runtime that libnbody_mi355x.so pulled in, the process ends up with two HSA
            # runtimes and torch sees "No HIP GPUs". Loaded FIRST, both share torch's runtime.
            raise RuntimeError("import torch before the first rust_exp_amd.lib() call in this process "
                               "(two HIP runtimes would be loaded otherwise)")
        import torch

        from .engine import NBodyEngine

        self.torch = torch
        self.device = torch.device("cuda", device_index)
        torch.cuda.set_device(self.device)
        self.eng = NBodyEngine(device=device_index, mode=mode)
        self.source_half = bool(source_half)
        if self.source_half:
            self.eng.set_source_precision(16)
        self.posm = None
        self.posh = None

    def set_shard(self, rank, world):
        self.eng.set_shard(rank, world)

    def set_particles(self, st):
        torch = self.torch
        self.eng.set_particles(st["px"], st["py"], st["vx"], st["vy"], st["m"], st.get("pz"), st.get("vz"))
        nbytes = self.eng.positions_bytes()
        self.posm = torch.zeros(nbytes // 16, 4, dtype=torch.float32, device=self.device)
        # kernels run on torch's current stream so that collectives issued by torch order after them
        self.eng.set_stream(torch.cuda.current_stream(self.device).cuda_stream)
        self.eng.bind_positions(self.posm.data_ptr(), nbytes)
        if self.source_half:
            # the exchanged array is the half4 source copy: 8 B per body on the wire instead of 16
            self.posh = torch.zeros(nbytes // 16, 4, dtype=torch.float16, device=self.device)
            self.eng.bind_half_sources(self.posh.data_ptr(), nbytes // 2)
        torch.cuda.synchronize(self.device)

    def slab(self):
        return self.eng.slab()

    def step_local(self, dt):
        self.eng.step_local(dt)

    def step_local_barnes_hut(self, theta, dt):
        # every rank rebuilds the (identical) quadtree on its host from the gathered positions and evaluates
        # only its slab of targets on its GPU (tree replicas + slab of targets per GPU)
        self.eng.step_barnes_hut(theta, dt, 1)

    def positions_array(self):
        """The array the per-step all-gather moves: [n_pad, 4] (x, y, z, m), float32 or float16."""
        return self.posh if self.source_half else self.posm

    @property
    def positions_replicated(self):
        return not self.source_half   # fp16 mode: other ranks' fp32 positions are not kept current

    def get_particles(self):
        return self.eng.get_particles()


class ShardedNBody:
    def __init__(self, local_engine, group=None, always_exchange=False):
        import torch.distributed as dist

        self.dist = dist
        self.group = group
        self.always_exchange = always_exchange   # run the collective even at world size 1 (tests)
        self._staged = False
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.local = local_engine
        self.local.set_shard(self.rank, self.world)
        self.n = 0

    def set_particles(self, st):
        self.n = len(st["px"])
        self.local.set_particles(st)
        self.lo, self.hi = self.local.slab()
        assert (self.lo, self.hi) == reference_slab(self.n, self.rank, self.world)

    def _exchange(self):
        """One all-gather of the slabs of the (x,y,z,m) array."""
        if self.world == 1 and not (self.always_exchange and self.dist.is_initialized()):
            return
        dist = self.dist
        pos = self.local.positions_array()
        n, w = self.n, self.world
        if n % w == 0:
            body = pos[:n]
            if not self._staged:
                try:
                    # in place: the send slab IS the rank's slot of the receive buffer (ncclAllGather
                    # in-place form: sendbuff == recvbuff + rank * count)
                    dist.all_gather_into_tensor(body, body[self.lo:self.hi], group=self.group)
                    return
                except (RuntimeError, ValueError):
                    self._staged = True   # backend refuses aliasing: stage the slab once per step
            dist.all_gather_into_tensor(body, body[self.lo:self.hi].clone(), group=self.group)
        else:
            # ragged last slab (reference split): one broadcast per owner
            for r in range(w):
                lo, hi = reference_slab(n, r, w)
                if hi > lo:
                    src = dist.get_global_rank(self.group, r) if self.group is not None else r
                    dist.broadcast(pos[lo:hi], src=src, group=self.group)

    def step_brute_force(self, dt):
        """nb_step_brute_force (nbody.rs:106-162) across all ranks."""
        self.local.step_local(dt)
        self._exchange()

    def step_barnes_hut(self, theta, dt, nthreads=1):
        """nb_step_barnes_hut (nbody.rs:186-480) across all ranks: theta == 0 delegates to brute force
        (nbody.rs:197-200); otherwise tree replicas + slab of targets per rank, same single exchange."""
        if theta == 0.0:
            return self.step_brute_force(dt)
        if not getattr(self.local, "positions_replicated", True):
            raise RuntimeError("Barnes-Hut needs replicated fp32 positions (not the fp16 source copy)")
        self.local.step_local_barnes_hut(theta, dt)
        self._exchange()

    def gather_state(self):
        """Full fp32 state on every rank: velocities always live only on their owner; positions too when the
        exchanged array is the fp16 copy."""
        import torch

        st = self.local.get_particles()
        if self.world == 1:
            return st
        out = dict(st)
        keys = ["vx", "vy", "vz"]
        if not getattr(self.local, "positions_replicated", True):
            keys += ["px", "py", "pz"]
        nccl = self.dist.get_backend(self.group) == "nccl"
        for k in keys:
            full = torch.from_numpy(np.array(st[k], dtype=np.float32))
            for r in range(self.world):
                lo, hi = reference_slab(self.n, r, self.world)
                if hi > lo:
                    src = self.dist.get_global_rank(self.group, r) if self.group is not None else r
                    part = full[lo:hi].clone()
                    if nccl:
                        part = part.cuda()
                        self.dist.broadcast(part, src=src, group=self.group)
                        full[lo:hi] = part.cpu()
                    else:
                        self.dist.broadcast(part, src=src, group=self.group)
                        full[lo:hi] = part
            out[k] = full.numpy()
        return out
