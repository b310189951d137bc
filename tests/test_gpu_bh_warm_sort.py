"""GPU: the device tree build's sort that starts from last step's order (bh_sort.hip, round 5: k_sample_rank / k_keys_scatter /
k_bucket_sort; replaces the serial insert loop nbody.rs:410-415 together with the rest of the build).

(key, index) pairs are distinct, so the sorted order is unique and the tree must not depend on how it was reached: after EVERY
step of a run the device tree -- built warm, from the previous build's order -- equals the host tree (= the oracle's,
tests/test_gpu_bh_device_tree.py) bit for bit with the reference fold, and has its structure with exact sums; a build whose
buckets overflow is refused and the step redone on the host tree, never wrong."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _bit_equal_trees(host, dev):
    assert len(host) == len(dev), (len(host), len(dev))
    for k in ("skip", "interior"):
        assert np.array_equal(host[k], dev[k]), k
    for k in ("px", "py", "m", "s", "q"):
        bad = np.flatnonzero(host[k].view(np.uint32) != dev[k].view(np.uint32))
        assert bad.size == 0, (k, bad.size, bad[:5])


def _state(rx, ob, make, n):
    if make == "disk":
        return ob.random_disk(n, 41)                      # velocities U[-3.5, 3.5): bodies cross cells every step
    if make == "orbits":
        return ob.stable_orbits(n, 0.5, 30.0, 42)         # v = sqrt(1000): 0.3 length units per step, the hot case
    st = rx.plummer_sphere(n, dim=2)
    return ob.particles(st["px"], st["py"], st["vx"], st["vy"], st["m"])


@pytest.mark.parametrize("make,n,steps", [("orbits", 20000, 6), ("disk", 50000, 6), ("plummer", 65536, 4), ("orbits", 131072, 4),
                                          ("disk", 300000, 3)])
def test_warm_sort_builds_the_host_tree_bit_for_bit_after_every_step(rx, ob, make, n, steps):
    from rust_exp_amd.engine import NBX_STAT_BH_FALLBACKS, NBX_STAT_BH_LAST_TREE

    p = _state(rx, ob, make, n)
    e = rx.NBodyEngine()
    e.set_bh_fold("reference")                            # the class that promises the host tree node for node, at any size
    e.set_bh_tree("device")
    e.set_particles(p["px"], p["py"], p["vx"], p["vy"], p["m"])
    fallbacks0 = e.get_stat(NBX_STAT_BH_FALLBACKS)
    for k in range(steps):
        e.step_barnes_hut(0.5, 0.01, 1)                   # build k is cold for k = 0, warm from then on
        e.synchronize()
        _bit_equal_trees(e.bh_flat_dump(False), e.bh_flat_dump("device"))   # (the dump builds once more: warm, same state)
    # the steps ran on the device tree, warm builds included (a refused build would have counted as a fallback)
    assert e.get_stat(NBX_STAT_BH_LAST_TREE) == 1
    assert e.get_stat(NBX_STAT_BH_FALLBACKS) == fallbacks0


def test_warm_and_cold_sorts_step_to_the_same_bits(rx, ob):
    """The same 8 steps with the warm sort and with the library sort every step (NBX_INC_SORT=0 in a child process): positions and
    velocities bit-identical (the exact-sum class above 65 536 bodies: the tree does not depend on the sort)."""
    import subprocess
    import sys

    code = ("import sys, numpy as np; sys.path.insert(0, %r); import rust_exp_amd as rx\n"
            "st = rx.plummer_sphere(200000, dim=2)\n"
            "rng = np.random.default_rng(5); vx = rng.normal(0, 8, 200000).astype(np.float32); vy = rng.normal(0, 8, 200000).astype(np.float32)\n"
            "e = rx.NBodyEngine(); e.set_bh_tree('device'); e.set_particles(st['px'], st['py'], vx, vy, st['m'])\n"
            "for _ in range(8): e.step_barnes_hut(0.5, 0.01, 1)\n"
            "q = e.get_particles(); print(e.get_stat(rx.engine.NBX_STAT_BH_FALLBACKS)); np.save(sys.argv[1], np.stack([q['px'], q['py'], q['vx'], q['vy']]))\n"
            % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    out = []
    for flag in ("1", "0"):
        path = "/tmp/nbx_warm_%s.npy" % flag
        r = subprocess.run([sys.executable, "-c", code, path], env=dict(os.environ, NBX_INC_SORT=flag), stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-500:]
        assert r.stdout.strip().splitlines()[-1] == "0"          # no fallback either way
        out.append(np.load(path))
    assert np.array_equal(out[0].view(np.uint32), out[1].view(np.uint32))


def test_a_reshuffled_system_is_refused_not_wrong(rx, ob):
    """Last step's order says nothing about bodies that were all moved by hand: through nbx_set_particles the engine forgets the
    order (cold sort). If the order is stale anyway -- here: the SAME engine state, bodies teleported by one huge step -- the warm
    sort's buckets may overflow: that build is refused and redone from a cold sort (round 6: on the device; round 5: on the host
    tree); the result is the step of an engine whose order never comes from the warm sort (host tree: its Morton order is the
    library sort's, always -- ADVICE r05)."""
    from rust_exp_amd.engine import NBX_STAT_BH_FALLBACKS

    n = 120000
    p = ob.random_disk(n, 7)
    a = rx.NBodyEngine(); a.set_bh_tree("device")
    b = rx.NBodyEngine(); b.set_bh_tree("host")
    for e in (a, b):
        e.set_particles(p["px"], p["py"], p["vx"], p["vy"], p["m"])
        e.step_barnes_hut(0.5, 0.01, 1)
        e.step_barnes_hut(0.5, 3.0, 1)        # every body moves ~10 length units: the order is scrambled
        e.step_barnes_hut(0.5, 0.01, 1)       # warm build on a scrambled order: sorted all the same, or refused
        e.step_barnes_hut(0.5, 0.01, 1)
    qa, qb = a.get_particles(), b.get_particles()
    assert a.get_stat(NBX_STAT_BH_FALLBACKS) == 0          # whatever the warm sort made of it, the host tree was not needed
    for k in ("px", "py", "vx", "vy"):
        assert np.isfinite(qa[k]).all()
        # exact-sum device tree vs host tree: the fast mode's tolerance class, not bits; a wrong sort would be off by O(1)
        assert np.abs(qa[k] - qb[k]).max() <= 2e-2 * max(1.0, np.abs(qb[k]).max()), k


def _disk_with_coincident_bodies(ob, n, at_one_point, seed):
    """More bodies at ONE position than a bucket of the warm sort has slots (kBucketCap = 4 096): one 62-bit key -> one bucket ->
    overflow, deterministically, at every warm sort.  (The reference merges them into one leaf, nbody.rs:249-260.)"""
    p = ob.random_disk(n, seed)
    idx = np.random.default_rng(seed).choice(n, at_one_point, replace=False)
    # at the body nearest the origin: the reference's running f32 fold of k identical positions (nbody.rs:315-317) drifts by ~sqrt(k/3)
    # ulps -- 40 ulps for 5 000 bodies, 8e-5 at |x| = 20: nearly EPS, where the reference would SPLIT the blob again and the device
    # build refuse for that reason (seen in round 6); at |x| < 1 it is 2e-6
    c = int(np.argmin(np.abs(p["px"]) + np.abs(p["py"])))
    p["px"][idx] = p["px"][c]; p["py"][idx] = p["py"][c]
    p["vx"][idx] = 0.0; p["vy"][idx] = 0.0      # (they stay together: every later warm sort meets them again)
    p["m"][idx] = 1e-3    # light: a blob of thousands of unit masses at one point flings its neighbours by 20 length units a step -- chaos, not a test
    return p


def test_host_tree_steps_never_take_an_overflowed_order(rx, ob):
    """the Morton order a HOST-tree step walks its bodies in (n >= 65 536) came from the warm sample sort in
    round 5, and nothing on that path read the sort's overflow verdict: with more than 4 096 coincident bodies the order lost
    bodies and named body 0 several times -- the lost ones were never integrated again.  Round 6: that order always comes from the
    library sort.  Bit-exact mode, host tree, three steps: the oracle's state bit for bit (nbody.rs:186-480); fast mode: the same
    system within the fast tolerance, and every body moved by its own velocity."""
    n, dt = 70000, 0.001      # (56 000 units of mass within 23 length units: a ~ 5e3; at the reference's dt = 0.01 seven steps fling bodies
    p = _disk_with_coincident_bodies(ob, n, 5000, 3)   # thousands of units out, and a root box that wide makes the reference panic on its depth counter)
    s = rx.NBodyEngine(mode="strict"); s.set_bh_tree("host")
    f = rx.NBodyEngine(mode="fast"); f.set_bh_tree("host")
    for e in (s, f):
        e.set_particles(p["px"], p["py"], p["vx"], p["vy"], p["m"])
        for _ in range(3):
            e.step_barnes_hut(0.6, dt, 1)
    q = p.copy()
    for _ in range(3):
        assert ob.step_barnes_hut(q, 0.6, dt, 16) == 0
    a, b = s.get_particles(), f.get_particles()
    for k in ("px", "py", "vx", "vy"):
        assert np.array_equal(a[k].view(np.uint32), q[k].view(np.uint32)), k
    amax = float(np.abs(np.stack([q["vx"] - p["vx"], q["vy"] - p["vy"]])).max()) / (3 * dt)      # ~ max |a| over the three steps
    for k in ("px", "py"):
        assert np.abs(b[k] - q[k]).max() <= 1e-5 + 1e-4 * amax * (3 * dt) ** 2, k
    for k in ("vx", "vy"):
        assert np.abs(b[k] - q[k]).max() <= 1e-4 * amax * 3 * dt + 1e-5, k
    # nobody was left out of the integration: every body moved by about its own velocity (the kicks are ~ amax dt = a few units of speed)
    moved = np.hypot(b["px"] - p["px"], b["py"] - p["py"])
    speed = np.hypot(p["vx"], p["vy"])
    assert (np.abs(moved - 3 * dt * speed) <= 3 * dt * (2 * 3 * dt * amax) + 1e-6).all()
    assert (moved[speed > 1.0] > 0).all()


@pytest.mark.parametrize("async_", [1, 0])
def test_an_overflowed_warm_sort_is_redone_cold_on_the_device(rx, ob, async_):
    """More than 4 096 coincident bodies overflow every WARM sort of the device build (NBX_STAT_BH_REFUSAL 0x100000).  Round 6: the
    build is redone at once from a cold sort on the device (NBX_STAT_BH_COLD_RESORTS; the host tree is not asked), the next builds
    sort cold as well (hold-off 2, 4 .. 32), and -- ADVICE r05 (medium) -- the step that was already enqueued BEHIND the refused
    one (two steps in flight; it starts from an order that is no permutation) does nothing, is enqueued again and changes no
    bit: pipelined and waiting forms step to the same state, which is the host-tree engine's within the class tolerance."""
    from rust_exp_amd.engine import (NBX_OPT_BH_ASYNC, NBX_STAT_BH_COLD_RESORTS, NBX_STAT_BH_FALLBACKS, NBX_STAT_BH_LAST_TREE,
                                     NBX_STAT_BH_REFUSAL)

    n = 70000
    p = _disk_with_coincident_bodies(ob, n, 5000, 4)
    e = rx.NBodyEngine(); e.set_bh_tree("device"); e.set_option(NBX_OPT_BH_ASYNC, async_)
    w = rx.NBodyEngine(); w.set_bh_tree("device"); w.set_option(NBX_OPT_BH_ASYNC, 0)
    h = rx.NBodyEngine(); h.set_bh_tree("host")
    for g in (e, w, h):
        g.set_particles(p["px"], p["py"], p["vx"], p["vy"], p["m"])
        for _ in range(7):                                  # cold, warm (overflow), cold, cold, warm (overflow), cold x 4 ...
            g.step_barnes_hut(0.5, 0.001, 1)                # (dt: see test_host_tree_steps_never_take_an_overflowed_order)
    a, b, c = e.get_particles(), w.get_particles(), h.get_particles()
    assert e.get_stat(NBX_STAT_BH_REFUSAL) & 0x100000 and e.get_stat(NBX_STAT_BH_COLD_RESORTS) == 2
    assert e.get_stat(NBX_STAT_BH_FALLBACKS) == 0 and e.get_stat(NBX_STAT_BH_LAST_TREE) == 1
    for k in ("px", "py", "vx", "vy"):
        assert np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32)), k
        assert np.abs(a[k] - c[k]).max() <= 2e-2 * max(1.0, np.abs(c[k]).max()), k


@pytest.mark.parametrize("n", [16385, 16640, 33000])
def test_warm_sort_just_above_the_small_front(rx, ob, n):
    """The first sizes that take the warm sort (26-52 buckets, fewer workgroups than CUs): the device tree equals the host tree bit
    for bit after every step (reference fold)."""
    from rust_exp_amd.engine import NBX_STAT_BH_FALLBACKS

    from rust_exp_amd.engine import NBX_STAT_BH_CLASS_SWITCHES

    p = ob.random_disk(n, 5)
    e = rx.NBodyEngine()
    e.set_bh_tree("device")
    e.set_bh_fold("reference")
    e.set_particles(p["px"], p["py"], p["vx"], p["vy"], p["m"])
    f0 = e.get_stat(NBX_STAT_BH_FALLBACKS) + e.get_stat(NBX_STAT_BH_CLASS_SWITCHES)
    kept = 0
    for _ in range(4):
        e.step_barnes_hut(0.85, 0.01, 1)
        e.synchronize()
        if e.get_stat(NBX_STAT_BH_FALLBACKS) + e.get_stat(NBX_STAT_BH_CLASS_SWITCHES) == f0:   # (EPS clusters may hand a step of this class on: its own test)
            try:
                dev = e.bh_flat_dump("device")                # (the dump builds once more, the class asked for or nothing)
            except rx.NBodyError:
                dev = None
            if dev is not None:
                _bit_equal_trees(e.bh_flat_dump(False), dev)
                kept += 1
        f0 = e.get_stat(NBX_STAT_BH_FALLBACKS) + e.get_stat(NBX_STAT_BH_CLASS_SWITCHES)
    assert kept >= 1


def test_above_the_warm_sorts_range_the_library_sort_serves_every_step(rx, ob):
    """More than 4 096 x 640 bodies: no warm sort (its splitter table ends there); the steps run on the device tree all the same
    and leave a finite state."""
    from rust_exp_amd.engine import NBX_STAT_BH_FALLBACKS, NBX_STAT_BH_LAST_TREE

    n = 4096 * 640 + 4096
    st = rx.plummer_sphere(n, dim=2)
    e = rx.NBodyEngine()
    e.set_particles(st["px"], st["py"], st["vx"], st["vy"], st["m"])
    for _ in range(3):
        e.step_barnes_hut(0.5, 0.01, 1)
    q = e.get_particles()
    assert np.isfinite(q["px"]).all() and np.isfinite(q["vx"]).all()
    assert e.get_stat(NBX_STAT_BH_LAST_TREE) == 1 and e.get_stat(NBX_STAT_BH_FALLBACKS) == 0


def test_sharded_engines_take_the_warm_order_too(rx, ob, monkeypatch):
    """world > 1 on one GPU (control flow): every engine sorts ALL bodies (tree replicas) and walks its slab in the Morton order
    restricted to it; warm builds from the second step on.  Positions after four steps equal one plain engine's bit for bit
    (exact-sum device tree: the same tree whichever sort made the order)."""
    monkeypatch.setenv("NBX_GROUP_EXCHANGE", "copy")      # engines of one group may share the test GPU
    n = 100000
    st = rx.plummer_sphere(n, dim=2)
    rng = np.random.default_rng(3)
    vx = rng.normal(0, 5, n).astype(np.float32); vy = rng.normal(0, 5, n).astype(np.float32)
    plain = rx.NBodyEngine()
    plain.set_bh_tree("device")
    plain.set_particles(st["px"], st["py"], vx, vy, st["m"])
    g = rx.NBodyGroup([0, 0, 0])
    from rust_exp_amd.engine import NBX_OPT_BH_TREE
    g.set_option(NBX_OPT_BH_TREE, 1)
    g.set_particles(st["px"], st["py"], vx, vy, st["m"])
    for _ in range(4):
        plain.step_barnes_hut(0.5, 0.01, 1)
        g.step_barnes_hut(0.5, 0.01, 1)
    a, b = plain.get_particles(), g.get_particles()
    for k in ("px", "py", "vx", "vy"):
        assert np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32)), k
    g.close()


def test_mixed_sequences_keep_the_order_sorted_copy_of_the_positions_honest(rx, ob):
    """Round 5: the fused walk + kick-drift leaves the new positions once more in the walks' order, and the next build's sort reads
    them from there instead of gathering.  Everything else that moves bodies or replaces the order must take that copy out of use:
    all-pairs steps, force-only evaluations and tree dumps (a build without a kick-drift), new particles, bit-exact steps.
    A sequence that interleaves them all equals the same sequence with the library sort every step (NBX_INC_SORT=0), bit for bit."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, numpy as np; sys.path.insert(0, %r); import rust_exp_amd as rx\n"
            "n = 150000\n"
            "st = rx.plummer_sphere(n, dim=2)\n"
            "rng = np.random.default_rng(9); vx = rng.normal(0, 6, n).astype(np.float32); vy = rng.normal(0, 6, n).astype(np.float32)\n"
            "e = rx.NBodyEngine(); e.set_bh_tree('device'); e.set_particles(st['px'], st['py'], vx, vy, st['m'])\n"
            "out = []\n"
            "def snap():\n"
            "    q = e.get_particles(); out.append(np.stack([q['px'], q['py'], q['vx'], q['vy']]))\n"
            "for _ in range(3): e.step_barnes_hut(0.5, 0.01, 1)\n"
            "fx, fy, _ = e.forces(0.5); out.append(np.stack([fx, fy, fx, fy]))\n"       # a build without a kick-drift
            "for _ in range(2): e.step_barnes_hut(0.5, 0.01, 1)\n"
            "e.step_brute_force(0.01)\n"                                               # bodies move, the order does not
            "for _ in range(2): e.step_barnes_hut(0.5, 0.01, 1)\n"
            "snap()\n"
            "d = e.bh_flat_dump('device'); out.append(np.stack([d['px'], d['py'], d['m'], d['s']])[:, :n])\n"
            "e.step_barnes_hut(0.85, 0.01, 1); e.synchronize(); e.step_barnes_hut(0.5, 0.01, 1)\n"
            "q = e.get_particles(); e.set_particles(q['py'], q['px'], q['vy'], q['vx'], q['m'])\n"   # new particles (mirrored)
            "for _ in range(3): e.step_barnes_hut(0.5, 0.01, 1)\n"
            "snap()\n"
            "np.save(sys.argv[1], np.concatenate(out, axis=1))\n" % root)
    res = []
    for flag in ("1", "0"):
        path = "/tmp/nbx_mixed_%s.npy" % flag
        r = subprocess.run([sys.executable, "-c", code, path], env=dict(os.environ, NBX_INC_SORT=flag), stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-800:]
        res.append(np.load(path))
    assert res[0].shape == res[1].shape and np.isfinite(res[0]).all()
    assert np.array_equal(res[0].view(np.uint32), res[1].view(np.uint32))


def test_clumped_keys_take_the_network_and_ties_go_by_index(rx, ob):
    """Buckets whose keys clump -- EPS-scale clumps, six hundred bodies at ONE point (identical 62-bit keys: only the index orders
    them; 600^2 > 48 x the bucket's pairs; the second clump of 3 000 fills a bucket beyond the LDS staging area: the network's
    two-halves exchange) -- leave the counting sort for the bitonic network over (key, index).  Warm and cold (library sort) runs must step to the
    same bits, and the exact-sum device tree must stay within the fast mode's tolerance of the host tree's forces."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, numpy as np; sys.path.insert(0, %r); import rust_exp_amd as rx\n"
            "rng = np.random.default_rng(21); n = 60000\n"
            "x = rng.normal(0, 8, n).astype(np.float32); y = rng.normal(0, 8, n).astype(np.float32)\n"
            "k = n // 4\n"                                                        # a quarter of the bodies in clumps ~3e-5 wide around others
            "x[:k] = x[k:2 * k] + rng.normal(0, 3e-5, k).astype(np.float32); y[:k] = y[k:2 * k] + rng.normal(0, 3e-5, k).astype(np.float32)\n"
            "x[-600:] = x[-601]; y[-600:] = y[-601]\n"                           # six hundred bodies at one point: one sub-bucket of 600
            "x[100:3100] = x[99]; y[100:3100] = y[99]\n"                         # three thousand at another: a bucket beyond the staging area
            "m = rng.uniform(0.5, 1.5, n).astype(np.float32); v = np.zeros(n, np.float32)\n"
            "e = rx.NBodyEngine(); e.set_bh_tree('device'); e.set_bh_fold('exact'); e.set_particles(x, y, v, v, m)\n"
            "for _ in range(5): e.step_barnes_hut(0.5, 0.001, 1)\n"
            "q = e.get_particles(); fx, fy, _ = e.forces(0.5)\n"
            "h = rx.NBodyEngine(); h.set_bh_tree('host'); h.set_particles(q['px'], q['py'], q['vx'], q['vy'], q['m']); gx, gy, _ = h.forces(0.5)\n"
            "err = max(np.abs(fx - gx).max(), np.abs(fy - gy).max()) / max(np.abs(gx).max(), np.abs(gy).max())\n"
            "print(e.get_stat(rx.engine.NBX_STAT_BH_FALLBACKS), err)\n"
            "np.save(sys.argv[1], np.stack([q['px'], q['py'], q['vx'], q['vy']]))\n" % root)
    res = []
    for flag in ("1", "0"):
        path = "/tmp/nbx_clump_%s.npy" % flag
        r = subprocess.run([sys.executable, "-c", code, path], env=dict(os.environ, NBX_INC_SORT=flag), stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-800:]
        fallbacks, err = r.stdout.strip().splitlines()[-1].split()
        res.append((np.load(path), int(fallbacks), float(err)))
    assert np.array_equal(res[0][0].view(np.uint32), res[1][0].view(np.uint32))
    assert res[0][1] == res[1][1] == 0                  # every step ran on the device tree, whichever sort made the order
    assert res[0][2] <= 1e-2 and res[1][2] <= 1e-2, (res[0][2], res[1][2])
