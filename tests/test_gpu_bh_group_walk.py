"""GPU: the round-4 fast Barnes-Hut walk over child groups (bh_walk.hip, NBX_OPT_BH_WALK = 1, the default) against
  * the oracle's traversal (oracle/nbody_oracle.c, nbody.rs:333-377) -- forces to the fast mode's tolerance,
  * the node walk of rounds 1-3 (NBX_OPT_BH_WALK = 0) -- same decisions (equal work counters), forces to rounding,
  * its own per-lane form -- bit for bit (tests/test_gpu_bh_device_tree.py parametrises those over both walks),
and the threshold it decides with (bh_threshold.h) against the reference's own sqrt-and-divide test, float by float."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _engine(rx, p, walk, tree=None, fold=None, wave=None):
    from rust_exp_amd.engine import NBX_OPT_BH_WALK, NBX_OPT_BH_WAVE

    e = rx.NBodyEngine()
    if fold:
        e.set_bh_fold(fold)
    if tree:
        e.set_bh_tree(tree)
    e.set_option(NBX_OPT_BH_WALK, walk)
    if wave is not None:
        e.set_option(NBX_OPT_BH_WAVE, wave)
    e.set_particles(p["px"], p["py"], p["vx"], p["vy"], p["m"])
    return e


def _reference_accepts(s, x, theta):
    """nbody.rs:344-345 in numpy float32 (correctly rounded sqrt and divide): s / sqrt(dist_sq) < theta."""
    with np.errstate(divide="ignore", invalid="ignore"):
        return (np.float32(s) / np.sqrt(np.float32(x), dtype=np.float32)) < np.float32(theta)


def _threshold_cases():
    rng = np.random.default_rng(11)
    s = np.concatenate([
        np.exp(rng.uniform(np.log(1e-6), np.log(2e2), 4000)),            # node sizes of real trees, and far beyond
        [0.0, 1e-38, 1e-45, 1.0, 100.0, 3e38, np.inf, 0.5, 0.25, 64.0, 1e-20, 1e20],
        2.0 ** rng.integers(-20, 8, 500),                                  # lattice-like sizes: exact ties with power-of-two distances
    ]).astype(np.float32)
    theta = np.concatenate([
        rng.choice([0.05, 0.1, 0.25, 0.3, 0.5, 0.85, 0.95, 1.0, 2.0], 4000),
        [0.5, 0.5, 0.5, 0.0, -1.0, 0.5, 0.5, np.nan, 1e-30, 1e30, 0.5, 0.5],
        rng.choice([0.25, 0.5, 1.0], 500),
    ]).astype(np.float32)
    return s, theta


def test_take_threshold_is_the_references_test_on_host_and_device(rx):
    """T = bh_take_threshold(s, theta) must satisfy, in the reference's own f32 arithmetic: the reference does NOT accept
    dist_sq = T and DOES accept the next float above it (so, the test being monotone, accept <=> dist_sq > T everywhere);
    T = +inf where nothing is accepted.  Host (nbx_bh_take_threshold) and device (k_bh_thresholds) must agree bit for bit."""
    s, theta = _threshold_cases()
    L = rx.lib()
    host = np.array([L.nbx_bh_take_threshold(float(a), float(b)) for a, b in zip(s, theta)], dtype=np.float32)
    e = rx.NBodyEngine()
    dev = e.bh_take_thresholds(s, theta)
    assert np.array_equal(host.view(np.uint32), dev.view(np.uint32))
    T = host
    assert np.all(T >= 0)
    never = ~_reference_accepts(s, np.float32(np.inf), theta)
    assert np.all(np.isinf(T[never]))
    ok = ~never
    assert not np.any(_reference_accepts(s[ok], T[ok], theta[ok]))
    with np.errstate(over="ignore"):                  # (T = FLT_MAX -> +inf: the threshold of "accepted at infinity only")
        up = np.nextafter(T[ok], np.float32(np.inf), dtype=np.float32)
    assert np.all(_reference_accepts(s[ok], up, theta[ok]))
    # and on a window of floats around T: the decision is "dist_sq > T", float by float
    for k in (-3, -2, -1, 1, 2, 3, 50):
        x = (T[ok].view(np.uint32).astype(np.int64) + k)
        good = (x >= 0) & (x <= 0x7F800000)
        xf = x[good].astype(np.uint32).view(np.float32)
        assert np.array_equal(_reference_accepts(s[ok][good], xf, theta[ok][good]), xf > T[ok][good]), k


@pytest.mark.parametrize("make,n,theta", [("orbits", 3, 0.5), ("disk", 100, 0.85), ("orbits", 10000, 0.85), ("disk", 10000, 0.5),
                                          ("plummer", 65536, 0.5), ("plummer", 200000, 0.3), ("orbits", 300000, 1.0)])
def test_group_walk_makes_the_node_walks_decisions_and_the_oracles_forces(rx, ob, make, n, theta):
    if make == "plummer":
        st = rx.plummer_sphere(n, dim=2)
        p = ob.particles(st["px"], st["py"], st["vx"], st["vy"], st["m"])
    else:
        p = ob.stable_orbits(n, 0.5, 30.0, 3) if make == "orbits" else ob.random_disk(n, 3)
    tree = "host" if n < 1000 else "device"      # both builds; the host tree below 65 536 bodies is walked lane by lane
    eg = _engine(rx, p, 1, tree=tree, fold="reference" if n <= 65536 else "exact")
    en = _engine(rx, p, 0, tree=tree, fold="reference" if n <= 65536 else "exact")
    wg, wn = eg.bh_work(theta), en.bh_work(theta)
    assert wg == wn, (wg, wn)                    # same children tested, same pair laws: the same decisions
    gx, gy, _ = eg.forces(theta)
    nx, ny, _ = en.forces(theta)
    ec = _engine(rx, p, 2, tree=tree, fold="reference" if n <= 65536 else "exact")   # the compiled form of the group walk
    cx, cy, _ = ec.forces(theta)
    assert np.array_equal(gx.view(np.uint32), cx.view(np.uint32)) and np.array_equal(gy.view(np.uint32), cy.view(np.uint32))
    scale = max(np.abs(nx).max(), np.abs(ny).max())
    # same terms; they differ by the order of the additions and by d^2 being formed with or without an FMA: a few ulps of the
    # largest partial sum (the sun of nb_stable_orbits sums 10^4 terms of alternating sign: 1.4e-5 of its force)
    assert max(np.abs(gx - nx).max(), np.abs(gy - ny).max()) <= 2e-5 * scale
    if n <= 65536:                                # the tree is the oracle's, node for node: the oracle's own traversal applies
        rc, ofx, ofy = ob.bh_forces(p, theta, nthreads=8)
        assert rc == 0
        oscale = max(np.abs(ofx).max(), np.abs(ofy).max())
        err = np.maximum(np.abs(gx - ofx), np.abs(gy - ofy)) / oscale
        assert err.max() <= 2e-5 and np.percentile(err, 99) <= 1e-5, (err.max(), np.percentile(err, 99))


@pytest.mark.parametrize("n", [2000, 70000])
def test_group_walk_steps_like_the_node_walk(rx, ob, n):
    """Ten steps, device tree (gated, asynchronous steps included): positions and velocities of the two walks stay within the
    fast mode's per-step tolerance of each other, and the group walk's first step within it of the oracle's."""
    from conftest import fast_tolerances

    p = ob.stable_orbits(n, 0.5, 30.0, 8)
    tol_dp, tol_dv = fast_tolerances(ob, p, 0.01, steps=1) if n <= 4096 else (2e-4, 2e-2)
    eg, en = _engine(rx, p, 1), _engine(rx, p, 0)
    eg.step_barnes_hut(0.85, 0.01, 1)
    en.step_barnes_hut(0.85, 0.01, 1)
    a, b = eg.get_particles(), en.get_particles()
    assert np.abs(a["px"] - b["px"]).max() <= tol_dp and np.abs(a["vx"] - b["vx"]).max() <= tol_dv
    o = p.copy()
    assert ob.step_barnes_hut(o, 0.85, 0.01, 8) == 0
    assert np.abs(a["px"] - o["px"]).max() <= tol_dp and np.abs(a["py"] - o["py"]).max() <= tol_dp
    assert np.abs(a["vx"] - o["vx"]).max() <= tol_dv and np.abs(a["vy"] - o["vy"]).max() <= tol_dv
    for _ in range(9):
        eg.step_barnes_hut(0.85, 0.01, 1)
    eg.synchronize()
    c = eg.get_particles()
    assert np.all(np.isfinite(c["px"])) and np.abs(c["px"] - a["px"]).max() > 0


def test_group_walk_on_a_deep_tree_spills_its_stack_correctly(rx, ob):
    """Two tight clumps far apart plus a sparse halo: the leaves under one wave sit ~24 levels down, so the wave's stack of
    pending sibling groups passes its 64 register-resident entries and uses the LDS spill.  Wave walk == lane walk, bit for bit,
    and both within tolerance of the oracle (the tight clump is 1 500 bodies 2 EPS wide: beyond what the exact-sum class's chain replay
    reproduces, so the build hands this system to the host tree -- tests/test_gpu_bh_chains.py)."""
    from rust_exp_amd.engine import NBX_OPT_BH_WAVE

    rng = np.random.default_rng(4)
    pts = []
    for cx, cy, scale in ((-40.0, -40.0, 1e-3), (40.0, 40.0, 2e-4), (0.0, 0.0, 30.0)):
        m = 1500
        pts.append(np.stack([cx + scale * rng.standard_normal(m), cy + scale * rng.standard_normal(m)], 1))
    xy = np.concatenate(pts).astype(np.float32)
    xy = xy[rng.permutation(len(xy))]
    n = len(xy)
    p = ob.particles(xy[:, 0], xy[:, 1], np.zeros(n), np.zeros(n), rng.uniform(0.1, 1.5, n))
    res = []
    for wave in (0, 1):
        e = _engine(rx, p, 1, tree="device", fold="exact", wave=wave)
        res.append(e.forces(0.3)[:2])
    for a, b in zip(res[0], res[1]):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    rc, ofx, ofy = ob.bh_forces(p, 0.3, nthreads=8)
    assert rc == 0
    scale = max(np.abs(ofx).max(), np.abs(ofy).max())
    err = np.maximum(np.abs(res[1][0] - ofx), np.abs(res[1][1] - ofy)) / scale
    assert np.percentile(err, 99.9) <= 2e-5 and err.max() <= 2e-3, (np.percentile(err, 99.9), err.max())


def test_group_walk_degenerate_trees(rx, ob):
    """One body (the root is a leaf), two coincident bodies (merged leaf), bodies on a vertical line (root box of zero width:
    s = 0 everywhere), theta at and below zero (nothing is ever accepted: every leaf is reached)."""
    cases = {
        "one": ob.particles([1.0], [2.0], [0.0], [0.0], [1.0]),
        "coincident": ob.particles([1.0, 1.0], [2.0, 2.0], [0, 0], [0, 0], [1.0, 2.0]),
        "vertical": ob.particles(np.zeros(300), np.linspace(-20, 20, 300), np.zeros(300), np.zeros(300), np.ones(300)),
    }
    for name, p in cases.items():
        for theta in (0.5, 1e-6, -0.25):
            for tree, fold in (("host", None), ("device", "reference"), ("device", "exact")):
                e = _engine(rx, p, 1, tree=tree, fold=fold)
                gx, gy, _ = e.forces(theta)
                # (the exact-sum class against the oracle's traversal with exactly summed nodes: 300 bodies on a line take the ROOT at
                #  theta 0.5 -- s = 0 -- and the reference's running f32 fold of its centre is 4e-5 of max|F| off the exact one)
                rc, ofx, ofy = ob.bh_forces_exact(p, theta) if fold == "exact" else ob.bh_forces(p, theta)
                assert rc == 0
                scale = max(np.abs(ofx).max(), np.abs(ofy).max(), 1e-30)
                assert max(np.abs(gx - ofx).max(), np.abs(gy - ofy).max()) <= 1e-5 * scale, (name, theta, tree)


@pytest.mark.parametrize("n,async_,tree", [(300, 1, "device"), (3000, 1, "device"), (12000, 0, "device"), (12000, 1, "device"),
                                          (100000, 1, "device"), (600000, 1, "device"), (70000, 1, "host")])
def test_kick_drift_folded_into_the_walk_changes_no_bit(rx, ob, n, async_, tree):
    """NBX_OPT_BH_FUSE_KICK (round 4, default on): the walk kernel applies the kick-drift
    and the velocity kill itself.  Same operations on the same acceleration: positions and velocities equal the separate
    kick-drift kernel's bit for bit, step after step -- bodies that cross the +-55 kill box included -- in the waiting and the
    pipelined form of the step, and when a refused device build hands the step to the host tree."""
    from rust_exp_amd.engine import NBX_OPT_BH_ASYNC, NBX_STAT_BH_FALLBACKS, NBX_OPT_BH_FUSE_KICK, NBX_STAT_BH_LAST_TREE

    p = ob.stable_orbits(n, 0.5, 30.0, 77)
    rng = np.random.default_rng(n)
    fast = rng.choice(n, min(max(2, n // 50), 240), replace=False)     # bodies near the right wall of the kill box
    half = len(fast) // 2
    p["px"][fast] = np.linspace(54.0, 54.9, len(fast)).astype(np.float32)
    p["vx"][fast[:half]] = np.float32(1000.0)                          # these leave it in the first step (dx = 10), the others stay
    p["vx"][fast[half:]] = np.float32(-5.0)
    res = []
    for fuse in (0, 1):
        e = _engine(rx, p, 1, tree=tree)       # (host tree: the wave walk -- and with it the folded kick -- from 65 536 bodies on)
        e.set_option(NBX_OPT_BH_ASYNC, async_)
        e.set_option(NBX_OPT_BH_FUSE_KICK, fuse)
        e.step_barnes_hut(0.6, 0.01, 1)
        first = e.get_particles()
        killed = np.abs(first["px"][fast]) > 55.0                  # the velocity kill of nbody.rs:466-471 took place
        assert killed[:half].all() and not killed[half:].any()
        assert np.all(first["vx"][fast[:half]] == 0.0) and np.all(first["vx"][fast[half:]] != 0.0)
        for _ in range(4):
            e.step_barnes_hut(0.6, 0.01, 1)
        res.append(e.get_particles())
        assert e.get_stat(NBX_STAT_BH_LAST_TREE) == (1 if tree == "device" else 0) and e.get_stat(NBX_STAT_BH_FALLBACKS) == 0
    for k in ("px", "py", "vx", "vy"):
        assert np.array_equal(res[0][k].view(np.uint32), res[1][k].view(np.uint32)), k
    assert np.abs(res[0]["px"] - p["px"]).max() > 0


@pytest.mark.parametrize("fold", ["reference", "exact"])
def test_a_refused_build_stops_the_folded_kick_too(rx, ob, fold):
    """The gate of the pipelined step (bh_gate.h) in the kernel that now ends it: a device build that must refuse (node pool
    exhausted by thousands of 18-level chains; EPS triples under the reference fold) leaves the state alone, the step is redone on
    the host tree, the step enqueued behind it is enqueued again -- the same states as with the separate kick-drift kernel."""
    from rust_exp_amd.engine import NBX_STAT_BH_FALLBACKS, NBX_OPT_BH_FUSE_KICK

    rng = np.random.default_rng(8)
    x = rng.uniform(-20, 20, 4000).astype(np.float32); y = rng.uniform(-20, 20, 4000).astype(np.float32)
    x2 = np.concatenate([x, x + np.float32(2e-4), x[:3] + np.float32(3e-5), x[:3] - np.float32(2e-5)])
    y2 = np.concatenate([y, y, y[:3] + np.float32(1e-5), y[:3] + np.float32(4e-5)])
    n = len(x2)
    q = ob.particles(x2, y2, rng.normal(0, 1, n), rng.normal(0, 1, n), np.ones(n))
    res, fallbacks = [], []
    for fuse in (0, 1):
        e = _engine(rx, q, 1, tree="device", fold=fold)
        e.set_option(NBX_OPT_BH_FUSE_KICK, fuse)
        for _ in range(4):
            e.step_barnes_hut(0.5, 0.01, 1)
        res.append(e.get_particles())
        fallbacks.append(e.get_stat(NBX_STAT_BH_FALLBACKS))
    assert fallbacks[0] >= 1 and fallbacks[0] == fallbacks[1]
    for k in ("px", "py", "vx", "vy"):
        assert np.array_equal(res[0][k].view(np.uint32), res[1][k].view(np.uint32)), k


def test_walk_trace_reports_every_walk(rx):
    """nbx_bh_walk_trace (tools/bh_walk_trace.py): one record per workgroup of the walk kernel -- start <= end on the device-wide
    clock, the groups it loaded, and every chunk of 64 bodies exactly once."""
    n = 100000
    st = rx.plummer_sphere(n, dim=2)
    e = rx.NBodyEngine()
    e.set_bh_fold("exact")
    e.set_particles(st["px"], st["py"], st["vx"], st["vy"], st["m"])
    fx0, fy0, _ = e.forces(0.5)
    tr = e.bh_walk_trace(0.5)
    ran = tr[:, 1] > 0
    bpw = 64
    while bpw > 2 and (n + bpw - 1) // bpw < 4096:          # the launcher's rule: at least 4 096 walks, 2 ... 64 bodies each
        bpw >>= 1
    assert bpw == 16 and ran.sum() == (n + bpw - 1) // bpw
    assert np.all(tr[ran, 1] >= tr[ran, 0])
    turns = (tr[ran, 2] & np.uint64(0x7FFFFFFF)).astype(np.int64)
    chunk = (tr[ran, 2] >> np.uint64(32)).astype(np.int64)
    assert turns.min() >= 1 and turns.max() < 5000
    assert np.array_equal(np.sort(chunk), np.arange(ran.sum()))
    fx1, fy1, _ = e.forces(0.5)                             # tracing changed nothing
    assert np.array_equal(fx0.view(np.uint32), fx1.view(np.uint32)) and np.array_equal(fy0.view(np.uint32), fy1.view(np.uint32))


def test_a_walk_that_outgrows_its_register_stack_is_redone_with_the_spill(rx, ob):
    """64 entries of pending groups live in VGPR lanes.  A chain of 24 nested cells towards one corner of a 65 536-wide box, with a
    small clump (an interior node of four leaves) in each of the three sibling quadrants on every level: the bodies in the innermost
    cell open every clump (theta = 0.3: the diagonal sibling is 0.47 of its distance wide), and while the walk descends the chain (slot 0 on every level) three opened siblings per level wait on
    the stack -- 72 entries.  The hand-scheduled loop then leaves with its overflow flag and the wave redoes its walk in the
    compiled form with the LDS spill: the trace must show such a walk, and the forces must equal the per-lane walk's bit for bit."""
    pts = [(65536.0, -65536.0)]                           # pins the root box to [0, 65536] x [-65536, 0]
    for L in range(1, 25):
        w = 65536.0 / 2 ** L                              # the chain's cell on level L is [0, w] x [-w, 0] (upper left: slot 0)
        for cx, cy in ((1.5 * w, -0.5 * w), (0.5 * w, -1.5 * w), (1.5 * w, -1.5 * w)):
            for dx, dy in ((-0.2, -0.2), (0.2, -0.2), (-0.2, 0.2), (0.2, 0.2)):
                pts.append((cx + dx * w, cy + dy * w))
    w = 65536.0 / 2 ** 24
    for dx, dy in ((0.0, 0.0), (0.3, -0.3), (0.6, -0.2), (0.2, -0.7)):
        pts.append((dx * w, dy * w))                      # the innermost cell's bodies (every clump is >= 4e-4 across: no EPS merge)
    xy = np.array(pts, dtype=np.float32)
    n = len(xy)
    p = ob.particles(xy[:, 0], xy[:, 1], np.zeros(n), np.zeros(n), np.ones(n))
    res, spilled = [], 0
    for wave in (0, 1):
        e = _engine(rx, p, 1, tree="device", fold="exact", wave=wave)
        res.append(e.forces(0.3)[:2])
        if wave:
            tr = e.bh_walk_trace(0.3)
            spilled = int(((tr[:, 2] >> np.uint64(31)) & np.uint64(1)).sum())
    assert spilled > 0, "no walk outgrew the register stack: the case does not exercise the spill"
    for a, b in zip(res[0], res[1]):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    # ... and against the all-pairs sum of the oracle: the walk's approximation at theta = 0.3, nothing worse
    fx, fy = ob.brute_forces(p)
    scale = max(np.abs(fx).max(), np.abs(fy).max())
    assert max(np.abs(res[1][0] - fx).max(), np.abs(res[1][1] - fy).max()) <= 0.05 * scale


def test_blocking_host_waits_give_the_same_state(rx):
    """NBX_SPIN_US = 0: the host waits of the stepping path block at once instead of polling first (engine_internal.h).  Read once
    per process, so the blocking form runs in a child process: same bits after the same steps."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, zlib, numpy as np; sys.path.insert(0, %r); import rust_exp_amd as rx\n"
            "st = rx.plummer_sphere(20000, dim=2); e = rx.NBodyEngine()\n"
            "e.set_particles(st['px'], st['py'], st['vx'], st['vy'], st['m'])\n"
            "for _ in range(6):\n    e.step_barnes_hut(0.6, 0.01, 1)\n    e.synchronize()\n"
            "p = e.get_particles(); print(zlib.crc32(np.concatenate([p[k] for k in ('px', 'py', 'vx', 'vy')]).tobytes()))\n" % root)
    sums = []
    for spin in ("0", "400"):
        env = dict(os.environ, NBX_SPIN_US=spin)
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        sums.append(out.stdout.strip().splitlines()[-1])
    assert sums[0] == sums[1] and sums[0].isdigit()
