"""GPU: quadtree built ON THE DEVICE (bh_build.hip; the fast mode's default from 1 024 bodies on) against the host build,
which is node-for-node the oracle's tree.  Two classes (NBX_OPT_BH_FOLD):
  * fold = reference (round 3; on request in the fast mode since round 6): interior masses and centres are the reference's own f32 running
    fold in arrival order (nbody.rs:303-320) -> the flattened tree equals the host tree BIT FOR BIT, or the build reports EPS
    clusters it cannot reproduce node for node and the step runs on the host tree;
  * fold = exact (round 2; the fast mode's default at every size since round 6 -- by cost): same node set / skip pointers / node sizes / leaf records -- including the
    reference's EPS merge (round 6: chains of any length, tests/test_gpu_bh_chains.py) -- with interior records that are roundings of the EXACT sums (the reference's fold
    drifts, 6e-4 at a million bodies), so forces are compared with the oracle AND with the fp64 arbiter
    (oracle/nbody_oracle.c orc_bh_forces_exact)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def engines(rx, p, fold=None):
    e = rx.NBodyEngine()
    if fold:
        e.set_bh_fold(fold)
    e.set_particles(p["px"], p["py"], p["vx"], p["vy"], p["m"])
    return e


def _bit_equal_trees(host, dev):
    assert len(host) == len(dev), (len(host), len(dev))
    for k in ("skip", "interior"):
        assert np.array_equal(host[k], dev[k]), k
    for k in ("px", "py", "m", "s", "q"):
        bad = np.flatnonzero(host[k].view(np.uint32) != dev[k].view(np.uint32))
        assert bad.size == 0, (k, bad.size, bad[:5], host[k][bad[:5]], dev[k][bad[:5]])


@pytest.mark.parametrize("make,n", [("disk", 600), ("disk", 5000), ("orbits", 10000), ("orbits", 20000), ("plummer", 65536),
                                    ("disk", 65536), ("plummer", 100000)])
def test_device_tree_with_the_reference_fold_equals_the_host_tree_bit_for_bit(rx, ob, make, n):
    """every record of the device-built flattened tree -- interior (px, py, m) included -- equals the host
    (= oracle) tree's, bit for bit: the f32 running fold of nbody.rs:303-320 replayed in arrival order per node (k_emit for
    nodes of <= 8 bodies, k_fold_big: rank / bitmap ordering + the m and p chains).  100 000 bodies: the same with the fold
    forced on above its default range (two index windows of the bitmap path)."""
    from rust_exp_amd.engine import NBX_OPT_BH_FOLD

    if make == "disk":
        p = ob.random_disk(n, 41)
    elif make == "orbits":
        p = ob.stable_orbits(n, 0.5, 30.0, 42)
    else:
        st = rx.plummer_sphere(n, dim=2)
        p = ob.particles(st["px"], st["py"], st["vx"], st["vy"], st["m"])
    e = engines(rx, p, fold="reference")
    assert e.query_option(NBX_OPT_BH_FOLD) == 1
    _bit_equal_trees(e.bh_flat_dump(False), e.bh_flat_dump("device"))
    # and the forces through it are the host-tree forces bit for bit (same walk over the same records), hence within the
    # fast mode's 2e-5 of the oracle for EVERY body
    a = engines(rx, p); a.set_bh_tree("host")
    b = engines(rx, p, fold="reference"); b.set_bh_tree("device")
    from rust_exp_amd.engine import NBX_STAT_BH_LAST_TREE

    for theta in (0.5, 0.85):
        fx, fy, _ = a.forces(theta)
        gx, gy, _ = b.forces(theta)
        assert a.get_stat(NBX_STAT_BH_LAST_TREE) == 0 and b.get_stat(NBX_STAT_BH_LAST_TREE) == 1
        assert np.array_equal(fx.view(np.uint32), gx.view(np.uint32)) and np.array_equal(fy.view(np.uint32), gy.view(np.uint32))
        if n <= 20000:
            rc, ofx, ofy = ob.bh_forces(p, theta, nthreads=8)
            scale = max(np.abs(ofx).max(), np.abs(ofy).max())
            assert rc == 0 and np.abs(gx - ofx).max() <= 2e-5 * scale and np.abs(gy - ofy).max() <= 2e-5 * scale


def test_device_tree_reference_fold_replays_eps_clusters(rx, ob):
    """Blobs of three bodies within EPS (nbody.rs:249-260: arrivals folded into a blob while they stay within EPS of its
    moving centre): the reference-fold class replays every connected cluster's arrivals in index order on the device (k_blobs)
    and the flattened tree is the host tree, bit for bit -- no hand-over.  (The exact-sum class merges pairs only and tolerates
    up to max(16, n/2000) bodies left behind.)"""
    from rust_exp_amd.engine import NBX_STAT_BH_FALLBACKS, NBX_STAT_BH_LAST_TREE

    rng = np.random.default_rng(17)
    x = rng.uniform(-20, 20, 6000).astype(np.float32); y = rng.uniform(-20, 20, 6000).astype(np.float32)
    # three triples within EPS of each other
    x = np.concatenate([x, x[:3] + np.float32(3e-5), x[:3] - np.float32(2e-5)])
    y = np.concatenate([y, y[:3] + np.float32(1e-5), y[:3] + np.float32(4e-5)])
    n = len(x)
    p = ob.particles(x, y, np.zeros(n), np.zeros(n), rng.uniform(0.5, 2.0, n))
    a = engines(rx, p); a.set_bh_tree("host")
    b = engines(rx, p, fold="reference")                 # device tree, reference fold (rounds 3-5: the default up to 65 536 bodies)
    c = engines(rx, p, fold="exact")
    fx, fy, _ = a.forces(0.5)
    gx, gy, _ = b.forces(0.5)
    hx, hy, _ = c.forces(0.5)
    assert b.get_stat(NBX_STAT_BH_FALLBACKS) == 0 and b.get_stat(NBX_STAT_BH_LAST_TREE) == 1
    assert c.get_stat(NBX_STAT_BH_FALLBACKS) == 0 and c.get_stat(NBX_STAT_BH_LAST_TREE) == 1
    assert np.array_equal(gx.view(np.uint32), fx.view(np.uint32)) and np.array_equal(gy.view(np.uint32), fy.view(np.uint32))
    host, dev = b.bh_flat_dump(False), b.bh_flat_dump("device")
    _bit_equal_trees(host, dev)
    assert int((host["interior"] == 0).sum()) == 6000    # three bodies per blob, one leaf each
    scale = max(np.abs(fx).max(), np.abs(fy).max())
    assert np.abs(hx - fx).max() <= 2e-3 * scale


def _clumps(rng, n_base, n_clumps, spread, box=20.0, masses=(0.5, 2.0), max_members=6):
    x = rng.uniform(-box, box, n_base).astype(np.float32); y = rng.uniform(-box, box, n_base).astype(np.float32)
    xs, ys = [x], [y]
    for c in range(n_clumps):
        k = int(rng.integers(1, max_members))
        xs.append((x[c] + rng.normal(0, spread, k)).astype(np.float32))
        ys.append((y[c] + rng.normal(0, spread, k)).astype(np.float32))
    x, y = np.concatenate(xs), np.concatenate(ys)
    perm = rng.permutation(len(x))
    x, y = x[perm], y[perm]
    return x, y, rng.uniform(masses[0], masses[1], len(x)).astype(np.float32)


@pytest.mark.parametrize("seed,n_base,n_clumps,spread,on_device", [
    (1, 4000, 300, 1.5e-5, True), (2, 20000, 600, 1e-5, True), (3, 600, 150, 2e-5, True), (4, 60000, 500, 1.5e-5, True),
    (12, 8000, 800, 3e-5, True), (5, 4000, 300, 4e-5, None), (6, 4000, 300, 1e-4, None), (7, 20000, 2000, 7e-5, None),
    (8, 600, 200, 2e-4, None), (9, 3000, 3000, 6e-5, False)])
def test_device_tree_reference_fold_clusters_in_random_arrival_order(rx, ob, seed, n_base, n_clumps, spread, on_device):
    """Hundreds of clusters of 2 .. 6 bodies, everything arriving in random order (members of a cluster interleaved with thousands
    of other bodies; clusters that straddle cell boundaries of every level).  Tight clusters (every member within EPS of the
    moving centre) the device build replays and files under their centre's path: the host tree bit for bit.  Looser ones leave
    unmerged bodies a fraction of EPS beside a blob of several bodies, its leaf is then ~18 levels deep and the blob's successive
    centres often sit in different cells at that depth: the build says so (NBX_STAT_BH_REFUSAL: 0x80) and the evaluation runs on the class
    below -- since round 6 the exact-sum DEVICE build in the fast mode, the host tree only when that refuses as well (more
    unmerged bodies than it tolerates).  Kept or handed to the host, the forces are the host tree's, bit for bit; a system made of nothing but clusters (the last case: more bodies to move
    than the build lists, more nodes than its pool holds) goes to the host build as a whole."""
    from rust_exp_amd.engine import NBX_STAT_BH_FALLBACKS, NBX_STAT_BH_LAST_TREE

    rng = np.random.default_rng(1000 + seed)
    x, y, m = _clumps(rng, n_base, n_clumps, spread)
    n = len(x)
    p = ob.particles(x, y, np.zeros(n), np.zeros(n), m)
    from rust_exp_amd.engine import NBX_STAT_BH_CLASS_SWITCHES, NBX_STAT_BH_REFUSAL

    h = engines(rx, p); h.set_bh_tree("host")
    d = engines(rx, p, fold="reference"); d.set_bh_tree("device")   # (rounds 3-5: the default class up to 65 536 bodies)
    fx, fy, _ = h.forces(0.85)
    gx, gy, _ = d.forces(0.85)
    sw, fb = d.get_stat(NBX_STAT_BH_CLASS_SWITCHES), d.get_stat(NBX_STAT_BH_FALLBACKS)
    device = sw == 0 and fb == 0                           # the reference-fold build was kept
    assert (d.get_stat(NBX_STAT_BH_LAST_TREE) == 1) == (fb == 0)
    why = d.get_stat(NBX_STAT_BH_REFUSAL)                  # the build says why it handed the system on (the reasons of the class asked for
    assert (why == 0) == device and (device or why & (0x10000 | 1 | 4 | 8 | 16 | 32 | 64 | 128))   # stay on record when the class below refuses too)
    same = np.array_equal(fx.view(np.uint32), gx.view(np.uint32)) and np.array_equal(fy.view(np.uint32), gy.view(np.uint32))
    if device or fb == 1:
        assert same                                        # the host tree's forces bit for bit: its own records, or its own build
    else:
        # round 6: refused in the fast mode -> served by the exact-sum DEVICE build: that engine's forces bit for bit (its class:
        # pairs merged, a few bodies of bigger clusters left unmerged)
        assert sw == 1
        x_ = engines(rx, p, fold="exact"); x_.set_bh_tree("device")
        ex, ey, _ = x_.forces(0.85)
        assert x_.get_stat(NBX_STAT_BH_LAST_TREE) == 1
        assert np.array_equal(ex.view(np.uint32), gx.view(np.uint32)) and np.array_equal(ey.view(np.uint32), gy.view(np.uint32))
        err = np.maximum(np.abs(gx - fx), np.abs(gy - fy)) / max(np.abs(fx).max(), np.abs(fy).max())
        assert np.percentile(err, 99.0) <= 4e-4, np.percentile(err, 99.0)
    if on_device is not None:
        assert device == on_device
    if device:
        host, dev = d.bh_flat_dump(False), d.bh_flat_dump("device")
        _bit_equal_trees(host, dev)
        assert int((host["interior"] == 0).sum()) < n - n_clumps // 4     # blobs did form


@pytest.mark.parametrize("make,n", [("disk", 5000), ("orbits", 20000), ("plummer", 100000), ("tiny", 2), ("one", 1)])
def test_device_tree_has_the_host_trees_structure(rx, ob, make, n):
    if make == "disk":
        p = ob.random_disk(n, 41)
    elif make == "orbits":
        p = ob.stable_orbits(n, 0.5, 30.0, 42)
    elif make == "plummer":
        st = rx.plummer_sphere(n, dim=2)
        p = ob.particles(st["px"], st["py"], st["vx"], st["vy"], st["m"])
    else:
        p = ob.random_disk(n, 43)
    # keep only bodies farther apart than EPS from every other body in both dims: no reference EPS merges,
    # so the two trees must have identical shape
    if len(p) > 2:
        from scipy.spatial import cKDTree

        pts = np.stack([p["px"], p["py"]], 1).astype(np.float64)
        pairs = cKDTree(pts).query_pairs(2.0e-4, p=np.inf, output_type="ndarray")
        keep = np.ones(len(p), bool)
        keep[pairs[:, 1]] = False
        p = p[keep]
    e = engines(rx, p, fold="exact")
    host = e.bh_flat_dump(False)
    dev = e.bh_flat_dump("device")
    assert len(host) == len(dev)
    assert np.array_equal(host["skip"], dev["skip"]) and np.array_equal(host["interior"], dev["interior"])
    assert np.array_equal(host["s"].view(np.uint32), dev["s"].view(np.uint32))           # node sizes: same f32 midpoints
    leaf = host["interior"] == 0
    for k in ("px", "py", "m"):
        assert np.array_equal(host[k][leaf].view(np.uint32), dev[k][leaf].view(np.uint32)), k   # leaf = the particle
    inner = ~leaf
    if inner.any():
        # interior mass / centre of mass: the reference folds particle by particle in f32 (a 100k-term running
        # sum drifts by ~6e-4 relative: the host root of the Plummer case holds 1000.6 for a true 1000.0), the
        # device folds child by child. Both are roundings of the same quantity:
        msum = p["m"].astype(np.float64).sum()
        assert abs(float(dev["m"][0]) - msum) <= abs(float(host["m"][0]) - msum) + 1e-6 * msum   # device root at least as accurate
        assert np.abs(host["m"][inner] - dev["m"][inner]).max() <= 2e-3 * host["m"][inner].max()
        scale = max(np.abs(host["px"]).max(), np.abs(host["py"]).max(), 1e-3)
        assert np.abs(host["px"][inner] - dev["px"][inner]).max() <= 2e-3 * scale
        assert np.abs(host["py"][inner] - dev["py"][inner]).max() <= 2e-3 * scale
        assert np.median(np.abs(host["px"][inner] - dev["px"][inner])) <= 1e-6 * scale


@pytest.mark.parametrize("theta", [0.5, 0.85])
def test_device_tree_forces_and_step_match_host_tree(rx, ob, theta):
    p = ob.stable_orbits(50000, 0.5, 30.0, 44)
    a, b = engines(rx, p), engines(rx, p)                # (b: the fast mode's default class since round 6 -- exact sums)
    a.set_bh_tree("host")
    b.set_bh_tree("device")
    fx, fy, _ = a.forces(theta)
    gx, gy, _ = b.forces(theta)
    scale = max(np.abs(fx).max(), np.abs(fy).max())
    assert np.abs(gx - fx).max() <= 5e-5 * scale and np.abs(gy - fy).max() <= 5e-5 * scale
    for _ in range(3):
        a.step_barnes_hut(theta, 0.01, 1)
        b.step_barnes_hut(theta, 0.01, 1)
    pa, pb = a.get_particles(), b.get_particles()
    # bodies grazing the 1000-mass sun (|a| ~ 4e4) amplify rounding-level force differences: bound the bulk
    # tightly and the worst case loosely
    assert np.median(np.abs(pa["px"] - pb["px"])) <= 1e-6 and np.abs(pa["px"] - pb["px"]).max() <= 5e-3
    assert np.median(np.abs(pa["vx"] - pb["vx"])) <= 1e-4 and np.abs(pa["vx"] - pb["vx"]).max() <= 0.5
    # the velocity-kill box (nbody.rs:466-471) applies on this path too
    q = ob.particles([0.0, 56.0, 3.0], [0.0, 0.0, 3.0], [0.0, 1.0, 0.0], [0.0, 1.0, 0.0], [1000.0, 1.0, 1.0])
    c = engines(rx, q, fold="exact")
    c.set_bh_tree("device")
    c.step_barnes_hut(0.5, 0.01, 1)
    st = c.get_particles()
    assert st["vx"][1] == 0 and st["vy"][1] == 0


def _structure_equal(host, dev):
    assert len(host) == len(dev), (len(host), len(dev))
    assert np.array_equal(host["skip"], dev["skip"]) and np.array_equal(host["interior"], dev["interior"])
    assert np.array_equal(host["s"].view(np.uint32), dev["s"].view(np.uint32))
    leaf = host["interior"] == 0
    for k in ("px", "py", "m"):
        assert np.array_equal(host[k][leaf].view(np.uint32), dev[k][leaf].view(np.uint32)), k
    return int(leaf.sum())


@pytest.mark.parametrize("n0,k,seed", [(5000, 800, 3), (5000, 2500, 4), (100000, 5000, 5), (300, 150, 6)])
def test_device_tree_reproduces_the_reference_eps_merge_in_arrival_order(rx, ob, n0, k, seed):
    """nbody.rs:249-260: k bodies get a partner closer than EPS in both axes, and the whole system arrives in RANDOM order (so
    half the partners come first, and other bodies arrive in between).  The reference merges a pair exactly when the later one
    finds the earlier one alone in a leaf that holds both; the device build decides the same from the sorted keys and the body
    indices.  Same node count, same skip pointers, same boxes, and every leaf -- merged blobs folded in arrival order included
    -- bit-equal to the host (= oracle) tree."""
    rng = np.random.default_rng(seed)
    x = rng.uniform(-20, 20, n0).astype(np.float32); y = rng.uniform(-20, 20, n0).astype(np.float32)
    from scipy.spatial import cKDTree

    keep = np.ones(n0, bool)       # base bodies farther than 4 EPS from each other: every cluster is a pair
    keep[cKDTree(np.stack([x, y], 1).astype(np.float64)).query_pairs(4.0e-4, p=np.inf, output_type="ndarray")[:, 1]] = False
    x, y = x[keep], y[keep]
    k = min(k, len(x))
    dx = rng.uniform(-9e-5, 9e-5, k).astype(np.float32); dy = rng.uniform(-9e-5, 9e-5, k).astype(np.float32)
    x = np.concatenate([x, x[:k] + dx]); y = np.concatenate([y, y[:k] + dy])
    perm = rng.permutation(len(x))
    x, y = x[perm], y[perm]
    n = len(x)
    p = ob.particles(x, y, np.zeros(n), np.zeros(n), rng.uniform(0.5, 2.0, n))
    e = engines(rx, p, fold="exact")
    host, dev = e.bh_flat_dump(False), e.bh_flat_dump("device")
    leaves = _structure_equal(host, dev)
    assert n - k <= leaves < n - k // 2          # most pairs merged (those split by a cell boundary at arrival time did not)
    # reference-fold class on the same system: the whole tree bit for bit, unless a blob's centre left its first member's cell
    # (then the build says so and the caller takes the host tree)
    r = engines(rx, p, fold="reference")
    try:
        _bit_equal_trees(host, r.bh_flat_dump("device"))
    except rx.NBodyError as ex:
        assert "refused" in str(ex)
    rc, ofx, ofy = ob.bh_forces(p, 0.5, nthreads=8)
    e.set_bh_tree("device")
    fx, fy, _ = e.forces(0.5)
    scale = max(np.abs(ofx).max(), np.abs(ofy).max())
    err = np.maximum(np.abs(fx - ofx), np.abs(fy - ofy)) / scale
    # same nodes, same leaves; interior centres differ in the last bit or two (exact mean vs running fold), which flips an
    # opening decision s/d < theta now and then (about one per million visits): that body is then off by one node's
    # Barnes-Hut approximation error. Everyone else is within the walk's fp32 rounding.
    assert rc == 0 and np.percentile(err, 99.9) <= 2e-5 and err.max() <= 2e-3, (np.percentile(err, 99.9), err.max())


def test_device_tree_duplicates_with_partners(rx, ob):
    """Bodies of one level-31 cell (exact duplicates) share a leaf in arrival order, and a close partner merges with them exactly
    like the reference does (nbody.rs:249-260) -- same tree as the host's.  (Clusters of three or more DISTINCT positions within
    EPS, which rounds 2-5 left to the host build when there were many: tests/test_gpu_bh_chains.py.)"""
    from rust_exp_amd.engine import NBX_STAT_BH_FALLBACKS, NBX_STAT_BH_LAST_TREE

    rng = np.random.default_rng(7)
    x = rng.uniform(-20, 20, 3000).astype(np.float32)
    y = rng.uniform(-20, 20, 3000).astype(np.float32)
    x = np.concatenate([x, x[:500] + np.float32(3e-5), x[:100], x[:100]])     # EPS-close pairs + exact triplicates
    y = np.concatenate([y, y[:500], y[:100], y[:100]])
    n = len(x)
    p = ob.particles(x, y, np.zeros(n), np.zeros(n), rng.uniform(0.5, 2.0, n))
    e = engines(rx, p, fold="exact")
    leaves = _structure_equal(e.bh_flat_dump(False), e.bh_flat_dump("device"))
    assert leaves == 3000                       # 500 pairs, 100 of them with two more bodies on top: one leaf each
    e.set_bh_tree("device")
    gx, gy, _ = e.forces(0.3)
    assert e.get_stat(NBX_STAT_BH_FALLBACKS) == 0 and e.get_stat(NBX_STAT_BH_LAST_TREE) == 1
    rc, ofx, ofy = ob.bh_forces(p, 0.3, nthreads=8)
    err = np.maximum(np.abs(gx - ofx), np.abs(gy - ofy)) / max(np.abs(ofx).max(), np.abs(ofy).max())
    assert rc == 0 and np.percentile(err, 99.9) <= 2e-5 and err.max() <= 2e-3


@pytest.mark.parametrize("make,n", [("orbits", 50000), ("disk", 20000), ("plummer", 262144)])
@pytest.mark.parametrize("theta", [0.5, 0.85])
def test_device_tree_exact_sums_are_at_least_as_close_to_exact_arithmetic_as_the_reference_fold(rx, ob, make, n, theta):
    """Forces through the fast walk of the device-built tree vs (a) the oracle and (b) the fp64 arbiter (the reference's tree
    and laws with exact node sums).  The reference's interior masses / centres are an f32 running fold over up to n bodies
    (nbody.rs:303-320) and drift; the device's are exact sums rounded once.  Stated tolerance of this path:
        |F_dev - F_arbiter| <= 2e-5 max|F|                            (fp32 rounding of the walk only)
        |F_dev - F_oracle|  <= |F_oracle - F_arbiter| + 2e-5 max|F|   (what separates it from the reference is the reference's own drift)
    for 99.9 % of the bodies; the rest may sit on a flipped opening decision (centres that differ in the last bits put s/d on
    the other side of theta about once per million visits) and are bounded by one node's approximation error: 2e-3 max|F|."""
    from rust_exp_amd.engine import NBX_STAT_BH_LAST_TREE

    if make == "orbits":
        p = ob.stable_orbits(n, 0.5, 30.0, 44)
    elif make == "disk":
        p = ob.random_disk(n, 41)
    else:
        st = rx.plummer_sphere(n, dim=2)
        p = ob.particles(st["px"], st["py"], st["vx"], st["vy"], st["m"])
    rc, ofx, ofy = ob.bh_forces(p, theta, nthreads=16)
    rc2, ex, ey = ob.bh_forces_exact(p, theta, nthreads=16)
    assert rc == 0 and rc2 == 0
    e = engines(rx, p)                       # fast mode, n >= 512 -> device tree; exact sums (round 6: the default at every size)
    from rust_exp_amd.engine import NBX_OPT_BH_FOLD
    assert e.query_option(NBX_OPT_BH_FOLD) == -1
    fx, fy, _ = e.forces(theta)
    assert e.get_stat(NBX_STAT_BH_LAST_TREE) == 1
    scale = max(np.abs(ex).max(), np.abs(ey).max())
    if n <= 65536:
        # the reference fold (on request since round 6): within the fast mode's 2e-5 of the oracle for every single body
        d = engines(rx, p, fold="reference")
        dx, dy, _ = d.forces(theta)
        assert d.get_stat(NBX_STAT_BH_LAST_TREE) == 1
        assert max(np.abs(dx - ofx).max(), np.abs(dy - ofy).max()) <= 2e-5 * max(np.abs(ofx).max(), np.abs(ofy).max())
    dev_arb = np.maximum(np.abs(fx - ex), np.abs(fy - ey)) / scale
    orc_arb = np.maximum(np.abs(ofx - ex), np.abs(ofy - ey)) / scale
    dev_orc = np.maximum(np.abs(fx - ofx), np.abs(fy - ofy)) / scale
    q = lambda a: float(np.percentile(a, 99.9))   # noqa: E731
    assert q(dev_arb) <= 2e-5 and dev_arb.max() <= 2e-3, (q(dev_arb), dev_arb.max(), q(orc_arb))
    assert q(dev_orc) <= q(orc_arb) + 2e-5 and dev_orc.max() <= orc_arb.max() + 2e-3, (q(dev_orc), q(orc_arb))


def test_tree_choice_by_mode_and_size(rx, ob):
    """NBX_OPT_BH_TREE = -1 (default): device build in the fast mode from 512 bodies on (exactly summed nodes: that mode's default class
    since round 6; 1 024 when the reference fold is asked for), host build below and in the bit-exact mode; 0 / 1 force one or the other."""
    from rust_exp_amd.engine import NBX_STAT_BH_LAST_TREE, NBX_OPT_BH_TREE

    for n, mode, want in ((4096, "fast", 1), (512, "fast", 1), (511, "fast", 0), (20000, "strict", 0)):
        p = ob.random_disk(n, 3)
        e = rx.NBodyEngine(mode=mode)
        assert e.get_option(NBX_OPT_BH_TREE) == -1
        e.set_particles(p["px"], p["py"], p["vx"], p["vy"], p["m"])
        e.step_barnes_hut(0.5, 0.01, 1)
        assert e.get_stat(NBX_STAT_BH_LAST_TREE) == want, (n, mode)
    e.set_bh_tree("device")                  # strict takes the device build only on request, and only with the reference fold
    e.step_barnes_hut(0.5, 0.01, 1)
    assert e.get_stat(NBX_STAT_BH_LAST_TREE) == 1
    e.set_bh_fold("exact")
    e.step_barnes_hut(0.5, 0.01, 1)
    assert e.get_stat(NBX_STAT_BH_LAST_TREE) == 0
    f = rx.NBodyEngine()
    f.set_bh_tree("host")
    f.set_particles(p["px"], p["py"], p["vx"], p["vy"], p["m"])
    f.step_barnes_hut(0.5, 0.01, 1)
    assert f.get_stat(NBX_STAT_BH_LAST_TREE) == 0
    for n, want in ((512, 1), (511, 0)):     # exactly summed nodes: no root chain, the device build pays earlier
        p = ob.random_disk(n, 3)
        g = engines(rx, p, fold="exact")
        g.step_barnes_hut(0.5, 0.01, 1)
        assert g.get_stat(NBX_STAT_BH_LAST_TREE) == want, n


@pytest.mark.parametrize("make,n", [("disk", 3000), ("orbits", 10000), ("clusters", 4892), ("plummer", 40000)])
def test_strict_mode_on_the_device_tree_is_still_the_oracle(rx, ob, make, n):
    """The bit-exact mode keeps the reference-faithful host build unless the caller selects the device build; with the reference
    fold that tree is the host tree bit for bit (or refused and built on the host), so three bit-exact steps on it are the
    oracle's three steps, bit for bit -- clusters of bodies within EPS included.  With exactly summed nodes (NBX_OPT_BH_FOLD = 0)
    the request is ignored: that tree is not the reference's."""
    from rust_exp_amd.engine import NBX_STAT_BH_LAST_TREE

    if make == "disk":
        p = ob.random_disk(n, 45)
    elif make == "orbits":
        p = ob.stable_orbits(n, 0.5, 30.0, 46)
    elif make == "clusters":
        x, y, m = _clumps(np.random.default_rng(1001), 4000, 300, 1.5e-5)
        assert len(x) == n
        p = ob.particles(x, y, np.zeros(n), np.zeros(n), m)
    else:
        st = rx.plummer_sphere(n, dim=2)
        p = ob.particles(st["px"], st["py"], st["vx"], st["vy"], st["m"])
    e = rx.NBodyEngine(mode="strict")
    e.set_bh_tree("device")
    e.set_particles(p["px"], p["py"], p["vx"], p["vy"], p["m"])
    q = p.copy()
    for k in range(3):
        e.step_barnes_hut(0.6, 0.01, 1)
        # (the clusters fall in on themselves: a later step may be one the device build hands to the host build)
        assert e.get_stat(NBX_STAT_BH_LAST_TREE) == 1 or (make == "clusters" and k > 0)
        ob.step_barnes_hut(q, 0.6, 0.01, 1)
    st = e.get_particles()
    for k in ("px", "py", "vx", "vy"):
        assert np.array_equal(st[k].view(np.uint32), q[k].view(np.uint32)), k
    fx, fy, _ = e.forces(0.6)
    rc, ofx, ofy = ob.bh_forces(q, 0.6, nthreads=8)
    assert rc == 0 and np.array_equal(fx.view(np.uint32), ofx.view(np.uint32)) and np.array_equal(fy.view(np.uint32), ofy.view(np.uint32))
    e.set_bh_fold("exact")
    e.step_barnes_hut(0.6, 0.01, 1)
    assert e.get_stat(NBX_STAT_BH_LAST_TREE) == 0
    f = rx.NBodyEngine(mode="strict")        # default tree choice: the host build
    f.set_particles(p["px"], p["py"], p["vx"], p["vy"], p["m"])
    f.step_barnes_hut(0.6, 0.01, 1)
    assert f.get_stat(NBX_STAT_BH_LAST_TREE) == 0


def test_device_tree_node_pool_overflow_falls_back_to_host_build(rx, ob):
    """Thousands of pairs 2e-4 apart (farther than EPS in x, so nobody merges them: nbody.rs:249) force ~18-level chains,
    more than the 4 nodes per body the device pool holds: the device build reports pool exhaustion and the evaluation takes
    the reference-faithful host build instead. Which build ran is ASSERTED (NBX_STAT_BH_FALLBACKS / NBX_STAT_BH_LAST_TREE),
    and the result is then the host-tree result bit for bit."""
    from rust_exp_amd.engine import NBX_STAT_BH_FALLBACKS, NBX_STAT_BH_LAST_TREE

    rng = np.random.default_rng(8)
    x = rng.uniform(-20, 20, 4000).astype(np.float32)
    y = rng.uniform(-20, 20, 4000).astype(np.float32)
    x2 = np.concatenate([x, x + np.float32(2e-4)])
    y2 = np.concatenate([y, y])
    assert np.all(np.abs(x2[4000:] - x2[:4000]) >= 1.5e-4)
    n = len(x2)
    p = ob.particles(x2, y2, np.zeros(n), np.zeros(n), np.ones(n))
    a = rx.NBodyEngine(); a.set_bh_tree("host"); a.set_particles(p["px"], p["py"], p["vx"], p["vy"], p["m"])
    b = rx.NBodyEngine(); b.set_bh_tree("device"); b.set_bh_fold("exact"); b.set_particles(p["px"], p["py"], p["vx"], p["vy"], p["m"])
    fx, fy, _ = a.forces(0.5)
    assert a.get_stat(NBX_STAT_BH_LAST_TREE) == 0 and a.get_stat(NBX_STAT_BH_FALLBACKS) == 0
    gx, gy, _ = b.forces(0.5)
    assert b.get_stat(NBX_STAT_BH_FALLBACKS) == 1 and b.get_stat(NBX_STAT_BH_LAST_TREE) == 0
    from rust_exp_amd.engine import NBX_STAT_BH_REFUSAL
    assert b.get_stat(NBX_STAT_BH_REFUSAL) == 0x10000      # "the node pool overflowed"
    assert b.bh_host_timing()["nodes"] == a.bh_host_timing()["nodes"] > 4 * n
    assert np.array_equal(gx.view(np.uint32), fx.view(np.uint32)) and np.array_equal(gy.view(np.uint32), fy.view(np.uint32))
    b.step_barnes_hut(0.5, 0.01, 1)
    assert b.get_stat(NBX_STAT_BH_FALLBACKS) == 2
    assert np.isfinite(b.get_particles()["px"]).all()
    # a well-separated system of the same size stays on the device
    q = ob.random_disk(n, 5)
    b.set_particles(q["px"], q["py"], q["vx"], q["vy"], q["m"])
    b.forces(0.5)
    assert b.get_stat(NBX_STAT_BH_LAST_TREE) == 1 and b.get_stat(NBX_STAT_BH_FALLBACKS) == 2


@pytest.mark.parametrize("walk", [1, 2, 0])
@pytest.mark.parametrize("n,theta", [(1, 0.5), (70, 0.5), (5000, 0.3), (100000, 0.85)])
def test_wave_uniform_walk_is_bit_identical_to_the_per_lane_walk(rx, ob, n, theta, walk):
    """NBX_OPT_BH_WAVE: one walk per wave (scalar record loads; lanes inside a subtree kept as a scalar mask -- child-group walk,
    walk = 1 -- or parked on accepted subtrees -- node walk of rounds 1-3, walk = 0) must make every body's decisions and sums
    exactly as the per-lane walk of the same kind does."""
    from rust_exp_amd.engine import NBX_OPT_BH_WALK, NBX_OPT_BH_WAVE

    p = ob.stable_orbits(n, 0.5, 30.0, 61) if n > 1 else ob.random_disk(1, 61)
    res = []
    for wave in (0, 1):
        e = engines(rx, p)
        e.set_bh_tree("device")
        e.set_option(NBX_OPT_BH_WALK, walk)
        e.set_option(NBX_OPT_BH_WAVE, wave)
        fx, fy, _ = e.forces(theta)
        e.step_barnes_hut(theta, 0.01, 1)
        st = e.get_particles()
        res.append((fx, fy, st["px"], st["vx"]))
    for a, b in zip(res[0], res[1]):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.mark.parametrize("walk", [1, 2, 0])
@pytest.mark.parametrize("theta", [0.5, 0.25, 1.0])
def test_wave_uniform_walk_on_a_lattice_takes_the_exact_test_path(rx, ob, theta, walk):
    """Bodies on a power-of-two lattice put many node centres at distances where s/d equals theta exactly or to within the
    1e-5 band: those visits take the reference's own sqrt-and-divide test (node walk: take_node's exact path; child-group walk:
    the exact threshold of bh_threshold.h sits right there). Decisions and sums must still be the per-lane walk's, bit for bit,
    and the forces must agree with the oracle's Barnes-Hut to the fast-mode tolerance."""
    from rust_exp_amd.engine import NBX_OPT_BH_WALK, NBX_OPT_BH_WAVE

    side = 64
    gx, gy = np.meshgrid(np.arange(side, dtype=np.float32), np.arange(side, dtype=np.float32))
    x = (gx.ravel() - 31.5).astype(np.float32)
    y = (gy.ravel() - 31.5).astype(np.float32)
    n = x.size
    rng = np.random.default_rng(5)
    order = rng.permutation(n)
    p = ob.particles(x[order], y[order], np.zeros(n), np.zeros(n), np.ones(n))
    res = []
    for wave in (0, 1):
        e = engines(rx, p, fold="exact")
        e.set_bh_tree("device")
        e.set_option(NBX_OPT_BH_WALK, walk)
        e.set_option(NBX_OPT_BH_WAVE, wave)
        fx, fy, _ = e.forces(theta)
        res.append((fx, fy))
    for a, b in zip(res[0], res[1]):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    # against the oracle: a lattice is all ties, and the device tree's exactly summed centres of mass may land on the other side
    # of one than the reference's f32 running fold -- each flip replaces a node by its children, an error of the size of the
    # theta approximation itself; everything else agrees to fp32 rounding
    rc, ofx, ofy = ob.bh_forces(p, theta)
    assert rc == 0
    scale = max(np.abs(ofx).max(), np.abs(ofy).max())
    err = np.maximum(np.abs(res[1][0] - ofx), np.abs(res[1][1] - ofy)) / scale
    assert np.median(err) <= 2e-5 and err.max() <= 0.05, (np.median(err), err.max())


@pytest.mark.parametrize("case", ["identical", "collinear_x", "collinear_y", "two_clusters_far_apart", "three"])
def test_device_tree_degenerate_inputs(rx, ob, case):
    """Zero-extent boxes, 31-level chains and bodies that share a key: the scan-based build must stay finite, keep the
    pre-order invariants (skip pointers nest, the root spans the array, root mass = total mass) and agree with
    all-pairs where the approximation is exact (theta -> tiny)."""
    rng = np.random.default_rng(11)
    if case == "identical":
        n = 300
        x = np.full(n, 1.25, np.float32); y = np.full(n, -3.5, np.float32)
    elif case == "collinear_x":
        n = 2000
        x = rng.uniform(-30, 30, n).astype(np.float32); y = np.full(n, 2.0, np.float32)
    elif case == "collinear_y":
        n = 2000
        x = np.full(n, -7.0, np.float32); y = rng.uniform(-30, 30, n).astype(np.float32)
    elif case == "two_clusters_far_apart":
        n = 4000
        x = np.concatenate([rng.normal(-40, 1e-3, n // 2), rng.normal(40, 1e-3, n // 2)]).astype(np.float32)
        y = np.concatenate([rng.normal(0, 1e-3, n // 2), rng.normal(0, 1e-3, n // 2)]).astype(np.float32)
    else:
        n = 3
        x = np.array([0.0, 1.0, 1.0], np.float32); y = np.array([0.0, 0.0, 1.0], np.float32)
    m = rng.uniform(0.5, 2.0, n).astype(np.float32)
    p = ob.particles(x, y, np.zeros(n), np.zeros(n), m)
    e = engines(rx, p, fold="exact")
    e.set_bh_tree("device")
    dev = e.bh_flat_dump("device")
    k = len(dev)
    assert dev["skip"][0] == k
    assert np.all(dev["skip"] > np.arange(k)) and np.all(dev["skip"] <= k)
    inner = dev["interior"] == 1
    # children start right after an interior node and every subtree ends inside its parent's span
    assert np.all(dev["skip"][np.flatnonzero(inner) + 1] <= dev["skip"][inner])
    assert abs(float(dev["m"][0]) - float(m.astype(np.float64).sum())) <= 1e-5 * float(m.sum())
    assert np.array_equal(dev["q"][inner], dev["s"][inner] * dev["s"][inner]) and np.all(dev["q"][~inner] == -1.0)
    bx, by, _ = e.forces(1e-3)
    assert np.isfinite(bx).all() and np.isfinite(by).all()
    # against the host (= reference) tree through the same traversal. Two reference quirks show up here and must be
    # shared, not "fixed": coincident bodies sit in one leaf whose running-fold centre of mass (nbody.rs:315-318) lands
    # an ulp off their position, so each feels a spurious m*M*ulp/EPS pull; and the opening test uses the x-extent
    # only (nbody.rs:341), so a vertical line of bodies (s = 0 everywhere) accepts the root for everyone.
    h = engines(rx, p)
    h.set_bh_tree("host")
    hx, hy, _ = h.forces(1e-3)
    scale = max(np.abs(hx).max(), np.abs(hy).max(), 1e-6)
    # 1e-3: the host's centres of mass carry the reference's running-fold drift (up to 6e-4 relative), the device's are
    # exact means -- e.g. the vertical line keeps x = -7 exactly on the device (zero x-force) but not on the host
    assert np.abs(bx - hx).max() <= 1e-3 * scale and np.abs(by - hy).max() <= 1e-3 * scale
    if case == "identical":
        host = h.bh_flat_dump(False)
        assert np.array_equal(host[-1:]["px"].view(np.uint32), dev[-1:]["px"].view(np.uint32))
        assert np.array_equal(host[-1:]["m"].view(np.uint32), dev[-1:]["m"].view(np.uint32))
    if case in ("collinear_x", "two_clusters_far_apart", "three"):
        fx, fy, _ = e.forces(0.0)     # theta -> 0 opens everything: all-pairs up to summation order
        scale = max(np.abs(fx).max(), np.abs(fy).max(), 1e-6)
        assert np.abs(bx - fx).max() <= 1e-4 * scale and np.abs(by - fy).max() <= 1e-4 * scale
    e.step_barnes_hut(0.5, 0.01, 1)
    st = e.get_particles()
    assert np.isfinite(st["px"]).all() and np.isfinite(st["vx"]).all()


@pytest.mark.parametrize("fold", ["reference", "exact"])
def test_steps_enqueued_without_waiting_for_the_build_verdict(rx, ob, fold):
    """NBX_OPT_BH_ASYNC (round 3, default on): a Barnes-Hut step on the device tree is enqueued without a host wait in the
    middle -- walk and kick-drift check the build's verdict on the device -- and the host reads the verdict at the next call
    that needs the state.  Same state as the waiting form (NBX_OPT_BH_ASYNC = 0), bit for bit, over several back-to-back steps;
    and when the build must refuse (EPS triples under the reference fold, an exhausted node pool under either), the gated
    kernels leave the state alone and the step is redone on the host tree: the host-tree result, bit for bit."""
    from rust_exp_amd.engine import NBX_OPT_BH_ASYNC, NBX_STAT_BH_FALLBACKS, NBX_STAT_BH_LAST_TREE

    p = ob.stable_orbits(12000, 0.5, 30.0, 51)
    outs = []
    for async_ in (1, 0):
        e = engines(rx, p, fold=fold)
        e.set_option(NBX_OPT_BH_ASYNC, async_)
        for _ in range(6):
            e.step_barnes_hut(0.7, 0.01, 1)
        assert e.get_stat(NBX_STAT_BH_LAST_TREE) == 1 and e.get_stat(NBX_STAT_BH_FALLBACKS) == 0
        outs.append(e.get_particles())
    for k in ("px", "py", "vx", "vy"):
        assert np.array_equal(outs[0][k].view(np.uint32), outs[1][k].view(np.uint32)), k
    assert np.abs(outs[0]["px"] - p["px"]).max() > 0.1

    # a system the device build refuses: thousands of pairs 2e-4 apart in x (no EPS merge) -> ~18-level chains, more nodes than
    # the pool holds (both classes); plus EPS triples (refused by the reference fold alone)
    rng = np.random.default_rng(8)
    x = rng.uniform(-20, 20, 4000).astype(np.float32); y = rng.uniform(-20, 20, 4000).astype(np.float32)
    x2 = np.concatenate([x, x + np.float32(2e-4), x[:3] + np.float32(3e-5), x[:3] - np.float32(2e-5)])
    y2 = np.concatenate([y, y, y[:3] + np.float32(1e-5), y[:3] + np.float32(4e-5)])
    n = len(x2)
    q = ob.particles(x2, y2, rng.normal(0, 1, n), rng.normal(0, 1, n), np.ones(n))
    h = engines(rx, q); h.set_bh_tree("host")
    d = engines(rx, q, fold=fold); d.set_bh_tree("device")
    h.step_barnes_hut(0.5, 0.01, 1)
    # three steps enqueued back to back: the first is refused (and poisons the second, which is enqueued again after the redo);
    # the bodies have moved apart by then, so the later builds succeed
    for _ in range(3):
        d.step_barnes_hut(0.5, 0.01, 1)
    assert d.get_stat(NBX_STAT_BH_FALLBACKS) >= 1
    e1 = engines(rx, q, fold=fold); e1.set_bh_tree("device"); e1.set_option(NBX_OPT_BH_ASYNC, 0)
    e1.step_barnes_hut(0.5, 0.01, 1)
    a, b = h.get_particles(), e1.get_particles()
    assert e1.get_stat(NBX_STAT_BH_FALLBACKS) == 1 and e1.get_stat(NBX_STAT_BH_LAST_TREE) == 0
    for k in ("px", "py", "vx", "vy"):                       # the refused step == the host-tree step, bit for bit
        assert np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32)), k
    for _ in range(2):
        e1.step_barnes_hut(0.5, 0.01, 1)
    a, b = e1.get_particles(), d.get_particles()             # waiting form == pipelined form over the whole sequence
    assert e1.get_stat(NBX_STAT_BH_FALLBACKS) == d.get_stat(NBX_STAT_BH_FALLBACKS)
    for k in ("px", "py", "vx", "vy"):
        assert np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32)), k
    fb = d.get_stat(NBX_STAT_BH_FALLBACKS)
    # new state, new verdict: a well-separated system right behind it stays on the device, nothing is left pending
    d.set_particles(p["px"], p["py"], p["vx"], p["vy"], p["m"])
    d.step_barnes_hut(0.5, 0.01, 1)
    d.set_particles(p["px"], p["py"], p["vx"], p["vy"], p["m"])      # replaces the state while that step's verdict is still unread
    d.step_barnes_hut(0.5, 0.01, 1)
    assert d.get_stat(NBX_STAT_BH_LAST_TREE) == 1 and d.get_stat(NBX_STAT_BH_FALLBACKS) == fb
    ref = engines(rx, p, fold=fold); ref.set_option(NBX_OPT_BH_ASYNC, 0); ref.step_barnes_hut(0.5, 0.01, 1)
    a, b = ref.get_particles(), d.get_particles()
    for k in ("px", "py", "vx", "vy"):
        assert np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32)), k


def test_reference_fold_merges_between_non_neighbouring_entities(rx, ob):
    """Found by tests/fuzz_fast.py (seed 5214): 1 000 bodies within 0.03 of the origin, a fifth of them copies 1e-5 away from
    another body.  Two bodies within EPS need not be neighbours in key order -- a third body of their common cell can sit
    between them -- and the reference still merges them when that third body arrived later; a neighbours-only merge misses it
    (10 nodes too many: the exact-sum class keeps that looser contract).  The reference-fold class must reproduce it (k_blobs
    replays the cluster, k_place moves the merged body next to its entity) or hand the step to the host build."""
    from rust_exp_amd.engine import NBX_STAT_BH_FALLBACKS, NBX_STAT_BH_LAST_TREE

    rng = np.random.default_rng(5214)
    n = int(rng.choice([2, 3, 17, 255, 256, 257, 1000, 4097, 9000, 20000, 70000, 150000]))
    scale = float(rng.choice([1e-2, 1.0, 30.0, 3e3]))
    assert (n, scale) == (1000, 0.01)
    x = (rng.normal(0, 1, n) * scale).astype(np.float32)
    y = (rng.normal(0, 1, n) * scale).astype(np.float32)
    assert rng.random() < 0.5
    k = n // 5
    x[:k] = x[k:2 * k] + (rng.normal(0, 1e-3, k) * scale).astype(np.float32)
    y[:k] = y[k:2 * k] + (rng.normal(0, 1e-3, k) * scale).astype(np.float32)
    p = ob.particles(x, y, np.zeros(n), np.zeros(n), np.full(n, 0.37))
    h = engines(rx, p); h.set_bh_tree("host")
    d = engines(rx, p, fold="reference"); d.set_bh_tree("device")   # device tree, reference fold (below 1 024 bodies that class defaults to the host build)
    from rust_exp_amd.engine import NBX_STAT_BH_CLASS_SWITCHES

    fx, fy, _ = h.forces(0.85)
    gx, gy, _ = d.forces(0.85)
    kept = d.get_stat(NBX_STAT_BH_CLASS_SWITCHES) == 0 and d.get_stat(NBX_STAT_BH_FALLBACKS) == 0
    if kept:                                            # since the replay of whole clusters (k_blobs): the device tree itself
        assert np.array_equal(fx.view(np.uint32), gx.view(np.uint32)) and np.array_equal(fy.view(np.uint32), gy.view(np.uint32))
        _bit_equal_trees(d.bh_flat_dump(False), d.bh_flat_dump("device"))
    # the same build asked for by the BIT-EXACT mode (its only device class): kept -> that tree, refused -> the host build; a step
    # through either gives the host-tree step bit for bit
    s_, hs = rx.NBodyEngine(mode="strict"), rx.NBodyEngine(mode="strict")
    s_.set_bh_tree("device"); s_.set_bh_fold("reference"); hs.set_bh_tree("host")
    for e in (s_, hs):
        e.set_particles(p["px"], p["py"], p["vx"], p["vy"], p["m"])
        e.step_barnes_hut(0.85, 0.01, 1)
    assert (s_.get_stat(NBX_STAT_BH_LAST_TREE) == 1) == kept and s_.get_stat(NBX_STAT_BH_FALLBACKS) == (0 if kept else 1)
    a, b = hs.get_particles(), s_.get_particles()
    for kk in ("px", "py", "vx", "vy"):
        assert np.array_equal(a[kk].view(np.uint32), b[kk].view(np.uint32)), kk


def test_refusals_in_a_row_back_off_to_the_host_build(rx, ob):
    """A system the device build refuses tends to stay that way for many steps: after the second refused build in a row the next 2, 4, 8 ..
    steps go straight to the host build, then one step tries the device again (engine_internal.h note_refusal).  Every step is
    the host-tree step, bit for bit; the number of device builds attempted is read from the profile; a new state resets it."""
    from rust_exp_amd.engine import NBX_K_TREE_BUILD, NBX_STAT_BH_FALLBACKS, NBX_STAT_BH_LAST_TREE

    rng = np.random.default_rng(77)
    n0 = 4000
    x = rng.uniform(-20, 20, n0).astype(np.float32); y = rng.uniform(-20, 20, n0).astype(np.float32)
    x2 = np.concatenate([x, x + np.float32(2e-4)])      # thousands of pairs 2e-4 apart, at rest: ~18-level chains, the node
    y2 = np.concatenate([y, y])                         # pool overflows at every step
    n = len(x2)
    q = ob.particles(x2, y2, np.zeros(n), np.zeros(n), np.full(n, 1e-6))
    from rust_exp_amd.engine import NBX_OPT_BH_ASYNC
    h = engines(rx, q); h.set_bh_tree("host")
    steps = 12
    for _ in range(steps):
        h.step_barnes_hut(0.85, 1e-4, 1)
    a = h.get_particles()
    for async_ in (1, 0):
        d = engines(rx, q)
        d.set_option(NBX_OPT_BH_ASYNC, async_)
        d.profile(True)
        for _ in range(steps):
            d.step_barnes_hut(0.85, 1e-4, 1)
        b = d.get_particles()
        for k in ("px", "py", "vx", "vy"):
            assert np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32)), k
        assert d.get_stat(NBX_STAT_BH_FALLBACKS) == steps and d.get_stat(NBX_STAT_BH_LAST_TREE) == 0
        # attempts: steps 1, 2 (refused twice -> 2 host steps), 5 (-> 4 host steps), 10 (-> 8 host steps); the pipelined form had
        # step 2's build in flight when step 1's verdict arrived (poisoned, enqueued again): one build more
        assert d.profile_read(NBX_K_TREE_BUILD)[1] == (5 if async_ else 4)
    # a new state starts afresh: the device tree serves it at once
    p = ob.stable_orbits(6000, 0.5, 30.0, 3)
    d.set_particles(p["px"], p["py"], p["vx"], p["vy"], p["m"])
    d.step_barnes_hut(0.85, 0.01, 1)
    assert d.get_stat(NBX_STAT_BH_LAST_TREE) == 1 and d.get_stat(NBX_STAT_BH_FALLBACKS) == steps


def test_strict_mode_on_the_device_tree_keeps_the_references_depth_panic(rx, ob):
    """The reference panics when its depth COUNTER passes 50 (nbody.rs:230-232) -- and the counter grows by two per level while a
    leaf is split down, so two bodies 1.5e-4 apart (not "too close") in a box 4e4 wide, 28 levels to separate, trip it although no
    node is deeper than 29.  The device build has no such counter: in the bit-exact mode it therefore leaves every tree with a
    leaf below level 25 to the host build, which counts like the reference (NBX_STAT_BH_REFUSAL 0x200), and the caller gets the
    reference's panic as NBX_ERR_TREE_DEPTH; found by tests/fuzz_strict.py seed 20206.  The fast mode documents that it has no
    depth panic and builds the tree."""
    from rust_exp_amd.engine import NBX_STAT_BH_LAST_TREE, NBX_STAT_BH_REFUSAL

    rng = np.random.default_rng(5)
    n0 = 1500
    x = np.concatenate([[1.0, 1.0 + 1.5e-4], rng.uniform(-2e4, 2e4, n0)]).astype(np.float32)
    y = np.concatenate([[1.0, 1.0], rng.uniform(-2e4, 2e4, n0)]).astype(np.float32)
    n = len(x)
    p = ob.particles(x, y, np.zeros(n), np.zeros(n), np.ones(n))
    rc, _, _ = ob.bh_forces(p, 0.5, nthreads=1)
    assert rc != 0                                      # the oracle (= the reference) panics on this system
    e = rx.NBodyEngine(mode="strict")
    e.set_bh_tree("device")
    e.set_particles(p["px"], p["py"], p["vx"], p["vy"], p["m"])
    with pytest.raises(rx.NBodyError) as err:
        e.forces(0.5)
    assert err.value.code == -4 and e.get_stat(NBX_STAT_BH_REFUSAL) == 0x200
    f = engines(rx, p); f.set_bh_tree("device")
    fx, fy, _ = f.forces(0.5)
    assert f.get_stat(NBX_STAT_BH_LAST_TREE) == 1 and np.isfinite(fx).all() and np.isfinite(fy).all()


def _disk_with_a_long_eps_chain(ob, n_disk, chain, seed):
    """A random disk + `chain` bodies on a line, 1.9 EPS apart: every link is shorter than 2 EPS (one connected component for the
    reference-fold replay: more than its 96 bodies -> that class must refuse, NBX_STAT_BH_REFUSAL bit 4) and longer than EPS
    (nothing merges, nbody.rs:249: the exact-sum class builds the reference's node set without leaving anybody behind -- it replays a
    chain this long in pieces, and blobs at the cuts would count as left behind: at 1.5 EPS the disk's tide had pushed 25 of 130
    bodies within EPS of a neighbour after four steps, 26 of them at cuts, 16 are tolerated at this size)."""
    rng = np.random.default_rng(seed)
    d = ob.random_disk(n_disk, seed)
    cx = np.float32(3.0) + np.arange(chain, dtype=np.float32) * np.float32(1.9e-4)
    cy = np.full(chain, 2.0, np.float32)
    order = rng.permutation(n_disk + chain)                 # the chain's bodies arrive anywhere in the sequence
    cat = lambda a, b: np.concatenate([np.asarray(a, np.float32), np.asarray(b, np.float32)])[order]   # noqa: E731
    zero = np.zeros(chain, np.float32)
    # (masses of the chain small: it stays a chain for the steps below)
    return ob.particles(cat(d["px"], cx), cat(d["py"], cy), cat(d["vx"], zero), cat(d["vy"], zero), cat(d["m"], np.full(chain, 1e-6)))


@pytest.mark.parametrize("async_", [1, 0])
def test_a_refused_reference_fold_build_is_served_by_the_exact_sum_device_build(rx, ob, async_):
    """Round 6: in the FAST mode a reference-fold device build that refuses (nbody.rs:249-260's merge, :303-320's
    fold: what the replay cannot reproduce node for node) is redone on the exact-sum DEVICE build -- 8 x faster than the host build
    at 65 536 bodies and inside the fast mode's stated tolerance -- and so is the back-off run behind refusals in a row; the host
    build serves only the bit-exact mode and refusals of the exact-sum class itself.  The step is then the exact-sum engine's
    step bit for bit, its forces within the exact-sum class's bounds of the oracle's fp64 arbiter (orc_bh_forces_exact)."""
    from rust_exp_amd.engine import (NBX_OPT_BH_ASYNC, NBX_STAT_BH_CLASS_SWITCHES, NBX_STAT_BH_FALLBACKS, NBX_STAT_BH_LAST_TREE,
                                     NBX_STAT_BH_REFUSAL)

    p = _disk_with_a_long_eps_chain(ob, 20000, 110, 77)
    theta = 0.5
    r = engines(rx, p, fold="reference"); r.set_option(NBX_OPT_BH_ASYNC, async_)
    x = engines(rx, p, fold="exact"); x.set_option(NBX_OPT_BH_ASYNC, async_)
    # forces: refused -> exact-sum device tree, never the host
    fx, fy, _ = r.forces(theta)
    assert r.get_stat(NBX_STAT_BH_CLASS_SWITCHES) == 1 and r.get_stat(NBX_STAT_BH_FALLBACKS) == 0
    assert r.get_stat(NBX_STAT_BH_LAST_TREE) == 1 and r.get_stat(NBX_STAT_BH_REFUSAL) & 4
    gx, gy, _ = x.forces(theta)
    assert x.get_stat(NBX_STAT_BH_CLASS_SWITCHES) == 0 and x.get_stat(NBX_STAT_BH_FALLBACKS) == 0
    assert np.array_equal(fx.view(np.uint32), gx.view(np.uint32)) and np.array_equal(fy.view(np.uint32), gy.view(np.uint32))
    rc, ex, ey = ob.bh_forces_exact(p, theta, nthreads=16)
    assert rc == 0
    scale = max(np.abs(ex).max(), np.abs(ey).max())
    err = np.maximum(np.abs(fx - ex), np.abs(fy - ey)) / scale
    assert np.percentile(err, 99.9) <= 2e-5 and err.max() <= 2e-3, (np.percentile(err, 99.9), err.max())
    # steps: six in a row, every one refused by the reference-fold class or inside its back-off run (refusals 1, 2 -> run of 2 ->
    # refusal 3 -> run of 4 ...): all six served by the exact-sum device build, none by the host
    for _ in range(6):
        r.step_barnes_hut(theta, 0.01, 1)
        x.step_barnes_hut(theta, 0.01, 1)
    a, b = r.get_particles(), x.get_particles()
    assert r.get_stat(NBX_STAT_BH_CLASS_SWITCHES) == 1 + 6 and r.get_stat(NBX_STAT_BH_FALLBACKS) == 0
    assert r.get_stat(NBX_STAT_BH_LAST_TREE) == 1
    for k in ("px", "py", "vx", "vy"):
        assert np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32)), k
    # the bit-exact mode on the same system: only the reference fold serves it, so the refusal goes to the HOST build -- the
    # host-tree step bit for bit
    s = rx.NBodyEngine(mode="strict"); s.set_bh_tree("device"); s.set_bh_fold("reference"); s.set_option(NBX_OPT_BH_ASYNC, async_)
    h = rx.NBodyEngine(mode="strict"); h.set_bh_tree("host")
    for e in (s, h):
        e.set_particles(p["px"], p["py"], p["vx"], p["vy"], p["m"])
        e.step_barnes_hut(theta, 0.01, 1)
    assert s.get_stat(NBX_STAT_BH_FALLBACKS) == 1 and s.get_stat(NBX_STAT_BH_CLASS_SWITCHES) == 0 and s.get_stat(NBX_STAT_BH_LAST_TREE) == 0
    a, b = s.get_particles(), h.get_particles()
    for k in ("px", "py", "vx", "vy"):
        assert np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32)), k


def test_the_fast_modes_default_tree_class_is_chosen_by_cost(rx, ob):
    """NBX_OPT_BH_FOLD = -1 (round 6): the fast mode builds exact sums on the device from 512 bodies on at EVERY size (the
    reference fold's build is 2.6 x .. 12.7 x the exact-sum build's between 2 000 and 65 536 bodies, profiles/r06_bh_sizes.jsonl:
    never within the 1.5 x of the cost rule) -- the default engine's forces are the exact-sum engine's bit for bit, not the
    reference-fold engine's; the bit-exact mode keeps the host build."""
    from rust_exp_amd.engine import NBX_STAT_BH_LAST_TREE

    for n in (600, 10000, 65536):
        p = ob.stable_orbits(n, 0.5, 30.0, 5)
        d, x, r = engines(rx, p), engines(rx, p, fold="exact"), engines(rx, p, fold="reference")
        r.set_bh_tree("device")
        fd, fx, fr = d.forces(0.85), x.forces(0.85), r.forces(0.85)
        assert d.get_stat(NBX_STAT_BH_LAST_TREE) == 1 and x.get_stat(NBX_STAT_BH_LAST_TREE) == 1
        assert np.array_equal(fd[0].view(np.uint32), fx[0].view(np.uint32)) and np.array_equal(fd[1].view(np.uint32), fx[1].view(np.uint32))
        if n >= 10000:
            assert not np.array_equal(fd[0].view(np.uint32), fr[0].view(np.uint32))   # (the classes do differ: the sun's chain of 10^4 terms)
    s = rx.NBodyEngine(mode="strict")
    s.set_particles(p["px"], p["py"], p["vx"], p["vy"], p["m"])
    s.forces(0.85)
    assert s.get_stat(NBX_STAT_BH_LAST_TREE) == 0
