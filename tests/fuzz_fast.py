#!/usr/bin/env python3
"""Fuzz campaign of the FAST mode, run by hand on a GPU box (pytest does not collect it):
    python tests/fuzz_fast.py [first_seed] [count]
Random sizes / scales / mass ranges / clumps as in fuzz_strict.py. Per case: everything finite; fast all-pairs forces
within 5e-5 max|F| sqrt(N)/64 of the bit-exact kernel's; fast Barnes-Hut (host tree) within 1e-4 max|F| of the bit-exact walk;
device-built tree (EPS merge reproduced: pairs since round 2, chains of any length since round 6) vs host tree through the same
walk: 99.9 % of the bodies within 2e-4 max|F| (4e-4 above 100 000 bodies: the reference's own f32 node folds drift by 2e-4 .. 3e-4 at 150 000 bodies --
the device's sums are exact -- and a handful of opening decisions flip), everyone within 5e-3 max|F| (1e-2 at theta = 0.85: a
flipped decision costs one node's approximation error, which grows with theta; seed 41296 reached 5.2e-3).
Round 3: a third of the cases have ONE common mass plus 0 / 1 / 5 / 30 exceptions (the unit-mass sweep + K2 correction from
16 384 bodies on); from 1 024 to 65 536 bodies a third engine asks for the reference's running fold (the default class of rounds 3-5; on
request since round 6): whenever that build is kept its forces must equal the host tree's BIT FOR BIT, and when it refuses the exact-sum
DEVICE build must have served the evaluation (bit for bit the exact-sum engine's forces), not the host; and four Barnes-Hut steps enqueued back to
back (the two-slot pipeline without a host wait) must leave the state of four waited-for steps, bit for bit.
Late round 3: 40 % of the cases get 5 / 50 / 300 clusters of 2 .. 6 bodies around EPS wide in shuffled arrival order (the device
build replays whole clusters: k_blobs).  Round 6: those systems are served by the exact-sum device build too (chains replayed in
arrival order, bh_build.hip 3b) and judged against the oracle's fp64 arbiter on the reference's tree: the device tree's forces must be
as close to it as the host (= reference) tree's own (see the comment at the check)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rust_exp_amd as rx  # noqa: E402


def make_case(seed):
    """The case of one seed: positions, velocities, masses, theta, number of clusters added."""
    rng = np.random.default_rng(seed)
    n = int(rng.choice([2, 3, 17, 255, 256, 257, 1000, 4097, 9000, 20000, 70000, 150000]))
    scale = float(rng.choice([1e-2, 1.0, 30.0, 3e3]))
    x = (rng.normal(0, 1, n) * scale).astype(np.float32)
    y = (rng.normal(0, 1, n) * scale).astype(np.float32)
    if rng.random() < 0.5 and n > 50:
        k = n // 5
        x[:k] = x[k:2 * k] + (rng.normal(0, 1e-3, k) * scale).astype(np.float32)
        y[:k] = y[k:2 * k] + (rng.normal(0, 1e-3, k) * scale).astype(np.float32)
    clumps = 0
    if rng.random() < 0.4 and n > 50:
        # clusters of 2 .. 6 bodies around EPS wide (absolute: the reference's EPS is 1e-4 whatever the scale), arrival order
        # shuffled: blobs of several members, blobs with unmerged neighbours, merges across cell boundaries (k_blobs / k_place)
        clumps = int(rng.choice([5, 50, 300]))
        spread = float(rng.choice([1e-5, 3e-5, 6e-5, 1.5e-4]))
        xs, ys = [x], [y]
        for c in rng.choice(n, min(clumps, n), replace=False):
            kk = int(rng.integers(1, 6))
            xs.append((x[c] + rng.normal(0, spread, kk)).astype(np.float32))
            ys.append((y[c] + rng.normal(0, spread, kk)).astype(np.float32))
        x, y = np.concatenate(xs), np.concatenate(ys)
        perm = rng.permutation(len(x))
        x, y = x[perm], y[perm]
        n = len(x)
    mk = rng.choice(["unit", "wide", "common"])
    if mk == "common":
        m = np.full(n, rng.choice([0.37, 1.0, 2.5e-3]), np.float32)
        k_exc = int(rng.choice([0, 1, 5, 30]))
        if k_exc and n > 2 * k_exc:
            m[rng.choice(n, k_exc, replace=False)] = (10.0 ** rng.uniform(-2, 3, k_exc)).astype(np.float32)
    else:
        m = {"unit": rng.uniform(0.5, 2.0, n), "wide": 10.0 ** rng.uniform(-3, 3, n)}[mk].astype(np.float32)
    vx = rng.normal(0, 1, n).astype(np.float32); vy = rng.normal(0, 1, n).astype(np.float32)
    theta = float(rng.choice([0.3, 0.5, 0.85]))
    return x, y, vx, vy, m, theta, mk, clumps, scale


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    bad = 0
    kept = [[0, 0], [0, 0]]     # [clumps added?][device tree kept?]
    kept_ref = [0, 0, 0]        # reference-fold class asked for: kept / served by the exact-sum device build / by the host build
    t0 = time.time()
    for seed in range(first, first + count):
        x, y, vx, vy, m, theta, mk, clumps, scale = make_case(seed)
        n = len(x)
        why = []
        try:
            def eng(mode, tree=0):
                e = rx.NBodyEngine(mode=mode)
                e.set_bh_tree("device" if tree else "host")
                e.set_particles(x, y, vx, vy, m)
                return e
            fs, ff = eng("strict"), eng("fast")
            sx, sy, _ = fs.forces(0.0)
            fx, fy, _ = ff.forces(0.0)
            sc = max(np.abs(sx).max(), np.abs(sy).max(), 1e-30)
            if not (np.isfinite(fx).all() and np.isfinite(fy).all()):
                why.append("brute not finite")
            elif max(np.abs(fx - sx).max(), np.abs(fy - sy).max()) > (2.0 if clumps else 1.0) * 5e-5 * sc * max(1.0, np.sqrt(n) / 64):
                # summation-order class, grows with sqrt(N); wide mass ranges sit near it, and injected clusters (pair terms up to
                # 1e4 times the far field, cancelling) past it: seed 70692 reached 1.07 x the plain bound
                # Past the bound (1 of 12 000 cases in round 6: seed 411270, 70 000 bodies at the 3e3 scale, one common mass of 0.37 with 30
                # exceptions up to 918): which of the two is off?  The bit-exact kernel's ascending f32 sum is the reference's arithmetic,
                # not the truth; the oracle's float64 sum arbitrates, as in tests/test_gpu_randomized.py -- the fast kernel must be as close
                # to it as the reference's own arithmetic is (x 2, or the fp32 rounding of a single sum).
                from oracle import binding as ob

                tx, ty = ob.brute_forces_f64(ob.particles(x, y, vx, vy, m))
                e_fast = max(np.abs(fx - tx).max(), np.abs(fy - ty).max()) / sc
                e_strict = max(np.abs(sx - tx).max(), np.abs(sy - ty).max()) / sc
                if e_fast > max(2.0 * e_strict, 2e-6):
                    why.append("brute tol %.2e (vs float64: fast %.2e, bit-exact %.2e)" % (max(np.abs(fx - sx).max(), np.abs(fy - sy).max()) / sc, e_fast, e_strict))
            try:
                bsx, bsy, _ = fs.forces(theta)
                tree_ok = True
            except rx.NBodyError:
                tree_ok = False
            if tree_ok:
                bfx, bfy, _ = ff.forces(theta)
                bsc = max(np.abs(bsx).max(), np.abs(bsy).max(), 1e-30)
                if not np.isfinite(bfx).all():
                    why.append("bh not finite")
                elif max(np.abs(bfx - bsx).max(), np.abs(bfy - bsy).max()) > 1e-4 * bsc:
                    why.append("bh tol %.2e" % (max(np.abs(bfx - bsx).max(), np.abs(bfy - bsy).max()) / bsc))
                fd = eng("fast", 1)
                dx_, dy_, _ = fd.forces(theta)
                err = np.maximum(np.abs(dx_ - bfx), np.abs(dy_ - bfy)) / bsc
                if not np.isfinite(dx_).all():
                    why.append("device tree not finite")
                elif clumps:
                    # Systems with injected clusters (up to half the bodies sit in one).  Rounds 3-5 (pairs-only merge) sent most of them
                    # to the host build and held the rest to the host tree's forces: p99 <= 4e-4, up to 8 * max(16, n/2000) bodies
                    # beyond the max-error bound.  Round 6 replays the chains on the device (bh_build.hip 3b), so the device tree serves
                    # them -- same nodes, same leaves (tests/test_gpu_bh_chains.py, tests/test_chain_model.py) -- and what is left to
                    # compare are the INTERIOR records: exact sums here, the reference's f32 running fold there.  A cluster's near field
                    # is 1e4 times the far field, and at theta = 0.85 the last bits of a centre flip the opening decision of the tiny
                    # nodes around a blob for percents of the bodies: measured on the cases a 12 000-case campaign flagged, the host
                    # tree's forces are 8e-4 .. 3.5e-3 (p99), up to 1.1e-2 (max) off the fp64 arbiter on the SAME tree
                    # (orc_bh_forces_exact), the device tree's 2e-4 .. 1.8e-3, up to 4.3e-3.  So the arbiter is the yardstick: at the
                    # 99th and 99.9th percentile and at the maximum the device tree must be within the bounds every other system is held
                    # to (2e-4 / 4e-4; 5e-3 / 1e-2) OR as close to the arbiter as the reference's own tree is (25 % + the walk's fp32
                    # rounding of slack).  At the 3e3 scale neighbouring floats are 2.4e-4 apart, more than EPS: a cluster is then a
                    # handful of bodies on one or two positions whose folded centre drifts by an ulp and makes the reference split bodies
                    # of ONE position (nbody.rs:315-317), which no order of keys can express (NBX_STAT_BH_CHAIN_APPROX counts them):
                    # there the old allowance of bodies beyond the max-error bound stays.
                    from oracle import binding as ob

                    rc_a, ax_, ay_ = ob.bh_forces_exact(ob.particles(x, y, vx, vy, m), theta, nthreads=os.cpu_count() or 8)
                    if rc_a != 0:
                        why.append("oracle arbiter rc %d" % rc_a)
                    else:
                        e_dev = np.maximum(np.abs(dx_ - ax_), np.abs(dy_ - ay_)) / bsc
                        e_ref = np.maximum(np.abs(bfx - ax_), np.abs(bfy - ay_)) / bsc
                        top = 1e-2 if theta > 0.8 else 5e-3
                        allowed = min(8 * max(16, n // 2000), max(8, n // 20)) if scale >= 1e3 else 0
                        far = e_dev > np.maximum(top, 1.25 * e_ref.max() + 2e-5)
                        if int(far.sum()) > allowed:
                            why.append("device tree (clustered, exact sums): %d bodies beyond the max-error bound (%d allowed), max %.2e (host tree vs arbiter %.2e)"
                                       % (int(far.sum()), allowed, e_dev.max(), e_ref.max()))
                        else:
                            bulk = 4e-4 if n > 100000 else 2e-4
                            for q, bound in (() if scale >= 1e3 else ((99.0, bulk), (99.9, bulk if n >= 2000 else top))):   # (3e3 scale: the count above is the check)
                                if np.percentile(e_dev[~far], q) > max(bound, 1.25 * np.percentile(e_ref, q) + 2e-5):
                                    why.append("device tree (clustered, exact sums) further from the fp64 arbiter than the host tree at p%g: %.2e vs %.2e"
                                               % (q, np.percentile(e_dev[~far], q), np.percentile(e_ref, q)))
                                    break
                elif np.percentile(err, 99.9) > (4e-4 if n > 100000 else 2e-4) or err.max() > (1e-2 if theta > 0.8 else 5e-3):
                    why.append("device tree p99.9 %.2e max %.2e" % (np.percentile(err, 99.9), err.max()))
                from rust_exp_amd.engine import NBX_OPT_BH_ASYNC, NBX_STAT_BH_LAST_TREE
                kept[int(clumps > 0)][int(fd.get_stat(NBX_STAT_BH_LAST_TREE) == 1)] += 1
                if 1024 <= n <= 65536:
                    # the reference-fold class (on request since round 6): kept -> the host tree, bit for bit; refused -> served by the
                    # exact-sum DEVICE build (NBX_STAT_BH_CLASS_SWITCHES), i.e. fd's forces bit for bit -- never by the host build
                    # unless the exact-sum class refuses as well
                    from rust_exp_amd.engine import NBX_STAT_BH_CLASS_SWITCHES, NBX_STAT_BH_FALLBACKS
                    fr = rx.NBodyEngine(mode="fast"); fr.set_bh_tree("device"); fr.set_bh_fold("reference"); fr.set_particles(x, y, vx, vy, m)
                    rx_, ry_, _ = fr.forces(theta)
                    sw, fb = fr.get_stat(NBX_STAT_BH_CLASS_SWITCHES), fr.get_stat(NBX_STAT_BH_FALLBACKS)
                    same = lambda a, b: np.array_equal(a.view(np.uint32), b.view(np.uint32))   # noqa: E731
                    if sw == 0 and fb == 0:
                        if not (same(rx_, bfx) and same(ry_, bfy)):
                            why.append("reference-fold device tree != host tree (%d words)" % int((rx_.view(np.uint32) != bfx.view(np.uint32)).sum()))
                    elif fb == 0 and fd.get_stat(NBX_STAT_BH_LAST_TREE) == 1:
                        if not (same(rx_, dx_) and same(ry_, dy_)):
                            why.append("refused reference-fold build not served by the exact-sum device build's forces")
                    kept_ref[0 if sw == 0 and fb == 0 else (1 if fb == 0 else 2)] += 1
                # (once the bodies have moved a tree may be one the reference panics on -- two bodies an ulp apart at |x| ~ 3000
                #  need cells no f32 midpoint can make: the library then reports the reference's panic, seed 91685 -- not a failure)
                try:
                    if n >= 512:
                        pa, pb = eng("fast", 1), eng("fast", 1)
                        pb.set_option(NBX_OPT_BH_ASYNC, 0)
                        for _ in range(4):
                            pa.step_barnes_hut(theta, 0.01, 1); pb.step_barnes_hut(theta, 0.01, 1)
                        sa, sb = pa.get_particles(), pb.get_particles()
                        if any(not np.array_equal(sa[k].view(np.uint32), sb[k].view(np.uint32)) for k in ("px", "py", "vx", "vy")):
                            why.append("pipelined steps != waited-for steps")
                    for e in (ff, fd):
                        e.step_barnes_hut(theta, 0.01, 1); e.step_brute_force(0.01)
                        st = e.get_particles()
                        if not (np.isfinite(st["px"]).all() and np.isfinite(st["vx"]).all()):
                            why.append("state not finite")
                except rx.NBodyError as ex:
                    if ex.code not in (-4, -5):      # NBX_ERR_TREE_DEPTH, NBX_ERR_TREE: the reference's own panics
                        raise
        except Exception as ex:   # noqa: BLE001
            why.append("exception " + repr(ex))
        if why:
            bad += 1
            print("FAIL seed", seed, "n", n, "scale", scale, "masses", mk, "theta", theta, why)
    print("fuzz fast: %d cases, %d failures, %.1f s; device tree kept / handed over: %d / %d without added clusters, %d / %d with; "
          "reference-fold class asked for (1 024 .. 65 536 bodies): %d kept, %d served by the exact-sum device build, %d by the host build"
          % (count, bad, time.time() - t0, kept[0][1], kept[0][0], kept[1][1], kept[1][0], kept_ref[0], kept_ref[1], kept_ref[2]))


if __name__ == "__main__":
    main()
