#!/usr/bin/env python3
"""CPU model of the exact-sum device build's chain replay (bh_build.hip section 3b: k_chain_links / k_chain_heads / k_chain), the way the
rule was found and its parameters fixed in round 6 -- numpy for the keys and the links, plain Python for the replay of a segment -- and
compared with the ORACLE's tree (test infrastructure like tests/f32_exact.py: tests/test_chain_model.py holds the rule to the oracle on
every CPU run; pytest does not collect this file):
    python tests/chain_model.py [seed]          20 000 bodies + 600 chains of 2-6 bodies within EPS, random arrival order
    from chain_model import compare; compare(particles, W=1 << 30, K=3, LINK=np.float32(2e-4))
`compare` returns how many leaves / nodes the model's tree and the oracle's differ by, and the segment statistics.  Parameters: K = how many
sorted places ahead a body links boundaries, LINK = the link distance (2 EPS shipped), CUTW / MCUT = where chains longer than CUTW + 2 MCUT are
cut (32 / 14 shipped), LOOSECUT = also cut chains longer than this at boundaries no pair within EPS spans (0 = off: shipped).
Results that fixed them: docs/rounds/r06.md section 2."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import binding as ob  # noqa: E402
f32 = np.float32
EPS = f32(1e-4)
LEVELS = 31

def path_keys(x, y, box):
    x1 = np.full(x.shape, box[0], f32); y1 = np.full(x.shape, box[1], f32)
    x2 = np.full(x.shape, box[2], f32); y2 = np.full(x.shape, box[3], f32)
    key = np.zeros(x.shape, np.uint64)
    for l in range(LEVELS):
        cx = ((x1 + x2) * f32(0.5)).astype(f32); cy = ((y1 + y2) * f32(0.5)).astype(f32)
        low = y < cy; left = x < cx
        q = np.where(low, 2, 0) + np.where(left, 0, 1)
        y2 = np.where(low, cy, y2); y1 = np.where(low, y1, cy)
        x2 = np.where(left, cx, x2); x1 = np.where(left, x1, cx)
        key = (key << np.uint64(2)) | q.astype(np.uint64)
    return key

def path_key1(x, y, box):
    return int(path_keys(np.array([x], f32), np.array([y], f32), box)[0])

def common(a, b):
    d = int(a) ^ int(b)
    if d == 0: return LEVELS
    return (62 - d.bit_length()) // 2

def fold(c, q):
    cx, cy, cm = c; qx, qy, qm = q
    if cm == 0: return (qx, qy, qm)
    inv = f32(1.0) / f32(cm + qm)
    return (f32(f32(f32(cx * cm) + f32(qx * qm)) * inv), f32(f32(f32(cy * cm) + f32(qy * qm)) * inv), f32(cm + qm))

def chain_model(px, py, m, W=32, RUNLINK=8, LINK=EPS, SCAN=256, stats=None, ret_part=False, K=1, box=None, CUTW=32, MCUT=14):
    n = len(px)
    if box is None: box = (px.min(), py.min(), px.max(), py.max())
    keys = path_keys(px, py, box)
    order = np.argsort(keys, kind="stable")
    ks = keys[order]; xs = px[order]; ys = py[order]; ms = m[order]; idx = order.astype(np.int64)
    start = np.ones(n, bool); start[1:] = ks[1:] != ks[:-1]
    es = np.flatnonzero(start); ee = np.append(es[1:], n); ne = len(es)
    runlen = np.repeat(ee - es, ee - es)
    longrun = runlen > RUNLINK
    # body-level links: boundary b (between b-1 and b), b = 1..n-1, covered if some j < b <= t <= j+K is within LINK
    cov = np.zeros(n + 1, bool)
    for d in range(1, K + 1):
        if n > d:
            cl = (np.abs(xs[:-d] - xs[d:]) < LINK) & (np.abs(ys[:-d] - ys[d:]) < LINK)   # pair (j, j+d)
            for t in range(1, d + 1):                 # boundaries j+1 .. j+d
                cov[t:n - d + t] |= cl
    lk = cov.copy(); lk[0] = False; lk[n] = False
    lk[1:n] &= ~longrun[:-1] & ~longrun[1:]
    # long chains: cut at multiples of CUTW whose MCUT boundaries on either side are all linked
    cut = np.zeros(n + 1, bool)
    for bnd in range(CUTW, n, CUTW):
        lo, hi = bnd - MCUT, bnd + MCUT
        if lo >= 1 and hi <= n - 1 and lk[lo:hi + 1].all(): cut[bnd] = True
    L = lk & ~cut
    link = None
    out_key = ks.copy(); out_pos = np.arange(n)          # out_pos[j] = source sorted position written at output slot j
    nseg = 0; nsim = 0; left_behind = 0; seglens = []
    blobs_multi = 0
    heads_pos = np.flatnonzero(~L[:-1] & L[1:])          # body p is a head: boundary p not linked, boundary p+1 linked
    for a in heads_pos:
        z = a + 1
        while L[z]: z += 1
        nseg += 1
        seglens.append(z - a)
        # --- simulate
        members = list(range(a, z))
        arr = sorted(members, key=lambda j: idx[j])
        heads = []                                   # dicts: c=(cx,cy,cm), rep, members
        for j in arr:
            kB = int(ks[j]); iB = idx[j]
            best = -1; cnt = 0; X = None
            for h in heads:
                c = common(h["rep"], kB)
                if c > best: best, cnt, X = c, 1, h
                elif c == best: cnt += 1
            merged = False
            if X is not None and cnt == 1:
                cx, cy, cm = X["c"]
                if abs(f32(cx - xs[j])) < EPS and abs(f32(cy - ys[j])) < EPS:
                    # rival scan outside
                    rival = False
                    t = a - 1; s = 0
                    while t >= 0 and s < SCAN:
                        if common(ks[t], kB) < best: break
                        if idx[t] < iB: rival = True; break
                        t -= 1; s += 1
                    t = z; s = 0
                    while not rival and t < n and s < SCAN:
                        if common(ks[t], kB) < best: break
                        if idx[t] < iB: rival = True; break
                        t += 1; s += 1
                    if not rival:
                        X["c"] = fold(X["c"], (xs[j], ys[j], ms[j]))
                        X["rep"] = path_key1(X["c"][0], X["c"][1], box)
                        X["members"].append(j)
                        merged = True
            if not merged:
                heads.append({"c": (xs[j], ys[j], ms[j]), "rep": kB, "members": [j], "first": kB})
        kprev = int(ks[a - 1]) if a > 0 else -1
        knext = int(ks[z]) if z < n else 1 << 63
        for h in heads:
            if len(h["members"]) > 1:
                blobs_multi += 1
                if not (kprev < h["rep"] < knext):
                    left_behind += 1
                    bj = max(h["members"], key=lambda j: (common(h["rep"], ks[j]), -j))
                    h["rep"] = int(ks[bj])
        heads.sort(key=lambda h: h["rep"])
        o = a
        for h in heads:
            for j in h["members"]:               # already in arrival order
                out_key[o] = h["rep"]; out_pos[o] = j; o += 1
        assert o == z
    if stats is not None:
        sl = np.array(seglens) if seglens else np.zeros(1, int)
        stats.update(seg_max=int(sl.max()), seg_p99=float(np.percentile(sl, 99)), seg_bodies=int(sl.sum()), seg_gt32=int((sl > 32).sum()), seg_gt64=int((sl > 64).sum()), segments=nseg, blobs_multi=blobs_multi, left_behind=left_behind, entities=ne)
    if ret_part:
        st = np.ones(n, bool); st[1:] = out_key[1:] != out_key[:-1]
        s_ = np.flatnonzero(st); e_ = np.append(s_[1:], n)
        oi = idx[out_pos]
        return [list(oi[a:z]) for a, z in zip(s_, e_)]
    # leaves: runs of equal out_key, folded in idx order
    oidx = idx[out_pos]; ox = xs[out_pos]; oy = ys[out_pos]; om = ms[out_pos]
    st = np.ones(n, bool); st[1:] = out_key[1:] != out_key[:-1]
    s_ = np.flatnonzero(st); e_ = np.append(s_[1:], n)
    leaves = []
    single = (e_ - s_) == 1
    for a, z in zip(s_[~single], e_[~single]):
        jj = sorted(range(a, z), key=lambda j: oidx[j])
        c = (f32(0), f32(0), f32(0))
        for j in jj: c = fold(c, (ox[j], oy[j], om[j]))
        leaves.append(c)
    lv = np.zeros((int(single.sum()) + len(leaves), 3), f32)
    ss = s_[single]
    lv[: len(ss), 0] = ox[ss]; lv[: len(ss), 1] = oy[ss]; lv[: len(ss), 2] = om[ss]
    if leaves: lv[len(ss):] = np.array(leaves, f32)
    # node count from the entity keys
    uk = out_key[s_]
    def cd(a, b):
        d = a ^ b
        r = np.full(d.shape, LEVELS, np.int64)
        nz = d != 0
        bl = np.zeros(d.shape, np.int64)
        dd = d[nz].astype(np.uint64)
        bl[nz] = np.floor(np.log2(dd.astype(np.float64))).astype(np.int64) + 1   # ok for 62-bit? check below
        # exact bit length
        v = dd.copy(); b = np.zeros(len(dd), np.int64)
        for sh in (32, 16, 8, 4, 2, 1):
            t = v >> np.uint64(sh); mk = t != 0
            b[mk] += sh; v[mk] = t[mk]
        bl[nz] = b + 1
        r[nz] = (62 - bl[nz]) // 2
        return r
    if len(uk) > 1:
        c = cd(uk[1:], uk[:-1])
        cl = np.concatenate([[-1], c]); cr = np.concatenate([c, [-1]])
    else:
        cl = np.array([-1]); cr = np.array([-1])
    leaf = np.minimum(LEVELS, 1 + np.maximum(cl, cr))
    nodes = int((leaf - cl).sum())
    return lv, nodes

def oracle_leaves(p):
    rc, d = ob.bh_tree_dump(p)
    assert rc == 0
    lf = d[(d[:, 7] == 0) & (d[:, 6] != 0)]
    return lf[:, 4:7].copy(), len(d), d

def compare(p, **kw):
    st = {}
    lv, nodes = chain_model(p["px"], p["py"], p["m"], stats=st, **kw)
    ol, on, d = oracle_leaves(p)
    # flattened tree of the product omits empty leaves; count oracle nodes without empty leaves
    on_ne = int(((d[:, 7] != 0) | (d[:, 6] != 0)).sum())
    a = set(map(bytes, np.ascontiguousarray(lv).view(np.uint8).reshape(len(lv), 12)))
    b = set(map(bytes, np.ascontiguousarray(ol).view(np.uint8).reshape(len(ol), 12)))
    st.update(model_leaves=len(lv), oracle_leaves=len(ol), model_nodes=nodes, oracle_nodes=on_ne, only_model=len(a - b), only_oracle=len(b - a))
    return st

if __name__ == "__main__":
    rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
    n0 = 20000
    x = rng.uniform(-20, 20, n0).astype(f32); y = rng.uniform(-20, 20, n0).astype(f32)
    # chains: pick 600 seeds, grow chains of 2..6 bodies each step <= 0.9 EPS from the previous
    xs = [x]; ys = [y]
    for s in rng.choice(n0, 600, replace=False):
        L = rng.integers(1, 6)
        cx, cy = x[s], y[s]
        for _ in range(L):
            cx = f32(cx + rng.uniform(-9e-5, 9e-5)); cy = f32(cy + rng.uniform(-9e-5, 9e-5))
            xs.append(np.array([cx], f32)); ys.append(np.array([cy], f32))
    x = np.concatenate(xs); y = np.concatenate(ys)
    perm = rng.permutation(len(x)); x, y = x[perm], y[perm]
    n = len(x)
    p = ob.particles(x, y, np.zeros(n), np.zeros(n), rng.uniform(0.5, 2.0, n))
    print("links at 1 EPS, sorted neighbours only:", compare(p))
    print("links at 2 EPS, sorted neighbours only:", compare(p, LINK=f32(2e-4)))
    print("links at 2 EPS, three places ahead (shipped):", compare(p, LINK=f32(2e-4), K=3))
