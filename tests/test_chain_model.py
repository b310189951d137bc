"""CPU: the RULE the exact-sum device build's chain replay implements (bh_build.hip section 3b), restated in numpy / plain Python
(tests/chain_model.py) and held against the oracle's tree (oracle/nbody_oracle.c: nbody.rs:226-331 line by line).  nbody.rs:249-260
says WHEN two bodies merge in terms of the tree as it stands at the arrival; the device has no such tree, only sorted path keys, so the
kernel's rule is a restatement: "B arrives at the leaf of the ONE earlier entity whose (centre's) path shares the most leading digits
with B's key; it merges iff it is within EPS of that entity's running centre; a blob is filed under its centre's path; an earlier arrival
outside the chain at least as deep in B's path is a rival".  If that restatement were wrong the GPU tests would show it only as a wrong
tree; here it is checked on its own: same leaves (blobs folded in arrival order, bit for bit) and same node count as the oracle's tree."""
import os
import sys

import numpy as np
import pytest




def _chains(rng, n0, seeds, longest, step=9e-5, box=20.0):
    x = rng.uniform(-box, box, n0).astype(np.float32)
    y = rng.uniform(-box, box, n0).astype(np.float32)
    xs, ys = [x], [y]
    for s in rng.choice(n0, seeds, replace=False):
        cx, cy = x[s], y[s]
        for _ in range(int(rng.integers(1, longest + 1))):
            cx = np.float32(cx + rng.uniform(-step, step))
            cy = np.float32(cy + rng.uniform(-step, step))
            xs.append(np.array([cx], np.float32))
            ys.append(np.array([cy], np.float32))
    x, y = np.concatenate(xs), np.concatenate(ys)
    order = rng.permutation(len(x))
    return x[order], y[order]


@pytest.mark.parametrize("seed,n0,seeds,longest", [(1, 1200, 250, 5), (2, 400, 200, 10), (5, 3000, 300, 4)])
def test_the_replays_rule_makes_the_oracles_tree(ob, seed, n0, seeds, longest):
    import chain_model as cm

    rng = np.random.default_rng(seed)
    x, y = _chains(rng, n0, seeds, longest)
    n = len(x)
    p = ob.particles(x, y, np.zeros(n), np.zeros(n), rng.uniform(0.5, 2.0, n))
    st = cm.compare(p, K=3, LINK=np.float32(2e-4))          # the shipped parameters: links at 2 EPS, three sorted places ahead
    assert st["only_oracle"] == 0 and st["only_model"] == 0, st
    assert st["model_leaves"] == st["oracle_leaves"] < n - seeds // 2 and st["model_nodes"] == st["oracle_nodes"], st
    assert st["blobs_multi"] >= seeds // 2                   # (blobs did form)


def test_a_blob_is_filed_under_its_centres_path(ob):
    """tests/test_gpu_bh_chains.py::test_a_blob_travels_by_its_centre_..., on the model: A and B merge astride a cell's midline, the
    centre lies on B's side; C splits the leaf and the blob goes where its CENTRE is (nbody.rs:271-281)."""
    import chain_model as cm

    a, b = (-15.0 - 3.0e-5, 1.0, 1.0), (-15.0 + 5.0e-5, 1.0, 3.0)
    for pair in ([a, b], [b, a]):
        pts = [(-30.0, -30.0, 1.0), (30.0, 30.0, 1.0)] + pair + [(-15.0 - 3.0e-5, 1.0 - 1.5e-4, 1.0)]
        p = ob.particles([q[0] for q in pts], [q[1] for q in pts], np.zeros(5), np.zeros(5), [q[2] for q in pts])
        st = cm.compare(p, K=3, LINK=np.float32(2e-4))
        assert st["only_oracle"] == 0 and st["model_leaves"] == 4 and st["model_nodes"] == st["oracle_nodes"], st
