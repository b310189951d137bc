"""TEST INFRASTRUCTURE ONLY -- a second, independent restatement of the reference's Barnes-Hut step
(rs-src/nbody.rs:186-480) in plain Python with numpy.float32 scalars, recursive like the original, written from the Rust
source and not from oracle/nbody_oracle.c.  It exists to pin the C oracle's tree build and traversal the same way
oracle/nbody_numpy.py pins its brute-force step: tests require the two to agree bit for bit (small cases only --
this is pure-Python loops).  PARITY UNPINNED by the reference itself (it has no tests or vectors): what this adds is a
restatement-vs-restatement check in two languages."""
import numpy as np

F = np.float32
EPS = F(0.0001)            # nbody.rs:17
VP_WDH = F(100.0)          # nbody.rs:13
VP_ORG_X = F(0.0)          # nbody.rs:14
VP_ORG_Y = F(0.0)          # nbody.rs:15


class TreePanic(Exception):
    """A Rust panic! / assert! of the reference (recursion > 50, identical positions, non-positive mass, degenerate split)."""


def force(px1, py1, m1, px2, py2, m2):                    # nbody.rs:164-184
    dx = px2 - px1
    dy = py2 - py1
    dist_sq = dx * dx + dy * dy
    f = (m1 * m2) / (dist_sq + EPS)
    return f * dx, f * dy


class Node:                                               # nbody.rs:207-217
    __slots__ = ("x1", "y1", "x2", "y2", "px", "py", "m", "children")

    def __init__(self, x1, y1, x2, y2):
        self.x1, self.y1, self.x2, self.y2 = x1, y1, x2, y2
        self.px = F(0.0); self.py = F(0.0); self.m = F(0.0)
        self.children = None

    def insert(self, px, py, m, depth):                    # nbody.rs:226-284
        if depth > 50:
            raise TreePanic("recursion")
        if self.children is not None:
            self.add_mass(px, py, m)
            self.children[self.quadrant_from_point(px, py)].insert(px, py, m, depth + 1)
            return
        too_close = abs(self.px - px) < EPS and abs(self.py - py) < EPS
        if self.m == F(0.0) or too_close:
            self.add_mass(px, py, m)
            return
        if not (self.px != px or self.py != py):
            raise TreePanic("identical positions")
        po, qo, mo = self.px, self.py, self.m
        self.px = F(0.0); self.py = F(0.0); self.m = F(0.0)
        self.create_children()
        self.insert(po, qo, mo, depth + 1)
        self.insert(px, py, m, depth + 1)

    def create_children(self):                             # nbody.rs:286-301
        cx = (self.x1 + self.x2) * F(0.5)
        cy = (self.y1 + self.y2) * F(0.5)
        if not (cx > self.x1 or cx < self.x2 or cy > self.y1 or cy < self.y2):
            raise TreePanic("subdivision")
        self.children = [Node(self.x1, cy, cx, self.y2),   # UL
                         Node(cx, cy, self.x2, self.y2),   # UR
                         Node(self.x1, self.y1, cx, cy),   # LL
                         Node(cx, self.y1, self.x2, cy)]   # LR

    def add_mass(self, px, py, m):                         # nbody.rs:303-320
        if not m > F(0.0):
            raise TreePanic("mass")
        if self.m == F(0.0):
            self.px, self.py, self.m = px, py, m
        else:
            inv_msum = F(1.0) / (self.m + m)
            self.px = (self.px * self.m + px * m) * inv_msum
            self.py = (self.py * self.m + py * m) * inv_msum
            self.m = self.m + m

    def quadrant_from_point(self, x, y):                   # nbody.rs:322-331 ; indices UL, UR, LL, LR = 0..3
        cx = (self.x1 + self.x2) * F(0.5)
        cy = (self.y1 + self.y2) * F(0.5)
        if y < cy:
            return 2 if x < cx else 3
        return 0 if x < cx else 1

    def compute_force(self, px, py, m, theta):             # nbody.rs:333-377
        if self.children is not None:
            s = self.x2 - self.x1
            dx = self.px - px
            dy = self.py - py
            with np.errstate(divide="ignore", invalid="ignore"):
                d = np.sqrt(dx * dx + dy * dy)
                approximate = s / d < theta
            if approximate:
                return force(px, py, m, self.px, self.py, self.m)
            fx = F(0.0); fy = F(0.0)
            for c in self.children:
                ax, ay = c.compute_force(px, py, m, theta)
                fx = fx + ax
                fy = fy + ay
            return fx, fy
        if self.px == px and self.py == py:
            return F(0.0), F(0.0)
        if self.m == F(0.0):
            return F(0.0), F(0.0)
        return force(px, py, m, self.px, self.py, self.m)


def step_barnes_hut(px, py, vx, vy, m, theta, dt):
    """One nb_step_barnes_hut (theta != 0) on float32 arrays, in place. Raises TreePanic where the reference panics."""
    theta = F(theta); dt = F(dt)
    n = len(px)
    x1 = y1 = np.finfo(np.float32).max                     # nbody.rs:388-398
    x2 = y2 = np.finfo(np.float32).min
    for i in range(n):
        x1 = px[i] if px[i] < x1 else x1
        y1 = py[i] if py[i] < y1 else y1
        x2 = px[i] if px[i] > x2 else x2
        y2 = py[i] if py[i] > y2 else y2
    tree = Node(F(x1), F(y1), F(x2), F(y2))
    for i in range(n):                                     # nbody.rs:410-415
        tree.insert(px[i], py[i], m[i], 0)
    lim = VP_WDH * F(0.55)
    for i in range(n):                                     # nbody.rs:443-471 (old positions live in the tree snapshot)
        fx, fy = tree.compute_force(px[i], py[i], m[i], theta)
        vx[i] = vx[i] + dt * fx / m[i]
        vy[i] = vy[i] + dt * fy / m[i]
        px[i] = px[i] + dt * vx[i]
        py[i] = py[i] + dt * vy[i]
        if abs(VP_ORG_X - px[i]) > lim or abs(VP_ORG_Y - py[i]) > lim:
            vx[i] = F(0.0)
            vy[i] = F(0.0)
    return tree


# ---- fp64 arbiter (NOT in the reference), second restatement of orc_bh_forces_exact -------------------------------------
def build_tree(px, py, m):
    """The reference tree of nbody.rs:388-415 for float32 arrays (same code path as step_barnes_hut above)."""
    n = len(px)
    x1 = y1 = np.finfo(np.float32).max
    x2 = y2 = np.finfo(np.float32).min
    for i in range(n):
        x1 = px[i] if px[i] < x1 else x1
        y1 = py[i] if py[i] < y1 else y1
        x2 = px[i] if px[i] > x2 else x2
        y2 = py[i] if py[i] > y2 else y2
    tree = Node(F(x1), F(y1), F(x2), F(y2))
    for i in range(n):
        tree.insert(px[i], py[i], m[i], 0)
    return tree


def _exact_sums(node, memo):
    """(mass, mass*x, mass*y) of a node as Python floats (fp64): a leaf is its f32 record, an interior node the sum of its
    children -- exact to fp64 rounding, free of the reference's f32 running fold."""
    if node.children is None:
        v = (float(node.m), float(node.m) * float(node.px), float(node.m) * float(node.py))
    else:
        parts = [_exact_sums(c, memo) for c in node.children]
        v = (sum(p[0] for p in parts), sum(p[1] for p in parts), sum(p[2] for p in parts))
    memo[id(node)] = v
    return v


def bh_forces_exact(px, py, m, theta):
    """Forces through the reference's tree, opening law s/d < theta and pair law, with exact node sums and fp64 arithmetic."""
    import math

    tree = build_tree(px, py, m)
    memo = {}
    _exact_sums(tree, memo)
    eps = float(EPS)
    theta = float(F(theta))

    def pair(x, y, mi, cx, cy, mc):
        dx, dy = cx - x, cy - y
        f = mi * mc / (dx * dx + dy * dy + eps)
        return f * dx, f * dy

    def walk(node, x32, y32, x, y, mi):
        if node.children is not None:
            mm, mx, my = memo[id(node)]
            cx, cy = mx / mm, my / mm
            s = float(node.x2 - node.x1)                   # the f32 box width (nbody.rs:341)
            d = math.sqrt((cx - x) ** 2 + (cy - y) ** 2)
            if d > 0.0 and s / d < theta:
                return pair(x, y, mi, cx, cy, mm)
            fx = fy = 0.0
            for c in node.children:
                ax, ay = walk(c, x32, y32, x, y, mi)
                fx += ax; fy += ay
            return fx, fy
        if (node.px == x32 and node.py == y32) or node.m == F(0.0):    # nbody.rs:365, :368
            return 0.0, 0.0
        return pair(x, y, mi, float(node.px), float(node.py), float(node.m))

    out = np.zeros((len(px), 2))
    for i in range(len(px)):
        out[i] = walk(tree, px[i], py[i], float(px[i]), float(py[i]), float(m[i]))
    return out[:, 0], out[:, 1]
