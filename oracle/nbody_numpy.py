"""Independent numpy-float32 restatement of the reference brute-force step and draw colour math.
TEST INFRASTRUCTURE ONLY (pins oracle/nbody_oracle.c: the two must agree bit-for-bit).

Follows /root/reference/rs-src/nbody.rs:
  force()                :164-184   f = (m1*m2)/((dx*dx+dy*dy)+EPS); (f*dx, f*dy)
  nb_step_brute_force    :106-162   per i: sequential f32 sum over ascending j != i, then kick-drift
Sequential summation is reproduced with np.cumsum(dtype=float32) (a strictly left-to-right
running sum), NOT np.sum (pairwise).
"""
import numpy as np

EPS = np.float32(0.0001)  # nbody.rs:17


def brute_forces(px, py, m, i0=0, i1=None):
    px = np.asarray(px, np.float32)
    py = np.asarray(py, np.float32)
    m = np.asarray(m, np.float32)
    n = len(px)
    i1 = n if i1 is None else i1
    fx = np.zeros(i1 - i0, np.float32)
    fy = np.zeros(i1 - i0, np.float32)
    for i in range(i0, i1):
        dx = px - px[i]                       # :174  px2 - px1 (f32)
        dy = py - py[i]                       # :175
        dist_sq = dx * dx + dy * dy           # :176  two roundings for the products, one for the add
        f = (m[i] * m) / (dist_sq + EPS)      # :180
        cx = np.delete(f * dx, i)             # :136 skip by index; :183
        cy = np.delete(f * dy, i)
        if len(cx):
            fx[i - i0] = np.cumsum(cx, dtype=np.float32)[-1]   # :141 sequential ascending-j sum
            fy[i - i0] = np.cumsum(cy, dtype=np.float32)[-1]
    return fx, fy


def step_brute_force(px, py, vx, vy, m, dt):
    """Returns new (px,py,vx,vy). nbody.rs:149-161."""
    dt = np.float32(dt)
    px, py, vx, vy, m = (np.asarray(a, np.float32) for a in (px, py, vx, vy, m))
    fx, fy = brute_forces(px, py, m)
    vx = vx + (dt * fx) / m                   # :155
    vy = vy + (dt * fy) / m                   # :156
    px = px + dt * vx                         # :158 updated v
    py = py + dt * vy
    return px, py, vx, vy
