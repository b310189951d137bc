"""The stated fp32 tolerance of the fast mode (SURVEY.md 8(d) / DESIGN.md section 4), as a function of the case:

    1 step:   max|dv| <= 1e-5 * max|a| * dt * max(1, sqrt(N)/64),   max|dp| <= max(1e-5, dt * (that) + 4e-6)
    k steps:  max|dp| <= k * (1-step bound),   max|dv| <= 2.5 * k * (1-step bound)

(k = 10 gives the survey's 1e-4 / 5e-3 on its 4 096-body case where max|a| ~ 2e3.)  `amax` = max over bodies of |a_i| on
the initial state (a = F/m, nbody.rs:140-142,:155).  Shared by the parity tests (tests/conftest.py) and bench.py --verify.
"""
import math


def fast_step_tolerances(amax, n, dt, steps=1):
    """(position bound, velocity bound) after `steps` fast-mode steps against the f32 reference arithmetic."""
    v1 = 1e-5 * float(amax) * dt * max(1.0, math.sqrt(n) / 64.0)
    # positions follow from p' = p + dt * v': a velocity difference dv moves a body by dt * dv (the survey's flat 1e-5 is that
    # on its 4 096-body case; the sqrt(N) growth of the velocity bound carries over at half a million bodies)
    p1 = max(1e-5, dt * v1 + 4e-6)
    if steps <= 1:
        return p1, v1
    return p1 * steps, 2.5 * steps * v1
