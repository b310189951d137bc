"""TEST INFRASTRUCTURE ONLY -- an independent restatement of the reference's nb_draw (rs-src/nbody.rs:482-617) in plain
Python with numpy.float32 scalars, written from the Rust source; tests require the C oracle's orc_draw to match it
pixel for pixel.  atan2 is libm's atan2f (what Rust's f32::atan2 calls), reached through ctypes."""
import ctypes
import ctypes.util

import numpy as np

F = np.float32
VP_WDH = F(100.0); VP_ORG_X = F(0.0); VP_ORG_Y = F(0.0)      # nbody.rs:13-15
_libm = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
_libm.atan2f.restype = ctypes.c_float
_libm.atan2f.argtypes = [ctypes.c_float, ctypes.c_float]
PI = F(3.14159274101257324)                                   # std::f32::consts::PI


def rgb_to_abgr32(r, g, b, factor):                           # nbody.rs:585-593
    r = int(F(r) * F(factor)); g = int(F(g) * F(factor)); b = int(F(b) * F(factor))
    return (min(r, 255) << 0) | (min(b, 255) << 16) | (min(g, 255) << 8)


def add_abgr32(c1, c2):                                       # nbody.rs:595-617
    out = 0
    for sh in (24, 16, 8, 0):
        out |= min(255, ((c1 >> sh) & 0xFF) + ((c2 >> sh) & 0xFF)) << sh
    return out


def draw(px, py, vx, vy, w, h):
    fb = [0] * (w * h)                                        # :490
    aspect = F(h) / F(w)                                      # :494
    x1 = VP_ORG_X - VP_WDH / F(2.0)                           # :497-500
    y1 = (VP_ORG_Y - VP_WDH / F(2.0)) * aspect
    x2 = VP_ORG_X + VP_WDH / F(2.0)
    y2 = (VP_ORG_Y + VP_WDH / F(2.0)) * aspect
    scalex = (F(1.0) / (x2 - x1)) * F(w)                      # :503-506
    scaley = (F(1.0) / (y2 - y1)) * F(h)
    col_body = rgb_to_abgr32(255, 215, 130, 0.3)              # :520-521
    col_tail = rgb_to_abgr32(255, 215, 130, 0.25)
    dirs = [(1, 0), (1, 1), (0, 1), (-1, 1), (-1, 0), (-1, -1), (0, -1), (1, -1)]   # E NE N NW W SW S SE
    for k in range(len(px)):
        x = (F(px[k]) - x1) * scalex                          # :525-526
        y = (F(py[k]) - y1) * scaley
        for i in range(2):
            if i == 0:
                xo, yo, col = int(x), int(y), col_body        # `as i32` truncates toward zero
            else:
                angle = F(_libm.atan2f(float(vy[k]), float(vx[k])))                  # :541
                octant = int(F(8.0) * angle / (F(2.0) * PI) + F(8.0)) % 8            # :542
                xo = int(x) - dirs[octant][0]
                yo = int(y) - dirs[octant][1]
                col = col_tail
            if xo < 0 or xo >= w or yo < 0 or yo >= h:        # :559
                continue
            fb[xo + yo * w] = add_abgr32(fb[xo + yo * w], col)
    cx, cy = w // 2, h // 2                                   # :571-577 (callers keep w, h >= 3)
    for dx, dy in ((0, 0), (1, 0), (0, 1), (-1, 0), (0, -1)):
        fb[cx + dx + (cy + dy) * w] = 0x00FF00FF
    return np.array(fb, dtype=np.uint32).reshape(h, w)
