// draw.hip -- nb_draw on the device: replaces the per-frame download of the whole
// particle state (N x 32 B) by a splat kernel + one w*h*4-byte framebuffer download.
//
// Reference semantics (nbody.rs:482-617): for every particle add the body colour 0x0027404C at (x,y) and the
// tail colour 0x0020353F at (x,y) - dir[octant(v)], each with a per-channel SATURATING add, then overwrite
// the 5-pixel magenta centre cross.  Saturating addition of non-negative terms is order independent:
//     channel = min(255, nb * body_channel + nt * tail_channel)
// so the device counts body hits and tail hits per pixel with integer atomics (exact, order free) and a
// resolve pass turns counts into colours.  Pixel coordinates use the reference's f32 expression
// ((p - origin) * scale, truncated toward zero, `as i32` saturating / NaN -> 0 = v_cvt_i32_f32).
//
// BIT-EXACT with the host path / the reference, tails included.  The tail octant is
// ((8*atan2(vy,vx)/(2*pi) + 8) as i32) % 8 evaluated in f32 with the platform libm's atan2f (nbody.rs:541-542), a step
// function of the velocity whose steps sit wherever that f32 expression happens to cross an integer.  The device decides
// the octant from a double-precision evaluation of the same expression (same f32 constants) whenever that value is
// farther than 1e-5 from an integer -- the f32 evaluation (libm error <= 1 ulp of the angle, one rounding in the
// divide, one in the add: <= 1.1e-6 in total) then lands on the same side -- and handles the exact cases every libm
// agrees on (vy = +-0: angle +-0 or +-pi; vx = +-0: +-pi/2, where 8a/2pi is exactly an integer) itself.  Everything
// else -- diagonal velocities, directions within 1e-5/1.27 rad of a multiple of 45 degrees -- is AMBIGUOUS: the particle's
// pixel and velocity go to a short list and the HOST evaluates the reference expression with its own atan2f for those few
// and adds their tail hits to the downloaded framebuffer (a saturating add of one more non-negative term commutes with
// the resolve).  The same pattern as the Barnes-Hut opening test: cheap decision + exact fallback in a narrow band.
#include "kernels.h"

namespace nbx {

__global__ __launch_bounds__(kTile) void k_draw_count(const float4* __restrict__ posm, const float4* __restrict__ vel,
                                                      const int n, const int w, const int h, const float x1,
                                                      const float y1, const float scalex, const float scaley,
                                                      uint2* __restrict__ counts, unsigned* __restrict__ amb_count,
                                                      DrawAmbiguous* __restrict__ amb)
{
    const int k = blockIdx.x * kTile + threadIdx.x;
    if (k >= n) return;
    const float4 p = posm[k];
    const float4 v = vel[k];
    const int xi = (int)__fmul_rn(__fsub_rn(p.x, x1), scalex);   // nbody.rs:525, :536
    const int yi = (int)__fmul_rn(__fsub_rn(p.y, y1), scaley);   // nbody.rs:526, :537
    if (xi >= 0 && xi < w && yi >= 0 && yi < h) atomicAdd(&counts[xi + yi * w].x, 1u);   // :559-565
    int oct = 0;                                                                           // NaN angle: `as i32` gives 0
    if (v.x == v.x && v.y == v.y) {
        if (v.y == 0.0f) {
            oct = (__float_as_uint(v.x) >> 31) ? 4 : 0;       // atan2f(+-0, x): +-0 for x = +0 or x > 0, +-pi for x = -0 or x < 0
        } else if (v.x == 0.0f) {
            oct = v.y > 0.0f ? 2 : 6;                         // +-pi/2: 8 * (pi_f/2) / (2 pi_f) = 2 exactly
        } else {
            const double a = atan2((double)v.y, (double)v.x);
            const double t = 8.0 * a / (2.0 * (double)3.14159274f) + 8.0;                 // :541-542 with the f32 constants
            const double fl = floor(t);
            if (t - fl < 1e-5 || fl + 1.0 - t < 1e-5) {       // too close to a step of the f32 expression: the host decides
                const unsigned slot = atomicAdd(amb_count, 1u);
                amb[slot] = DrawAmbiguous{xi, yi, v.x, v.y};  // capacity n: cannot overflow
                return;
            }
            oct = (int)fl % 8;
        }
    }
    const int dxs[8] = {1, 1, 0, -1, -1, -1, 0, 1};
    const int dys[8] = {0, 1, 1, 1, 0, -1, -1, -1};
    const int xt = xi - dxs[oct], yt = yi - dys[oct];                                     // :553-554
    if (xt >= 0 && xt < w && yt >= 0 && yt < h) atomicAdd(&counts[xt + yt * w].y, 1u);
}

__device__ __forceinline__ unsigned sat_channel(unsigned nb, unsigned cb, unsigned nt, unsigned ct)
{
    // counts can be large: clamp before multiplying (255 hits of any non-zero channel already saturate)
    const unsigned a = (nb > 255u ? 255u : nb) * cb + (nt > 255u ? 255u : nt) * ct;
    return a > 255u ? 255u : a;
}

__global__ __launch_bounds__(kTile) void k_draw_resolve(const uint2* __restrict__ counts, const int w, const int h,
                                                        unsigned* __restrict__ fb)
{
    const int i = blockIdx.x * kTile + threadIdx.x;
    if (i >= w * h) return;
    const uint2 c = counts[i];
    // col_body = 0x0027404C (R 76, G 64, B 39), col_tail = 0x0020353F (R 63, G 53, B 32)   nbody.rs:520-521
    unsigned px = sat_channel(c.x, 76u, c.y, 63u) | (sat_channel(c.x, 64u, c.y, 53u) << 8) |
                  (sat_channel(c.x, 39u, c.y, 32u) << 16);
    if (w >= 3 && h >= 3) {                                   // nbody.rs:571-577
        const int x = i % w, y = i / w, cx = w / 2, cy = h / 2;
        if ((y == cy && (x == cx || x == cx + 1 || x == cx - 1)) || (x == cx && (y == cy + 1 || y == cy - 1)))
            px = 0x00FF00FFu;
    }
    fb[i] = px;
}

hipError_t launch_draw(const float4* posm, const float4* vel, int n, int w, int h, float x1, float y1, float scalex,
                       float scaley, void* counts, unsigned* fb, unsigned* amb_count, DrawAmbiguous* amb, hipStream_t stream)
{
    hipError_t e = hipMemsetAsync(counts, 0, sizeof(uint2) * (size_t)w * (size_t)h, stream);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(amb_count, 0, sizeof(unsigned), stream);
    if (e != hipSuccess) return e;
    if (n > 0)
        hipLaunchKernelGGL(k_draw_count, dim3((n + kTile - 1) / kTile), dim3(kTile), 0, stream, posm, vel, n, w, h, x1, y1,
                           scalex, scaley, static_cast<uint2*>(counts), amb_count, amb);
    hipLaunchKernelGGL(k_draw_resolve, dim3((w * h + kTile - 1) / kTile), dim3(kTile), 0, stream,
                       static_cast<const uint2*>(counts), w, h, fb);
    return hipGetLastError();
}

}  // namespace nbx
