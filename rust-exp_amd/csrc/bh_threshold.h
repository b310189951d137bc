// bh_threshold.h -- the reference's opening test (nbody.rs:341-345) as ONE exact comparison per node.
//
//   reference:   s = x2 - x1 ;  dist_sq = dx*dx + dy*dy ;  accept  <=>  fl( s / fl(sqrt(dist_sq)) ) < theta
//
// sqrt and divide are correctly rounded, hence monotone: for a node of size s >= 0 the left-hand side never grows when dist_sq
// grows.  So for every (s, theta) there is one float T with
//
//                accept  <=>  dist_sq > T                      (dist_sq any float >= 0, +inf included; NaN -> not accepted)
//
// bh_take_threshold(s, theta) finds that T by evaluating the reference's own expression on neighbouring floats (a guess from
// (s/theta)^2, its neighbours one by one, and -- should the guess ever be far off -- a bisection over the float bit patterns,
// which order like the floats they encode).  A walk that compares
// the reference's dist_sq (unfused: fl(fl(dx*dx) + fl(dy*dy))) with T makes the reference's decision for EVERY body and node --
// no band around the boundary, no second test -- at the price of one v_cmp.  Rounds 1-3 compared q = s*s with theta^2 * d^2
// and re-made decisions inside a 1e-5 band with the reference's arithmetic: 2 multiplies, 2 compares and a mask test per visit.
//
// T >= 0 always (dist_sq = 0 gives s/0 = +inf or NaN: never accepted, nbody.rs:345);  theta <= 0, NaN theta or NaN s: nothing
// is accepted, T = +inf.  Host and device compute the same T: it is a property of (s, theta), not of the search.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

namespace nbx {

__host__ __device__ inline float bh_bits_to_float(uint32_t u)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __uint_as_float(u);
#else
    float f;
    memcpy(&f, &u, 4);
    return f;
#endif
}
__host__ __device__ inline uint32_t bh_float_to_bits(float f)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __float_as_uint(f);
#else
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
#endif
}

// the reference's test, in the reference's arithmetic (nbody.rs:344-345); translation units that include this header are
// compiled with -ffp-contract=off, and hipcc's sqrtf and '/' are correctly rounded (its default, like Rust's f32::sqrt and '/')
__host__ __device__ inline bool bh_reference_accepts(float s, float dist_sq, float theta)
{
    return s / sqrtf(dist_sq) < theta;
}

__host__ __device__ inline float bh_take_threshold(float s, float theta)
{
    const uint32_t kInf = 0x7F800000u;
    // a guess from (s/theta)^2 in double, then its neighbours one by one: the threshold sits within a few floats of the guess
    // (two roundings of 2^-24 each, the sqrt halving the first), so two or three evaluations of the reference's test settle it
    const double r = (double)s / (double)theta;
    const double g = r * r;
    const float gf = g < 3.0e38 ? (float)g : 3.0e38f;            // (a NaN guess lands on 3e38 too)
    const uint32_t ug = bh_float_to_bits(gf);                    // gf >= 0: a valid position on the pattern axis
    uint32_t lo = 0u, hi = kInf;                                 // bisection bounds if the probing does not settle it
    if (bh_reference_accepts(s, gf, theta)) {                    // T is below the guess
        hi = ug;
        for (uint32_t k = 1u; k <= 4u && k <= ug; k++) {
            if (!bh_reference_accepts(s, bh_bits_to_float(ug - k), theta)) return bh_bits_to_float(ug - k);
            hi = ug - k;
        }
        if (hi == 0u) return 0.0f;                               // (cannot happen: dist_sq = 0 is never accepted)
    } else {                                                     // T is the guess or above it
        lo = ug;
        for (uint32_t k = 1u; k <= 4u && ug + k <= kInf; k++) {
            if (bh_reference_accepts(s, bh_bits_to_float(ug + k), theta)) return bh_bits_to_float(ug + k - 1u);
            lo = ug + k;
        }
        if (lo >= kInf || !bh_reference_accepts(s, bh_bits_to_float(kInf), theta)) return bh_bits_to_float(kInf);   // never accepted
    }
    // invariant: !accepts(lo) && accepts(hi); the bit patterns of non-negative floats order like the floats
    while (hi - lo > 1u) {
        const uint32_t mid = lo + (hi - lo) / 2u;
        if (bh_reference_accepts(s, bh_bits_to_float(mid), theta)) hi = mid;
        else lo = mid;
    }
    return bh_bits_to_float(lo);                                 // the largest dist_sq the reference does NOT accept
}

}  // namespace nbx
