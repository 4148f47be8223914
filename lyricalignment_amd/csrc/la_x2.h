// la_x2.h -- device helpers of the "f16x2" scheme (la_f32x2.hip): a float32 value scaled by a power of two splits exactly into two IEEE
// halves hi + lo (22 bits), and a float32 product becomes three f16 MFMA products lo.hi + hi.lo + hi.hi accumulated in float32.
#pragma once
#include "la_common.h"

namespace la {
namespace x2 {

// power of two s with mx * s in [2^13, 2^14); mx == 0 (or not finite) -> 1.  Returns s, *inv = 1 / s (both exact).
__device__ __forceinline__ float scale_for(float mx, float *inv) {
    if (!(mx > 0.f) || !(mx < 3.0e38f)) { *inv = 1.f; return 1.f; }
    int e;
    (void)frexpf(mx, &e);                    // mx = m 2^e, m in [0.5, 1)
    int sh = 14 - e;
    sh = sh > 126 ? 126 : (sh < -126 ? -126 : sh);
    *inv = ldexpf(1.f, -sh);
    return ldexpf(1.f, sh);
}

// x (already scaled) -> f16 hi | f16 lo << 16
__device__ __forceinline__ unsigned pack_hi_lo(float x) {
    const _Float16 h = (_Float16)x;
    const _Float16 l = (_Float16)(x - (float)h);
    return (unsigned)__builtin_bit_cast(unsigned short, h) | ((unsigned)__builtin_bit_cast(unsigned short, l) << 16);
}

}  // namespace x2
}  // namespace la
