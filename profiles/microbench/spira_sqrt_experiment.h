// RECORDED EXPERIMENT (round 1), not product code: a range-checked correctly rounded square root for gfx950, Float32 and
// Float64, checked by profiles/microbench/sqrt_check.hip against the compiler's expansion over 2^32 inputs each.
// The product (csrc/spira_device.h) uses the compiler's builtin: this variant measured 1.5-2.5 % slower.
//
// The compiler's expansion (`__builtin_sqrt*`, correctly rounded by default under hipcc) spends about a third of
// its instructions on scaling tiny inputs into range and on passing 0 / inf / NaN through.  When EVERY active lane of
// the wave holds a positive, finite, normal input above the scaling threshold (one integer compare per lane, one
// scalar branch per wave) the same refinement runs without those steps: identical arithmetic, hence identical bits
// (0 mismatches over 2^32 inputs per precision).  Measured on k_bounce it is 1.5-2.5 % SLOWER than the compiler's
// branch-free expansion (the extra scalar branch per root costs more than the seven instructions it saves), so it is
// opt-in (-DSPIRA_SQRT_FAST) and the product uses the builtin; kept with its checker as a recorded experiment.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace spira {

__device__ __forceinline__ bool sqrt_in_range(float x) { return (__float_as_uint(x) - 0x0F800000u) < (0x7F800000u - 0x0F800000u); }            // 2^-96 <= x < inf
__device__ __forceinline__ bool sqrt_in_range(double x) { return ((uint32_t)__double2hiint(x) - 0x10000000u) < (0x7FF00000u - 0x10000000u); }   // 2^-767 <= x < inf

__device__ __forceinline__ float sqrt_core(float x) {               // requires sqrt_in_range(x)
    const float s = __builtin_amdgcn_sqrtf(x);                      // v_sqrt_f32: within 1 ulp
    const float s_dn = __uint_as_float(__float_as_uint(s) - 1u), s_up = __uint_as_float(__float_as_uint(s) + 1u);
    const float r_dn = __builtin_fmaf(-s_dn, s, x), r_up = __builtin_fmaf(-s_up, s, x);
    float r = r_dn <= 0.0f ? s_dn : s;
    r = r_up > 0.0f ? s_up : r;
    return r;
}
__device__ __forceinline__ double sqrt_core(double x) {             // requires sqrt_in_range(x)
    const double y = __builtin_amdgcn_rsq(x);                       // v_rsq_f64
    double g = x * y, h = 0.5 * y;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    double d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, x);
    return __builtin_fma(d, h, g);
}

#ifdef SPIRA_SQRT_FAST
__device__ __forceinline__ float sqrt_rn(float x) { return __all(sqrt_in_range(x)) ? sqrt_core(x) : __builtin_sqrtf(x); }
__device__ __forceinline__ double sqrt_rn(double x) { return __all(sqrt_in_range(x)) ? sqrt_core(x) : __builtin_sqrt(x); }
#else                                                               // default: the compiler's expansion
__device__ __forceinline__ float sqrt_rn(float x) { return __builtin_sqrtf(x); }
__device__ __forceinline__ double sqrt_rn(double x) { return __builtin_sqrt(x); }
#endif

}  // namespace spira
