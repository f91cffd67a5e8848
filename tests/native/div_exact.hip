// GPU check of the contract behind k_path's speculative division (spira_device.h: SpecDiv, normalize(a, SpecDiv&), root_over):
//   whenever the exponent window a SpecDiv has folded is respected (outside_window() == false), every quotient has the bits of the
//   compiler's IEEE expansion of a / b — unit_vector a / sqrt(a.a) component by component, roots n / 2a (a root whose numerator is
//   below 2^-600 / 2^-78, zero included, may differ but is, like the IEEE one, below 1e-70 / 1e-9 in magnitude: rejected against t_min).
// Operands: random mantissas (plus all-ones / all-zeros / nearly-all-ones), both signs, exponents spread over the window and beyond it
// (so flagged cases occur and are counted, not compared), concentrated at the window's edges; vectors with components up to 2^-360
// (Float64) / 2^-62 (Float32) of their length, with +0 and -0 components.
// usage: div_exact <blocks> <iterations per thread> <seed>; prints per precision the mismatches among the unflagged cases, exit 0 iff none
#include <cstdio>
#include <cstdlib>
#include "../../julia-spira_amd/csrc/spira_device.h"

using namespace spira;

__device__ __forceinline__ uint32_t mixh(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <class T> struct Fmt;
template <> struct Fmt<double> { static constexpr int bias = 1023, span = 360, edge_lo = 1023 - 350, edge_hi = 1023 + 350, emax = 0x7FF, sq_span = 180, small = 360; };
template <> struct Fmt<float> { static constexpr int bias = 127, span = 48, edge_lo = 127 - 45, edge_hi = 127 + 45, emax = 0xFF, sq_span = 24, small = 62; };

__device__ __forceinline__ double assemble(uint32_t sign, uint32_t e, uint32_t a, uint32_t b, uint32_t sel, double) {
    uint64_t m = ((uint64_t)(a & 0xFFFFFu) << 32) | b;
    if (sel == 0) m = 0xFFFFFFFFFFFFFull; else if (sel == 1) m = 0; else if (sel == 2) m = 0xFFFFFFFFFFFFFull - (b & 3u);
    return __longlong_as_double((long long)(((uint64_t)sign << 63) | ((uint64_t)e << 52) | m));
}
__device__ __forceinline__ float assemble(uint32_t sign, uint32_t e, uint32_t a, uint32_t, uint32_t sel, float) {
    uint32_t m = a & 0x7FFFFFu;
    if (sel == 0) m = 0x7FFFFFu; else if (sel == 1) m = 0; else if (sel == 2) m = 0x7FFFFFu - (a >> 24 & 3u);
    return __uint_as_float((sign << 31) | (e << 23) | m);
}
// a value whose exponent lies within +-span of the bias (mostly), anywhere (1/16), or exactly on a window edge (1/16)
template <class T> __device__ T value(uint32_t k, uint32_t i, int span) {
    const uint32_t a = mixh(k ^ (0x9e3779b9u * (i + 1))), b = mixh(a + i), c = mixh(b ^ k);
    int e = Fmt<T>::bias - span + (int)((c >> 4) % (uint32_t)(2 * span + 1));
    if ((c & 15u) == 0) e = (int)((c >> 4) % (uint32_t)(Fmt<T>::emax + 1));
    if ((c & 15u) == 1) e = (c & 0x100u) ? Fmt<T>::edge_lo - (int)(c >> 9 & 1u) : Fmt<T>::edge_hi - (int)(c >> 9 & 1u);
    return assemble(a >> 31, (uint32_t)e, a, b, c >> 20 & 31u, (T)0);
}

__device__ __noinline__ double ieee_div(double a, double b) { return a / b; }
__device__ __noinline__ float ieee_div(float a, float b) { return a / b; }
__device__ __forceinline__ bool same(double a, double b) { return __double_as_longlong(a) == __double_as_longlong(b) || (a != a && b != b); }
__device__ __forceinline__ bool same(float a, float b) { return __float_as_uint(a) == __float_as_uint(b) || (a != a && b != b); }

template <class T> __device__ T small_num();
template <> __device__ double small_num<double>() { return 2.409919865102884e-181; }   // 2^-600
template <> __device__ float small_num<float>() { return 3.308722450212111e-24f; }       // 2^-78

template <class T>
__global__ void k_check(uint32_t seed, uint32_t iters, unsigned long long *out) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long bad_n = 0, ok_n = 0, bad_r = 0, ok_r = 0, bad_s = 0, ok_s = 0;
    for (uint32_t it = 0; it < iters; ++it) {
        const uint32_t k = mixh(seed ^ mixh(tid * 0x9e3779b9u + it));
        // ---- unit_vector: components of very different magnitudes, one of them sometimes zero / negative zero
        Vec<T> a = mk<T>(value<T>(k, 0, Fmt<T>::sq_span), value<T>(k, 1, Fmt<T>::sq_span), value<T>(k, 2, Fmt<T>::sq_span));
        if ((k & 7u) == 1) a.y = a.x * value<T>(k, 3, Fmt<T>::small);            // a component far below (or above) the others
        if ((k & 0xFFu) == 2) a.z = (T)0;
        if ((k & 0xFFu) == 3) a.x = -(T)0;
        {
            SpecDiv g;
            const Vec<T> q = normalize(a, g);
            if (!outside_window<T>(g)) {
                const T len = sqrt_rn(dot(a, a));
                ++ok_n;
                bad_n += !same(q.x, ieee_div(a.x, len)) + !same(q.y, ieee_div(a.y, len)) + !same(q.z, ieee_div(a.z, len));
            }
        }
        // ---- the slimmed square root: inside the divisor window it has the bits of the compiler's
        {
            const T x = abs_t(value<T>(k, 7, Fmt<T>::span));
            if (mag_word(x) >= ExpWindow<T>::lo && mag_word(x) < ExpWindow<T>::hi) { ++ok_s; bad_s += !same(sqrt_moderate(x), sqrt_rn(x)); }
            SpecDiv g;
            if (mag_word(x) < ExpWindow<T>::hi) bad_s += !same(root_sqrt<T>(x, g), sqrt_rn(x));   // (the upper bound is root_operands' business) zero and subnormals included
            const T z = (T)0;
            bad_s += !same(root_sqrt<T>(z, g), z);
        }
        // ---- pixel coordinates (:398-399): (i - 1 + xi) / (W - 1), numerator +0 or 2^-21 .. 2^31, no window check in the kernel
        {
            const uint32_t W1 = 1u + (mixh(k ^ 0x51u) % 0x7FFFFFFEu), i1 = mixh(k ^ 0x52u) % (W1 + 1u);
            const T xi = (k & 0x3000u) ? (T)(mixh(k ^ 0x53u) >> 11) * (T)(1.0 / 2097152.0) : (T)0;
            const T n = (T)i1 + xi, dd = (T)W1;
            const Recip<T> rc = recip_of(dd);
            SpecDiv g;
            bad_s += !same(pixel_quotient<T>(n, rc, g), ieee_div(n, dd));
        }
        // ---- the triangle test's f = 1 / a (:161): divisors of both signs, |a| >= 1e-8 as the caller guarantees
        {
            T aa = value<T>(k, 8, Fmt<T>::span);
            if (abs_t(aa) < (T)1e-8) aa = (T)1e-8;
            SpecDiv g;
            const T f = tri_recip<T>(aa, g);
            if (!outside_window<T>(g)) { ++ok_s; bad_s += !same(f, ieee_div((T)1.0, aa)); }
        }
        // ---- roots over 2a: numerators -b -+ sqrt(disc), bounded through b*b and disc as in closest_hit_local()
        {
            const T two_a = abs_t(value<T>(k, 4, Fmt<T>::span)), bq = value<T>(k, 5, Fmt<T>::sq_span + 4), disc = abs_t(value<T>(k, 6, Fmt<T>::span));
            const T sq = (k & 0x300u) ? sqrt_rn(disc) : abs_t(bq);                 // sometimes exactly |b|: one numerator is exactly zero
            const T n0 = -bq - sq, n1 = -bq + sq;
            SpecDiv g;
            const RootDiv<T> rd = root_divisor<T>(two_a, g);
            root_operands<T>(bq * bq, (k & 0x300u) ? disc : bq * bq, g);
            const T r0 = root_over<T>(n0, rd, g), r1 = root_over<T>(n1, rd, g);
            if (!outside_window<T>(g)) {                  // a numerator far below the window: both quotients tiny, whatever they are
                const T w0 = ieee_div(n0, two_a), w1 = ieee_div(n1, two_a);
                const T tiny = sizeof(T) == 8 ? (T)1e-70 : (T)1e-9;
                ++ok_r;
                bad_r += !(same(r0, w0) || (abs_t(n0) < small_num<T>() && abs_t(r0) < tiny && abs_t(w0) < tiny));
                bad_r += !(same(r1, w1) || (abs_t(n1) < small_num<T>() && abs_t(r1) < tiny && abs_t(w1) < tiny));
            }
        }
    }
    atomicAdd(out, bad_n); atomicAdd(out + 1, ok_n); atomicAdd(out + 2, bad_r); atomicAdd(out + 3, ok_r); atomicAdd(out + 4, bad_s); atomicAdd(out + 5, ok_s);
}

template <class T> int run(const char *name, int blocks, uint32_t iters, uint32_t seed) {
    unsigned long long *d, h[6] = {0, 0, 0, 0, 0, 0};
    if (hipMalloc(&d, sizeof h) != hipSuccess || hipMemset(d, 0, sizeof h) != hipSuccess) return 2;
    hipLaunchKernelGGL(k_check<T>, dim3(blocks), dim3(256), 0, 0, seed, iters, d);
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost) != hipSuccess) return 2;
    const unsigned long long n = (unsigned long long)blocks * 256 * iters;
    printf("%s unit_vector: %llu mismatching quotients in %llu unflagged vectors of %llu; roots: %llu mismatching in %llu unflagged pairs of %llu; "
           "square roots, 1/a, pixel quotients: %llu mismatching, %llu window-checked cases\n", name, h[0], h[1], n, h[2], h[3], n, h[4], h[5]);
    (void)hipFree(d);
    if (h[1] < n / 8 || h[3] < n / 8 || h[5] < n / 8) { printf("%s: too few unflagged cases, the test has no power\n", name); return 3; }
    return (h[0] || h[2] || h[4]) ? 1 : 0;
}

int main(int argc, char **argv) {
    const int blocks = argc > 1 ? atoi(argv[1]) : 4096;
    const uint32_t iters = argc > 2 ? (uint32_t)atoi(argv[2]) : 1024, seed = argc > 3 ? (uint32_t)strtoul(argv[3], 0, 0) : 1u;
    const int a = run<double>("f64", blocks, iters, seed), b = run<float>("f32", blocks, iters, seed);
    return a | b;
}
