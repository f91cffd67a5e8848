// spira_device.h — gfx950 device code of the SPIRA path-trace hot path (HIP, CDNA4 only).
//
// One ray per lane.  The per-segment arithmetic follows examples/julia-raytracer.jl of the
// reference statement by statement (citations inline) so that path GEOMETRY is bit-identical
// to a same-precision CPU evaluation; the translation unit is built with -ffp-contract=off
// because Julia never fuses a*b+c.  Everything else (iterative throughput form, SoA queues,
// compaction, LDS scene, counter-based RNG) is this build's own design — see DESIGN.md.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "spira_fastdiv.h"

namespace spira {

// correctly rounded square root: hipcc's default expansion of the builtin (IEEE, like Julia's sqrt)
__device__ __forceinline__ float sqrt_rn(float x) { return __builtin_sqrtf(x); }
__device__ __forceinline__ double sqrt_rn(double x) { return __builtin_sqrt(x); }

constexpr int kBlock = 256;          // 4 waves of 64
// k_bounce register budget (second __launch_bounds__ argument = minimum waves per SIMD), from a same-device A/B
// of builds (S1 / S3 Msamples/s): f32 4: 22 766 / 4 296, 5: 22 795 / 4 308, 6: 17 844 / 3 601, 8: 9 037 / 2 050;
// f64 3: 15 368 / 2 876, 4: 15 723 / 2 964, 5: 11 425 / 2 339.  (f64 without a bound takes 132 VGPRs -> 3 waves.)
#ifndef SPIRA_WAVES_F32
#define SPIRA_WAVES_F32 5
#endif
#ifndef SPIRA_WAVES_F64
#define SPIRA_WAVES_F64 4
#endif
#define SPIRA_WAVES_PER_SIMD(T) (sizeof(T) == 8 ? SPIRA_WAVES_F64 : SPIRA_WAVES_F32)
constexpr uint32_t kMaxTries = 64;   // bounded rejection sampling (P(exhaust) ~ 2e-21)

// ------------------------------------------------------------------ small vector algebra
// Operation order mirrors Vec3 of examples/julia-raytracer.jl:11-41.
template <class T> struct Vec { T x, y, z; };

template <class T> __device__ __forceinline__ Vec<T> mk(T x, T y, T z) { Vec<T> r; r.x = x; r.y = y; r.z = z; return r; }
template <class T> __device__ __forceinline__ Vec<T> operator+(Vec<T> a, Vec<T> b) { return mk<T>(a.x + b.x, a.y + b.y, a.z + b.z); }
template <class T> __device__ __forceinline__ Vec<T> operator-(Vec<T> a, Vec<T> b) { return mk<T>(a.x - b.x, a.y - b.y, a.z - b.z); }
template <class T> __device__ __forceinline__ Vec<T> operator*(Vec<T> a, T b) { return mk<T>(a.x * b, a.y * b, a.z * b); }
template <class T> __device__ __forceinline__ Vec<T> operator/(Vec<T> a, T b) { return mk<T>(a.x / b, a.y / b, a.z / b); }
template <class T> __device__ __forceinline__ Vec<T> mulv(Vec<T> a, Vec<T> b) { return mk<T>(a.x * b.x, a.y * b.y, a.z * b.z); }
template <class T> __device__ __forceinline__ T dot(Vec<T> a, Vec<T> b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <class T> __device__ __forceinline__ Vec<T> cross(Vec<T> a, Vec<T> b) {
    return mk<T>(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ float abs_t(float x) { return __builtin_fabsf(x); }
__device__ __forceinline__ double abs_t(double x) { return __builtin_fabs(x); }
template <class T> __device__ __forceinline__ Vec<T> normalize(Vec<T> a) { return a / sqrt_rn(dot(a, a)); }  // :27-28

// ------------------------------------------------------------------ speculative IEEE division (k_path, k_path_metal, k_variant_metal, k_variant_cpu)
// hipcc expands a / b (correctly rounded, as Julia's) into v_div_scale x2, v_rcp, a Newton refinement of the reciprocal, a quotient with
// one (Float64) or two (Float32) residual corrections, v_div_fmas and v_div_fixup: 11 instructions each, and a segment of a path holds
// two vector / scalar divisions (unit normal :139, unit direction :349 / :357) and the roots of its sphere tests (:126, :133, all over
// 2a).  While the exponents of the operands are moderate, v_div_scale returns its operand, v_div_fmas is a plain fma and v_div_fixup
// passes its argument through, so the quotient IS the fma chain of `quotient()` below — bit for bit — and its refined reciprocal is the
// same for every numerator over one denominator: 5 (3) instructions once, then 3 (5) per quotient, in Float64 (Float32).
// What "moderate" costs to test decides whether this pays (measured on S1: unguarded +9 %, a guarded branch per division -1 %), so the
// test is not made where the division stands.  A SpecDiv lane folds magnitudes it relies on into a running minimum / maximum (the
// squares unit_vector forms anyway, b*b and the discriminant of a sphere test: non-negative values, whose high words order like the
// values — one v_max3_u32 / v_min3_u32 each), the wave looks at them once, at the end of the pass, and a wave that saw anything
// outside the window — a subnormal, a huge value, Inf, NaN — reports itself in PathArgs::redo; the exact instantiation of the kernel
// (ExactDiv: the compiler's division everywhere), launched right behind, renders the pass of exactly those waves again and overwrites
// what they wrote.  The only case that is common, a component of a vector that is exactly zero (a 2^-20-grained random offset
// cancelling, an axis-aligned surface, a cross product with a coordinate axis), is handled on the spot: +-0 / length is that zero.
// Results are therefore those of the IEEE division in every case; tests/native/div_exact.hip compares 2^32 quotients per precision over and beyond the window.
struct ExactDiv {};
struct SpecDiv { uint32_t lo = 0xFFFFFFFFu, hi = 0u; };
// magnitude word of a NON-NEGATIVE value (+0 included; a NaN of either sign lands above +Inf): orders like the value
__device__ __forceinline__ uint32_t mag_word(double x) { return (uint32_t)__double2hiint(x); }
__device__ __forceinline__ uint32_t mag_word(float x) { return __float_as_uint(x); }
// The window.  Divisors — 2a of the roots, the length in unit_vector via the sum of squares — within 2^-350 .. 2^350 (Float32:
// 2^-45 .. 2^45; v_div_scale_f32 starts scaling at an exponent difference of 96, _f64 at 768); b*b and the discriminant of a sphere
// test below the same upper bound (so |b| and the square root, hence each numerator -b -+ sqrt, stay below its square root, doubled);
// the single squares of a vector to normalise at least 2^-600 (2^-78) unless the component is a zero: a component may be far smaller
// than the length, its quotient only has to stay a normal number (>= 2^-300 / 2^175, resp. 2^-39 / 2^22).
// A root's numerator has NO lower bound — it is exactly zero for every ray that starts on the sphere it is tested against whenever
// 4a*cc drowns in b*b (:118), which is common.  Far below the window the shared-reciprocal quotient may differ from the IEEE one,
// but both are then smaller than 2^-250 (Float32: 2^-33) in magnitude (|n| < 2^-600 resp. 2^-78, 2a >= 2^-350 resp. 2^-45), the scan
// rejects both against t_min (:120 / :127) and never looks at their value again; closest_hit_local() checks that t_min is large
// enough for this argument.
template <class T> struct ExpWindow;
template <> struct ExpWindow<double> { static constexpr uint32_t lo = (1023u - 350u) << 20, hi = (1023u + 350u) << 20, lo_sq = (1023u - 600u) << 20; };
template <> struct ExpWindow<float> { static constexpr uint32_t lo = (127u - 45u) << 23, hi = (127u + 45u) << 23, lo_sq = (127u - 78u) << 23; };
template <class T> __device__ __forceinline__ bool outside_window(const SpecDiv &g) { return g.lo < ExpWindow<T>::lo || g.hi >= ExpWindow<T>::hi; }
template <class T> __device__ __forceinline__ bool outside_window(const ExactDiv &) { return false; }

template <class T> struct Recip { T d, r; };
__device__ __forceinline__ Recip<double> recip_of(double d) {
    Recip<double> rc; rc.d = d;
    const double r0 = __builtin_amdgcn_rcp(d);
    const double r1 = __builtin_fma(r0, __builtin_fma(-d, r0, 1.0), r0);
    rc.r = __builtin_fma(r1, __builtin_fma(-d, r1, 1.0), r1);
    return rc;
}
__device__ __forceinline__ Recip<float> recip_of(float d) {
    Recip<float> rc; rc.d = d;
    const float r0 = __builtin_amdgcn_rcpf(d);
    rc.r = __builtin_fmaf(__builtin_fmaf(-d, r0, 1.0f), r0, r0);
    return rc;
}
__device__ __forceinline__ double quotient(double n, const Recip<double> &rc) {      // n / rc.d while both are moderate
    const double q = n * rc.r;
    return __builtin_fma(__builtin_fma(-rc.d, q, n), rc.r, q);
}
__device__ __forceinline__ float quotient(float n, const Recip<float> &rc) {
    const float q0 = n * rc.r;
    const float q1 = __builtin_fmaf(__builtin_fmaf(-rc.d, q0, n), rc.r, q0);
    return __builtin_fmaf(__builtin_fmaf(-rc.d, q1, n), rc.r, q1);
}
// The compiler's IEEE square root, minus what moderate operands never use: hipcc scales x up when it is below 2^-767 (Float32: 2^-96)
// and the result back down, and patches x = 0 / Inf through at the end — 8 (6) of its 18 (15) instructions.  The rest, below, is the
// same instruction sequence, so for x inside SpecDiv's window the result has the same bits (tests/native/div_exact.hip).
__device__ __forceinline__ double sqrt_moderate(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = y * 0.5;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g); h = __builtin_fma(h, r, h);
    g = __builtin_fma(__builtin_fma(-g, g, x), h, g);
    return __builtin_fma(__builtin_fma(-g, g, x), h, g);
}
__device__ __forceinline__ float sqrt_moderate(float x) {
    const float sq = __builtin_amdgcn_sqrtf(x);                               // within one ulp: pick among it and its neighbours
    const float dn = __uint_as_float(__float_as_uint(sq) - 1u), up = __uint_as_float(__float_as_uint(sq) + 1u);
    const float vp = __builtin_fmaf(-dn, sq, x), vs = __builtin_fmaf(-up, sq, x);
    float r = (vp <= 0.0f) ? dn : sq;
    r = (vs > 0.0f) ? up : r;
    return r;
}
// unit_vector, :27-28
template <class T> __device__ __forceinline__ Vec<T> normalize(Vec<T> a, ExactDiv &) { return normalize(a); }
template <class T> __device__ __forceinline__ Vec<T> normalize(Vec<T> a, SpecDiv &g) {
    const T sx = a.x * a.x, sy = a.y * a.y, sz = a.z * a.z;
    const T s = sx + sy + sz;                                                // dot(a, a), same association
    // sx >= 2^-600 puts |a.x| >= 2^-300; s < 2^350 puts every |a.i| and the length below 2^175; the length is at least each |a.i|
    const uint32_t ms = mag_word(s);
    g.hi = max(g.hi, ms);
    g.lo = min(g.lo, ms);                                                    // the length as a divisor
    const Recip<T> rc = recip_of(sqrt_moderate(s));
    Vec<T> q = mk<T>(quotient(a.x, rc), quotient(a.y, rc), quotient(a.z, rc));
    const uint32_t least = min(min(mag_word(sx), mag_word(sy)), mag_word(sz));
    if (__builtin_expect(__any(least < ExpWindow<T>::lo_sq), 0)) {           // wave-uniform: a small square — fine if its component is a zero
        // +-0 / length is that zero (the fma chain turns -0 into +0: patched here); anything else this small leaves the window
        if (mag_word(sx) < ExpWindow<T>::lo_sq) { if (a.x == (T)0) q.x = a.x; else g.hi = 0xFFFFFFFFu; }
        if (mag_word(sy) < ExpWindow<T>::lo_sq) { if (a.y == (T)0) q.y = a.y; else g.hi = 0xFFFFFFFFu; }
        if (mag_word(sz) < ExpWindow<T>::lo_sq) { if (a.z == (T)0) q.z = a.z; else g.hi = 0xFFFFFFFFu; }
    }
    return q;
}
// all roots of one ray's sphere tests are quotients over 2a (:126, :133)
template <class T> struct RootDiv { T two_a; Recip<T> rc; };
template <class T> __device__ __forceinline__ RootDiv<T> root_divisor(T two_a, ExactDiv &) { RootDiv<T> r; r.two_a = two_a; r.rc.d = two_a; r.rc.r = 0; return r; }
template <class T> __device__ __forceinline__ RootDiv<T> root_divisor(T two_a, SpecDiv &g) {
    RootDiv<T> r; r.two_a = two_a; r.rc = recip_of(two_a);
    const uint32_t m = mag_word(two_a);                                      // 2a = 2 d.d >= +0
    g.lo = min(g.lo, m); g.hi = max(g.hi, m);
    return r;
}
// a sphere test that has roots: b*b and its discriminant bound the numerators -b -+ sqrt(disc)
template <class T> __device__ __forceinline__ void root_operands(T, T, ExactDiv &) {}
template <class T> __device__ __forceinline__ void root_operands(T bb, T disc, SpecDiv &g) { g.hi = max(max(g.hi, mag_word(bb)), mag_word(disc)); }
// sqrt(disc), :125.  The upper bound is folded by root_operands(); a discriminant below the window — exactly zero is not rare in
// Float32, where b*b and 4a*cc are a few ulps apart for every grazing ray — takes the compiler's square root on the spot.
template <class T> __device__ __forceinline__ T root_sqrt(T disc, ExactDiv &) { return sqrt_rn(disc); }
template <class T> __device__ __forceinline__ T root_sqrt(T disc, SpecDiv &) {
    T sq = sqrt_moderate(disc);
    if (__builtin_expect(mag_word(disc) < ExpWindow<T>::lo, 0)) sq = sqrt_rn(disc);  // rare; (a plain divergent branch: a wave-wide vote here
    return sq;                                                                       //  would keep the sphere loop from being unrolled)
}
template <class T> __device__ __forceinline__ T root_over(T n, const RootDiv<T> &r, ExactDiv &) { return n / r.two_a; }
template <class T> __device__ __forceinline__ T root_over(T n, const RootDiv<T> &r, SpecDiv &) { return quotient(n, r.rc); }
// the smallest t_min for which a root far below the window is rejected whatever its exact value (see ExpWindow)
template <class T> __device__ __forceinline__ void root_t_min(T, ExactDiv &) {}
template <class T> __device__ __forceinline__ void root_t_min(T t_min, SpecDiv &g) { if (!(t_min >= (sizeof(T) == 8 ? (T)1e-70 : (T)1e-9))) g.hi = 0xFFFFFFFFu; }

// ------------------------------------------------------------------ 16/32-byte packets
template <class T> struct alignas(4 * sizeof(T)) Pack4 { T x, y, z, w; };
template <class T> struct alignas(2 * sizeof(T)) Pack2 { T x, y; };
template <class T> struct alignas(sizeof(T)) Pack3 { T x, y, z; };      // radiance triples: no padding moved through HBM

template <class T> struct Bits;
template <> struct Bits<float> {
    static __device__ __forceinline__ float from_u32(uint32_t u) { return __uint_as_float(u); }
    static __device__ __forceinline__ uint32_t to_u32(float f) { return __float_as_uint(f); }
};
template <> struct Bits<double> {
    static __device__ __forceinline__ double from_u32(uint32_t u) { return __longlong_as_double((long long)u); }
    static __device__ __forceinline__ uint32_t to_u32(double f) { return (uint32_t)__double_as_longlong(f); }
};

// ------------------------------------------------------------------ counter-based RNG (DESIGN.md "RNG")
__host__ __device__ __forceinline__ uint32_t mix32(uint32_t x) {   // lowbias32
    x ^= x >> 16; x *= 0x7feb352dU;
    x ^= x >> 15; x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}
struct RngKey { uint32_t hA, hB, hBr; };
__device__ __forceinline__ RngKey rng_key(uint32_t sA, uint32_t sB, uint32_t pixel, uint32_t sample, uint32_t bounce) {
    uint32_t sb = (sample << 8) | bounce;
    RngKey k;
    k.hA = mix32(mix32(sA + pixel) ^ sb);
    k.hB = mix32(mix32(sB ^ pixel) + sb);
    k.hBr = (k.hB << 16) | (k.hB >> 16);
    return k;
}
template <class T>
__device__ __forceinline__ void rng3(const RngKey &k, uint32_t t, T &u0, T &u1, T &u2, const T s = (T)(1.0 / 2097152.0)) {
    uint32_t a = mix32((k.hA + t * 0x9E3779B9u) ^ k.hBr);
    uint32_t b = mix32(a + k.hB);
    u0 = (T)(a >> 11) * s;
    u1 = (T)(b >> 11) * s;
    u2 = (T)(((a & 0x7FFu) << 10) | (b & 0x3FFu)) * s;
}

// ------------------------------------------------------------------ scene in LDS
// Layout of the dynamic LDS block (all offsets multiples of 32 bytes):
//   sph  : n_spheres   x Pack4<T>  {cx, cy, cz, r*r}   (one ds_read_b128 per sphere test, f32)
//   tri  : n_triangles x 3 x Pack4<T> {v0,n.x} {e1,n.y} {e2,n.z}   (n = unit geometric normal)
//   mat  : n_materials x 2 x Pack4<T> {diffuse, specular} {emission, roughness}
//   smat : n_spheres   x int32 (0-based material)   tmat : n_triangles x int32
template <class T> struct SceneLds {
    const Pack4<T> *sph;
    const Pack4<T> *tri;
    const Pack4<T> *mat;
    const int *smat;
    const int *tmat;
    uint32_t n_spheres, n_triangles;
    // large meshes: 8-wide quantised BVH in global memory (spira_bvh.h), L2 / Infinity-Cache resident
    const uint4 *bvh_nodes;     // 5 x uint4 per node slot
    const Pack4<T> *bvh_tris;
    const uint4 *bvh_tris32;    // Float64 scenes: the Float32 screening records of the walk (3 x uint4 per triangle, normalised frame), else NULL
    uint32_t n_bvh_tris;
    const Pack4<T> *bvh_root;   // LDS: {min}, {max} of the whole mesh's padded box (caller's coordinates, render precision) — a ray that misses
                                // it never touches the tree — and {centre, scale} of the normalised frame the node boxes live in
    const T *spd;           // SPIRA_EXT_SPECTRAL: kSpdRows x kSpdN table (include/spira_spd.h) staged behind the scene, else unused
};

constexpr int kSpdN = 36, kSpdRows = 6;      // include/spira_spd.h
constexpr uint32_t kExtDielectric = 0x00020000u, kExtSpectral = 0x00040000u;   // SPIRA_EXT_* (include/spira_hip.h)

template <class T> struct SceneGlobal {     // flat arrays exactly as passed through the C ABI
    const T *spheres5;      // prepare_scene_data, src/spira-metal-optimized.jl:515-529
    const T *materials8;    // :531-541
    const T *triangles10;   // staged into LDS (small counts); NULL / 0 when the mesh goes through the BVH
    uint32_t n_spheres, n_materials, n_triangles;
    const uint4 *bvh_nodes;         // spira_bvh.h: 5 x uint4 per node slot, slot 0 = root
    const Pack4<T> *bvh_tris;       // 3 packets per triangle, node order
    const uint4 *bvh_tris32;        // Float64: 3 x uint4 per triangle, the same order — {v0, index} {e1, L} {e2, 0} in Float32, normalised frame (spira_bvh.h)
    const Pack4<T> *bvh_frame;      // 3 packets: root box min, max (caller's coordinates), {centre, scale}
    uint32_t n_bvh_tris;
    uint32_t bvh_slots;             // node slots in bvh_nodes
    const T *spd;           // device copy of the SPD table, or NULL (extension off)
};

// box tests only (conservative: the boxes are padded by 1e-4 of the mesh's extent, these reciprocals are good to 1e-7 / 1e-15)
__device__ __forceinline__ float rcp_fast(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ double rcp_fast(double x) {       // v_rcp_f64 + one Newton step (3 instructions; the IEEE division is ~30).  x = 0: NaN, which
    const double r = __builtin_amdgcn_rcp(x);                 // the NaN-ignoring min / max of box_entry() read as "no constraint from this axis"
    return __builtin_fma(r, __builtin_fma(-x, r, 1.0), r);
}
__device__ __forceinline__ float min_nn(float a, float b) { return __builtin_fminf(a, b); }  // NaN-ignoring
__device__ __forceinline__ float max_nn(float a, float b) { return __builtin_fmaxf(a, b); }
__device__ __forceinline__ double min_nn(double a, double b) { return __builtin_fmin(a, b); }
__device__ __forceinline__ double max_nn(double a, double b) { return __builtin_fmax(a, b); }

template <class T> __host__ __device__ inline size_t scene_lds_bytes(uint32_t ns, uint32_t nm, uint32_t nt) {
    size_t b = (size_t)ns * sizeof(Pack4<T>) + (size_t)nt * 3 * sizeof(Pack4<T>) + (size_t)nm * 2 * sizeof(Pack4<T>);
    b += ((size_t)ns + nt) * sizeof(int);
    b = (b + 31) & ~(size_t)31;
    b += (size_t)kSpdRows * kSpdN * sizeof(T);        // the SPD table's slot (1.7 KB in Float64; filled only in spectral mode)
    b = (b + 31) & ~(size_t)31;
    b += 3 * sizeof(Pack4<T>);                        // the mesh's bounding box and normalised frame (BVH scenes)
    return (b + 31) & ~(size_t)31;
}

template <class T>
__device__ __forceinline__ SceneLds<T> stage_scene(const SceneGlobal<T> &g, unsigned char *lds) {
    Pack4<T> *sph = reinterpret_cast<Pack4<T> *>(lds);
    Pack4<T> *tri = sph + g.n_spheres;
    Pack4<T> *mat = tri + 3 * (size_t)g.n_triangles;
    int *smat = reinterpret_cast<int *>(mat + 2 * (size_t)g.n_materials);
    int *tmat = smat + g.n_spheres;
    for (uint32_t i = threadIdx.x; i < g.n_spheres; i += blockDim.x) {
        const T *s = g.spheres5 + 5 * (size_t)i;
        Pack4<T> p; p.x = s[0]; p.y = s[1]; p.z = s[2]; p.w = s[3] * s[3];     // radius*radius, :117
        sph[i] = p;
        smat[i] = (int)s[4] - 1;                                               // 1-based float -> 0-based
    }
    for (uint32_t i = threadIdx.x; i < g.n_triangles; i += blockDim.x) {
        const T *t = g.triangles10 + 10 * (size_t)i;
        Pack4<T> v0, e1, e2;
        v0.x = t[0]; v0.y = t[1]; v0.z = t[2]; v0.w = 0;
        e1.x = t[3] - t[0]; e1.y = t[4] - t[1]; e1.z = t[5] - t[2]; e1.w = 0;  // edge1 = v1 - v0, :149
        e2.x = t[6] - t[0]; e2.y = t[7] - t[1]; e2.z = t[8] - t[2]; e2.w = 0;  // edge2 = v2 - v0, :150
        // the geometric normal depends on the triangle only: normalize(cross(edge1, edge2)), :105-109, evaluated here once
        // per workgroup instead of once per hit (same operations on the same values, so the same bits)
        const Vec<T> nrm = normalize(cross(mk<T>(e1.x, e1.y, e1.z), mk<T>(e2.x, e2.y, e2.z)));
        v0.w = nrm.x; e1.w = nrm.y; e2.w = nrm.z;
        tri[3 * i] = v0; tri[3 * i + 1] = e1; tri[3 * i + 2] = e2;
        tmat[i] = (int)t[9] - 1;
    }
    for (uint32_t i = threadIdx.x; i < g.n_materials; i += blockDim.x) {
        const T *m = g.materials8 + 8 * (size_t)i;
        Pack4<T> a, b;
        a.x = m[0]; a.y = m[1]; a.z = m[2]; a.w = m[6];     // diffuse|albedo, specular|metallic
        b.x = m[3]; b.y = m[4]; b.z = m[5]; b.w = m[7];     // emission, roughness
        mat[2 * i] = a; mat[2 * i + 1] = b;
    }
    const size_t spd_off = (((size_t)g.n_spheres * sizeof(Pack4<T>) + (size_t)g.n_triangles * 3 * sizeof(Pack4<T>) + (size_t)g.n_materials * 2 * sizeof(Pack4<T>) +
                             ((size_t)g.n_spheres + g.n_triangles) * sizeof(int)) + 31) & ~(size_t)31;
    T *spd = reinterpret_cast<T *>(lds + spd_off);
    if (g.spd)
        for (uint32_t i = threadIdx.x; i < (uint32_t)(kSpdRows * kSpdN); i += blockDim.x) spd[i] = g.spd[i];
    Pack4<T> *root = reinterpret_cast<Pack4<T> *>(lds + ((spd_off + (size_t)kSpdRows * kSpdN * sizeof(T) + 31) & ~(size_t)31));
    if (g.n_bvh_tris && threadIdx.x < 3) root[threadIdx.x] = g.bvh_frame[threadIdx.x];
    __syncthreads();
    SceneLds<T> sc;
    sc.sph = sph; sc.tri = tri; sc.mat = mat; sc.smat = smat; sc.tmat = tmat;
    sc.n_spheres = g.n_spheres; sc.n_triangles = g.n_triangles;
    sc.bvh_nodes = g.bvh_nodes; sc.bvh_tris = g.bvh_tris; sc.bvh_tris32 = g.bvh_tris32; sc.n_bvh_tris = g.n_bvh_tris;
    sc.spd = spd;
    sc.bvh_root = root;
    return sc;
}

// ------------------------------------------------------------------ per-render constants
template <class T> struct RenderConst {
    FastDiv fd_tile, fd_width, fd_stripe;   // tile_pixels, width, stripe_h
    Vec<T> cam_origin, cam_llc, cam_hor, cam_ver;
    uint32_t width, height;      // full image
    uint32_t spp, max_depth;
    uint32_t sA, sB;             // mixed seed halves
    uint32_t flags;
    // tile
    uint32_t rows;               // rows rendered by this call
    uint32_t row0, stripe_h, stripe_count, stripe_rank;
    uint32_t tile_pixels;        // rows * width
    uint32_t slots;              // k: sample slots per pass
    uint32_t sample0;            // index of the first sample of this call (progressive accumulation; else 0)
};

// local output row -> reference loop row j (1-based, j = 1 is v = 0, the image bottom)
template <class T> __device__ __forceinline__ uint32_t ref_row_j(const RenderConst<T> &rc, uint32_t lr) {
    uint32_t y = rc.row0 + lr;
    if (rc.stripe_count > 1) {
        const uint32_t sq = fastdiv(lr, rc.fd_stripe);           // lr / stripe_h
        y = (sq * rc.stripe_count + rc.stripe_rank) * rc.stripe_h + (lr - sq * rc.stripe_h);
    }
    return (rc.flags & 0x00001000u /*SPIRA_ROWS_BOTTOM_UP*/) ? y + 1 : rc.height - y;   // hdr_data[height-j+1, i], :408
}

// ------------------------------------------------------------------ closest hit (semantics A)
// Möller–Trumbore, examples/julia-raytracer.jl:145-187, on precomputed edges.  Returns true and t when the
// triangle is hit with t_min <= t <= t_max (the comparison with the running closest hit is the caller's).
// f = 1 / a, :161.  SpecDiv: |a| >= 1e-8 has just been checked (:157, inside the window's lower bound in both precisions), the upper bound is folded
template <class T> __device__ __forceinline__ T tri_recip(T aa, ExactDiv &) { return (T)1.0 / aa; }
template <class T> __device__ __forceinline__ T tri_recip(T aa, SpecDiv &g) {
    g.hi = max(g.hi, mag_word(abs_t(aa)));
    return quotient((T)1.0, recip_of(aa));
}
template <class T, class P>
__device__ __forceinline__ bool triangle_test(const Pack4<T> v0, const Pack4<T> e1p, const Pack4<T> e2p, Vec<T> o, Vec<T> d, T t_min, T t_max, T &t_out, P &pol) {
    Vec<T> e1 = mk<T>(e1p.x, e1p.y, e1p.z), e2 = mk<T>(e2p.x, e2p.y, e2p.z);
    Vec<T> h = cross(d, e2);                               // :153
    T aa = dot(e1, h);                                     // :154
    if (abs_t(aa) < (T)1e-8) return false;                 // :157
    T f = tri_recip<T>(aa, pol);                           // :161
    Vec<T> sv = o - mk<T>(v0.x, v0.y, v0.z);               // :162
    T u = f * dot(sv, h);                                  // :163
    if (u < (T)0.0 || u > (T)1.0) return false;            // :165
    Vec<T> q = cross(sv, e1);                              // :169
    T v = f * dot(d, q);                                   // :170
    if (v < (T)0.0 || u + v > (T)1.0) return false;        // :172
    T t = f * dot(e2, q);                                  // :177
    if (t < t_min || t > t_max) return false;              // :179
    t_out = t;
    return true;
}
template <class T>
__device__ __forceinline__ bool triangle_test(const Pack4<T> v0, const Pack4<T> e1p, const Pack4<T> e2p, Vec<T> o, Vec<T> d, T t_min, T t_max, T &t_out) {
    ExactDiv exact;
    return triangle_test<T>(v0, e1p, e2p, o, d, t_min, t_max, t_out, exact);
}


constexpr int kBvhStackD = 64;

// Conservative ray / padded-box slab test: entry distance, or -1 when the box cannot contain a hit <= best.
template <class T>
__device__ __forceinline__ T box_entry(const Pack4<T> mn, const Pack4<T> mx, Vec<T> o, Vec<T> inv, T best) {
    T x1 = (mn.x - o.x) * inv.x, x2 = (mx.x - o.x) * inv.x;
    T y1 = (mn.y - o.y) * inv.y, y2 = (mx.y - o.y) * inv.y;
    T z1 = (mn.z - o.z) * inv.z, z2 = (mx.z - o.z) * inv.z;
    T enter = max_nn(max_nn(min_nn(x1, x2), min_nn(y1, y2)), min_nn(z1, z2));
    T exit_ = min_nn(min_nn(max_nn(x1, x2), max_nn(y1, y2)), max_nn(z1, z2));
    return (enter <= exit_ && exit_ >= (T)0 && enter <= best) ? max_nn(enter, (T)0) : (T)-1;
}

// ------------------------------------------------------------------ 8-wide quantised BVH (layout and rationale: spira_bvh.h)
// Returns exactly what the reference's linear scan over the same triangles returns: minimal t, ties to the LATER triangle of the
// caller's array (`t > closest_so_far` rejects, :179).  Box tests run in Float32 in the mesh's normalised frame whatever the render
// precision (they only prune, conservatively); the triangle test is the scan's own arithmetic in T on the caller's coordinates.
// A ray is assumed to have a direction of unit length (every ray of the kernels is normalised): its parameter then measures
// distance, which is what the clamp of 1/d below and the padding of the boxes are sized for.
struct Bvh8Ray {
    float ox, oy, oz;        // origin in the normalised frame: the point where the ray enters the mesh's box (or its own origin inside it)
    float ix, iy, iz;        // 1 / d, magnitude clamped to 2^40 (a slab the ray runs parallel to: inside -> (-huge, huge), outside -> beyond every exit)
    uint32_t oct;            // bit k: d_k < 0
    float best;              // closest hit so far as a distance from (ox, oy, oz), normalised units, rounded up
    float tmin;              // Float64 walks (their Float32 screen): t_min as a distance from (ox, oy, oz), normalised units, rounded down
    float amin;              // ... and the scan's |a| >= 1e-8 (:157) in the normalised frame's units, rounded up: 1e-8 scale^2
};
template <class T> struct Bvh8Walk {      // one lane's traversal state
    uint32_t G;              // current node group: child_base << 8 | hit children still to visit, bit (slot ^ oct): ascending = front to back
    uint32_t tw;             // triangles of the current node still to test: hit leaf slots << 24 | the node's tri_base
    uint32_t rank;           // the node's leaf ranks (4 bits per slot): the triangle of leaf slot s is tri_base + rank_s
    int sp;
    T t0;                    // ray parameter of (ox, oy, oz)
    uint32_t c0, c1, c2, c3, nc;   // Float64 walk: the triangles its Float32 screen could not reject (newest first), nc of them: their exact tests wait for a batch
};
constexpr uint32_t kBvhCand = 4;

__device__ __forceinline__ float bvh8_rcp(float dx, bool neg) {
    const float a = __builtin_fmaxf(__builtin_fabsf(dx), 9.094947017729282e-13f);     // 2^-40
    const float r = __builtin_amdgcn_rcpf(a);
    return neg ? -r : r;
}
__device__ __forceinline__ float bvh8_best(float x) { return x * 1.00000095367431640625f; }      // (1 + 2^-20): rounded up for sure

// Root box test (render precision, caller's coordinates) and the ray's Float32 side.  false: the ray cannot hit the mesh before `closest`.
template <class T>
__device__ __forceinline__ bool bvh8_enter(const SceneLds<T> &sc, Vec<T> o, Vec<T> d, T closest, Bvh8Ray &r, T &t0) {
    const Vec<T> inv = mk<T>(rcp_fast(d.x), rcp_fast(d.y), rcp_fast(d.z));
    const T te = box_entry<T>(sc.bvh_root[0], sc.bvh_root[1], o, inv, closest);
    if (te < (T)0) return false;
    t0 = te;
    const Pack4<T> fr = sc.bvh_root[2];
    const Vec<T> on = ((o + d * te) - mk<T>(fr.x, fr.y, fr.z)) * fr.w;
    r.ox = (float)on.x; r.oy = (float)on.y; r.oz = (float)on.z;
    const float dx = (float)d.x, dy = (float)d.y, dz = (float)d.z;
    const bool nx = dx < 0.0f, ny = dy < 0.0f, nz = dz < 0.0f;
    r.oct = (nx ? 1u : 0u) | (ny ? 2u : 0u) | (nz ? 4u : 0u);
    r.ix = bvh8_rcp(dx, nx); r.iy = bvh8_rcp(dy, ny); r.iz = bvh8_rcp(dz, nz);
    r.best = bvh8_best((float)((closest - te) * fr.w));
    r.tmin = 0.0f; r.amin = 0.0f;      // (set by bvh8_begin)
    return true;
}

typedef float F2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ F2 bvh8_bytes(uint32_t w0, uint32_t w1, int k) {      // byte k of both words as floats (v_cvt_f32_ubyteK x 2): children k and k + 4
    F2 r; r.x = (float)((w0 >> (8 * k)) & 0xFFu); r.y = (float)((w1 >> (8 * k)) & 0xFFu); return r;
}

// One node: slab tests of its 8 children, two at a time (v_pk_fma_f32).  Out: ih = hit child NODES, bit (slot ^ oct); lh = hit leaf slots.
__device__ __forceinline__ void bvh8_node(const uint4 w0, const uint4 w1, const uint4 w2, const uint4 w3, const uint4 w4, const Bvh8Ray &r,
                                          uint32_t &ih, uint32_t &lh) {
    const float sx = __uint_as_float((w0.w & 0xFFu) << 23), sy = __uint_as_float(((w0.w >> 8) & 0xFFu) << 23), sz = __uint_as_float(((w0.w >> 16) & 0xFFu) << 23);
    const uint32_t imask = w0.w >> 24;
    const float ax = sx * r.ix, ay = sy * r.iy, az = sz * r.iz;
    const float bx = (__uint_as_float(w0.x) - r.ox) * r.ix, by = (__uint_as_float(w0.y) - r.oy) * r.iy, bz = (__uint_as_float(w0.z) - r.oz) * r.iz;
    const F2 ax2 = {ax, ax}, ay2 = {ay, ay}, az2 = {az, az}, bx2 = {bx, bx}, by2 = {by, by}, bz2 = {bz, bz};
    // near / far planes per axis by the sign of the direction: lo_x = w2.xy, lo_y = w2.zw, lo_z = w3.xy, hi_x = w3.zw, hi_y = w4.xy, hi_z = w4.zw
    const bool nx = (r.oct & 1u) != 0, ny = (r.oct & 2u) != 0, nz = (r.oct & 4u) != 0;
    const uint32_t nxw[2] = {nx ? w3.z : w2.x, nx ? w3.w : w2.y}, fxw[2] = {nx ? w2.x : w3.z, nx ? w2.y : w3.w};
    const uint32_t nyw[2] = {ny ? w4.x : w2.z, ny ? w4.y : w2.w}, fyw[2] = {ny ? w2.z : w4.x, ny ? w2.w : w4.y};
    const uint32_t nzw[2] = {nz ? w4.z : w3.x, nz ? w4.w : w3.y}, fzw[2] = {nz ? w3.x : w4.z, nz ? w3.y : w4.w};
    uint32_t hlo = 0, hhi = 0;                                        // children 0..3, children 4..7
#pragma unroll
    for (int k = 3; k >= 0; --k) {
        const F2 tnx = __builtin_elementwise_fma(bvh8_bytes(nxw[0], nxw[1], k), ax2, bx2), tfx = __builtin_elementwise_fma(bvh8_bytes(fxw[0], fxw[1], k), ax2, bx2);
        const F2 tny = __builtin_elementwise_fma(bvh8_bytes(nyw[0], nyw[1], k), ay2, by2), tfy = __builtin_elementwise_fma(bvh8_bytes(fyw[0], fyw[1], k), ay2, by2);
        const F2 tnz = __builtin_elementwise_fma(bvh8_bytes(nzw[0], nzw[1], k), az2, bz2), tfz = __builtin_elementwise_fma(bvh8_bytes(fzw[0], fzw[1], k), az2, bz2);
        const float tn0 = __builtin_fmaxf(__builtin_fmaxf(tnx.x, tny.x), __builtin_fmaxf(tnz.x, 0.0f)), tf0 = __builtin_fminf(__builtin_fminf(tfx.x, tfy.x), __builtin_fminf(tfz.x, r.best));
        const float tn1 = __builtin_fmaxf(__builtin_fmaxf(tnx.y, tny.y), __builtin_fmaxf(tnz.y, 0.0f)), tf1 = __builtin_fminf(__builtin_fminf(tfx.y, tfy.y), __builtin_fminf(tfz.y, r.best));
        hlo = hlo + hlo + ((tn0 <= tf0) ? 1u : 0u);                   // (descending k: child k ends up at bit k)
        hhi = hhi + hhi + ((tn1 <= tf1) ? 1u : 0u);
    }
    const uint32_t hits = hlo | (hhi << 4);
    uint32_t ihs = hits & imask;
    lh = hits & ~imask;
    if (nx) ihs = ((ihs & 0x55u) << 1) | ((ihs >> 1) & 0x55u);       // bit s -> bit s ^ oct
    if (ny) ihs = ((ihs & 0x33u) << 2) | ((ihs >> 2) & 0x33u);
    if (nz) ihs = ((ihs & 0x0Fu) << 4) | (ihs >> 4);
    ih = ihs;
}

// a triangle's three packets out of the words a trip loaded (Float32: 3 x uint4, Float64: 6)
__device__ __forceinline__ void bvh8_tri_words(const uint4 w0, const uint4 w1, const uint4 w2, const uint4, const uint4, const uint4, Pack4<float> &v0, Pack4<float> &e1, Pack4<float> &e2) {
    v0.x = __uint_as_float(w0.x); v0.y = __uint_as_float(w0.y); v0.z = __uint_as_float(w0.z); v0.w = __uint_as_float(w0.w);
    e1.x = __uint_as_float(w1.x); e1.y = __uint_as_float(w1.y); e1.z = __uint_as_float(w1.z); e1.w = __uint_as_float(w1.w);
    e2.x = __uint_as_float(w2.x); e2.y = __uint_as_float(w2.y); e2.z = __uint_as_float(w2.z); e2.w = __uint_as_float(w2.w);
}
__device__ __forceinline__ void bvh8_tri_words(const uint4 w0, const uint4 w1, const uint4 w2, const uint4 w3, const uint4 w4, const uint4 w5, Pack4<double> &v0, Pack4<double> &e1, Pack4<double> &e2) {
    v0.x = __hiloint2double((int)w0.y, (int)w0.x); v0.y = __hiloint2double((int)w0.w, (int)w0.z); v0.z = __hiloint2double((int)w1.y, (int)w1.x); v0.w = __hiloint2double((int)w1.w, (int)w1.z);
    e1.x = __hiloint2double((int)w2.y, (int)w2.x); e1.y = __hiloint2double((int)w2.w, (int)w2.z); e1.z = __hiloint2double((int)w3.y, (int)w3.x); e1.w = __hiloint2double((int)w3.w, (int)w3.z);
    e2.x = __hiloint2double((int)w4.y, (int)w4.x); e2.y = __hiloint2double((int)w4.w, (int)w4.z); e2.z = __hiloint2double((int)w5.y, (int)w5.x); e2.w = __hiloint2double((int)w5.w, (int)w5.z);
}

// start at the root: a group holding only slot 0 (child_base 0, bit 0 ^ oct: slot = bit ^ oct = 0)
template <class T> __device__ __forceinline__ void bvh8_begin(Bvh8Walk<T> &w, Bvh8Ray &r, T t0, T t_min, T scale) {
    w.G = 1u << r.oct; w.tw = 0; w.rank = 0; w.sp = 0; w.t0 = t0; w.c0 = w.c1 = w.c2 = w.c3 = 0; w.nc = 0;
    // t_min seen from the entry point, rounded DOWN (a smaller bound rejects less): one part in 2^20 and a denormal-proof 2^-60 below the converted value
    const float tm = (float)((t_min - t0) * scale);
    r.tmin = tm - __builtin_fabsf(tm) * 9.5367431640625e-7f - 8.673617379884035e-19f;
    const float sc32 = (float)scale;                     // (a scale beyond Float32: amin = Inf, no triangle is ever "certain"; below it: 0, every |a| above its error bound is)
    r.amin = ((1.0000001e-8f * sc32) * sc32) * 1.00000095367431640625f;
}

// ---- Float64 walks screen their triangles in Float32: BUILT, VERIFIED, MEASURED SLOWER, NOT THE DEFAULT (-DSPIRA_BVH_SCREEN) ----
// Round 4, same box, alternating libraries, 1080p spp 64 depth 12, Float64: mesh stress scene S5 67.4 -> 71.1 ms per frame (+5.5 %), BASELINE configs[4]
// 5.78 -> 5.91 ms (+2 %), although a ray needs 14 % fewer trips (17.7 -> 15.2, the Float32 walk's count).  Why: a trip is bound by its memory round trip, not by
// its arithmetic (5 100 .. 5 450 cycles per wave-trip whichever arms run, profiles/experiments/mesh_stats.py), and the deferred exact tests are round trips
// of their own (one per group of finished lanes, every ~5.5 trips).  The first form — exact test in the lane's next trip — lost 8 %: with 64 lanes some lane
// holds a candidate in 92 % of the trips, so the wave ran the Float64 arm almost every trip anyway.  Kept as an experiment build with its contract test.
// A Float64 triangle record is 96 bytes and its test ~75 Float64 instructions; four out of five tests of a walk reject.  So a Float64 walk tests the
// Float32 copy of the triangle (48 bytes, the record a Float32 walk loads; normalised frame, ray from its entry point into the mesh's box) with every
// comparison widened by a bound on what Float32 rounding can have done to its two sides.  A triangle that MAY pass becomes a candidate of the ray, and the
// candidates are put through the scan's own Float64 test, which alone decides (so the result is exactly the linear scan's, as before) — not in the walk's
// loop, where one lane of 64 with a candidate would make the whole wave run the Float64 arm (measured: 92 % of the trips, S5 +8 % frame time), but in
// batches: when the lanes that finished their walk hand over their results, or when a lane's list is full.  A triangle that passes FOR SURE bounds the hit
// distance from above, and the walk prunes with that bound.
// "May pass" must hold for every triangle the exact test accepts.  With u = 2^-24, all inputs rounded once from their Float64 values (|coordinates| <=
// ~0.55 in the normalised frame, |d| = 1), Le = max(|e1|_inf, |e2|_inf) (stored with the record, rounded up), sm = |sv|_inf, plain left-to-right arithmetic:
//   |d(sv)|_inf <= u (|P| + |v0| + |sv|)_inf <= 2.2 u + u sm        |d(h)|_inf <= 8 u Le  (h = d x e2, |h|_inf <= 2 Le)
//   |d(a)|  <= 48 u Le^2                                            (a = e1 . h)
//   |d(nu)|, |d(nv)| <= u Le (19.2 + 42 sm)                         (nu = sv . h;  q = sv x e1, |d(q)|_inf <= u Le (6.4 + 6 sm);  nv = d . q)
//   |d(nt)| <= u Le^2 (19.2 + 42 sm)                                (nt = e2 . q)
// used below as EL = (32 + 64 sm) u Le, ET = EL Le, EA = 64 u Le^2 (margins of 1.3 .. 1.6 over the bounds; the sums and products of the comparisons
// themselves add a few u of their own, inside those margins).  The exact test's conditions in these terms, s = sign(a):
//   u >= 0: s nu >= 0;  v >= 0: s nv >= 0;  u + v <= 1: s (nu + nv) <= |a|;  t in [t_min, closest]: tmin |a| <= s nt <= best |a|
// (u <= 1 follows from v >= 0 and u + v <= 1).  |a| <= EA: the sign is not known, the triangle may pass.  The Float64 evaluation of the exact test
// deviates from these real-number conditions by parts in 2^53 of the same terms: eight orders of magnitude inside the bounds.
// Returns 0: the exact test certainly rejects; 1: it may accept; 2: it certainly accepts unless a closer hit is known — then t_hi bounds the hit distance
// (normalised units from the entry point) from above.  Class 2 needs every condition to hold by the same margins the other way round, |a| >= 1e-8 (:157) included.
// tests/native/tri_screen.hip draws 2^30 ray / triangle pairs (rays through edges and vertices, grazing rays, slivers, far-away meshes): no triangle the
// exact test accepts is rejected, every class-2 triangle is accepted by the exact test at a distance below its bound.
__device__ __forceinline__ int tri_screen_f32(const uint4 u0, const uint4 u1, const uint4 u2, const Bvh8Ray &r, float dx, float dy, float dz, float &t_hi) {
    const float e1x = __uint_as_float(u1.x), e1y = __uint_as_float(u1.y), e1z = __uint_as_float(u1.z), Le = __uint_as_float(u1.w);
    const float e2x = __uint_as_float(u2.x), e2y = __uint_as_float(u2.y), e2z = __uint_as_float(u2.z);
    const float svx = r.ox - __uint_as_float(u0.x), svy = r.oy - __uint_as_float(u0.y), svz = r.oz - __uint_as_float(u0.z);
    const float hx = dy * e2z - dz * e2y, hy = dz * e2x - dx * e2z, hz = dx * e2y - dy * e2x;
    const float a = e1x * hx + e1y * hy + e1z * hz;
    const float nu = svx * hx + svy * hy + svz * hz;
    const float qx = svy * e1z - svz * e1y, qy = svz * e1x - svx * e1z, qz = svx * e1y - svy * e1x;
    const float nv = dx * qx + dy * qy + dz * qz;
    const float nt = e2x * qx + e2y * qy + e2z * qz;
    const float sm = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(svx), __builtin_fabsf(svy)), __builtin_fabsf(svz));
    const float EL = (1.9073486328125e-6f + 3.814697265625e-6f * sm) * Le;        // (32 + 64 sm) 2^-24 Le
    const float ET = EL * Le, EA = (3.814697265625e-6f * Le) * Le;                // 64 2^-24 Le^2
    const float aa = __builtin_fabsf(a);
    t_hi = 0.0f;
    if (!(aa > EA)) return 1;                                                      // (also a NaN from an overflow: the exact test decides)
    const bool neg = a < 0.0f;
    const float su = neg ? -nu : nu, sv_ = neg ? -nv : nv, st = neg ? -nt : nt;
    const float suv = su + sv_, EUV = 2.0f * EL + EA, tlo = r.tmin * aa, EM = ET + __builtin_fabsf(r.tmin) * EA;
    if (su < -EL || sv_ < -EL || suv > aa + EUV || st > r.best * aa + (ET + r.best * EA) || st < tlo - EM) return 0;
    // inside by the same margins, beyond t_min by them, the determinant clear of 1e-8: a hit for sure (if nothing closer is known)
    const float alo = aa - EA;
    if (su > EL && sv_ > EL && suv < aa - EUV && st > tlo + EM + 1.9073486328125e-6f * __builtin_fabsf(tlo) && alo > r.amin) {
        t_hi = (((st + ET) * 1.0000002384185791015625f) * __builtin_amdgcn_rcpf(alo * 0.99999988079071044921875f)) * 1.00000095367431640625f;
        return 2;
    }
    return 1;
}

// Float64 walks: the exact (scan's own, Float64) test of the newest candidate of this lane's list.  One memory round trip (the 96-byte record).
template <class T>
__device__ __forceinline__ void bvh8_resolve_one(const SceneLds<T> &sc, Bvh8Walk<T> &w, Bvh8Ray &r, Vec<T> o, Vec<T> d, T t_min, int base, T &closest, int &prim, uint32_t &slot) {
    const uint32_t ti = w.c0;
    w.c0 = w.c1; w.c1 = w.c2; w.c2 = w.c3; --w.nc;
    const Pack4<T> v0 = sc.bvh_tris[3 * (size_t)ti], e1 = sc.bvh_tris[3 * (size_t)ti + 1], e2 = sc.bvh_tris[3 * (size_t)ti + 2];
    T t;
    if (triangle_test<T>(v0, e1, e2, o, d, t_min, closest, t)) {
        const int p = base + (int)Bits<T>::to_u32(v0.w);
        if (t < closest || p > prim) {                                // t == closest: the later object wins
            closest = t; prim = p; slot = ti;
            r.best = __builtin_fminf(r.best, bvh8_best((float)((t - w.t0) * sc.bvh_root[2].w)));
        }
    }
}

// One trip of a lane's walk = ONE memory round trip: a lane with triangles pending tests the next one (render precision, caller's
// coordinates: the scan's own test), any other lane visits its nearest pending child node; both kinds of lane issue their loads
// together, through one per-lane pointer, before either computes.  (A node visit followed by a loop over its triangles made a wave
// pay one dependent round trip per triangle of its unluckiest lane: ~10 000 cycles per wave-step on config 5.)  A group that is
// exhausted pops the stack.  Returns false when the walk is over.  Stack levels < KL live in the wave's LDS scratch ([level][lane]).
template <class T, int KL>
__device__ __forceinline__ bool bvh8_step(const SceneLds<T> &sc, Bvh8Walk<T> &w, Bvh8Ray &r, Vec<T> o, Vec<T> d, T t_min, int base, T &closest, int &prim, uint32_t &slot,
                                          uint32_t *lds_stack, uint32_t *stack, uint32_t lane) {
    constexpr bool kWide = sizeof(T) == 8;
    // Float32: TWO items per trip — the lane's pending triangle AND, once that was the node's last pending one, its next node visit, both fetched in the same
    // round trip (the order of tests does not change the result: minimal t, ties to the later triangle).  Trips per ray 12.0 -> 10.4; with the 128 registers the
    // second launch of a mesh pass has in Float32 that is config 5 3.83 -> 3.69 ms.  (It lost 5 % while that kernel was held to 96 registers and spilled 60 of
    // them, and in Float64 — six more words to hold — it still gains nothing: 5.13 -> 5.20 ms; Float64 keeps one item per trip, below.)
#ifdef SPIRA_BVH_DUAL_F64
    constexpr bool kDual = true;
#else
    constexpr bool kDual = !kWide;
#endif
#ifdef SPIRA_BVH_SCREEN
    constexpr bool kScreen = kWide;          // experiment build (see tri_screen_f32): measured 2 .. 6 % SLOWER than testing every triangle in Float64 where it is met
#else
    constexpr bool kScreen = false;
#endif
    if constexpr (kScreen) {
        // Float64: the Float32 walk's trip (a node's last pending triangle AND the next node visit in one round trip, 5 + 3 loads), the triangle SCREENED in
        // Float32 (tri_screen_f32): a triangle that may be hit joins the lane's candidate list; its Float64 test is run by bvh8_resolve_one(), in batches.
        const bool do_tri = (w.tw >> 24) != 0;
        uint32_t ti = 0;
        const uint4 *pt = sc.bvh_tris32;
        if (do_tri) {
            const uint32_t s_ = (uint32_t)__builtin_ctz(w.tw >> 24);
            ti = (w.tw & 0x00FFFFFFu) + ((w.rank >> (4u * s_)) & 15u);
            w.tw &= ~(0x01000000u << s_);
            pt = sc.bvh_tris32 + 3 * (size_t)ti;
        }
        bool do_node = (w.tw >> 24) == 0;
        if (do_node && !(w.G & 0xFFu)) {
            if (w.sp == 0) do_node = false;
            else { --w.sp; w.G = (KL > 0 && w.sp < KL) ? lds_stack[w.sp * 64 + lane] : stack[w.sp - KL]; }
        }
        const uint4 *pn = sc.bvh_nodes;
        if (do_node) {
            const uint32_t pos = (uint32_t)__builtin_ctz(w.G);
            const uint32_t idx = (w.G >> 8) + (pos ^ r.oct);
            w.G &= w.G - 1u;
            if (w.G & 0xFFu) {
                if (KL > 0 && w.sp < KL) lds_stack[w.sp * 64 + lane] = w.G; else stack[w.sp - KL] = w.G;
                ++w.sp;
            }
            pn = sc.bvh_nodes + 5 * (size_t)idx;
        }
        const uint4 n0 = pn[0], n1 = pn[1], n2 = pn[2], n3 = pn[3], n4 = pn[4];
        const uint4 u0 = pt[0], u1 = pt[1], u2 = pt[2];
        asm volatile("" :: "v"(n0.x), "v"(n0.y), "v"(n0.z), "v"(n0.w), "v"(n1.x), "v"(n1.y), "v"(n1.z), "v"(n1.w), "v"(n2.x), "v"(n2.y), "v"(n2.z), "v"(n2.w) : "memory");
        asm volatile("" :: "v"(n3.x), "v"(n3.y), "v"(n3.z), "v"(n3.w), "v"(n4.x), "v"(n4.y), "v"(n4.z), "v"(n4.w) : "memory");
        asm volatile("" :: "v"(u0.x), "v"(u0.y), "v"(u0.z), "v"(u0.w), "v"(u1.x), "v"(u1.y), "v"(u1.z), "v"(u1.w), "v"(u2.x), "v"(u2.y), "v"(u2.z), "v"(u2.w) : "memory");
        if (do_tri) {
            float t_hi;
            const int cls = tri_screen_f32(u0, u1, u2, r, (float)d.x, (float)d.y, (float)d.z, t_hi);
            if (cls != 0) {
                // a full list is emptied on the spot (rare: a ray grazing many triangles): this lane's candidates all go through the exact test now
                if (__builtin_expect(w.nc == kBvhCand, 0)) { while (w.nc) bvh8_resolve_one<T>(sc, w, r, o, d, t_min, base, closest, prim, slot); }
                w.c3 = w.c2; w.c2 = w.c1; w.c1 = w.c0; w.c0 = ti; ++w.nc;
                if (cls == 2) r.best = __builtin_fminf(r.best, t_hi);
            }
        }
        if (do_node) {
            uint32_t ih, lh;
            bvh8_node(n0, n1, n2, n3, n4, r, ih, lh);
            w.G = (n1.x << 8) | ih;
            w.tw = n1.y | (lh << 24);
            w.rank = n1.z;
        }
        return (w.tw >> 24) != 0 || (w.G & 0xFFu) != 0 || w.sp > 0;
    }
    if constexpr (kDual) {
        const bool do_tri = (w.tw >> 24) != 0;
        uint32_t ti = 0;
        const uint4 *pt = reinterpret_cast<const uint4 *>(sc.bvh_tris);
        if (do_tri) {
            const uint32_t s_ = (uint32_t)__builtin_ctz(w.tw >> 24);
            ti = (w.tw & 0x00FFFFFFu) + ((w.rank >> (4u * s_)) & 15u);
            w.tw &= ~(0x01000000u << s_);
            pt = reinterpret_cast<const uint4 *>(sc.bvh_tris + 3 * (size_t)ti);
        }
        bool do_node = (w.tw >> 24) == 0;
        if (do_node && !(w.G & 0xFFu)) {
            if (w.sp == 0) do_node = false;
            else { --w.sp; w.G = (KL > 0 && w.sp < KL) ? lds_stack[w.sp * 64 + lane] : stack[w.sp - KL]; }
        }
        const uint4 *pn = sc.bvh_nodes;
        if (do_node) {
            const uint32_t pos = (uint32_t)__builtin_ctz(w.G);
            const uint32_t idx = (w.G >> 8) + (pos ^ r.oct);
            w.G &= w.G - 1u;
            if (w.G & 0xFFu) {
                if (KL > 0 && w.sp < KL) lds_stack[w.sp * 64 + lane] = w.G; else stack[w.sp - KL] = w.G;
                ++w.sp;
            }
            pn = sc.bvh_nodes + 5 * (size_t)idx;
        }
        const uint4 n0 = pn[0], n1 = pn[1], n2 = pn[2], n3 = pn[3], n4 = pn[4];
        const uint4 u0 = pt[0], u1 = pt[1], u2 = pt[2], uz = make_uint4(0, 0, 0, 0);
        uint4 u3 = uz, u4 = uz, u5 = uz;
        if constexpr (kWide) { u3 = pt[3]; u4 = pt[4]; u5 = pt[5]; }
        asm volatile("" :: "v"(n0.x), "v"(n0.y), "v"(n0.z), "v"(n0.w), "v"(n1.x), "v"(n1.y), "v"(n1.z), "v"(n1.w), "v"(n2.x), "v"(n2.y), "v"(n2.z), "v"(n2.w) : "memory");
        asm volatile("" :: "v"(n3.x), "v"(n3.y), "v"(n3.z), "v"(n3.w), "v"(n4.x), "v"(n4.y), "v"(n4.z), "v"(n4.w) : "memory");
        asm volatile("" :: "v"(u0.x), "v"(u0.y), "v"(u0.z), "v"(u0.w), "v"(u1.x), "v"(u1.y), "v"(u1.z), "v"(u1.w), "v"(u2.x), "v"(u2.y), "v"(u2.z), "v"(u2.w) : "memory");
        if constexpr (kWide) asm volatile("" :: "v"(u3.x), "v"(u3.y), "v"(u3.z), "v"(u3.w), "v"(u4.x), "v"(u4.y), "v"(u4.z), "v"(u4.w), "v"(u5.x), "v"(u5.y), "v"(u5.z), "v"(u5.w) : "memory");
        if (do_tri) {
            Pack4<T> v0, e1, e2;
            bvh8_tri_words(u0, u1, u2, u3, u4, u5, v0, e1, e2);
            T t;
            if (triangle_test<T>(v0, e1, e2, o, d, t_min, closest, t)) {
                const int p = base + (int)Bits<T>::to_u32(v0.w);
                if (t < closest || p > prim) {
                    closest = t; prim = p; slot = ti;
                    r.best = bvh8_best((float)((t - w.t0) * sc.bvh_root[2].w));
                }
            }
        }
        if (do_node) {
            uint32_t ih, lh;
            bvh8_node(n0, n1, n2, n3, n4, r, ih, lh);
            w.G = (n1.x << 8) | ih;
            w.tw = n1.y | (lh << 24);
            w.rank = n1.z;
        }
        return (w.tw >> 24) != 0 || (w.G & 0xFFu) != 0 || w.sp > 0;
    }
    const bool is_tri = (w.tw >> 24) != 0;
    const uint4 *ptr;
    uint32_t ti = 0;
    if (is_tri) {
        const uint32_t s_ = (uint32_t)__builtin_ctz(w.tw >> 24);           // the next hit leaf slot; its triangle: tri_base + rank of the slot
        ti = (w.tw & 0x00FFFFFFu) + ((w.rank >> (4u * s_)) & 15u);
        w.tw &= ~(0x01000000u << s_);
        ptr = reinterpret_cast<const uint4 *>(sc.bvh_tris + 3 * (size_t)ti);
    } else {
        const uint32_t pos = (uint32_t)__builtin_ctz(w.G);                 // (the low byte is never empty here)
        const uint32_t idx = (w.G >> 8) + (pos ^ r.oct);
        w.G &= w.G - 1u;
        if (w.G & 0xFFu) {                                                 // siblings left: keep the group
            if (KL > 0 && w.sp < KL) lds_stack[w.sp * 64 + lane] = w.G; else stack[w.sp - KL] = w.G;
            ++w.sp;
        }
        ptr = sc.bvh_nodes + 5 * (size_t)idx;
    }
    // Every lane loads the same number of words (a node is 5 x 16 bytes, a triangle 3 in Float32 and 6 in Float64; the arrays are padded
    // by one record): a wave-instruction costs the memory pipeline the same whichever lanes take part, and loads under a condition
    // would be sunk behind the other arm's arithmetic — a second round trip.  The empty asm pins all of them ahead of both arms.
    uint4 w0, w1, w2, w3, w4, w5 = make_uint4(0, 0, 0, 0);
    if constexpr (kWide) {
        // Float64: written out as the six aligned 16-byte loads they are.  Left to itself the compiler regroups the words by their later use
        // into loads at odd offsets (dwordx4 at +4, dwordx2 at +0, dwordx3 at +68 ...): -2.5 % on config 5.  (In Float32 its grouping is the
        // faster one by 5 %: register allocation.)
        typedef uint32_t U4 __attribute__((ext_vector_type(4)));
        U4 q0, q1, q2, q3, q4, q5;
        asm volatile("global_load_dwordx4 %0, %6, off\n\tglobal_load_dwordx4 %1, %6, off offset:16\n\tglobal_load_dwordx4 %2, %6, off offset:32\n\t"
                     "global_load_dwordx4 %3, %6, off offset:48\n\tglobal_load_dwordx4 %4, %6, off offset:64\n\tglobal_load_dwordx4 %5, %6, off offset:80\n\t"
                     "s_waitcnt vmcnt(0)"
                     : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3), "=&v"(q4), "=&v"(q5) : "v"(ptr) : "memory");
        w0 = make_uint4(q0.x, q0.y, q0.z, q0.w); w1 = make_uint4(q1.x, q1.y, q1.z, q1.w); w2 = make_uint4(q2.x, q2.y, q2.z, q2.w);
        w3 = make_uint4(q3.x, q3.y, q3.z, q3.w); w4 = make_uint4(q4.x, q4.y, q4.z, q4.w); w5 = make_uint4(q5.x, q5.y, q5.z, q5.w);
    } else {
        w0 = ptr[0]; w1 = ptr[1]; w2 = ptr[2]; w3 = ptr[3]; w4 = ptr[4];
        asm volatile("" :: "v"(w0.x), "v"(w0.y), "v"(w0.z), "v"(w0.w), "v"(w1.x), "v"(w1.y), "v"(w1.z), "v"(w1.w), "v"(w2.x), "v"(w2.y), "v"(w2.z), "v"(w2.w) : "memory");
        asm volatile("" :: "v"(w3.x), "v"(w3.y), "v"(w3.z), "v"(w3.w), "v"(w4.x), "v"(w4.y), "v"(w4.z), "v"(w4.w) : "memory");
    }
    if (is_tri) {
        Pack4<T> v0, e1, e2;
        bvh8_tri_words(w0, w1, w2, w3, w4, w5, v0, e1, e2);
        T t;
        if (triangle_test<T>(v0, e1, e2, o, d, t_min, closest, t)) {
            const int p = base + (int)Bits<T>::to_u32(v0.w);
            if (t < closest || p > prim) {                                // t == closest: the later object wins
                closest = t; prim = p; slot = ti;
                r.best = bvh8_best((float)((t - w.t0) * sc.bvh_root[2].w));
            }
        }
    } else {
        uint32_t ih, lh;
        bvh8_node(w0, w1, w2, w3, w4, r, ih, lh);
        w.G = (w1.x << 8) | ih;
        w.tw = w1.y | (lh << 24);
        w.rank = w1.z;
    }
    if ((w.tw >> 24) == 0 && !(w.G & 0xFFu)) {
        if (w.sp == 0) return false;
        --w.sp;
        w.G = (KL > 0 && w.sp < KL) ? lds_stack[w.sp * 64 + lane] : stack[w.sp - KL];
    }
    return true;
}

// The screened trip of a traversal SESSION's Float64 walk (k_path): the same trip as the kScreen branch of bvh8_step, but the lane carries nothing of its ray
// in Float64 — direction in Float32, origin = the entry point inside Bvh8Ray — because the Float64 tests of its candidates run outside the walk's loop,
// from the parked entry re-read there.  Returns 0: the walk is over; 1: go on; 2: the candidate list is full (nothing was done: the caller empties it).
enum : int { kWalkDone = 0, kWalkOn = 1, kWalkFull = 2 };
template <int KL>
__device__ __forceinline__ int bvh8_step_screened(const uint4 *nodes, const uint4 *tris32, Bvh8Walk<double> &w, Bvh8Ray &r, float dx, float dy, float dz,
                                                  uint32_t *lds_stack, uint32_t *stack, uint32_t lane) {
    if (__builtin_expect(w.nc == kBvhCand, 0)) return kWalkFull;
    const bool do_tri = (w.tw >> 24) != 0;
    uint32_t ti = 0;
    const uint4 *pt = tris32;
    if (do_tri) {
        const uint32_t s_ = (uint32_t)__builtin_ctz(w.tw >> 24);
        ti = (w.tw & 0x00FFFFFFu) + ((w.rank >> (4u * s_)) & 15u);
        w.tw &= ~(0x01000000u << s_);
        pt = tris32 + 3 * (size_t)ti;
    }
    bool do_node = (w.tw >> 24) == 0;
    if (do_node && !(w.G & 0xFFu)) {
        if (w.sp == 0) do_node = false;
        else { --w.sp; w.G = (KL > 0 && w.sp < KL) ? lds_stack[w.sp * 64 + lane] : stack[w.sp - KL]; }
    }
    const uint4 *pn = nodes;
    if (do_node) {
        const uint32_t pos = (uint32_t)__builtin_ctz(w.G);
        const uint32_t idx = (w.G >> 8) + (pos ^ r.oct);
        w.G &= w.G - 1u;
        if (w.G & 0xFFu) {
            if (KL > 0 && w.sp < KL) lds_stack[w.sp * 64 + lane] = w.G; else stack[w.sp - KL] = w.G;
            ++w.sp;
        }
        pn = nodes + 5 * (size_t)idx;
    }
    const uint4 n0 = pn[0], n1 = pn[1], n2 = pn[2], n3 = pn[3], n4 = pn[4];
    const uint4 u0 = pt[0], u1 = pt[1], u2 = pt[2];
    asm volatile("" :: "v"(n0.x), "v"(n0.y), "v"(n0.z), "v"(n0.w), "v"(n1.x), "v"(n1.y), "v"(n1.z), "v"(n1.w), "v"(n2.x), "v"(n2.y), "v"(n2.z), "v"(n2.w) : "memory");
    asm volatile("" :: "v"(n3.x), "v"(n3.y), "v"(n3.z), "v"(n3.w), "v"(n4.x), "v"(n4.y), "v"(n4.z), "v"(n4.w) : "memory");
    asm volatile("" :: "v"(u0.x), "v"(u0.y), "v"(u0.z), "v"(u0.w), "v"(u1.x), "v"(u1.y), "v"(u1.z), "v"(u1.w), "v"(u2.x), "v"(u2.y), "v"(u2.z), "v"(u2.w) : "memory");
    if (do_tri) {
        float t_hi;
        const int cls = tri_screen_f32(u0, u1, u2, r, dx, dy, dz, t_hi);
        if (cls != 0) {
            w.c3 = w.c2; w.c2 = w.c1; w.c1 = w.c0; w.c0 = ti; ++w.nc;
            if (cls == 2) r.best = __builtin_fminf(r.best, t_hi);
        }
    }
    if (do_node) {
        uint32_t ih, lh;
        bvh8_node(n0, n1, n2, n3, n4, r, ih, lh);
        w.G = (n1.x << 8) | ih;
        w.tw = n1.y | (lh << 24);
        w.rank = n1.z;
    }
    return ((w.tw >> 24) != 0 || (w.G & 0xFFu) != 0 || w.sp > 0) ? kWalkOn : kWalkDone;
}

// `closest` / `prim` come in holding the best hit so far (spheres, LDS triangles) and go out updated; `slot` is the hit triangle's
// position in the reordered array.
template <class T, int KL = 0>
__device__ __forceinline__ void bvh_closest_hit(const SceneLds<T> &sc, Vec<T> o, Vec<T> d, T t_min, T &closest, int &prim, uint32_t &slot, uint32_t *lds_stack = nullptr) {
    Bvh8Ray r; T t0;
    if (!bvh8_enter<T>(sc, o, d, closest, r, t0)) return;
    Bvh8Walk<T> w;
    bvh8_begin<T>(w, r, t0, t_min, sc.bvh_root[2].w);
    uint32_t stack[kBvhStackD - KL];
    const uint32_t lane = threadIdx.x & 63;
    const int base = (int)(sc.n_spheres + sc.n_triangles);
    while (bvh8_step<T, KL>(sc, w, r, o, d, t_min, base, closest, prim, slot, lds_stack, stack, lane)) {}
    while (w.nc) bvh8_resolve_one<T>(sc, w, r, o, d, t_min, base, closest, prim, slot);      // (Float64: the candidates of the screened walk; Float32: never any)
}

// Linear scan with a shrinking t_max, examples/julia-raytracer.jl:242-258; spheres :113-142, triangles
// :145-187 (LDS-resident), then the BVH for large meshes.  Returns the object index (spheres first, then
// triangles in the caller's order) or -1.
// The LDS-resident part of the scan (spheres, then the small triangle set); `closest` / `prim` come in as "nothing yet" and go out updated.
// TRI = false: an instantiation for scenes without LDS-resident triangles (spheres only, or spheres + a BVH mesh) — the triangle scan and
// the triangle arms of the shading code are not compiled in, which is worth 3 % on S1 in Float64 through register allocation alone.
// CAM = true (camera rays of k_path): every camera ray starts at the camera's origin, so `oc` (:114) and `dot(oc, oc) - radius^2` (:117) of a sphere are the
// same for all of them — `cam[s]` holds them {oc.x, oc.y, oc.z, c}, evaluated once per workgroup by the same operations on the same values (same bits),
// and the ray is left with half the arithmetic of a sphere test (9 of 18 operations).
template <class T, class P, bool TRI = true, bool CAM = false>
__device__ __forceinline__ void closest_hit_local(const SceneLds<T> &sc, Vec<T> o, Vec<T> d, T t_min, T &closest, int &prim, P &pol, const Pack4<T> *cam = nullptr) {
    closest = (T)INFINITY;                                     // t_max = Inf, :335
    prim = -1;
    const T a = dot(d, d);                                     // :115 (same value for every sphere)
    const T two_a = (T)2.0 * a;
    const T four_a = (T)4 * a;
    const RootDiv<T> over_2a = root_divisor<T>(two_a, pol);
    root_t_min<T>(t_min, pol);
    // software-pipelined LDS reads: the next sphere's packet is requested before this one is tested, so the
    // ds_read latency overlaps the arithmetic instead of stalling every iteration at s_waitcnt lgkmcnt(0)
    const Pack4<T> *sph = CAM ? cam : sc.sph;
    Pack4<T> c_next = sph[0];              // (slot 0 always exists in the LDS image: the block is never empty)
#pragma unroll 2       // two tests per trip: the packet rotation (c = c_next) turns into register renaming, +1..3 %
    for (uint32_t s = 0; s < sc.n_spheres; ++s) {
        const Pack4<T> c = c_next;
        c_next = sph[s + 1 < sc.n_spheres ? s + 1 : s];
        Vec<T> oc; T cc;
        if (CAM) { oc = mk<T>(c.x, c.y, c.z); cc = c.w; }
        else {
            oc = o - mk<T>(c.x, c.y, c.z);                     // :114
            cc = dot(oc, oc) - c.w;                            // :117
        }
        T b = (T)2.0 * dot(oc, d);                             // :116
        const T bb = b * b;
        T disc = bb - four_a * cc;                             // :118
        if (!(disc < 0)) {                                     // :120
            root_operands<T>(bb, disc, pol);
            T sq = root_sqrt<T>(disc, pol);                    // :125
            T root = root_over<T>(-b - sq, over_2a, pol);          // :126
            bool ok = !(root < t_min || root > closest);       // :130
            // :127,:131-132.  The second root is only worth a division when the first one fell short of t_min: if the first is
            // beyond `closest`, the second (-b + sq >= -b - sq, same positive divisor, correctly rounded division is monotone) is too.
            if (!ok && root < t_min) {
                root = root_over<T>(-b + sq, over_2a, pol);
                ok = !(root < t_min || root > closest);
            }
            if (ok) { closest = root; prim = (int)s; }         // :137, :252
        }
    }
    if (TRI && sc.n_triangles) {
        Pack4<T> v0n = sc.tri[0], e1n = sc.tri[1], e2n = sc.tri[2];
#pragma unroll 2       // as for the spheres: +3 % on S3 in f64, neutral elsewhere
        for (uint32_t i = 0; i < sc.n_triangles; ++i) {
            const Pack4<T> v0 = v0n, e1 = e1n, e2 = e2n;
            const uint32_t nx = i + 1 < sc.n_triangles ? i + 1 : i;
            v0n = sc.tri[3 * nx]; e1n = sc.tri[3 * nx + 1]; e2n = sc.tri[3 * nx + 2];
            T t;
            if (triangle_test<T>(v0, e1, e2, o, d, t_min, closest, t, pol)) {
                closest = t; prim = (int)(sc.n_spheres + i);
            }
        }
    }
}

// Does the ray come near the mesh at all (its LDS-resident bounding box, conservative slab test, within the closest hit so far)?
template <class T>
__device__ __forceinline__ bool mesh_box_hit(const SceneLds<T> &sc, Vec<T> o, Vec<T> d, T closest) {
    const Vec<T> inv = mk<T>(rcp_fast(d.x), rcp_fast(d.y), rcp_fast(d.z));
    return box_entry<T>(sc.bvh_root[0], sc.bvh_root[1], o, inv, closest) >= (T)0;
}

template <class T>
__device__ __forceinline__ void closest_hit_local(const SceneLds<T> &sc, Vec<T> o, Vec<T> d, T t_min, T &closest, int &prim) {
    ExactDiv exact;
    closest_hit_local<T>(sc, o, d, t_min, closest, prim, exact);
}
template <class T, bool BVH, class P, bool TRI = true>
__device__ __forceinline__ int closest_hit(const SceneLds<T> &sc, Vec<T> o, Vec<T> d, T t_min, T &t_hit, uint32_t &slot, P &pol) {
    T closest; int prim;
    closest_hit_local<T, P, TRI>(sc, o, d, t_min, closest, prim, pol);
    slot = 0;
    if (BVH) bvh_closest_hit<T>(sc, o, d, t_min, closest, prim, slot);
    t_hit = closest;
    return prim;
}
template <class T, bool BVH>
__device__ __forceinline__ int closest_hit(const SceneLds<T> &sc, Vec<T> o, Vec<T> d, T t_min, T &t_hit, uint32_t &slot) {
    ExactDiv exact;
    return closest_hit<T, BVH>(sc, o, d, t_min, t_hit, slot, exact);
}

// random_in_unit_sphere, examples/julia-raytracer.jl:309-316; tries t = 1..kMaxTries.
template <class T> __device__ __forceinline__ Vec<T> random_in_unit_sphere(const RngKey &k) {
    Vec<T> p = mk<T>(0, 0, 0);
    for (uint32_t t = 1; t <= kMaxTries; ++t) {
        T u0, u1, u2;
        rng3<T>(k, t, u0, u1, u2, (T)(1.0 / 1048576.0));            // 2*u: the doubling of :311 is exact, fold it into the scale
        Vec<T> q = mk<T>(u0, u1, u2) - mk<T>(1, 1, 1);              // :311
        if (dot(q, q) < (T)1.0) { p = q; break; }                   // :312
    }
    return p;
}

// ------------------------------------------------------------------ one path segment (semantics A)
// ray_color, examples/julia-raytracer.jl:328-367, in iterative form:
//   L += beta * emission ; beta *= specular*diffuse | 0.5*diffuse ; on a miss L += beta * sky.
// A segment is split in two so that the random vector it needs (which depends only on the RNG key, not
// on the geometry) can be produced by ANY lane of the workgroup between the halves:
//   segment_front : intersection, radiance terms, throughput, everything of the scatter that does not
//                   need the random vector (o becomes the hit point)
//   segment_back  : new direction from the pending vector and random_in_unit_sphere()
enum : uint32_t { kDead = 0, kDiffuse = 1, kSpecRough = 2, kMirror = 3, kDielectric = 4 };
template <class T> struct Pending { Vec<T> v; T rough; uint32_t kind; };   // v = pos + n (diffuse) | reflected (specular) | normal (dielectric)
struct SegInfo { int prim; bool alive; bool has_contrib; };

// ---- extensions (SPIRA_EXT_*; the reference only names them, README.md:10 — the build's own, parity unpinned) ----
// Compiled into the EXT = true instantiations only; the default kernels do not carry a single instruction of this.
//   SPIRA_EXT_DIELECTRIC  a material with roughness < 0 is a smooth dielectric of refractive index -roughness, tinted by
//                         `diffuse`: Snell refraction, Schlick's reflectance, total internal reflection; one uniform draw
//                         (try 1 of the bounce's key) chooses between reflection and refraction.
//   SPIRA_EXT_SPECTRAL    hero-wavelength transport: every path draws one wavelength (the third uniform of the pixel-jitter
//                         try, so no draw is added), RGB reflectances / emissions / the sky are uplifted to that wavelength
//                         with the basis rows of the SPD table, and the path starts with the wavelength's RGB response
//                         (rows w) as its throughput, so radiance accumulates as linear sRGB like in the RGB mode.
template <class T> struct ExtState {
    uint32_t flags;
    T bR, bG, bB;            // spectral: the uplift basis at this path's wavelength
};

template <class T> __device__ __forceinline__ T spd_lookup(const T *tab, int row, int i, T f) {
    const T *t = tab + row * kSpdN;
    return t[i] + (t[i + 1] - t[i]) * f;
}
// The path's wavelength coordinate x in [0, 35) (380 + 10 x nm) from its bounce-0 key; fills the basis, returns the RGB response.
template <class T>
__device__ __forceinline__ Vec<T> ext_wavelength(const SceneLds<T> &sc, uint32_t sA, uint32_t sB, uint32_t pixel, uint32_t sample, ExtState<T> &ext) {
    T xu, xv, u2;
    rng3<T>(rng_key(sA, sB, pixel, sample, 0), 0, xu, xv, u2);
    const T x = u2 * (T)(kSpdN - 1);
    int i = (int)x;
    i = i > kSpdN - 2 ? kSpdN - 2 : i;
    const T f = x - (T)i;
    ext.bR = spd_lookup<T>(sc.spd, 0, i, f); ext.bG = spd_lookup<T>(sc.spd, 1, i, f); ext.bB = spd_lookup<T>(sc.spd, 2, i, f);
    return mk<T>(spd_lookup<T>(sc.spd, 3, i, f), spd_lookup<T>(sc.spd, 4, i, f), spd_lookup<T>(sc.spd, 5, i, f));
}
template <class T> __device__ __forceinline__ T ext_uplift(const ExtState<T> &ext, const Vec<T> c) { return (c.x * ext.bR + c.y * ext.bG) + c.z * ext.bB; }

// sky term of a miss, :365-366
template <class T> __device__ __forceinline__ Vec<T> sky_term(const Vec<T> d, const Vec<T> beta) {
    T ts = (T)0.5 * (d.y + (T)1.0);
    Vec<T> sky = mk<T>(1.0, 1.0, 1.0) * ((T)1.0 - ts) + mk<T>((T)0.5, (T)0.7, (T)1.0) * ts;
    return mulv(beta, sky);
}
template <class T, bool EXT> __device__ __forceinline__ Vec<T> sky_term_x(const Vec<T> d, const Vec<T> beta, const ExtState<T> *ext) {
    if (EXT && (ext->flags & kExtSpectral)) {
        T ts = (T)0.5 * (d.y + (T)1.0);
        Vec<T> sky = mk<T>(1.0, 1.0, 1.0) * ((T)1.0 - ts) + mk<T>((T)0.5, (T)0.7, (T)1.0) * ts;
        return beta * ext_uplift<T>(*ext, sky);
    }
    return sky_term<T>(d, beta);
}

// material index (0-based) of a hit object
template <class T, bool BVH, bool TRI = true> __device__ __forceinline__ int material_of(const SceneLds<T> &sc, int prim, uint32_t slot) {
    if ((!TRI && !BVH) || prim < (int)sc.n_spheres) return sc.smat[prim];
    if (TRI && (!BVH || prim < (int)(sc.n_spheres + sc.n_triangles))) return sc.tmat[prim - (int)sc.n_spheres];
    return (int)Bits<T>::to_u32(sc.bvh_tris[3 * (size_t)slot + 1].w);
}

// Everything that follows a hit at `pos` (= o + d*t) except the random vector: normal, emitted term, and — when
// the path scatters — throughput and the pending direction data.  Returns whether the emitted term is non-zero.
template <class T, bool BVH, bool EXT, class P, bool TRI = true>
__device__ __forceinline__ bool shade_hit(const SceneLds<T> &sc, const Vec<T> pos, const Vec<T> d, int prim, uint32_t slot, Vec<T> &beta,
                                          bool scatter, Vec<T> &contrib, Pending<T> &pend, const ExtState<T> *ext, P &pol) {
    Vec<T> n;
    if ((!TRI && !BVH) || prim < (int)sc.n_spheres) {
        const Pack4<T> c = sc.sph[prim];
        n = normalize(pos - mk<T>(c.x, c.y, c.z), pol);                      // :139
    } else if (TRI && (!BVH || prim < (int)(sc.n_spheres + sc.n_triangles))) {
        int ti = prim - (int)sc.n_spheres;
        n = mk<T>(sc.tri[3 * ti].w, sc.tri[3 * ti + 1].w, sc.tri[3 * ti + 2].w);        // :105-109, precomputed by stage_scene
    } else {
        const Pack4<T> e1p = sc.bvh_tris[3 * (size_t)slot + 1], e2p = sc.bvh_tris[3 * (size_t)slot + 2];
        n = normalize(cross(mk<T>(e1p.x, e1p.y, e1p.z), mk<T>(e2p.x, e2p.y, e2p.z)));   // :105-109
    }
    const int mi = material_of<T, BVH, TRI>(sc, prim, slot);
    const Pack4<T> ma = sc.mat[2 * mi], mb = sc.mat[2 * mi + 1];
    Vec<T> diffuse = mk<T>(ma.x, ma.y, ma.z), emission = mk<T>(mb.x, mb.y, mb.z);
    T specular = ma.w, roughness = mb.w;
    if (EXT && (ext->flags & kExtSpectral)) {                                // the material at the path's wavelength
        const T sd = ext_uplift<T>(*ext, diffuse), se = ext_uplift<T>(*ext, emission);
        diffuse = mk<T>(sd, sd, sd); emission = mk<T>(se, se, se);
    }
    const bool has_contrib = (emission.x != 0 || emission.y != 0 || emission.z != 0);
    contrib = mulv(beta, emission);                                          // emitted, :339
    pend.kind = kDead; pend.rough = 0; pend.v = mk<T>(0, 0, 0);
    if (scatter) {
        if (EXT && (ext->flags & kExtDielectric) && roughness < (T)0.0) {
            pend.v = n; pend.kind = kDielectric; pend.rough = -roughness;    // direction: dielectric_resolve(), once the key is known
            beta = mulv(beta, diffuse);
        } else if (specular > (T)0.0) {                                      // :342
            pend.v = d - n * ((T)2 * dot(d, n));                             // reflect, :323-325, :344
            pend.kind = roughness > (T)0.0 ? kSpecRough : kMirror;           // :346
            pend.rough = roughness;
            beta = mulv(beta * specular, diffuse);                           // :353
        } else {
            pend.v = pos + n;                                                // first half of :356
            pend.kind = kDiffuse;
            beta = mulv(beta * (T)0.5, diffuse);                             // :360
        }
    }
    return has_contrib;
}

template <class T, bool BVH, bool EXT = false>
__device__ __forceinline__ bool shade_hit(const SceneLds<T> &sc, const Vec<T> pos, const Vec<T> d, int prim, uint32_t slot, Vec<T> &beta,
                                          bool scatter, Vec<T> &contrib, Pending<T> &pend, const ExtState<T> *ext = nullptr) {
    ExactDiv exact;
    return shade_hit<T, BVH, EXT>(sc, pos, d, prim, slot, beta, scatter, contrib, pend, ext, exact);
}

// SPIRA_EXT_DIELECTRIC: turn a pending dielectric interaction (pend.v = geometric normal, pend.rough = refractive index) into
// its outgoing direction (left in pend.v as a kMirror, i.e. "normalise and go").  u = first uniform of try 1 of the bounce's key.
template <class T>
__device__ __forceinline__ void dielectric_resolve(Pending<T> &pend, const Vec<T> d, const RngKey &key) {
    const Vec<T> n = pend.v;
    const T ior = pend.rough;
    const T cosd = dot(d, n);
    const bool entering = cosd < (T)0.0;
    const Vec<T> nn = entering ? n : mk<T>(-n.x, -n.y, -n.z);
    const T eta = entering ? (T)1.0 / ior : ior;
    const T ci = entering ? -cosd : cosd;
    const T s2 = (eta * eta) * ((T)1.0 - ci * ci);
    T r0 = ((T)1.0 - ior) / ((T)1.0 + ior); r0 = r0 * r0;
    const T x = (T)1.0 - ci, x2 = x * x;
    const T refl = r0 + ((T)1.0 - r0) * ((x2 * x2) * x);                    // Schlick
    T u, u1, u2;
    rng3<T>(key, 1, u, u1, u2);
    if (s2 > (T)1.0 || u < refl) pend.v = d - nn * ((T)2 * dot(d, nn));      // (total internal) reflection
    else pend.v = d * eta + nn * (eta * ci - sqrt_rn((T)1.0 - s2));         // Snell
    pend.kind = kMirror;
}

template <class T, bool BVH, bool EXT = false>
__device__ __forceinline__ SegInfo segment_front(const SceneLds<T> &sc, Vec<T> &o, const Vec<T> d, Vec<T> &beta, bool scatter,
                                                 Vec<T> &contrib, T &t_out, Pending<T> &pend, const ExtState<T> *ext = nullptr) {
    SegInfo info;
    T t;
    uint32_t slot;
    int prim = closest_hit<T, BVH>(sc, o, d, (T)0.001, t, slot);             // :335
    info.prim = prim;
    t_out = prim >= 0 ? t : (T)0;
    if (prim < 0) {                                                          // miss: sky, :365-366
        pend.kind = kDead; pend.rough = 0; pend.v = mk<T>(0, 0, 0);
        contrib = sky_term_x<T, EXT>(d, beta, ext);
        info.alive = false; info.has_contrib = true;
        return info;
    }
    const Vec<T> pos = o + d * t;                                            // point_at, :138 / :183
    info.has_contrib = shade_hit<T, BVH, EXT>(sc, pos, d, prim, slot, beta, scatter, contrib, pend, ext);
    info.alive = scatter;
    if (scatter) o = pos;
    return info;
}

template <class T, class P>
__device__ __forceinline__ Vec<T> segment_back(const Vec<T> pos, const Pending<T> &pend, const Vec<T> rnd, P &pol) {
    if (pend.kind == kDiffuse) return normalize((pend.v + rnd) - pos, pol);  // :356-357
    Vec<T> reflected = pend.v;
    if (pend.kind == kSpecRough) reflected = reflected + rnd * pend.rough;   // :347
    return normalize(reflected, pol);                                        // :349
}
template <class T>
__device__ __forceinline__ Vec<T> segment_back(const Vec<T> pos, const Pending<T> &pend, const Vec<T> rnd) {
    ExactDiv exact;
    return segment_back<T>(pos, pend, rnd, exact);
}

// Whole segment by one lane (megakernel / trace kernels).
template <class T, bool BVH, bool EXT = false>
__device__ __forceinline__ SegInfo trace_segment(const SceneLds<T> &sc, const RenderConst<T> &rc, Vec<T> &o, Vec<T> &d,
                                                 Vec<T> &beta, uint32_t pixel, uint32_t sample, uint32_t bounce,
                                                 bool scatter, Vec<T> &contrib, T &t_out, const ExtState<T> *ext = nullptr) {
    Pending<T> pend;
    const Vec<T> d_in = d;
    SegInfo info = segment_front<T, BVH, EXT>(sc, o, d, beta, scatter, contrib, t_out, pend, ext);
    if (info.alive) {
        Vec<T> rnd = mk<T>(0, 0, 0);
        if (EXT && pend.kind == kDielectric) dielectric_resolve<T>(pend, d_in, rng_key(rc.sA, rc.sB, pixel, sample, bounce));
        if (pend.kind == kDiffuse || pend.kind == kSpecRough)
            rnd = random_in_unit_sphere<T>(rng_key(rc.sA, rc.sB, pixel, sample, bounce));
        d = segment_back<T>(o, pend, rnd);
    }
    return info;
}

// Camera ray for (reference loop indices i, j; both 1-based) — examples/julia-raytracer.jl:398-400, :298-306
template <class T, class P>
__device__ __forceinline__ void camera_ray(const RenderConst<T> &rc, uint32_t i, uint32_t j, uint32_t pixel, uint32_t sample,
                                           Vec<T> &o, Vec<T> &d, P &pol) {
    T xu, xv, unused;
    rng3<T>(rng_key(rc.sA, rc.sB, pixel, sample, 0), 0, xu, xv, unused);
    T u = ((T)(i - 1) + xu) / (T)(rc.width - 1);                             // :398
    T v = ((T)(j - 1) + xv) / (T)(rc.height - 1);                            // :399
    o = rc.cam_origin;
    d = normalize(((rc.cam_llc + rc.cam_hor * u) + rc.cam_ver * v) - o, pol);   // :303
}
template <class T>
__device__ __forceinline__ void camera_ray(const RenderConst<T> &rc, uint32_t i, uint32_t j, uint32_t pixel, uint32_t sample,
                                           Vec<T> &o, Vec<T> &d) {
    ExactDiv exact;
    camera_ray<T>(rc, i, j, pixel, sample, o, d, exact);
}
// wave-uniform value -> scalar registers
__device__ __forceinline__ float to_scalar(float x) { return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(x))); }
__device__ __forceinline__ double to_scalar(double x) {
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)__double2loint(x)), hi = __builtin_amdgcn_readfirstlane((uint32_t)__double2hiint(x));
    return __hiloint2double((int)hi, (int)lo);
}
// The divisors of the pixel coordinates (:398-399), W - 1 and H - 1, with their refined reciprocals: once per wave, in scalar registers.
// The quotients need no window check: a numerator (i - 1) + xi is +0 or within 2^-21 .. 2^31, a divisor within 1 .. 2^31.
template <class T> struct PixelDiv { Recip<T> w1, h1; };
template <class T> __device__ __forceinline__ PixelDiv<T> pixel_divisors(const RenderConst<T> &rc) {
    PixelDiv<T> p;
    p.w1 = recip_of((T)(rc.width - 1)); p.h1 = recip_of((T)(rc.height - 1));
    p.w1.d = to_scalar(p.w1.d); p.w1.r = to_scalar(p.w1.r); p.h1.d = to_scalar(p.h1.d); p.h1.r = to_scalar(p.h1.r);
    return p;
}
template <class T> __device__ __forceinline__ T pixel_quotient(T n, const Recip<T> &rc, ExactDiv &) { return n / rc.d; }
template <class T> __device__ __forceinline__ T pixel_quotient(T n, const Recip<T> &rc, SpecDiv &) { return quotient(n, rc); }
// the same with the twelve camera values read from LDS (k_path: they would otherwise hold 12 / 24 SGPRs through the whole sub-chunk loop)
template <class T, class P>
__device__ __forceinline__ void camera_ray_lds(const RenderConst<T> &rc, const T *cam, const PixelDiv<T> &pd, uint32_t i, uint32_t j, uint32_t pixel, uint32_t sample,
                                               Vec<T> &o, Vec<T> &d, P &pol) {
    T xu, xv, unused;
    rng3<T>(rng_key(rc.sA, rc.sB, pixel, sample, 0), 0, xu, xv, unused);
    T u = pixel_quotient<T>((T)(i - 1) + xu, pd.w1, pol);                    // :398
    T v = pixel_quotient<T>((T)(j - 1) + xv, pd.h1, pol);                    // :399
    o = mk<T>(cam[0], cam[1], cam[2]);
    const Vec<T> llc = mk<T>(cam[3], cam[4], cam[5]), hor = mk<T>(cam[6], cam[7], cam[8]), ver = mk<T>(cam[9], cam[10], cam[11]);
    d = normalize(((llc + hor * u) + ver * v) - o, pol);                     // :303
}

// q (index inside the pass batch, slot-major) -> pixel / sample
template <class T>
__device__ __forceinline__ void path_of(const RenderConst<T> &rc, uint32_t q, uint32_t pass, uint32_t &i, uint32_t &j,
                                        uint32_t &pixel, uint32_t &sample) {
    uint32_t slot = fastdiv(q, rc.fd_tile);
    uint32_t pl = q - slot * rc.tile_pixels;
    uint32_t lr = fastdiv(pl, rc.fd_width);
    uint32_t lx = pl - lr * rc.width;
    j = ref_row_j(rc, lr);
    i = lx + 1;
    pixel = (j - 1) * rc.width + lx;
    sample = rc.sample0 + pass * rc.slots + slot;
}

// ------------------------------------------------------------------ ray queues (SoA of 16/32-byte packets)
// A[i] = {o.x o.y o.z d.x}  B[i] = {d.y d.z beta.x beta.y}  C[i] = {beta.z, bits(q)}   — 40 B (f32) / 80 B (f64) per ray
template <class T> struct RayQueue { Pack4<T> *A; Pack4<T> *B; Pack2<T> *C; };

struct Stats {                       // device-side counters (one per context)
    unsigned long long segments, rays_enqueued, radiance_rmw, radiance_store, redone_waves, rays_parked;
    unsigned long long mesh_wave_trips, mesh_lane_trips;      // traversal sessions: trips of the walk loop, and the lanes that took part in them (bvh8_step calls)
};

template <class T> struct BounceArgs {
    SceneGlobal<T> scene;
    RenderConst<T> rc;
    RayQueue<T> qin, qout;
    Pack3<T> *L;                     // per-path radiance of the pass batch (slot-major), 12 / 24 B each
    const uint32_t *cnt_in;          // [NW] rays waiting in each wave's region of qin (bounce >= 1)
    uint32_t *cnt_out;               // [NW] survivors this bounce leaves in each region of qout
    uint32_t *blk_stats;             // [NW][4] segments traced, radiance RMWs, radiance stores of this launch
    uint32_t cap;                    // region size in rays (a multiple of R*64)
    Stats *stats;
    uint32_t bounce;
    uint32_t pass;
    uint32_t n_first;                // FIRST: number of paths in this pass
};

// Order this wave's LDS traffic: DS operations of one wave execute in issue order, so a wave-scope
// fence (no instructions, only a compiler barrier) is all that lanes of ONE wave need to exchange data.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Per-bounce wavefront kernel.  FIRST generates the camera ray in registers (no queue read)
// and stores the path's first radiance term; later bounces read compacted rays, RMW the path
// radiance only when the segment contributes, and append survivors to the other queue.
//
// Every WAVE is an autonomous worker: the launch has NW = 4*gridDim.x waves, the same for every
// bounce of a pass, and wave w owns the fixed region [w*cap, (w+1)*cap) of BOTH queues (cap >= the rays
// w starts with; survivors only shrink).  It appends its survivors there and leaves their number in
// cnt_out[w]; the next bounce's wave w reads exactly that region.  Hence: no global atomic, no LDS
// atomic, no workgroup barrier after the scene is staged, no inter-wave traffic; the same CU slot / XCD
// (w mod 8 workgroups under round-robin dispatch) touches a region in consecutive bounces.  FIRST deals
// the pass's 64*R-ray sub-chunks round-robin over the waves, which spreads every image region over all
// waves, so the regions stay balanced as rays die.
//
// A wave handles sub-chunks of R*64 rays in three phases:
//   1. every lane: R x (load | generate ray, closest hit, radiance terms, throughput); rays that scatter
//      and need random_in_unit_sphere() append their RNG key to the wave's work list in LDS (slot index
//      from a ballot prefix; the list length is a wave-uniform register);
//   2. the wave drains the list cooperatively: a lane keeps trying one entry until it is accepted, then
//      takes the next unclaimed entry (ballot prefix again) — lanes do not idle behind the slowest
//      rejection loop of the wave (mean 1.9 tries, wave maximum ~6);
//   3. every lane: finish the directions; compaction by wave64 ballot + popcount prefix straight into
//      the wave's region; survivors land in consecutive slots, so the 16-byte packet stores coalesce.
// Results do not depend on which lane produced a random vector: it is a pure function of its key.
template <class T, bool FIRST, int R, bool BVH>
__global__ __launch_bounds__(kBlock, SPIRA_WAVES_PER_SIMD(T)) void k_bounce(const BounceArgs<T> a) {
    extern __shared__ __attribute__((aligned(32))) unsigned char lds_raw[];
    constexpr uint32_t WPB = kBlock / 64, SUB = 64 * R;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t NW = gridDim.x * WPB, wid = blockIdx.x * WPB + wave;
    const RenderConst<T> &rc = a.rc;
    const bool scatter = (a.bounce + 1 < rc.max_depth);
    // this wave's rays: FIRST -> sub-chunks wid, wid+NW, ... of [0, n_first); else its own queue region
    uint32_t n_mine = 0;
    if (!FIRST) {
        uint32_t n_blk = 0;
#pragma unroll
        for (uint32_t w = 0; w < WPB; ++w) n_blk += a.cnt_in[blockIdx.x * WPB + w];
        n_mine = a.cnt_in[wid];
        if (n_blk == 0) {                           // block-uniform: nothing left in this workgroup's regions
            if (lane == 0) { if (scatter) a.cnt_out[wid] = 0; a.blk_stats[4 * wid] = 0; a.blk_stats[4 * wid + 1] = 0; a.blk_stats[4 * wid + 2] = 0; a.blk_stats[4 * wid + 3] = 0; }
            return;
        }
    }
    const SceneLds<T> sc = stage_scene<T>(a.scene, lds_raw);     // the only workgroup barrier of the kernel
    // the wave's private work list sits behind the scene in the one dynamic LDS block
    Pack4<T> *s_rnd = reinterpret_cast<Pack4<T> *>(lds_raw + scene_lds_bytes<T>(a.scene.n_spheres, a.scene.n_materials, a.scene.n_triangles)) + wave * SUB;
    const uint32_t region = wid * a.cap;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    uint32_t n_rmw = 0, n_store = 0, n_seg = 0, fill = 0;

    const uint32_t limit = FIRST ? a.n_first : n_mine;
    const uint32_t n_sub = (limit + SUB - 1) / SUB;
    for (uint32_t sub = FIRST ? wid : 0u; sub < n_sub; sub += FIRST ? NW : 1u) {
        Vec<T> o[R], beta[R];
        Pending<T> pend[R];
        uint32_t q[R], ent[R];
        uint32_t n_list = 0;
        // ---------------- phase 1
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint32_t idx = sub * SUB + r * 64 + lane;
            pend[r].kind = kDead;
            bool want = false;
            RngKey key;
            if (idx < limit) {
                uint32_t pixel, sample, pi, pj;
                Vec<T> d;
                if (FIRST) {
                    q[r] = idx;
                    path_of<T>(rc, idx, a.pass, pi, pj, pixel, sample);
                    camera_ray<T>(rc, pi, pj, pixel, sample, o[r], d);
                    beta[r] = mk<T>(1, 1, 1);
                } else {
                    const Pack4<T> A = a.qin.A[region + idx], B = a.qin.B[region + idx];
                    const Pack2<T> C = a.qin.C[region + idx];
                    o[r] = mk<T>(A.x, A.y, A.z);
                    d = mk<T>(A.w, B.x, B.y);
                    beta[r] = mk<T>(B.z, B.w, C.x);
                    q[r] = Bits<T>::to_u32(C.y);
                }
                Vec<T> contrib; T t_hit;
                SegInfo si = segment_front<T, BVH>(sc, o[r], d, beta[r], scatter, contrib, t_hit, pend[r]);
                ++n_seg;
                // Path radiance L[q].  Bit 31 of the queued path index says "L[q] already holds radiance".  While it
                // is clear a term is a plain 16-byte STORE (0 + x == x exactly): the load -> add -> store round trip,
                // which parked every contributing wave for a full memory latency, only remains for paths that met
                // an emitter earlier.  Every path stores exactly once more at its end (its last term, or zero).
                // (No-return float atomics were tried instead: 30 % slower on S1 — scattered 4-byte atomics run at
                // the memory side at ~1/17 of the coalesced rate, MI355X_MICROARCH.md.)
                const uint32_t qi = q[r] & 0x7FFFFFFFu;
                const bool has_l = (q[r] >> 31) != 0;
                if (si.has_contrib) {
                    Pack3<T> l; l.x = contrib.x; l.y = contrib.y; l.z = contrib.z;
                    if (has_l) { const Pack3<T> l0 = a.L[qi]; l.x = l0.x + contrib.x; l.y = l0.y + contrib.y; l.z = l0.z + contrib.z; ++n_rmw; }
                    else ++n_store;
                    a.L[qi] = l;
                    q[r] |= 0x80000000u;
                } else if (!si.alive && !has_l) {              // path ends without ever having contributed
                    Pack3<T> l; l.x = 0; l.y = 0; l.z = 0;
                    a.L[qi] = l;
                    ++n_store;
                }
                want = (pend[r].kind == kDiffuse || pend[r].kind == kSpecRough);
                if (want) {
                    if (!FIRST) path_of<T>(rc, qi, a.pass, pi, pj, pixel, sample);
                    key = rng_key(rc.sA, rc.sB, pixel, sample, a.bounce);
                }
            }
            if (scatter) {
                const unsigned long long m = __ballot(want);
                if (want) {                                   // append to the wave's work list
                    ent[r] = n_list + __popcll(m & lt_mask);
                    uint32_t *kw = reinterpret_cast<uint32_t *>(&s_rnd[ent[r]]);
                    kw[0] = key.hA; kw[1] = key.hB;
                }
                n_list += (uint32_t)__popcll(m);
            }
        }
        if (!scatter) continue;        // uniform: the last bounce neither scatters nor enqueues
        wave_lds_sync();
        // ---------------- phase 2: cooperative random_in_unit_sphere() over the wave's work list
        {
            uint32_t e = lane, t = 1, next = 64;
            bool have = e < n_list;
            RngKey k; k.hA = 0; k.hB = 0; k.hBr = 0;
            if (have) {
                const uint32_t *kw = reinterpret_cast<const uint32_t *>(&s_rnd[e]);
                k.hA = kw[0]; k.hB = kw[1]; k.hBr = (k.hB << 16) | (k.hB >> 16);
            }
            while (__any(have)) {
                bool done = false;
                if (have) {
                    T u0, u1, u2;
                    rng3<T>(k, t, u0, u1, u2, (T)(1.0 / 1048576.0));
                    Vec<T> c = mk<T>(u0, u1, u2) - mk<T>(1, 1, 1);                   // :311
                    done = dot(c, c) < (T)1.0;                                      // :312
                    if (!done && t == kMaxTries) { c = mk<T>(0, 0, 0); done = true; }
                    if (done) { Pack4<T> w; w.x = c.x; w.y = c.y; w.z = c.z; w.w = 0; s_rnd[e] = w; }
                    ++t;
                }
                const unsigned long long m = __ballot(done);
                if (done) {                                   // take the next unclaimed entry
                    e = next + __popcll(m & lt_mask); t = 1;
                    have = e < n_list;
                    if (have) {
                        const uint32_t *kw = reinterpret_cast<const uint32_t *>(&s_rnd[e]);
                        k.hA = kw[0]; k.hB = kw[1]; k.hBr = (k.hB << 16) | (k.hB >> 16);
                    }
                }
                next += (uint32_t)__popcll(m);
            }
        }
        wave_lds_sync();
        // ---------------- phase 3: directions, compaction into this wave's region of the out queue
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const bool alive = pend[r].kind != kDead;
            const unsigned long long m = __ballot(alive);
            if (alive) {
                Vec<T> rnd = mk<T>(0, 0, 0);
                if (pend[r].kind != kMirror) { const Pack4<T> w = s_rnd[ent[r]]; rnd = mk<T>(w.x, w.y, w.z); }
                const Vec<T> nd = segment_back<T>(o[r], pend[r], rnd);
                const uint32_t dst = region + fill + __popcll(m & lt_mask);
                Pack4<T> A, B; Pack2<T> C;
                A.x = o[r].x; A.y = o[r].y; A.z = o[r].z; A.w = nd.x;
                B.x = nd.y; B.y = nd.z; B.z = beta[r].x; B.w = beta[r].y;
                C.x = beta[r].z; C.y = Bits<T>::from_u32(q[r]);
                a.qout.A[dst] = A; a.qout.B[dst] = B; a.qout.C[dst] = C;
            }
            fill += (uint32_t)__popcll(m);
        }
        wave_lds_sync();      // the list slots are rewritten by the next sub-chunk's phase 1
    }
    // per-wave results: survivors in the region, statistics (summed by k_resolve; no atomics)
    for (int sft = 32; sft > 0; sft >>= 1) { n_rmw += __shfl_down(n_rmw, sft); n_seg += __shfl_down(n_seg, sft); n_store += __shfl_down(n_store, sft); }
    if (lane == 0) {
        if (scatter) a.cnt_out[wid] = fill;
        a.blk_stats[4 * wid] = n_seg;
        a.blk_stats[4 * wid + 1] = n_rmw;
        a.blk_stats[4 * wid + 2] = n_store;
        a.blk_stats[4 * wid + 3] = FIRST ? 0u : n_seg;          // rays read from a queue == rays written to one
    }
}

// ====================================================================== the persistent wave-autonomous path kernel
// One launch per pass (mesh scenes: two — thin waves that park the rays reaching the mesh's box, then fat waves that walk them; see
// PathArgs::mesh_mode).  Differences from k_bounce (kept as SPIRA_KERNEL_BOUNCE, the round-1 organisation):
//   * the queues hold HITS, not rays: the intersection of segment k+1 runs at the END of stage k, right after the new
//     direction is known, so a ray that leaves the scene adds its sky term and never touches memory; what is queued is
//     {hit point, incoming direction, throughput, path index, hit reference}.  Every lane that enters the shading code of a
//     stage holds a hit (in k_bounce the later bounces ran their shading at ~50 % lane utilisation behind the hit/miss
//     branch), and on an open scene fewer than half as many packets go through HBM;
//   * a wave never waits for any other wave: it owns its region of both queues, so it simply carries on with the next
//     stage on what it has just written — all max_depth stages in one launch, the survivor count in a register, no count
//     arrays, no per-bounce launches and launch tails; the scene is staged into LDS once per workgroup per pass.
// Stage 0 generates camera rays (sub-chunks dealt round-robin over the waves) and intersects them; stage k >= 1 reads the
// hits of segment k.  Each stage: (1) shade the hit — emitted radiance, throughput, what the scatter needs — and put the
// RNG keys on the wave's LDS work list; (2) drain the list cooperatively (random_in_unit_sphere); (3) new direction, closest
// hit of the next segment, sky term on a miss, ballot/popcount compaction of the hits into the wave's region.
// The arithmetic per segment is the same statements in the same order as k_bounce / k_mega, so results are bit-identical.
template <class T> struct PathArgs {
    SceneGlobal<T> scene;
    RenderConst<T> rc;
    RayQueue<T> q[2];                // stage k writes q[k & 1] and (k >= 1) reads q[(k + 1) & 1]
    uint32_t *qref[2];               // Float32 only: the hit reference of each queued packet (Float64 packs it beside q)
    uint2 *qkey[2];                  // sphere scenes without extensions: the path's half-made RNG key beside each packet (k_path, kCarry)
    uint32_t cam_consts;             // the launch's LDS block has room for one packet per sphere behind the camera: the camera rays' share of a sphere test (closest_hit_local, CAM)
    Pack3<T> *L;                     // per-path radiance of the pass batch (slot-major)
    uint32_t *blk_stats;             // [NW][4] segments, radiance RMWs, radiance stores, packets enqueued
    uint32_t cap;                    // region size in packets (a multiple of R*64)
    uint32_t pass;
    uint32_t n_first;                // number of paths in this pass
    uint32_t dense_pct;              // dense continuation threshold in % (0 = always go through the queue)
    Pack4<T> *mesh_list;             // BVH scenes: per wave `cap` entries of 3 packets — rays that reach the mesh's bounding box wait here
                                     // for a traversal session (this wave's, or the fat wave's that takes the list over); NULL = traverse in place
    uint32_t mesh_min_batch;         // parked rays wait (over several rounds if need be) until the wave holds this many — or has nothing else to do
    uint32_t refill_free;            // a traversal session hands new rays to the free lanes once this many lanes are free
    // Mesh scenes run a pass as TWO launches (mesh_mode 1 then 2; 0 = one launch, sessions inside it).  Late in a pass a wave of the
    // first launch holds a handful of mesh rays at best — sessions of 40 rays ran at 0.22 lane utilisation — so the first launch (many
    // thin waves: all the work that never touches the mesh) only PARKS the rays that reach the mesh's box and lets their paths rest;
    // the second launch (few fat waves, wave w taking over the lists of first-launch waves w*k .. w*k+k-1) traverses them in large
    // refilled sessions and carries those paths to their end on its own queue regions.
    uint32_t mesh_mode;
    uint32_t *mesh_count;            // [NW of the first launch] parked rays per wave (written by mode 1, read by mode 2)
    uint32_t resume_k, resume_nw;    // first-launch waves per second-launch wave (<= 16, divides resume_nw), and their total number.  Mode 1 deals
                                     // its sub-chunks so that the k waves one fat wave takes over work on image blocks far apart (load balance)
    uint32_t *redo;                  // [NW] speculative division (SpecDiv above): the SPEC launch leaves 1 for a wave that has to be rendered again,
    uint32_t redo_only;              // the exact launch behind it (redo_only = 1) renders exactly those waves.  NULL / 0: one exact launch.
    Stats *stats;                    // redo_only: counts the waves rendered again
};

// queue word C.y: the path index (bit 31 = "L[q] already holds radiance") and, in Float64, the hit reference beside it
__device__ __forceinline__ float pack_qref(uint32_t q, uint32_t, float) { return __uint_as_float(q); }
__device__ __forceinline__ double pack_qref(uint32_t q, uint32_t ref, double) { return __longlong_as_double((long long)((unsigned long long)q | ((unsigned long long)ref << 32))); }
__device__ __forceinline__ uint32_t unpack_q(float w) { return __float_as_uint(w); }
__device__ __forceinline__ uint32_t unpack_q(double w) { return (uint32_t)(unsigned long long)__double_as_longlong(w); }
__device__ __forceinline__ uint32_t unpack_ref(double w) { return (uint32_t)((unsigned long long)__double_as_longlong(w) >> 32); }

#ifdef SPIRA_MESH_STATS
// experiment builds only (make stats): wave-summed traversal counters, read by spira_debug_mesh_stats()
// 0 wave cycles, 1 session cycles, 2 sessions, 3 wave-steps, 4 lane-steps, 5 refill blocks, 6 rays traversed, 7 rounds, 8 triangle tests (lane), 9 walk-loop cycles
__device__ unsigned long long g_mesh_dbg[32];        // [16..31]: the same for the second launch of a mesh pass (mesh_mode 2)
#define MESH_STAT(...) __VA_ARGS__
#else
#define MESH_STAT(...)
#endif

// MODE (mesh scenes, PathArgs::mesh_mode): 0 = one launch with the traversal sessions inside, 1 = first of two launches (parks, never walks: the
// session code is not compiled in — it cost the first launch 9 % through register allocation alone), 2 = second launch (no camera rays).
#ifndef SPIRA_CARRY_KEY
#define SPIRA_CARRY_KEY 1        // paths carry their half-made RNG key through the hit queue (k_path, kCarry); 0: derived from the path index at every scatter
#endif
#ifndef SPIRA_WAVES_B_F32
#define SPIRA_WAVES_B_F32 4      // the second launch of a mesh pass runs 16 fat waves per CU = 4 per SIMD whatever the kernel allows: 128 registers instead of 96
#endif                           // (35 spilled VGPRs less) is config 5 Float32 4.33 -> 4.11 ms.  (The parking launch with 4 / 3 waves per SIMD: +2 % / +4 % frame time.)
#ifndef SPIRA_WAVES_B_F64
#define SPIRA_WAVES_B_F64 SPIRA_WAVES_F64
#endif
#ifndef SPIRA_WAVES_PATH_F32
#define SPIRA_WAVES_PATH_F32 6   // the Float32 path kernels (k_path but its second mesh launch, k_path_metal) at 6 waves per SIMD = 80 registers: the instantiation with the LDS triangle
#endif                           // scan wants 90 at 5 (closed box S3: 19.6 -> 18.6 ms), the mesh parking launch 82 (config 5: 3.87 -> 3.81 ms; before the in-place traversal left it, it wanted
                                 // 96 + 35 spilled and lost 6 % at 6), S1's takes 75 either way, k_path_metal 47-53.  (k_bounce stays at SPIRA_WAVES_F32: it loses at 6.)
template <class T, int R, bool BVH, bool EXT, bool SPEC, int MODE = 0, bool TRI = true>
__global__ __launch_bounds__(kBlock, MODE == 2 ? (sizeof(T) == 8 ? SPIRA_WAVES_B_F64 : SPIRA_WAVES_B_F32) : (sizeof(T) == 4 ? SPIRA_WAVES_PATH_F32 : SPIRA_WAVES_F64)) void k_path(const PathArgs<T> a) {
    extern __shared__ __attribute__((aligned(32))) unsigned char lds_raw[];
    constexpr uint32_t WPB = kBlock / 64, SUB = 64 * R;
    constexpr bool kRefArray = sizeof(T) == 4;
    // The RNG key of a scatter is mix32(mix32(sA + pixel) ^ sb), mix32(mix32(sB ^ pixel) + sb) with sb = sample << 8 | bounce (rng_key).  With
    // bounce < 256 the `| bounce` is an XOR / an addition of its own, so a path can carry ka = mix32(sA + pixel) ^ (sample << 8) and
    // kb = mix32(sB ^ pixel) + (sample << 8) — two words made once from the camera ray's pixel and sample — and a scatter derives its key with two mixes
    // instead of two divisions (path index -> pixel, sample) and four.  Same bits; the mesh and extension instantiations keep the derivation from q
    // (their parked entries have no room for the words, and the extensions need pixel and sample anyway).
    constexpr bool kCarry = SPIRA_CARRY_KEY && !BVH && !EXT;
    constexpr uint32_t kStageShift = 25, kRefMask = (1u << kStageShift) - 1u;      // a hit reference needs < 2^25; the packet's stage rides above it
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t NW = gridDim.x * WPB, wid = blockIdx.x * WPB + wave;
    const RenderConst<T> &rc = a.rc;
    using Pol = typename std::conditional<SPEC, SpecDiv, ExactDiv>::type;
    Pol pol;                                                     // how this instantiation divides (see SpecDiv)
    if (!SPEC && a.redo_only) {                                  // the launch behind a speculative one: only what that one reported
        uint32_t any = 0;
        for (uint32_t w = 0; w < WPB; ++w) any |= a.redo[blockIdx.x * WPB + w];
        if (!any) return;                                        // (workgroup-uniform, ahead of the barrier in stage_scene)
    }
    T *cam_lds = reinterpret_cast<T *>(lds_raw + scene_lds_bytes<T>(a.scene.n_spheres, a.scene.n_materials, a.scene.n_triangles) + WPB * SUB * sizeof(Pack4<T>));
    if (threadIdx.x < 12) {                                      // camera: origin, lower-left corner, horizontal, vertical (visible after stage_scene's barrier)
        const Vec<T> *cv = threadIdx.x < 3 ? &rc.cam_origin : (threadIdx.x < 6 ? &rc.cam_llc : (threadIdx.x < 9 ? &rc.cam_hor : &rc.cam_ver));
        const uint32_t cc = threadIdx.x % 3;
        cam_lds[threadIdx.x] = cc == 0 ? cv->x : (cc == 1 ? cv->y : cv->z);
    }
    Pack4<T> *cam_sph = reinterpret_cast<Pack4<T> *>(reinterpret_cast<unsigned char *>(cam_lds) + 128);
    // (not in the extension instantiations: the second copy of the camera rays' scan costs the Float64 one 140 bytes of scratch per lane — glass scene 5.8 -> 6.6 ms)
    constexpr bool kCam = !EXT;
    if (kCam && MODE != 2 && a.cam_consts)                       // (the second launch of a mesh pass has no camera rays)
        for (uint32_t i = threadIdx.x; i < a.scene.n_spheres; i += blockDim.x) {
            const T *sp = a.scene.spheres5 + 5 * (size_t)i;
            const Vec<T> oc = rc.cam_origin - mk<T>(sp[0], sp[1], sp[2]);         // :114 with the origin every camera ray has
            Pack4<T> k; k.x = oc.x; k.y = oc.y; k.z = oc.z; k.w = dot(oc, oc) - sp[3] * sp[3];      // :117 (radius*radius as stage_scene makes it)
            cam_sph[i] = k;
        }
    const SceneLds<T> sc = stage_scene<T>(a.scene, lds_raw);     // the only workgroup barrier of the kernel
    const PixelDiv<T> pix_div = pixel_divisors<T>(rc);
    if (!SPEC && a.redo_only) {
        if (!a.redo[wid]) return;                                // wave-uniform; no workgroup barrier follows
        if (lane == 0) atomicAdd(&a.stats->redone_waves, 1ull);
    }
    Pack4<T> *s_rnd = reinterpret_cast<Pack4<T> *>(lds_raw + scene_lds_bytes<T>(a.scene.n_spheres, a.scene.n_materials, a.scene.n_triangles)) + wave * SUB;
    constexpr uint32_t mesh_mode = BVH ? (uint32_t)MODE : 0u;
    const uint32_t kseg = mesh_mode == 2u ? a.resume_k : 1u;
    if (mesh_mode == 2u && wid * kseg >= a.resume_nw) return;    // wave-uniform; no workgroup barrier follows
    const uint32_t region = wid * kseg * a.cap;                  // (a fat wave owns the regions of the k waves it takes over)
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const uint32_t ref_base = sc.n_spheres + sc.n_triangles;     // references >= ref_base are BVH triangle slots
    // Dense continuation (a.dense_pct > 0, max_depth <= 128; host default 70 %): when at least dense_pct % of a
    // sub-chunk's scattered rays hit again, they
    // stay in registers and go straight into their next stage instead of through the queue — compaction only where it pays (a closed
    // scene never touches the queues; an open one compacts as before).  Packets then carry their own stage (7 bits above the hit
    // reference), because a wave's region may hold hits of different segments.
    const bool mixed = mesh_mode != 0u || rc.max_depth <= 128u;      // (the two launches of a mesh pass exist for max_depth <= 128 only: a compile-time fact there,
    const uint32_t dense_pct = mixed ? a.dense_pct : 0u;
    // Deferred mesh traversal (BVH scenes, max_depth <= 128): a ray that reaches the mesh's bounding box is not traversed where it
    // stands — a few lanes of every wave would walk the tree in global memory while the others wait — but parked on the wave's
    // mesh list; sessions (below) walk the parked rays with all 64 lanes, refilled as rays finish; a hit then enters the out queue as a
    // packet of its own stage.  (On the 81 920-triangle scene of config 5 the in-place traversal was 70 % of the frame time.)
    const bool defer = mesh_mode != 0u || (BVH && mixed && a.mesh_list != nullptr);      //  and so is `defer`: neither launch carries the in-place traversal or the stage-less packets of deeper
                                                                                         //  renders — config 5: Float32 4.05 -> 3.88 ms, Float64 5.26 -> 5.11 ms, no spilled VGPR left in the Float64 kernels)
    Pack4<T> *mlist = a.mesh_list + 3 * (size_t)region;
    uint32_t n_rmw = 0, n_store = 0, n_seg = 0, n_enq = 0, n_park = 0;   // (n_park: entries written to the mesh list, wave-uniform)
    uint32_t n_wtrips = 0, n_ltrips = 0;                         // trips of the traversal sessions' walk loop and the lanes walking in them (wave-uniform: scalar registers)
    uint32_t n_in = 0;                                           // packets waiting in this wave's region (rounds >= 1)
    const uint32_t n_sub_first = (a.n_first + SUB - 1) / SUB;

    // park a ray on the wave's mesh list (entry = 3 packets: {o, d.x} {d.y, d.z, beta.xy} {beta.z, closest so far, q, object so far + stage of the hit})
    auto park = [&](uint32_t slot_i, const Vec<T> o_, const Vec<T> d_, const Vec<T> beta_, T closest_, uint32_t q_, int prim_, uint32_t stage_hit) {
        Pack4<T> A, B, C;
        A.x = o_.x; A.y = o_.y; A.z = o_.z; A.w = d_.x;
        B.x = d_.y; B.y = d_.z; B.z = beta_.x; B.w = beta_.y;
        C.x = beta_.z; C.y = closest_; C.z = Bits<T>::from_u32(q_); C.w = Bits<T>::from_u32((uint32_t)(prim_ + 1) | (stage_hit << kStageShift));
        mlist[3 * (size_t)slot_i] = A; mlist[3 * (size_t)slot_i + 1] = B; mlist[3 * (size_t)slot_i + 2] = C;
    };
    // Rounds: every packet of the wave's queue advances one stage per round; parked rays re-enter as hit packets after a traversal
    // session, possibly several rounds after they were parked (packets carry their own stage).  Ends when queue and list are empty.
    uint32_t mfill = 0;                                              // entries on the mesh list
    // mode 2: the list starts as the k lists the first launch left, segment j at its wave's region; pfx[j] = entries ahead of segment j
    constexpr int kMaxSeg = 16;
    uint32_t pfx[kMaxSeg + 1];
    bool segmented = false;
    {
        uint32_t cnt = 0;
        if (mesh_mode == 2u && lane < kseg && wid * kseg + lane < a.resume_nw) cnt = a.mesh_count[wid * kseg + lane];
        pfx[0] = 0;
#pragma unroll
        for (int j = 0; j < kMaxSeg; ++j) pfx[j + 1] = pfx[j] + (uint32_t)__builtin_amdgcn_readlane((int)cnt, j);
        if (mesh_mode == 2u) { mfill = pfx[kMaxSeg]; segmented = true; }
    }
    // Mode 1: wave w*k + i works on the sub-chunks of "logical" wave i*(NW/k) + w, so that the k waves a fat wave of the second launch takes over
    // (w*k .. w*k+k-1: contiguous regions) cover image blocks NW/k sub-chunks apart — adjacent blocks see the mesh together or not at all, and
    // fat waves made of them ran 1.5x apart.  (Any bijection will do: the assignment only has to be a pure function of the wave index.)
    const uint32_t first_sub = mesh_mode == 1u ? (wid % a.resume_k) * (NW / a.resume_k) + wid / a.resume_k : wid;
    MESH_STAT(unsigned long long dbg_t0 = __builtin_readcyclecounter(); unsigned long long dbg_sess = 0, dbg_walk = 0; uint32_t dbg_ns = 0, dbg_ws = 0, dbg_ls = 0, dbg_rf = 0, dbg_rays = 0, dbg_rounds = 0, dbg_ws1 = 0, dbg_ls1 = 0, dbg_h[4] = {0, 0, 0, 0};)
    for (uint32_t round = 0; ; ++round) {
        MESH_STAT(++dbg_rounds;)
        const bool first = mesh_mode != 2u && round == 0;
        const RayQueue<T> qin = a.q[(round + 1) & 1], qout = a.q[round & 1];
        const uint32_t *rin = a.qref[(round + 1) & 1];
        uint32_t *rout = a.qref[round & 1];
        const uint2 *kin = a.qkey[(round + 1) & 1];
        uint2 *kout = a.qkey[round & 1];
        const uint32_t limit = first ? a.n_first : n_in;
        const uint32_t n_sub = first ? n_sub_first : (n_in + SUB - 1) / SUB;
        uint32_t fill = 0;
        for (uint32_t sub = first ? first_sub : 0u; sub < n_sub; sub += first ? NW : 1u) {
            Vec<T> o[R], beta[R];
            Pending<T> pend[R];                           // between trips pend[r].v holds the direction the hit was reached along
            ExtState<T> ex[R];                            // EXT instantiations only (dead otherwise)
            uint32_t q[R], ent[R], stg[R], ref[R];
            uint32_t ka[R], kb[R];                        // kCarry: the path's half-made RNG key
            bool valid[R];
            // ---------------- the sub-chunk's rays: camera rays + their closest hit (round 0) or queued hits
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const uint32_t idx = sub * SUB + r * 64 + lane;
                valid[r] = false; stg[r] = round; ref[r] = 0; q[r] = 0; ka[r] = 0; kb[r] = 0;
                pend[r].kind = kDead; pend[r].rough = 0; pend[r].v = mk<T>(0, 0, 0);
                o[r] = mk<T>(0, 0, 0); beta[r] = mk<T>(1, 1, 1);
                bool parked = false; T park_t = 0; int park_prim = -1;
                if (idx < limit) {
                    if (first) {
                        uint32_t pixel, sample, pi, pj;
                        Vec<T> d;
                        q[r] = idx;
                        path_of<T>(rc, idx, a.pass, pi, pj, pixel, sample);
                        if (kCarry) { ka[r] = mix32(rc.sA + pixel) ^ (sample << 8); kb[r] = mix32(rc.sB ^ pixel) + (sample << 8); }
                        camera_ray_lds<T>(rc, cam_lds, pix_div, pi, pj, pixel, sample, o[r], d, pol);
                        if (EXT) { ex[r].flags = rc.flags; if (rc.flags & kExtSpectral) beta[r] = ext_wavelength<T>(sc, rc.sA, rc.sB, pixel, sample, ex[r]); }
                        T t; uint32_t slot = 0;
                        int prim;
                        if (defer) {
                            if (kCam && a.cam_consts) closest_hit_local<T, Pol, TRI, true>(sc, o[r], d, (T)0.001, t, prim, pol, cam_sph);
                            else closest_hit_local<T, Pol, TRI>(sc, o[r], d, (T)0.001, t, prim, pol);   // :335, spheres and LDS triangles
                            parked = mesh_box_hit<T>(sc, o[r], d, t);
                        } else if (kCam && !BVH && a.cam_consts) { closest_hit_local<T, Pol, TRI, true>(sc, o[r], d, (T)0.001, t, prim, pol, cam_sph); }
                        else prim = closest_hit<T, BVH, Pol, TRI>(sc, o[r], d, (T)0.001, t, slot, pol);     // :335
                        ++n_seg;
                        if (parked) { park_t = t; park_prim = prim; pend[r].v = d; }          // parked below, in uniform control flow
                        else if (prim < 0) {                      // the camera ray leaves the scene: sky, :365-366
                            const Vec<T> c = sky_term_x<T, EXT>(d, beta[r], &ex[r]);
                            Pack3<T> l; l.x = c.x; l.y = c.y; l.z = c.z;
                            a.L[idx] = l;
                            ++n_store;
                        } else {
                            valid[r] = true;
                            o[r] = o[r] + d * t;                  // point_at, :138 / :183
                            ref[r] = (BVH && prim >= (int)ref_base) ? ref_base + slot : (uint32_t)prim;
                            pend[r].v = d;
                        }
                    } else {
                        const Pack4<T> A = qin.A[region + idx], B = qin.B[region + idx];
                        const Pack2<T> C = qin.C[region + idx];
                        o[r] = mk<T>(A.x, A.y, A.z);              // the hit point
                        pend[r].v = mk<T>(A.w, B.x, B.y);         // the direction it was reached along
                        beta[r] = mk<T>(B.z, B.w, C.x);
                        q[r] = unpack_q(C.y);
                        uint32_t w;
                        if constexpr (kRefArray) w = rin[region + idx]; else w = unpack_ref(C.y);
                        if (mixed) { stg[r] = w >> kStageShift; ref[r] = w & kRefMask; } else ref[r] = w;
                        if (kCarry) { const uint2 k = kin[region + idx]; ka[r] = k.x; kb[r] = k.y; }
                        valid[r] = true;
                    }
                }
                if (defer && first) {                             // camera rays that reach the mesh's box: onto the mesh list (stage of the hit: 0)
                    const unsigned long long mp = __ballot(parked);
                    if (parked) park(mfill + __popcll(mp & lt_mask), o[r], pend[r].v, beta[r], park_t, q[r], park_prim, 0u);
                    mfill += (uint32_t)__popcll(mp); n_park += (uint32_t)__popcll(mp);
                }
            }
            // ---------------- stages on these rays; normally ONE trip, more while the hits stay dense
            while (true) {
                uint32_t n_list = 0, n_valid = 0;
                // ---- phase 1: the hit of segment stg[r]
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    bool want = false;
                    RngKey key;
                    const Vec<T> d = pend[r].v;
                    pend[r].kind = kDead;
                    if (valid[r]) {
                        uint32_t pixel = 0, sample = 0, pi, pj;
                        const bool scatter = stg[r] + 1 < rc.max_depth;
                        int prim = (int)ref[r]; uint32_t slot = 0;
                        if (BVH && ref[r] >= ref_base) { prim = (int)ref_base; slot = ref[r] - ref_base; }
                        const uint32_t qi = q[r] & 0x7FFFFFFFu;
                        const bool has_l = (q[r] >> 31) != 0;
                        if (EXT) {
                            path_of<T>(rc, qi, a.pass, pi, pj, pixel, sample);
                            ex[r].flags = rc.flags;
                            if (rc.flags & kExtSpectral) (void)ext_wavelength<T>(sc, rc.sA, rc.sB, pixel, sample, ex[r]);
                        }
                        Vec<T> contrib;
                        const bool has_contrib = shade_hit<T, BVH, EXT, Pol, TRI>(sc, o[r], d, prim, slot, beta[r], scatter, contrib, pend[r], &ex[r], pol);
                        // Path radiance L[q]: while bit 31 of q is clear a term is a plain store (0 + x == x exactly); the
                        // load -> add -> store round trip only remains for paths that met an emitter earlier.
                        if (has_contrib) {
                            Pack3<T> l; l.x = contrib.x; l.y = contrib.y; l.z = contrib.z;
                            if (has_l) { const Pack3<T> l0 = a.L[qi]; l.x = l0.x + contrib.x; l.y = l0.y + contrib.y; l.z = l0.z + contrib.z; ++n_rmw; }
                            else ++n_store;
                            a.L[qi] = l;
                            q[r] |= 0x80000000u;
                        } else if (!scatter && !has_l) {          // the path ends here without ever having contributed
                            Pack3<T> l; l.x = 0; l.y = 0; l.z = 0;
                            a.L[qi] = l;
                            ++n_store;
                        }
                        want = (pend[r].kind == kDiffuse || pend[r].kind == kSpecRough);
                        if (want || (EXT && pend[r].kind == kDielectric)) {
                            if (kCarry && mixed) {                // (max_depth <= 128: the bounce fits the low byte)
                                key.hA = mix32(ka[r] ^ stg[r]); key.hB = mix32(kb[r] + stg[r]); key.hBr = (key.hB << 16) | (key.hB >> 16);
                            } else {
                                if (!EXT) path_of<T>(rc, qi, a.pass, pi, pj, pixel, sample);
                                key = rng_key(rc.sA, rc.sB, pixel, sample, stg[r]);
                            }
                            if (EXT && pend[r].kind == kDielectric) dielectric_resolve<T>(pend[r], d, key);     // -> kMirror
                        }
                    }
                    const unsigned long long mv = __ballot(pend[r].kind != kDead);
                    n_valid += (uint32_t)__popcll(mv);
                    const unsigned long long m = __ballot(want);
                    if (want) {                                   // append to the wave's work list
                        ent[r] = n_list + __popcll(m & lt_mask);
                        uint32_t *kw = reinterpret_cast<uint32_t *>(&s_rnd[ent[r]]);
                        kw[0] = key.hA; kw[1] = key.hB;
                    }
                    n_list += (uint32_t)__popcll(m);
                }
                if (n_valid == 0) break;           // wave-uniform: nobody scatters (last segments, or an empty tail)
                wave_lds_sync();
                // ---- phase 2: cooperative random_in_unit_sphere() over the wave's work list
                {
                    uint32_t e = lane, t = 1, next = 64;
                    bool have = e < n_list;
                    RngKey k; k.hA = 0; k.hB = 0; k.hBr = 0;
                    if (have) {
                        const uint32_t *kw = reinterpret_cast<const uint32_t *>(&s_rnd[e]);
                        k.hA = kw[0]; k.hB = kw[1]; k.hBr = (k.hB << 16) | (k.hB >> 16);
                    }
                    while (__any(have)) {
                        bool done = false;
                        if (have) {
                            T u0, u1, u2;
                            rng3<T>(k, t, u0, u1, u2, (T)(1.0 / 1048576.0));
                            Vec<T> c = mk<T>(u0, u1, u2) - mk<T>(1, 1, 1);                   // :311
                            done = dot(c, c) < (T)1.0;                                      // :312
                            if (!done && t == kMaxTries) { c = mk<T>(0, 0, 0); done = true; }
                            if (done) { Pack4<T> w; w.x = c.x; w.y = c.y; w.z = c.z; w.w = 0; s_rnd[e] = w; }
                            ++t;
                        }
                        const unsigned long long m = __ballot(done);
                        if (done) {                                   // take the next unclaimed entry
                            e = next + __popcll(m & lt_mask); t = 1;
                            have = e < n_list;
                            if (have) {
                                const uint32_t *kw = reinterpret_cast<const uint32_t *>(&s_rnd[e]);
                                k.hA = kw[0]; k.hB = kw[1]; k.hBr = (k.hB << 16) | (k.hB >> 16);
                            }
                        }
                        next += (uint32_t)__popcll(m);
                    }
                }
                wave_lds_sync();
                // ---- phase 3: direction, closest hit of segment stg[r] + 1
                uint32_t n_hit = 0;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    bool hit = false, parked3 = false;
                    T park_t3 = 0; int park_prim3 = -1;
                    if (pend[r].kind != kDead) {
                        Vec<T> rnd = mk<T>(0, 0, 0);
                        if (pend[r].kind != kMirror) { const Pack4<T> w = s_rnd[ent[r]]; rnd = mk<T>(w.x, w.y, w.z); }
                        const Vec<T> nd = segment_back<T>(o[r], pend[r], rnd, pol);
                        T t; uint32_t slot = 0;
                        int prim;
                        if (defer) {
                            closest_hit_local<T, Pol, TRI>(sc, o[r], nd, (T)0.001, t, prim, pol);   // :335 of the next level, LDS part
                            parked3 = mesh_box_hit<T>(sc, o[r], nd, t);
                        } else prim = closest_hit<T, BVH, Pol, TRI>(sc, o[r], nd, (T)0.001, t, slot, pol); // :335 of the next level
                        ++n_seg;
                        if (parked3) { park_t3 = t; park_prim3 = prim; pend[r].v = nd; }
                        else if (prim < 0) {                          // the path leaves the scene: its last term, :365-366
                            const Vec<T> c = sky_term_x<T, EXT>(nd, beta[r], &ex[r]);
                            const uint32_t qi = q[r] & 0x7FFFFFFFu;
                            Pack3<T> l; l.x = c.x; l.y = c.y; l.z = c.z;
                            if (q[r] >> 31) { const Pack3<T> l0 = a.L[qi]; l.x = l0.x + c.x; l.y = l0.y + c.y; l.z = l0.z + c.z; ++n_rmw; }
                            else ++n_store;
                            a.L[qi] = l;
                        } else {
                            hit = true;
                            o[r] = o[r] + nd * t;                     // point_at, :138 / :183
                            ref[r] = (BVH && prim >= (int)ref_base) ? ref_base + slot : (uint32_t)prim;
                            pend[r].v = nd;
                            ++stg[r];
                        }
                    }
                    if (defer) {                                  // rays that reach the mesh's box wait for the end of the round
                        const unsigned long long mp = __ballot(parked3);
                        if (parked3) park(mfill + __popcll(mp & lt_mask), o[r], pend[r].v, beta[r], park_t3, q[r], park_prim3, stg[r] + 1u);
                        mfill += (uint32_t)__popcll(mp); n_park += (uint32_t)__popcll(mp);
                    }
                    valid[r] = hit;
                    n_hit += (uint32_t)__popcll(__ballot(hit));
                }
                wave_lds_sync();      // the list slots are rewritten by the next trip's / sub-chunk's phase 1
                if (n_hit == 0) break;
                if (dense_pct && n_hit * 100u >= n_valid * dense_pct) continue;     // dense: the hits stay in registers
                // ---- compaction of the hits into this wave's region of the out queue
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const unsigned long long m = __ballot(valid[r]);
                    if (valid[r]) {
                        const uint32_t dst = region + fill + __popcll(m & lt_mask);
                        const Vec<T> nd = pend[r].v;
                        const uint32_t w = mixed ? (ref[r] | (stg[r] << kStageShift)) : ref[r];
                        Pack4<T> A, B; Pack2<T> C;
                        A.x = o[r].x; A.y = o[r].y; A.z = o[r].z; A.w = nd.x;
                        B.x = nd.y; B.y = nd.z; B.z = beta[r].x; B.w = beta[r].y;
                        C.x = beta[r].z; C.y = pack_qref(q[r], w, (T)0);
                        qout.A[dst] = A; qout.B[dst] = B; qout.C[dst] = C;
                        if constexpr (kRefArray) rout[dst] = w;
                        if constexpr (kCarry) kout[dst] = make_uint2(ka[r], kb[r]);
                    }
                    fill += (uint32_t)__popcll(m);
                }
                break;
            }
        }
        if (defer && mesh_mode != 1u && mfill && (mfill >= a.mesh_min_batch || fill == 0)) {
            // ---------------- traversal session over the wave's parked rays.  A divergent node fetch costs the CU the same whether 64 lanes
            // take part or one (profiles/r03_gather_chase.txt), so the wave is kept full: whenever `refill_free` lanes have finished their
            // ray, their results are emitted (hits compacted into the out queue, sky terms for the others) and they take the next parked
            // rays; only the list's last rays run down to the slowest.  (The work list's LDS is idle now: the first stack levels live there.)
            constexpr int kLdsStack = (int)(SUB * sizeof(Pack4<T>) / (64 * sizeof(uint32_t)));      // 8 levels in Float32, 16 in Float64
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");     // the list was written by this wave's own lanes
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            uint32_t *lstack = reinterpret_cast<uint32_t *>(s_rnd);
            uint32_t stack[kBvhStackD - kLdsStack];
            uint32_t next = 0;                                         // next unread entry (wave-uniform)
            MESH_STAT(const unsigned long long dbg_s0 = __builtin_readcyclecounter(); ++dbg_ns; dbg_rays += mfill;)
#ifdef SPIRA_BVH_SCREEN
            constexpr bool kScreened = sizeof(T) == 8;
#else
            constexpr bool kScreened = false;
#endif
            if constexpr (kScreened) {
            // ---- Float64: the walk runs on Float32 screens (bvh8_step_screened) and a walking lane holds nothing of its ray in Float64: the parked entry is read
            // again when the lane has finished, its candidates go through the scan's own Float64 test there (all finished lanes together: one round trip per
            // candidate of the lane that holds the most), and the result is emitted as in the Float32 session below.
            bool walking = false, finished = false, full = false;      // finished: a result waits to be emitted; full: the candidate list has to be emptied
            uint32_t ent_i = 0;                                        // the lane's parked entry (index into a.mesh_list)
            float dxf = 0, dyf = 0, dzf = 1;
            Bvh8Ray ry; ry.ox = ry.oy = ry.oz = 0; ry.ix = ry.iy = ry.iz = 1; ry.oct = 0; ry.best = 0; ry.tmin = 0; ry.amin = 0;
            Bvh8Walk<T> wk; wk.G = 0; wk.tw = 0; wk.rank = 0; wk.sp = 0; wk.t0 = 0; wk.c0 = wk.c1 = wk.c2 = wk.c3 = 0; wk.nc = 0;
            auto decode = [&](const Pack4<T> A, const Pack4<T> B, const Pack4<T> C, Vec<T> &o_, Vec<T> &d_, Vec<T> &beta_, T &closest, uint32_t &q_, int &prim, uint32_t &slot, uint32_t &stage_hit) {
                o_ = mk<T>(A.x, A.y, A.z); d_ = mk<T>(A.w, B.x, B.y); beta_ = mk<T>(B.z, B.w, C.x);
                closest = C.y;
                q_ = Bits<T>::to_u32(C.z);
                const uint32_t pw = Bits<T>::to_u32(C.w);
                prim = (int)(pw & kRefMask) - 1;                       // the closest object so far: a sphere / LDS triangle, or (an entry written back below) a mesh triangle
                stage_hit = pw >> kStageShift;
                slot = 0;
                if (prim >= (int)ref_base) { slot = (uint32_t)prim - ref_base; prim = (int)ref_base; }
            };
            while (true) {
                // ---- lanes whose candidate list is full: their candidates go through the exact test now, the result so far is written back into the entry (rare)
                if (__any(full)) {
                    if (full) {
                        Pack4<T> *ent = a.mesh_list + 3 * (size_t)ent_i;
                        const Pack4<T> A = ent[0], B = ent[1];
                        Pack4<T> C = ent[2];
                        Vec<T> o_, d_, beta_; T closest; uint32_t q_, slot, stage_hit; int prim;
                        decode(A, B, C, o_, d_, beta_, closest, q_, prim, slot, stage_hit);
                        while (wk.nc) bvh8_resolve_one<T>(sc, wk, ry, o_, d_, (T)0.001, (int)ref_base, closest, prim, slot);
                        C.y = closest;
                        C.w = Bits<T>::from_u32((uint32_t)((prim >= (int)ref_base ? (int)(ref_base + slot) : prim) + 1) | (stage_hit << kStageShift));
                        ent[2] = C;
                        full = false;
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                }
                // ---- emit what is finished (wave-uniform block)
                {
                    Vec<T> o_ = mk<T>(0, 0, 0), d_ = mk<T>(0, 0, 1), beta_ = mk<T>(0, 0, 0);
                    T closest = 0; uint32_t q_ = 0, stage_hit = 0, slot = 0; int prim = -1;
                    if (finished) {
                        // the entry and the newest candidate's Float64 record in ONE round trip (most rays hold at most one candidate)
                        const Pack4<T> *ent = a.mesh_list + 3 * (size_t)ent_i;
                        const bool has = wk.nc != 0;
                        const uint32_t ti = has ? wk.c0 : 0u;
                        const Pack4<T> *rec = sc.bvh_tris + 3 * (size_t)ti;
                        const Pack4<T> A = ent[0], B = ent[1], C = ent[2];
                        const Pack4<T> v0 = rec[0], e1 = rec[1], e2 = rec[2];
                        decode(A, B, C, o_, d_, beta_, closest, q_, prim, slot, stage_hit);
                        if (has) {
                            wk.c0 = wk.c1; wk.c1 = wk.c2; wk.c2 = wk.c3; --wk.nc;
                            T t;
                            if (triangle_test<T>(v0, e1, e2, o_, d_, (T)0.001, closest, t)) {
                                const int p = (int)ref_base + (int)Bits<T>::to_u32(v0.w);
                                if (t < closest || p > prim) { closest = t; prim = p; slot = ti; }      // t == closest: the later object wins
                            }
                        }
                    }
                    while (__any(finished && wk.nc != 0)) {
                        if (finished && wk.nc != 0) bvh8_resolve_one<T>(sc, wk, ry, o_, d_, (T)0.001, (int)ref_base, closest, prim, slot);
                    }
                    const unsigned long long mh = __ballot(finished && prim >= 0);
                    if (finished) {
                        if (prim < 0) {                                   // the ray leaves the scene after all: sky, :365-366
                            const uint32_t qi = q_ & 0x7FFFFFFFu;
                            ExtState<T> ex1; ex1.flags = rc.flags; ex1.bR = 0; ex1.bG = 0; ex1.bB = 0;
                            if (EXT && (rc.flags & kExtSpectral)) {
                                uint32_t pi, pj, pixel, sample;
                                path_of<T>(rc, qi, a.pass, pi, pj, pixel, sample);
                                (void)ext_wavelength<T>(sc, rc.sA, rc.sB, pixel, sample, ex1);
                            }
                            const Vec<T> c = sky_term_x<T, EXT>(d_, beta_, &ex1);
                            Pack3<T> l; l.x = c.x; l.y = c.y; l.z = c.z;
                            if (q_ >> 31) { const Pack3<T> l0 = a.L[qi]; l.x = l0.x + c.x; l.y = l0.y + c.y; l.z = l0.z + c.z; ++n_rmw; }
                            else ++n_store;
                            a.L[qi] = l;
                        } else {
                            const Vec<T> hp = o_ + d_ * closest;          // point_at, :138 / :183
                            const uint32_t w_ = ((prim >= (int)ref_base) ? ref_base + slot : (uint32_t)prim) | (stage_hit << kStageShift);
                            const uint32_t dst = region + fill + __popcll(mh & lt_mask);
                            Pack4<T> A, B; Pack2<T> C;
                            A.x = hp.x; A.y = hp.y; A.z = hp.z; A.w = d_.x;
                            B.x = d_.y; B.y = d_.z; B.z = beta_.x; B.w = beta_.y;
                            C.x = beta_.z; C.y = pack_qref(q_, w_, (T)0);
                            qout.A[dst] = A; qout.B[dst] = B; qout.C[dst] = C;
                            if constexpr (kRefArray) rout[dst] = w_;
                        }
                    }
                    fill += (uint32_t)__popcll(mh);
                    finished = false;
                }
                // ---- refill the free lanes
                const unsigned long long mf = __ballot(!walking);
                const uint32_t take = min((uint32_t)__popcll(mf), mfill - next);
                if (!walking && (uint32_t)__popcll(mf & lt_mask) < take) {
                    const uint32_t e = next + (uint32_t)__popcll(mf & lt_mask);
                    ent_i = region + e;
                    if (segmented) {                               // entry e of the k lists taken over: which list, which entry of it
                        uint32_t seg = 0, off = e;
#pragma unroll
                        for (int j = 1; j < kMaxSeg; ++j) if (e >= pfx[j]) { seg = (uint32_t)j; off = e - pfx[j]; }
                        ent_i = (wid * kseg + seg) * a.cap + off;
                    }
                    const Pack4<T> *ent = a.mesh_list + 3 * (size_t)ent_i;
                    const Pack4<T> A = ent[0], B = ent[1], C = ent[2];
                    const Vec<T> o_ = mk<T>(A.x, A.y, A.z), d_ = mk<T>(A.w, B.x, B.y);
                    T t0;
                    if (bvh8_enter<T>(sc, o_, d_, C.y, ry, t0)) { bvh8_begin<T>(wk, ry, t0, (T)0.001, sc.bvh_root[2].w); dxf = (float)d_.x; dyf = (float)d_.y; dzf = (float)d_.z; walking = true; }
                    else { wk.nc = 0; finished = true; }
                }
                next += take;
                if (!__any(walking)) { if (__any(finished)) continue; break; }
                // ---- walk; leave the loop when enough lanes are free for a refill to pay (or, with the list exhausted, when all are done), or a list is full
                const bool more = next < mfill;
                uint32_t n_now = (uint32_t)__popcll(__ballot(walking));
                while (true) {
                    ++n_wtrips; n_ltrips += n_now;
                    if (walking) {
                        const int st_ = bvh8_step_screened<kLdsStack>(sc.bvh_nodes, sc.bvh_tris32, wk, ry, dxf, dyf, dzf, lstack, stack, lane);
                        if (st_ == kWalkDone) { walking = false; finished = true; }
                        else if (st_ == kWalkFull) full = true;
                    }
                    const uint32_t n_walk = (uint32_t)__popcll(__ballot(walking));
                    n_now = n_walk;
                    if (n_walk == 0 || (more && 64u - n_walk >= a.refill_free) || __any(full)) break;
                }
            }
            } else {
            bool walking = false, finished = false;                    // finished: a result waits to be emitted
            Vec<T> o_ = mk<T>(0, 0, 0), d_ = mk<T>(0, 0, 1), beta_ = mk<T>(0, 0, 0);
            T closest = 0; uint32_t q_ = 0, stage_hit = 0, slot = 0; int prim = -1;
            Bvh8Ray ry; ry.ox = ry.oy = ry.oz = 0; ry.ix = ry.iy = ry.iz = 1; ry.oct = 0; ry.best = 0; ry.tmin = 0; ry.amin = 0;
            Bvh8Walk<T> wk; wk.G = 0; wk.tw = 0; wk.rank = 0; wk.sp = 0; wk.t0 = 0; wk.c0 = wk.c1 = wk.c2 = wk.c3 = 0; wk.nc = 0;
            while (true) {
                // ---- emit what is finished, refill the free lanes (wave-uniform block)
                const unsigned long long mh = __ballot(finished && prim >= 0);
                if (finished) {
                    if (prim < 0) {                                   // the ray leaves the scene after all: sky, :365-366
                        const uint32_t qi = q_ & 0x7FFFFFFFu;
                        ExtState<T> ex1; ex1.flags = rc.flags; ex1.bR = 0; ex1.bG = 0; ex1.bB = 0;
                        if (EXT && (rc.flags & kExtSpectral)) {
                            uint32_t pi, pj, pixel, sample;
                            path_of<T>(rc, qi, a.pass, pi, pj, pixel, sample);
                            (void)ext_wavelength<T>(sc, rc.sA, rc.sB, pixel, sample, ex1);
                        }
                        const Vec<T> c = sky_term_x<T, EXT>(d_, beta_, &ex1);
                        Pack3<T> l; l.x = c.x; l.y = c.y; l.z = c.z;
                        if (q_ >> 31) { const Pack3<T> l0 = a.L[qi]; l.x = l0.x + c.x; l.y = l0.y + c.y; l.z = l0.z + c.z; ++n_rmw; }
                        else ++n_store;
                        a.L[qi] = l;
                    } else {
                        const Vec<T> hp = o_ + d_ * closest;          // point_at, :138 / :183
                        const uint32_t w_ = ((prim >= (int)ref_base) ? ref_base + slot : (uint32_t)prim) | (stage_hit << kStageShift);
                        const uint32_t dst = region + fill + __popcll(mh & lt_mask);
                        Pack4<T> A, B; Pack2<T> C;
                        A.x = hp.x; A.y = hp.y; A.z = hp.z; A.w = d_.x;
                        B.x = d_.y; B.y = d_.z; B.z = beta_.x; B.w = beta_.y;
                        C.x = beta_.z; C.y = pack_qref(q_, w_, (T)0);
                        qout.A[dst] = A; qout.B[dst] = B; qout.C[dst] = C;
                        if constexpr (kRefArray) rout[dst] = w_;
                    }
                }
                fill += (uint32_t)__popcll(mh);
                finished = false;
                const unsigned long long mf = __ballot(!walking);
                const uint32_t take = min((uint32_t)__popcll(mf), mfill - next);
                if (!walking && (uint32_t)__popcll(mf & lt_mask) < take) {
                    const uint32_t e = next + (uint32_t)__popcll(mf & lt_mask);
                    const Pack4<T> *ent = mlist + 3 * (size_t)e;
                    if (segmented) {                               // entry e of the k lists taken over: which list, which entry of it
                        uint32_t seg = 0, off = e;
#pragma unroll
                        for (int j = 1; j < kMaxSeg; ++j) if (e >= pfx[j]) { seg = (uint32_t)j; off = e - pfx[j]; }
                        ent = a.mesh_list + 3 * ((size_t)(wid * kseg + seg) * a.cap + off);
                    }
                    const Pack4<T> A = ent[0], B = ent[1], C = ent[2];
                    o_ = mk<T>(A.x, A.y, A.z); d_ = mk<T>(A.w, B.x, B.y); beta_ = mk<T>(B.z, B.w, C.x);
                    closest = C.y;
                    q_ = Bits<T>::to_u32(C.z);
                    const uint32_t pw = Bits<T>::to_u32(C.w);
                    prim = (int)(pw & kRefMask) - 1;
                    stage_hit = pw >> kStageShift;
                    slot = 0;
                    T t0;
                    if (bvh8_enter<T>(sc, o_, d_, closest, ry, t0)) { bvh8_begin<T>(wk, ry, t0, (T)0.001, sc.bvh_root[2].w); walking = true; }
                    else finished = true;
                }
                next += take;
                if (!__any(walking)) { if (__any(finished)) continue; break; }
                // ---- walk; leave the loop when enough lanes are free for a refill to pay (or, with the list exhausted, when all are done)
                const bool more = next < mfill;
                MESH_STAT(++dbg_rf; const unsigned long long dbg_w0 = __builtin_readcyclecounter();)
                uint32_t n_now = (uint32_t)__popcll(__ballot(walking));
                while (true) {
                    ++n_wtrips; n_ltrips += n_now;
                    MESH_STAT(++dbg_ws; { const uint32_t nl = (uint32_t)__popcll(__ballot(walking)); dbg_ls += nl; if (segmented) { ++dbg_ws1; dbg_ls1 += nl; } ++dbg_h[nl <= 8 ? 0 : (nl <= 24 ? 1 : (nl <= 48 ? 2 : 3))]; })
                    if (walking && !bvh8_step<T, kLdsStack>(sc, wk, ry, o_, d_, (T)0.001, (int)ref_base, closest, prim, slot, lstack, stack, lane)) { walking = false; finished = true; }
                    const uint32_t n_walk = (uint32_t)__popcll(__ballot(walking));
                    n_now = n_walk;
                    if (n_walk == 0 || (more && 64u - n_walk >= a.refill_free)) break;
                }

                MESH_STAT(dbg_walk += __builtin_readcyclecounter() - dbg_w0;)
            }
            }
            mfill = 0;
            segmented = false;                                         // from here on the list is the wave's own (the lists taken over are spent)
            MESH_STAT(dbg_sess += __builtin_readcyclecounter() - dbg_s0;)
        }
        n_in = fill;
        n_enq += fill;
        if (n_in == 0 && (mfill == 0 || mesh_mode == 1u)) break;      // wave-uniform: every path of this wave has ended (mode 1: or rests on the list)
        // The wave now reads what its own lanes have just written: on one CU (one vector L1, one L2) a workgroup-scope
        // release/acquire (wait for the stores) is all that takes.  No other wave ever touches this region.
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    for (int sft = 32; sft > 0; sft >>= 1) { n_rmw += __shfl_down(n_rmw, sft); n_seg += __shfl_down(n_seg, sft); n_store += __shfl_down(n_store, sft); }
    const bool again = SPEC && __any(outside_window<T>(pol));    // some quotient of this wave may not be the IEEE one: the exact launch redoes the wave
    MESH_STAT(if (lane == 0) { unsigned long long *g = g_mesh_dbg + (mesh_mode == 2u ? 16 : 0);
                               atomicAdd(&g[0], __builtin_readcyclecounter() - dbg_t0); atomicAdd(&g[1], dbg_sess); atomicAdd(&g[2], (unsigned long long)dbg_ns);
                               atomicAdd(&g[3], (unsigned long long)dbg_ws); atomicAdd(&g[4], (unsigned long long)dbg_ls); atomicAdd(&g[5], (unsigned long long)dbg_rf);
                               atomicAdd(&g[6], (unsigned long long)dbg_rays); atomicAdd(&g[7], (unsigned long long)dbg_rounds); atomicAdd(&g[9], dbg_walk); atomicAdd(&g[10], 1ull); atomicAdd(&g[11], (unsigned long long)dbg_ws1); atomicAdd(&g[12], (unsigned long long)dbg_ls1);
                               atomicAdd(&g[8], (unsigned long long)dbg_h[0]); atomicAdd(&g[13], (unsigned long long)dbg_h[1]); atomicAdd(&g[14], (unsigned long long)dbg_h[2]); atomicAdd(&g[15], (unsigned long long)dbg_h[3]); })
    if (lane == 0) {
        if (mesh_mode == 2u) {                     // on top of what the first launch's wave left in the row
            const uint32_t row = 4 * wid * kseg;
            a.blk_stats[row] += n_seg; a.blk_stats[row + 1] += n_rmw; a.blk_stats[row + 2] += n_store; a.blk_stats[row + 3] += n_enq;
        } else {
            a.blk_stats[4 * wid] = n_seg;
            a.blk_stats[4 * wid + 1] = n_rmw;
            a.blk_stats[4 * wid + 2] = n_store;
            a.blk_stats[4 * wid + 3] = n_enq;          // wave-uniform
            if (mesh_mode == 1u) a.mesh_count[wid] = mfill;
        }
        if (SPEC) a.redo[wid] = again ? 1u : 0u;
        if (BVH && n_park && !again && !(SPEC && a.redo_only == 2u)) atomicAdd(&a.stats->rays_parked, (unsigned long long)n_park);      // (a wave rendered again counts there)
        if (BVH && mesh_mode != 1u && n_wtrips && !again && !(SPEC && a.redo_only == 2u)) { atomicAdd(&a.stats->mesh_wave_trips, (unsigned long long)n_wtrips); atomicAdd(&a.stats->mesh_lane_trips, (unsigned long long)n_ltrips); }
    }
}

// Megakernel: one lane walks whole paths in registers (the non-wavefront comparison point) — with path REGENERATION: a lane
// whose path has ended starts its next path at once instead of idling until the longest path of the wave is done, so every
// trip of the loop intersects one ray per lane (open scenes: 2.2 segments per path on average, 8 at most).
template <class T, bool BVH, bool EXT>
__global__ __launch_bounds__(kBlock) void k_mega(const BounceArgs<T> a) {
    extern __shared__ __attribute__((aligned(32))) unsigned char lds_raw[];
    const SceneLds<T> sc = stage_scene<T>(a.scene, lds_raw);
    const RenderConst<T> &rc = a.rc;
    unsigned long long nseg = 0;
    const uint32_t stride = gridDim.x * kBlock;
    uint32_t idx = blockIdx.x * kBlock + threadIdx.x;
    bool fresh = true;
    uint32_t pixel = 0, sample = 0, b = 0;
    Vec<T> o = mk<T>(0, 0, 0), d = mk<T>(0, 0, 1), beta = mk<T>(1, 1, 1), Lacc = mk<T>(0, 0, 0);
    ExtState<T> ex; ex.flags = rc.flags; ex.bR = 0; ex.bG = 0; ex.bB = 0;
    while (idx < a.n_first) {
        if (fresh) {
            uint32_t pi, pj;
            path_of<T>(rc, idx, a.pass, pi, pj, pixel, sample);
            camera_ray<T>(rc, pi, pj, pixel, sample, o, d);
            beta = mk<T>(1, 1, 1); Lacc = mk<T>(0, 0, 0); b = 0;
            if (EXT && (rc.flags & kExtSpectral)) beta = ext_wavelength<T>(sc, rc.sA, rc.sB, pixel, sample, ex);
            fresh = false;
        }
        Vec<T> contrib; T t_hit;
        SegInfo si = trace_segment<T, BVH, EXT>(sc, rc, o, d, beta, pixel, sample, b, b + 1 < rc.max_depth, contrib, t_hit, &ex);
        ++nseg;
        if (b == 0) { if (si.has_contrib) Lacc = contrib; }
        else if (si.has_contrib) Lacc = Lacc + contrib;
        ++b;
        if (!si.alive || b == rc.max_depth) {
            Pack3<T> l; l.x = Lacc.x; l.y = Lacc.y; l.z = Lacc.z;
            a.L[idx] = l;
            idx += stride;
            fresh = true;
        }
    }
    for (int sft = 32; sft > 0; sft >>= 1) nseg += __shfl_down(nseg, sft);
    if ((threadIdx.x & 63) == 0 && nseg) atomicAdd(&a.stats->segments, nseg);
}

// Diagnostic: trace chosen paths and record every segment (prim, t, direction) — used by the
// parity tests to compare path geometry bit for bit with the CPU restatement.
template <class T, bool BVH, bool EXT>
__global__ __launch_bounds__(64) void k_trace(const BounceArgs<T> a, const uint32_t *ijs, uint32_t n_paths, int *prims, T *ts,
                                              T *dirs, T *radiance) {
    extern __shared__ __attribute__((aligned(32))) unsigned char lds_raw[];
    const SceneLds<T> sc = stage_scene<T>(a.scene, lds_raw);
    const RenderConst<T> &rc = a.rc;
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_paths) return;
    uint32_t i = ijs[3 * p], j = ijs[3 * p + 1], sample = ijs[3 * p + 2];
    uint32_t pixel = (j - 1) * rc.width + (i - 1);
    Vec<T> o, d, beta = mk<T>(1, 1, 1), Lacc = mk<T>(0, 0, 0);
    camera_ray<T>(rc, i, j, pixel, sample, o, d);
    ExtState<T> ex; ex.flags = rc.flags; ex.bR = 0; ex.bG = 0; ex.bB = 0;
    if (EXT && (rc.flags & kExtSpectral)) beta = ext_wavelength<T>(sc, rc.sA, rc.sB, pixel, sample, ex);
    for (uint32_t b = 0; b < rc.max_depth; ++b) {
        Vec<T> contrib; T t_hit;
        Vec<T> dir_in = d;
        SegInfo si = trace_segment<T, BVH, EXT>(sc, rc, o, d, beta, pixel, sample, b, b + 1 < rc.max_depth, contrib, t_hit, &ex);
        size_t k = (size_t)p * rc.max_depth + b;
        prims[k] = si.prim; ts[k] = t_hit;
        dirs[3 * k] = dir_in.x; dirs[3 * k + 1] = dir_in.y; dirs[3 * k + 2] = dir_in.z;
        if (b == 0) { if (si.has_contrib) Lacc = contrib; }
        else if (si.has_contrib) Lacc = Lacc + contrib;
        if (!si.alive) { for (uint32_t bb = b + 1; bb < rc.max_depth; ++bb) prims[(size_t)p * rc.max_depth + bb] = -2; break; }
    }
    radiance[3 * p] = Lacc.x; radiance[3 * p + 1] = Lacc.y; radiance[3 * p + 2] = Lacc.z;
}

// ====================================================================== secondary integrator variants
// Same scene/LDS/finalize machinery, different estimator.  Both run one lane per path (CPU variant) or per
// pixel (METAL variant, whose RNG state is carried from sample to sample); spheres only, like their sources.
template <class T> __device__ __forceinline__ Vec<T> sa_normalize(Vec<T> a) {     // StaticArrays: inv(norm(a)) * a
    T inv = (T)1.0 / sqrt_rn(dot(a, a));
    return mk<T>(inv * a.x, inv * a.y, inv * a.z);
}
template <class T> __device__ __forceinline__ Vec<T> sa_normalize(Vec<T> a, ExactDiv &) { return sa_normalize(a); }
template <class T> __device__ __forceinline__ Vec<T> sa_normalize(Vec<T> a, SpecDiv &g) {     // 1 / length through the speculative reciprocal; the rest are products
    const T s = a.x * a.x + a.y * a.y + a.z * a.z;                                  // dot(a, a)
    const uint32_t ms = mag_word(s);
    g.hi = max(g.hi, ms); g.lo = min(g.lo, ms);
    const T inv = quotient((T)1.0, recip_of(sqrt_moderate(s)));
    return mk<T>(inv * a.x, inv * a.y, inv * a.z);
}

// trace_ray of render_with_cpu, src/spira-metal-optimized.jl:1351-1412, iteratively:
//   L = beta * (emission | sky) ends the path; beta *= albedo*0.5 (diffuse) | albedo (metallic).
// Returns the number of segments; rec* (optional) record the per-segment trace.
template <class T, class P>
__device__ __forceinline__ uint32_t path_cpu(const SceneLds<T> &sc, const RenderConst<T> &rc, const PixelDiv<T> &pd, uint32_t i, uint32_t j, uint32_t sample,
                                             Vec<T> &L, int *rec_prims, T *rec_ts, T *rec_dirs, P &pol) {
    const uint32_t pixel = (j - 1) * rc.width + (i - 1);
    T xu, xv, unused;
    rng3<T>(rng_key(rc.sA, rc.sB, pixel, sample, 0), 0, xu, xv, unused);
    T u = pixel_quotient<T>((T)(i - 1) + xu, pd.w1, pol);                           // :1428
    T v = pixel_quotient<T>((T)(j - 1) + xv, pd.h1, pol);                           // :1429
    Vec<T> o = rc.cam_origin;
    Vec<T> d = sa_normalize(((rc.cam_llc + rc.cam_hor * u) + rc.cam_ver * v) - o, pol);  // :1431-1432 (Ray ctor normalises, :297)
    Vec<T> beta = mk<T>(1, 1, 1);
    L = mk<T>(0, 0, 0);
    uint32_t nseg = 0;
    for (uint32_t b = 0; b < rc.max_depth; ++b) {                                   // depth <= 0 -> BLACK, :1352
        ++nseg;
        bool hit = false; int prim = -1, mi = 0;
        T closest = (T)1e20f;                                                       // INF, :287
        Vec<T> n = mk<T>(0, 0, 0);
        for (uint32_t s = 0; s < sc.n_spheres; ++s) {                               // :1362-1385
            const Pack4<T> c = sc.sph[s];
            Vec<T> ctr = mk<T>(c.x, c.y, c.z);
            Vec<T> oc = o - ctr;
            T a = (T)1.0;                                                           // :1364
            T half_b = dot(oc, d);
            T cc = dot(oc, oc) - c.w;
            const T hb2 = half_b * half_b;
            T disc = hb2 - a * cc;                                                  // :1367
            if (disc > 0) {
                root_operands<T>(hb2, disc, pol);
                T sq = root_sqrt<T>(disc, pol);
                T root = (-half_b - sq) / a;                                        // :1373
                if (root < (T)0.001f) root = (-half_b + sq) / a;                    // :1374-1376
                if (root > (T)0.001f && root < closest) {                           // :1378
                    closest = root; hit = true; prim = (int)s;
                    n = sa_normalize((o + d * closest) - ctr, pol);                 // :1381
                    mi = sc.smat[s];
                }
            }
        }
        if (rec_prims) { rec_prims[b] = prim; rec_ts[b] = hit ? closest : (T)0; rec_dirs[3 * b] = d.x; rec_dirs[3 * b + 1] = d.y; rec_dirs[3 * b + 2] = d.z; }
        if (!hit) {                                                                 // sky, :1411-1412
            T ts = (T)0.5 * (d.y + (T)1.0);
            L = mulv(beta, mk<T>(1, 1, 1) * ((T)1.0 - ts) + mk<T>((T)0.5, (T)0.7, (T)1.0) * ts);
            break;
        }
        const Pack4<T> ma = sc.mat[2 * mi], mb = sc.mat[2 * mi + 1];
        if (mb.x > 0 || mb.y > 0 || mb.z > 0) { L = mulv(beta, mk<T>(mb.x, mb.y, mb.z)); break; }   // :1392-1394: emitters end the path
        if (b + 1 >= rc.max_depth) break;                                           // next level returns BLACK
        const RngKey k = rng_key(rc.sA, rc.sB, pixel, sample, b);
        T lobe, r0, r1, r2, u0_, u1_;
        rng3<T>(k, 1, lobe, u0_, u1_);
        rng3<T>(k, 2, r0, r1, r2);
        const Vec<T> rv = mk<T>(r0, r1, r2) - mk<T>((T)0.5, (T)0.5, (T)0.5);        // rand(Vec3) - 0.5f0
        const Vec<T> pos = o + d * closest;                                         // :1388
        if (lobe > ma.w) {                                                          // rand > metallic -> diffuse, :1397
            Vec<T> target = (pos + n) + sa_normalize(rv, pol);                      // :1399
            d = sa_normalize(sa_normalize(target - pos, pol), pol);                 // :1400 + Ray ctor
            beta = mulv(mk<T>(ma.x, ma.y, ma.z), beta) * (T)0.5;                    // albedo .* L .* 0.5, :1401
        } else {
            Vec<T> reflected = d - n * ((T)2.0 * dot(d, n));                        // :1404
            d = sa_normalize(sa_normalize(reflected + rv * mb.w, pol), pol);        // :1405 + ctor
            beta = mulv(mk<T>(ma.x, ma.y, ma.z), beta);                             // :1406
        }
        o = pos;
    }
    return nseg;
}
template <class T>
__device__ __forceinline__ uint32_t path_cpu(const SceneLds<T> &sc, const RenderConst<T> &rc, uint32_t i, uint32_t j, uint32_t sample,
                                             Vec<T> &L, int *rec_prims, T *rec_ts, T *rec_dirs) {
    ExactDiv exact;
    PixelDiv<T> pd;
    pd.w1.d = (T)(rc.width - 1); pd.w1.r = 0; pd.h1.d = (T)(rc.height - 1); pd.h1.r = 0;      // (ExactDiv divides by .d)
    return path_cpu<T>(sc, rc, pd, i, j, sample, L, rec_prims, rec_ts, rec_dirs, exact);
}

__device__ __forceinline__ float lcg_uniform(uint32_t &st, float) { st = st * 1664525u + 1013904223u; return (float)(st & 0x00FFFFFFu) / (float)0x01000000; }
__device__ __forceinline__ double lcg_uniform(uint32_t &st, double) { st = st * 1664525u + 1013904223u; return (double)(st & 0x00FFFFFFu) / (double)0x01000000; }

// sin/cos of 2*pi*r, r in [0,1): the same fixed polynomial, in the same plain arithmetic, as the oracle's.
template <class T> __device__ __forceinline__ void sincos_turn(T r, T &sn, T &cs) {
    T t = r * (T)4.0;
    int k = (int)(t + (T)0.5);
    T f = t - (T)k;
    T th = f * (T)1.57079632679489661923;
    T x2 = th * th;
    const double SC[8] = {-1.0 / 6, 1.0 / 120, -1.0 / 5040, 1.0 / 362880, -1.0 / 39916800, 1.0 / 6227020800.0, -1.0 / 1307674368000.0,
                          1.0 / 355687428096000.0};
    const double CC[9] = {-1.0 / 2, 1.0 / 24, -1.0 / 720, 1.0 / 40320, -1.0 / 3628800, 1.0 / 479001600, -1.0 / 87178291200.0,
                          1.0 / 20922789888000.0, -1.0 / 6402373705728000.0};
    constexpr int ns = sizeof(T) == 4 ? 4 : 8, nc = sizeof(T) == 4 ? 5 : 9;
    T ps = 0, pc = 0;
#pragma unroll
    for (int i = ns - 1; i >= 0; --i) ps = (ps + (T)SC[i]) * x2;
#pragma unroll
    for (int i = nc - 1; i >= 0; --i) pc = (pc + (T)CC[i]) * x2;
    T s0 = th + th * ps, c0 = (T)1.0 + pc;
    switch (k & 3) {
    case 0: sn = s0; cs = c0; break;
    case 1: sn = c0; cs = -s0; break;
    case 2: sn = -s0; cs = -c0; break;
    default: sn = -c0; cs = s0; break;
    }
}

// ---- the pieces of path_trace, src/spira_path_trace_kernel.metal:140-269, shared by the one-lane-per-pixel kernel
// (k_variant_metal) and the wavefront kernel (k_path_metal)
// Closest sphere, intersect_sphere :109-136 + the selection loop :181-189 (`t < closest_t`: ties keep the EARLIER sphere).
// (P: ExactDiv, or SpecDiv in k_variant_metal's speculative instantiation — the roots are quotients over a = d.d, a root below the
// window is rejected against EPSILON like the IEEE one: see ExpWindow)
template <class T, class P>
__device__ __forceinline__ int metal_intersect(const SceneLds<T> &sc, const Vec<T> o, const Vec<T> d, T &closest, P &pol) {
    const T EPSILON = (T)0.0001f, INF_ = (T)1e20f;                                  // :6-7
    closest = INF_;
    int hit = -1;
    const T a = dot(d, d);
    const RootDiv<T> over_a = root_divisor<T>(a, pol);
    root_t_min<T>(EPSILON, pol);
    for (uint32_t s = 0; s < sc.n_spheres; ++s) {
        const Pack4<T> c = sc.sph[s];
        Vec<T> oc = o - mk<T>(c.x, c.y, c.z);
        T half_b = dot(oc, d);
        T cc = dot(oc, oc) - c.w;
        const T hb2 = half_b * half_b;
        T disc = hb2 - a * cc;                                                      // :117
        T t = INF_;
        if (disc > (T)0.0) {
            root_operands<T>(hb2, disc, pol);
            T sq = root_sqrt<T>(disc, pol);
            T root = root_over<T>(-half_b - sq, over_a, pol);                       // :120
            if (!(root > EPSILON)) root = root_over<T>(-half_b + sq, over_a, pol);  // :127
            if (root > EPSILON) t = root;                                           // :123 / :130
        }
        if (t < closest) { closest = t; hit = (int)s; }                             // :184-188
    }
    return hit;
}
template <class T>
__device__ __forceinline__ int metal_intersect(const SceneLds<T> &sc, const Vec<T> o, const Vec<T> d, T &closest) {
    ExactDiv exact;
    return metal_intersect<T>(sc, o, d, closest, exact);
}

// sky term of a miss, :192-198
template <class T> __device__ __forceinline__ Vec<T> metal_sky(const Vec<T> d, const Vec<T> thr) {
    T ts = (T)0.5 * (d.y + (T)1.0);
    return mulv(thr, mk<T>(1, 1, 1) * ((T)1.0 - ts) + mk<T>((T)0.5, (T)0.7, (T)1.0) * ts);
}

// Everything that follows a hit, :200-247: normal (the winning sphere's `normalize(hit_point - center)`, :123/:130 — the same
// expression on the same values as the reference's per-candidate normal), flip towards the ray, emitted term, scatter (LCG draws
// in the reference's order), throughput, Russian roulette after depth 3, throughput cut-off.  o/d: in = the segment's ray and
// (o) its hit point, out = the scattered ray.  Returns whether the path goes on; `emitted` is thr_in * emission.
template <class T, class P>
__device__ __forceinline__ bool metal_shade(const SceneLds<T> &sc, int hit, uint32_t depth, uint32_t &st, Vec<T> &o, Vec<T> &d, Vec<T> &thr,
                                            Vec<T> &emitted, bool &has_emission, P &pol) {
    const T EPSILON = (T)0.0001f;
    const Pack4<T> c = sc.sph[hit];
    const Vec<T> hit_point = o;                                                     // :203 (computed by the caller: o + d * closest)
    Vec<T> n = normalize(hit_point - mk<T>(c.x, c.y, c.z), pol);
    const int mi = sc.smat[hit];
    const Pack4<T> ma = sc.mat[2 * mi], mb = sc.mat[2 * mi + 1];
    if (dot(d, n) > (T)0.0) n = mk<T>(-n.x, -n.y, -n.z);                            // :207-209
    emitted = mulv(thr, mk<T>(mb.x, mb.y, mb.z));                                   // :212
    has_emission = (mb.x != 0 || mb.y != 0 || mb.z != 0);
    Vec<T> scatter_origin = hit_point + n * EPSILON;                                // :215
    Vec<T> nd;
    T xi = lcg_uniform(st, (T)0);
    if (xi < ma.w) {                                                                // metal, :219
        nd = d - n * ((T)2.0 * dot(d, n));                                          // reflect, :220
        if (mb.w > (T)0.0) {                                                        // :221
            Vec<T> pv = mk<T>(0, 0, 0);
            for (int guard = 0; guard < 4096; ++guard) {                            // random_unit_vector, :61-70 (bounded: every wave must finish)
                T a0 = lcg_uniform(st, (T)0), a1 = lcg_uniform(st, (T)0), a2 = lcg_uniform(st, (T)0);
                pv = mk<T>(a0 * (T)2.0 - (T)1.0, a1 * (T)2.0 - (T)1.0, a2 * (T)2.0 - (T)1.0);
                if (dot(pv, pv) < (T)1.0) break;
            }
            nd = normalize(nd + normalize(pv, pol) * mb.w, pol);                    // :222
        }
    } else {                                                                        // cosine hemisphere, :73-93
        T r1 = lcg_uniform(st, (T)0), r2 = lcg_uniform(st, (T)0), sn, cs;
        sincos_turn<T>(r1, sn, cs);
        T sr = root_sqrt<T>(r2, pol);                                               // (arguments in [0, 1]: only the lower side needs care)
        T hx = cs * sr, hy = sn * sr;
        T zz = (T)1.0 - hx * hx - hy * hy;
        T hz = root_sqrt<T>(zz > (T)0.0 ? zz : (T)0.0, pol);
        Vec<T> helper = abs_t(n.x) > (T)0.1 ? mk<T>(0, 1, 0) : mk<T>(1, 0, 0);      // :89
        Vec<T> ua = normalize(cross(helper, n), pol);
        Vec<T> va = cross(n, ua);
        nd = normalize((ua * hx + va * hy) + n * hz, pol);                          // :93
    }
    o = scatter_origin; d = nd;
    thr = mulv(thr, mk<T>(ma.x, ma.y, ma.z));                                       // :232
    if (depth > 3) {                                                                // :236-243
        T pc = thr.x > thr.y ? thr.x : thr.y; pc = pc > thr.z ? pc : thr.z;
        pc = pc < (T)0.95f ? pc : (T)0.95f;
        xi = lcg_uniform(st, (T)0);
        if (xi > pc) return false;
        thr = thr / pc;
    }
    { T mx = thr.x > thr.y ? thr.x : thr.y; mx = mx > thr.z ? mx : thr.z; if (mx < (T)0.01f) return false; }   // :246
    return true;
}
template <class T>
__device__ __forceinline__ bool metal_shade(const SceneLds<T> &sc, int hit, uint32_t depth, uint32_t &st, Vec<T> &o, Vec<T> &d, Vec<T> &thr,
                                            Vec<T> &emitted, bool &has_emission) {
    ExactDiv exact;
    return metal_shade<T>(sc, hit, depth, st, o, d, thr, emitted, has_emission, exact);
}

// camera ray of a sample, :158-170 (x, y = gid, 0-based, y = 0 is v = 0)
template <class T>
__device__ __forceinline__ void metal_camera_ray(const RenderConst<T> &rc, uint32_t x, uint32_t y, uint32_t &st, Vec<T> &o, Vec<T> &d) {
    T xi = lcg_uniform(st, (T)0);
    T u_j = ((T)x + xi) / (T)rc.width;                                              // :161
    xi = lcg_uniform(st, (T)0);
    T v_j = ((T)y + xi) / (T)rc.height;                                             // :162
    o = rc.cam_origin;
    d = normalize(((rc.cam_llc + rc.cam_hor * u_j) + rc.cam_ver * v_j) - o);        // :167-170
}
// the same through a division policy; pd: the divisors W and H with their reciprocals (numerators x + xi: +0 or 2^-24 .. 2^31)
template <class T, class P>
__device__ __forceinline__ void metal_camera_ray(const RenderConst<T> &rc, const PixelDiv<T> &pd, uint32_t x, uint32_t y, uint32_t &st, Vec<T> &o, Vec<T> &d, P &pol) {
    T xi = lcg_uniform(st, (T)0);
    T u_j = pixel_quotient<T>((T)x + xi, pd.w1, pol);                               // :161
    xi = lcg_uniform(st, (T)0);
    T v_j = pixel_quotient<T>((T)y + xi, pd.h1, pol);                               // :162
    o = rc.cam_origin;
    d = normalize(((rc.cam_llc + rc.cam_hor * u_j) + rc.cam_ver * v_j) - o, pol);   // :167-170
}

// One sample of path_trace, one lane walking the whole path.
template <class T>
__device__ __forceinline__ uint32_t path_metal(const SceneLds<T> &sc, const RenderConst<T> &rc, uint32_t x, uint32_t y, uint32_t &st,
                                               Vec<T> &acc, int *rec_prims, T *rec_ts, T *rec_dirs) {
    Vec<T> o, d;
    metal_camera_ray<T>(rc, x, y, st, o, d);
    Vec<T> thr = mk<T>(1, 1, 1);
    acc = mk<T>(0, 0, 0);
    uint32_t nseg = 0;
    for (uint32_t depth = 0; depth < rc.max_depth; ++depth) {                       // :176
        ++nseg;
        T closest;
        const int hit = metal_intersect<T>(sc, o, d, closest);
        if (rec_prims) { rec_prims[depth] = hit; rec_ts[depth] = hit >= 0 ? closest : (T)0; rec_dirs[3 * depth] = d.x; rec_dirs[3 * depth + 1] = d.y; rec_dirs[3 * depth + 2] = d.z; }
        if (hit == -1) { acc = acc + metal_sky<T>(d, thr); break; }                 // :192-198
        o = o + d * closest;                                                        // hit_point, :203
        Vec<T> emitted; bool has_e;
        const bool go_on = metal_shade<T>(sc, hit, depth, st, o, d, thr, emitted, has_e);
        acc = acc + emitted;                                                        // :212
        if (!go_on) break;
    }
    return nseg;
}

__device__ __forceinline__ uint32_t metal_state0(uint32_t sA, uint32_t sB, uint32_t pixel) { return mix32(mix32(sA + pixel) ^ sB); }

// SEM 1 = render_with_cpu semantics: one lane per path, result into L (then k_resolve / k_finalize as usual).
template <class T, bool SPEC>
__global__ __launch_bounds__(kBlock) void k_variant_cpu(const BounceArgs<T> a, uint32_t *redo, int redo_only) {
    extern __shared__ __attribute__((aligned(32))) unsigned char lds_raw[];
    const uint32_t wave_id = (blockIdx.x * kBlock + threadIdx.x) >> 6;
    typename std::conditional<SPEC, SpecDiv, ExactDiv>::type pol;            // as in k_variant_metal
    if (!SPEC && redo_only) {
        uint32_t any = 0;
        for (uint32_t w = 0; w < kBlock / 64; ++w) any |= redo[blockIdx.x * (kBlock / 64) + w];
        if (!any) return;
    }
    const SceneLds<T> sc = stage_scene<T>(a.scene, lds_raw);
    const RenderConst<T> &rc = a.rc;
    const PixelDiv<T> pix_div = pixel_divisors<T>(rc);
    if (!SPEC && redo_only) {
        if (!redo[wave_id]) return;
        if ((threadIdx.x & 63) == 0) atomicAdd(&a.stats->redone_waves, 1ull);
    }
    unsigned long long nseg = 0;
    for (uint32_t idx = blockIdx.x * kBlock + threadIdx.x; idx < a.n_first; idx += gridDim.x * kBlock) {
        uint32_t pixel, sample, pi, pj;
        path_of<T>(rc, idx, a.pass, pi, pj, pixel, sample);
        Vec<T> Lp;
        nseg += path_cpu<T>(sc, rc, pix_div, pi, pj, sample, Lp, nullptr, nullptr, nullptr, pol);
        Pack3<T> l; l.x = Lp.x; l.y = Lp.y; l.z = Lp.z;
        a.L[idx] = l;
    }
    for (int sft = 32; sft > 0; sft >>= 1) nseg += __shfl_down(nseg, sft);
    const bool again = SPEC && (redo_only == 2 || __any(outside_window<T>(pol)));
    if ((threadIdx.x & 63) == 0) {
        if (nseg && !again) atomicAdd(&a.stats->segments, nseg);
        if (SPEC) redo[wave_id] = again ? 1u : 0u;
    }
}

// SEM 2 = the .metal kernel's semantics: one lane per pixel walks all spp samples (its LCG state runs through
// them, .metal :155/:268) and leaves their SUM in accum (the `output_hdr_image[p] += L` of :264).  The sample and depth loops
// are flattened into one loop with REGENERATION: a lane whose path has ended starts the pixel's next sample (then its next
// pixel) at once, so every trip intersects one ray per lane instead of idling behind the wave's longest path.
// Progressive use (`resume`): start from the sums already in accum and, when rng_states is given, from the LCG
// states a previous call left there (the `rng_states[pixel_idx] = rng_state` of :268).
// SPEC: speculative division (SpecDiv; fresh renders only — a wave that has to be rendered again must find its inputs untouched, and a
// progressive call updates sums and LCG states in place).  redo: [waves] the speculative launch's report; redo_only: the exact launch
// behind it renders the reported waves.  A reported wave does not add its segments to the statistics: the second rendering does.
template <class T, bool SPEC>
__global__ __launch_bounds__(kBlock) void k_variant_metal(const BounceArgs<T> a, Pack4<T> *accum, uint32_t *rng_states, int resume, uint32_t *redo, int redo_only) {
    extern __shared__ __attribute__((aligned(32))) unsigned char lds_raw[];
    const uint32_t wave_id = (blockIdx.x * kBlock + threadIdx.x) >> 6;
    typename std::conditional<SPEC, SpecDiv, ExactDiv>::type pol;
    if (!SPEC && redo_only) {
        uint32_t any = 0;
        for (uint32_t w = 0; w < kBlock / 64; ++w) any |= redo[blockIdx.x * (kBlock / 64) + w];
        if (!any) return;                                      // (workgroup-uniform, ahead of the barrier in stage_scene)
    }
    const SceneLds<T> sc = stage_scene<T>(a.scene, lds_raw);
    const RenderConst<T> &rc = a.rc;
    PixelDiv<T> pix_div;
    pix_div.w1 = recip_of((T)rc.width); pix_div.h1 = recip_of((T)rc.height);
    pix_div.w1.d = to_scalar(pix_div.w1.d); pix_div.w1.r = to_scalar(pix_div.w1.r); pix_div.h1.d = to_scalar(pix_div.h1.d); pix_div.h1.r = to_scalar(pix_div.h1.r);
    if (!SPEC && redo_only) {
        if (!redo[wave_id]) return;                            // wave-uniform; no workgroup barrier follows
        if ((threadIdx.x & 63) == 0) atomicAdd(&a.stats->redone_waves, 1ull);
    }
    unsigned long long nseg = 0;
    const uint32_t stride = gridDim.x * kBlock;
    uint32_t pl = blockIdx.x * kBlock + threadIdx.x;
    bool new_pixel = true, new_path = true;
    uint32_t pi = 1, pj = 1, st = 0, s = 0, depth = 0;
    Vec<T> sum = mk<T>(0, 0, 0), acc = mk<T>(0, 0, 0), thr = mk<T>(1, 1, 1), o = mk<T>(0, 0, 0), d = mk<T>(0, 0, 1);
    while (pl < rc.tile_pixels) {
        if (new_pixel) {
            uint32_t pixel, sample;
            path_of<T>(rc, pl, 0, pi, pj, pixel, sample);
            st = ((resume & 2) && rng_states) ? rng_states[pl] : metal_state0(rc.sA, rc.sB, pixel);
            sum = mk<T>(0, 0, 0);
            if (resume & 1) { const Pack4<T> l0 = accum[pl]; sum = mk<T>(l0.x, l0.y, l0.z); }
            s = 0; new_pixel = false; new_path = true;
        }
        if (new_path) {
            metal_camera_ray<T>(rc, pix_div, pi - 1, pj - 1, st, o, d, pol);
            thr = mk<T>(1, 1, 1); acc = mk<T>(0, 0, 0); depth = 0;
            new_path = false;
        }
        ++nseg;
        T closest;
        const int hit = metal_intersect<T>(sc, o, d, closest, pol);
        bool ended;
        if (hit == -1) { acc = acc + metal_sky<T>(d, thr); ended = true; }            // :192-198
        else {
            o = o + d * closest;                                                      // hit_point, :203
            Vec<T> emitted; bool has_e;
            const bool go_on = metal_shade<T>(sc, hit, depth, st, o, d, thr, emitted, has_e, pol);
            acc = acc + emitted;                                                      // :212
            ++depth;
            ended = !go_on || depth == rc.max_depth;
        }
        if (ended) {
            sum = sum + acc;                                                          // output_hdr_image[p] += L, :264
            new_path = true;
            if (++s == rc.spp) {
                Pack4<T> l; l.x = sum.x; l.y = sum.y; l.z = sum.z; l.w = 0;
                accum[pl] = l;
                if (rng_states) rng_states[pl] = st;
                pl += stride;
                new_pixel = true;
            }
        }
    }
    for (int sft = 32; sft > 0; sft >>= 1) nseg += __shfl_down(nseg, sft);
    const bool again = SPEC && (redo_only == 2 || __any(outside_window<T>(pol)));      // redo_only == 2 on the speculative launch: report every wave (tests)
    if ((threadIdx.x & 63) == 0) {
        if (nseg && !again) atomicAdd(&a.stats->segments, nseg);
        if (SPEC) redo[wave_id] = again ? 1u : 0u;
    }
}

// SEM 2 in wavefront form (k_path's organisation applied to the .metal estimator).  The estimator's LCG state runs from sample
// to sample of a pixel (.metal :155/:268), so only ONE sample per pixel can be in flight: the parallelism is the tile's pixels,
// not pixels x spp.  Every wave owns a fixed block of pixels (and the matching region of both hit queues) and walks, for each
// sample in turn, all stages on it: stage 0 draws the jitter and intersects the camera ray, stage k shades the hits of depth k
// (normal flip, emitted term, metal / cosine-hemisphere scatter, Russian roulette after depth 3, throughput cut-off — the LCG
// state travels in the packet), intersects the scattered ray and compacts the hits by ballot + popcount.  A path that ends
// adds its radiance to the pixel's running sum (`output_hdr_image[p] += L`, :264) and parks the LCG state for the next sample.
// Same statements in the same order as k_variant_metal: bit-identical sums and states.
template <class T> struct MetalArgs {
    SceneGlobal<T> scene;
    RenderConst<T> rc;
    RayQueue<T> q[2];
    uint2 *qx[2];                    // per packet: {hit sphere, LCG state}
    Pack3<T> *L;                     // [tile_pixels] radiance of the pixel's sample in flight (only for paths that met an emitter)
    Pack4<T> *accum;                 // [tile_pixels] running sums
    uint32_t *rng_states;            // [tile_pixels]
    uint32_t *blk_stats;             // [NW][4]
    uint32_t ppw;                    // pixels per wave (a multiple of 64) == region size
    int resume;                      // bit 0: continue the sums in accum, bit 1: continue the LCG states in rng_states
    uint32_t *redo;                  // speculative division (fresh renders only), as PathArgs::redo / redo_only (2 on the speculative launch: report every wave)
    int redo_only;
    Stats *stats;
};

template <class T, int R, bool SPEC>
__global__ __launch_bounds__(kBlock, sizeof(T) == 4 ? SPIRA_WAVES_PATH_F32 : SPIRA_WAVES_F64) void k_path_metal(const MetalArgs<T> a) {
    extern __shared__ __attribute__((aligned(32))) unsigned char lds_raw[];
    constexpr uint32_t WPB = kBlock / 64, SUB = 64 * R;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t wid = blockIdx.x * WPB + wave;
    const RenderConst<T> &rc = a.rc;
    typename std::conditional<SPEC, SpecDiv, ExactDiv>::type pol;
    if (!SPEC && a.redo_only) {
        uint32_t any = 0;
        for (uint32_t w = 0; w < WPB; ++w) any |= a.redo[blockIdx.x * WPB + w];
        if (!any) return;                                        // (workgroup-uniform, ahead of the barrier in stage_scene)
    }
    const SceneLds<T> sc = stage_scene<T>(a.scene, lds_raw);
    PixelDiv<T> pix_div;
    pix_div.w1 = recip_of((T)rc.width); pix_div.h1 = recip_of((T)rc.height);
    pix_div.w1.d = to_scalar(pix_div.w1.d); pix_div.w1.r = to_scalar(pix_div.w1.r); pix_div.h1.d = to_scalar(pix_div.h1.d); pix_div.h1.r = to_scalar(pix_div.h1.r);
    if (!SPEC && a.redo_only) {
        if (!a.redo[wid]) return;                                // wave-uniform; no workgroup barrier follows
        if (lane == 0) atomicAdd(&a.stats->redone_waves, 1ull);
    }
    const uint32_t region = wid * a.ppw;                          // first pixel of this wave == first slot of its queue regions
    const uint32_t n_pix = region < rc.tile_pixels ? (rc.tile_pixels - region < a.ppw ? rc.tile_pixels - region : a.ppw) : 0u;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    uint32_t n_seg = 0, n_enq = 0, n_rmw = 0, n_store = 0;
    // initial sums / states of this wave's pixels
    for (uint32_t i = lane; i < n_pix; i += 64) {
        const uint32_t pl = region + i;
        if (!(a.resume & 1)) { Pack4<T> z; z.x = 0; z.y = 0; z.z = 0; z.w = 0; a.accum[pl] = z; }
        if (!(a.resume & 2)) {
            uint32_t pixel, sample, pi, pj;
            path_of<T>(rc, pl, 0, pi, pj, pixel, sample);
            a.rng_states[pl] = metal_state0(rc.sA, rc.sB, pixel);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");

    // a path ends: pixel sum += the sample's radiance (last term `c`, on top of what L holds when the path met an emitter)
    auto finish = [&](uint32_t pl, bool has_l, bool has_c, const Vec<T> c, uint32_t st) {
        Vec<T> tot = mk<T>(0, 0, 0);
        if (has_l) { const Pack3<T> l0 = a.L[pl]; tot = mk<T>(l0.x, l0.y, l0.z); if (has_c) tot = tot + c; ++n_rmw; }
        else if (has_c) tot = c;
        Pack4<T> sum = a.accum[pl];
        sum.x = sum.x + tot.x; sum.y = sum.y + tot.y; sum.z = sum.z + tot.z;      // sum = sum + c, k_variant_metal's order
        a.accum[pl] = sum;
        a.rng_states[pl] = st;
        ++n_store;
    };

    for (uint32_t s = 0; s < rc.spp; ++s) {
        uint32_t n_in = 0;
        for (uint32_t stage = 0; stage < rc.max_depth; ++stage) {
            const bool first = stage == 0;
            const RayQueue<T> qin = a.q[(stage + 1) & 1], qout = a.q[stage & 1];
            const uint2 *xin = a.qx[(stage + 1) & 1];
            uint2 *xout = a.qx[stage & 1];
            const uint32_t limit = first ? n_pix : n_in;
            uint32_t fill = 0;
            for (uint32_t sub = 0; sub * SUB < limit; ++sub) {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const uint32_t idx = sub * SUB + r * 64 + lane;
                    bool hit_next = false;
                    Vec<T> o = mk<T>(0, 0, 0), d = mk<T>(0, 0, 0), thr = mk<T>(1, 1, 1);
                    uint32_t q = 0, st = 0, ref = 0;
                    if (idx < limit) {
                        int hit;
                        bool alive = true;
                        if (first) {
                            q = region + idx;
                            uint32_t pixel, sample, pi, pj;
                            path_of<T>(rc, q, 0, pi, pj, pixel, sample);
                            st = a.rng_states[q];
                            metal_camera_ray<T>(rc, pix_div, pi - 1, pj - 1, st, o, d, pol);
                            T closest;
                            hit = metal_intersect<T>(sc, o, d, closest, pol);
                            ++n_seg;
                            if (hit < 0) { finish(q, false, true, metal_sky<T>(d, thr), st); alive = false; }
                            else o = o + d * closest;                                   // hit_point, :203
                        } else {
                            const Pack4<T> A = qin.A[region + idx], B = qin.B[region + idx];
                            const Pack2<T> C = qin.C[region + idx];
                            const uint2 X = xin[region + idx];
                            o = mk<T>(A.x, A.y, A.z); d = mk<T>(A.w, B.x, B.y); thr = mk<T>(B.z, B.w, C.x);
                            q = Bits<T>::to_u32(C.y); hit = (int)X.x; st = X.y;
                        }
                        if (alive) {
                            const uint32_t pl = q & 0x7FFFFFFFu;
                            bool has_l = (q >> 31) != 0;
                            Vec<T> emitted; bool has_e;
                            const bool go_on = metal_shade<T>(sc, hit, stage, st, o, d, thr, emitted, has_e, pol);
                            if (go_on && stage + 1 < rc.max_depth) {
                                if (has_e) {                                          // the path continues: park the emitted term in L
                                    Pack3<T> l; l.x = emitted.x; l.y = emitted.y; l.z = emitted.z;
                                    if (has_l) { const Pack3<T> l0 = a.L[pl]; l.x = l0.x + emitted.x; l.y = l0.y + emitted.y; l.z = l0.z + emitted.z; }
                                    a.L[pl] = l;
                                    has_l = true; q |= 0x80000000u;
                                }
                                T closest;
                                const int nh = metal_intersect<T>(sc, o, d, closest, pol);
                                ++n_seg;
                                if (nh < 0) finish(pl, has_l, true, metal_sky<T>(d, thr), st);
                                else { hit_next = true; o = o + d * closest; ref = (uint32_t)nh; }
                            } else finish(pl, has_l, has_e, emitted, st);              // Russian roulette, cut-off or max_depth
                        }
                    }
                    const unsigned long long m = __ballot(hit_next);
                    if (hit_next) {
                        const uint32_t dst = region + fill + __popcll(m & lt_mask);
                        Pack4<T> A, B; Pack2<T> C;
                        A.x = o.x; A.y = o.y; A.z = o.z; A.w = d.x;
                        B.x = d.y; B.y = d.z; B.z = thr.x; B.w = thr.y;
                        C.x = thr.z; C.y = Bits<T>::from_u32(q);
                        qout.A[dst] = A; qout.B[dst] = B; qout.C[dst] = C;
                        xout[dst] = make_uint2(ref, st);
                    }
                    fill += (uint32_t)__popcll(m);
                }
            }
            n_in = fill;
            n_enq += fill;
            // the wave reads next what its own lanes have just written (queues, sums, LCG states): wait for the stores
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            if (n_in == 0) break;
        }
    }
    for (int sft = 32; sft > 0; sft >>= 1) { n_seg += __shfl_down(n_seg, sft); n_rmw += __shfl_down(n_rmw, sft); n_store += __shfl_down(n_store, sft); }
    const bool again = SPEC && (a.redo_only == 2 || __any(outside_window<T>(pol)));
    if (lane == 0) {
        a.blk_stats[4 * wid] = n_seg;
        a.blk_stats[4 * wid + 1] = n_rmw;
        a.blk_stats[4 * wid + 2] = n_store;
        a.blk_stats[4 * wid + 3] = n_enq;
        if (SPEC) a.redo[wid] = again ? 1u : 0u;
    }
}

// Folds per-wave statistics rows into the render totals (used after k_path_metal, which has no resolve pass).
static __global__ void k_fold_stats(const uint32_t *blk_stats, uint32_t n_rows, Stats *stats) {
    unsigned long long seg = 0, enq = 0, rmw = 0, sto = 0;
    for (uint32_t i = threadIdx.x; i < n_rows; i += blockDim.x) { seg += blk_stats[4 * i]; rmw += blk_stats[4 * i + 1]; sto += blk_stats[4 * i + 2]; enq += blk_stats[4 * i + 3]; }
    for (int sft = 32; sft > 0; sft >>= 1) { seg += __shfl_down(seg, sft); enq += __shfl_down(enq, sft); rmw += __shfl_down(rmw, sft); sto += __shfl_down(sto, sft); }
    if (threadIdx.x == 0) { stats->segments += seg; stats->rays_enqueued += enq; stats->radiance_rmw += rmw; stats->radiance_store += sto; }
}

// Diagnostic trace for the secondary variants (same outputs as k_trace).
template <class T, int SEM>
__global__ __launch_bounds__(64) void k_trace_variant(const BounceArgs<T> a, const uint32_t *ijs, uint32_t n_paths, int *prims, T *ts,
                                                      T *dirs, T *radiance) {
    extern __shared__ __attribute__((aligned(32))) unsigned char lds_raw[];
    const SceneLds<T> sc = stage_scene<T>(a.scene, lds_raw);
    const RenderConst<T> &rc = a.rc;
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_paths) return;
    uint32_t i = ijs[3 * p], j = ijs[3 * p + 1], sample = ijs[3 * p + 2];
    int *pr = prims + (size_t)p * rc.max_depth;
    T *pt = ts + (size_t)p * rc.max_depth, *pd = dirs + 3 * (size_t)p * rc.max_depth;
    Vec<T> c = mk<T>(0, 0, 0);
    uint32_t nseg = 0;
    if (SEM == 1) {
        nseg = path_cpu<T>(sc, rc, i, j, sample, c, pr, pt, pd);
    } else {
        uint32_t st = metal_state0(rc.sA, rc.sB, (j - 1) * rc.width + (i - 1));
        for (uint32_t s = 0; s <= sample; ++s) nseg = path_metal<T>(sc, rc, i - 1, j - 1, st, c, pr, pt, pd);
    }
    for (uint32_t b = nseg; b < rc.max_depth; ++b) pr[b] = -2;
    radiance[3 * p] = c.x; radiance[3 * p + 1] = c.y; radiance[3 * p + 2] = c.z;
}

// Per-pass resolve: accum[pix] += L[slot][pix] for slot = 0..k_eff-1, in sample order — the
// `color = color + ray_color(...)` of examples/julia-raytracer.jl:401 in the same order.
// Workgroup 0 also folds the pass's per-wave statistics (n_rows x {segments, radiance RMWs, radiance stores, rays
// enqueued}) into the render totals.
template <class T>
__global__ __launch_bounds__(kBlock) void k_resolve(Pack4<T> *accum, const Pack3<T> *L, uint32_t tile_pixels, uint32_t k_eff, int first_pass,
                                                    const uint32_t *blk_stats, uint32_t n_rows, Stats *stats) {
    for (uint32_t p = blockIdx.x * kBlock + threadIdx.x; p < tile_pixels; p += gridDim.x * kBlock) {
        Pack4<T> acc;
        if (first_pass) { acc.x = 0; acc.y = 0; acc.z = 0; acc.w = 0; } else acc = accum[p];
        // eight independent 12/24-byte loads in flight per lane, then the adds in sample order (the sum order is
        // the reference's; only the loads are batched)
        uint32_t s = 0;
        for (; s + 8 <= k_eff; s += 8) {
            Pack3<T> l[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) l[k] = L[(size_t)(s + k) * tile_pixels + p];
#pragma unroll
            for (int k = 0; k < 8; ++k) { acc.x = acc.x + l[k].x; acc.y = acc.y + l[k].y; acc.z = acc.z + l[k].z; }
        }
        for (; s < k_eff; ++s) {
            const Pack3<T> l = L[(size_t)s * tile_pixels + p];
            acc.x = acc.x + l.x; acc.y = acc.y + l.y; acc.z = acc.z + l.z;
        }
        accum[p] = acc;
    }
    if (blockIdx.x == 0 && blk_stats) {
        __shared__ unsigned long long red[4];
        if (threadIdx.x < 4) red[threadIdx.x] = 0;
        __syncthreads();
        unsigned long long seg = 0, enq = 0, rmw = 0, sto = 0;
        for (uint32_t i = threadIdx.x; i < n_rows; i += kBlock) {
            seg += blk_stats[4 * i];
            rmw += blk_stats[4 * i + 1];
            sto += blk_stats[4 * i + 2];
            enq += blk_stats[4 * i + 3];
        }
        for (int sft = 32; sft > 0; sft >>= 1) { seg += __shfl_down(seg, sft); enq += __shfl_down(enq, sft); rmw += __shfl_down(rmw, sft); sto += __shfl_down(sto, sft); }
        if ((threadIdx.x & 63) == 0) { atomicAdd(&red[0], seg); atomicAdd(&red[1], enq); atomicAdd(&red[2], rmw); atomicAdd(&red[3], sto); }
        __syncthreads();
        if (threadIdx.x == 0) { stats->segments += red[0]; stats->rays_enqueued += red[1]; stats->radiance_rmw += red[2]; stats->radiance_store += red[3]; }
    }
}

// ACES / gamma display transforms (to_acescg examples/julia-raytracer.jl:370-384;
// gpu_tone_map_kernel! src/spira-metal-optimized.jl:1128-1144; clamp+sqrt :1441-1442)
template <class T> __host__ __device__ inline T aces1(T x) {
    const T a = (T)2.51, b = (T)0.03, c = (T)2.43, d = (T)0.59, e = (T)0.14;
    T v = (x * (a * x + b)) / (x * (c * x + d) + e);
    return v < 0 ? (T)0 : (v > 1 ? (T)1 : v);
}
__host__ __device__ inline float sqrt_any(float x) { return __builtin_sqrtf(x); }
__host__ __device__ inline double sqrt_any(double x) { return __builtin_sqrt(x); }
// ====================================================================== SPIRA_SEM_HYBRID: render_hybrid_gpu as written
// The estimator of src/spira-metal-optimized.jl:1228-1343 (what render() executes on a Metal / CUDA machine), statement by statement like the oracle's
// oracle_render_hybrid: the whole image advances in lock step, and one condition is image-wide — `if sum(hit_results[:, 1]) == 0 break` (:1303) ends a
// sample for EVERY pixel when no ray of the image hit anything at this depth.  So a sample is max_depth + 1 launches of one lane per pixel:
//   phase 0                 K3 raygen (:610-697, per-pixel xorshift32 :412-426) + K4 intersection of depth 1 (:700-799)
//   phase p = 1..max-1      if any ray hit at depth p: K5 scatter of depth p (:862-989), then K4 of depth p + 1
//   phase max               if any ray hit at depth max: K5, then K6 shade (:1071-1105) with contribution 0.5^max (:1328), K7 tone map (:1128-1144), sum
// `flags[(sample - 1) * (max_depth + 1) + depth]` is set by any lane that hits at that depth (a plain store of 1: no atomic needed) and read by the next
// launch; a depth nobody reached stays 0, so a sample that broke stays broken.  The reference relaunches ~12 kernels per sphere per depth with host
// round trips; here a 1080p frame at spp 64, depth 8 is 576 launches back to back on one stream.  Not a performance path: a fidelity path.
template <class T> struct HybridPx { Vec<T> d, o, pt, n; };      // per-pixel ray state between launches (SoA planes in `state`: 12 values) + mat, rng
template <class T> struct HybridArgs {
    SceneGlobal<T> scene;
    RenderConst<T> rc;
    T *state;                // [12][P]: d.xyz, o.xyz, pt.xyz, n.xyz
    uint32_t *mat;           // [P] material (1-based) of the hit of the last intersection, 0 = none
    uint32_t *rng;           // [P] xorshift32 states
    Pack4<T> *accum;         // [P] sums of the tone-mapped samples, in OUTPUT row order
    uint32_t *flags;         // [spp][max_depth + 1]
    Stats *stats;
    uint32_t sample;         // 1-based (:1275)
    uint32_t phase;          // 0 .. max_depth
};
__device__ __forceinline__ uint32_t xorshift32_step(uint32_t s) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; }      // :412-417
template <class T> __device__ __forceinline__ T xs_uniform(uint32_t s) { return (T)(float)((double)s / 4294967295.0); }       // Float32(state / typemax(UInt32)), :420-426
template <class T> __device__ __forceinline__ T aces_sqrt(T x) {                                                               // :1133-1143
    const T a = (T)2.51f, b = (T)0.03f, c = (T)2.43f, d = (T)0.59f, e = (T)0.14f;
    T r = (x * (a * x + b)) / (x * (c * x + d) + e);
    r = r < (T)0 ? (T)0 : (r > (T)1 ? (T)1 : r);
    return sqrt_rn(r);
}
template <class T>
__global__ __launch_bounds__(kBlock) void k_hybrid(const HybridArgs<T> a) {
    extern __shared__ __attribute__((aligned(32))) unsigned char lds_raw[];
    const RenderConst<T> &rc = a.rc;
    const uint32_t P = rc.width * rc.height, D = rc.max_depth;
    uint32_t *fl = a.flags + (size_t)(a.sample - 1) * (D + 1);
    if (a.phase > 0 && fl[a.phase] == 0) return;                  // the image-wide break (:1303), or a sample that broke earlier (workgroup-uniform: ahead of the barrier)
    const SceneLds<T> sc = stage_scene<T>(a.scene, lds_raw);
    unsigned long long nseg = 0;
    for (uint32_t k = blockIdx.x * kBlock + threadIdx.x; k < P; k += gridDim.x * kBlock) {
        Vec<T> o, d;
        uint32_t st = a.rng[k];
        if (a.phase == 0) {
            // ---- K3
            st = xorshift32_step(st + a.sample);                                                  // :632
            st = xorshift32_step(st); const T rand1 = xs_uniform<T>(st);                          // :635-636
            st = xorshift32_step(st); const T rand2 = xs_uniform<T>(st);                          // :637-638
            const uint32_t row = k / rc.width, col = k - row * rc.width;                          // :651-652
            const T u_center = (T)col / (T)(rc.width - 1), v_center = (T)row / (T)(rc.height - 1);      // :656-657
            const T pw = (T)1.0 / (T)(rc.width - 1), ph = (T)1.0 / (T)(rc.height - 1);            // :669-670
            const T u = u_center + (rand1 - (T)0.5) * (T)0.5 * pw;                                // :672
            const T v = v_center + (rand2 - (T)0.5) * (T)0.5 * ph;                                // :673
            const T dx = rc.cam_llc.x + u * rc.cam_hor.x + v * rc.cam_ver.x - rc.cam_origin.x;    // :676
            const T dy = rc.cam_llc.y + u * rc.cam_hor.y + v * rc.cam_ver.y - rc.cam_origin.y;
            const T dz = rc.cam_llc.z + u * rc.cam_hor.z + v * rc.cam_ver.z - rc.cam_origin.z;
            const T len_sq = dx * dx + dy * dy + dz * dz;                                         // :681
            const T inv_len = len_sq > (T)0 ? sqrt_rn((T)1.0 / len_sq) : (T)0;                    // :682
            o = rc.cam_origin;
            d = mk<T>(dx * inv_len, dy * inv_len, dz * inv_len);                                  // :689-691
        } else {
            // ---- K5 on the hit of depth `phase`
            d = mk<T>(a.state[k], a.state[(size_t)P + k], a.state[2 * (size_t)P + k]);
            const Vec<T> pt = mk<T>(a.state[6 * (size_t)P + k], a.state[7 * (size_t)P + k], a.state[8 * (size_t)P + k]);
            const uint32_t mat = a.mat[k];
            o = pt;                                                                               // :883-885: (0, 0, 0) for a ray that missed
            if (mat) {                                                                            // :888
                const Pack4<T> ma = sc.mat[2 * (mat - 1)], mb = sc.mat[2 * (mat - 1) + 1];
                const T metallic = ma.w, roughness = mb.w;                                        // :892-893
                const T nx = a.state[9 * (size_t)P + k], ny = a.state[10 * (size_t)P + k], nz = a.state[11 * (size_t)P + k];
                const T dot_prod = d.x * nx + d.y * ny + d.z * nz;                                // :904
                if (metallic > (T)0) {                                                            // :907
                    T rx = d.x - (T)2.0 * dot_prod * nx, ry = d.y - (T)2.0 * dot_prod * ny, rz = d.z - (T)2.0 * dot_prod * nz;      // :908-910
                    if (roughness > (T)0) {                                                       // :912
                        st = xorshift32_step(st); T r1 = xs_uniform<T>(st) - (T)0.5;              // :914-925
                        st = xorshift32_step(st); T r2 = xs_uniform<T>(st) - (T)0.5;
                        st = xorshift32_step(st); T r3 = xs_uniform<T>(st) - (T)0.5;
                        const T nl = sqrt_rn(r1 * r1 + r2 * r2 + r3 * r3);                        // :927
                        if (nl > (T)1e-5f) { const T il = (T)1.0 / nl; r1 *= il; r2 *= il; r3 *= il; }      // :928-933
                        rx += roughness * r1; ry += roughness * r2; rz += roughness * r3;         // :935-937
                        const T il = (T)1.0 / sqrt_rn(rx * rx + ry * ry + rz * rz);               // :939
                        rx *= il; ry *= il; rz *= il;
                    }
                    d = mk<T>(rx, ry, rz);                                                        // :944-946
                } else {                                                                          // :947
                    T lx = 0, ly = 0, lz = 0;
                    for (int tries = 0; tries < 64; ++tries) {                                    // while true, :950-963 (bounded like the oracle)
                        st = xorshift32_step(st); const T r1 = xs_uniform<T>(st);
                        st = xorshift32_step(st); const T r2 = xs_uniform<T>(st);
                        st = xorshift32_step(st); const T r3 = xs_uniform<T>(st);
                        lx = r1 * (T)2.0 - (T)1.0; ly = r2 * (T)2.0 - (T)1.0; lz = r3 * (T)2.0 - (T)1.0;
                        if (lx * lx + ly * ly + lz * lz <= (T)1.0) break;
                        if (tries == 63) { lx = 0; ly = 0; lz = 0; }
                    }
                    const T ddx = nx + lx, ddy = ny + ly, ddz = nz + lz;                          // :966-968
                    const T ls = ddx * ddx + ddy * ddy + ddz * ddz;                               // :970
                    if (ls < (T)1e-5f) d = mk<T>(nx, ny, nz);                                     // :971-974
                    else { const T il = (T)1.0 / sqrt_rn(ls); d = mk<T>(ddx * il, ddy * il, ddz * il); }      // :976-979
                }
            }
            if (a.phase == D) {
                // ---- K6 with the hit of depth max and the NEW direction, contribution = 0.5^max (:1328); K7; the running sum (:1334)
                T contribution = (T)1.0;
                for (uint32_t i = 0; i < D; ++i) contribution *= (T)0.5;
                T cr, cg, cb;
                if (mat) {                                                                        // :1084-1096
                    const Pack4<T> ma = sc.mat[2 * (mat - 1)], mb = sc.mat[2 * (mat - 1) + 1];
                    cr = ma.x * contribution + mb.x; cg = ma.y * contribution + mb.y; cb = ma.z * contribution + mb.z;
                } else {                                                                          // :1097-1102
                    const T t = (T)0.5 * (d.y + (T)1.0);
                    cr = ((T)1.0 - t) + t * (T)0.5; cg = ((T)1.0 - t) + t * (T)0.7f; cb = ((T)1.0 - t) + t * (T)1.0;
                }
                const uint32_t row = k / rc.width, col = k - row * rc.width;
                const uint32_t y = (rc.flags & 0x00001000u /*SPIRA_ROWS_BOTTOM_UP*/) ? row : rc.height - 1 - row;      // the row flip of :1177-1188
                Pack4<T> acc = a.accum[(size_t)y * rc.width + col];
                acc.x += aces_sqrt<T>(cr); acc.y += aces_sqrt<T>(cg); acc.z += aces_sqrt<T>(cb);
                a.accum[(size_t)y * rc.width + col] = acc;
                a.rng[k] = st;
                continue;
            }
        }
        // ---- K4: the intersection of depth phase + 1
        uint32_t mat = 0; T best = (T)1e20f;                                                      // :713-715
        Vec<T> pt = mk<T>(0, 0, 0), nn = mk<T>(0, 0, 0);                                          // :718-719
        for (uint32_t s = 0; s < sc.n_spheres; ++s) {                                             // :724
            const T cx = a.scene.spheres5[5 * (size_t)s], cy = a.scene.spheres5[5 * (size_t)s + 1], cz = a.scene.spheres5[5 * (size_t)s + 2], rad = a.scene.spheres5[5 * (size_t)s + 3];
            const T ocx = o.x - cx, ocy = o.y - cy, ocz = o.z - cz;                               // :735-737
            const T aa = d.x * d.x + d.y * d.y + d.z * d.z;                                       // :740
            const T half_b = ocx * d.x + ocy * d.y + ocz * d.z;                                   // :741-743
            const T c = ocx * ocx + ocy * ocy + ocz * ocz - rad * rad;                            // :744
            const T disc = half_b * half_b - aa * c;                                              // :747
            if (!(disc > (T)0)) continue;                                                         // :750
            const T sq = sqrt_rn(disc);                                                           // :759
            const T t1 = (-half_b - sq) / aa, t2 = (-half_b + sq) / aa;                           // :760-761
            const T t = t1 > (T)0.001f ? t1 : t2;                                                 // :764
            if (t <= (T)0.001f || t >= best) continue;                                            // :767
            best = t; mat = (uint32_t)a.scene.spheres5[5 * (size_t)s + 4];                        // :772-774
            pt = mk<T>(o.x + t * d.x, o.y + t * d.y, o.z + t * d.z);                              // :777-779
            const T nx = pt.x - cx, ny = pt.y - cy, nz = pt.z - cz;                               // :782-784
            const T il = (T)1.0 / sqrt_rn(nx * nx + ny * ny + nz * nz);                           // :787
            nn = mk<T>(nx * il, ny * il, nz * il);                                                // :788-790
        }
        ++nseg;
        if (mat) fl[a.phase + 1] = 1u;                                                            // some ray of the image hit at this depth
        a.state[k] = d.x; a.state[(size_t)P + k] = d.y; a.state[2 * (size_t)P + k] = d.z;
        a.state[6 * (size_t)P + k] = pt.x; a.state[7 * (size_t)P + k] = pt.y; a.state[8 * (size_t)P + k] = pt.z;
        a.state[9 * (size_t)P + k] = nn.x; a.state[10 * (size_t)P + k] = nn.y; a.state[11 * (size_t)P + k] = nn.z;
        a.mat[k] = mat;
        a.rng[k] = st;
    }
    for (int sft = 32; sft > 0; sft >>= 1) nseg += __shfl_down(nseg, sft);
    if ((threadIdx.x & 63) == 0 && nseg) atomicAdd(&a.stats->segments, nseg);
}
// the per-pixel xorshift32 states of a render: rand(UInt32, width * height) in the reference (:1258), derived from the seed here (like SPIRA_SEM_METAL's)
static __global__ void k_hybrid_init(uint32_t *rng, uint32_t n, uint32_t sA, uint32_t sB) {
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) rng[k] = mix32(mix32(sA + k) ^ sB);
}

template <class T> __host__ __device__ inline T post1(T x, uint32_t post) {
    switch (post) {
    case 0x000u: return aces1<T>(x);
    case 0x100u: return sqrt_any(aces1<T>(x));
    case 0x200u: { T v = x < 0 ? (T)0 : (x > 1 ? (T)1 : x); return sqrt_any(v); }
    default: return x;
    }
}

// Progressive accumulation: load the caller's planar running sums into the accumulator.
template <class T>
__global__ __launch_bounds__(kBlock) void k_load_accum(Pack4<T> *accum, const T *sum_planar, uint32_t tile_pixels) {
    for (uint32_t p = blockIdx.x * kBlock + threadIdx.x; p < tile_pixels; p += gridDim.x * kBlock) {
        Pack4<T> l; l.x = sum_planar[p]; l.y = sum_planar[tile_pixels + p]; l.z = sum_planar[2 * (size_t)tile_pixels + p]; l.w = 0;
        accum[p] = l;
    }
}

// Finalize: color / samples_per_pixel (:405), planar outputs, optional display transform.
template <class T>
__global__ __launch_bounds__(kBlock) void k_finalize(const Pack4<T> *accum, uint32_t tile_pixels, uint32_t spp, uint32_t post,
                                                     T *out_hdr, T *out_img) {
    for (uint32_t p = blockIdx.x * kBlock + threadIdx.x; p < tile_pixels; p += gridDim.x * kBlock) {
        const Pack4<T> acc = accum[p];
        T r = acc.x / (T)spp, g = acc.y / (T)spp, b = acc.z / (T)spp;
        if (out_hdr) { out_hdr[p] = r; out_hdr[tile_pixels + p] = g; out_hdr[2 * (size_t)tile_pixels + p] = b; }
        if (out_img) {
            out_img[p] = post1<T>(r, post); out_img[tile_pixels + p] = post1<T>(g, post);
            out_img[2 * (size_t)tile_pixels + p] = post1<T>(b, post);
        }
    }
}

}  // namespace spira
