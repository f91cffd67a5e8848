/*
 * spira_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.  PARITY UNPINNED (see below).
 *
 * CPU restatement (plain C) of the path-trace integrator of jenkinsm13/julia-spira, variant A:
 * examples/julia-raytracer.jl (render :387-421, ray_color :328-367, hit :113-258, Camera
 * :261-306, sampling :309-325, to_acescg :370-384).  Instantiated in Float64 (*_f64, the
 * reference's precision) and Float32 (*_f32, the mirror the f32 HIP kernels are checked
 * against).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library; nothing under julia-spira_amd/ links, imports or calls it.
 *
 * PARITY UNPINNED: the reference has no golden vectors / KATs / fixtures for this path
 * (tests/bunny-test.jl:59 asserts only the image size), cannot run here (Julia is not
 * installed, nothing was denied) and never seeds its RNG (Random.default_rng(), Xoshiro256++
 * for julia >= 1.7, Project.toml:23 `julia = "1.8"`, call sites examples/julia-raytracer.jl
 * :311,:398,:399), so its random stream cannot be reproduced.  What IS checked
 * (tests/test_oracle_kat.py): the analytic known answers derivable from the cited formulae
 *  and the one property the reference's own test pins (output size).
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/spira_hip.h" /* plain-data spira_params + flag values only */
#include "../include/spira_spd.h" /* the SPD table of the spectral extension: data shared with the product */

/* xorshift32 of src/spira-metal-optimized.jl:412-417 and its conversion :420-426 (defined below; used by the SPIRA_SEM_HYBRID restatement) */
uint32_t oracle_xorshift32(uint32_t s);
float oracle_xorshift_uniform(uint32_t s);

#define ORACLE_MAX_TRIES 64u

/* lowbias32 integer hash (Chris Wellons, public domain) — DESIGN.md "RNG" */
static inline uint32_t oracle_mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU;
    x ^= x >> 15; x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}

static inline void oracle_seed_mix(uint64_t seed, uint32_t *sA, uint32_t *sB) {
    uint32_t lo = (uint32_t)seed, hi = (uint32_t)(seed >> 32);
    *sA = oracle_mix32(oracle_mix32(lo + 0x9E3779B9u) ^ hi);
    *sB = oracle_mix32(oracle_mix32(hi + 0x85EBCA6Bu) ^ lo);
}

/* local output row -> global output row (include/spira_hip.h, "Tiling") */
static inline uint32_t oracle_global_row(const spira_params *p, uint32_t r) {
    if (p->stripe_count <= 1) return p->row0 + r;
    return ((r / p->stripe_h) * p->stripe_count + p->stripe_rank) * p->stripe_h + (r % p->stripe_h);
}

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)

/* ---- Float64 instantiation ---- */
#define REAL double
#define SUF(x) CAT(x, _f64)
#define SQRT sqrt
#define FABS fabs
#define TAN tan
/* Julia: deg2rad(z::AbstractFloat) = z * (oftype(z, pi) / 180) */
#define DEG2RAD(x) ((x) * (3.14159265358979323846 / 180.0))
#include "spira_oracle_impl.h"
#undef REAL
#undef SUF
#undef SQRT
#undef FABS
#undef TAN
#undef DEG2RAD

/* ---- Float32 instantiation ---- */
#define REAL float
#define SUF(x) CAT(x, _f32)
#define SQRT sqrtf
#define FABS fabsf
#define TAN tanf
#define DEG2RAD(x) ((x) * ((float)3.14159265358979323846 / 180.0f))
#include "spira_oracle_impl.h"
#undef REAL
#undef SUF
#undef SQRT
#undef FABS
#undef TAN
#undef DEG2RAD

/* Generator streams of the reference's GPU variants, for known-answer tests only. */
/* LCG of src/spira_path_trace_kernel.metal:52-58 */
uint32_t oracle_lcg_next(uint32_t state) { return state * 1664525u + 1013904223u; }
float oracle_lcg_uniform(uint32_t state) { return (float)(state & 0x00FFFFFFu) / (float)0x01000000; }
/* xorshift32 of src/spira-metal-optimized.jl:412-417 and the conversion :420-426
 * (division in Float64, then rounded to Float32) */
uint32_t oracle_xorshift32(uint32_t s) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; }
float oracle_xorshift_uniform(uint32_t s) { return (float)((double)s / 4294967295.0); }

uint32_t oracle_mix32_export(uint32_t x) { return oracle_mix32(x); }
int oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
