// VALU issue-rate microbenchmark (gfx950): wave64 instructions per second and SIMD cycles per instruction for
// independent streams of one opcode at 1..8 waves per SIMD.  Every stream is inline assembly, so the compiler can
// neither fold the arithmetic nor pack two v_fma_f32 into one v_pk_fma_f32 (the SLP vectoriser does that to plain
// C++ — the first version of this file measured v_pk_fma_f32 and reported twice the real v_fma_f32 rate).
// Used to place k_bounce's VALU instruction rate against the real issue ceiling (DESIGN.md §4 "Roofline").
// build: hipcc -O3 --offload-arch=gfx950 valu_peak.hip -o valu_peak
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(op)  op(0) op(1) op(2) op(3) op(4) op(5) op(6) op(7)
#define F32_3(i)  "v_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define PK_3(i)   "v_pk_fma_f32 %" #i ", %" #i ", %4, %5\n"
#define F64_3(i)  "v_fma_f64 %" #i ", %" #i ", %4, %5\n"
#define MULLO(i)  "v_mul_lo_u32 %" #i ", %" #i ", %8\n"
#define MUL24(i)  "v_mul_u32_u24 %" #i ", %" #i ", %8\n"
#define SQRT32(i) "v_sqrt_f32 %" #i ", %" #i "\n"
#define RCP32(i)  "v_rcp_f32 %" #i ", %" #i "\n"
#define RCP64(i)  "v_rcp_f64 %" #i ", %" #i "\n"
#define XOR32(i)  "v_xor_b32 %" #i ", %" #i ", %8\n"
#define CND32(i)  "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define CND64(i)  "v_cndmask_b32_e64 %" #i ", %" #i ", %8, %9\n"
#define MOV32(i)  "v_mov_b32 %" #i ", %8\n"
#define ADD32(i)  "v_add_f32 %" #i ", %" #i ", %8\n"
#define MUL32(i)  "v_mul_f32 %" #i ", %" #i ", %8\n"
#define CMP32(i)  "v_cmp_lt_f32 vcc, %" #i ", %8\n"
#define CMP32S(i) "v_cmp_lt_f32_e64 %9, %" #i ", %8\n"
#define ADD64(i)  "v_add_f64 %" #i ", %" #i ", %4\n"
#define MUL64(i)  "v_mul_f64 %" #i ", %" #i ", %4\n"
#define CMP64(i)  "v_cmp_lt_f64 vcc, %" #i ", %4\n"
#define DSC32(i)  "v_div_scale_f32 %" #i ", vcc, %" #i ", %8, %" #i "\n"
#define DFM32(i)  "v_div_fmas_f32 %" #i ", %" #i ", %8, %8\n"
#define DFX32(i)  "v_div_fixup_f32 %" #i ", %" #i ", %8, %8\n"
#define DSC64(i)  "v_div_scale_f64 %" #i ", vcc, %" #i ", %4, %" #i "\n"
#define DFM64(i)  "v_div_fmas_f64 %" #i ", %" #i ", %4, %4\n"
#define DFX64(i)  "v_div_fixup_f64 %" #i ", %" #i ", %4, %4\n"
#define RSQ64(i)  "v_rsq_f64 %" #i ", %" #i "\n"
#define LSHR(i)   "v_lshrrev_b32 %" #i ", 15, %" #i "\n"
#define CVT(i)    "v_cvt_f32_u32 %" #i ", %" #i "\n"
#define CVT64(i)  "v_cvt_f64_u32 %" #i ", %8\n"
#define LDEXP64(i) "v_ldexp_f64 %" #i ", %" #i ", %8\n"
#define MOV64(i)  "v_mov_b64 %" #i ", %4\n"
#define CNDV64(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %8, vcc\n"
#define ADDC(i)   "v_addc_co_u32 %" #i ", vcc, %" #i ", %8, vcc\n"
#define REP4(op)  op(0) op(1) op(2) op(3)

template <int KIND>
__global__ void k(float *out, int iters) {
    float a0 = threadIdx.x * 1e-3f + 1, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3;
    unsigned u0 = threadIdx.x, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3, u4 = u0 + 4, u5 = u0 + 5, u6 = u0 + 6, u7 = u0 + 7;
    const float c1 = 1.0001f, c2 = 0.5f;
    const double e1 = 1.0001, e2 = 0.5;
    const unsigned m = 0x7feb352du;
    unsigned long long mask = 0x5555555555555555ull + (unsigned long long)iters;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (KIND == 0) asm volatile(REP8(F32_3) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c1), "v"(c2));
            if (KIND == 1) asm volatile(REP4(PK_3) : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(e1), "v"(e2));        // 64-bit register pairs = 2 x f32
            if (KIND == 2) asm volatile(REP4(F64_3) : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(e1), "v"(e2));
            if (KIND == 3) asm volatile(REP8(MULLO) : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(m));
            if (KIND == 4) asm volatile(REP8(MUL24) : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(m));
            if (KIND == 5) asm volatile(REP8(SQRT32) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            if (KIND == 6) asm volatile(REP8(RCP32) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            if (KIND == 7) asm volatile(REP4(RCP64) : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));
            if (KIND == 8) asm volatile(REP8(XOR32) : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(m));
#define U8 "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7)
#define A8 "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
#define D4 "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3)
            if (KIND == 10) asm volatile(REP8(CND64) : U8 : "v"(m), "s"(mask));
            if (KIND == 11) asm volatile(REP8(MOV32) : U8 : "v"(m));
            if (KIND == 12) asm volatile(REP8(ADD32) : A8 : "v"(c2));
            if (KIND == 13) asm volatile(REP8(MUL32) : A8 : "v"(c1));
            if (KIND == 14) asm volatile(REP8(CMP32) : A8 : "v"(c1) : "vcc");
            if (KIND == 15) asm volatile(REP4(ADD64) : D4 : "v"(e2));
            if (KIND == 16) asm volatile(REP4(MUL64) : D4 : "v"(e1));
            if (KIND == 17) asm volatile(REP4(CMP64) : D4 : "v"(e1) : "vcc");
            if (KIND == 18) asm volatile(REP8(DSC32) : A8 : "v"(c1) : "vcc");
            if (KIND == 19) asm volatile(REP8(DFM32) : A8 : "v"(c1) : "vcc");
            if (KIND == 20) asm volatile(REP8(DFX32) : A8 : "v"(c1));
            if (KIND == 21) asm volatile(REP4(DSC64) : D4 : "v"(e1) : "vcc");
            if (KIND == 22) asm volatile(REP4(DFM64) : D4 : "v"(e1) : "vcc");
            if (KIND == 23) asm volatile(REP4(DFX64) : D4 : "v"(e1));
            if (KIND == 24) asm volatile(REP4(RSQ64) : D4);
            if (KIND == 25) asm volatile(REP8(LSHR) : U8);
            if (KIND == 26) asm volatile(REP8(CVT) : U8);
            if (KIND == 27) asm volatile(REP4(CVT64) : D4, "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4) : );
            if (KIND == 28) asm volatile(REP4(LDEXP64) : D4, "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4));
            if (KIND == 29) asm volatile(REP4(MOV64) : D4 : "v"(e1));
            if (KIND == 30) asm volatile(REP8(CMP32S) : A8 : "v"(c1), "s"(mask));
            if (KIND == 31) asm volatile(REP8(CNDV64) : U8 : "v"(m) : "vcc");
            if (KIND == 32) asm volatile("v_cmp_lt_u32 vcc, %0, %8\n" REP8(CND32) : U8 : "v"(m) : "vcc");
            if (KIND == 33) asm volatile("s_mov_b64 vcc, %9\n" REP8(CND32) : U8 : "v"(m), "s"(mask) : "vcc");
            if (KIND == 9) asm volatile(REP8(CND32) : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(m) : "vcc");
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(d0 + d1 + d2 + d3) + (float)(u0 + u1 + u2 + u3 + u4 + u5 + u6 + u7);
}

template <int KIND> void run(const char *name, int per_iter, float *d) {
    for (int wps : {1, 4, 8}) {
        int blocks = 256 * wps;   // 256-thread blocks = 4 waves = 1 per SIMD; wps blocks per CU
        int iters = 4000;
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 100);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        double winstr = (double)blocks * 4 * iters * per_iter;
        printf("%-22s waves/SIMD=%d  %8.3f ms  %.3e wave-instr/s  = %5.2f cycles/instr/SIMD @2.4GHz\n", name, wps, ms, winstr / (ms * 1e-3),
               1024 * 2.4e9 / (winstr / (ms * 1e-3)));
    }
}

int main() {
    float *d; (void)hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
    run<0>("v_fma_f32", 64, d);
    run<1>("v_pk_fma_f32", 32, d);
    run<2>("v_fma_f64", 32, d);
    run<3>("v_mul_lo_u32", 64, d);
    run<4>("v_mul_u32_u24", 64, d);
    run<5>("v_sqrt_f32", 64, d);
    run<6>("v_rcp_f32", 64, d);
    run<7>("v_rcp_f64", 32, d);
    run<8>("v_xor_b32", 64, d);
    run<9>("v_cndmask_b32 (vcc)", 64, d);
    run<10>("v_cndmask_b32 (sgpr)", 64, d);
    run<31>("v_cndmask_e64 (vcc)", 64, d);
    run<32>("v_cmp+8 cndmask_e32", 72, d);
    run<33>("s_mov vcc+8 cndmask", 64, d);
    run<11>("v_mov_b32", 64, d);
    run<12>("v_add_f32", 64, d);
    run<13>("v_mul_f32", 64, d);
    run<14>("v_cmp_lt_f32 vcc", 64, d);
    run<30>("v_cmp_lt_f32 sgpr", 64, d);
    run<15>("v_add_f64", 32, d);
    run<16>("v_mul_f64", 32, d);
    run<17>("v_cmp_lt_f64", 32, d);
    run<18>("v_div_scale_f32", 64, d);
    run<19>("v_div_fmas_f32", 64, d);
    run<20>("v_div_fixup_f32", 64, d);
    run<21>("v_div_scale_f64", 32, d);
    run<22>("v_div_fmas_f64", 32, d);
    run<23>("v_div_fixup_f64", 32, d);
    run<24>("v_rsq_f64", 32, d);
    run<25>("v_lshrrev_b32", 64, d);
    run<26>("v_cvt_f32_u32", 64, d);
    run<27>("v_cvt_f64_u32", 32, d);
    run<28>("v_ldexp_f64", 32, d);
    run<29>("v_mov_b64", 32, d);
    return 0;
}
