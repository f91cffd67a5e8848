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
#define MAX32(i)  "v_max_f32 %" #i ", %" #i ", %8\n"
#define MAX3F(i)  "v_max3_f32 %" #i ", %" #i ", %8, %9\n"
#define MIN3U(i)  "v_min3_u32 %" #i ", %" #i ", %8, %8\n"
#define CVTUB(i)  "v_cvt_f32_ubyte1 %" #i ", %" #i "\n"
#define AND32(i)  "v_and_b32 %" #i ", %" #i ", %8\n"
#define LSHLOR(i) "v_lshl_or_b32 %" #i ", %" #i ", 3, %8\n"
#define BFE(i)    "v_bfe_u32 %" #i ", %" #i ", 8, 8\n"
#define PERM(i)   "v_perm_b32 %" #i ", %" #i ", %8, %8\n"
#define ADDU(i)   "v_add_u32 %" #i ", %" #i ", %8\n"
#define MULHI(i)  "v_mul_hi_u32 %" #i ", %" #i ", %8\n"
#define MAX64(i)  "v_max_f64 %" #i ", %" #i ", %4\n"
#define CVT3264(i) "v_cvt_f32_f64 %" #i ", %8\n"
#define SQRT64(i) "v_sqrt_f64 %" #i ", %" #i "\n"
#define FMAC32(i) "v_fmac_f32 %" #i ", %8, %9\n"
#define MBCNT(i)  "v_mbcnt_lo_u32_b32 %" #i ", %8, %" #i "\n"
#define FFBL(i)   "v_ffbl_b32 %" #i ", %" #i "\n"
#define MAD24(i)  "v_mad_u32_u24 %" #i ", %" #i ", %8, %8\n"
#define BFI(i)    "v_bfi_b32 %" #i ", %8, %" #i ", %8\n"
#define REP4(op)  op(0) op(1) op(2) op(3)
// mixed streams: every other instruction a v_fma_f32 (Float64 rows: v_fma_f64); the partner's MARGINAL cost = pair - the v_fma alone — what the opcode costs inside
// real code, where a same-opcode stream's serialisation (compares writing one SGPR pair, selects reading it ...) does not occur
#define PAIR32(X) "v_fma_f32 %0, %0, %16, %17\n" X(8) "v_fma_f32 %1, %1, %16, %17\n" X(9) "v_fma_f32 %2, %2, %16, %17\n" X(10) "v_fma_f32 %3, %3, %16, %17\n" X(11) \
                  "v_fma_f32 %4, %4, %16, %17\n" X(12) "v_fma_f32 %5, %5, %16, %17\n" X(13) "v_fma_f32 %6, %6, %16, %17\n" X(14) "v_fma_f32 %7, %7, %16, %17\n" X(15)
#define P_CMPS(i)  "v_cmp_lt_f32_e64 %19, %" #i ", %16\n"
#define P_CMPV(i)  "v_cmp_lt_f32 vcc, %" #i ", %16\n"
#define P_CND(i)   "v_cndmask_b32_e64 %" #i ", %" #i ", %18, %19\n"
#define P_MOV(i)   "v_mov_b32 %" #i ", %18\n"
#define P_XOR(i)   "v_xor_b32 %" #i ", %" #i ", %18\n"
#define P_ADDU(i)  "v_add_u32 %" #i ", %" #i ", %18\n"
#define P_MAX(i)   "v_max_f32 %" #i ", %" #i ", %16\n"
#define P_MULLO(i) "v_mul_lo_u32 %" #i ", %" #i ", %18\n"
#define P_CVTUB(i) "v_cvt_f32_ubyte1 %" #i ", %" #i "\n"
#define P_PERM(i)  "v_perm_b32 %" #i ", %" #i ", %18, %18\n"
#define P_LSHR(i)  "v_lshrrev_b32 %" #i ", 15, %" #i "\n"
#define P_FMA(i)   "v_fma_f32 %" #i ", %" #i ", %16, %17\n"
#define P_RCP(i)   "v_rcp_f32 %" #i ", %" #i "\n"

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
            if (KIND == 40) asm volatile(REP8(MAX32) : A8 : "v"(c1));
            if (KIND == 41) asm volatile(REP8(MAX3F) : A8 : "v"(c1), "v"(c2));
            if (KIND == 42) asm volatile(REP8(MIN3U) : U8 : "v"(m));
            if (KIND == 43) asm volatile(REP8(CVTUB) : U8);
            if (KIND == 44) asm volatile(REP8(AND32) : U8 : "v"(m));
            if (KIND == 45) asm volatile(REP8(LSHLOR) : U8 : "v"(m));
            if (KIND == 46) asm volatile(REP8(BFE) : U8);
            if (KIND == 47) asm volatile(REP8(PERM) : U8 : "v"(m));
            if (KIND == 48) asm volatile(REP8(ADDU) : U8 : "v"(m));
            if (KIND == 49) asm volatile(REP8(MULHI) : U8 : "v"(m));
            if (KIND == 50) asm volatile(REP4(MAX64) : D4 : "v"(e1));
            if (KIND == 51) asm volatile(REP8(CVT3264) : A8, "+v"(d0) : );
            if (KIND == 52) asm volatile(REP4(SQRT64) : D4);
            if (KIND == 53) asm volatile(REP8(FMAC32) : A8 : "v"(c1), "v"(c2));
            if (KIND == 54) asm volatile(REP8(MBCNT) : U8 : "v"(m));
            if (KIND == 55) asm volatile(REP8(FFBL) : U8);
            if (KIND == 56) asm volatile(REP8(MAD24) : U8 : "v"(m));
            if (KIND == 57) asm volatile(REP8(BFI) : U8 : "v"(m));
#define PAIRARGS : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(c1), "v"(c2), "v"(m), "s"(mask)
            if (KIND == 60) asm volatile(PAIR32(P_CMPS) PAIRARGS);
            if (KIND == 61) asm volatile(PAIR32(P_CMPV) PAIRARGS : "vcc");
            if (KIND == 62) asm volatile(PAIR32(P_CND) PAIRARGS);
            if (KIND == 63) asm volatile(PAIR32(P_MOV) PAIRARGS);
            if (KIND == 64) asm volatile(PAIR32(P_XOR) PAIRARGS);
            if (KIND == 65) asm volatile(PAIR32(P_ADDU) PAIRARGS);
            if (KIND == 66) asm volatile(PAIR32(P_MAX) PAIRARGS);
            if (KIND == 67) asm volatile(PAIR32(P_MULLO) PAIRARGS);
            if (KIND == 68) asm volatile(PAIR32(P_CVTUB) PAIRARGS);
            if (KIND == 69) asm volatile(PAIR32(P_PERM) PAIRARGS);
            if (KIND == 70) asm volatile(PAIR32(P_LSHR) PAIRARGS);
            if (KIND == 71) asm volatile(PAIR32(P_FMA) PAIRARGS);
            if (KIND == 72) asm volatile(PAIR32(P_RCP) PAIRARGS);
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
    run<40>("v_max_f32", 64, d);
    run<41>("v_max3_f32", 64, d);
    run<42>("v_min3_u32", 64, d);
    run<43>("v_cvt_f32_ubyte1", 64, d);
    run<44>("v_and_b32", 64, d);
    run<45>("v_lshl_or_b32", 64, d);
    run<46>("v_bfe_u32", 64, d);
    run<47>("v_perm_b32", 64, d);
    run<48>("v_add_u32", 64, d);
    run<49>("v_mul_hi_u32", 64, d);
    run<50>("v_max_f64", 32, d);
    run<51>("v_cvt_f32_f64", 64, d);
    run<52>("v_sqrt_f64", 32, d);
    run<53>("v_fmac_f32", 64, d);
    run<54>("v_mbcnt_lo_u32_b32", 64, d);
    run<55>("v_ffbl_b32", 64, d);
    run<56>("v_mad_u32_u24", 64, d);
    run<57>("v_bfi_b32", 64, d);
    // pairs: cycles per PAIR (one v_fma_f32 + one partner); the partner's marginal cost = this - v_fma_f32's row
    run<71>("pair:fma+v_fma_f32", 64, d);
    run<60>("pair:fma+v_cmp_lt_f32_sgpr", 64, d);
    run<61>("pair:fma+v_cmp_lt_f32_vcc", 64, d);
    run<62>("pair:fma+v_cndmask_b32", 64, d);
    run<63>("pair:fma+v_mov_b32", 64, d);
    run<64>("pair:fma+v_xor_b32", 64, d);
    run<65>("pair:fma+v_add_u32", 64, d);
    run<66>("pair:fma+v_max_f32", 64, d);
    run<67>("pair:fma+v_mul_lo_u32", 64, d);
    run<68>("pair:fma+v_cvt_f32_ubyte1", 64, d);
    run<69>("pair:fma+v_perm_b32", 64, d);
    run<70>("pair:fma+v_lshrrev_b32", 64, d);
    run<72>("pair:fma+v_rcp_f32", 64, d);
    return 0;
}
