// VALU issue-rate microbenchmark (gfx950): wave-instructions/s for independent v_fma_f32 / v_mul_lo_u32 /
// v_sqrt_f32 streams at 1..8 waves per SIMD.  Used to place k_bounce's VALU instruction rate against the
// real issue ceiling (DESIGN.md "Roofline").  build: hipcc -O3 --offload-arch=gfx950 valu_peak.hip -o valu_peak
#include <hip/hip_runtime.h>
#include <cstdio>
template <int KIND>
__global__ void k(float *out, int iters) {
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    unsigned u0 = threadIdx.x, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3, u4 = u0 + 4, u5 = u0 + 5, u6 = u0 + 6, u7 = u0 + 7;
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                a0 = __builtin_fmaf(a0, 1.0001f, 0.5f); a1 = __builtin_fmaf(a1, 1.0001f, 0.5f); a2 = __builtin_fmaf(a2, 1.0001f, 0.5f); a3 = __builtin_fmaf(a3, 1.0001f, 0.5f);
                a4 = __builtin_fmaf(a4, 1.0001f, 0.5f); a5 = __builtin_fmaf(a5, 1.0001f, 0.5f); a6 = __builtin_fmaf(a6, 1.0001f, 0.5f); a7 = __builtin_fmaf(a7, 1.0001f, 0.5f);
            }
        } else if (KIND == 1) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                u0 *= 0x7feb352du; u1 *= 0x7feb352du; u2 *= 0x7feb352du; u3 *= 0x7feb352du; u4 *= 0x7feb352du; u5 *= 0x7feb352du; u6 *= 0x7feb352du; u7 *= 0x7feb352du;
            }
        } else if (KIND == 2) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                a0 = __builtin_amdgcn_sqrtf(a0); a1 = __builtin_amdgcn_sqrtf(a1); a2 = __builtin_amdgcn_sqrtf(a2); a3 = __builtin_amdgcn_sqrtf(a3);
                a4 = __builtin_amdgcn_sqrtf(a4); a5 = __builtin_amdgcn_sqrtf(a5); a6 = __builtin_amdgcn_sqrtf(a6); a7 = __builtin_amdgcn_sqrtf(a7);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {   // xor-shift: two full-rate int ops per line element
                u0 ^= u0 >> 15; u1 ^= u1 >> 15; u2 ^= u2 >> 15; u3 ^= u3 >> 15; u4 ^= u4 >> 15; u5 ^= u5 >> 15; u6 ^= u6 >> 15; u7 ^= u7 >> 15;
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(u0 + u1 + u2 + u3 + u4 + u5 + u6 + u7);
}
template <int KIND> void run(const char *name, int per_iter, float *d) {
    for (int wps : {1, 2, 4, 8}) {
        int blocks = 256 * wps;   // 256-thread blocks = 4 waves = 1 per SIMD; wps blocks per CU
        int iters = 20000;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 100);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double winstr = (double)blocks * 4 * iters * per_iter;
        printf("%-12s waves/SIMD=%d  %.3f ms  %.3e wave-instr/s  = %.2f cycles/instr/SIMD @2.4GHz\n", name, wps, ms, winstr / (ms * 1e-3),
               1024 * 2.4e9 / (winstr / (ms * 1e-3)));
    }
}
int main() {
    float *d; hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
    run<0>("v_fma_f32", 64, d);
    run<1>("v_mul_lo_u32", 64, d);
    run<2>("v_sqrt_f32", 64, d);
    run<3>("xorshift", 128, d);
    return 0;
}
