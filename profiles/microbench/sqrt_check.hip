// Bit-exactness check of julia-spira_amd/csrc/spira_sqrt.h on the device: the range-checked refinement
// (sqrt_core) against the compiler's correctly rounded expansion, over ALL 2^32 Float32 bit patterns and over 2^32
// Float64 inputs (every exponent, hashed mantissas).  Prints the number of mismatches (must be 0) and how many
// inputs took the fast path.   build: hipcc -O3 --offload-arch=gfx950 -I../../julia-spira_amd/csrc sqrt_check.hip -o sqrt_check
#include "spira_sqrt_experiment.h"
#include <cstdio>

__device__ __forceinline__ uint64_t mix64(uint64_t z) { z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31); }

__global__ void check32(unsigned long long *bad, unsigned long long *fast, uint32_t base) {
    const uint32_t bits = base + blockIdx.x * blockDim.x + threadIdx.x;
    const float x = __uint_as_float(bits);
    const bool in = spira::sqrt_in_range(x);
    const float want = __builtin_sqrtf(x);
    const float got = in ? spira::sqrt_core(x) : want;
    if (__float_as_uint(got) != __float_as_uint(want)) atomicAdd(bad, 1ull);
    const unsigned long long m = __ballot(in);
    if ((threadIdx.x & 63) == 0 && (blockIdx.x & 255) == 0) atomicAdd(fast, 256ull * __popcll(m));   // sampled count
}
__global__ void check64(unsigned long long *bad, unsigned long long *fast, uint64_t base) {
    const uint64_t i = base + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    // exponent from the low 11 bits of the index (all 2048, sign from bit 11), mantissa hashed; every 4096th input is a
    // perfect square or its neighbour (the hard cases for rounding)
    uint64_t bits = ((i & 0x7FFull) << 52) | ((i >> 11 & 1ull) << 63) | (mix64(i) & 0xFFFFFFFFFFFFFull);
    double x = __longlong_as_double((long long)bits);
    if ((i & 0xFFF000ull) == 0) { const double r = (double)(uint32_t)mix64(i ^ 0x5bd1e995u); x = r * r; if (i & 1) x = __longlong_as_double(__double_as_longlong(x) + ((i & 2) ? 1 : -1)); }
    const bool in = spira::sqrt_in_range(x);
    const double want = __builtin_sqrt(x);
    const double got = in ? spira::sqrt_core(x) : want;
    if (__double_as_longlong(got) != __double_as_longlong(want)) atomicAdd(bad, 1ull);
    const unsigned long long m = __ballot(in);
    if ((threadIdx.x & 63) == 0 && (blockIdx.x & 255) == 0) atomicAdd(fast, 256ull * __popcll(m));
}

int main() {
    unsigned long long *d, h[4] = {0, 0, 0, 0};
    (void)hipMalloc(&d, sizeof(h)); (void)hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    for (uint32_t part = 0; part < 16; ++part) hipLaunchKernelGGL(check32, dim3(1u << 20), dim3(256), 0, 0, d, d + 1, part << 28);
    for (uint64_t part = 0; part < 16; ++part) hipLaunchKernelGGL(check64, dim3(1u << 20), dim3(256), 0, 0, d + 2, d + 3, part << 28);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("f32: %llu mismatches over 2^32 inputs, about %llu on the fast path\n", h[0], h[1]);
    printf("f64: %llu mismatches over 2^32 inputs, about %llu on the fast path\n", h[2], h[3]);
    return (h[0] || h[2]) ? 1 : 0;
}
