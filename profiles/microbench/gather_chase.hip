// Divergent pointer-chase microbenchmark (gfx950): what does ONE step of a per-lane BVH walk cost?
// Every active lane fetches NB x 16 bytes at its own (data-dependent) record of a table and derives the next
// record from what it read — the memory shape of bvh_closest_hit (spira_device.h).  Sweeps: record bytes
// (16..128), active lanes per wave (64, 32, 16, 4, 1: the straggler regime of a traversal batch), table size
// (L1-, L2-, Infinity-Cache-, HBM-resident) and the table in LDS instead.  Output: nanoseconds per wave-step per CU
// slot (= time x CUs / (waves x steps)); with W waves per CU the latency one wave sees per step is W times that.
// build: hipcc -O3 --offload-arch=gfx950 gather_chase.hip -o gather_chase
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

template <int NB>
__global__ __launch_bounds__(256) void chase_global(const uint4 *tab, uint32_t mask, int steps, int active, uint32_t *out) {
    const uint32_t lane = threadIdx.x & 63;
    if ((int)lane >= active) return;
    uint32_t idx = mix(blockIdx.x * 256 + threadIdx.x) & mask;
    uint32_t acc = 0;
    for (int s = 0; s < steps; ++s) {
        uint32_t h = 0;
#pragma unroll
        for (int k = 0; k < NB; ++k) { const uint4 v = tab[(size_t)idx * NB + k]; h += v.x ^ v.y ^ v.z ^ v.w; }
        acc += h;
        idx = mix(h + s) & mask;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int NB>
__global__ __launch_bounds__(256) void chase_lds(const uint4 *tab, uint32_t mask, int steps, int active, uint32_t *out) {
    extern __shared__ uint4 lds[];
    for (uint32_t i = threadIdx.x; i < (mask + 1) * NB; i += 256) lds[i] = tab[i];
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63;
    if ((int)lane >= active) return;
    uint32_t idx = mix(blockIdx.x * 256 + threadIdx.x) & mask;
    uint32_t acc = 0;
    for (int s = 0; s < steps; ++s) {
        uint32_t h = 0;
#pragma unroll
        for (int k = 0; k < NB; ++k) { const uint4 v = lds[(size_t)idx * NB + k]; h += v.x ^ v.y ^ v.z ^ v.w; }
        acc += h;
        idx = mix(h + s) & mask;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

// coherent variant: all lanes of a wave walk the SAME record chain (a wave-uniform node): the cost of a broadcast fetch
template <int NB>
__global__ __launch_bounds__(256) void chase_uniform(const uint4 *tab, uint32_t mask, int steps, int active, uint32_t *out) {
    const uint32_t lane = threadIdx.x & 63;
    if ((int)lane >= active) return;
    uint32_t idx = mix(blockIdx.x * 4 + (threadIdx.x >> 6)) & mask;
    uint32_t acc = 0;
    for (int s = 0; s < steps; ++s) {
        uint32_t h = 0;
#pragma unroll
        for (int k = 0; k < NB; ++k) { const uint4 v = tab[(size_t)idx * NB + k]; h += v.x ^ v.y ^ v.z ^ v.w; }
        acc += h;
        idx = mix(h + s) & mask;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int NB>
static void run(const char *kind, int mode, const uint4 *tab, size_t table_bytes, int wgs_per_cu, int steps, int active, uint32_t *out) {
    const int cus = 256;
    const uint32_t n_rec = (uint32_t)(table_bytes / (16 * NB));
    uint32_t p2 = 1; while (p2 * 2 <= n_rec) p2 *= 2;
    const uint32_t mask = p2 - 1;
    const int grid = cus * wgs_per_cu;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        if (mode == 0) chase_global<NB><<<grid, 256>>>(tab, mask, steps, active, out);
        else if (mode == 1) chase_lds<NB><<<grid, 256, (size_t)p2 * NB * 16>>>(tab, mask, steps, active, out);
        else chase_uniform<NB><<<grid, 256>>>(tab, mask, steps, active, out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double waves_per_cu = wgs_per_cu * 4.0;
    const double ns_per_wave_step_per_cu = best * 1e6 / (waves_per_cu * steps);
    printf("%-8s rec=%3dB table=%9zuB waves/CU=%2.0f active=%2d  %8.3f ms  %7.1f ns/wave-step/CU  (one wave sees %8.1f ns/step)  %6.2f Glane-steps/s\n",
           kind, NB * 16, (size_t)p2 * NB * 16, waves_per_cu, active, best, ns_per_wave_step_per_cu, ns_per_wave_step_per_cu * waves_per_cu,
           (double)grid * 4 * active * steps / (best * 1e6));
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
}

int main() {
    const size_t max_bytes = 1ull << 30;
    std::vector<uint32_t> h(max_bytes / 4);
    uint32_t s = 12345u;
    for (auto &v : h) { s = s * 1664525u + 1013904223u; v = s; }
    uint4 *tab; uint32_t *out;
    CK(hipMalloc(&tab, max_bytes)); CK(hipMalloc(&out, 256 * 8 * 256 * 4));
    CK(hipMemcpy(tab, h.data(), max_bytes, hipMemcpyHostToDevice));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(chase_lds<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(chase_lds<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(chase_lds<5>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const int steps = 2000;
    const size_t sizes[] = {16u << 10, 1u << 20, 6u << 20, 64u << 20, 1u << 30};
    const int actives[] = {64, 32, 16, 4, 1};
    printf("== global, 5 WG/CU (20 waves/CU)\n");
    for (size_t sz : sizes)
        for (int a : actives) {
            run<1>("global", 0, tab, sz, 5, steps, a, out);
            run<2>("global", 0, tab, sz, 5, steps, a, out);
            run<4>("global", 0, tab, sz, 5, steps, a, out);
            run<5>("global", 0, tab, sz, 5, steps, a, out);
            run<8>("global", 0, tab, sz, 5, steps, a, out);
        }
    printf("== global, 2 WG/CU (8 waves/CU) and 1 WG/CU\n");
    for (size_t sz : {size_t(1u << 20), size_t(6u << 20)})
        for (int a : {64, 4}) {
            run<4>("global", 0, tab, sz, 2, steps, a, out);
            run<4>("global", 0, tab, sz, 1, steps, a, out);
            run<5>("global", 0, tab, sz, 2, steps, a, out);
        }
    printf("== LDS table (one copy per workgroup)\n");
    for (int a : actives) {
        run<1>("lds", 1, tab, 32u << 10, 4, steps, a, out);
        run<4>("lds", 1, tab, 32u << 10, 4, steps, a, out);
        run<5>("lds", 1, tab, 32u << 10, 4, steps, a, out);
        run<4>("lds", 1, tab, 64u << 10, 2, steps, a, out);
    }
    printf("== wave-uniform record (all lanes the same address)\n");
    for (size_t sz : {size_t(1u << 20), size_t(6u << 20)})
        for (int a : {64, 1}) {
            run<4>("uniform", 2, tab, sz, 5, steps, a, out);
            run<5>("uniform", 2, tab, sz, 5, steps, a, out);
        }
    return 0;
}
