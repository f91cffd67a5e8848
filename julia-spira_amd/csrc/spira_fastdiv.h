// spira_fastdiv.h — division by a launch constant (host + device; no HIP headers, so host-only tools can include it).
// Granlund-Montgomery round-up method, exact for every uint32 n and d >= 1 (one v_mul_hi_u32 instead of the
// ~20-instruction udiv expansion; path_of runs per segment).
#pragma once
#include <stdint.h>
#if defined(__HIPCC__)
#define SPIRA_HD __host__ __device__
#else
#define SPIRA_HD
#endif

namespace spira {

struct FastDiv { uint32_t magic, sh1, sh2; };
inline FastDiv fastdiv_make(uint32_t d) {
    FastDiv f;
    uint32_t l = 0;
    while (l < 32 && (1ull << l) < d) ++l;                       // l = ceil(log2 d)
    f.magic = (uint32_t)((((1ull << l) - d) << 32) / d + 1);
    f.sh1 = l < 1 ? l : 1;
    f.sh2 = l ? l - 1 : 0;
    return f;
}
SPIRA_HD inline uint32_t fastdiv(uint32_t n, const FastDiv &f) {
#ifdef __HIP_DEVICE_COMPILE__
    uint32_t t = __umulhi(f.magic, n);
#else
    uint32_t t = (uint32_t)(((uint64_t)f.magic * n) >> 32);
#endif
    return (t + ((n - t) >> f.sh1)) >> f.sh2;
}

}  // namespace spira
