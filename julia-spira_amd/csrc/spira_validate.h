// spira_validate.h — host-side checks of the caller's scene arrays (no HIP headers: also built into the sanitizer harness
// tests/native/host_sanitize.cpp).  Returns 0 or a negative SPIRA_E_* code and a static message.
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>

namespace spira {

template <class T>
int scene_arrays_check(const T *spheres5, const T *materials8, const T *triangles10, uint32_t n_spheres, uint32_t n_materials, uint32_t nt,
                       const char **msg) {
    constexpr int kInvalid = -1;
    auto bad = [&](const char *m) { *msg = m; return kInvalid; };
    if (!materials8) return bad("materials8 is NULL");
    if (n_spheres && !spheres5) return bad("spheres5 is NULL");
    if (nt && !triangles10) return bad("triangles10 is NULL");
    if (n_materials < 1) return bad("n_materials must be >= 1");
    for (uint32_t i = 0; i < n_spheres; ++i) {
        const T *s = spheres5 + 5 * (size_t)i;
        if (!(std::isfinite(s[0]) && std::isfinite(s[1]) && std::isfinite(s[2]) && std::isfinite(s[3]))) return bad("sphere with a non-finite centre or radius");
        const T m = s[4];
        if (!(m >= 1 && m <= (T)n_materials) || m != std::floor(m)) return bad("sphere material index out of range (1-based, stored as a float)");
    }
    for (uint32_t i = 0; i < nt; ++i) {
        const T *t = triangles10 + 10 * (size_t)i;
        for (int k = 0; k < 9; ++k)
            if (!std::isfinite(t[k])) return bad("triangle with a non-finite vertex");      // (the BVH builder bins centroids: inf / NaN has no bin)
        const T m = t[9];
        if (!(m >= 1 && m <= (T)n_materials) || m != std::floor(m)) return bad("triangle material index out of range");
    }
    for (uint32_t i = 0; i < 8 * (size_t)n_materials; ++i)
        if (std::isnan(materials8[i])) return bad("material with a NaN field");
    return 0;
}

}  // namespace spira
