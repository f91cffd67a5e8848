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

// "Ordinary magnitudes": every coordinate is zero or within 2^-20 .. 2^20 (Float32) / 2^-64 .. 2^64 (Float64) and every radius within
// that range.  Not a validity rule — any finite scene renders, with the same results — but the predictor of whether k_path's
// speculative division (spira_device.h, SpecDiv) will have to render waves a second time.
template <class T> inline bool magnitude_moderate(T v, bool zero_ok) {
    const T lo = sizeof(T) == 8 ? (T)5.421010862427522e-20 : (T)9.5367431640625e-07, hi = (T)1 / lo;
    const T m = std::fabs(v);
    return (zero_ok && m == 0) || (m >= lo && m <= hi);
}
template <class T>
bool scene_scale_moderate(const T *spheres5, const T *triangles10, uint32_t n_spheres, uint32_t nt) {
    for (uint32_t i = 0; i < n_spheres; ++i) {
        const T *s = spheres5 + 5 * (size_t)i;
        if (!(magnitude_moderate(s[0], true) && magnitude_moderate(s[1], true) && magnitude_moderate(s[2], true) && magnitude_moderate(s[3], false))) return false;
    }
    for (uint32_t i = 0; i < nt; ++i)
        for (int k = 0; k < 9; ++k)
            if (!magnitude_moderate(triangles10[10 * (size_t)i + k], true)) return false;
    return true;
}
template <class T>
bool camera_scale_moderate(const T *camera12) {
    for (int k = 0; k < 12; ++k)
        if (!magnitude_moderate(camera12[k], true)) return false;
    return true;
}

}  // namespace spira
