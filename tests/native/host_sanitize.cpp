// host_sanitize.cpp — host-only harness built with -fsanitize=address,undefined (tests/test_native_sanitize.py).
// GPU AddressSanitizer is not available on the pool, so everything of the product that runs on the HOST is exercised here
// under ASan + UBSan: the BVH builder (spira_bvh.h) over degenerate meshes, the scene validation (spira_validate.h), the
// magic-number division (spira_fastdiv.h) and the triangle hash; the CPU oracle is linked in and run under the sanitizers too.
// The tree is checked semantically: a host traversal with the kernels' rules (8-wide quantised nodes, Float32 slab tests in the mesh's
// normalised frame, octant-ordered descent, leaf = Moller-Trumbore in T, ties to the later triangle) must return exactly what a
// linear scan over the caller's array returns.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../../julia-spira_amd/csrc/spira_bvh.h"
#include "../../julia-spira_amd/csrc/spira_fastdiv.h"
#include "../../julia-spira_amd/csrc/spira_validate.h"
#include "../../include/spira_hip.h"

#ifndef SPIRA_NO_ORACLE
extern "C" {
int oracle_render_f64(const double *, const double *, const double *, const double *, const spira_params *, double *, double *, int, uint64_t *);
int oracle_render_f32(const float *, const float *, const float *, const float *, const spira_params *, float *, float *, int, uint64_t *);
int oracle_render_variant_f32(const float *, const float *, const float *, const spira_params *, float *, float *, int, uint64_t *);
}
#endif

static int g_fail = 0;
#define CHECK(c) do { if (!(c)) { std::fprintf(stderr, "CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #c); ++g_fail; } } while (0)

template <class T> struct V { T x, y, z; };
template <class T> static V<T> sub(V<T> a, V<T> b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
template <class T> static V<T> cross(V<T> a, V<T> b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
template <class T> static T dot(V<T> a, V<T> b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

// examples/julia-raytracer.jl:145-187 on precomputed edges (same statements as triangle_test of spira_device.h)
template <class T> static bool tri_test(V<T> v0, V<T> e1, V<T> e2, V<T> o, V<T> d, T t_min, T t_max, T &t_out) {
    V<T> h = cross(d, e2);
    T a = dot(e1, h);
    if (std::fabs(a) < (T)1e-8) return false;
    T f = (T)1.0 / a;
    V<T> s = sub(o, v0);
    T u = f * dot(s, h);
    if (u < 0 || u > 1) return false;
    V<T> q = cross(s, e1);
    T v = f * dot(d, q);
    if (v < 0 || u + v > 1) return false;
    T t = f * dot(e2, q);
    if (t < t_min || t > t_max) return false;
    t_out = t;
    return true;
}

template <class T> static uint32_t bits_of(T w) {
    if (sizeof(T) == 4) { uint32_t u; std::memcpy(&u, &w, 4); return u; }
    uint64_t u; std::memcpy(&u, &w, 8); return (uint32_t)u;
}

// Host mirror of the kernels' walk (spira_device.h: bvh8_enter / bvh8_node / bvh8_step): root box in T, child boxes in Float32 in the
// normalised frame with 1/d clamped to 2^40, hit children in ascending (slot ^ octant) order, triangles in T on the caller's coordinates.
// (One item per trip, as the Float64 kernel walks; the Float32 kernel takes a node's last pending triangle and the next node in one trip — the same sets of
// nodes and triangles up to pruning, the same result: the device tests compare both with the linear scan bit for bit.)
struct Stats8 { uint64_t nodes = 0, tris = 0; int max_sp = 0; };
template <class T>
static void traverse(const spira::RawVec<uint32_t> &nodes, const spira::RawVec<spira::HostPack4<T>> &tris, const spira::BvhFrame<T> &fr, V<T> o, V<T> d, T t_min,
                     T &closest, int &prim, Stats8 &st) {
    const V<T> inv = {(T)1 / d.x, (T)1 / d.y, (T)1 / d.z};
    T te;
    {
        T x1 = (fr.root_mn[0] - o.x) * inv.x, x2 = (fr.root_mx[0] - o.x) * inv.x, y1 = (fr.root_mn[1] - o.y) * inv.y, y2 = (fr.root_mx[1] - o.y) * inv.y,
          z1 = (fr.root_mn[2] - o.z) * inv.z, z2 = (fr.root_mx[2] - o.z) * inv.z;
        T en = std::fmax(std::fmax(std::fmin(x1, x2), std::fmin(y1, y2)), std::fmin(z1, z2));
        T ex = std::fmin(std::fmin(std::fmax(x1, x2), std::fmax(y1, y2)), std::fmax(z1, z2));
        if (!(en <= ex && ex >= 0 && en <= closest)) return;
        te = std::fmax(en, (T)0);
    }
    const float ox = (float)(((o.x + d.x * te) - fr.centre[0]) * fr.scale), oy = (float)(((o.y + d.y * te) - fr.centre[1]) * fr.scale),
                oz = (float)(((o.z + d.z * te) - fr.centre[2]) * fr.scale);
    const float dx = (float)d.x, dy = (float)d.y, dz = (float)d.z;
    const bool ng[3] = {dx < 0, dy < 0, dz < 0};
    const uint32_t oct = (ng[0] ? 1u : 0u) | (ng[1] ? 2u : 0u) | (ng[2] ? 4u : 0u);
    auto rcp = [](float x, bool neg) { float a = std::fmax(std::fabs(x), std::ldexp(1.0f, -spira::kBvhInvClampExp)); return neg ? -1.0f / a : 1.0f / a; };
    const float iv[3] = {rcp(dx, ng[0]), rcp(dy, ng[1]), rcp(dz, ng[2])}, on[3] = {ox, oy, oz};
    float best = (float)((closest - te) * fr.scale) * 1.00000095367431640625f;
    uint32_t stack[spira::kBvhStack];
    int sp = 0;
    uint32_t G = 1u << oct;
    while (true) {
        const uint32_t pos = (uint32_t)__builtin_ctz(G), idx = (G >> 8) + (pos ^ oct);
        G &= G - 1u;
        if (G & 0xFFu) { CHECK(sp < spira::kBvhStack); stack[sp++] = G; st.max_sp = std::max(st.max_sp, sp); }
        CHECK(idx < fr.n_slots);
        const uint32_t *w = &nodes[(size_t)idx * spira::kBvhNodeDwords];
        ++st.nodes;
        const uint32_t imask = w[3] >> 24;
        uint32_t hits = 0;
        for (int i = 0; i < 8; ++i) {
            float tn = 0.0f, tf = best;
            for (int k = 0; k < 3; ++k) {
                const float step = spira::bits_float(((w[3] >> (8 * k)) & 0xFFu) << 23);
                const float a = step * iv[k], b = (spira::bits_float(w[k]) - on[k]) * iv[k];
                const uint32_t lo = (w[8 + 2 * k + (i >> 2)] >> (8 * (i & 3))) & 0xFFu, hi = (w[14 + 2 * k + (i >> 2)] >> (8 * (i & 3))) & 0xFFu;
                const float tnk = std::fmaf((float)(ng[k] ? hi : lo), a, b), tfk = std::fmaf((float)(ng[k] ? lo : hi), a, b);
                tn = std::fmax(tn, tnk); tf = std::fmin(tf, tfk);
            }
            if (tn <= tf) hits |= 1u << i;
        }
        uint32_t ih = 0, lh = hits & ~imask;
        for (int sl = 0; sl < 8; ++sl) if (hits & imask & (1u << sl)) ih |= 1u << (sl ^ oct);
        G = (w[4] << 8) | ih;
        uint32_t tm = lh;                               // hit leaf slots; the triangle of slot s: tri_base + rank_s (4 bits per slot)
        while (tm) {
            const uint32_t sl = (uint32_t)__builtin_ctz(tm), i = w[5] + ((w[6] >> (4 * sl)) & 15u);
            tm &= tm - 1u;
            CHECK((size_t)i * 3 + 2 < tris.size());
            const auto &a = tris[3 * (size_t)i], &b = tris[3 * (size_t)i + 1], &c = tris[3 * (size_t)i + 2];
            T t;
            ++st.tris;
            if (tri_test<T>({a.x, a.y, a.z}, {b.x, b.y, b.z}, {c.x, c.y, c.z}, o, d, t_min, closest, t)) {
                int p = (int)bits_of(a.w);
                if (t < closest || p > prim) { closest = t; prim = p; best = (float)((t - te) * fr.scale) * 1.00000095367431640625f; }
            }
        }
        if (!(G & 0xFFu)) { if (sp == 0) break; G = stack[--sp]; }
    }
}

template <class T>
static void check_mesh(const char *name, const std::vector<T> &t10, uint32_t n_rays, uint32_t seed, unsigned n_threads = 0) {
    const uint32_t n = (uint32_t)(t10.size() / 10);
    spira::RawVec<uint32_t> nodes;
    spira::RawVec<spira::HostPack4<T>> tris;
    spira::BvhFrame<T> fr{};
    const bool ok = spira::bvh_build<T>(t10.data(), n, nodes, tris, fr, n_threads);
    CHECK(ok);
    if (!ok) return;
    CHECK(tris.size() == 3 * (size_t)n && nodes.size() == (size_t)fr.n_slots * spira::kBvhNodeDwords && fr.n_slots >= 1 && fr.depth < spira::kBvhStack - 2);
    std::vector<char> seen(n, 0);                       // every triangle exactly once
    for (uint32_t i = 0; i < n; ++i) { uint32_t oi = bits_of(tris[3 * (size_t)i].w); CHECK(oi < n && !seen[oi]); if (oi < n) seen[oi] = 1; }
    std::mt19937 rng(seed);
    std::uniform_real_distribution<double> U(-1, 1);
    uint32_t hits = 0;
    Stats8 st;
    for (uint32_t r = 0; r < n_rays; ++r) {
        const uint32_t k = rng() % n;                   // aim at a random triangle's first vertex (+ noise) from a random origin
        const double far = (r % 7 == 3) ? 40.0 : 4.0;   // some origins far outside the mesh, some (below) on a triangle of it
        V<T> o = {(T)(far * U(rng)), (T)(far * U(rng)), (T)(far * U(rng))};
        if (r % 5 == 1) { const uint32_t k2 = rng() % n; o = {t10[10 * (size_t)k2 + 3], t10[10 * (size_t)k2 + 4], t10[10 * (size_t)k2 + 5]}; }
        V<T> tgt = {t10[10 * (size_t)k] + (T)(0.3 * U(rng)), t10[10 * (size_t)k + 1] + (T)(0.3 * U(rng)), t10[10 * (size_t)k + 2] + (T)(0.3 * U(rng))};
        V<T> d = sub(tgt, o);
        if (r % 11 == 5) d.x = 0;                       // axis-parallel rays: the clamped reciprocal
        if (r % 13 == 6) { d.y = 0; d.z = 0; }
        T len = std::sqrt(dot(d, d));
        if (!(len > 0)) continue;
        d = {d.x / len, d.y / len, d.z / len};
        T c_lin = INFINITY; int p_lin = -1;
        for (uint32_t i = 0; i < n; ++i) {              // examples/julia-raytracer.jl:242-258
            const T *t = &t10[10 * (size_t)i];
            V<T> v0 = {t[0], t[1], t[2]}, e1 = {(T)(t[3] - t[0]), (T)(t[4] - t[1]), (T)(t[5] - t[2])}, e2 = {(T)(t[6] - t[0]), (T)(t[7] - t[1]), (T)(t[8] - t[2])};
            T tt;
            if (tri_test<T>(v0, e1, e2, o, d, (T)0.001, c_lin, tt)) { c_lin = tt; p_lin = (int)i; }
        }
        T c_bvh = INFINITY; int p_bvh = -1;
        traverse<T>(nodes, tris, fr, o, d, (T)0.001, c_bvh, p_bvh, st);
        CHECK(p_lin == p_bvh && (p_lin < 0 || std::memcmp(&c_lin, &c_bvh, sizeof(T)) == 0));
        hits += p_lin >= 0;
    }
    std::printf("%-28s n=%-7u slots=%-7u depth=%-3d rays=%u hits=%u  nodes/ray=%.1f tris/ray=%.1f stack=%d\n", name, n, fr.n_slots, fr.depth, n_rays, hits,
                (double)st.nodes / n_rays, (double)st.tris / n_rays, st.max_sp);
}

template <class T> static std::vector<T> soup(uint32_t n, uint32_t seed, double size) {
    std::mt19937 rng(seed);
    std::uniform_real_distribution<double> U(-1, 1);
    std::vector<T> t(10 * (size_t)n);
    for (uint32_t i = 0; i < n; ++i) {
        double b[3] = {3 * U(rng), 3 * U(rng), 3 * U(rng)};
        for (int v = 0; v < 3; ++v) for (int k = 0; k < 3; ++k) t[10 * (size_t)i + 3 * v + k] = (T)(b[k] + (v ? size * U(rng) : 0));
        t[10 * (size_t)i + 9] = 1;
    }
    return t;
}

template <class T> static void meshes() {
    check_mesh<T>("random soup", soup<T>(3000, 1, 0.8), 3000, 11);
    check_mesh<T>("one triangle", soup<T>(1, 2, 1.0), 300, 12);
    check_mesh<T>("two triangles", soup<T>(2, 3, 1.0), 300, 13);
    {   // 300 copies of the same triangle: all centroids coincide, leaves would overflow without the list split; ties -> the LAST copy
        std::vector<T> one = soup<T>(1, 4, 1.5), t;
        for (int i = 0; i < 300; ++i) t.insert(t.end(), one.begin(), one.end());
        check_mesh<T>("300 identical triangles", t, 400, 14);
    }
    {   // zero-area triangles (two equal vertices / all three equal) mixed into a soup
        std::vector<T> t = soup<T>(500, 5, 0.7);
        for (uint32_t i = 0; i < 500; i += 5) for (int k = 0; k < 3; ++k) { t[10 * (size_t)i + 3 + k] = t[10 * (size_t)i + k]; if (i % 10 == 0) t[10 * (size_t)i + 6 + k] = t[10 * (size_t)i + k]; }
        check_mesh<T>("zero-area triangles", t, 1500, 15);
    }
    {   // a flat grid (all z equal: one axis has zero extent) with many equal centroids per cell (each cell's triangle repeated 5 times)
        std::vector<T> t;
        for (int gx = 0; gx < 20; ++gx) for (int gy = 0; gy < 20; ++gy) for (int rep = 0; rep < 5; ++rep) {
            T x = (T)(gx * 0.25 - 2.5), y = (T)(gy * 0.25 - 2.5);
            T tri[10] = {x, y, 0, (T)(x + 0.25), y, 0, x, (T)(y + 0.25), 0, 1};
            t.insert(t.end(), tri, tri + 10);
        }
        check_mesh<T>("flat grid, repeated cells", t, 2000, 16);
    }
    {   // huge coordinate range: padding and the slab test must stay conservative
        std::vector<T> t = soup<T>(400, 6, 0.5);
        for (size_t i = 0; i < t.size(); ++i) if (i % 10 != 9) t[i] *= (i / 10 % 2 ? (T)1000 : (T)0.001);
        check_mesh<T>("mixed scales", t, 1000, 17);
    }
}

// The build runs on a pool of host threads (spira_bvh.h: chunk-parallel passes near the root, a shared queue of subtrees, level-parallel collapse):
// whatever the number of threads, the node and triangle arrays must come out byte for byte the same — and equal the linear scan.
template <class T> static void determinism(uint32_t n, uint32_t seed) {
    const std::vector<T> t10 = soup<T>(n, seed, 0.05);
    spira::RawVec<uint32_t> nodes1;
    spira::RawVec<spira::HostPack4<T>> tris1;
    spira::BvhFrame<T> fr1{};
    CHECK(spira::bvh_build<T>(t10.data(), n, nodes1, tris1, fr1, 1));
    for (unsigned nt : {2u, 3u, 8u, 13u}) {
        for (int rep = 0; rep < 2; ++rep) {
            spira::RawVec<uint32_t> nodes;
            spira::RawVec<spira::HostPack4<T>> tris;
            spira::BvhFrame<T> fr{};
            CHECK(spira::bvh_build<T>(t10.data(), n, nodes, tris, fr, nt));
            CHECK(nodes.size() == nodes1.size() && tris.size() == tris1.size() && fr.n_slots == fr1.n_slots && fr.depth == fr1.depth);
            CHECK(nodes.size() == nodes1.size() && std::memcmp(nodes.data(), nodes1.data(), nodes.size() * sizeof(uint32_t)) == 0);
            CHECK(tris.size() == tris1.size() && std::memcmp(tris.data(), tris1.data(), tris.size() * sizeof(tris[0])) == 0);
            CHECK(std::memcmp(&fr, &fr1, sizeof fr) == 0);
        }
    }
    std::printf("determinism %-6s n=%-6u slots=%-6u depth=%d: 1 == 2 == 3 == 8 == 13 threads\n", sizeof(T) == 4 ? "f32" : "f64", n, fr1.n_slots, fr1.depth);
}

// The build's thread pool on its own: thousands of short generations of varying job counts (serial one-job runs and sleepy gaps in between, more threads than
// cores), every job of every generation executed exactly once.  (Round 4: a worker that was late for one generation could pair the next generation's job count
// with the old claim word and run a job nobody waited for — one build in ~40 hung or crashed.  run() now closes the claim word before it rewrites the descriptor.)
static void pool_stress(int reps, int runs) {
    for (int rep = 0; rep < reps; ++rep) {
        spira::HostPool pool(16);
        for (int r = 0; r < runs; ++r) {
            const size_t n = 1 + (size_t)((r * 7 + rep) % 41);
            std::vector<std::atomic<int>> hits(n);
            for (auto &h : hits) h = 0;
            std::atomic<int> total{0};
            pool.run(n, [&](size_t j) { volatile double x = 1; for (int i = 0; i < 100 + (int)(j % 5) * 200; ++i) x = x * 1.000001 + 1e-9; if (j < n) hits[j]++; total++; });
            bool ok = total == (int)n;
            for (size_t j = 0; j < n; ++j) ok = ok && hits[j] == 1;
            CHECK(ok);
            if (!ok) return;
            if (r % 50 == 0) std::this_thread::sleep_for(std::chrono::microseconds(200));
        }
    }
    std::printf("pool stress: %d pools x %d generations, every job exactly once\n", reps, runs);
}

static void fastdiv_checks() {
    std::mt19937 rng(7);
    const uint32_t ds[] = {1, 2, 3, 5, 7, 8, 64, 1000, 1920, 2073600, 132710400, 0x7FFFFFFF, 0x80000000u, 0xFFFFFFFFu};
    for (uint32_t d : ds) {
        const spira::FastDiv f = spira::fastdiv_make(d);
        const uint32_t probes[] = {0, 1, d - 1, d, d + 1, 2 * d - 1, 2 * d, 0x7FFFFFFFu, 0xFFFFFFFEu, 0xFFFFFFFFu};
        for (uint32_t n : probes) CHECK(spira::fastdiv(n, f) == n / d);
        for (int i = 0; i < 20000; ++i) { uint32_t n = rng(); CHECK(spira::fastdiv(n, f) == n / d); }
    }
    for (int i = 0; i < 20000; ++i) { uint32_t d = rng() | 1u, n = rng(); CHECK(spira::fastdiv(n, spira::fastdiv_make(d)) == n / d); }
    unsigned char buf[37];
    for (int i = 0; i < 37; ++i) buf[i] = (unsigned char)(i * 7);
    uint64_t h0 = spira::bytes_hash64(buf, 0), h1 = spira::bytes_hash64(buf + 1, 36), h2 = spira::bytes_hash64(buf, 37);    // unaligned start, odd length
    CHECK(h0 != h1 && h1 != h2);
}

static void validation_checks() {
    const char *msg = nullptr;
    float s5[10] = {0, 0, 0, 1, 1, 1, 1, 1, 0.5f, 2}, m8[16] = {0};
    CHECK(spira::scene_arrays_check<float>(s5, m8, nullptr, 2, 2, 0, &msg) == 0);
    s5[9] = 3; CHECK(spira::scene_arrays_check<float>(s5, m8, nullptr, 2, 2, 0, &msg) == SPIRA_E_INVALID);
    s5[9] = 1.5f; CHECK(spira::scene_arrays_check<float>(s5, m8, nullptr, 2, 2, 0, &msg) == SPIRA_E_INVALID);
    s5[9] = NAN; CHECK(spira::scene_arrays_check<float>(s5, m8, nullptr, 2, 2, 0, &msg) == SPIRA_E_INVALID);
    s5[9] = 2; s5[5] = INFINITY; CHECK(spira::scene_arrays_check<float>(s5, m8, nullptr, 2, 2, 0, &msg) == SPIRA_E_INVALID);
    s5[5] = 1;
    {   // the scale predictor of k_path's speculative division: zero coordinates are ordinary, zero / tiny / huge radii and coordinates are not
        float a5[10] = {0, 0, 0, 1, 1, 1, -2.5f, 1000.f, 0.5f, 2};
        CHECK(spira::scene_scale_moderate<float>(a5, nullptr, 2, 0));
        a5[3] = 0; CHECK(!spira::scene_scale_moderate<float>(a5, nullptr, 2, 0));                  // radius 0
        a5[3] = 1; a5[6] = 1e-30f; CHECK(!spira::scene_scale_moderate<float>(a5, nullptr, 2, 0));
        a5[6] = 3e7f; CHECK(!spira::scene_scale_moderate<float>(a5, nullptr, 2, 0));
        a5[6] = 5e5f; CHECK(spira::scene_scale_moderate<float>(a5, nullptr, 2, 0));                 // 2^19: inside 2^-20 .. 2^20
        double d5[5] = {1e18, 0, -1e-18, 1e-10, 1}, tt[10] = {0, 0, 0, 1, 0, 0, 0, 1e200, 0, 1}, cam[12] = {0, 1, 5, -1, 0, 0, 2, 0, 0, 0, 1.1, 0};
        CHECK(spira::scene_scale_moderate<double>(d5, nullptr, 1, 0));                             // 2^-64 .. 2^64 in Float64
        CHECK(!spira::scene_scale_moderate<double>(d5, tt, 1, 1));
        CHECK(spira::camera_scale_moderate<double>(cam));
        cam[4] = 1e-300; CHECK(!spira::camera_scale_moderate<double>(cam));
        cam[4] = NAN; CHECK(!spira::camera_scale_moderate<double>(cam));
    }
    double t10[10] = {0, 0, 0, 1, 0, 0, 0, 1, 0, 1}, m8d[8] = {0};
    CHECK(spira::scene_arrays_check<double>(nullptr, m8d, t10, 0, 1, 1, &msg) == 0);
    t10[4] = NAN; CHECK(spira::scene_arrays_check<double>(nullptr, m8d, t10, 0, 1, 1, &msg) == SPIRA_E_INVALID);
    t10[4] = 0; t10[9] = 0; CHECK(spira::scene_arrays_check<double>(nullptr, m8d, t10, 0, 1, 1, &msg) == SPIRA_E_INVALID);
    CHECK(spira::scene_arrays_check<double>(nullptr, nullptr, nullptr, 0, 1, 0, &msg) == SPIRA_E_INVALID);
    spira::RawVec<uint32_t> nodes;
    spira::RawVec<spira::HostPack4<float>> tris;
    spira::BvhFrame<float> fr{};
    CHECK(!spira::bvh_build<float>(nullptr, 0, nodes, tris, fr));                                 // empty
    CHECK(!spira::bvh_build<float>(nullptr, (1u << 24) + 1, nodes, tris, fr));                    // over the 2^24 limit: rejected before any read
}

#ifndef SPIRA_NO_ORACLE
static void oracle_checks() {   // the checker itself under ASan/UBSan: S2-like scene, all estimators, extensions, tilings, row orders
    const double sph[25] = {0, -100.5, -1, 100, 1, 0, 0, -1, 0.5, 2, 1, 0, -1, 0.5, 3, -1, 0, -1, 0.5, 4, 0, 2, 0, 0.5, 5};
    const double mat[48] = {0.8, 0.8, 0.2, 0, 0, 0, 0, 1, 0.8, 0.2, 0.2, 0, 0, 0, 0, 1, 0.8, 0.6, 0.2, 0, 0, 0, 0.8, 0.3, 0.9, 0.9, 0.9, 0, 0, 0, 0, -1.5,
                            0.8, 0.8, 0.8, 4, 4, 4, 0, 1, 0.2, 0.8, 0.2, 0, 0, 0, 0, 1};
    const double tri[10] = {-0.5, 0, -2, 0.5, 0, -2, 0, 1, -2, 6};
    const double cam[12] = {0, 1, 3, -1.4, -0.2, 1.6, 2.9, 0, 0, 0, 1.5, -0.6};
    std::vector<double> hdr(3 * 23 * 41), img(3 * 23 * 41);
    uint64_t seg = 0;
    for (uint32_t flags : {0u, (uint32_t)SPIRA_EXT_DIELECTRIC, (uint32_t)SPIRA_EXT_SPECTRAL, (uint32_t)(SPIRA_EXT_DIELECTRIC | SPIRA_EXT_SPECTRAL | SPIRA_ROWS_BOTTOM_UP | SPIRA_POST_ACES_GAMMA)}) {
        spira_params p{}; p.width = 41; p.height = 23; p.spp = 3; p.max_depth = 6; p.n_spheres = 5; p.n_materials = 6; p.n_triangles = 1; p.flags = flags; p.seed = 99;
        CHECK(oracle_render_f64(sph, mat, tri, cam, &p, hdr.data(), img.data(), 2, &seg) == 0 && seg > 41 * 23 * 3);
        p.rows = 8; p.stripe_h = 4; p.stripe_count = 3; p.stripe_rank = 2;        // rows 8..11, 20..22 (ragged)
        p.rows = 7;
        CHECK(oracle_render_f64(sph, mat, tri, cam, &p, hdr.data(), img.data(), 1, &seg) == 0);
    }
    float sphf[25], matf[48], camf[12];
    for (int i = 0; i < 25; ++i) sphf[i] = (float)sph[i];
    for (int i = 0; i < 48; ++i) matf[i] = (float)std::fabs(mat[i]);
    for (int i = 0; i < 12; ++i) camf[i] = (float)cam[i];
    std::vector<float> hf(3 * 23 * 41), imf(3 * 23 * 41);
    for (uint32_t sem : {(uint32_t)SPIRA_SEM_CPU, (uint32_t)SPIRA_SEM_METAL}) {
        spira_params p{}; p.width = 41; p.height = 23; p.spp = 4; p.max_depth = 9; p.n_spheres = 5; p.n_materials = 6; p.flags = sem; p.seed = 5;
        CHECK(oracle_render_variant_f32(sphf, matf, camf, &p, hf.data(), imf.data(), 2, &seg) == 0 && seg > 0);
    }
    for (float v : hf) CHECK(std::isfinite(v));
}
#endif

int main(int argc, char **argv) {
    // all three regimes of the parallel build: > 16 384 items (chunk-parallel splits), the shared queue, the plain recursion; small meshes with a forced pool
    determinism<float>(40000, 21);
    determinism<double>(40000, 22);
    determinism<float>(700, 23);
    determinism<double>(5, 24);
    pool_stress(argc > 1 ? 12 : 120, 300);
    if (argc > 1 && std::strcmp(argv[1], "determinism") == 0) {      // (the ThreadSanitizer build runs this part only)
        if (g_fail) { std::fprintf(stderr, "%d check(s) failed\n", g_fail); return 1; }
        std::printf("host sanitize harness: all checks passed\n");
        return 0;
    }
    check_mesh<float>("big soup, 8 threads", soup<float>(40000, 25, 0.05), 150, 18, 8);
    check_mesh<double>("big soup, 3 threads", soup<double>(40000, 26, 0.05), 150, 19, 3);
    meshes<float>();
    meshes<double>();
    fastdiv_checks();
    validation_checks();
#ifndef SPIRA_NO_ORACLE
    oracle_checks();
#endif
    if (g_fail) { std::fprintf(stderr, "%d check(s) failed\n", g_fail); return 1; }
    std::printf("host sanitize harness: all checks passed\n");
    return 0;
}
