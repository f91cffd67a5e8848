// spira_bvh.h — host-side BVH builder for triangle meshes (the MI355X answer to the reference's
// "BVH", which is a plain list: examples/julia-raytracer.jl:231-258, and to the flat median-split BVH of
// examples/julia-raytracer-optimized.jl:1327-1419).
//
// Contract with the kernels (spira_device.h, bvh_closest_hit): the traversal must return EXACTLY what
// the reference's linear closest-hit scan returns — the minimal t over all triangles that pass the
// Möller–Trumbore test, ties going to the triangle that comes LATER in the caller's array
// (`t > closest_so_far` rejects, so an equal t replaces the earlier hit, :179/:219-224).  The tree only
// prunes: boxes are padded, the slab test is conservative, and the leaf test is the same arithmetic as
// the linear scan.  tests/test_gpu_parity.py checks GPU(BVH) == oracle(linear scan) bit for bit.
//
// Layout: interior nodes only, each holding BOTH children's boxes (one 64-byte fetch per visit, f32):
//   node[4k+0] = {left.min.xyz , bits(left ref)}   node[4k+1] = {left.max.xyz , 0}
//   node[4k+2] = {right.min.xyz, bits(right ref)}  node[4k+3] = {right.max.xyz, 0}
//   ref: bit31 = leaf; leaf: bits 24..30 = triangle count (1..64), bits 0..23 = first triangle in the
//   reordered array; interior: node index.  An absent child has an inverted box and ref = kBvhNone.
// Triangles are reordered leaf by leaf: tri[3i] = {v0, bits(original index)}, tri[3i+1] = {e1, bits(material0)},
// tri[3i+2] = {e2, 0}, with e1 = v1 - v0, e2 = v2 - v0 computed in the render precision (:149-150).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

namespace spira {

constexpr uint32_t kBvhLeafFlag = 0x80000000u;
constexpr uint32_t kBvhNone = 0xFFFFFFFFu;
constexpr uint32_t kBvhMaxLeaf = 64;
constexpr uint32_t kBvhMaxTris = 1u << 24;
constexpr int kBvhStack = 64;

template <class T> struct HostPack4 { T x, y, z, w; };

template <class T> inline T bits_to_real(uint32_t u) {
    T r;
    if constexpr (sizeof(T) == 4) { std::memcpy(&r, &u, 4); }
    else { uint64_t v = u; std::memcpy(&r, &v, 8); }
    return r;
}

template <class T> struct BvhBuild {
    struct Item { double c[3], mn[3], mx[3]; uint32_t idx; };
    std::vector<Item> items;
    std::vector<HostPack4<T>> nodes;      // 4 packets per interior node
    std::vector<uint32_t> order;          // reordered position -> original triangle index
    double pad = 0;
    int max_depth = 0;

    static uint32_t leaf_ref(uint32_t first, uint32_t count) { return kBvhLeafFlag | (count << 24) | first; }

    void bounds(uint32_t first, uint32_t count, double mn[3], double mx[3], double cmn[3], double cmx[3]) const {
        for (int k = 0; k < 3; ++k) { mn[k] = cmn[k] = std::numeric_limits<double>::infinity(); mx[k] = cmx[k] = -mn[k]; }
        for (uint32_t i = first; i < first + count; ++i)
            for (int k = 0; k < 3; ++k) {
                mn[k] = std::min(mn[k], items[i].mn[k]); mx[k] = std::max(mx[k], items[i].mx[k]);
                cmn[k] = std::min(cmn[k], items[i].c[k]); cmx[k] = std::max(cmx[k], items[i].c[k]);
            }
    }

    // returns the child reference of the subtree over items [first, first+count); writes its box
    uint32_t build(uint32_t first, uint32_t count, int depth, double mn[3], double mx[3]) {
        double cmn[3], cmx[3];
        bounds(first, count, mn, mx, cmn, cmx);
        max_depth = std::max(max_depth, depth);
        if (count <= 4 || depth >= kBvhStack - 8) {
            if (count <= kBvhMaxLeaf) return leaf_ref(first, count);
        }
        // binned SAH over the three axes (16 bins)
        constexpr int NB = 16;
        int best_axis = -1, best_bin = -1;
        double best_cost = std::numeric_limits<double>::infinity();
        auto area = [](const double a[3], const double b[3]) {
            double e0 = std::max(0.0, b[0] - a[0]), e1 = std::max(0.0, b[1] - a[1]), e2 = std::max(0.0, b[2] - a[2]);
            return e0 * e1 + e1 * e2 + e2 * e0;
        };
        for (int ax = 0; ax < 3; ++ax) {
            double ext = cmx[ax] - cmn[ax];
            if (!(ext > 0)) continue;
            double bmn[NB][3], bmx[NB][3];
            uint32_t bcnt[NB] = {0};
            for (int b = 0; b < NB; ++b) for (int k = 0; k < 3; ++k) { bmn[b][k] = std::numeric_limits<double>::infinity(); bmx[b][k] = -bmn[b][k]; }
            for (uint32_t i = first; i < first + count; ++i) {
                int b = std::min(NB - 1, (int)((items[i].c[ax] - cmn[ax]) / ext * NB));
                ++bcnt[b];
                for (int k = 0; k < 3; ++k) { bmn[b][k] = std::min(bmn[b][k], items[i].mn[k]); bmx[b][k] = std::max(bmx[b][k], items[i].mx[k]); }
            }
            double lmn[3], lmx[3], rmn[3], rmx[3], larea[NB];
            uint32_t lcnt[NB], c = 0;
            for (int k = 0; k < 3; ++k) { lmn[k] = std::numeric_limits<double>::infinity(); lmx[k] = -lmn[k]; }
            for (int b = 0; b < NB - 1; ++b) {
                c += bcnt[b];
                for (int k = 0; k < 3; ++k) { lmn[k] = std::min(lmn[k], bmn[b][k]); lmx[k] = std::max(lmx[k], bmx[b][k]); }
                lcnt[b] = c; larea[b] = c ? area(lmn, lmx) : 0;
            }
            for (int k = 0; k < 3; ++k) { rmn[k] = std::numeric_limits<double>::infinity(); rmx[k] = -rmn[k]; }
            c = 0;
            for (int b = NB - 1; b > 0; --b) {
                c += bcnt[b];
                for (int k = 0; k < 3; ++k) { rmn[k] = std::min(rmn[k], bmn[b][k]); rmx[k] = std::max(rmx[k], bmx[b][k]); }
                if (lcnt[b - 1] == 0 || c == 0) continue;
                double cost = larea[b - 1] * lcnt[b - 1] + area(rmn, rmx) * c;
                if (cost < best_cost) { best_cost = cost; best_axis = ax; best_bin = b; }
            }
        }
        uint32_t mid;
        if (best_axis >= 0) {
            double ext = cmx[best_axis] - cmn[best_axis];
            auto it = std::partition(items.begin() + first, items.begin() + first + count, [&](const Item &t) {
                int b = std::min(NB - 1, (int)((t.c[best_axis] - cmn[best_axis]) / ext * NB));
                return b < best_bin;
            });
            mid = (uint32_t)(it - items.begin());
        } else {
            mid = first + count / 2;      // all centroids coincide: split the list in two
        }
        if (mid == first || mid == first + count) mid = first + count / 2;
        uint32_t me = (uint32_t)(nodes.size() / 4);
        nodes.resize(nodes.size() + 4);
        double lmn[3], lmx[3], rmn[3], rmx[3];
        uint32_t lref = build(first, mid - first, depth + 1, lmn, lmx);
        uint32_t rref = build(mid, first + count - mid, depth + 1, rmn, rmx);
        nodes[4 * me + 0] = {(T)(lmn[0] - pad), (T)(lmn[1] - pad), (T)(lmn[2] - pad), bits_to_real<T>(lref)};
        nodes[4 * me + 1] = {(T)(lmx[0] + pad), (T)(lmx[1] + pad), (T)(lmx[2] + pad), (T)0};
        nodes[4 * me + 2] = {(T)(rmn[0] - pad), (T)(rmn[1] - pad), (T)(rmn[2] - pad), bits_to_real<T>(rref)};
        nodes[4 * me + 3] = {(T)(rmx[0] + pad), (T)(rmx[1] + pad), (T)(rmx[2] + pad), (T)0};
        return me;
    }
};

// Builds the tree over n triangles (caller's triangles10 layout).  Outputs: nodes (4 packets per interior
// node, root = node 0) and tris (3 packets per triangle, leaf order).  Returns false if a limit is hit.
template <class T>
bool bvh_build(const T *triangles10, uint32_t n, std::vector<HostPack4<T>> &nodes, std::vector<HostPack4<T>> &tris, int *depth_out) {
    if (n == 0 || n > kBvhMaxTris) return false;
    BvhBuild<T> b;
    b.items.resize(n);
    double amax = 0;
    for (uint32_t i = 0; i < n; ++i) {
        const T *t = triangles10 + 10 * (size_t)i;
        auto &it = b.items[i];
        it.idx = i;
        for (int k = 0; k < 3; ++k) {
            double a = (double)t[k], bb = (double)t[3 + k], c = (double)t[6 + k];
            it.mn[k] = std::min(a, std::min(bb, c)); it.mx[k] = std::max(a, std::max(bb, c));
            it.c[k] = (a + bb + c) / 3.0;
            amax = std::max(amax, std::max(std::fabs(it.mn[k]), std::fabs(it.mx[k])));
        }
    }
    // padding: far above the rounding of a hit point o + t*d computed in T (a few ulps of the coordinates)
    b.pad = 1e-4 * amax + 1e-6;
    double mn[3], mx[3];
    b.nodes.reserve(4 * (size_t)n);
    uint32_t root = b.build(0, n, 0, mn, mx);
    const T inf = std::numeric_limits<T>::infinity();
    if (root & kBvhLeafFlag) {      // tiny mesh: make an interior root with one real child and one absent child
        b.nodes.resize(4);
        b.nodes[0] = {(T)(mn[0] - b.pad), (T)(mn[1] - b.pad), (T)(mn[2] - b.pad), bits_to_real<T>(root)};
        b.nodes[1] = {(T)(mx[0] + b.pad), (T)(mx[1] + b.pad), (T)(mx[2] + b.pad), (T)0};
        b.nodes[2] = {inf, inf, inf, bits_to_real<T>(kBvhNone)};
        b.nodes[3] = {-inf, -inf, -inf, (T)0};
    }
    if (b.max_depth >= kBvhStack - 2) return false;
    nodes.swap(b.nodes);
    tris.resize(3 * (size_t)n);
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t oi = b.items[i].idx;
        const T *t = triangles10 + 10 * (size_t)oi;
        tris[3 * (size_t)i + 0] = {t[0], t[1], t[2], bits_to_real<T>(oi)};
        tris[3 * (size_t)i + 1] = {(T)(t[3] - t[0]), (T)(t[4] - t[1]), (T)(t[5] - t[2]), bits_to_real<T>((uint32_t)t[9] - 1u)};   // edge1 = v1 - v0, :149
        tris[3 * (size_t)i + 2] = {(T)(t[6] - t[0]), (T)(t[7] - t[1]), (T)(t[8] - t[2]), (T)0};                                // edge2 = v2 - v0, :150
    }
    if (depth_out) *depth_out = b.max_depth;
    return true;
}

inline uint64_t bytes_hash64(const void *p, size_t n) {
    const unsigned char *c = (const unsigned char *)p;
    uint64_t h = 0x9E3779B97F4A7C15ull ^ (uint64_t)n;
    size_t i = 0;
    for (; i + 8 <= n; i += 8) { uint64_t v; std::memcpy(&v, c + i, 8); h = (h ^ v) * 0xff51afd7ed558ccdull; h ^= h >> 32; }
    for (; i < n; ++i) { h = (h ^ c[i]) * 0x100000001b3ull; }
    return h ^ (h >> 29);
}

}  // namespace spira
