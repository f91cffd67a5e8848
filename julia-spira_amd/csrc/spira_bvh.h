// spira_bvh.h — host-side builder of the mesh acceleration structure: an 8-wide BVH with quantised child boxes (the MI355X
// answer to the reference's "BVH", which is a plain list: examples/julia-raytracer.jl:231-258, and to the flat median-split
// BVH of examples/julia-raytracer-optimized.jl:1327-1419).  No HIP headers: also built into tests/native/host_sanitize.cpp.
//
// Contract with the kernels (spira_device.h, bvh8_*): the traversal must return EXACTLY what the reference's linear
// closest-hit scan returns — the minimal t over all triangles that pass the Möller–Trumbore test, ties going to the triangle
// that comes LATER in the caller's array (`t > closest_so_far` rejects, so an equal t replaces the earlier hit, :179/:219-224).
// The tree only prunes: boxes are padded and quantised outward, the slab test is conservative, and the leaf test is the same
// arithmetic as the linear scan, in the render precision, on the caller's coordinates.
//
// Why this shape (profiles/r03_gather_chase.txt): a per-lane (divergent) 16-byte load costs the CU ~7.5 ns per wave-instruction
// whether 64 lanes are active or one, and a dependent step from L2 takes one wave 0.6-1.4 us at 20 waves per CU.  So a node
// visit must fetch few bytes and a ray must need few dependent visits: 8 children per 80-byte node (a binary node with two
// Float32 boxes is 64 bytes for 2 children, in Float64 128), boxes in Float32 whatever the render precision — a box test only
// has to be conservative — and the first slots of the breadth-first array are small enough to be staged into LDS.
//
// Frame: boxes live in the mesh's NORMALISED frame x_n = (x - centre) * scale (scale a power of two, largest extent -> (0.5, 1]),
// so their Float32 arithmetic is well conditioned for a mesh of any size anywhere; a ray enters it at the root box, which is
// tested in the render precision in the caller's coordinates (root_mn / root_mx below).
//
// Node slot = 20 dwords (80 bytes):
//   0..2   p.xyz      Float32 origin of the node's quantisation grid (normalised frame)
//   3      ex | ey<<8 | ez<<16 | imask<<24     biased exponents of the grid steps (step = 2^(e-127)); imask bit s: child slot s is a node
//   4      child_base  slot index of child slot 0 (a block of 8 consecutive slots; slots of absent / leaf children are unused holes)
//   5      tri_base    first triangle (reordered array) of this node's leaf children (one triangle per leaf child, in slot order; < 2^24)
//   6      rank[8]     4 bits per slot: leaf child s holds triangle tri_base + rank_s (its rank among the node's leaf slots); else 0
//   7      reserved (0)
//   8..19  qlo_x[8] qlo_y[8] qlo_z[8] qhi_x[8] qhi_y[8] qhi_z[8]   child boxes on the grid, one byte each (an empty child: lo 255, hi 0)
// Child slots are assigned so that slot index bit k says "on the positive side along axis k" (greedy assignment on centroid
// offsets): visiting hit children in ascending (slot XOR ray octant) order is approximately front to back.
// Triangles are reordered node by node (breadth first): tri[3i] = {v0, bits(original index)}, tri[3i+1] = {e1, bits(material0)},
// tri[3i+2] = {e2, 0}, with e1 = v1 - v0, e2 = v2 - v0 computed in the render precision (:149-150).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

namespace spira {

constexpr uint32_t kBvhMaxTris = 1u << 24;
constexpr int kBvhStack = 64;                 // traversal stack levels (one entry per 8-wide level at most)
constexpr uint32_t kBvhNodeDwords = 20;
constexpr uint32_t kBvhLeafTris = 1;          // triangles per leaf child: its quantised box is the triangle's own (1: 7.8 walk trips per camera ray on config 5; 3: 8.6)
constexpr int kBvhInvClampExp = 40;           // traversal: |1/d| is clamped to 2^40 (normalised frame: coordinates within ~1)

template <class T> struct HostPack4 { T x, y, z, w; };

template <class T> inline T bits_to_real(uint32_t u) {
    T r;
    if constexpr (sizeof(T) == 4) { std::memcpy(&r, &u, 4); }
    else { uint64_t v = u; std::memcpy(&r, &v, 8); }
    return r;
}
inline uint32_t float_bits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
inline float bits_float(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }

// What the device needs beside the node and triangle arrays.
template <class T> struct BvhFrame {
    T root_mn[3], root_mx[3];     // padded box of the whole mesh in the caller's coordinates (rounded outward in T)
    T centre[3];                  // normalised frame: x_n = (x - centre) * scale
    T scale;
    uint32_t n_slots;             // node slots (80 bytes each), breadth first: slot 0 = root
    int depth;                    // levels of 8-wide nodes
};

template <class T> struct Bvh8Build {
    struct Item { double c[3], mn[3], mx[3]; uint32_t idx; };          // normalised frame, boxes padded
    struct BNode { double mn[3], mx[3]; int left, right; uint32_t first, count; };   // binary SAH tree; left < 0: leaf over items [first, first+count)
    std::vector<Item> items;
    std::vector<BNode> bn;

    static double area(const double a[3], const double b[3]) {
        double e0 = std::max(0.0, b[0] - a[0]), e1 = std::max(0.0, b[1] - a[1]), e2 = std::max(0.0, b[2] - a[2]);
        return e0 * e1 + e1 * e2 + e2 * e0;
    }

    // binned SAH (16 bins, three axes); median split of the list when the centroids coincide or the tree gets deep
    int build_binary(uint32_t first, uint32_t count, int depth) {
        const int me = (int)bn.size();
        bn.push_back(BNode{});
        double mn[3], mx[3], cmn[3], cmx[3];
        for (int k = 0; k < 3; ++k) { mn[k] = cmn[k] = std::numeric_limits<double>::infinity(); mx[k] = cmx[k] = -mn[k]; }
        for (uint32_t i = first; i < first + count; ++i)
            for (int k = 0; k < 3; ++k) {
                mn[k] = std::min(mn[k], items[i].mn[k]); mx[k] = std::max(mx[k], items[i].mx[k]);
                cmn[k] = std::min(cmn[k], items[i].c[k]); cmx[k] = std::max(cmx[k], items[i].c[k]);
            }
        for (int k = 0; k < 3; ++k) { bn[me].mn[k] = mn[k]; bn[me].mx[k] = mx[k]; }
        bn[me].first = first; bn[me].count = count; bn[me].left = bn[me].right = -1;
        if (count <= kBvhLeafTris) return me;
        constexpr int NB = 16;
        int best_axis = -1, best_bin = -1;
        double best_cost = std::numeric_limits<double>::infinity();
        if (depth < 40)
            for (int ax = 0; ax < 3; ++ax) {
                const double ext = cmx[ax] - cmn[ax];
                if (!(ext > 0)) continue;
                double bmn[NB][3], bmx[NB][3];
                uint32_t bcnt[NB] = {0};
                for (int b = 0; b < NB; ++b) for (int k = 0; k < 3; ++k) { bmn[b][k] = std::numeric_limits<double>::infinity(); bmx[b][k] = -bmn[b][k]; }
                for (uint32_t i = first; i < first + count; ++i) {
                    const int b = std::min(NB - 1, (int)((items[i].c[ax] - cmn[ax]) / ext * NB));
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
                    const double cost = larea[b - 1] * lcnt[b - 1] + area(rmn, rmx) * c;
                    if (cost < best_cost) { best_cost = cost; best_axis = ax; best_bin = b; }
                }
            }
        uint32_t mid;
        if (best_axis >= 0) {
            const double ext = cmx[best_axis] - cmn[best_axis], lo = cmn[best_axis];
            const int ax = best_axis, bb = best_bin;
            auto it = std::partition(items.begin() + first, items.begin() + first + count, [&](const Item &t) {
                return std::min(NB - 1, (int)((t.c[ax] - lo) / ext * NB)) < bb;
            });
            mid = (uint32_t)(it - items.begin());
        } else mid = first + count / 2;
        if (mid == first || mid == first + count) mid = first + count / 2;
        const int l = build_binary(first, mid - first, depth + 1);
        const int r = build_binary(mid, first + count - mid, depth + 1);
        bn[me].left = l; bn[me].right = r;
        return me;
    }
};

// Builds the structure over n triangles (caller's triangles10 layout).  Outputs: nodes (kBvhNodeDwords per slot, slot 0 = root),
// tris (3 packets per triangle, node order), frame.  Returns false if a limit is hit.
template <class T>
bool bvh_build(const T *triangles10, uint32_t n, std::vector<uint32_t> &nodes, std::vector<HostPack4<T>> &tris, BvhFrame<T> &frame) {
    if (n == 0 || n > kBvhMaxTris) return false;
    // ---- frame: centre and power-of-two scale from the bounds of all vertices
    double lo[3], hi[3], amax = 0;
    for (int k = 0; k < 3; ++k) { lo[k] = std::numeric_limits<double>::infinity(); hi[k] = -lo[k]; }
    for (uint32_t i = 0; i < n; ++i)
        for (int v = 0; v < 3; ++v)
            for (int k = 0; k < 3; ++k) {
                const double x = (double)triangles10[10 * (size_t)i + 3 * v + k];
                lo[k] = std::min(lo[k], x); hi[k] = std::max(hi[k], x); amax = std::max(amax, std::fabs(x));
            }
    double ext = 0, centre[3];
    for (int k = 0; k < 3; ++k) {
        centre[k] = (double)(T)(lo[k] * 0.5 + hi[k] * 0.5);        // (halves first: lo + hi may overflow)
        ext = std::max(ext, std::max(hi[k] - centre[k], centre[k] - lo[k]) * 2);
    }
    const int emax = sizeof(T) == 8 ? 1000 : 120;
    int se = 0;
    if (ext > 0 && std::isfinite(ext)) { int e2; std::frexp(ext, &e2); se = std::max(-emax, std::min(emax, -e2)); }       // ext * 2^se in [0.5, 1)
    else if (!std::isfinite(ext)) se = -emax;
    const double scale = std::ldexp(1.0, se);
    // Padding (normalised units): far above the Float32 rounding of the box arithmetic (~1e-7 of the extent) and above the rounding of a
    // hit point o + t*d computed in T (a few ulps of the coordinates: amax * eps_T) — the same 1e-4 of round 1-2's boxes.
    const double amax_n = amax * scale;
    const double pad = sizeof(T) == 4 ? 1e-4 * std::max(1.0, amax_n) : 1e-4 + 1e-9 * amax_n;
    Bvh8Build<T> b;
    b.items.resize(n);
    for (uint32_t i = 0; i < n; ++i) {
        const T *t = triangles10 + 10 * (size_t)i;
        auto &it = b.items[i];
        it.idx = i;
        for (int k = 0; k < 3; ++k) {
            const double a = ((double)t[k] - centre[k]) * scale, bb = ((double)t[3 + k] - centre[k]) * scale, c = ((double)t[6 + k] - centre[k]) * scale;
            it.mn[k] = std::min(a, std::min(bb, c)) - pad; it.mx[k] = std::max(a, std::max(bb, c)) + pad;
            it.c[k] = (a + bb + c) / 3.0;
        }
    }
    b.bn.reserve(2 * (size_t)n);
    const int broot = b.build_binary(0, n, 0);
    // ---- root box in the caller's coordinates, rounded outward in T
    for (int k = 0; k < 3; ++k) {
        const double mnw = b.bn[broot].mn[k] / scale + centre[k], mxw = b.bn[broot].mx[k] / scale + centre[k];
        T a = (T)mnw, c = (T)mxw;
        const T big = std::numeric_limits<T>::max();
        a = std::nextafter(std::nextafter(a, -big), -big); c = std::nextafter(std::nextafter(c, big), big);
        frame.root_mn[k] = a; frame.root_mx[k] = c; frame.centre[k] = (T)centre[k];
    }
    frame.scale = (T)scale;
    // ---- collapse to 8-wide nodes, breadth first
    struct Pending { int bnode; uint32_t slot; int level; };
    std::vector<Pending> queue;
    queue.push_back({broot, 0u, 1});
    nodes.assign(kBvhNodeDwords, 0u);
    std::vector<uint32_t> order;
    order.reserve(n);
    int depth = 0;
    for (size_t qi = 0; qi < queue.size(); ++qi) {
        const Pending cur = queue[qi];
        depth = std::max(depth, cur.level);
        if (cur.level >= kBvhStack - 2) return false;
        // the up to 8 entries of this node: open the entry with the largest box until 8 or all are leaves
        int ent[8], ne = 0;
        const auto &root = b.bn[cur.bnode];
        if (root.left < 0) ent[ne++] = cur.bnode;
        else { ent[ne++] = root.left; ent[ne++] = root.right; }
        while (ne < 8) {
            int pick = -1; double pa = -1;
            for (int i = 0; i < ne; ++i)
                if (b.bn[ent[i]].left >= 0) { const double a = Bvh8Build<T>::area(b.bn[ent[i]].mn, b.bn[ent[i]].mx); if (a > pa) { pa = a; pick = i; } }
            if (pick < 0) break;
            const int o = ent[pick];
            ent[pick] = b.bn[o].left; ent[ne++] = b.bn[o].right;
        }
        // node box, centre
        double nmn[3], nmx[3], nc[3];
        for (int k = 0; k < 3; ++k) { nmn[k] = std::numeric_limits<double>::infinity(); nmx[k] = -nmn[k]; }
        for (int i = 0; i < ne; ++i) for (int k = 0; k < 3; ++k) { nmn[k] = std::min(nmn[k], b.bn[ent[i]].mn[k]); nmx[k] = std::max(nmx[k], b.bn[ent[i]].mx[k]); }
        for (int k = 0; k < 3; ++k) nc[k] = 0.5 * (nmn[k] + nmx[k]);
        // greedy slot assignment: slot bit k set <=> the child sits on the positive side along axis k
        int slot_of[8], ent_at[8];
        bool slot_used[8] = {false}, ent_done[8] = {false};
        for (int s = 0; s < 8; ++s) ent_at[s] = -1;
        for (int round = 0; round < ne; ++round) {
            double bestc = -std::numeric_limits<double>::infinity(); int bi = -1, bs = -1;
            for (int i = 0; i < ne; ++i) {
                if (ent_done[i]) continue;
                double off[3];
                for (int k = 0; k < 3; ++k) off[k] = 0.5 * (b.bn[ent[i]].mn[k] + b.bn[ent[i]].mx[k]) - nc[k];
                for (int s = 0; s < 8; ++s) {
                    if (slot_used[s]) continue;
                    const double c = ((s & 1) ? off[0] : -off[0]) + ((s & 2) ? off[1] : -off[1]) + ((s & 4) ? off[2] : -off[2]);
                    if (c > bestc) { bestc = c; bi = i; bs = s; }
                }
            }
            ent_done[bi] = true; slot_used[bs] = true; slot_of[bi] = bs; ent_at[bs] = ent[bi];
        }
        (void)slot_of;
        // quantisation grid
        float p[3]; uint32_t eb[3];
        for (int k = 0; k < 3; ++k) {
            float pf = (float)nmn[k];
            if ((double)pf > nmn[k]) pf = std::nextafter(pf, -std::numeric_limits<float>::infinity());
            p[k] = pf;
            int e = -120;
            const double span = nmx[k] - (double)pf;
            if (span > 0) { int e2; std::frexp(span / 255.0, &e2); e = std::max(-120, e2); }       // 2^e2 > span / 255
            while ((double)pf + 255.0 * std::ldexp(1.0, e) < nmx[k]) ++e;                          // (the grid must reach the far side exactly)
            if (e > 120) return false;
            eb[k] = (uint32_t)(e + 127);
        }
        uint32_t imask = 0, n_int = 0, rank_word = 0;
        uint8_t q[6][8];
        for (int s = 0; s < 8; ++s) { for (int a = 0; a < 3; ++a) { q[a][s] = 255; q[3 + a][s] = 0; } }
        const uint32_t tri_base = (uint32_t)order.size();
        uint32_t tri_off = 0;
        for (int s = 0; s < 8; ++s) {
            if (ent_at[s] < 0) continue;
            const auto &c = b.bn[ent_at[s]];
            for (int k = 0; k < 3; ++k) {
                const double step = std::ldexp(1.0, (int)eb[k] - 127);
                double ql = std::floor((c.mn[k] - (double)p[k]) / step), qh = std::ceil((c.mx[k] - (double)p[k]) / step);
                ql = std::min(255.0, std::max(0.0, ql)); qh = std::min(255.0, std::max(0.0, qh));
                if ((double)p[k] + ql * step > c.mn[k] && ql > 0) ql -= 1;
                if ((double)p[k] + qh * step < c.mx[k] && qh < 255) qh += 1;
                if ((double)p[k] + ql * step > c.mn[k] || (double)p[k] + qh * step < c.mx[k]) return false;   // cannot happen: p + 255 * step >= node max
                q[k][s] = (uint8_t)ql; q[3 + k][s] = (uint8_t)qh;
            }
            if (c.left >= 0) { imask |= 1u << s; ++n_int; }
            else {
                if (c.count != 1) return false;
                rank_word |= tri_off << (4 * s);
                order.push_back(c.first);
                ++tri_off;
            }
        }
        uint32_t child_base = 0;
        if (n_int) {
            child_base = (uint32_t)(nodes.size() / kBvhNodeDwords);
            if ((uint64_t)child_base + 8 > (1u << 24)) return false;                              // a stack entry holds 24 bits of it
            nodes.resize(nodes.size() + 8 * kBvhNodeDwords, 0u);
            for (int s = 0; s < 8; ++s) {
                if (!(imask & (1u << s))) {           // a hole: give it empty children so that a stray visit finds nothing
                    uint32_t *h = &nodes[((size_t)child_base + s) * kBvhNodeDwords];
                    h[8] = h[9] = h[10] = h[11] = h[12] = h[13] = 0xFFFFFFFFu;
                    continue;
                }
                queue.push_back({ent_at[s], child_base + (uint32_t)s, cur.level + 1});
            }
        }
        uint32_t *w = &nodes[(size_t)cur.slot * kBvhNodeDwords];
        w[0] = float_bits(p[0]); w[1] = float_bits(p[1]); w[2] = float_bits(p[2]);
        w[3] = eb[0] | (eb[1] << 8) | (eb[2] << 16) | (imask << 24);
        w[4] = child_base; w[5] = tri_base;
        w[6] = rank_word;
        w[7] = 0;
        for (int a = 0; a < 6; ++a) {
            w[8 + 2 * a] = q[a][0] | (q[a][1] << 8) | (q[a][2] << 16) | ((uint32_t)q[a][3] << 24);
            w[9 + 2 * a] = q[a][4] | (q[a][5] << 8) | (q[a][6] << 16) | ((uint32_t)q[a][7] << 24);
        }
    }
    if (order.size() != n) return false;
    frame.n_slots = (uint32_t)(nodes.size() / kBvhNodeDwords);
    frame.depth = depth;
    tris.resize(3 * (size_t)n);
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t oi = b.items[order[i]].idx;
        const T *t = triangles10 + 10 * (size_t)oi;
        tris[3 * (size_t)i + 0] = {t[0], t[1], t[2], bits_to_real<T>(oi)};
        tris[3 * (size_t)i + 1] = {(T)(t[3] - t[0]), (T)(t[4] - t[1]), (T)(t[5] - t[2]), bits_to_real<T>((uint32_t)t[9] - 1u)};   // edge1 = v1 - v0, :149
        tris[3 * (size_t)i + 2] = {(T)(t[6] - t[0]), (T)(t[7] - t[1]), (T)(t[8] - t[2]), (T)0};                                // edge2 = v2 - v0, :150
    }
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
