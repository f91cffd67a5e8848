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
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <limits>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <thread>
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

// ---------------------------------------------------------------------------------------------------------------- host threads of a build
// The build of a mesh's tree is the first thing a render of a new mesh waits for (tests/bunny-test.jl:37-60 renders its scene once), so it runs
// on the host cores the process may use: a pool that lives for one build (its threads are created in ~0.5 ms and joined at the end: a library
// should not keep threads around), jobs handed out through one atomic counter.  Everything is split into jobs of FIXED size and every
// reduction merges exact quantities (min / max / integer counts), so the arrays that come out do not depend on the number of threads
// (tests/native/host_sanitize.cpp builds with 1, 2, 3 and 8 threads and compares them byte for byte).
inline unsigned build_threads() {
    if (const char *e = std::getenv("SPIRA_BUILD_THREADS")) { const long v = std::strtol(e, nullptr, 10); if (v >= 1) return (unsigned)std::min<long>(v, 64); }
    unsigned n = std::max(1u, std::thread::hardware_concurrency());
    if (FILE *f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {          // a container's CPU quota (cgroup v2): "max 100000" or "<quota> <period>"
        char q[64]; long long period = 0;
        if (std::fscanf(f, "%63s %lld", q, &period) == 2 && std::strcmp(q, "max") != 0 && period > 0) {
            const long long quota = std::strtoll(q, nullptr, 10);
            if (quota > 0) n = (unsigned)std::max<long long>(1, std::min<long long>(n, (quota + period - 1) / period));
        }
        std::fclose(f);
    }
    return std::min(n, 16u);
}

inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#else
    std::this_thread::yield();
#endif
}

// A build is a burst of a few hundred short jobs within ~10 ms.  Workers spin for the next job for ~20 us (waking a sleeping thread per job cost
// more than the jobs: 40 ms of a build's 60), then block on a condition variable — a thread that only ever spins stays on the core it was
// created on until the load balancer moves it (seconds, in a VM), one that is WOKEN is placed on an idle core.  The pool dies with the build.
class HostPool {
  public:
    explicit HostPool(unsigned n_threads) {
        for (unsigned i = 1; i < std::max(1u, n_threads); ++i) {
            try { workers_.emplace_back([this] { loop(); }); } catch (...) { break; }      // (no more threads to be had: the build runs on what there is)
        }
    }
    ~HostPool() {
        { std::lock_guard<std::mutex> lk(mu_); stop_.store(true); }
        cv_.notify_all();
        for (auto &t : workers_) t.join();
    }
    HostPool(const HostPool &) = delete;
    HostPool &operator=(const HostPool &) = delete;
    unsigned size() const { return (unsigned)workers_.size() + 1; }
    // fn(job) for job in [0, n_jobs), on all threads (the caller's included); returns when every job is done
    void run(size_t n_jobs, const std::function<void(size_t)> &fn) {
        if (n_jobs == 0) return;
        if (workers_.empty() || n_jobs == 1) { for (size_t j = 0; j < n_jobs; ++j) fn(j); return; }
        // 1. CLOSE the claim word before anything of the job descriptor changes: a worker that is late for the previous generation (it saw that
        //    generation, was descheduled, and only now reads fn / n / next) must fail its tag check — with the old tag still in `next_` it would pair
        //    the NEW n with the OLD tag, claim an index the old generation never had, and run a job nobody waits for (found by a stress run: one build
        //    in ~40 hung with done == n + 1, or crashed in a dead std::function)
        const uint64_t g = gen_.load(std::memory_order_relaxed) + 1;
        next_.store((g << 32) | 0xFFFFFFFFull, std::memory_order_seq_cst);      // tag g, index beyond any n: nothing can be claimed
        // 2. the descriptor of generation g
        fn_.store(&fn, std::memory_order_relaxed); n_jobs_.store(n_jobs, std::memory_order_relaxed);
        done_.store(0, std::memory_order_relaxed);
        // 3. open it, publish it
        next_.store(g << 32, std::memory_order_release);
        gen_.store(g);                                         // (seq_cst: ordered against the sleepers' count, below)
        if (sleepers_.load() != 0) { { std::lock_guard<std::mutex> lk(mu_); } cv_.notify_all(); }
        work(g);
        // the call ends when every JOB is done, not when every worker has looked in: a worker the system has not scheduled yet holds nobody up,
        // and when it does wake it finds the generation tag of `next_` moved on and claims nothing
        while (done_.load(std::memory_order_acquire) != n_jobs) cpu_relax();
    }

  private:
    // jobs are claimed through one word {generation, next index}: a claim that succeeds belongs to the generation the claimer saw, which is
    // therefore still running (run() waits for that job), so `fn` is alive while it executes
    void work(uint64_t g) {
        // (acquire loads, in this order: the descriptor is read BEFORE the claim word, so a claim word that still carries this generation's tag
        //  proves the descriptor read is this generation's too — run() closes the word before it rewrites the descriptor)
        const std::function<void(size_t)> *fn = fn_.load(std::memory_order_acquire);
        const size_t n = n_jobs_.load(std::memory_order_acquire);
        for (;;) {
            uint64_t cur = next_.load(std::memory_order_acquire);
            for (;;) {
                if ((cur >> 32) != (g & 0xFFFFFFFFull) || (cur & 0xFFFFFFFFull) >= n) return;
                if (next_.compare_exchange_weak(cur, cur + 1, std::memory_order_acq_rel, std::memory_order_acquire)) break;
            }
            (*fn)((size_t)(cur & 0xFFFFFFFFull));
            done_.fetch_add(1, std::memory_order_release);
        }
    }
    void loop() {
        uint64_t seen = 0;
        for (;;) {
            uint64_t g = seen;
            for (uint32_t spins = 0; spins < 2000 && (g = gen_.load()) == seen && !stop_.load(std::memory_order_relaxed); ++spins) cpu_relax();
            if (g == seen) {
                std::unique_lock<std::mutex> lk(mu_);
                sleepers_.fetch_add(1);
                cv_.wait(lk, [&] { return (g = gen_.load()) != seen || stop_.load(); });
                sleepers_.fetch_sub(1);
            }
            if (stop_.load()) return;
            seen = g;
            work(g);
        }
    }
    std::vector<std::thread> workers_;
    std::atomic<const std::function<void(size_t)> *> fn_{nullptr};
    std::atomic<size_t> n_jobs_{0}, done_{0};
    std::atomic<uint64_t> next_{0}, gen_{0};
    std::atomic<bool> stop_{false};
    std::atomic<unsigned> sleepers_{0};
    std::mutex mu_;
    std::condition_variable cv_;
};

// std::vector whose resize() leaves trivially constructible elements uninitialised: the arrays of a build are written in full by parallel jobs,
// and zero-filling 8 MB on one thread first would cost as much as the jobs
template <class U> struct DefaultInitAlloc : std::allocator<U> {
    template <class V> struct rebind { using other = DefaultInitAlloc<V>; };
    DefaultInitAlloc() = default;
    template <class V> DefaultInitAlloc(const DefaultInitAlloc<V> &) {}
    template <class V, class... A> void construct(V *p, A &&...a) {
        if constexpr (sizeof...(A) == 0) ::new ((void *)p) V; else ::new ((void *)p) V(std::forward<A>(a)...);
    }
};
template <class U> using RawVec = std::vector<U, DefaultInitAlloc<U>>;

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
    static constexpr int NB = 16;                      // SAH bins per axis
    static constexpr uint32_t kChunk = 2048;           // items per job of the parallel passes (fixed: results must not depend on the thread count)
    static constexpr uint32_t kTaskItems = 1024;       // a subtree of at most this many items is ONE job of the shared queue, built by the plain recursion
    static constexpr uint32_t kBigItems = 16384;       // a node over more than this many items is split by the whole pool (chunk-parallel passes)
    struct Bounds { double mn[3], mx[3], cmn[3], cmx[3]; };
    struct Bins { double bmn[3][NB][3], bmx[3][NB][3]; uint32_t cnt[3][NB]; };
    RawVec<Item> items, scratch;
    RawVec<BNode> bn;

    static double area(const double a[3], const double b[3]) {
        double e0 = std::max(0.0, b[0] - a[0]), e1 = std::max(0.0, b[1] - a[1]), e2 = std::max(0.0, b[2] - a[2]);
        return e0 * e1 + e1 * e2 + e2 * e0;
    }
    static void bounds_clear(Bounds &b) { for (int k = 0; k < 3; ++k) { b.mn[k] = b.cmn[k] = std::numeric_limits<double>::infinity(); b.mx[k] = b.cmx[k] = -b.mn[k]; } }
    static void bounds_add(Bounds &b, const Item *it, uint32_t n) {
        for (uint32_t i = 0; i < n; ++i)
            for (int k = 0; k < 3; ++k) {
                b.mn[k] = std::min(b.mn[k], it[i].mn[k]); b.mx[k] = std::max(b.mx[k], it[i].mx[k]);
                b.cmn[k] = std::min(b.cmn[k], it[i].c[k]); b.cmx[k] = std::max(b.cmx[k], it[i].c[k]);
            }
    }
    static void bounds_merge(Bounds &b, const Bounds &o) {
        for (int k = 0; k < 3; ++k) { b.mn[k] = std::min(b.mn[k], o.mn[k]); b.mx[k] = std::max(b.mx[k], o.mx[k]); b.cmn[k] = std::min(b.cmn[k], o.cmn[k]); b.cmx[k] = std::max(b.cmx[k], o.cmx[k]); }
    }
    static int bin_of(double c, double lo, double ext) { return std::min(NB - 1, (int)((c - lo) / ext * NB)); }
    // Bins are cleared lazily: only the counts are zeroed (192 bytes instead of 2.7 KB — most nodes of a tree hold a handful of items), a bin's
    // box is valid only where its count is non-zero.
    static void bins_clear(Bins &b) { std::memset(b.cnt, 0, sizeof b.cnt); }
    // all three axes in one pass over the items (an axis along which the centroids coincide gets no bins)
    static void bins_add(Bins &b, const Bounds &bd, const Item *it, uint32_t n) {
        double ext[3]; bool use[3];
        for (int ax = 0; ax < 3; ++ax) { ext[ax] = bd.cmx[ax] - bd.cmn[ax]; use[ax] = ext[ax] > 0; }
        for (uint32_t i = 0; i < n; ++i)
            for (int ax = 0; ax < 3; ++ax) {
                if (!use[ax]) continue;
                const int q = bin_of(it[i].c[ax], bd.cmn[ax], ext[ax]);
                if (b.cnt[ax][q]++ == 0) for (int k = 0; k < 3; ++k) { b.bmn[ax][q][k] = it[i].mn[k]; b.bmx[ax][q][k] = it[i].mx[k]; }
                else for (int k = 0; k < 3; ++k) { b.bmn[ax][q][k] = std::min(b.bmn[ax][q][k], it[i].mn[k]); b.bmx[ax][q][k] = std::max(b.bmx[ax][q][k], it[i].mx[k]); }
            }
    }
    static void bins_merge(Bins &b, const Bins &o) {
        for (int ax = 0; ax < 3; ++ax)
            for (int i = 0; i < NB; ++i) {
                if (!o.cnt[ax][i]) continue;
                if (!b.cnt[ax][i]) for (int k = 0; k < 3; ++k) { b.bmn[ax][i][k] = o.bmn[ax][i][k]; b.bmx[ax][i][k] = o.bmx[ax][i][k]; }
                else for (int k = 0; k < 3; ++k) { b.bmn[ax][i][k] = std::min(b.bmn[ax][i][k], o.bmn[ax][i][k]); b.bmx[ax][i][k] = std::max(b.bmx[ax][i][k], o.bmx[ax][i][k]); }
                b.cnt[ax][i] += o.cnt[ax][i];
            }
    }
    // binned SAH over the three axes: the cheapest plane (axes in order, planes from the far side; the first of equal costs wins).  Only planes in
    // front of a non-empty bin are priced: across an empty bin neither side changes, the cost is the same number, and an equal cost never replaces
    // the best — so this IS the full sweep, at the price of the occupied bins (a node of two items: one plane per axis instead of fifteen).
    static bool choose_split(const Bounds &bd, const Bins &b, int &best_axis, int &best_bin) {
        best_axis = -1; best_bin = -1;
        double best_cost = std::numeric_limits<double>::infinity();
        for (int ax = 0; ax < 3; ++ax) {
            if (!(bd.cmx[ax] - bd.cmn[ax] > 0)) continue;
            double lmn[3], lmx[3], rmn[3], rmx[3], larea[NB];
            uint32_t lcnt[NB], c = 0;
            double la = 0;
            for (int k = 0; k < 3; ++k) { lmn[k] = std::numeric_limits<double>::infinity(); lmx[k] = -lmn[k]; }
            for (int i = 0; i < NB - 1; ++i) {
                if (b.cnt[ax][i]) {
                    c += b.cnt[ax][i];
                    for (int k = 0; k < 3; ++k) { lmn[k] = std::min(lmn[k], b.bmn[ax][i][k]); lmx[k] = std::max(lmx[k], b.bmx[ax][i][k]); }
                    la = area(lmn, lmx);
                }
                lcnt[i] = c; larea[i] = la;
            }
            for (int k = 0; k < 3; ++k) { rmn[k] = std::numeric_limits<double>::infinity(); rmx[k] = -rmn[k]; }
            c = 0;
            for (int i = NB - 1; i > 0; --i) {
                if (!b.cnt[ax][i]) continue;
                c += b.cnt[ax][i];
                for (int k = 0; k < 3; ++k) { rmn[k] = std::min(rmn[k], b.bmn[ax][i][k]); rmx[k] = std::max(rmx[k], b.bmx[ax][i][k]); }
                if (lcnt[i - 1] == 0) continue;
                const double cost = larea[i - 1] * lcnt[i - 1] + area(rmn, rmx) * c;
                if (cost < best_cost) { best_cost = cost; best_axis = ax; best_bin = i; }
            }
        }
        return best_axis >= 0;
    }

    // One node: its box, and — unless it is a leaf — the split of its items [first, first + count) into [first, mid) and [mid, first + count):
    // binned SAH, median split of the list when the centroids coincide or the tree gets deep.  Returns mid (0: a leaf).
    uint32_t split_node(BNode &nd, uint32_t first, uint32_t count, int depth) {
        Bounds bd; bounds_clear(bd); bounds_add(bd, items.data() + first, count);
        for (int k = 0; k < 3; ++k) { nd.mn[k] = bd.mn[k]; nd.mx[k] = bd.mx[k]; }
        nd.first = first; nd.count = count; nd.left = nd.right = -1;
        if (count <= kBvhLeafTris) return 0;
        int ax = -1, bb = -1;
        if (depth < 40) { Bins b; bins_clear(b); bins_add(b, bd, items.data() + first, count); choose_split(bd, b, ax, bb); }
        uint32_t mid;
        if (ax >= 0) {
            const double ext = bd.cmx[ax] - bd.cmn[ax], lo = bd.cmn[ax];
            auto it = std::partition(items.begin() + first, items.begin() + first + count, [&](const Item &t) { return bin_of(t.c[ax], lo, ext) < bb; });
            mid = (uint32_t)(it - items.begin());
        } else mid = first + count / 2;
        if (mid == first || mid == first + count) mid = first + count / 2;
        return mid;
    }
    // The plain recursion (one thread) over a subtree whose 2 * count - 1 nodes are bn[base ...] (every leaf holds ONE item, so the count is exact);
    // `next` hands out the slots.
    int build_serial(int me, int &next, uint32_t first, uint32_t count, int depth) {
        const uint32_t mid = split_node(bn[me], first, count, depth);
        if (!mid) return me;
        const int l = next++, r = next++;
        build_serial(l, next, first, mid - first, depth + 1);
        build_serial(r, next, mid, first + count - mid, depth + 1);
        bn[me].left = l; bn[me].right = r;
        return me;
    }

    // A node over more than kBigItems items runs its passes (bounds, bins, stable partition) as jobs of kChunk items on the whole pool: near the root
    // there are fewer nodes than threads.  Returns mid.
    uint32_t split_big(HostPool &pool, BNode &nd, uint32_t first, uint32_t count, int depth) {
        nd.first = first; nd.count = count; nd.left = nd.right = -1;
        const uint32_t n_chunks = (count + kChunk - 1) / kChunk;
        auto span = [&](size_t j, uint32_t &o, uint32_t &n) { o = first + (uint32_t)j * kChunk; n = std::min(kChunk, first + count - o); };
        std::vector<Bounds> pb(n_chunks);
        pool.run(n_chunks, [&](size_t j) { uint32_t o, n; span(j, o, n); bounds_clear(pb[j]); bounds_add(pb[j], items.data() + o, n); });
        Bounds bd = pb[0];
        for (uint32_t j = 1; j < n_chunks; ++j) bounds_merge(bd, pb[j]);
        for (int k = 0; k < 3; ++k) { nd.mn[k] = bd.mn[k]; nd.mx[k] = bd.mx[k]; }
        int ax = -1, bb = -1;
        if (depth < 40) {
            RawVec<Bins> pbin(n_chunks);
            pool.run(n_chunks, [&](size_t j) { uint32_t o, n; span(j, o, n); bins_clear(pbin[j]); bins_add(pbin[j], bd, items.data() + o, n); });
            for (uint32_t j = 1; j < n_chunks; ++j) bins_merge(pbin[0], pbin[j]);
            choose_split(bd, pbin[0], ax, bb);
        }
        uint32_t mid = first + count / 2;
        if (ax >= 0) {
            // stable partition through the scratch array: every chunk counts its left items, a prefix over the chunks places them
            const double ext = bd.cmx[ax] - bd.cmn[ax], lo = bd.cmn[ax];
            std::vector<uint32_t> nl(n_chunks + 1, 0);
            pool.run(n_chunks, [&](size_t j) {
                uint32_t o, n, c = 0; span(j, o, n);
                for (uint32_t i = 0; i < n; ++i) c += bin_of(items[o + i].c[ax], lo, ext) < bb ? 1u : 0u;
                nl[j + 1] = c;
            });
            for (uint32_t j = 0; j < n_chunks; ++j) nl[j + 1] += nl[j];
            const uint32_t n_left = nl[n_chunks];
            pool.run(n_chunks, [&](size_t j) {
                uint32_t o, n; span(j, o, n);
                uint32_t l = first + nl[j], r = first + n_left + ((uint32_t)j * kChunk - nl[j]);
                for (uint32_t i = 0; i < n; ++i) { const Item &t = items[o + i]; if (bin_of(t.c[ax], lo, ext) < bb) scratch[l++] = t; else scratch[r++] = t; }
            });
            pool.run(n_chunks, [&](size_t j) { uint32_t o, n; span(j, o, n); std::memcpy(&items[o], &scratch[o], (size_t)n * sizeof(Item)); });
            mid = first + n_left;
            if (mid == first || mid == first + count) mid = first + count / 2;
        }
        return mid;
    }

    // The binary tree over items [0, n): 2n - 1 nodes in bn (numbered in whatever order the threads get to them — nothing downstream depends on
    // the numbers, only on the shape and the boxes, and those are a pure function of the items).  Three regimes by node size:
    //   > kBigItems   the main thread walks these few nodes, each split running chunk-parallel on the pool (split_big)
    //   > kTaskItems  a job of the shared queue: whichever thread takes it splits it alone and puts both halves back
    //   otherwise     a job of the shared queue: the whole subtree by the plain recursion
    int build(HostPool &pool, uint32_t n) {
        bn.resize(2 * (size_t)n - 1);
        scratch.resize(n > kBigItems ? n : 0);
        std::atomic<int> next_node{1};
        struct Job { int node; uint32_t first, count; int depth; };
        std::vector<Job> queue;                        // LIFO: the biggest pieces are split first, their halves are taken up at once
        {
            std::vector<Job> big{{0, 0u, n, 0}};
            while (!big.empty()) {
                const Job j = big.back(); big.pop_back();
                if (j.count <= kBigItems) { queue.push_back(j); continue; }
                const uint32_t mid = split_big(pool, bn[j.node], j.first, j.count, j.depth);
                const int l = next_node.fetch_add(2), r = l + 1;
                bn[j.node].left = l; bn[j.node].right = r;
                big.push_back({r, mid, j.first + j.count - mid, j.depth + 1});
                big.push_back({l, j.first, mid - j.first, j.depth + 1});
            }
        }
        scratch.clear(); scratch.shrink_to_fit();
        std::mutex qmu;
        std::atomic<uint32_t> items_left{n};           // items not yet inside a finished subtree: 0 = the tree is complete
        pool.run(pool.size(), [&](size_t) {
            for (uint32_t idle = 0;;) {
                Job j; bool have = false;
                { std::lock_guard<std::mutex> lk(qmu); if (!queue.empty()) { j = queue.back(); queue.pop_back(); have = true; } }
                if (!have) {
                    if (items_left.load(std::memory_order_acquire) == 0) return;
                    if (++idle < 64) cpu_relax(); else std::this_thread::yield();
                    continue;
                }
                idle = 0;
                if (j.count > kTaskItems) {
                    const uint32_t mid = split_node(bn[j.node], j.first, j.count, j.depth);      // (count > 1: never a leaf)
                    const int l = next_node.fetch_add(2), r = l + 1;
                    bn[j.node].left = l; bn[j.node].right = r;
                    std::lock_guard<std::mutex> lk(qmu);
                    queue.push_back({r, mid, j.first + j.count - mid, j.depth + 1});
                    queue.push_back({l, j.first, mid - j.first, j.depth + 1});
                } else {
                    int next = next_node.fetch_add(2 * (int)j.count - 2);                          // the subtree's nodes below its root, one block
                    build_serial(j.node, next, j.first, j.count, j.depth);
                    items_left.fetch_sub(j.count, std::memory_order_acq_rel);
                }
            }
        });
        return 0;
    }
};

// Builds the structure over n triangles (caller's triangles10 layout).  Outputs: nodes (kBvhNodeDwords per slot, slot 0 = root),
// tris (3 packets per triangle, node order), frame.  Returns false if a limit is hit.  n_threads = 0: build_threads().
template <class T>
bool bvh_build(const T *triangles10, uint32_t n, RawVec<uint32_t> &nodes, RawVec<HostPack4<T>> &tris, BvhFrame<T> &frame, unsigned n_threads = 0,
               RawVec<HostPack4<float>> *tris32 = nullptr) {
    if (n == 0 || n > kBvhMaxTris) return false;
    HostPool pool(n_threads ? n_threads : (n < 4096 ? 1u : build_threads()));      // (a small mesh is built faster than threads are started)
    using B8 = Bvh8Build<T>;
    constexpr uint32_t kChunk = B8::kChunk;
    const uint32_t n_chunks = (n + kChunk - 1) / kChunk;
    // ---- frame: centre and power-of-two scale from the bounds of all vertices
    struct VB { double lo[3], hi[3], amax; };
    std::vector<VB> vb(n_chunks);
    pool.run(n_chunks, [&](size_t j) {
        VB b; b.amax = 0;
        for (int k = 0; k < 3; ++k) { b.lo[k] = std::numeric_limits<double>::infinity(); b.hi[k] = -b.lo[k]; }
        const uint32_t i1 = std::min(n, ((uint32_t)j + 1) * kChunk);
        for (uint32_t i = (uint32_t)j * kChunk; i < i1; ++i)
            for (int v = 0; v < 3; ++v)
                for (int k = 0; k < 3; ++k) {
                    const double x = (double)triangles10[10 * (size_t)i + 3 * v + k];
                    b.lo[k] = std::min(b.lo[k], x); b.hi[k] = std::max(b.hi[k], x); b.amax = std::max(b.amax, std::fabs(x));
                }
        vb[j] = b;
    });
    double lo[3], hi[3], amax = 0;
    for (int k = 0; k < 3; ++k) { lo[k] = std::numeric_limits<double>::infinity(); hi[k] = -lo[k]; }
    for (const VB &b : vb) { for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], b.lo[k]); hi[k] = std::max(hi[k], b.hi[k]); } amax = std::max(amax, b.amax); }
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
    B8 b;
    b.items.resize(n);
    pool.run(n_chunks, [&](size_t j) {
        const uint32_t i1 = std::min(n, ((uint32_t)j + 1) * kChunk);
        for (uint32_t i = (uint32_t)j * kChunk; i < i1; ++i) {
            const T *t = triangles10 + 10 * (size_t)i;
            auto &it = b.items[i];
            it.idx = i;
            for (int k = 0; k < 3; ++k) {
                const double a = ((double)t[k] - centre[k]) * scale, bb = ((double)t[3 + k] - centre[k]) * scale, c = ((double)t[6 + k] - centre[k]) * scale;
                it.mn[k] = std::min(a, std::min(bb, c)) - pad; it.mx[k] = std::max(a, std::max(bb, c)) + pad;
                it.c[k] = (a + bb + c) / 3.0;
            }
        }
    });
    const int broot = b.build(pool, n);
    // ---- root box in the caller's coordinates, rounded outward in T
    for (int k = 0; k < 3; ++k) {
        const double mnw = b.bn[broot].mn[k] / scale + centre[k], mxw = b.bn[broot].mx[k] / scale + centre[k];
        T a = (T)mnw, c = (T)mxw;
        const T big = std::numeric_limits<T>::max();
        a = std::nextafter(std::nextafter(a, -big), -big); c = std::nextafter(std::nextafter(c, big), big);
        frame.root_mn[k] = a; frame.root_mx[k] = c; frame.centre[k] = (T)centre[k];
    }
    frame.scale = (T)scale;
    // ---- collapse to 8-wide nodes, breadth first, one level at a time: the nodes of a level are worked out as parallel jobs (which binary nodes they
    // swallow, slot assignment, quantised boxes), then numbered one after the other in the level's order — slots and triangle positions come out exactly as a
    // first-in-first-out walk would hand them out
    struct Pending { int bnode; uint32_t slot; };
    struct Made { int ent_at[8]; uint32_t imask, n_int, n_leaf, rank_word; uint32_t leaf_first[8]; uint32_t w[kBvhNodeDwords]; bool ok; };
    RawVec<Pending> level, next_level;
    level.push_back({broot, 0u});
    nodes.clear();
    nodes.reserve((size_t)n * kBvhNodeDwords / 2);
    nodes.resize(kBvhNodeDwords);
    RawVec<uint32_t> order;
    order.reserve(n);
    RawVec<Made> made;
    int depth = 0;
    auto make_node = [&](const Pending &cur, Made &m) {
        m.ok = false; m.imask = 0; m.n_int = 0; m.n_leaf = 0; m.rank_word = 0;
        std::memset(m.w, 0, sizeof m.w);
        // the up to 8 entries of this node: open the entry with the largest box until 8 or all are leaves
        int ent[8], ne = 0;
        const auto &root = b.bn[cur.bnode];
        if (root.left < 0) ent[ne++] = cur.bnode;
        else { ent[ne++] = root.left; ent[ne++] = root.right; }
        while (ne < 8) {
            int pick = -1; double pa = -1;
            for (int i = 0; i < ne; ++i)
                if (b.bn[ent[i]].left >= 0) { const double a = B8::area(b.bn[ent[i]].mn, b.bn[ent[i]].mx); if (a > pa) { pa = a; pick = i; } }
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
        bool slot_used[8] = {false}, ent_done[8] = {false};
        for (int s = 0; s < 8; ++s) m.ent_at[s] = -1;
        double off[8][3];
        for (int i = 0; i < ne; ++i) for (int k = 0; k < 3; ++k) off[i][k] = 0.5 * (b.bn[ent[i]].mn[k] + b.bn[ent[i]].mx[k]) - nc[k];
        for (int round = 0; round < ne; ++round) {
            double bestc = -std::numeric_limits<double>::infinity(); int bi = -1, bs = -1;
            for (int i = 0; i < ne; ++i) {
                if (ent_done[i]) continue;
                for (int s = 0; s < 8; ++s) {
                    if (slot_used[s]) continue;
                    const double c = ((s & 1) ? off[i][0] : -off[i][0]) + ((s & 2) ? off[i][1] : -off[i][1]) + ((s & 4) ? off[i][2] : -off[i][2]);
                    if (c > bestc) { bestc = c; bi = i; bs = s; }
                }
            }
            ent_done[bi] = true; slot_used[bs] = true; m.ent_at[bs] = ent[bi];
        }
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
            if (e > 120) return;
            eb[k] = (uint32_t)(e + 127);
        }
        uint8_t q[6][8];
        for (int s = 0; s < 8; ++s) { for (int a = 0; a < 3; ++a) { q[a][s] = 255; q[3 + a][s] = 0; } }
        for (int s = 0; s < 8; ++s) {
            if (m.ent_at[s] < 0) continue;
            const auto &c = b.bn[m.ent_at[s]];
            for (int k = 0; k < 3; ++k) {
                const double step = std::ldexp(1.0, (int)eb[k] - 127);
                double ql = std::floor((c.mn[k] - (double)p[k]) / step), qh = std::ceil((c.mx[k] - (double)p[k]) / step);
                ql = std::min(255.0, std::max(0.0, ql)); qh = std::min(255.0, std::max(0.0, qh));
                if ((double)p[k] + ql * step > c.mn[k] && ql > 0) ql -= 1;
                if ((double)p[k] + qh * step < c.mx[k] && qh < 255) qh += 1;
                if ((double)p[k] + ql * step > c.mn[k] || (double)p[k] + qh * step < c.mx[k]) return;   // cannot happen: p + 255 * step >= node max
                q[k][s] = (uint8_t)ql; q[3 + k][s] = (uint8_t)qh;
            }
            if (c.left >= 0) { m.imask |= 1u << s; ++m.n_int; }
            else {
                if (c.count != 1) return;
                m.rank_word |= m.n_leaf << (4 * s);
                m.leaf_first[m.n_leaf++] = c.first;
            }
        }
        uint32_t *w = m.w;
        w[0] = float_bits(p[0]); w[1] = float_bits(p[1]); w[2] = float_bits(p[2]);
        w[3] = eb[0] | (eb[1] << 8) | (eb[2] << 16) | (m.imask << 24);
        w[6] = m.rank_word;
        for (int a = 0; a < 6; ++a) {
            w[8 + 2 * a] = q[a][0] | (q[a][1] << 8) | (q[a][2] << 16) | ((uint32_t)q[a][3] << 24);
            w[9 + 2 * a] = q[a][4] | (q[a][5] << 8) | (q[a][6] << 16) | ((uint32_t)q[a][7] << 24);
        }
        m.ok = true;
    };
    RawVec<uint32_t> child_base_of, tri_base_of;
    while (!level.empty()) {
        ++depth;
        if (depth >= kBvhStack - 2) return false;
        made.resize(level.size());
        constexpr size_t kNodesPerJob = 64;
        const size_t n_jobs = (level.size() + kNodesPerJob - 1) / kNodesPerJob;
        pool.run(n_jobs, [&](size_t j) {
            const size_t i1 = std::min(level.size(), (j + 1) * kNodesPerJob);
            for (size_t i = j * kNodesPerJob; i < i1; ++i) make_node(level[i], made[i]);
        });
        // number the level: child blocks and triangle positions in the level's order (a prefix sum, the only serial part), then write in parallel
        child_base_of.resize(level.size()); tri_base_of.resize(level.size());
        uint32_t slots = (uint32_t)(nodes.size() / kBvhNodeDwords), n_order = (uint32_t)order.size(), n_next = 0;
        for (size_t i = 0; i < level.size(); ++i) {
            const Made &m = made[i];
            if (!m.ok) return false;
            tri_base_of[i] = n_order; n_order += m.n_leaf;
            child_base_of[i] = m.n_int ? slots : 0u;
            if (m.n_int) { if ((uint64_t)slots + 8 > (1u << 24)) return false; slots += 8; n_next += m.n_int; }      // a stack entry holds 24 bits of a child base
        }
        nodes.resize((size_t)slots * kBvhNodeDwords);          // (default-initialised: the jobs below write every word of every new slot)
        order.resize(n_order);
        next_level.resize(n_next);
        // where each node's children go in the next level's list: the same prefix, over n_int
        RawVec<uint32_t> next_at(level.size());
        { uint32_t c = 0; for (size_t i = 0; i < level.size(); ++i) { next_at[i] = c; c += made[i].n_int; } }
        pool.run(n_jobs, [&](size_t j) {
            const size_t i1 = std::min(level.size(), (j + 1) * kNodesPerJob);
            for (size_t i = j * kNodesPerJob; i < i1; ++i) {
                Made &m = made[i];
                for (uint32_t t = 0; t < m.n_leaf; ++t) order[tri_base_of[i] + t] = m.leaf_first[t];
                if (m.n_int) {
                    uint32_t at = next_at[i];
                    for (int s = 0; s < 8; ++s) {
                        uint32_t *h = &nodes[((size_t)child_base_of[i] + s) * kBvhNodeDwords];
                        if (!(m.imask & (1u << s))) {           // a hole: give it empty children so that a stray visit finds nothing
                            std::memset(h, 0, kBvhNodeDwords * sizeof(uint32_t));
                            h[8] = h[9] = h[10] = h[11] = h[12] = h[13] = 0xFFFFFFFFu;
                            continue;
                        }
                        next_level[at++] = {m.ent_at[s], child_base_of[i] + (uint32_t)s};      // (its words are written when its own level is made)
                    }
                }
                m.w[4] = child_base_of[i]; m.w[5] = tri_base_of[i];
                std::memcpy(&nodes[(size_t)level[i].slot * kBvhNodeDwords], m.w, sizeof m.w);
            }
        });
        level.swap(next_level);
    }
    if (order.size() != n) return false;
    frame.n_slots = (uint32_t)(nodes.size() / kBvhNodeDwords);
    frame.depth = depth;
    tris.resize(3 * (size_t)n);
    if (tris32) tris32->resize(3 * (size_t)n);
    pool.run(n_chunks, [&](size_t j) {
        const uint32_t i1 = std::min(n, ((uint32_t)j + 1) * kChunk);
        for (uint32_t i = (uint32_t)j * kChunk; i < i1; ++i) {
            const uint32_t oi = b.items[order[i]].idx;
            const T *t = triangles10 + 10 * (size_t)oi;
            const T e1[3] = {(T)(t[3] - t[0]), (T)(t[4] - t[1]), (T)(t[5] - t[2])};      // edge1 = v1 - v0, :149
            const T e2[3] = {(T)(t[6] - t[0]), (T)(t[7] - t[1]), (T)(t[8] - t[2])};      // edge2 = v2 - v0, :150
            tris[3 * (size_t)i + 0] = {t[0], t[1], t[2], bits_to_real<T>(oi)};
            tris[3 * (size_t)i + 1] = {e1[0], e1[1], e1[2], bits_to_real<T>((uint32_t)t[9] - 1u)};
            tris[3 * (size_t)i + 2] = {e2[0], e2[1], e2[2], (T)0};
            if (tris32) {
                // The Float32 screening record of a Float64 walk (spira_device.h, tri_maybe_f32): the same triangle in the normalised frame, rounded to
                // Float32 — v0 with an absolute error of 2^-24 of a coordinate (|coordinate| <= ~0.55), the edges with a relative one — and
                // L >= max(|e1|_inf, |e2|_inf), the scale of the screen's error bounds.
                float v[3], f1[3], f2[3], L = 0.0f;
                for (int k = 0; k < 3; ++k) {
                    v[k] = (float)(((double)t[k] - centre[k]) * scale);
                    f1[k] = (float)((double)e1[k] * scale); f2[k] = (float)((double)e2[k] * scale);
                    L = std::max(L, std::max(std::fabs(f1[k]), std::fabs(f2[k])));
                }
                L = std::nextafter(L, std::numeric_limits<float>::infinity());
                (*tris32)[3 * (size_t)i + 0] = {v[0], v[1], v[2], bits_float(oi)};
                (*tris32)[3 * (size_t)i + 1] = {f1[0], f1[1], f1[2], L};
                (*tris32)[3 * (size_t)i + 2] = {f2[0], f2[1], f2[2], 0.0f};
            }
        }
    });
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
