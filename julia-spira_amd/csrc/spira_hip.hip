// spira_hip.hip — host side of libspira_hip.so: device contexts, workspaces, the wavefront
// pass loop and the C ABI declared in include/spira_hip.h.  gfx950 (MI355X) only.
//
// Replaces the host loops of render_hybrid_gpu (src/spira-metal-optimized.jl:1228-1343): the
// reference launches >= spp*(1 + max_depth*(12*n_spheres + 4)) synchronous kernels with host
// round trips per depth; here one pass = one launch of the persistent path kernel
// (mesh scenes: a parking launch + a fat-wave launch; SPIRA_KERNEL_BOUNCE: max_depth bounce kernels)
// + 1 resolve kernel, fully asynchronous on one HIP stream, the live-ray counts staying on the
// device, and a pass carries `slots` samples of every pixel of the tile at once.
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <rccl/rccl.h>      // types and prototypes only: librccl is opened at run time (dlopen), never linked
#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/spira_hip.h"
#include "../../include/spira_spd.h"
#include "spira_device.h"
#include "spira_bvh.h"
#include "spira_validate.h"

// The library is built from this one file as THREE translation units (Makefile), because what the optimiser does to one family of kernels it undoes
// on another (profiles/r03_compiler_flags.md):
//   SPIRA_TU_F32      render_impl<float> / trace_impl<float> and every kernel they launch, with -fno-slp-vectorize.  The SLP vectoriser pairs Float32
//                     operations into v_pk_mul/add/fma_f32 and pays for every pair with register moves: without it S1 runs 13 % and the closed box S3
//                     20 % faster in Float32 (Float64 has no packed arithmetic to form: indifferent, the mesh scene 4 % better off WITH the pass).
//   SPIRA_TU_F64MESH  the Float64 path kernels of mesh scenes (k_path<double, ., BVH = true, ...>), with the compiler's defaults.
//   SPIRA_TU_MAIN     the C ABI, the host runtime and every other Float64 kernel, with -mllvm -two-entry-phi-node-folding-threshold=1: SimplifyCFG then
//                     turns far fewer two-sided branches into selects, which is 4 % of k_path on S1 in Float64 (1 % on the closed box) — and 5.5 % the
//                     other way on the mesh kernels, hence their own unit.  (No effect on the Float32 kernels or the secondary Float64 ones.)
// None of the macros defined (make stats, tests): one translation unit, the compiler's defaults.  The state below is shared by all units (inline
// variables of a named namespace: one instance in the library); the functions further down are internal to each unit.
#if (defined(SPIRA_TU_MAIN) + defined(SPIRA_TU_F32) + defined(SPIRA_TU_F64MESH)) > 1
#error "SPIRA_TU_MAIN, SPIRA_TU_F32 and SPIRA_TU_F64MESH are three different translation units"
#endif
struct spira_scene;
// A host-output frame rendered as `count` consecutive row slabs (render_host_slabs): slab `index` > 0 continues the call of slab 0 — same scene (already in
// the context's store), same counters and event brackets (they add up), and the caller holds the context's lock across all of them.
struct SlabCtl { uint32_t index, count; };
namespace spira_tu {      // defined in the SPIRA_TU_F32 unit, called from the SPIRA_TU_MAIN one
int render_impl_f32(const spira_scene *h, const float *spheres5, const float *materials8, const float *triangles10, const float *camera12, const spira_params *p,
                    float *out_hdr, float *out_img, bool out_on_device, void *user_stream, bool progressive, uint32_t sample0, uint32_t *rng_states, const SlabCtl *slab);
int trace_impl_f32(const float *spheres5, const float *materials8, const float *triangles10, const float *camera12, const spira_params *p,
                   uint32_t n_paths, const uint32_t *ijs, int *prims, float *ts, float *dirs, float *radiance);
// defined in the SPIRA_TU_F64MESH unit: launch_path<double> of a mesh scene (PathArgs::mesh_mode 0 or 1) and launch_path_resume<double> (mode 2)
int launch_path_mesh_f64(int R, dim3 grid, size_t lds, hipStream_t st, const spira::PathArgs<double> &a, int spec);
void launch_path_resume_f64(int R, dim3 grid, size_t lds, hipStream_t st, const spira::PathArgs<double> &a);
}

namespace spira_host {

inline thread_local std::string tl_err;
inline thread_local int tl_device = 0;
inline thread_local hipError_t tl_lds_optin = hipSuccess;      // a refused LDS opt-in of this thread's call (launch_lds / lds_optin_failed), whichever unit launched

inline int fail(int code, const std::string &msg) { tl_err = msg; return code; }

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess)                                                                           \
            return fail(SPIRA_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));                \
    } while (0)

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return 0;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        hipError_t e = hipMalloc(&p, bytes);
        if (e != hipSuccess) return fail(SPIRA_E_HIP, std::string("hipMalloc: ") + hipGetErrorString(e));
        cap = bytes;
        return 0;
    }
    void release() { if (p) { (void)hipFree(p); p = nullptr; cap = 0; } }
};

// A scene resident on one device: the flat arrays exactly as the ABI takes them (+ the BVH of a large mesh).
// The context owns one (re-uploaded per call, the BVH cached by a hash of the triangle bytes); every
// spira_scene handle owns one (validated, built and uploaded once by spira_scene_create_*).
struct SceneStore {
    DevBuf arrays, bvh_nodes, bvh_tris, bvh_tris32;      // (bvh_tris32: Float64 scenes only — the Float32 screening records of the walk)
    uint32_t ns = 0, nm = 0, nt = 0;
    uint64_t bvh_hash = 0; uint32_t bvh_n = 0, bvh_slots = 0; int bvh_prec = 0, bvh_depth = 0;
    bool moderate = false;                    // every coordinate / radius of ordinary magnitude (spira::scene_scale_moderate): speculative division pays
    void release() { arrays.release(); bvh_nodes.release(); bvh_tris.release(); bvh_tris32.release(); bvh_hash = 0; bvh_n = 0; }
};

struct Ctx {
    bool init = false;
    int device = -1;
    int num_cus = 256;
    hipStream_t stream = nullptr;
    hipStream_t copy_stream = nullptr;        // host-output frames rendered as row slabs: a slab's copies run here, beside the next slab's kernels
    hipEvent_t ev_slab = nullptr;
    DevBuf qA[2], qB[2], qC[2], qR[2], qK[2], qX[2], mesh_list, mesh_count, redo, L, accum, counts, blkstats, stats, out_tmp, trace, rng;
    DevBuf hyb_state, hyb_mat, hyb_flags;          // SPIRA_SEM_HYBRID: per-pixel ray state between its launches
    DevBuf spd32, spd64;                          // SPIRA_EXT_SPECTRAL: the SPD table, uploaded once per precision
    DevBuf multi_tile, multi_stack, multi_full;   // spira_render_multi_*: this device's tile; device 0: the gathered tiles, the frame
    SceneStore scene;                         // the scene of the current call (host-array entry points)
    spira::Stats *h_stats = nullptr;          // pinned
    void *h_stage = nullptr; size_t h_stage_cap = 0;   // pinned staging of a host-output frame (copy_out below)
    std::vector<hipEvent_t> stage_ev;         // one per chunk in flight
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    hipEvent_t ev_done = nullptr;             // end of the last call that used the workspaces, on whatever stream it ran
    bool have_done = false;
    std::vector<hipEvent_t> ev_pool;          // profile mode: pairs around bounce launches
    size_t ev_used = 0;
    std::vector<hipEvent_t> ev_mid;           // mesh passes run as two launches: one event between them (-> spira_counters.walk_kernel_ms)
    std::vector<size_t> ev_mid_end;           // ... and the ev_pool index of the event that closes that pass's bracket
    size_t ev_mid_used = 0;
    spira_counters last{};
    bool last_valid = false, last_pending = false;
    hipStream_t last_stream = nullptr;
    std::recursive_mutex mu;                  // (recursive: a host-output frame rendered as row slabs holds it across its slabs' render calls)
};

constexpr int kMaxDevices = 16;
inline Ctx g_ctx[kMaxDevices];

}  // namespace spira_host
using namespace spira_host;

// The opaque scene handle of the C ABI (spira_scene_create_* / spira_scene_destroy).
struct spira_scene {
    uint32_t magic;        // kSceneMagic while alive
    int device;
    int prec;              // sizeof(T) the scene was created in
    SceneStore store;
    // spira_scene_create_multi_*: the same scene resident on devices 1 .. n_replicas-1 as well (this handle is device 0's)
    int n_replicas = 1;
    spira_scene *replica[16] = {};
};

namespace {
constexpr uint32_t kSceneMagic = 0x53504952u;   // "SPIR"

int get_ctx(Ctx **out) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(SPIRA_E_NO_DEVICE, std::string("no HIP device: ") + (e != hipSuccess ? hipGetErrorString(e) : "device count is 0"));
    if (tl_device < 0 || tl_device >= n || tl_device >= kMaxDevices) return fail(SPIRA_E_INVALID, "device index out of range");
    Ctx &c = g_ctx[tl_device];
    HIP_TRY(hipSetDevice(tl_device));
    std::lock_guard<std::recursive_mutex> init_lock(c.mu);     // two threads must not initialise one context twice
    if (!c.init) {
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, tl_device));
        if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
            return fail(SPIRA_E_NO_DEVICE, std::string("libspira_hip is built for gfx950 only, found ") + prop.gcnArchName);
        c.num_cus = prop.multiProcessorCount;
        c.device = tl_device;
        HIP_TRY(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
        HIP_TRY(hipEventCreate(&c.ev_start));
        HIP_TRY(hipEventCreate(&c.ev_stop));
        HIP_TRY(hipEventCreateWithFlags(&c.ev_done, hipEventDisableTiming));
        HIP_TRY(hipHostMalloc((void **)&c.h_stats, sizeof(spira::Stats), hipHostMallocDefault));
        c.init = true;
    }
    *out = &c;
    return 0;
}

// SPIRA_LOG_TIMING=1: host-side phase times of a call on stderr (where a first call's milliseconds go: context, validation, tree build, uploads, launches)
struct Lap {
    bool on; const char *what; std::chrono::steady_clock::time_point t0, t;
    explicit Lap(const char *w) : on(std::getenv("SPIRA_LOG_TIMING") != nullptr), what(w) { if (on) { t0 = t = std::chrono::steady_clock::now(); std::fprintf(stderr, "[spira %s]", what); } }
    void operator()(const char *phase) {
        if (!on) return;
        const auto n = std::chrono::steady_clock::now();
        std::fprintf(stderr, " %s %.3f", phase, std::chrono::duration<double, std::milli>(n - t).count());
        t = n;
    }
    ~Lap() { if (on) std::fprintf(stderr, " | total %.3f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count()); }
};

uint32_t env_u32(const char *name, uint32_t dflt) {
    const char *s = std::getenv(name);
    if (!s || !*s) return dflt;
    return (uint32_t)std::strtoul(s, nullptr, 10);
}

uint32_t stripe_rows(uint32_t height, uint32_t sh, uint32_t n, uint32_t r) {
    if (sh == 0 || n == 0) return 0;
    uint32_t rows = 0;
    for (uint32_t y0 = r * sh; y0 < height; y0 += n * sh) rows += std::min(sh, height - y0);
    return rows;
}

// Scene arrays: pointers, counts, material indices, finiteness (spira_validate.h) and the LDS budget.
template <class T>
int validate_scene(const T *spheres5, const T *materials8, const T *triangles10, uint32_t n_spheres, uint32_t n_materials, uint32_t nt) {
    if (n_spheres > SPIRA_MAX_LDS_SPHERES) return fail(SPIRA_E_LIMIT, "more than 1024 spheres");
    if (nt > SPIRA_MAX_TRIANGLES) return fail(SPIRA_E_LIMIT, "more than 2^24 triangles");
    const char *msg = nullptr;
    if (int rc = spira::scene_arrays_check<T>(spheres5, materials8, triangles10, n_spheres, n_materials, nt, &msg)) return fail(rc, msg);
    if (spira::scene_lds_bytes<T>(n_spheres, n_materials, nt > SPIRA_LDS_TRIANGLES ? 0 : nt) > 120 * 1024)
        return fail(SPIRA_E_LIMIT, "scene does not fit in LDS");
    return 0;
}

// Render parameters (nt = triangles actually present in the scene of this call).
int validate_params(const void *camera12, const spira_params *p, uint32_t nt, uint32_t *rows_out) {
    if (!p) return fail(SPIRA_E_INVALID, "params is NULL");
    if (!camera12) return fail(SPIRA_E_INVALID, "camera12 is NULL");
    if (p->width < 2 || p->height < 2) return fail(SPIRA_E_INVALID, "width and height must be >= 2 (u = (i-1+rand)/(W-1))");
    if ((uint64_t)p->width * p->height > 0x7FFFFFFFull) return fail(SPIRA_E_LIMIT, "image larger than 2^31 pixels");
    if (p->spp < 1 || p->spp > SPIRA_MAX_SPP) return fail(SPIRA_E_LIMIT, "spp out of range [1, 2^24]");
    if (p->max_depth > SPIRA_MAX_DEPTH) return fail(SPIRA_E_LIMIT, "max_depth > 255");
    const uint32_t sem = p->flags & SPIRA_SEM_MASK;
    if (sem != SPIRA_SEM_A && sem != SPIRA_SEM_CPU && sem != SPIRA_SEM_METAL && sem != SPIRA_SEM_HYBRID) return fail(SPIRA_E_UNSUPPORTED, "unknown integrator semantics");
    if (sem != SPIRA_SEM_A && nt) return fail(SPIRA_E_UNSUPPORTED, "SPIRA_SEM_CPU / SPIRA_SEM_METAL / SPIRA_SEM_HYBRID are sphere-only, like their sources");
    if (sem == SPIRA_SEM_HYBRID && p->rows != 0)
        return fail(SPIRA_E_UNSUPPORTED, "SPIRA_SEM_HYBRID renders whole images only (the reference ends a sample when no ray of the IMAGE hits anything): rows must be 0");
    uint32_t kern = p->flags & SPIRA_KERNEL_MASK;
    if (kern != SPIRA_KERNEL_DEFAULT && kern != SPIRA_KERNEL_WAVEFRONT && kern != SPIRA_KERNEL_MEGA && kern != SPIRA_KERNEL_BOUNCE) return fail(SPIRA_E_UNSUPPORTED, "unknown kernel organisation");
    if (p->flags & (SPIRA_EXT_DIELECTRIC | SPIRA_EXT_SPECTRAL)) {
        if (sem != SPIRA_SEM_A) return fail(SPIRA_E_UNSUPPORTED, "SPIRA_EXT_* extensions apply to SPIRA_SEM_A only");
        if (kern == SPIRA_KERNEL_BOUNCE) return fail(SPIRA_E_UNSUPPORTED, "SPIRA_EXT_* extensions are not built into the per-bounce organisation");
    }
    uint32_t rows = p->rows;
    if (rows == 0) rows = p->height;
    else if (p->stripe_count > 1) {
        if (p->stripe_h == 0 || p->stripe_rank >= p->stripe_count) return fail(SPIRA_E_INVALID, "bad stripe parameters");
        if (rows != stripe_rows(p->height, p->stripe_h, p->stripe_count, p->stripe_rank))
            return fail(SPIRA_E_INVALID, "rows != spira_stripe_rows(height, stripe_h, stripe_count, stripe_rank)");
    } else if ((uint64_t)p->row0 + rows > p->height) return fail(SPIRA_E_INVALID, "row0 + rows > height");
    *rows_out = rows;
    return 0;
}

template <class T>
void fill_const(spira::RenderConst<T> &rc, const T *cam, const spira_params *p, uint32_t rows, uint32_t slots) {
    rc.cam_origin = {cam[0], cam[1], cam[2]};
    rc.cam_llc = {cam[3], cam[4], cam[5]};
    rc.cam_hor = {cam[6], cam[7], cam[8]};
    rc.cam_ver = {cam[9], cam[10], cam[11]};
    rc.width = p->width; rc.height = p->height; rc.spp = p->spp; rc.max_depth = p->max_depth;
    uint32_t lo = (uint32_t)p->seed, hi = (uint32_t)(p->seed >> 32);
    rc.sA = spira::mix32(spira::mix32(lo + 0x9E3779B9u) ^ hi);
    rc.sB = spira::mix32(spira::mix32(hi + 0x85EBCA6Bu) ^ lo);
    rc.flags = p->flags;
    rc.rows = rows;
    if (p->rows == 0) { rc.row0 = 0; rc.stripe_h = 0; rc.stripe_count = 0; rc.stripe_rank = 0; }
    else { rc.row0 = p->row0; rc.stripe_h = p->stripe_h; rc.stripe_count = p->stripe_count; rc.stripe_rank = p->stripe_rank; }
    rc.tile_pixels = rows * p->width;
    rc.slots = slots;
    rc.sample0 = 0;
    rc.fd_tile = spira::fastdiv_make(rc.tile_pixels);
    rc.fd_width = spira::fastdiv_make(rc.width);
    rc.fd_stripe = spira::fastdiv_make(rc.stripe_h ? rc.stripe_h : 1);
}

// The magic-number division is exact by construction; verify it anyway on the values a render can see.
bool fastdiv_selfcheck(uint32_t d, uint32_t n_max) {
    const spira::FastDiv f = spira::fastdiv_make(d);
    uint32_t probes[] = {0u, 1u, d - 1, d, d + 1, 2 * d - 1, 2 * d, n_max / 2, n_max - 1, n_max, 0x7FFFFFFFu, 0xFFFFFFFFu};
    for (uint32_t n : probes)
        if (spira::fastdiv(n, f) != n / d) return false;
    uint32_t x = 0x12345u;
    for (int i = 0; i < 512; ++i) {
        x = spira::mix32(x + i);
        if (spira::fastdiv(x, f) != x / d) return false;
    }
    return true;
}

template <class T>
void scene_pointers(const SceneStore &s, spira::SceneGlobal<T> &g) {
    const bool use_bvh = s.nt > SPIRA_LDS_TRIANGLES;
    const uint32_t nt_lds = use_bvh ? 0 : s.nt;
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t ns_b = (size_t)s.ns * 5 * sizeof(T), nm_b = (size_t)s.nm * 8 * sizeof(T);
    char *base = (char *)s.arrays.p;
    g.spheres5 = (const T *)base;
    g.materials8 = (const T *)(base + up(ns_b));
    g.triangles10 = (const T *)(base + up(ns_b) + up(nm_b));
    g.n_spheres = s.ns; g.n_materials = s.nm; g.n_triangles = nt_lds;
    g.bvh_nodes = use_bvh ? (const uint4 *)s.bvh_nodes.p : nullptr;
    g.bvh_frame = use_bvh ? (const spira::Pack4<T> *)s.bvh_tris.p : nullptr;          // 3 packets ahead of the triangles
    g.bvh_tris = use_bvh ? (const spira::Pack4<T> *)s.bvh_tris.p + 3 : nullptr;
    g.bvh_tris32 = (use_bvh && sizeof(T) == 8) ? (const uint4 *)s.bvh_tris32.p : nullptr;
    g.n_bvh_tris = use_bvh ? s.nt : 0;
    g.bvh_slots = use_bvh ? s.bvh_slots : 0;
}

// A mesh's tree as built on the host: one build can be uploaded to several devices (spira_scene_create_multi_*).
template <class T> struct HostBvh {
    spira::RawVec<uint32_t> nodes;
    spira::RawVec<spira::HostPack4<T>> tris;
    spira::RawVec<spira::HostPack4<float>> tris32;          // Float64 only
    spira::HostPack4<T> frame[3];
    uint32_t slots = 0; int depth = 0; bool built = false;
};
template <class T>
int host_bvh_build(const T *triangles10, uint32_t nt, HostBvh<T> &hb) {
    spira::BvhFrame<T> fr{};
#ifdef SPIRA_BVH_SCREEN
    constexpr bool kScreenRecords = sizeof(T) == 8;      // experiment build: Float64 walks screen their triangles in Float32 (spira_device.h, tri_screen_f32)
#else
    constexpr bool kScreenRecords = false;
#endif
    if (!spira::bvh_build<T>(triangles10, nt, hb.nodes, hb.tris, fr, 0, kScreenRecords ? &hb.tris32 : nullptr))
        return fail(SPIRA_E_LIMIT, "BVH build failed (tree too deep / too many triangles)");
    hb.frame[0] = {fr.root_mn[0], fr.root_mn[1], fr.root_mn[2], (T)0}; hb.frame[1] = {fr.root_mx[0], fr.root_mx[1], fr.root_mx[2], (T)0};
    hb.frame[2] = {fr.centre[0], fr.centre[1], fr.centre[2], fr.scale};
    hb.slots = fr.n_slots; hb.depth = fr.depth; hb.built = true;
    return 0;
}

// Upload host arrays into `s`.  The small arrays go asynchronously on `st`; a mesh above SPIRA_LDS_TRIANGLES gets a
// BVH built on the host (once per distinct triangle array: keyed by a hash of its bytes) and copied synchronously —
// `prev_done` (the event of the last call that may still be traversing the old tree) is waited for first.
template <class T>
int scene_upload(SceneStore &s, hipStream_t st, hipEvent_t prev_done, const T *spheres5, const T *materials8, const T *triangles10,
                 uint32_t n_spheres, uint32_t n_materials, uint32_t nt, HostBvh<T> *shared = nullptr) {
    const bool use_bvh = nt > SPIRA_LDS_TRIANGLES;
    const uint32_t nt_lds = use_bvh ? 0 : nt;
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t ns_b = (size_t)n_spheres * 5 * sizeof(T), nm_b = (size_t)n_materials * 8 * sizeof(T), nt_b = (size_t)nt_lds * 10 * sizeof(T);
    Lap lap("scene_upload");
    if (int rc = s.arrays.ensure(up(ns_b) + up(nm_b) + up(nt_b) + 256)) return rc;
    s.ns = n_spheres; s.nm = n_materials; s.nt = nt;
    s.moderate = spira::scene_scale_moderate<T>(spheres5, triangles10, n_spheres, nt);
    lap("alloc+scale");
    spira::SceneGlobal<T> g;
    scene_pointers<T>(s, g);
    if (ns_b) HIP_TRY(hipMemcpyAsync((void *)g.spheres5, spheres5, ns_b, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync((void *)g.materials8, materials8, nm_b, hipMemcpyHostToDevice, st));
    if (nt_b) HIP_TRY(hipMemcpyAsync((void *)g.triangles10, triangles10, nt_b, hipMemcpyHostToDevice, st));
    if (use_bvh) {
        // the context's store keeps the tree of the last mesh it saw, found again by a hash of the triangle bytes; a handle's store is filled once (no hash)
        const uint64_t h = shared ? 0 : spira::bytes_hash64(triangles10, (size_t)nt * 10 * sizeof(T));
        lap("hash");
        if (shared || s.bvh_hash != h || s.bvh_n != nt || s.bvh_prec != (int)sizeof(T)) {
            HostBvh<T> local;
            HostBvh<T> &hb = shared ? *shared : local;
            if (!hb.built) { if (int rc = host_bvh_build<T>(triangles10, nt, hb)) return rc; }
            lap("bvh_build");
            if (prev_done) HIP_TRY(hipEventSynchronize(prev_done));      // nobody still reads the tree that is about to be replaced
            // (+ one record of padding each: a walk's trip loads 5 / 6 x 16 bytes from a node or a triangle alike, spira_device.h bvh8_step)
            if (int rc = s.bvh_nodes.ensure(hb.nodes.size() * sizeof(hb.nodes[0]) + 128)) return rc;
            if (int rc = s.bvh_tris.ensure(sizeof hb.frame + hb.tris.size() * sizeof(hb.tris[0]) + 128)) return rc;
            const size_t tris32_b = hb.tris32.size() * sizeof(spira::HostPack4<float>);
            if (tris32_b) { if (int rc = s.bvh_tris32.ensure(tris32_b + 128)) return rc; }
            lap("hipMalloc");
            // The three arrays go up as asynchronous copies out of the vectors pinned in place (hipHostRegister: ~11 MB for 82 k triangles in Float64;
            // a hipMemcpy from pageable memory is staged by the runtime chunk by chunk on this thread), one wait at the end: the host vectors may die
            // with this scope.  Pinning refused (a limit on locked memory): the plain copies.
            const size_t nodes_b = hb.nodes.size() * sizeof(hb.nodes[0]), tris_b = hb.tris.size() * sizeof(hb.tris[0]);
            const bool pin_n = hipHostRegister(hb.nodes.data(), nodes_b, hipHostRegisterDefault) == hipSuccess;
            const bool pin_t = hipHostRegister(hb.tris.data(), tris_b, hipHostRegisterDefault) == hipSuccess;
            const bool pin_s = tris32_b && hipHostRegister(hb.tris32.data(), tris32_b, hipHostRegisterDefault) == hipSuccess;
            if (!pin_n || !pin_t || (tris32_b && !pin_s)) (void)hipGetLastError();
            hipError_t e1 = hipMemcpyAsync(s.bvh_nodes.p, hb.nodes.data(), nodes_b, hipMemcpyHostToDevice, st);
            hipError_t e2 = hipMemcpyAsync(s.bvh_tris.p, hb.frame, sizeof hb.frame, hipMemcpyHostToDevice, st);
            hipError_t e3 = hipMemcpyAsync((char *)s.bvh_tris.p + sizeof hb.frame, hb.tris.data(), tris_b, hipMemcpyHostToDevice, st);
            hipError_t e5 = tris32_b ? hipMemcpyAsync(s.bvh_tris32.p, hb.tris32.data(), tris32_b, hipMemcpyHostToDevice, st) : hipSuccess;
            hipError_t e4 = hipStreamSynchronize(st);
            if (pin_n) (void)hipHostUnregister(hb.nodes.data());
            if (pin_t) (void)hipHostUnregister(hb.tris.data());
            if (pin_s) (void)hipHostUnregister(hb.tris32.data());
            HIP_TRY(e1); HIP_TRY(e2); HIP_TRY(e3); HIP_TRY(e5); HIP_TRY(e4);
            lap("pin+copy");
            const int depth = hb.depth;
            s.bvh_slots = hb.slots;
            s.bvh_hash = h; s.bvh_n = nt; s.bvh_prec = (int)sizeof(T); s.bvh_depth = depth;
        }
    }
    return 0;
}

// Launch with a dynamic LDS block; above 64 KB the function has to be told first (up to the CU's 160 KB).
// A refused opt-in is remembered (thread-local) and turned into SPIRA_E_LIMIT by lds_optin_failed() before the call returns.
template <class K, class... Args>
void launch_lds(K kernel, dim3 grid, dim3 block, size_t lds, hipStream_t st, Args... args) {
    if (lds > 64 * 1024) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { tl_lds_optin = e; return; }          // do not launch a kernel that cannot get its LDS
    }
    hipLaunchKernelGGL(kernel, grid, block, lds, st, args...);
}
int lds_optin_failed() {
    if (tl_lds_optin == hipSuccess) return 0;
    const hipError_t e = tl_lds_optin;
    tl_lds_optin = hipSuccess;
    return fail(SPIRA_E_LIMIT, std::string("the device refused the kernel's dynamic LDS size (hipFuncSetAttribute): ") + hipGetErrorString(e));
}

template <class T, bool FIRST, bool BVH>
void launch_bounce_r(int R, dim3 grid, size_t lds, hipStream_t st, const spira::BounceArgs<T> &a) {
    switch (R) {
    case 2: launch_lds(spira::k_bounce<T, FIRST, 2, BVH>, grid, dim3(spira::kBlock), lds, st, a); break;
    default: launch_lds(spira::k_bounce<T, FIRST, 1, BVH>, grid, dim3(spira::kBlock), lds, st, a); break;
    }
}
template <class T, bool FIRST>
void launch_bounce(int R, dim3 grid, size_t lds, hipStream_t st, const spira::BounceArgs<T> &a) {
    if (a.scene.n_bvh_tris) launch_bounce_r<T, FIRST, true>(R, grid, lds, st, a);
    else launch_bounce_r<T, FIRST, false>(R, grid, lds, st, a);
}

// k_path.  `spec`: the speculative-division instantiation first (PathArgs::redo = its per-wave report), then the exact one over the
// waves it reported (spira_device.h, SpecDiv) — two launches, the second normally a grid of workgroups that return at once.
// spec == 2 (SPIRA_SPEC_DIV=2, tests): every wave is reported, i.e. the whole pass is rendered twice.
// MODE: PathArgs::mesh_mode as a template argument (mesh scenes: 0 one launch, 1 the parking launch of two; 2 is launch_path_resume below).
// TRI = false: the scene holds no LDS-resident triangles (spheres only, or spheres + a BVH mesh): instantiations without the triangle scan
template <class T, bool BVH, int MODE>
int launch_path_mode(int R, dim3 grid, size_t lds, hipStream_t st, spira::PathArgs<T> a, int spec) {
    const bool ext = (a.rc.flags & (SPIRA_EXT_DIELECTRIC | SPIRA_EXT_SPECTRAL)) != 0;
    const bool tri = a.scene.n_triangles != 0;
    const dim3 blk(spira::kBlock);
    if (ext) {           // extension instantiations (R = 2 only); like the default kernels, without the LDS triangle scan where the scene has none
        if (spec) {
            a.redo_only = spec == 2 ? 2 : 0;          // (2: every wave will be rendered again, whatever it reports)
            if (tri) launch_lds(spira::k_path<T, 2, BVH, true, true, MODE, true>, grid, blk, lds, st, a);
            else launch_lds(spira::k_path<T, 2, BVH, true, true, MODE, false>, grid, blk, lds, st, a);
            if (spec == 2) HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)a.redo, 1, (size_t)grid.x * (spira::kBlock / 64), st));
            a.redo_only = 1;
        } else { a.redo = nullptr; a.redo_only = 0; }
        if (tri) launch_lds(spira::k_path<T, 2, BVH, true, false, MODE, true>, grid, blk, lds, st, a);
        else launch_lds(spira::k_path<T, 2, BVH, true, false, MODE, false>, grid, blk, lds, st, a);
    } else if (R == 2) {
        if (spec) {
            a.redo_only = spec == 2 ? 2 : 0;
            if (tri) launch_lds(spira::k_path<T, 2, BVH, false, true, MODE, true>, grid, blk, lds, st, a);
            else launch_lds(spira::k_path<T, 2, BVH, false, true, MODE, false>, grid, blk, lds, st, a);
            if (spec == 2) HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)a.redo, 1, (size_t)grid.x * (spira::kBlock / 64), st));
            a.redo_only = 1;
        } else { a.redo = nullptr; a.redo_only = 0; }
        if (tri) launch_lds(spira::k_path<T, 2, BVH, false, false, MODE, true>, grid, blk, lds, st, a);
        else launch_lds(spira::k_path<T, 2, BVH, false, false, MODE, false>, grid, blk, lds, st, a);
    } else {
        a.redo = nullptr; a.redo_only = 0;
        launch_lds(spira::k_path<T, 1, BVH, false, false, MODE>, grid, blk, lds, st, a);
    }
    return 0;
}
template <class T>
int launch_path(int R, dim3 grid, size_t lds, hipStream_t st, const spira::PathArgs<T> &a, int spec) {
    if (!a.scene.n_bvh_tris) return launch_path_mode<T, false, 0>(R, grid, lds, st, a, spec);
#ifdef SPIRA_TU_MAIN
    if constexpr (sizeof(T) == 8) return spira_tu::launch_path_mesh_f64(R, grid, lds, st, a, spec);
    else
#endif
    {
        if (a.mesh_mode == 1) return launch_path_mode<T, true, 1>(R, grid, lds, st, a, spec);
        return launch_path_mode<T, true, 0>(R, grid, lds, st, a, spec);
    }
}

// the second launch of a mesh pass (PathArgs::mesh_mode 2): the exact instantiation — its waves add to radiance the first launch
// already stored, so they could not be rendered again, and its divisions are a small share of the frame's
template <class T>
void launch_path_resume(int R, dim3 grid, size_t lds, hipStream_t st, spira::PathArgs<T> a) {
    const bool ext = (a.rc.flags & (SPIRA_EXT_DIELECTRIC | SPIRA_EXT_SPECTRAL)) != 0;
    const dim3 blk(spira::kBlock);
    a.redo = nullptr; a.redo_only = 0;
    if (ext && a.scene.n_triangles) launch_lds(spira::k_path<T, 2, true, true, false, 2, true>, grid, blk, lds, st, a);
    else if (ext) launch_lds(spira::k_path<T, 2, true, true, false, 2, false>, grid, blk, lds, st, a);
    else if (R == 2 && a.scene.n_triangles) launch_lds(spira::k_path<T, 2, true, false, false, 2, true>, grid, blk, lds, st, a);
    else if (R == 2) launch_lds(spira::k_path<T, 2, true, false, false, 2, false>, grid, blk, lds, st, a);
    else launch_lds(spira::k_path<T, 1, true, false, false, 2>, grid, blk, lds, st, a);
}

// launch_path_resume<T> of whichever translation unit holds the mesh kernels of T
template <class T>
void launch_path_resume_entry(int R, dim3 grid, size_t lds, hipStream_t st, const spira::PathArgs<T> &a) {
#ifdef SPIRA_TU_MAIN
    if constexpr (sizeof(T) == 8) spira_tu::launch_path_resume_f64(R, grid, lds, st, a);
    else
#endif
        launch_path_resume<T>(R, grid, lds, st, a);
}

int profile_events(Ctx &c, size_t need) {
    while (c.ev_pool.size() < need) {
        hipEvent_t e;
        HIP_TRY(hipEventCreate(&e));
        c.ev_pool.push_back(e);
    }
    return 0;
}

// Every call that touches the context's workspaces first makes its stream wait for the previous call's end
// (which may have run on ANOTHER stream and not have been synchronised): the workspaces are shared per device.
int order_after_previous(Ctx &c, hipStream_t st) {
    if (c.have_done) HIP_TRY(hipStreamWaitEvent(st, c.ev_done, 0));
    return 0;
}
int mark_done(Ctx &c, hipStream_t st) {
    HIP_TRY(hipEventRecord(c.ev_done, st));
    c.have_done = true;
    return 0;
}

// The caller's output buffer is usually fresh from the allocator (`render` of either reference surface returns a new array): its pages do not exist yet,
// and the first touch of 12 000 of them inside the copy costs 2-3 ms.  The threads that will move the frame in ask the kernel for the pages (writable,
// contents untouched) while the GPU is still rendering; only pages that lie wholly inside the buffer.  A kernel that does not know the request says
// EINVAL and the copy faults the pages in as before.
void prefault_destination(char *p, size_t n) {
    const uintptr_t pg = 4096, a = ((uintptr_t)p + pg - 1) & ~(pg - 1), b = ((uintptr_t)p + n) & ~(pg - 1);
    if (b > a) (void)madvise((void *)a, b - a, 23 /* MADV_POPULATE_WRITE (Linux 5.14) */);
}

// A frame for a host-pointer caller (`render` of either reference surface returns a host array).  hipMemcpy into pageable memory runs at
// ~9 GB/s on this box (the runtime stages it on one thread): 5.7 ms for a 1080p Float64 frame, as long as rendering it.  Instead: device ->
// pinned staging in 8 MB chunks (one event each), and four host threads move the chunks on into the caller's memory as they arrive: 4.3 ms
// (most of what is left is the first touch of the caller's freshly allocated pages, which no copy strategy removes).
// Synchronous (the host-pointer entries are); small outputs take the plain copy.  Returns with the stream drained up to the copies.
int copy_out(Ctx &c, hipStream_t st, void *const dst[2], const void *const src[2], size_t bytes_each) {
    const size_t kChunk = (size_t)std::max<uint32_t>(1, env_u32("SPIRA_STAGE_CHUNK_MB", 8)) << 20, kMinStaged = 4u << 20, kMaxStaged = 512u << 20;      // (threads and chunk size: flat between 4 and 16 threads, 2 and 8 MB; frames beyond 512 MB take the plain copy rather than pin as much host memory)
    const int n_out = (dst[0] ? 1 : 0) + (dst[1] ? 1 : 0);
    if (!n_out) return 0;
    const size_t total = bytes_each * (size_t)n_out;
    bool staged = bytes_each >= kMinStaged && total <= kMaxStaged;
    if (staged && c.h_stage_cap < total) {
        if (c.h_stage) { (void)hipHostFree(c.h_stage); c.h_stage = nullptr; c.h_stage_cap = 0; }
        if (hipHostMalloc(&c.h_stage, total, hipHostMallocDefault) == hipSuccess) c.h_stage_cap = total;
        else { c.h_stage = nullptr; (void)hipGetLastError(); staged = false; }      // no pinned memory to be had: the plain copy still works
    }
    if (!staged) {
        for (int k = 0; k < 2; ++k) if (dst[k]) HIP_TRY(hipMemcpyAsync(dst[k], src[k], bytes_each, hipMemcpyDeviceToHost, st));
        return 0;
    }
    struct Piece { char *dst; size_t off, len; };
    std::vector<Piece> pieces;
    size_t off = 0;
    for (int k = 0; k < 2; ++k) {
        if (!dst[k]) continue;
        for (size_t o = 0; o < bytes_each; o += kChunk) {
            const size_t len = std::min(kChunk, bytes_each - o);
            HIP_TRY(hipMemcpyAsync((char *)c.h_stage + off, (const char *)src[k] + o, len, hipMemcpyDeviceToHost, st));
            if (c.stage_ev.size() <= pieces.size()) {
                hipEvent_t e;
                HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
                c.stage_ev.push_back(e);
            }
            HIP_TRY(hipEventRecord(c.stage_ev[pieces.size()], st));
            pieces.push_back({(char *)dst[k] + o, off, len});
            off += len;
        }
    }
    const int n_thr = (int)std::min<size_t>(std::max<uint32_t>(1, env_u32("SPIRA_STAGE_THREADS", 4)), pieces.size());
    std::vector<hipError_t> errs((size_t)n_thr, hipSuccess);
    const bool prefault = env_u32("SPIRA_PREFAULT", 1) != 0;
    auto mover = [&](int t) {
        (void)hipSetDevice(c.device);
        if (prefault) for (size_t i = (size_t)t; i < pieces.size(); i += (size_t)n_thr) prefault_destination(pieces[i].dst, pieces[i].len);
        for (size_t i = (size_t)t; i < pieces.size(); i += (size_t)n_thr) {
            const hipError_t e = hipEventSynchronize(c.stage_ev[i]);
            if (e != hipSuccess) { errs[(size_t)t] = e; return; }
            std::memcpy(pieces[i].dst, (const char *)c.h_stage + pieces[i].off, pieces[i].len);
        }
    };
    std::vector<std::thread> thr;
    for (int t = 1; t < n_thr; ++t) thr.emplace_back(mover, t);
    mover(0);
    for (auto &t : thr) t.join();
    for (hipError_t e : errs) if (e != hipSuccess) return fail(SPIRA_E_HIP, std::string("copy_out: ") + hipGetErrorString(e));
    return 0;
}

// The scene of a call: host arrays (uploaded into the context's store) or a handle (already resident).
template <class T>
int acquire_scene(Ctx &c, hipStream_t st, const spira_scene *h, const T *spheres5, const T *materials8, const T *triangles10,
                  const spira_params *p, spira::SceneGlobal<T> &g, bool reuse = false) {
    if (h) { scene_pointers<T>(h->store, g); return 0; }
    if (reuse) { scene_pointers<T>(c.scene, g); return 0; }      // (a later slab of the call that uploaded it)
    const uint32_t nt = triangles10 ? p->n_triangles : 0;
    if (int rc = scene_upload<T>(c.scene, st, c.have_done ? c.ev_done : nullptr, spheres5, materials8, triangles10, p->n_spheres, p->n_materials, nt)) return rc;
    scene_pointers<T>(c.scene, g);
    return 0;
}

// SPIRA_EXT_SPECTRAL: the SPD table of include/spira_spd.h in the render precision, resident per context.
template <class T>
int attach_spd(Ctx &c, hipStream_t st, const spira_params *p, spira::SceneGlobal<T> &g) {
    g.spd = nullptr;
    if (!(p->flags & SPIRA_EXT_SPECTRAL)) return 0;
    DevBuf &b = sizeof(T) == 4 ? c.spd32 : c.spd64;
    if (!b.p) {
        static_assert(SPIRA_SPD_N == spira::kSpdN && SPIRA_SPD_ROWS == spira::kSpdRows, "SPD table shape");
        T host[SPIRA_SPD_ROWS * SPIRA_SPD_N];
        for (int r = 0; r < SPIRA_SPD_ROWS; ++r)
            for (int i = 0; i < SPIRA_SPD_N; ++i) host[r * SPIRA_SPD_N + i] = (T)spira_spd_table[r][i];
        if (int rc = b.ensure(sizeof host)) return rc;
        HIP_TRY(hipMemcpy(b.p, host, sizeof host, hipMemcpyHostToDevice));
    }
    (void)st;
    g.spd = (const T *)b.p;
    return 0;
}

template <class T>
int check_handle(const spira_scene *h) {
    if (!h || h->magic != kSceneMagic) return fail(SPIRA_E_INVALID, "scene handle is NULL or was destroyed");
    if (h->prec != (int)sizeof(T)) return fail(SPIRA_E_INVALID, "scene handle was created in the other precision");
    if (h->device != tl_device) return fail(SPIRA_E_INVALID, "scene handle belongs to another device (spira_set_device)");
    return 0;
}

// ---- workspaces of the persistent organisation, sized and checked in ONE place.
// PathPlan says what the passes of a call may touch (derived from the launch geometry); ensure_path_plan() sizes every buffer a launch can be handed;
// verify_path_args() — called right before every k_path launch — checks each pointer of PathArgs against the capacity of the buffer it points into for
// the grid about to be launched, and refuses the launch (SPIRA_E_LIMIT) instead of letting a kernel write past a buffer nobody sized.  (Round 3's
// fuzz found exactly that: a depth-1 mesh render writing parked rays' hits into queues only `max_depth > 1` used to size.)
struct PathPlan {
    uint64_t waves = 0;        // NW of the largest pass (G_max * waves per workgroup)
    uint64_t packets = 0;      // NW * cap of the largest pass: entries of every per-packet array (worst case: every path queued / parked once)
    uint64_t batch = 0;        // paths of the largest pass (entries of L)
    bool queues = false;       // hit queues: max_depth > 1, or a mesh scene (a parked camera ray's hit comes back from its session as a packet)
    bool mesh = false;         // mesh lists (deferred traversal)
    bool two_pass = false;     // per-wave parked counts handed from the parking launch to the fat-wave launch
    bool spec = false;         // per-wave redo flags of the speculative-division launch
};
template <class T>
int ensure_path_plan(Ctx &c, const PathPlan &pl) {
    using P4 = spira::Pack4<T>;
    using P2 = spira::Pack2<T>;
    if (pl.queues)
        for (int i = 0; i < 2; ++i) {
            if (int rc = c.qA[i].ensure(pl.packets * sizeof(P4))) return rc;
            if (int rc = c.qB[i].ensure(pl.packets * sizeof(P4))) return rc;
            if (int rc = c.qC[i].ensure(pl.packets * sizeof(P2))) return rc;
            if (sizeof(T) == 4) { if (int rc = c.qR[i].ensure(pl.packets * sizeof(uint32_t))) return rc; }
            if (!pl.mesh) { if (int rc = c.qK[i].ensure(pl.packets * sizeof(uint2))) return rc; }      // carried RNG keys (sphere scenes; the extension launches leave them unused)
        }
    if (pl.mesh) { if (int rc = c.mesh_list.ensure(3 * pl.packets * sizeof(P4))) return rc; }
    if (pl.two_pass) { if (int rc = c.mesh_count.ensure(pl.waves * sizeof(uint32_t))) return rc; }
    if (pl.spec) { if (int rc = c.redo.ensure(pl.waves * sizeof(uint32_t))) return rc; }
    if (int rc = c.blkstats.ensure(pl.waves * 4 * sizeof(uint32_t))) return rc;
    if (int rc = c.L.ensure(pl.batch * sizeof(spira::Pack3<T>))) return rc;
    return c.stats.ensure(sizeof(spira::Stats));
}
template <class T>
int verify_path_args(const Ctx &c, const spira::PathArgs<T> &a, uint32_t first_launch_blocks) {
    using P4 = spira::Pack4<T>;
    using P2 = spira::Pack2<T>;
    const uint64_t nw = (uint64_t)first_launch_blocks * (spira::kBlock / 64), n = nw * a.cap;      // (a fat wave of the second launch owns the regions of the waves it takes over: the same n)
    auto covers = [](const DevBuf &b, const void *ptr, uint64_t bytes) { return ptr == b.p && b.p != nullptr && b.cap >= bytes; };
    const char *bad = nullptr;
    const bool mesh = a.scene.n_bvh_tris != 0;
    if ((uint64_t)a.n_first > n) bad = "the pass does not fit its queue regions";
    if (a.rc.max_depth > 1 || mesh)
        for (int i = 0; i < 2 && !bad; ++i) {
            if (!covers(c.qA[i], a.q[i].A, n * sizeof(P4)) || !covers(c.qB[i], a.q[i].B, n * sizeof(P4)) || !covers(c.qC[i], a.q[i].C, n * sizeof(P2))) bad = "hit queues";
            else if (sizeof(T) == 4 && !covers(c.qR[i], a.qref[i], n * sizeof(uint32_t))) bad = "hit reference arrays";
            else if (!mesh && !covers(c.qK[i], a.qkey[i], n * sizeof(uint2))) bad = "carried RNG keys";
        }
    if (!bad && a.mesh_list && !covers(c.mesh_list, a.mesh_list, 3 * n * sizeof(P4))) bad = "mesh lists";
    if (!bad && a.mesh_mode != 0 && (!a.mesh_list || !covers(c.mesh_count, a.mesh_count, nw * sizeof(uint32_t)) || a.resume_k == 0 || a.resume_k > 16 || nw % a.resume_k != 0 || a.resume_nw != nw))
        bad = "parked-ray counts of a two-launch mesh pass";
    if (!bad && a.redo && !covers(c.redo, a.redo, nw * sizeof(uint32_t))) bad = "redo flags";
    if (!bad && !covers(c.blkstats, a.blk_stats, nw * 4 * sizeof(uint32_t))) bad = "per-wave statistics";
    if (!bad && !covers(c.L, a.L, (uint64_t)a.n_first * sizeof(spira::Pack3<T>))) bad = "per-path radiance";
    if (!bad && !covers(c.stats, a.stats, sizeof(spira::Stats))) bad = "counters";
    if (bad) return fail(SPIRA_E_LIMIT, std::string("internal: a workspace is smaller than the launch needs (") + bad + ")");
    return 0;
}

template <class T>
int render_impl(const spira_scene *h, const T *spheres5, const T *materials8, const T *triangles10, const T *camera12, const spira_params *p,
                T *out_hdr, T *out_img, bool out_on_device, void *user_stream,
                bool progressive = false, uint32_t sample0 = 0, uint32_t *rng_states = nullptr, const SlabCtl *slab = nullptr) {
    // progressive: out_hdr is the caller's running SUM (in/out), samples [sample0, sample0 + spp) are added to it
    uint32_t rows = 0;
    const bool cont = slab && slab->index > 0;      // a later slab of a host-output frame: continues slab 0's call (SlabCtl)
    if (!p) return fail(SPIRA_E_INVALID, "params is NULL");
    tl_lds_optin = hipSuccess;           // (a flag an earlier call of this thread left behind by returning early must not fail this one)
    Lap lap("render");
    if (h) { if (int rc = check_handle<T>(h)) return rc; }
    else if (!cont) { if (int rc = validate_scene<T>(spheres5, materials8, triangles10, p->n_spheres, p->n_materials, triangles10 ? p->n_triangles : 0)) return rc; }
    if (int rc = validate_params(camera12, p, h ? h->store.nt : (triangles10 ? p->n_triangles : 0), &rows)) return rc;
    lap("validate");
    if (!out_hdr && !out_img) return fail(SPIRA_E_INVALID, "both outputs are NULL");
    if (progressive && (uint64_t)sample0 + p->spp > SPIRA_MAX_SPP) return fail(SPIRA_E_LIMIT, "sample0 + spp exceeds 2^24");
    if (progressive && (p->flags & SPIRA_SEM_MASK) == SPIRA_SEM_HYBRID) return fail(SPIRA_E_UNSUPPORTED, "SPIRA_SEM_HYBRID has no accumulate entry (its image is a mean of tone-mapped samples)");
    if (progressive && (p->flags & SPIRA_SEM_MASK) == SPIRA_SEM_METAL && sample0 > 0 && !rng_states)
        return fail(SPIRA_E_INVALID, "SPIRA_SEM_METAL with sample0 > 0 needs rng_states (the LCG states the previous call left); "
                                     "without them every call would replay the samples of the first");
    Ctx *cp = nullptr;
    if (int rc = get_ctx(&cp)) return rc;
    Ctx &c = *cp;
    std::lock_guard<std::recursive_mutex> lock(c.mu);
    hipStream_t st = out_on_device ? (hipStream_t)user_stream : c.stream;
    if (int rc = order_after_previous(c, st)) return rc;
    lap("context");

    const uint32_t W = p->width;
    const uint64_t tile_pixels = (uint64_t)rows * W;
    // default pass size: 160 Mi rays (a 1080p x 64 spp frame is one pass); ~16 GB (f32) / 31 GB (f64) of the 288 GB
    uint32_t target = p->batch_rays ? p->batch_rays : env_u32("SPIRA_BATCH_RAYS", 160u << 20);
    uint64_t slots64 = std::max<uint64_t>(1, target / tile_pixels);
    slots64 = std::min<uint64_t>(slots64, p->spp);
    slots64 = (p->spp + (p->spp + slots64 - 1) / slots64 - 1) / ((p->spp + slots64 - 1) / slots64);      // equal passes: spp 256 at 80 slots -> 4 x 64, not 3 x 80 + 16
    if (slots64 * tile_pixels > 0x7FFFFFFFull) return fail(SPIRA_E_LIMIT, "tile too large: rows*width must be < 2^31");
    const uint32_t slots = (uint32_t)slots64;
    const uint64_t batch = slots64 * tile_pixels;
    const uint32_t sem = p->flags & SPIRA_SEM_MASK;
    // the secondary variants run one lane per path / per pixel: no queues, no bounce kernels
    const bool mega = (p->flags & SPIRA_KERNEL_MASK) == SPIRA_KERNEL_MEGA || sem != SPIRA_SEM_A;
    const bool metal_wavefront = sem == SPIRA_SEM_METAL && (p->flags & SPIRA_KERNEL_MASK) == SPIRA_KERNEL_WAVEFRONT;
    uint64_t metal_launches = 0;
    const bool per_bounce = !mega && (p->flags & SPIRA_KERNEL_MASK) == SPIRA_KERNEL_BOUNCE;     // round-1 organisation: one launch per bounce
    const bool persistent = !mega && !per_bounce;                                                // k_path: one launch per pass
    const bool profile = ((p->flags & SPIRA_FLAG_PROFILE) != 0 && per_bounce) || persistent;     // k_path launches are always bracketed (2 events per pass)
    int R = (int)env_u32("SPIRA_R", 2);
    if (R != 1 && R != 2) R = 2;
    if (p->flags & (SPIRA_EXT_DIELECTRIC | SPIRA_EXT_SPECTRAL)) R = 2;      // the extension instantiations exist for R = 2 only

    // ---- launch geometry: NW = 4*G autonomous waves per bounce kernel, each owning `cap` rays of both queues
    // workgroups per CU: the persistent kernel runs a whole pass per launch, so its launch tail is one workgroup's share of the
    // pass: 32 per CU (8 rounds of resident workgroups) measured best on S1 (16: -3.5 %, 64: -1 %, 128: -5 %; S3 likes 64-128, +1.7 %)
    // Mesh scenes want fewer, fatter waves: a wave's round ends with the dense traversal of the rays it parked at the mesh's box, and a
    // traversal batch costs its slowest ray's chain of dependent node fetches whether it holds 64 rays or 10 (config 5, 81 920 triangles:
    // f32 32 per CU 7.54 ms, 16: 6.78, 8: 7.29, 4: 7.04; f64 32: 11.67, 8: 10.29, 4: 9.68; re-measured with the round's final kernels: f32 16: 6.83, 8: 7.02, 32: 7.39,
    // f64 4: 9.54, 8: 10.13, 16: 10.49 — and counts that are not powers of two lose 10-40 %: the grid no longer divides evenly over 8 XCDs x 32 CUs).
    const uint32_t nt_scene = h ? h->store.nt : (triangles10 ? p->n_triangles : 0);
    const bool mesh_two_pass = persistent && nt_scene > SPIRA_LDS_TRIANGLES && p->max_depth <= 128 && env_u32("SPIRA_DEFER_MESH", 1) && env_u32("SPIRA_MESH_TWO_PASS", 1);
    const uint32_t blocks_per_cu = !persistent ? 16 : ((nt_scene > SPIRA_LDS_TRIANGLES && !mesh_two_pass) ? 4 : 32);
    const uint32_t max_blocks = (uint32_t)c.num_cus * env_u32("SPIRA_BLOCKS_PER_CU", blocks_per_cu);
    const uint32_t wpb = spira::kBlock / 64;
    const uint32_t sub = 64 * R;                                   // rays per wave sub-chunk
    auto geometry = [&](uint64_t n_first, uint32_t &G, uint32_t &cap) {
        const uint64_t n_sub = (n_first + sub - 1) / sub;
        G = (uint32_t)std::min<uint64_t>((n_sub + wpb - 1) / wpb, max_blocks);
        const uint64_t nw = (uint64_t)G * wpb;
        cap = (uint32_t)(((n_sub + nw - 1) / nw) * sub);
    };
    uint32_t G_max = 0, cap_max = 0;
    geometry(batch, G_max, cap_max);
    const uint64_t q_rays = (uint64_t)cap_max * G_max * wpb;
    if (q_rays > 0xFFFFFFFFull) return fail(SPIRA_E_LIMIT, "pass too large");

    // ---- workspaces (cached per device, grown on demand; sized for 288 GB HBM: no chunking of a pass)
    using P4 = spira::Pack4<T>;
    using P2 = spira::Pack2<T>;
    const bool mesh_scene = nt_scene > SPIRA_LDS_TRIANGLES;
    const bool defer_mesh = persistent && mesh_scene && p->max_depth <= 128 && env_u32("SPIRA_DEFER_MESH", 1) != 0;
    // speculative division (spira_device.h, SpecDiv): decided here once, because it needs a workspace (the per-wave redo flags) — see the launch below
    PathPlan plan;
    plan.waves = (uint64_t)G_max * wpb; plan.packets = q_rays; plan.batch = batch;
    // (max_depth == 1 needs no queue — except on a mesh scene of the persistent organisation: a parked camera ray's hit comes back from its
    //  traversal session as a packet.)
    plan.queues = !mega && (p->max_depth > 1 || (persistent && mesh_scene));
    plan.mesh = defer_mesh; plan.two_pass = defer_mesh && mesh_two_pass; plan.spec = persistent && env_u32("SPIRA_SPEC_DIV", 1) != 0;
    if (persistent) { if (int rc = ensure_path_plan<T>(c, plan)) return rc; }
    else {
        if (plan.queues)
            for (int i = 0; i < 2; ++i) {
                if (int rc = c.qA[i].ensure(q_rays * sizeof(P4))) return rc;
                if (int rc = c.qB[i].ensure(q_rays * sizeof(P4))) return rc;
                if (int rc = c.qC[i].ensure(q_rays * sizeof(P2))) return rc;
            }
        if (sem != SPIRA_SEM_HYBRID) { if (int rc = c.L.ensure(batch * sizeof(spira::Pack3<T>))) return rc; }
        // per-wave survivor counts exist in the per-bounce organisation only; statistics: one row per wave per launch
        if (per_bounce) { if (int rc = c.counts.ensure((size_t)(p->max_depth + 2) * G_max * wpb * sizeof(uint32_t))) return rc; }
        if (int rc = c.blkstats.ensure((size_t)(per_bounce ? p->max_depth + 1 : 1) * G_max * wpb * 4 * sizeof(uint32_t))) return rc;
        if (int rc = c.stats.ensure(sizeof(spira::Stats))) return rc;
    }
    if (int rc = c.accum.ensure(tile_pixels * sizeof(P4))) return rc;

    lap("workspaces");
    spira::BounceArgs<T> a{};
    if (int rc = acquire_scene<T>(c, st, h, spheres5, materials8, triangles10, p, a.scene, cont)) return rc;
    if (int rc = attach_spd<T>(c, st, p, a.scene)) return rc;
    lap("scene");
    const bool scene_moderate = h ? h->store.moderate : c.scene.moderate;
    fill_const<T>(a.rc, camera12, p, rows, slots);
    if (!fastdiv_selfcheck(a.rc.tile_pixels, (uint32_t)batch) || !fastdiv_selfcheck(a.rc.width, a.rc.tile_pixels) ||
        !fastdiv_selfcheck(a.rc.stripe_h ? a.rc.stripe_h : 1, rows))
        return fail(SPIRA_E_LIMIT, "internal: fast division self-check failed");
    a.L = (spira::Pack3<T> *)c.L.p;
    a.stats = (spira::Stats *)c.stats.p;

    const size_t lds = spira::scene_lds_bytes<T>(a.scene.n_spheres, a.scene.n_materials, a.scene.n_triangles);
    const uint32_t n_pass = (p->spp + slots - 1) / slots;

    T *d_hdr = out_hdr, *d_img = out_img;
    if (!out_on_device) {
        size_t plane3 = 3 * tile_pixels * sizeof(T);
        if (int rc = c.out_tmp.ensure(2 * plane3)) return rc;
        d_hdr = out_hdr ? (T *)c.out_tmp.p : nullptr;
        d_img = out_img ? (T *)((char *)c.out_tmp.p + plane3) : nullptr;
    }

    size_t n_prof = 0;
    if (profile) {
        n_prof = (size_t)n_pass * (persistent ? 1 : std::max<uint32_t>(p->max_depth, 1)) * 2;
        if (int rc = profile_events(c, (cont ? c.ev_used : 0) + n_prof)) return rc;
    }
    if (!cont) {                         // (a later slab adds its brackets and device counters to slab 0's)
        c.ev_used = 0;
        c.ev_mid_used = 0;
        HIP_TRY(hipMemsetAsync(c.stats.p, 0, sizeof(spira::Stats), st));
        HIP_TRY(hipEventRecord(c.ev_start, st));
    }
    uint64_t launches = 0;

    // progressive accumulation: the caller's running sums (and, METAL, LCG states) seed the accumulator
    a.rc.sample0 = progressive ? sample0 : 0;
    uint32_t *d_rng = nullptr;
    if (progressive) {
        const uint32_t lblocks = std::min<uint32_t>((uint32_t)((tile_pixels + spira::kBlock - 1) / spira::kBlock), max_blocks);
        if (!out_on_device) HIP_TRY(hipMemcpyAsync(d_hdr, out_hdr, 3 * tile_pixels * sizeof(T), hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL((spira::k_load_accum<T>), dim3(lblocks), dim3(spira::kBlock), 0, st, (P4 *)c.accum.p, (const T *)d_hdr, (uint32_t)tile_pixels);
        ++launches;
        if (rng_states) {
            d_rng = rng_states;
            if (!out_on_device) {
                if (int rc = c.rng.ensure(tile_pixels * sizeof(uint32_t))) return rc;
                d_rng = (uint32_t *)c.rng.p;
                if (sample0 > 0) HIP_TRY(hipMemcpyAsync(d_rng, rng_states, tile_pixels * sizeof(uint32_t), hipMemcpyHostToDevice, st));
            }
        }
    }

    if (p->max_depth == 0) {
        if (!progressive) HIP_TRY(hipMemsetAsync(c.accum.p, 0, tile_pixels * sizeof(P4), st));   // depth <= 0 -> Vec3(0,0,0), :330
    } else if (sem == SPIRA_SEM_HYBRID) {
        // render_hybrid_gpu as written (spira_device.h, k_hybrid): the whole image in lock step, max_depth + 1 launches per sample, all on this stream
        const uint64_t P = tile_pixels;
        if (P > 0xFFFFFFFFull / 2) return fail(SPIRA_E_LIMIT, "image too large for SPIRA_SEM_HYBRID");
        if (int rc = c.hyb_state.ensure(12 * P * sizeof(T))) return rc;
        if (int rc = c.hyb_mat.ensure(P * sizeof(uint32_t))) return rc;
        if (int rc = c.rng.ensure(P * sizeof(uint32_t))) return rc;
        const size_t n_flags = (size_t)p->spp * (p->max_depth + 1);
        if (int rc = c.hyb_flags.ensure(n_flags * sizeof(uint32_t))) return rc;
        HIP_TRY(hipMemsetAsync(c.hyb_flags.p, 0, n_flags * sizeof(uint32_t), st));
        HIP_TRY(hipMemsetAsync(c.accum.p, 0, P * sizeof(P4), st));
        HIP_TRY(hipMemsetAsync(c.hyb_mat.p, 0, P * sizeof(uint32_t), st));
        HIP_TRY(hipMemsetAsync(c.hyb_state.p, 0, 12 * P * sizeof(T), st));
        const uint32_t hblocks = std::min<uint32_t>((uint32_t)((P + spira::kBlock - 1) / spira::kBlock), max_blocks);
        hipLaunchKernelGGL(spira::k_hybrid_init, dim3(hblocks), dim3(spira::kBlock), 0, st, (uint32_t *)c.rng.p, (uint32_t)P, a.rc.sA, a.rc.sB);
        spira::HybridArgs<T> ha{};
        ha.scene = a.scene; ha.rc = a.rc; ha.state = (T *)c.hyb_state.p; ha.mat = (uint32_t *)c.hyb_mat.p; ha.rng = (uint32_t *)c.rng.p;
        ha.accum = (P4 *)c.accum.p; ha.flags = (uint32_t *)c.hyb_flags.p; ha.stats = (spira::Stats *)c.stats.p;
        for (uint32_t smp = 1; smp <= p->spp; ++smp)
            for (uint32_t ph = 0; ph <= p->max_depth; ++ph) {
                ha.sample = smp; ha.phase = ph;
                launch_lds(spira::k_hybrid<T>, dim3(hblocks), dim3(spira::kBlock), lds, st, ha);
            }
        if (int rc = lds_optin_failed()) return rc;
        launches += 1 + (uint64_t)p->spp * (p->max_depth + 1);
    } else if (sem == SPIRA_SEM_METAL && metal_wavefront) {
        // the .metal estimator in wavefront form: every wave owns a block of pixels and walks sample after sample on it
        spira::MetalArgs<T> ma{};
        ma.scene = a.scene; ma.rc = a.rc;
        const uint32_t g_res = (uint32_t)c.num_cus * (sizeof(T) == 8 ? SPIRA_WAVES_F64 : SPIRA_WAVES_F32);     // one resident round of workgroups
        const uint64_t nw0 = (uint64_t)g_res * wpb;
        ma.ppw = (uint32_t)((((tile_pixels + nw0 - 1) / nw0) + 63) / 64 * 64);
        const uint32_t Gm = (uint32_t)((tile_pixels + (uint64_t)ma.ppw * wpb - 1) / ((uint64_t)ma.ppw * wpb));
        const uint64_t slots_q = (uint64_t)Gm * wpb * ma.ppw;
        for (int i = 0; i < 2; ++i) {
            if (int rc = c.qA[i].ensure(slots_q * sizeof(P4))) return rc;
            if (int rc = c.qB[i].ensure(slots_q * sizeof(P4))) return rc;
            if (int rc = c.qC[i].ensure(slots_q * sizeof(P2))) return rc;
            if (int rc = c.qX[i].ensure(slots_q * sizeof(uint2))) return rc;
            ma.q[i] = {(P4 *)c.qA[i].p, (P4 *)c.qB[i].p, (P2 *)c.qC[i].p};
            ma.qx[i] = (uint2 *)c.qX[i].p;
        }
        if (!d_rng) {
            if (int rc = c.rng.ensure(tile_pixels * sizeof(uint32_t))) return rc;
            d_rng = (uint32_t *)c.rng.p;
        }
        if (int rc = c.blkstats.ensure((size_t)Gm * wpb * 4 * sizeof(uint32_t))) return rc;
        ma.L = (spira::Pack3<T> *)c.L.p; ma.accum = (P4 *)c.accum.p; ma.rng_states = d_rng; ma.blk_stats = (uint32_t *)c.blkstats.p;
        ma.resume = progressive ? (sample0 > 0 ? 3 : 1) : 0;          // bit 0: continue the sums, bit 1: continue the LCG states
        if (int rc = profile_events(c, 2)) return rc;
        HIP_TRY(hipEventRecord(c.ev_pool[c.ev_used++], st));
        int spec = (int)env_u32("SPIRA_SPEC_DIV", 1);      // speculative division as in k_path (fresh renders only: a progressive call updates sums and states in place)
        if (spec == 1 && !(scene_moderate && spira::camera_scale_moderate<T>(camera12))) spec = 0;
        if (spec == 3) spec = 1;
        if (ma.resume || R != 2) spec = 0;
        ma.stats = (spira::Stats *)c.stats.p; ma.redo = nullptr; ma.redo_only = 0;
        if (spec) {
            if (int rc = c.redo.ensure((size_t)Gm * wpb * sizeof(uint32_t))) return rc;
            ma.redo = (uint32_t *)c.redo.p;
            ma.redo_only = spec == 2 ? 2 : 0;
            launch_lds(spira::k_path_metal<T, 2, true>, dim3(Gm), dim3(spira::kBlock), lds, st, ma);
            ma.redo_only = 1;
            ++launches;
        }
        if (R == 2) launch_lds(spira::k_path_metal<T, 2, false>, dim3(Gm), dim3(spira::kBlock), lds, st, ma);
        else launch_lds(spira::k_path_metal<T, 1, false>, dim3(Gm), dim3(spira::kBlock), lds, st, ma);
        if (int rc = lds_optin_failed()) return rc;
        HIP_TRY(hipEventRecord(c.ev_pool[c.ev_used++], st));
        hipLaunchKernelGGL(spira::k_fold_stats, dim3(1), dim3(64), 0, st, (const uint32_t *)c.blkstats.p, Gm * wpb, (spira::Stats *)c.stats.p);
        launches += 2;
        metal_launches = 1;
    } else if (sem == SPIRA_SEM_METAL) {
        // one launch: every lane owns a pixel and walks its spp samples (the LCG state runs through them)
        uint32_t blocks = std::min<uint32_t>((uint32_t)((tile_pixels + spira::kBlock - 1) / spira::kBlock), max_blocks);
        a.pass = 0; a.n_first = (uint32_t)tile_pixels;
        const int resume = progressive ? (sample0 > 0 ? 3 : 1) : 0;       // bit 0: continue the sums, bit 1: continue the LCG states
        // speculative division as in k_path (SPIRA_SPEC_DIV): fresh renders of scenes of ordinary scale; the exact launch behind renders reported waves again
        int spec = (int)env_u32("SPIRA_SPEC_DIV", 1);
        if (spec == 1 && !(scene_moderate && spira::camera_scale_moderate<T>(camera12))) spec = 0;
        if (spec == 3) spec = 1;
        if (resume) spec = 0;
        if (spec) {
            if (int rc = c.redo.ensure((size_t)blocks * wpb * sizeof(uint32_t))) return rc;
            uint32_t *redo = (uint32_t *)c.redo.p;
            launch_lds(spira::k_variant_metal<T, true>, dim3(blocks), dim3(spira::kBlock), lds, st, a, (P4 *)c.accum.p, d_rng, resume, redo, spec == 2 ? 2 : 0);
            launch_lds(spira::k_variant_metal<T, false>, dim3(blocks), dim3(spira::kBlock), lds, st, a, (P4 *)c.accum.p, d_rng, resume, redo, 1);
            launches += 2;
        } else {
            launch_lds(spira::k_variant_metal<T, false>, dim3(blocks), dim3(spira::kBlock), lds, st, a, (P4 *)c.accum.p, d_rng, resume, (uint32_t *)nullptr, 0);
            ++launches;
        }
    } else {
        for (uint32_t pass = 0; pass < n_pass; ++pass) {
            const uint32_t k_eff = std::min(slots, p->spp - pass * slots);
            const uint32_t n_first = (uint32_t)((uint64_t)k_eff * tile_pixels);
            a.pass = pass;
            a.n_first = n_first;
            uint32_t G = 0, stat_rows = 0;
            if (sem == SPIRA_SEM_CPU) {
                uint32_t blocks = std::min<uint32_t>((n_first + spira::kBlock - 1) / spira::kBlock, max_blocks);
                int spec = (int)env_u32("SPIRA_SPEC_DIV", 1);      // speculative division as in k_path / k_variant_metal
                if (spec == 1 && !(scene_moderate && spira::camera_scale_moderate<T>(camera12))) spec = 0;
                if (spec == 3) spec = 1;
                if (spec) {
                    if (int rc = c.redo.ensure((size_t)max_blocks * wpb * sizeof(uint32_t))) return rc;
                    uint32_t *redo = (uint32_t *)c.redo.p;
                    launch_lds(spira::k_variant_cpu<T, true>, dim3(blocks), dim3(spira::kBlock), lds, st, a, redo, spec == 2 ? 2 : 0);
                    launch_lds(spira::k_variant_cpu<T, false>, dim3(blocks), dim3(spira::kBlock), lds, st, a, redo, 1);
                    launches += 2;
                } else {
                    launch_lds(spira::k_variant_cpu<T, false>, dim3(blocks), dim3(spira::kBlock), lds, st, a, (uint32_t *)nullptr, 0);
                    ++launches;
                }
            } else if (mega) {
                uint32_t blocks = std::min<uint32_t>((n_first + spira::kBlock - 1) / spira::kBlock, max_blocks);
                const bool ext = (p->flags & (SPIRA_EXT_DIELECTRIC | SPIRA_EXT_SPECTRAL)) != 0;
                if (a.scene.n_bvh_tris) { if (ext) launch_lds(spira::k_mega<T, true, true>, dim3(blocks), dim3(spira::kBlock), lds, st, a); else launch_lds(spira::k_mega<T, true, false>, dim3(blocks), dim3(spira::kBlock), lds, st, a); }
                else { if (ext) launch_lds(spira::k_mega<T, false, true>, dim3(blocks), dim3(spira::kBlock), lds, st, a); else launch_lds(spira::k_mega<T, false, false>, dim3(blocks), dim3(spira::kBlock), lds, st, a); }
                ++launches;
            } else if (persistent) {
                // one launch: every wave walks all max_depth stages on its own region of the hit queues
                spira::PathArgs<T> pa{};
                pa.scene = a.scene; pa.rc = a.rc; pa.L = a.L; pa.pass = pass; pa.n_first = n_first;
                // dense continuation threshold (same device, S1 1080p spp 64 depth 8, Msamples/s): f64 100 %: 20 218, 90: 20 563, 80: 20 953,
                // 70: 20 963, 60: 20 052; f32 90: 30 222, 80: 30 141, 70: 29 567, 60: 28 354 — a packet costs twice the bytes in Float64, so it
                // pays to keep a little more in registers there.  On the closed box S3 any threshold > 0 gives the full +22 % (f64).
                // Round 4 (packets carry the RNG key words: a queued hit costs more), S1 ms per frame, two rounds on one box: f64 80: 5.302 / 5.277, 75: 5.245 / 5.229,
                // 70: 5.221 / 5.240, 65: 5.272 / 5.299; f32 80: 3.382 / 3.414, 75: 3.341 / 3.354, 70: 3.340 / 3.351, 65: 3.374 / 3.351; configs[4] the same at 70 and 80.
                pa.dense_pct = std::min<uint32_t>(env_u32("SPIRA_DENSE_PCT", 70), 100);      // (Float32 re-measured on the no-SLP build of round 3, S1: 90: 37 900, 85: 38 500, 80: 38 500, 75: 38 500, 70: 37 500)
                // BVH scenes: the wave-owned lists of rays waiting for their dense traversal batch (3 packets per entry, `cap` entries per wave)
                pa.mesh_list = plan.mesh ? (P4 *)c.mesh_list.p : nullptr;
                geometry(n_first, G, pa.cap);
                stat_rows = G * wpb;
                for (int i = 0; i < 2; ++i) {
                    pa.q[i] = {(P4 *)c.qA[i].p, (P4 *)c.qB[i].p, (P2 *)c.qC[i].p};
                    pa.qref[i] = (uint32_t *)c.qR[i].p;
                    pa.qkey[i] = (uint2 *)c.qK[i].p;
                }
                pa.blk_stats = (uint32_t *)c.blkstats.p;
                pa.stats = (spira::Stats *)c.stats.p;
                // speculative division (spira_device.h, SpecDiv): +7 % on S1 while (almost) no wave has to be rendered again, which is what a scene
                // and camera of ordinary magnitudes give; a scene scaled to 1e-30 would have every wave rendered twice, so it is not tried there
                // SPIRA_SPEC_DIV: 0 off, 1 default, 2 report every wave (the whole pass is rendered twice), 3 on even where the predictor says no
                int spec = (int)env_u32("SPIRA_SPEC_DIV", 1);
                if (spec == 1 && !(scene_moderate && spira::camera_scale_moderate<T>(camera12))) spec = 0;
                if (spec == 3) spec = 1;
                if (spec) pa.redo = (uint32_t *)c.redo.p;      // (sized by the plan whenever SPIRA_SPEC_DIV != 0)
                pa.mesh_mode = 0; pa.mesh_count = nullptr; pa.resume_k = 1; pa.resume_nw = 0;
                pa.mesh_min_batch = std::max<uint32_t>(1, env_u32("SPIRA_MESH_MIN_BATCH", 128));
                pa.refill_free = std::min<uint32_t>(64, std::max<uint32_t>(1, env_u32("SPIRA_MESH_REFILL", 16)));
                size_t lds_a = lds + (size_t)wpb * sub * sizeof(P4) + 128;             // + one work list per wave + the camera
                // ... + one packet per sphere: what a sphere test of a CAMERA ray does not depend on the ray for (closest_hit_local, CAM) — where the block has the room
                pa.cam_consts = (env_u32("SPIRA_CAM_CONSTS", 1) && a.scene.n_spheres && lds_a + (size_t)a.scene.n_spheres * sizeof(P4) <= (size_t)160 * 1024) ? 1u : 0u;
                if (pa.cam_consts) lds_a += (size_t)a.scene.n_spheres * sizeof(P4);
                if (plan.two_pass) {
                    pa.mesh_mode = 1; pa.mesh_count = (uint32_t *)c.mesh_count.p;
                    // the fat waves of the second launch: about 16 per CU (4 per SIMD), each taking over k <= 16 first-launch waves; k divides their number
                    const uint32_t nw = G * wpb, fat = std::max<uint32_t>(1, (uint32_t)c.num_cus * env_u32("SPIRA_MESH_FAT_WAVES_PER_CU", 16));
                    uint32_t k = 16;
                    while (k > 1 && (nw % k != 0 || nw / k < fat)) k >>= 1;
                    pa.resume_k = k; pa.resume_nw = nw;
                }
                if (int rc = verify_path_args<T>(c, pa, G)) return rc;      // every pointer against the capacity of its buffer, for THIS grid
                HIP_TRY(hipEventRecord(c.ev_pool[c.ev_used++], st));
                if (int rc = launch_path<T>(R, dim3(G), lds_a, st, pa, spec)) return rc;
                if (int rc = lds_optin_failed()) return rc;      // (a kernel that was refused its LDS did not run: nothing that consumes its output is enqueued)
                launches += (spec && R == 2) ? 2 : 1;      // the speculative launch and its exact follow-up
                if (pa.mesh_mode == 1) {           // second launch: nw / k fat waves
                    if (c.ev_mid.size() <= c.ev_mid_used) {
                        hipEvent_t e;
                        HIP_TRY(hipEventCreate(&e));
                        c.ev_mid.push_back(e); c.ev_mid_end.push_back(0);
                    }
                    HIP_TRY(hipEventRecord(c.ev_mid[c.ev_mid_used], st));
                    c.ev_mid_end[c.ev_mid_used++] = c.ev_used;      // (the closing event of this pass's bracket is recorded next)
                    spira::PathArgs<T> pb = pa;
                    pb.mesh_mode = 2; pb.n_first = 0;
                    const uint32_t nwb = pa.resume_nw / pa.resume_k;
                    launch_path_resume_entry<T>(R, dim3((nwb + wpb - 1) / wpb), lds_a, st, pb);
                    if (int rc = lds_optin_failed()) return rc;
                    ++launches;
                }
                HIP_TRY(hipEventRecord(c.ev_pool[c.ev_used++], st));
            } else {
                geometry(n_first, G, a.cap);
                stat_rows = p->max_depth * G * wpb;
                const size_t nw = (size_t)G * wpb;
                for (uint32_t b = 0; b < p->max_depth; ++b) {
                    a.bounce = b;
                    int qi = b & 1;      // bounce b writes queue qi, reads queue qi^1
                    a.qout = {(P4 *)c.qA[qi].p, (P4 *)c.qB[qi].p, (P2 *)c.qC[qi].p};
                    a.qin = {(P4 *)c.qA[qi ^ 1].p, (P4 *)c.qB[qi ^ 1].p, (P2 *)c.qC[qi ^ 1].p};
                    a.cnt_in = (const uint32_t *)c.counts.p + (size_t)b * nw;
                    a.cnt_out = (uint32_t *)c.counts.p + (size_t)(b + 1) * nw;
                    a.blk_stats = (uint32_t *)c.blkstats.p + (size_t)b * nw * 4;
                    if (profile) HIP_TRY(hipEventRecord(c.ev_pool[c.ev_used++], st));
                    const size_t lds_b = lds + (size_t)wpb * sub * sizeof(P4);   // + one work list per wave (one slot per ray of a sub-chunk)
                    if (b == 0) launch_bounce<T, true>(R, dim3(G), lds_b, st, a);
                    else launch_bounce<T, false>(R, dim3(G), lds_b, st, a);
                    if (int rc = lds_optin_failed()) return rc;
                    if (profile) HIP_TRY(hipEventRecord(c.ev_pool[c.ev_used++], st));
                    ++launches;
                }
            }
            uint32_t rblocks = std::min<uint32_t>((uint32_t)((tile_pixels + spira::kBlock - 1) / spira::kBlock), max_blocks);
            hipLaunchKernelGGL((spira::k_resolve<T>), dim3(rblocks), dim3(spira::kBlock), 0, st, (P4 *)c.accum.p, (const spira::Pack3<T> *)c.L.p,
                               (uint32_t)tile_pixels, k_eff, (pass == 0 && !progressive) ? 1 : 0, mega ? (const uint32_t *)nullptr : (const uint32_t *)c.blkstats.p,
                               stat_rows, (spira::Stats *)c.stats.p);
            ++launches;
        }
    }
    {
        uint32_t fblocks = std::min<uint32_t>((uint32_t)((tile_pixels + spira::kBlock - 1) / spira::kBlock), max_blocks);
        if (progressive)      // hand the running sums back untouched (x / 1 is exact)
            hipLaunchKernelGGL((spira::k_finalize<T>), dim3(fblocks), dim3(spira::kBlock), 0, st, (const P4 *)c.accum.p, (uint32_t)tile_pixels,
                               1u, (uint32_t)SPIRA_POST_NONE, d_hdr, (T *)nullptr);
        else
            hipLaunchKernelGGL((spira::k_finalize<T>), dim3(fblocks), dim3(spira::kBlock), 0, st, (const P4 *)c.accum.p, (uint32_t)tile_pixels,
                               p->spp, sem == SPIRA_SEM_HYBRID ? (uint32_t)SPIRA_POST_NONE : (p->flags & SPIRA_POST_MASK), d_hdr, d_img);      // (HYBRID: the sum is already tone-mapped, K7 per sample)
        ++launches;
    }
    if (progressive && rng_states && !out_on_device)
        HIP_TRY(hipMemcpyAsync(rng_states, d_rng, tile_pixels * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    if (int rc = lds_optin_failed()) return rc;
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(c.ev_stop, st));
    HIP_TRY(hipMemcpyAsync(c.h_stats, c.stats.p, sizeof(spira::Stats), hipMemcpyDeviceToHost, st));
    lap("enqueue");

    if (!cont) c.last = spira_counters{};
    c.last.samples += (uint64_t)p->spp * tile_pixels;
    c.last.passes += p->max_depth ? n_pass : 0;
    c.last.launches += launches;
    c.last.bounce_launches += metal_launches ? metal_launches : ((mega || !p->max_depth) ? 0 : (uint64_t)n_pass * (persistent ? 1 : p->max_depth));
    c.last_valid = true;
    c.last_pending = true;
    c.last_stream = st;

    if (!out_on_device) {
        void *const dst[2] = {out_hdr, out_img};
        const void *const src[2] = {d_hdr, d_img};
        if (int rc = copy_out(c, st, dst, src, 3 * tile_pixels * sizeof(T))) return rc;
    }
    if (int rc = mark_done(c, st)) return rc;
    if (!out_on_device) HIP_TRY(hipStreamSynchronize(st));
    lap("copy_out+sync");
    return 0;
}

// render_impl<T> of whichever translation unit holds the kernels of T
template <class T>
int render_entry_plain(const spira_scene *h, const T *spheres5, const T *materials8, const T *triangles10, const T *camera12, const spira_params *p,
                       T *out_hdr, T *out_img, bool out_on_device, void *user_stream, bool progressive = false, uint32_t sample0 = 0, uint32_t *rng_states = nullptr,
                       const SlabCtl *slab = nullptr) {
#ifdef SPIRA_TU_MAIN
    if constexpr (sizeof(T) == 4)
        return spira_tu::render_impl_f32(h, spheres5, materials8, triangles10, camera12, p, out_hdr, out_img, out_on_device, user_stream, progressive, sample0, rng_states, slab);
    else
#endif
        return render_impl<T>(h, spheres5, materials8, triangles10, camera12, p, out_hdr, out_img, out_on_device, user_stream, progressive, sample0, rng_states, slab);
}

#ifdef SPIRA_TU_MAIN
// A large frame for a host-pointer caller, rendered as row slabs: slab k's planes go device -> pinned staging on a second stream while slab k + 1
// renders, and the host threads of copy_out move them on into the caller's memory.  The copy of a 1080p frame (Float64 HDR: 49.8 MB, 5 ms into
// pageable memory — as long as rendering it) then hides behind the kernels but for the last slab's share.  The RNG is keyed by the global pixel, so
// the slabs are, bit for bit, the rows of the frame rendered whole (tests/test_gpu_runtime.py); the counters of the call add up over its slabs.
// *done = false: the frame is not of that kind (small, striped, progressive, ...) and nothing was touched — the caller takes the plain path.
template <class T>
int render_host_slabs(const spira_scene *h, const T *spheres5, const T *materials8, const T *triangles10, const T *camera12, const spira_params *p,
                      T *out_hdr, T *out_img, bool *done) {
    *done = false;
    if (!p || !camera12 || (!out_hdr && !out_img)) return 0;
    const uint32_t S_env = env_u32("SPIRA_HOST_SLABS", 0xFFFFFFFFu);                // (unset: chosen below; 0 or 1: never)
    if (S_env < 2) return 0;
    if ((p->flags & SPIRA_SEM_MASK) == SPIRA_SEM_HYBRID) return 0;                   // whole images only
    if (p->rows != 0 && p->stripe_count > 1) return 0;                               // an interleaved tile (its rows are not consecutive image rows)
    // a mesh pass ends with the tail of its fat waves, and four small passes have four of them: configs[4] 6.2 -> 7.8 ms of device time, more than the copy hides
    if ((h ? h->store.nt : (triangles10 ? p->n_triangles : 0)) > SPIRA_LDS_TRIANGLES) return 0;
    const uint32_t W = p->width, rows = p->rows ? p->rows : p->height, row0 = p->rows ? p->row0 : 0;
    if (!W || !rows || !p->spp || (uint64_t)row0 + rows > p->height) return 0;      // (the plain path reports what is wrong)
    const int n_out = (out_hdr ? 1 : 0) + (out_img ? 1 : 0);
    const size_t plane3 = (size_t)3 * rows * W * sizeof(T), total = plane3 * (size_t)n_out;
    // a slab costs ~0.1 ms of device time (its own launches and their tails) and hides its share of the copy: two for a 1080p Float32 image (24.9 MB: 4.7 -> 4.0 ms
    // end to end), four from 32 MB on (1080p Float64 HDR, 49.8 MB: 7.3 -> 6.6 ms into touched memory; profiles/experiments/r04_host_slabs_probe.py)
    const uint32_t S = S_env != 0xFFFFFFFFu ? std::min<uint32_t>(S_env, 16) : (total < ((size_t)32 << 20) ? 2u : 4u);
    if (total < ((size_t)8 << 20) || total > ((size_t)512 << 20) || rows < 16 * S || (uint64_t)rows * W * p->spp < ((uint64_t)16 << 20)) return 0;
    Ctx *cp = nullptr;
    if (int rc = get_ctx(&cp)) return rc;
    Ctx &c = *cp;
    std::lock_guard<std::recursive_mutex> lock(c.mu);
    if (c.h_stage_cap < total) {
        if (c.h_stage) { (void)hipHostFree(c.h_stage); c.h_stage = nullptr; c.h_stage_cap = 0; }
        if (hipHostMalloc(&c.h_stage, total, hipHostMallocDefault) == hipSuccess) c.h_stage_cap = total;
        else { c.h_stage = nullptr; (void)hipGetLastError(); return 0; }                // no pinned memory to be had: the plain path still works
    }
    if (!c.copy_stream) HIP_TRY(hipStreamCreateWithFlags(&c.copy_stream, hipStreamNonBlocking));
    if (!c.ev_slab) HIP_TRY(hipEventCreateWithFlags(&c.ev_slab, hipEventDisableTiming));
    if (int rc = order_after_previous(c, c.stream)) return rc;                       // (out_tmp may still be read by the previous call)
    if (int rc = c.out_tmp.ensure(2 * plane3)) return rc;
    *done = true;
    Lap lap("host slabs");
    struct Piece { char *dst; size_t off, len; };
    std::vector<Piece> pieces;
    size_t off = 0;
    int rc_all = 0;
    for (uint32_t s = 0, r0 = 0; s < S && !rc_all; ++s) {
        const uint32_t rs = rows / S + (s < rows % S ? 1u : 0u);
        spira_params ps = *p;
        ps.row0 = row0 + r0; ps.rows = rs; ps.stripe_h = 0; ps.stripe_count = 0; ps.stripe_rank = 0;
        // the slab's planar block [3][rs][W] sits at element 3 * W * r0 of its output's device frame
        T *d_hdr = out_hdr ? (T *)c.out_tmp.p + (size_t)3 * W * r0 : nullptr;
        T *d_img = out_img ? (T *)((char *)c.out_tmp.p + plane3) + (size_t)3 * W * r0 : nullptr;
        const SlabCtl ctl{s, S};
        rc_all = render_entry_plain<T>(h, spheres5, materials8, triangles10, camera12, &ps, d_hdr, d_img, true, (void *)c.stream, false, 0, nullptr, &ctl);
        if (rc_all) break;
        hipError_t e = hipEventRecord(c.ev_slab, c.stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(c.copy_stream, c.ev_slab, 0);
        T *const dev[2] = {d_hdr, d_img};
        T *const host[2] = {out_hdr, out_img};
        for (int k = 0; k < 2 && e == hipSuccess; ++k) {
            if (!host[k]) continue;
            for (int pl = 0; pl < 3 && e == hipSuccess; ++pl) {
                const size_t len = (size_t)rs * W * sizeof(T);
                e = hipMemcpyAsync((char *)c.h_stage + off, dev[k] + (size_t)pl * rs * W, len, hipMemcpyDeviceToHost, c.copy_stream);
                if (e != hipSuccess) break;
                if (c.stage_ev.size() <= pieces.size()) {
                    hipEvent_t ev;
                    e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
                    if (e != hipSuccess) break;
                    c.stage_ev.push_back(ev);
                }
                e = hipEventRecord(c.stage_ev[pieces.size()], c.copy_stream);
                pieces.push_back({(char *)(host[k] + ((size_t)pl * rows + r0) * W), off, len});
                off += len;
            }
        }
        if (e != hipSuccess) rc_all = fail(SPIRA_E_HIP, std::string("host-output slabs: ") + hipGetErrorString(e));
        r0 += rs;
    }
    if (rc_all) {                        // drain what was enqueued; nothing of the frame is promised
        (void)hipStreamSynchronize(c.stream); (void)hipStreamSynchronize(c.copy_stream);
        return rc_all;
    }
    lap("enqueue");
    const int n_thr = (int)std::min<size_t>(std::max<uint32_t>(1, env_u32("SPIRA_STAGE_THREADS", 4)), pieces.size());
    std::vector<hipError_t> errs((size_t)n_thr, hipSuccess);
    const bool prefault = env_u32("SPIRA_PREFAULT", 1) != 0;
    auto mover = [&](int t) {
        (void)hipSetDevice(c.device);
        if (prefault) for (size_t i = (size_t)t; i < pieces.size(); i += (size_t)n_thr) prefault_destination(pieces[i].dst, pieces[i].len);
        for (size_t i = (size_t)t; i < pieces.size(); i += (size_t)n_thr) {
            const hipError_t e = hipEventSynchronize(c.stage_ev[i]);
            if (e != hipSuccess) { errs[(size_t)t] = e; return; }
            std::memcpy(pieces[i].dst, (const char *)c.h_stage + pieces[i].off, pieces[i].len);
        }
    };
    std::vector<std::thread> thr;
    for (int t = 1; t < n_thr; ++t) thr.emplace_back(mover, t);
    mover(0);
    for (auto &t : thr) t.join();
    lap("movers");
    HIP_TRY(hipStreamSynchronize(c.copy_stream));
    HIP_TRY(hipStreamSynchronize(c.stream));
    for (hipError_t e : errs) if (e != hipSuccess) return fail(SPIRA_E_HIP, std::string("host-output slabs: ") + hipGetErrorString(e));
    return 0;
}
#endif

template <class T>
int render_entry(const spira_scene *h, const T *spheres5, const T *materials8, const T *triangles10, const T *camera12, const spira_params *p,
                 T *out_hdr, T *out_img, bool out_on_device, void *user_stream, bool progressive = false, uint32_t sample0 = 0, uint32_t *rng_states = nullptr) {
#ifdef SPIRA_TU_MAIN
    if (!out_on_device && !progressive) {
        bool done = false;
        const int rc = render_host_slabs<T>(h, spheres5, materials8, triangles10, camera12, p, out_hdr, out_img, &done);
        if (done || rc) return rc;
    }
#endif
    return render_entry_plain<T>(h, spheres5, materials8, triangles10, camera12, p, out_hdr, out_img, out_on_device, user_stream, progressive, sample0, rng_states);
}

template <class T>
int trace_impl(const T *spheres5, const T *materials8, const T *triangles10, const T *camera12, const spira_params *p,
               uint32_t n_paths, const uint32_t *ijs, int *prims, T *ts, T *dirs, T *radiance) {
    uint32_t rows = 0;
    if (!p) return fail(SPIRA_E_INVALID, "params is NULL");
    tl_lds_optin = hipSuccess;
    if (int rc = validate_scene<T>(spheres5, materials8, triangles10, p->n_spheres, p->n_materials, triangles10 ? p->n_triangles : 0)) return rc;
    if (int rc = validate_params(camera12, p, triangles10 ? p->n_triangles : 0, &rows)) return rc;
    if (!n_paths || !ijs || !prims || !ts || !dirs || !radiance) return fail(SPIRA_E_INVALID, "NULL argument");
    if (p->max_depth < 1) return fail(SPIRA_E_INVALID, "max_depth must be >= 1");
    if ((p->flags & SPIRA_SEM_MASK) == SPIRA_SEM_HYBRID) return fail(SPIRA_E_UNSUPPORTED, "SPIRA_SEM_HYBRID has no per-path trace (its samples advance image-wide in lock step)");
    for (uint32_t k = 0; k < n_paths; ++k)
        if (ijs[3 * k] < 1 || ijs[3 * k] > p->width || ijs[3 * k + 1] < 1 || ijs[3 * k + 1] > p->height || ijs[3 * k + 2] >= p->spp)
            return fail(SPIRA_E_INVALID, "path (i, j, sample) out of range");
    Ctx *cp = nullptr;
    if (int rc = get_ctx(&cp)) return rc;
    Ctx &c = *cp;
    std::lock_guard<std::recursive_mutex> lock(c.mu);
    hipStream_t st = c.stream;
    if (int rc = order_after_previous(c, st)) return rc;
    spira::BounceArgs<T> a{};
    if (int rc = acquire_scene<T>(c, st, (const spira_scene *)nullptr, spheres5, materials8, triangles10, p, a.scene)) return rc;
    if (int rc = attach_spd<T>(c, st, p, a.scene)) return rc;
    fill_const<T>(a.rc, camera12, p, rows, 1);
    size_t nseg = (size_t)n_paths * p->max_depth;
    size_t b_ij = ((size_t)n_paths * 3 * sizeof(uint32_t) + 255) & ~(size_t)255;
    size_t b_pr = (nseg * sizeof(int) + 255) & ~(size_t)255;
    size_t b_ts = (nseg * sizeof(T) + 255) & ~(size_t)255;
    size_t b_di = (nseg * 3 * sizeof(T) + 255) & ~(size_t)255;
    size_t b_ra = ((size_t)n_paths * 3 * sizeof(T) + 255) & ~(size_t)255;
    if (int rc = c.trace.ensure(b_ij + b_pr + b_ts + b_di + b_ra)) return rc;
    char *base = (char *)c.trace.p;
    uint32_t *d_ij = (uint32_t *)base;
    int *d_pr = (int *)(base + b_ij);
    T *d_ts = (T *)(base + b_ij + b_pr);
    T *d_di = (T *)(base + b_ij + b_pr + b_ts);
    T *d_ra = (T *)(base + b_ij + b_pr + b_ts + b_di);
    HIP_TRY(hipMemcpyAsync(d_ij, ijs, (size_t)n_paths * 3 * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemsetAsync(d_ts, 0, b_ts + b_di, st));
    const size_t lds = spira::scene_lds_bytes<T>(a.scene.n_spheres, a.scene.n_materials, a.scene.n_triangles);
    const uint32_t sem = p->flags & SPIRA_SEM_MASK;
    if (sem == SPIRA_SEM_CPU) launch_lds(spira::k_trace_variant<T, 1>, dim3((n_paths + 63) / 64), dim3(64), lds, st, a, d_ij, n_paths, d_pr, d_ts, d_di, d_ra);
    else if (sem == SPIRA_SEM_METAL) launch_lds(spira::k_trace_variant<T, 2>, dim3((n_paths + 63) / 64), dim3(64), lds, st, a, d_ij, n_paths, d_pr, d_ts, d_di, d_ra);
    else {
        const bool ext = (p->flags & (SPIRA_EXT_DIELECTRIC | SPIRA_EXT_SPECTRAL)) != 0;
        const dim3 tg((n_paths + 63) / 64), tb(64);
        if (a.scene.n_bvh_tris) { if (ext) launch_lds(spira::k_trace<T, true, true>, tg, tb, lds, st, a, d_ij, n_paths, d_pr, d_ts, d_di, d_ra); else launch_lds(spira::k_trace<T, true, false>, tg, tb, lds, st, a, d_ij, n_paths, d_pr, d_ts, d_di, d_ra); }
        else { if (ext) launch_lds(spira::k_trace<T, false, true>, tg, tb, lds, st, a, d_ij, n_paths, d_pr, d_ts, d_di, d_ra); else launch_lds(spira::k_trace<T, false, false>, tg, tb, lds, st, a, d_ij, n_paths, d_pr, d_ts, d_di, d_ra); }
    }
    if (int rc = lds_optin_failed()) return rc;
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(prims, d_pr, nseg * sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(ts, d_ts, nseg * sizeof(T), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(dirs, d_di, nseg * 3 * sizeof(T), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(radiance, d_ra, (size_t)n_paths * 3 * sizeof(T), hipMemcpyDeviceToHost, st));
    if (int rc = mark_done(c, st)) return rc;
    HIP_TRY(hipStreamSynchronize(st));
    return 0;
}

// Camera constructor arithmetic, host side.  Statement order of
// examples/julia-raytracer.jl:280-291 == src/spira-metal-optimized.jl:332-345 (focus_dist = 1).
template <class T> struct HV { T x, y, z; };
template <class T> HV<T> hsub(HV<T> a, HV<T> b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
template <class T> HV<T> hscale(HV<T> a, T s) { return {a.x * s, a.y * s, a.z * s}; }
template <class T> HV<T> hdiv(HV<T> a, T s) { return {a.x / s, a.y / s, a.z / s}; }
template <class T> T hdot(HV<T> a, HV<T> b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <class T> HV<T> hcross(HV<T> a, HV<T> b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
template <class T> HV<T> hnorm(HV<T> a) { return hdiv(a, (T)std::sqrt(hdot(a, a))); }

template <class T>
void camera_impl(const T *position, const T *look_at, const T *up, T fov_deg, T aspect, T focus_dist, T *out12) {
    HV<T> pos{position[0], position[1], position[2]}, la{look_at[0], look_at[1], look_at[2]}, vup{up[0], up[1], up[2]};
    T theta = fov_deg * ((T)3.14159265358979323846 / (T)180);     // deg2rad(z) = z * (oftype(z, pi) / 180)
    T h = std::tan(theta / 2);
    T vh = (T)2.0 * h;
    T vw = aspect * vh;
    HV<T> w = hnorm(hsub(pos, la));
    HV<T> u = hnorm(hcross(vup, w));
    HV<T> v = hcross(w, u);
    HV<T> hor = hscale(u, focus_dist * vw);
    HV<T> ver = hscale(v, focus_dist * vh);
    HV<T> llc = hsub(hsub(hsub(pos, hdiv(hor, (T)2)), hdiv(ver, (T)2)), hscale(w, focus_dist));
    T o[12] = {pos.x, pos.y, pos.z, llc.x, llc.y, llc.z, hor.x, hor.y, hor.z, ver.x, ver.y, ver.z};
    std::memcpy(out12, o, sizeof o);
}

// n_devices == 0: a handle on the calling thread's device; n_devices >= 1: one validated and built ONCE, resident on devices 0 .. n_devices-1
template <class T>
int scene_create(const T *spheres5, const T *materials8, const T *triangles10, uint32_t n_spheres, uint32_t n_materials,
                        uint32_t n_triangles, spira_scene **out, int n_devices = 0) {
    if (!out) return fail(SPIRA_E_INVALID, "out is NULL");
    *out = nullptr;
    const uint32_t nt = triangles10 ? n_triangles : 0;
    if (int rc = validate_scene<T>(spheres5, materials8, triangles10, n_spheres, n_materials, nt)) return rc;
    const bool multi = n_devices > 0;
    if (multi && (n_devices > kMaxDevices || (n_devices > spira_device_count() && !env_u32("SPIRA_MULTI_REHEARSE", 0))))
        return fail(SPIRA_E_INVALID, "n_devices out of range (1 .. spira_device_count())");
    const int caller_device = tl_device;
    const int n_phys = multi ? std::min(n_devices, std::max(1, spira_device_count())) : 1;      // (rehearsal: every rank renders on device 0)
    HostBvh<T> hb;                                      // the mesh's tree: built on the first upload, reused by the others
    spira_scene *first = nullptr;
    int rc = 0;
    for (int d = 0; d < n_phys && !rc; ++d) {
        if (multi) tl_device = d;
        Ctx *cp = nullptr;
        if ((rc = get_ctx(&cp))) break;                // also selects the device
        spira_scene *h = new (std::nothrow) spira_scene();
        if (!h) { rc = fail(SPIRA_E_HIP, "out of host memory"); break; }
        h->magic = kSceneMagic; h->device = tl_device; h->prec = (int)sizeof(T);
        if (!first) first = h; else { first->replica[d] = h; }
        rc = scene_upload<T>(h->store, nullptr, nullptr, spheres5, materials8, triangles10, n_spheres, n_materials, nt, &hb);
        if (!rc && hipStreamSynchronize(nullptr) != hipSuccess) rc = fail(SPIRA_E_HIP, "hipStreamSynchronize failed after the scene upload");
        if (!rc) first->n_replicas = d + 1;
    }
    tl_device = caller_device;
    (void)hipSetDevice(caller_device);
    if (rc) {
        const std::string keep = tl_err;
        if (first) {
            for (int d = 1; d < kMaxDevices; ++d) if (first->replica[d]) { (void)hipSetDevice(first->replica[d]->device); first->replica[d]->store.release(); first->replica[d]->magic = 0; delete first->replica[d]; }
            (void)hipSetDevice(first->device); first->store.release(); first->magic = 0; delete first;
            (void)hipSetDevice(caller_device);
        }
        tl_err = keep;
        return rc;
    }
    *out = first;
    return 0;
}


// ======================================================================= multi-device render (one node)
// RCCL is opened at run time: libspira_hip.so has no link-time dependency on it, and when the host process already
// carries an RCCL (PyTorch does) that copy is the one found.
struct Rccl {
    void *handle = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    // one set of communicators per device count ever asked for (devices 0..n-1), kept until spira_shutdown: a host that alternates
    // between, say, 8-GPU frames and 4-GPU previews does not pay ncclCommInitAll (hundreds of ms) at every switch
    bool have[kMaxDevices + 1] = {};
    ncclComm_t comms_of[kMaxDevices + 1][kMaxDevices] = {};
    ncclComm_t *comms = nullptr;         // the set of the current call
    std::mutex mu;
};
Rccl g_rccl;

int rccl_load(Rccl &r) {
    if (r.handle) return 0;
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        r.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (r.handle) break;
    }
    if (!r.handle) return fail(SPIRA_E_UNSUPPORTED, std::string("RCCL not found (dlopen librccl.so.1): ") + dlerror());
#define SPIRA_RCCL_SYM(field, sym)                                                        \
    r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.handle, #sym));                \
    if (!r.field) { r.handle = nullptr; return fail(SPIRA_E_UNSUPPORTED, "RCCL symbol missing: " #sym); }
    SPIRA_RCCL_SYM(CommInitAll, ncclCommInitAll)
    SPIRA_RCCL_SYM(CommDestroy, ncclCommDestroy)
    SPIRA_RCCL_SYM(CommAbort, ncclCommAbort)
    SPIRA_RCCL_SYM(GroupStart, ncclGroupStart)
    SPIRA_RCCL_SYM(GroupEnd, ncclGroupEnd)
    SPIRA_RCCL_SYM(Send, ncclSend)
    SPIRA_RCCL_SYM(Recv, ncclRecv)
    SPIRA_RCCL_SYM(GetErrorString, ncclGetErrorString)
#undef SPIRA_RCCL_SYM
    return 0;
}

void rccl_release(Rccl &r) {
    for (int n = 1; n <= kMaxDevices; ++n) {
        if (!r.have[n]) continue;
        for (int i = 0; i < n; ++i) if (r.comms_of[n][i]) { (void)r.CommDestroy(r.comms_of[n][i]); r.comms_of[n][i] = nullptr; }
        r.have[n] = false;
    }
    r.comms = nullptr;
}

// after a failed exchange: the communicators of this device count may hold a half-issued group — abort them, the next call makes new ones
void rccl_abort(Rccl &r, int n) {
    if (!r.have[n]) return;
    for (int i = 0; i < n; ++i) if (r.comms_of[n][i]) { (void)r.CommAbort(r.comms_of[n][i]); r.comms_of[n][i] = nullptr; }
    r.have[n] = false;
    r.comms = nullptr;
}

int rccl_comms(Rccl &r, int n) {          // communicators for devices 0..n-1 (ncclCommInitAll once per n)
    if (int rc = rccl_load(r)) return rc;
    if (!r.have[n]) {
        int devs[kMaxDevices];
        for (int i = 0; i < n; ++i) devs[i] = i;
        ncclResult_t e = r.CommInitAll(r.comms_of[n], n, devs);
        if (e != ncclSuccess) return fail(SPIRA_E_HIP, std::string("ncclCommInitAll: ") + r.GetErrorString(e));
        r.have[n] = true;
    }
    r.comms = r.comms_of[n];
    return 0;
}

// Row permutation of the gathered tiles back to image order, for both outputs at once.
//   stack: [n][2][3][max_rows][W] (tile of rank r: hdr planes, then img planes; rows in the rank's local order)
//   full : [2][3][H][W]
// Global row y belongs to rank (y / stripe_h) % n, local row ((y / stripe_h) / n) * stripe_h + y % stripe_h (spira_params "Tiling").
template <class T>
__global__ void k_assemble(const T *stack, T *full, uint32_t n, uint32_t stripe_h, uint32_t max_rows, uint32_t W, uint32_t H) {
    const size_t total = (size_t)6 * H * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t x = (uint32_t)(i % W);
        const uint32_t y = (uint32_t)((i / W) % H);
        const uint32_t pl = (uint32_t)(i / ((size_t)W * H));            // 0..5: output * 3 + channel
        const uint32_t sq = y / stripe_h, r = sq % n, lr = (sq / n) * stripe_h + y % stripe_h;
        full[i] = stack[(((size_t)r * 6 + pl) * max_rows + lr) * W + x];
    }
}

constexpr uint32_t kMultiStripeH = 1;      // (single rows: equal tiles whatever the height; 8-row stripes cost a world-8 rank 4 % — distributed.py)

// `mh`: a scene resident on the devices (spira_scene_create_multi_*) — nothing is validated, hashed, built or uploaded per call — or NULL: host arrays
template <class T>
int render_multi_impl(const spira_scene *mh, const T *spheres5, const T *materials8, const T *triangles10, const T *camera12, const spira_params *p, int n_devices,
                      T *out_hdr, T *out_img) {
    if (!p) return fail(SPIRA_E_INVALID, "params is NULL");
    if (!out_hdr && !out_img) return fail(SPIRA_E_INVALID, "both outputs are NULL");
    if (p->rows != 0 || p->stripe_count != 0) return fail(SPIRA_E_INVALID, "spira_render_multi tiles the frame itself: rows / stripe_* must be 0");
    const int avail = spira_device_count();
    // SPIRA_MULTI_REHEARSE=1 (a one-GPU box): all ranks render on device 0 one after the other and their tiles reach the stack by
    // device copies instead of RCCL — exercises tiling, re-pitching and reassembly for any n; not a measurement
    const bool rehearse = env_u32("SPIRA_MULTI_REHEARSE", 0) != 0;
    if (n_devices < 1 || n_devices > kMaxDevices || (!rehearse && n_devices > avail)) return fail(SPIRA_E_INVALID, "n_devices out of range (1 .. spira_device_count())");
    if (avail < 1) return fail(SPIRA_E_NO_DEVICE, "no HIP device");
    // validate once on the calling thread, so that argument errors are reported before any thread or communicator exists
    if (mh) {
        if (mh->magic != kSceneMagic) return fail(SPIRA_E_INVALID, "scene handle is NULL or was destroyed");
        if (mh->prec != (int)sizeof(T)) return fail(SPIRA_E_INVALID, "scene handle was created in the other precision");
        if (mh->device != 0 || (!rehearse && mh->n_replicas < n_devices)) return fail(SPIRA_E_INVALID, "scene handle is not resident on devices 0 .. n_devices-1 (spira_scene_create_multi_*)");
        uint32_t rows = 0;
        if (int rc = validate_params(camera12, p, mh->store.nt, &rows)) return rc;
    } else {
        uint32_t rows = 0;
        const uint32_t nt = triangles10 ? p->n_triangles : 0;
        if (int rc = validate_scene<T>(spheres5, materials8, triangles10, p->n_spheres, p->n_materials, nt)) return rc;
        if (int rc = validate_params(camera12, p, nt, &rows)) return rc;
    }
    const uint32_t n = (uint32_t)n_devices, W = p->width, H = p->height;
    if (stripe_rows(H, kMultiStripeH, n, n - 1) == 0) return fail(SPIRA_E_INVALID, "image has fewer rows than devices");
    const uint32_t max_rows = stripe_rows(H, kMultiStripeH, n, 0);
    const size_t tile_elems = (size_t)6 * max_rows * W;                 // hdr planes + img planes, padded to the largest tile
    std::lock_guard<std::mutex> rl(g_rccl.mu);                           // one multi-device render at a time
    if (!rehearse) { if (int rc = rccl_comms(g_rccl, n_devices)) return rc; }

    std::vector<int> rcs(n, 0);
    std::vector<std::string> errs(n);
    std::atomic<bool> exchange_failed{false};      // (set by any worker thread whose gate reports a failed exchange)
    const int caller_device = tl_device;
    // Every device finishes (or fails) its allocation + render-enqueue phase before any of them enters the exchange: a device that
    // failed early must not leave device 0 waiting on the stream for a tile that will never be sent.
    std::mutex gate_mu;
    std::condition_variable gate_cv;
    uint32_t gate_arrived = 0;
    bool gate_failed = false;
    uint32_t gate_round = 0;
    auto gate = [&](bool ok) -> bool {        // returns whether ALL devices got here without an error (reusable: the exchange has one behind it too)
        std::unique_lock<std::mutex> lk(gate_mu);
        if (!ok) gate_failed = true;
        const uint32_t my_round = gate_round;
        if (++gate_arrived == n) { gate_arrived = 0; ++gate_round; gate_cv.notify_all(); }
        else gate_cv.wait(lk, [&] { return gate_round != my_round; });
        return !gate_failed;
    };
    auto worker = [&](uint32_t r) {
        tl_device = rehearse ? 0 : (int)r;
        auto bail = [&](int rc) { rcs[r] = rc; errs[r] = tl_err; };
        Ctx *cp = nullptr;
        hipStream_t st = nullptr;
        const uint32_t rows_r_all = (n == 1) ? H : stripe_rows(H, kMultiStripeH, n, r);
        auto phase1 = [&]() -> int {          // workspaces + this device's tile, enqueued on its stream
            if (int rc = get_ctx(&cp)) return rc;
            Ctx &c = *cp;
            st = c.stream;
            {
                std::lock_guard<std::recursive_mutex> lock(c.mu);
                if (int rc = c.multi_tile.ensure(2 * tile_elems * sizeof(T))) return rc;      // the tile + a scratch copy for ragged tiles
                if (r == 0 || rehearse) {
                    if (int rc = c.multi_stack.ensure((size_t)n * tile_elems * sizeof(T))) return rc;
                    if (int rc = c.multi_full.ensure((size_t)6 * H * W * sizeof(T))) return rc;
                }
            }
            spira_params tp = *p;
            tp.rows = rows_r_all;
            tp.row0 = 0; tp.stripe_h = kMultiStripeH; tp.stripe_count = n; tp.stripe_rank = r;
            if (n == 1) { tp.rows = 0; tp.stripe_h = 0; tp.stripe_count = 0; tp.stripe_rank = 0; }
            T *d_hdr = (T *)c.multi_tile.p, *d_img = d_hdr + (size_t)3 * max_rows * W;
            // render_impl writes each output as [3][rows][W] contiguously; the gather wants every tile at the pitch of the largest one
            // ([6][max_rows][W]).  A tile with fewer rows (ragged last stripes) is rendered into the second half of the buffer and its
            // six planes are re-pitched with one strided device copy.
            const spira_scene *hr = !mh ? nullptr : ((rehearse || r == 0) ? mh : mh->replica[r]);
            if (rows_r_all != max_rows) {
                T *scratch = d_hdr + tile_elems;
                if (int rc = render_entry<T>(hr, spheres5, materials8, triangles10, camera12, &tp, scratch, scratch + (size_t)3 * rows_r_all * W, true, st)) return rc;
                hipError_t e = hipMemcpy2DAsync(d_hdr, (size_t)max_rows * W * sizeof(T), scratch, (size_t)rows_r_all * W * sizeof(T), (size_t)rows_r_all * W * sizeof(T), 6,
                                                hipMemcpyDeviceToDevice, st);
                if (e != hipSuccess) return fail(SPIRA_E_HIP, std::string("hipMemcpy2DAsync: ") + hipGetErrorString(e));
                return 0;
            }
            return render_entry<T>(hr, spheres5, materials8, triangles10, camera12, &tp, d_hdr, d_img, true, st);
        };
        const int rc1 = phase1();
        if (rc1) bail(rc1);
        if (!rehearse) { if (!gate(rc1 == 0)) { if (st) (void)hipStreamSynchronize(st); return; } }
        else if (rc1) return;
        Ctx &c = *cp;
        // ---- the one exchange of the path: every tile to device 0 (RCCL point-to-point over xGMI; n-1 transfers arrive at once)
        if (rehearse) {
            hipError_t he = hipMemcpyAsync((T *)c.multi_stack.p + (size_t)r * tile_elems, c.multi_tile.p, tile_elems * sizeof(T), hipMemcpyDeviceToDevice, st);
            if (he != hipSuccess) { tl_err = std::string("hipMemcpyAsync: ") + hipGetErrorString(he); return bail(SPIRA_E_HIP); }
            if (r + 1 < n) return;                 // the last rank assembles
        } else {
        const ncclDataType_t dt = sizeof(T) == 4 ? ncclFloat32 : ncclFloat64;
        ncclResult_t e = g_rccl.GroupStart();
        if (e == ncclSuccess) e = g_rccl.Send(c.multi_tile.p, tile_elems, dt, 0, g_rccl.comms[r], st);
        if (e == ncclSuccess && r == 0)
            for (uint32_t src = 0; src < n && e == ncclSuccess; ++src)
                e = g_rccl.Recv((T *)c.multi_stack.p + (size_t)src * tile_elems, tile_elems, dt, (int)src, g_rccl.comms[0], st);
        ncclResult_t e2 = g_rccl.GroupEnd();
        if (e == ncclSuccess) e = e2;
        if (e != ncclSuccess) { tl_err = std::string("RCCL gather: ") + g_rccl.GetErrorString(e); bail(SPIRA_E_HIP); }
        // A rank whose send could not be issued leaves device 0's receive waiting for ever: every rank learns here whether ALL of them
        // issued their part, and when one did not, nobody synchronises on the exchange — the communicators are aborted after the threads join.
        if (!gate(e == ncclSuccess)) { exchange_failed = true; return; }
        }
        if (rehearse || r == 0) {
            const uint32_t blocks = (uint32_t)std::min<size_t>(((size_t)6 * H * W + 255) / 256, (size_t)c.num_cus * 16);
            hipLaunchKernelGGL((k_assemble<T>), dim3(blocks), dim3(256), 0, st, (const T *)c.multi_stack.p, (T *)c.multi_full.p, n, kMultiStripeH, max_rows, W, H);
            const size_t plane3 = (size_t)3 * H * W * sizeof(T);
            void *const dst[2] = {out_hdr, out_img};
            const void *const src[2] = {c.multi_full.p, (const char *)c.multi_full.p + plane3};
            {
                std::lock_guard<std::recursive_mutex> lock(c.mu);          // (the staging buffer belongs to the context)
                if (int rc = copy_out(c, st, dst, src, plane3)) return bail(rc);
            }
        }
        {
            std::lock_guard<std::recursive_mutex> lock(c.mu);
            if (int rc = mark_done(c, st)) return bail(rc);
        }
        hipError_t he = hipStreamSynchronize(st);
        if (he != hipSuccess) { tl_err = std::string("hipStreamSynchronize: ") + hipGetErrorString(he); return bail(SPIRA_E_HIP); }
    };
    if (rehearse) {
        for (uint32_t r = 0; r < n && !rcs[r ? r - 1 : 0]; ++r) worker(r);
    } else {
        std::vector<std::thread> threads;
        for (uint32_t r = 1; r < n; ++r) threads.emplace_back(worker, r);
        worker(0);                                    // device 0 on the calling thread
        for (auto &t : threads) t.join();
    }
    tl_device = caller_device;
    (void)hipSetDevice(caller_device);
    if (exchange_failed) rccl_abort(g_rccl, n_devices);
    for (uint32_t r = 0; r < n; ++r)
        if (rcs[r]) return fail(rcs[r], "device " + std::to_string(r) + ": " + errs[r]);
    if (exchange_failed) return fail(SPIRA_E_HIP, "RCCL gather failed on another device");
    return 0;
}

}  // namespace

#ifdef SPIRA_TU_F64MESH
int spira_tu::launch_path_mesh_f64(int R, dim3 grid, size_t lds, hipStream_t st, const spira::PathArgs<double> &a, int spec) {
    if (a.mesh_mode == 1) return launch_path_mode<double, true, 1>(R, grid, lds, st, a, spec);
    return launch_path_mode<double, true, 0>(R, grid, lds, st, a, spec);
}
void spira_tu::launch_path_resume_f64(int R, dim3 grid, size_t lds, hipStream_t st, const spira::PathArgs<double> &a) {
    launch_path_resume<double>(R, grid, lds, st, a);
}
#elif defined(SPIRA_TU_F32)
int spira_tu::render_impl_f32(const spira_scene *h, const float *spheres5, const float *materials8, const float *triangles10, const float *camera12, const spira_params *p,
                              float *out_hdr, float *out_img, bool out_on_device, void *user_stream, bool progressive, uint32_t sample0, uint32_t *rng_states, const SlabCtl *slab) {
    return render_impl<float>(h, spheres5, materials8, triangles10, camera12, p, out_hdr, out_img, out_on_device, user_stream, progressive, sample0, rng_states, slab);
}
int spira_tu::trace_impl_f32(const float *spheres5, const float *materials8, const float *triangles10, const float *camera12, const spira_params *p,
                             uint32_t n_paths, const uint32_t *ijs, int *prims, float *ts, float *dirs, float *radiance) {
    return trace_impl<float>(spheres5, materials8, triangles10, camera12, p, n_paths, ijs, prims, ts, dirs, radiance);
}
#else
// ======================================================================= C ABI (SPIRA_TU_MAIN, or the single translation unit)
extern "C" {

int spira_abi_version(void) { return SPIRA_ABI_VERSION; }
#ifndef SPIRA_BUILD_ID
#define SPIRA_BUILD_ID "unknown"
#endif
const char *spira_build_id(void) { return SPIRA_BUILD_ID; }
const char *spira_last_error(void) { return tl_err.c_str(); }

int spira_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { tl_err = std::string("hipGetDeviceCount: ") + hipGetErrorString(e); return 0; }
    return n;
}

int spira_set_device(int device) {
    int n = spira_device_count();
    if (device < 0 || device >= n || device >= kMaxDevices) return fail(SPIRA_E_INVALID, "device index out of range");
    tl_device = device;
    return 0;
}

#ifdef SPIRA_MESH_STATS
// experiment builds only (make stats; not part of include/spira_hip.h): the traversal counters of spira_device.h, optionally reset
extern "C" int spira_debug_mesh_stats(unsigned long long *out32, int reset) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (out32 && hipMemcpyFromSymbol(out32, HIP_SYMBOL(spira::g_mesh_dbg), 32 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) { unsigned long long z[32] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(spira::g_mesh_dbg), z, sizeof z) != hipSuccess) return -1; }
    return 0;
}
#endif

int spira_get_counters(spira_counters *out) {
    if (!out) return fail(SPIRA_E_INVALID, "out is NULL");
    Ctx *cp = nullptr;
    if (int rc = get_ctx(&cp)) return rc;
    Ctx &c = *cp;
    std::lock_guard<std::recursive_mutex> lock(c.mu);
    if (!c.last_valid) return fail(SPIRA_E_INVALID, "no render has run on this device");
    if (c.last_pending) {
        HIP_TRY(hipEventSynchronize(c.ev_stop));
        HIP_TRY(hipStreamSynchronize(c.last_stream));
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, c.ev_start, c.ev_stop));
        c.last.kernel_ms = ms;
        c.last.segments = c.h_stats->segments;
        c.last.rays_enqueued = c.h_stats->rays_enqueued;
        c.last.radiance_rmw = c.h_stats->radiance_rmw;
        c.last.radiance_stores = c.h_stats->radiance_store;
        c.last.redone_waves = c.h_stats->redone_waves;
        c.last.rays_parked = c.h_stats->rays_parked;
        c.last.mesh_wave_trips = c.h_stats->mesh_wave_trips;
        c.last.mesh_lane_trips = c.h_stats->mesh_lane_trips;
        double wms = 0;
        for (size_t i = 0; i < c.ev_mid_used; ++i) {
            float m = 0;
            HIP_TRY(hipEventElapsedTime(&m, c.ev_mid[i], c.ev_pool[c.ev_mid_end[i]]));
            wms += m;
        }
        c.last.walk_kernel_ms = wms;
        double bms = 0;
        for (size_t i = 0; i + 1 < c.ev_used; i += 2) {
            float m = 0;
            HIP_TRY(hipEventElapsedTime(&m, c.ev_pool[i], c.ev_pool[i + 1]));
            bms += m;
        }
        c.last.bounce_kernel_ms = bms;
        c.last_pending = false;
    }
    *out = c.last;
    return 0;
}

void spira_shutdown(void) {
    { std::lock_guard<std::mutex> rl(g_rccl.mu); if (g_rccl.handle) rccl_release(g_rccl); }
    for (int d = 0; d < kMaxDevices; ++d) {
        Ctx &c = g_ctx[d];
        std::lock_guard<std::recursive_mutex> lock(c.mu);
        if (!c.init) continue;
        (void)hipSetDevice(d);
        (void)hipDeviceSynchronize();
        for (int i = 0; i < 2; ++i) { c.qA[i].release(); c.qB[i].release(); c.qC[i].release(); c.qR[i].release(); c.qK[i].release(); c.qX[i].release(); }
        c.mesh_list.release(); c.mesh_count.release();
        c.redo.release(); c.L.release(); c.accum.release(); c.counts.release(); c.blkstats.release(); c.stats.release(); c.scene.release(); c.out_tmp.release(); c.trace.release(); c.rng.release(); c.multi_tile.release(); c.multi_stack.release(); c.multi_full.release(); c.spd32.release(); c.spd64.release(); c.hyb_state.release(); c.hyb_mat.release(); c.hyb_flags.release();
        for (hipEvent_t e : c.ev_pool) (void)hipEventDestroy(e);
        c.ev_pool.clear();
        for (hipEvent_t e : c.ev_mid) (void)hipEventDestroy(e);
        c.ev_mid.clear(); c.ev_mid_end.clear(); c.ev_mid_used = 0;
        (void)hipEventDestroy(c.ev_start); (void)hipEventDestroy(c.ev_stop); (void)hipEventDestroy(c.ev_done);
        c.have_done = false;
        (void)hipHostFree(c.h_stats);
        if (c.h_stage) { (void)hipHostFree(c.h_stage); c.h_stage = nullptr; c.h_stage_cap = 0; }
        for (hipEvent_t e : c.stage_ev) (void)hipEventDestroy(e);
        c.stage_ev.clear();
        (void)hipStreamDestroy(c.stream);
        if (c.copy_stream) { (void)hipStreamDestroy(c.copy_stream); c.copy_stream = nullptr; }
        if (c.ev_slab) { (void)hipEventDestroy(c.ev_slab); c.ev_slab = nullptr; }
        c.init = false; c.last_valid = false;
    }
}

int spira_camera_lookat_f32(const float lookfrom[3], const float lookat[3], const float vup[3], float vfov_deg, float aspect_ratio,
                            float out12[12]) {
    if (!lookfrom || !lookat || !vup || !out12) return fail(SPIRA_E_INVALID, "NULL argument");
    camera_impl<float>(lookfrom, lookat, vup, vfov_deg, aspect_ratio, 1.0f, out12);
    return 0;
}
int spira_camera_lookat_f64(const double position[3], const double look_at[3], const double up[3], double fov_deg, double aspect_ratio,
                            double focus_dist, double out12[12]) {
    if (!position || !look_at || !up || !out12) return fail(SPIRA_E_INVALID, "NULL argument");
    camera_impl<double>(position, look_at, up, fov_deg, aspect_ratio, focus_dist, out12);
    return 0;
}

int spira_render_f32(const float *s, const float *m, const float *t, const float cam[12], const spira_params *p, float *out_hdr, float *out_img) {
    return render_entry<float>(nullptr, s, m, t, cam, p, out_hdr, out_img, false, nullptr);
}
int spira_render_f64(const double *s, const double *m, const double *t, const double cam[12], const spira_params *p, double *out_hdr, double *out_img) {
    return render_entry<double>(nullptr, s, m, t, cam, p, out_hdr, out_img, false, nullptr);
}
int spira_render_device_f32(const float *s, const float *m, const float *t, const float cam[12], const spira_params *p, float *d_hdr,
                            float *d_img, void *stream) {
    return render_entry<float>(nullptr, s, m, t, cam, p, d_hdr, d_img, true, stream);
}
int spira_render_device_f64(const double *s, const double *m, const double *t, const double cam[12], const spira_params *p, double *d_hdr,
                            double *d_img, void *stream) {
    return render_entry<double>(nullptr, s, m, t, cam, p, d_hdr, d_img, true, stream);
}

int spira_accumulate_f32(const float *s, const float *m, const float *t, const float cam[12], const spira_params *p, uint32_t sample0,
                         float *sum_rgb, uint32_t *rng_states) {
    if (!sum_rgb) return fail(SPIRA_E_INVALID, "sum_rgb is NULL");
    return render_entry<float>(nullptr, s, m, t, cam, p, sum_rgb, nullptr, false, nullptr, true, sample0, rng_states);
}
int spira_accumulate_f64(const double *s, const double *m, const double *t, const double cam[12], const spira_params *p, uint32_t sample0,
                         double *sum_rgb, uint32_t *rng_states) {
    if (!sum_rgb) return fail(SPIRA_E_INVALID, "sum_rgb is NULL");
    return render_entry<double>(nullptr, s, m, t, cam, p, sum_rgb, nullptr, false, nullptr, true, sample0, rng_states);
}
int spira_accumulate_device_f32(const float *s, const float *m, const float *t, const float cam[12], const spira_params *p, uint32_t sample0,
                                float *d_sum_rgb, uint32_t *d_rng_states, void *stream) {
    if (!d_sum_rgb) return fail(SPIRA_E_INVALID, "d_sum_rgb is NULL");
    return render_entry<float>(nullptr, s, m, t, cam, p, d_sum_rgb, nullptr, true, stream, true, sample0, d_rng_states);
}
int spira_accumulate_device_f64(const double *s, const double *m, const double *t, const double cam[12], const spira_params *p, uint32_t sample0,
                                double *d_sum_rgb, uint32_t *d_rng_states, void *stream) {
    if (!d_sum_rgb) return fail(SPIRA_E_INVALID, "d_sum_rgb is NULL");
    return render_entry<double>(nullptr, s, m, t, cam, p, d_sum_rgb, nullptr, true, stream, true, sample0, d_rng_states);
}

// ---- scene handles: validate + build + upload once, render many times
int spira_scene_create_f32(const float *spheres5, const float *materials8, const float *triangles10, uint32_t n_spheres, uint32_t n_materials,
                           uint32_t n_triangles, spira_scene **out) {
    return scene_create<float>(spheres5, materials8, triangles10, n_spheres, n_materials, n_triangles, out);
}
int spira_scene_create_f64(const double *spheres5, const double *materials8, const double *triangles10, uint32_t n_spheres, uint32_t n_materials,
                           uint32_t n_triangles, spira_scene **out) {
    return scene_create<double>(spheres5, materials8, triangles10, n_spheres, n_materials, n_triangles, out);
}
int spira_scene_create_multi_f32(const float *spheres5, const float *materials8, const float *triangles10, uint32_t n_spheres, uint32_t n_materials,
                                 uint32_t n_triangles, int n_devices, spira_scene **out) {
    if (n_devices < 1) return fail(SPIRA_E_INVALID, "n_devices out of range (1 .. spira_device_count())");
    return scene_create<float>(spheres5, materials8, triangles10, n_spheres, n_materials, n_triangles, out, n_devices);
}
int spira_scene_create_multi_f64(const double *spheres5, const double *materials8, const double *triangles10, uint32_t n_spheres, uint32_t n_materials,
                                 uint32_t n_triangles, int n_devices, spira_scene **out) {
    if (n_devices < 1) return fail(SPIRA_E_INVALID, "n_devices out of range (1 .. spira_device_count())");
    return scene_create<double>(spheres5, materials8, triangles10, n_spheres, n_materials, n_triangles, out, n_devices);
}
int spira_render_multi_scene_f32(const spira_scene *scene, const float cam[12], const spira_params *p, int n_devices, float *out_hdr, float *out_img) {
    if (!scene) return fail(SPIRA_E_INVALID, "scene handle is NULL or was destroyed");
    return render_multi_impl<float>(scene, nullptr, nullptr, nullptr, cam, p, n_devices, out_hdr, out_img);
}
int spira_render_multi_scene_f64(const spira_scene *scene, const double cam[12], const spira_params *p, int n_devices, double *out_hdr, double *out_img) {
    if (!scene) return fail(SPIRA_E_INVALID, "scene handle is NULL or was destroyed");
    return render_multi_impl<double>(scene, nullptr, nullptr, nullptr, cam, p, n_devices, out_hdr, out_img);
}
int spira_scene_destroy(spira_scene *scene) {
    if (!scene) return 0;
    if (scene->magic != kSceneMagic) return fail(SPIRA_E_INVALID, "scene handle was already destroyed");
    for (int d = kMaxDevices - 1; d >= 0; --d) {           // the replicas of a multi-device handle first, then the handle itself
        spira_scene *h = d ? scene->replica[d] : scene;
        if (!h) continue;
        if (hipSetDevice(h->device) != hipSuccess) return fail(SPIRA_E_HIP, "hipSetDevice failed");
        h->store.release();                                // hipFree waits for work that still reads the buffers
        h->magic = 0;
        delete h;
    }
    (void)hipSetDevice(tl_device);
    return 0;
}
int spira_render_scene_f32(const spira_scene *scene, const float cam[12], const spira_params *p, float *out_hdr, float *out_img) {
    if (!scene) return fail(SPIRA_E_INVALID, "scene handle is NULL or was destroyed");
    return render_entry<float>(scene, nullptr, nullptr, nullptr, cam, p, out_hdr, out_img, false, nullptr);
}
int spira_render_scene_f64(const spira_scene *scene, const double cam[12], const spira_params *p, double *out_hdr, double *out_img) {
    if (!scene) return fail(SPIRA_E_INVALID, "scene handle is NULL or was destroyed");
    return render_entry<double>(scene, nullptr, nullptr, nullptr, cam, p, out_hdr, out_img, false, nullptr);
}
int spira_render_scene_device_f32(const spira_scene *scene, const float cam[12], const spira_params *p, float *d_hdr, float *d_img, void *stream) {
    if (!scene) return fail(SPIRA_E_INVALID, "scene handle is NULL or was destroyed");
    return render_entry<float>(scene, nullptr, nullptr, nullptr, cam, p, d_hdr, d_img, true, stream);
}
int spira_render_scene_device_f64(const spira_scene *scene, const double cam[12], const spira_params *p, double *d_hdr, double *d_img, void *stream) {
    if (!scene) return fail(SPIRA_E_INVALID, "scene handle is NULL or was destroyed");
    return render_entry<double>(scene, nullptr, nullptr, nullptr, cam, p, d_hdr, d_img, true, stream);
}

// ---- multi-device render on one node: interleaved 8-row stripes, one host thread + stream per device, one RCCL gather
int spira_render_multi_f32(const float *s, const float *m, const float *t, const float cam[12], const spira_params *p, int n_devices,
                           float *out_hdr, float *out_img) {
    return render_multi_impl<float>(nullptr, s, m, t, cam, p, n_devices, out_hdr, out_img);
}
int spira_render_multi_f64(const double *s, const double *m, const double *t, const double cam[12], const spira_params *p, int n_devices,
                           double *out_hdr, double *out_img) {
    return render_multi_impl<double>(nullptr, s, m, t, cam, p, n_devices, out_hdr, out_img);
}

int spira_trace_paths_f32(const float *s, const float *m, const float *t, const float cam[12], const spira_params *p, uint32_t n_paths,
                          const uint32_t *ijs, int *prims, float *ts, float *dirs, float *radiance) {
#ifdef SPIRA_TU_MAIN
    return spira_tu::trace_impl_f32(s, m, t, cam, p, n_paths, ijs, prims, ts, dirs, radiance);
#else
    return trace_impl<float>(s, m, t, cam, p, n_paths, ijs, prims, ts, dirs, radiance);
#endif
}
int spira_trace_paths_f64(const double *s, const double *m, const double *t, const double cam[12], const spira_params *p, uint32_t n_paths,
                          const uint32_t *ijs, int *prims, double *ts, double *dirs, double *radiance) {
    return trace_impl<double>(s, m, t, cam, p, n_paths, ijs, prims, ts, dirs, radiance);
}

int spira_tonemap_f32(float *values, uint64_t n, uint32_t post) {
    if (!values && n) return fail(SPIRA_E_INVALID, "values is NULL");
    post &= SPIRA_POST_MASK;
    for (uint64_t i = 0; i < n; ++i) values[i] = spira::post1<float>(values[i], post);
    return 0;
}

uint32_t spira_stripe_rows(uint32_t height, uint32_t stripe_h, uint32_t stripe_count, uint32_t stripe_rank) {
    if (stripe_count == 0 || stripe_rank >= stripe_count) return 0;
    return stripe_rows(height, stripe_h, stripe_count, stripe_rank);
}

}  // extern "C"
#endif  // SPIRA_TU_MAIN, or the single translation unit
