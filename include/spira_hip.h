/*
 * spira_hip.h — C ABI of libspira_hip.so, the MI355X (gfx950) path-trace backend for SPIRA.
 *
 * This is the drop-in boundary for ONE hot path of jenkinsm13/julia-spira: the per-pixel
 * Monte-Carlo path-trace integrator.  The reference has no FFI of its own (it is pure Julia);
 * the boundary is the Julia function surface, so every entry point below names the reference
 * function(s) it replaces (paths relative to the reference repo root):
 *
 *   spira_render_f32 / spira_render_device_f32
 *       replaces  render_hybrid_gpu(width,height,scene,camera; samples_per_pixel,max_depth)
 *                 src/spira-metal-optimized.jl:1228-1343   (the whole per-sample / per-depth
 *                 host loop with its K3..K10 kernels), reached from
 *                 render(scene,camera,W,H; ...)  src/spira-metal-optimized.jl:1453-1490
 *                 at the backend branch :1460-1479 (where `has_amdgpu` falls back to the CPU today),
 *       and       render(world,camera,W,H; samples_per_pixel,max_depth)
 *                 examples/julia-raytracer.jl:387-421 (the parity oracle's pixel loop).
 *   inputs        spheres5 / materials8 are exactly the flat arrays of
 *                 prepare_scene_data(scene)  src/spira-metal-optimized.jl:515-542;
 *                 camera12 is Camera_jl (origin, lower_left_corner, horizontal, vertical)
 *                 src/spira-metal-optimized.jl:360-365 == Camera fields :325-348 and
 *                 examples/julia-raytracer.jl:261-295;
 *                 triangles10 is [v0 v1 v2 material] per triangle, the flattened form of
 *                 Triangle(vertices, material) examples/julia-raytracer.jl:85-94.
 *   spira_render_f64 / spira_render_device_f64
 *       the same path computed in Float64, the precision of examples/julia-raytracer.jl.
 *   spira_camera_lookat_f32/_f64
 *       replaces  Camera(lookfrom,lookat,vup,vfov,aspect_ratio) src/spira-metal-optimized.jl:331-347
 *       and       Camera(; position, look_at, up, fov, aspect_ratio, focus_dist)
 *                 examples/julia-raytracer.jl:271-294.
 *   spira_tonemap_f32
 *       replaces  to_acescg examples/julia-raytracer.jl:370-384, gpu_tone_map_kernel!
 *                 src/spira-metal-optimized.jl:1128-1144 and the clamp+sqrt of
 *                 render_with_cpu :1441-1442.
 *   spira_accumulate_f32/_f64 (+ _device_)
 *       the progressive contract of src/spira_path_trace_kernel.metal:143-145,:252-268
 *       (current_sample_index, persisted rng_states, output += L), which no reference host code drives.
 *   flags & SPIRA_SEM_MASK selects which of the reference's FOUR estimators runs:
 *       ray_color (examples/julia-raytracer.jl:328-367, default), trace_ray of render_with_cpu
 *       (src/spira-metal-optimized.jl:1351-1412), path_trace (src/spira_path_trace_kernel.metal:140-269), render_hybrid_gpu's own
 *       (src/spira-metal-optimized.jl:1228-1343).
 *
 * Conventions: plain pointers and sizes only; every function returns 0 on success and a
 * negative SPIRA_E_* code otherwise; nothing throws, aborts or calls back across the ABI;
 * spira_last_error() returns a thread-local message.  Host-pointer entry points copy in/out;
 * *_device_* entry points take DEVICE pointers for the outputs (e.g. a torch tensor's
 * data_ptr()) plus the hipStream_t to run on (as void*; NULL = the null stream) and do not
 * synchronise.  Streams: all calls on one device share that device's workspaces (ray queues, radiance
 * buffers, the scene of host-array calls), so the library orders them itself — every call makes its
 * stream wait (hipStreamWaitEvent) for the end of the previous call on that device, whatever stream that
 * one ran on.  Calls on different streams are therefore safe without host synchronisation, and run one
 * after the other on the device.  There is NO CPU fallback: every render entry point fails with
 * SPIRA_E_NO_DEVICE when no HIP device is usable.
 */
#ifndef SPIRA_HIP_H
#define SPIRA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPIRA_ABI_VERSION 3      /* 3: spira_counters grew (rays_parked in round 3, the traversal counters and walk_kernel_ms in round 4) */

/* ---- error codes ---- */
#define SPIRA_OK            0
#define SPIRA_E_INVALID    -1   /* bad argument (null pointer, zero size, out-of-range index) */
#define SPIRA_E_NO_DEVICE  -2   /* no usable HIP device / runtime error at init */
#define SPIRA_E_HIP        -3   /* a HIP runtime call failed (see spira_last_error) */
#define SPIRA_E_LIMIT      -4   /* scene / parameters exceed a documented limit */
#define SPIRA_E_UNSUPPORTED -5  /* flag combination not implemented */

/* ---- flags (spira_params.flags) ---- */
/* integrator semantics (which of the reference's variants) */
#define SPIRA_SEM_MASK          0x0000000Fu
#define SPIRA_SEM_A             0x00000000u  /* examples/julia-raytracer.jl ray_color :328-367 (graded oracle) */
#define SPIRA_SEM_CPU           0x00000001u  /* render_with_cpu trace_ray src/spira-metal-optimized.jl:1351-1412 */
#define SPIRA_SEM_METAL         0x00000002u  /* path_trace src/spira_path_trace_kernel.metal:140-269 */
#define SPIRA_SEM_HYBRID        0x00000003u  /* render_hybrid_gpu src/spira-metal-optimized.jl:1228-1343 AS WRITTEN — what render() runs on a Metal / CUDA machine:
                                                the whole image in lock step (K3 raygen with per-pixel xorshift32 :610-697, K4 :700-799, the image-wide
                                                "nothing hit: end the sample" :1303, K5 scatter :862-989, contribution halved per depth :1328, K6 shade of
                                                the LAST bounce only :1071-1105, K7 ACES + sqrt per sample :1128-1144, K8 :1055-1068).  Restated, not
                                                repaired.  Spheres only, whole images only (rows == 0), no accumulate entry; out_hdr = out_img = the
                                                reference's (already tone-mapped) image; SPIRA_POST_* and SPIRA_KERNEL_* are ignored. */
/* kernel organisation */
#define SPIRA_KERNEL_MASK       0x000000F0u
#define SPIRA_KERNEL_DEFAULT    0x00000000u  /* the library's choice = the fastest organisation measured for the estimator:
                                                SPIRA_SEM_A: WAVEFRONT; SPIRA_SEM_METAL, SPIRA_SEM_CPU: one lane per pixel / path */
#define SPIRA_KERNEL_MEGA       0x00000010u  /* one lane walks whole paths in registers (with path regeneration) */
#define SPIRA_KERNEL_BOUNCE     0x00000020u  /* wavefront as in round 1: SoA RAY queues, one launch per bounce (comparison point) */
#define SPIRA_KERNEL_WAVEFRONT  0x00000030u  /* wavefront: SoA hit queues, ballot/popcount compaction, ONE persistent launch per pass in
                                                which every wave walks all bounces on its own queue region */
/* The organisation applies to SPIRA_SEM_A and, for WAVEFRONT vs the rest, to SPIRA_SEM_METAL (WAVEFRONT: every wave owns a block of
 * pixels and walks sample after sample on it, the LCG state travelling in the hit packet; otherwise one lane per pixel walks all its
 * samples — the faster form for that estimator).  SPIRA_SEM_CPU always runs one lane per path. */
/* display transform applied to out_img (out_hdr is always the linear mean) */
#define SPIRA_POST_MASK         0x00000F00u
#define SPIRA_POST_ACES         0x00000000u  /* clamp(aces(x),0,1)        examples/julia-raytracer.jl:370-384 */
#define SPIRA_POST_ACES_GAMMA   0x00000100u  /* sqrt(clamp(aces(x),0,1))  src/spira-metal-optimized.jl:1128-1144 */
#define SPIRA_POST_CLAMP_GAMMA  0x00000200u  /* sqrt(clamp(x,0,1))        src/spira-metal-optimized.jl:1441-1442 */
#define SPIRA_POST_NONE         0x00000300u  /* out_img = out_hdr */
/* row order of the outputs: default row 0 = image top, like hdr_data[height-j+1, i]
 * (examples/julia-raytracer.jl:408) and img[height-j+1, i] (src/spira-metal-optimized.jl:1445) */
#define SPIRA_ROWS_BOTTOM_UP    0x00001000u  /* row 0 = v=0 (bottom), the device-buffer order of :1177-1188 */

/* extensions.  The reference only NAMES these features (README.md:10, a comment at
 * src/spira_path_trace_kernel.metal:225): there is no reference code and so no parity to pin — the semantics below are the build's
 * own, restated in oracle/ and compared GPU vs oracle like everything else, and never part of the graded SPIRA_SEM_A runs.
 * SPIRA_SEM_A only; default (wavefront) and MEGA organisations. */
#define SPIRA_EXT_DIELECTRIC    0x00020000u  /* a material whose roughness is NEGATIVE is a smooth dielectric of refractive index
                                                -roughness, tinted by albedo: Snell refraction, Schlick reflectance, total internal
                                                reflection; one uniform draw per interaction picks reflection or refraction */
#define SPIRA_EXT_SPECTRAL      0x00040000u  /* hero-wavelength spectral transport: one wavelength per path (380..730 nm), RGB
                                                reflectance / emission / sky uplifted with the SPD tables of include/spira_spd.h
                                                (staged into LDS), radiance accumulated as linear sRGB through the wavelength's
                                                colour-matching response */

/* diagnostics */
#define SPIRA_FLAG_PROFILE      0x00010000u  /* SPIRA_KERNEL_BOUNCE: bracket every bounce launch with HIP events (slows the
                                                render).  The default organisation always brackets its one launch per pass, so
                                                spira_counters.bounce_kernel_ms is filled by every wavefront render. */

/* ---- limits ---- */
#define SPIRA_MAX_DEPTH        255u        /* bounce index is packed into 8 bits of the RNG key */
#define SPIRA_MAX_SPP          (1u << 24)  /* sample index is packed into 24 bits of the RNG key */
#define SPIRA_MAX_LDS_SPHERES  1024u        /* spheres are always an LDS-resident linear scan */
#define SPIRA_LDS_TRIANGLES    32u          /* up to this many triangles: LDS linear scan; more: device BVH */
#define SPIRA_MAX_TRIANGLES    (1u << 24)

/*
 * Render parameters: a superset of RenderParams_jl (src/spira-metal-optimized.jl:390-400).
 *
 * Tiling (multi-GPU): the image is width x height; this call renders `rows` output rows.
 * Output row r (0-based, in the row order chosen by the flags) of this call is global output
 * row  y = row0 + r                                   when stripe_count <= 1, and
 *      y = ((r / stripe_h) * stripe_count + stripe_rank) * stripe_h + (r % stripe_h)
 * when stripe_count > 1 (interleaved stripes of stripe_h rows, for load balance).  The RNG is
 * keyed by the GLOBAL pixel, so any tiling reproduces the untiled image bit for bit.
 * rows == 0 means "the whole image" (row0, stripe_* ignored).
 */
typedef struct spira_params {
    uint32_t width, height;
    uint32_t spp;            /* samples_per_pixel */
    uint32_t max_depth;      /* max_depth: maximum number of path segments */
    uint32_t n_spheres, n_materials, n_triangles;
    uint32_t flags;
    uint64_t seed;           /* absent in the reference (it never seeds its RNG) */
    uint32_t row0, rows;
    uint32_t stripe_h, stripe_count, stripe_rank;
    uint32_t batch_rays;     /* wavefront: target rays in flight per pass (0 = library default) */
} spira_params;              /* 64 bytes */

/* Counters of the last render on this thread's device context (for roofline arithmetic). */
typedef struct spira_counters {
    uint64_t samples;        /* camera paths started                                   */
    uint64_t segments;       /* path segments traced (ray/scene intersections)         */
    uint64_t rays_enqueued;  /* rays written to a queue (wavefront)                    */
    uint64_t radiance_rmw;   /* read-modify-writes of the per-path radiance (wavefront)*/
    uint64_t radiance_stores;/* plain 16/32-byte stores of the per-path radiance          */
    uint64_t passes;         /* wavefront passes                                       */
    uint64_t launches;       /* kernel launches                                        */
    double   kernel_ms;      /* device time of the render, HIP events on the render stream */
    double   bounce_kernel_ms;   /* device time spent in the dominant (bounce) kernel      */
    uint64_t bounce_launches;    /* launches of that kernel                                 */
    uint64_t redone_waves;       /* waves whose pass was rendered a second time with the compiler's division (speculative division) */
    uint64_t rays_parked;        /* mesh scenes: rays written to (and read back from) a wave's mesh list: 3 x 16/32 bytes each way */
    uint64_t mesh_wave_trips;    /* mesh scenes: trips of the traversal sessions' walk loop, summed over waves (one trip = one memory round trip of every walking lane) */
    uint64_t mesh_lane_trips;    /* ... and the lanes that took part in them: node visits + triangle tests; / (64 * mesh_wave_trips) = lane utilisation of the walk */
    double   walk_kernel_ms;     /* mesh scenes rendered as two launches: device time of the second (fat-wave, traversal) launches; part of bounce_kernel_ms */
} spira_counters;                /* 120 bytes */

/* ---- library / device ---- */
int         spira_abi_version(void);
const char *spira_build_id(void);             /* first 16 hex digits of the SHA-256 of the kernel sources + Makefile this library was built from (bench.py: which profile belongs to it) */
const char *spira_last_error(void);
int         spira_device_count(void);
int         spira_set_device(int device);      /* device used by subsequent calls on this thread */
int         spira_get_counters(spira_counters *out);
void        spira_shutdown(void);              /* frees cached device workspaces */

/* ---- camera (host arithmetic only; no device needed) ---- */
/* out12 = origin, lower_left_corner, horizontal, vertical.  focus_dist = 1 reproduces the
 * five-argument constructor of src/spira-metal-optimized.jl:331. */
int spira_camera_lookat_f32(const float lookfrom[3], const float lookat[3], const float vup[3],
                            float vfov_deg, float aspect_ratio, float out12[12]);
int spira_camera_lookat_f64(const double position[3], const double look_at[3], const double up[3],
                            double fov_deg, double aspect_ratio, double focus_dist, double out12[12]);

/* ---- render: host pointers ---- */
/* spheres5:   n_spheres   x [cx cy cz r material_index(1-based, stored as a float)]
 * materials8: n_materials x [albedo r g b, emission r g b, metallic|specular, roughness]
 * triangles10:n_triangles x [v0 xyz, v1 xyz, v2 xyz, material_index(1-based)] or NULL
 * out_hdr, out_img: planar, 3 planes of rows*width values (R plane, G plane, B plane);
 *                   either may be NULL.  Scene objects are intersected in the order
 *                   spheres[0..], then triangles[0..] (ties: the later object wins, as in the
 *                   closest-hit scan of examples/julia-raytracer.jl:242-258). */
int spira_render_f32(const float *spheres5, const float *materials8, const float *triangles10,
                     const float camera12[12], const spira_params *params,
                     float *out_hdr, float *out_img);
int spira_render_f64(const double *spheres5, const double *materials8, const double *triangles10,
                     const double camera12[12], const spira_params *params,
                     double *out_hdr, double *out_img);

/* ---- render: device output pointers, asynchronous on `stream` ---- */
int spira_render_device_f32(const float *spheres5, const float *materials8, const float *triangles10,
                            const float camera12[12], const spira_params *params,
                            float *d_out_hdr, float *d_out_img, void *stream);
int spira_render_device_f64(const double *spheres5, const double *materials8, const double *triangles10,
                            const double camera12[12], const spira_params *params,
                            double *d_out_hdr, double *d_out_img, void *stream);

/* ---- multi-device render on ONE node ----
 * Replaces the backend branch of render (src/spira-metal-optimized.jl:1460-1479) for a host that owns several GPUs: the
 * frame is dealt to devices 0..n_devices-1 as interleaved single rows (spira_params "Tiling"; the RNG is keyed by the
 * global pixel, so the result is bit-identical to a one-device render), each device renders its tile on its own stream
 * driven by its own host thread inside the library, no collective runs while rendering, and ONE RCCL exchange (grouped
 * ncclSend / ncclRecv = a gather; n-1 point-to-point transfers over xGMI that arrive at device 0 at once) brings the tiles
 * to device 0, which permutes the rows back to image order and copies the frame to the caller's HOST buffers.
 * params->rows / stripe_* must be 0.  librccl.so.1 is opened at run time (the copy the process already carries, if any);
 * SPIRA_E_UNSUPPORTED when none can be found; the communicators of a device count are made once (ncclCommInitAll) and kept
 * until spira_shutdown.  Afterwards spira_get_counters reports the tile of the calling thread's current device
 * (spira_set_device(d) first for device d's).  A failed exchange aborts the communicators and returns SPIRA_E_HIP: it never hangs.
 * Status: one device through RCCL and n = 2, 3, 8 rehearsed on one device are tested; n > 1 on separate GPUs has not run yet
 * (no multi-GPU box in the build pipeline). */
int spira_render_multi_f32(const float *spheres5, const float *materials8, const float *triangles10,
                           const float camera12[12], const spira_params *params, int n_devices,
                           float *out_hdr, float *out_img);
int spira_render_multi_f64(const double *spheres5, const double *materials8, const double *triangles10,
                           const double camera12[12], const spira_params *params, int n_devices,
                           double *out_hdr, double *out_img);

/* ---- scene handles: validate, build (BVH) and upload a scene ONCE, render it many times ----
 * Replaces the per-render uploads `sphere_data_gpu = MtlArray(sphere_data)` / `material_data_gpu = ...` of
 * render_hybrid_gpu (src/spira-metal-optimized.jl:1247-1254) after prepare_scene_data (:515-542): the host-array
 * entry points above re-validate every material index and hash the whole triangle array on every call (to find
 * the cached BVH), which for an 82 k-triangle mesh is host milliseconds per frame; a handle pays that once.
 * A handle belongs to the device current at creation (spira_set_device) and to one precision.  With a handle,
 * params->n_spheres / n_materials / n_triangles are ignored. */
typedef struct spira_scene spira_scene;
int spira_scene_create_f32(const float *spheres5, const float *materials8, const float *triangles10,
                           uint32_t n_spheres, uint32_t n_materials, uint32_t n_triangles, spira_scene **out);
int spira_scene_create_f64(const double *spheres5, const double *materials8, const double *triangles10,
                           uint32_t n_spheres, uint32_t n_materials, uint32_t n_triangles, spira_scene **out);
int spira_scene_destroy(spira_scene *scene);       /* NULL is a no-op; waits for renders still using it */
/* The same for spira_render_multi_*: ONE validation, ONE BVH build on the host, the scene resident on devices 0 .. n_devices-1
 * (the host-array entry points re-validate, re-hash and, on first use, rebuild the tree per device per call).  The handle is
 * device 0's; spira_scene_destroy frees all copies.  It also works with the single-device entry points on device 0. */
int spira_scene_create_multi_f32(const float *spheres5, const float *materials8, const float *triangles10,
                                 uint32_t n_spheres, uint32_t n_materials, uint32_t n_triangles, int n_devices, spira_scene **out);
int spira_scene_create_multi_f64(const double *spheres5, const double *materials8, const double *triangles10,
                                 uint32_t n_spheres, uint32_t n_materials, uint32_t n_triangles, int n_devices, spira_scene **out);
int spira_render_multi_scene_f32(const spira_scene *scene, const float camera12[12], const spira_params *params, int n_devices,
                                 float *out_hdr, float *out_img);
int spira_render_multi_scene_f64(const spira_scene *scene, const double camera12[12], const spira_params *params, int n_devices,
                                 double *out_hdr, double *out_img);
int spira_render_scene_f32(const spira_scene *scene, const float camera12[12], const spira_params *params,
                           float *out_hdr, float *out_img);
int spira_render_scene_f64(const spira_scene *scene, const double camera12[12], const spira_params *params,
                           double *out_hdr, double *out_img);
int spira_render_scene_device_f32(const spira_scene *scene, const float camera12[12], const spira_params *params,
                                  float *d_out_hdr, float *d_out_img, void *stream);
int spira_render_scene_device_f64(const spira_scene *scene, const double camera12[12], const spira_params *params,
                                  double *d_out_hdr, double *d_out_img, void *stream);

/* ---- progressive accumulation (checkpoint / resume / adaptive sampling) ----
 * The contract the reference's kernel was designed for and no host code uses: `current_sample_index`,
 * persisted `rng_states`, `output_hdr_image[p] += L` (src/spira_path_trace_kernel.metal:143-145, :252-268).
 * Renders samples [sample0, sample0 + params->spp) of every pixel of the tile and ADDS their radiance, in
 * sample order, to sum_rgb (planar 3 x rows x width running sums, caller-owned; zero them before the first
 * call).  image = sum_rgb / total samples; k calls of n samples leave bit for bit the sums of one call of k*n.
 * rng_states (rows*width words, or NULL) is used by SPIRA_SEM_METAL only, whose LCG state runs from sample to
 * sample: written by every call, read when sample0 > 0 — SPIRA_SEM_METAL with sample0 > 0 and rng_states == NULL
 * is SPIRA_E_INVALID (it would replay the first call's samples).  Host pointers, or device pointers + stream. */
int spira_accumulate_f32(const float *spheres5, const float *materials8, const float *triangles10,
                         const float camera12[12], const spira_params *params, uint32_t sample0,
                         float *sum_rgb, uint32_t *rng_states);
int spira_accumulate_f64(const double *spheres5, const double *materials8, const double *triangles10,
                         const double camera12[12], const spira_params *params, uint32_t sample0,
                         double *sum_rgb, uint32_t *rng_states);
int spira_accumulate_device_f32(const float *spheres5, const float *materials8, const float *triangles10,
                                const float camera12[12], const spira_params *params, uint32_t sample0,
                                float *d_sum_rgb, uint32_t *d_rng_states, void *stream);
int spira_accumulate_device_f64(const double *spheres5, const double *materials8, const double *triangles10,
                                const double camera12[12], const spira_params *params, uint32_t sample0,
                                double *d_sum_rgb, uint32_t *d_rng_states, void *stream);

/* ---- diagnostics: per-segment trace of chosen paths (parity tests compare geometry bitwise) ----
 * ijs: n_paths x [i, j, sample] with i in 1..width, j in 1..height (the loop indices of
 * examples/julia-raytracer.jl:392-397) and sample in 0..spp-1.  Outputs, per path and bounce b <
 * max_depth: prims = object index hit (spheres first, then triangles), -1 = miss, -2 = path
 * already ended; ts = hit distance; dirs = the segment's ray direction; radiance = the sample's
 * radiance (n_paths x 3). */
int spira_trace_paths_f32(const float *spheres5, const float *materials8, const float *triangles10,
                          const float camera12[12], const spira_params *params, uint32_t n_paths,
                          const uint32_t *ijs, int *prims, float *ts, float *dirs, float *radiance);
int spira_trace_paths_f64(const double *spheres5, const double *materials8, const double *triangles10,
                          const double camera12[12], const spira_params *params, uint32_t n_paths,
                          const uint32_t *ijs, int *prims, double *ts, double *dirs, double *radiance);

/* ---- post ---- */
/* In-place display transform of n host values (post = one of SPIRA_POST_*). Host arithmetic. */
int spira_tonemap_f32(float *values, uint64_t n, uint32_t post);

/* ---- helpers ---- */
/* Number of output rows a given stripe_rank renders (height rows, stripes of stripe_h). */
uint32_t spira_stripe_rows(uint32_t height, uint32_t stripe_h, uint32_t stripe_count, uint32_t stripe_rank);

#ifdef __cplusplus
}
#endif
#endif /* SPIRA_HIP_H */
