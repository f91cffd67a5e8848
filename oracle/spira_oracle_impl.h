/*
 * spira_oracle_impl.h — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the reference's path-trace integrator, included twice by spira_oracle.c:
 * once with REAL = double (the precision of examples/julia-raytracer.jl) and once with
 * REAL = float (same statements, Float32 arithmetic: the "bit mirror" the f32 HIP kernels are
 * compared with).  Line numbers cite /root/reference (jenkinsm13/julia-spira @ 2025-09-05).
 *
 * PARITY UNPINNED: the reference holds no golden vectors, known-answer tests or fixtures for
 * this path (its only assertion is size(image) == (64,64), tests/bunny-test.jl:59), it cannot
 * be executed in this pipeline (no Julia), and it never seeds its RNG, so its random stream is
 * not reproducible.  This file follows the reference statement by statement in everything
 * except the random stream, which is the build's own counter-based generator (oracle_rng3).
 *
 * Must be compiled with -ffp-contract=off: Julia does not fuse a*b+c, and the HIP kernels are
 * built the same way, so geometry is comparable bit for bit.
 */

/* ---------------------------------------------------------------------------------------- */
/* Vec3 and its operators: examples/julia-raytracer.jl:11-41                                  */
typedef struct { REAL x, y, z; } SUF(V3);
#define V3 SUF(V3)

static inline V3 SUF(v3)(REAL x, REAL y, REAL z) { V3 r = { x, y, z }; return r; }
static inline V3 SUF(add)(V3 a, V3 b) { return SUF(v3)(a.x + b.x, a.y + b.y, a.z + b.z); }   /* :21 */
static inline V3 SUF(sub)(V3 a, V3 b) { return SUF(v3)(a.x - b.x, a.y - b.y, a.z - b.z); }   /* :22 */
static inline V3 SUF(scale)(V3 a, REAL b) { return SUF(v3)(a.x * b, a.y * b, a.z * b); }     /* :23-24 */
static inline V3 SUF(divs)(V3 a, REAL b) { return SUF(v3)(a.x / b, a.y / b, a.z / b); }      /* :25 */
static inline REAL SUF(dot)(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }         /* :26 */
static inline REAL SUF(length)(V3 a) { return SQRT(SUF(dot)(a, a)); }                          /* :27 */
static inline V3 SUF(normalize)(V3 a) { return SUF(divs)(a, SUF(length)(a)); }                 /* :28 */
static inline V3 SUF(mulv)(V3 a, V3 b) { return SUF(v3)(a.x * b.x, a.y * b.y, a.z * b.z); }   /* :32-34 */
static inline V3 SUF(cross)(V3 a, V3 b) {                                                      /* :37-41 */
    return SUF(v3)(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}

/* Material: examples/julia-raytracer.jl:53-62 (fields of materials8: diffuse, emission,
 * specular, roughness — the same eight floats as Material_jl src/spira-metal-optimized.jl:379-384) */
typedef struct { V3 diffuse, emission; REAL specular, roughness; } SUF(Material);
#define Material SUF(Material)

/* HitRecord: examples/julia-raytracer.jl:65-71 (material kept as an index) */
typedef struct { REAL t; V3 position, normal; int material; int hit; int prim; } SUF(HitRecord);
#define HitRecord SUF(HitRecord)

typedef struct { V3 origin, direction; } SUF(Ray);                                             /* :44-47 */
#define Ray SUF(Ray)
static inline V3 SUF(point_at)(Ray r, REAL t) { return SUF(add)(r.origin, SUF(scale)(r.direction, t)); } /* :50 */

typedef struct {
    const REAL *spheres5, *materials8, *triangles10;
    uint32_t n_spheres, n_materials, n_triangles;
    V3 cam_origin, cam_llc, cam_hor, cam_ver;
    uint32_t sA, sB;          /* seed halves after oracle_seed_mix */
    uint32_t max_depth;
    uint32_t sem;
    uint32_t ext;             /* SPIRA_EXT_* bits (the build's own extensions; no reference code exists for them) */
} SUF(World);
#define World SUF(World)

static inline Material SUF(get_material)(const World *w, int idx1) {
    const REAL *m = w->materials8 + 8 * (size_t)(idx1 - 1);
    Material r;
    r.diffuse = SUF(v3)(m[0], m[1], m[2]);
    r.emission = SUF(v3)(m[3], m[4], m[5]);
    r.specular = m[6];
    r.roughness = m[7];
    return r;
}

/* Ray-sphere: examples/julia-raytracer.jl:113-142 */
static HitRecord SUF(hit_sphere)(const REAL *s5, Ray ray, REAL t_min, REAL t_max) {
    HitRecord miss; memset(&miss, 0, sizeof miss);
    V3 center = SUF(v3)(s5[0], s5[1], s5[2]);
    REAL radius = s5[3];
    V3 oc = SUF(sub)(ray.origin, center);                       /* :114 */
    REAL a = SUF(dot)(ray.direction, ray.direction);            /* :115 */
    REAL b = (REAL)2.0 * SUF(dot)(oc, ray.direction);           /* :116 */
    REAL c = SUF(dot)(oc, oc) - radius * radius;                /* :117 */
    REAL discriminant = b * b - (REAL)4 * a * c;                /* :118  (4*a)*c */
    if (discriminant < 0) return miss;                          /* :120 */
    REAL sqrtd = SQRT(discriminant);                            /* :125 */
    REAL root1 = (-b - sqrtd) / ((REAL)2.0 * a);                /* :126 */
    REAL root2 = (-b + sqrtd) / ((REAL)2.0 * a);                /* :127 */
    if (root1 < t_min || root1 > t_max) {                       /* :130 */
        root1 = root2;
        if (root1 < t_min || root1 > t_max) return miss;        /* :132 */
    }
    HitRecord rec;
    rec.t = root1;                                              /* :137 */
    rec.position = SUF(point_at)(ray, rec.t);                   /* :138 */
    rec.normal = SUF(normalize)(SUF(sub)(rec.position, center));/* :139 */
    rec.material = (int)s5[4];
    rec.hit = 1;
    rec.prim = 0;
    return rec;
}

/* Ray-triangle (Möller–Trumbore): examples/julia-raytracer.jl:145-187, normal :105-109 */
static HitRecord SUF(hit_triangle)(const REAL *t10, Ray ray, REAL t_min, REAL t_max) {
    HitRecord miss; memset(&miss, 0, sizeof miss);
    V3 v0 = SUF(v3)(t10[0], t10[1], t10[2]);
    V3 v1 = SUF(v3)(t10[3], t10[4], t10[5]);
    V3 v2 = SUF(v3)(t10[6], t10[7], t10[8]);
    V3 edge1 = SUF(sub)(v1, v0);                                /* :149 */
    V3 edge2 = SUF(sub)(v2, v0);                                /* :150 */
    V3 h = SUF(cross)(ray.direction, edge2);                    /* :153 */
    REAL a = SUF(dot)(edge1, h);                                /* :154 */
    if (FABS(a) < (REAL)1e-8) return miss;                      /* :157 */
    REAL f = (REAL)1.0 / a;                                     /* :161 */
    V3 s = SUF(sub)(ray.origin, v0);                            /* :162 */
    REAL u = f * SUF(dot)(s, h);                                /* :163 */
    if (u < (REAL)0.0 || u > (REAL)1.0) return miss;            /* :165 */
    V3 q = SUF(cross)(s, edge1);                                /* :169 */
    REAL v = f * SUF(dot)(ray.direction, q);                    /* :170 */
    if (v < (REAL)0.0 || u + v > (REAL)1.0) return miss;        /* :172 */
    REAL t = f * SUF(dot)(edge2, q);                            /* :177 */
    if (t < t_min || t > t_max) return miss;                    /* :179 */
    HitRecord rec;
    rec.t = t;
    rec.position = SUF(point_at)(ray, t);                       /* :183 */
    rec.normal = SUF(normalize)(SUF(cross)(edge1, edge2));      /* :184 -> :105-109 */
    rec.material = (int)t10[9];
    rec.hit = 1;
    rec.prim = 0;
    return rec;
}

/* Closest-hit linear scan with a shrinking t_max: examples/julia-raytracer.jl:242-258
 * (identical to HittableList :195-210).  Object order: spheres, then triangles, which is the
 * order of create_scene() :605-629. */
static HitRecord SUF(hit_world)(const World *w, Ray ray, REAL t_min, REAL t_max) {
    REAL closest_so_far = t_max;                                /* :244 */
    HitRecord result; memset(&result, 0, sizeof result);        /* :246 */
    for (uint32_t i = 0; i < w->n_spheres; ++i) {               /* :248 */
        HitRecord temp_rec = SUF(hit_sphere)(w->spheres5 + 5 * (size_t)i, ray, t_min, closest_so_far);
        if (temp_rec.hit) {                                     /* :250 */
            closest_so_far = temp_rec.t;                        /* :252 */
            result = temp_rec;
            result.prim = (int)i;
        }
    }
    for (uint32_t i = 0; i < w->n_triangles; ++i) {
        HitRecord temp_rec = SUF(hit_triangle)(w->triangles10 + 10 * (size_t)i, ray, t_min, closest_so_far);
        if (temp_rec.hit) {
            closest_so_far = temp_rec.t;
            result = temp_rec;
            result.prim = (int)(w->n_spheres + i);
        }
    }
    return result;
}

/* get_ray: examples/julia-raytracer.jl:298-306 (lens ignored, :299-300).
 * `llc + s*hor + t*ver - origin` parses as ((llc + s*hor) + t*ver) - origin. */
static Ray SUF(get_ray)(const World *w, REAL s, REAL t) {
    V3 origin = w->cam_origin;                                  /* :302 (offset = 0) */
    V3 d = SUF(sub)(SUF(add)(SUF(add)(w->cam_llc, SUF(scale)(w->cam_hor, s)), SUF(scale)(w->cam_ver, t)), origin);
    Ray r; r.origin = origin; r.direction = SUF(normalize)(d);  /* :303 */
    return r;
}

/* The build's counter-based RNG (DESIGN.md "RNG").  One call = one "try": three uniforms that
 * are exact multiples of 2^-21, identical in Float32 and Float64. */
typedef struct { uint32_t hA, hB; } SUF(RngKey);
#define RngKey SUF(RngKey)

static inline RngKey SUF(rng_key)(const World *w, uint32_t pixel, uint32_t sample, uint32_t bounce) {
    uint32_t sb = (sample << 8) | bounce;
    RngKey k;
    k.hA = oracle_mix32(oracle_mix32(w->sA + pixel) ^ sb);
    k.hB = oracle_mix32(oracle_mix32(w->sB ^ pixel) + sb);
    return k;
}

static inline void SUF(rng3)(RngKey k, uint32_t t, REAL *u0, REAL *u1, REAL *u2) {
    uint32_t a = oracle_mix32((k.hA + t * 0x9E3779B9u) ^ ((k.hB << 16) | (k.hB >> 16)));
    uint32_t b = oracle_mix32(a + k.hB);
    *u0 = (REAL)(a >> 11) * (REAL)(1.0 / 2097152.0);
    *u1 = (REAL)(b >> 11) * (REAL)(1.0 / 2097152.0);
    *u2 = (REAL)(((a & 0x7FFu) << 10) | (b & 0x3FFu)) * (REAL)(1.0 / 2097152.0);
}

/* random_in_unit_sphere: examples/julia-raytracer.jl:309-316.  The reference loops forever;
 * here tries are t = 1..ORACLE_MAX_TRIES (try 0 belongs to the pixel jitter), after which the
 * zero vector is returned (probability (1-pi/6)^64 ~ 2e-21 per call; the kernels do the same). */
static V3 SUF(random_in_unit_sphere)(RngKey k) {
    for (uint32_t t = 1; t <= ORACLE_MAX_TRIES; ++t) {
        REAL u0, u1, u2;
        SUF(rng3)(k, t, &u0, &u1, &u2);
        V3 p = SUF(sub)(SUF(scale)(SUF(v3)(u0, u1, u2), (REAL)2.0), SUF(v3)(1, 1, 1));  /* :311 */
        if (SUF(dot)(p, p) < (REAL)1.0) return p;                                      /* :312 */
    }
    return SUF(v3)(0, 0, 0);
}

/* reflect: examples/julia-raytracer.jl:323-325.  `2 * dot(v,n) * n` = (2*dot(v,n)) * n */
static inline V3 SUF(reflect)(V3 v, V3 n) {
    return SUF(sub)(v, SUF(scale)(n, (REAL)2 * SUF(dot)(v, n)));
}

/* ---- SPIRA_EXT_SPECTRAL (include/spira_hip.h): the path's wavelength and the uplift of an RGB triple to it.  The SPD table is
 * data shared with the product (include/spira_spd.h, generated by tools/make_spd_table.py); the arithmetic on it is restated here. */
typedef struct { REAL bR, bG, bB; } SUF(Basis);
#define Basis SUF(Basis)
static inline REAL SUF(spd_lookup)(int row, int i, REAL f) {
    REAL t0 = (REAL)spira_spd_table[row][i], t1 = (REAL)spira_spd_table[row][i + 1];
    return t0 + (t1 - t0) * f;
}
static V3 SUF(wavelength)(REAL u2, Basis *b) {       /* returns the wavelength's weighted linear-sRGB response */
    REAL x = u2 * (REAL)(SPIRA_SPD_N - 1);
    int i = (int)x;
    if (i > SPIRA_SPD_N - 2) i = SPIRA_SPD_N - 2;
    REAL f = x - (REAL)i;
    b->bR = SUF(spd_lookup)(0, i, f); b->bG = SUF(spd_lookup)(1, i, f); b->bB = SUF(spd_lookup)(2, i, f);
    return SUF(v3)(SUF(spd_lookup)(3, i, f), SUF(spd_lookup)(4, i, f), SUF(spd_lookup)(5, i, f));
}
static inline REAL SUF(uplift)(const Basis *b, V3 c) { return (c.x * b->bR + c.y * b->bG) + c.z * b->bB; }

/* Optional per-segment trace of one path (tests compare geometry bit for bit). */
typedef struct { int prim; REAL t; REAL dir[3]; } SUF(TraceSeg);
#define TraceSeg SUF(TraceSeg)

/* ray_color: examples/julia-raytracer.jl:328-367 (recursive, like the reference) */
static V3 SUF(ray_color_x)(const World *w, Ray ray, int depth, uint32_t pixel, uint32_t sample,
                           uint64_t *segments, TraceSeg *trace, const Basis *basis);
static V3 SUF(ray_color)(const World *w, Ray ray, int depth, uint32_t pixel, uint32_t sample,
                         uint64_t *segments, TraceSeg *trace) {
    return SUF(ray_color_x)(w, ray, depth, pixel, sample, segments, trace, NULL);
}
/* `basis` non-NULL = spectral mode: every RGB triple of the scene is replaced by its value at the path's wavelength. */
static V3 SUF(ray_color_x)(const World *w, Ray ray, int depth, uint32_t pixel, uint32_t sample,
                           uint64_t *segments, TraceSeg *trace, const Basis *basis) {
    if (depth <= 0) return SUF(v3)(0, 0, 0);                                      /* :330 */
    uint32_t bounce = w->max_depth - (uint32_t)depth;
    if (segments) ++*segments;
    HitRecord rec = SUF(hit_world)(w, ray, (REAL)0.001, (REAL)INFINITY);          /* :335 */
    if (trace) {
        trace[bounce].prim = rec.hit ? rec.prim : -1;
        trace[bounce].t = rec.hit ? rec.t : (REAL)0;
        trace[bounce].dir[0] = ray.direction.x; trace[bounce].dir[1] = ray.direction.y; trace[bounce].dir[2] = ray.direction.z;
    }
    if (rec.hit) {                                                                /* :337 */
        Material m = SUF(get_material)(w, rec.material);
        if (basis) {                                                              /* SPIRA_EXT_SPECTRAL */
            REAL sd = SUF(uplift)(basis, m.diffuse), se = SUF(uplift)(basis, m.emission);
            m.diffuse = SUF(v3)(sd, sd, sd); m.emission = SUF(v3)(se, se, se);
        }
        V3 emitted = m.emission;                                                  /* :339 */
        RngKey k = SUF(rng_key)(w, pixel, sample, bounce);
        if ((w->ext & SPIRA_EXT_DIELECTRIC) && m.roughness < (REAL)0.0) {         /* SPIRA_EXT_DIELECTRIC: Snell + Schlick */
            V3 n = rec.normal, d = ray.direction;
            REAL ior = -m.roughness;
            REAL cosd = SUF(dot)(d, n);
            int entering = cosd < (REAL)0.0;
            V3 nn = entering ? n : SUF(v3)(-n.x, -n.y, -n.z);
            REAL eta = entering ? (REAL)1.0 / ior : ior;
            REAL ci = entering ? -cosd : cosd;
            REAL s2 = (eta * eta) * ((REAL)1.0 - ci * ci);
            REAL r0 = ((REAL)1.0 - ior) / ((REAL)1.0 + ior); r0 = r0 * r0;
            REAL x = (REAL)1.0 - ci, x2 = x * x;
            REAL refl = r0 + ((REAL)1.0 - r0) * ((x2 * x2) * x);
            REAL u, u1, u2;
            SUF(rng3)(k, 1, &u, &u1, &u2);
            V3 dir;
            if (s2 > (REAL)1.0 || u < refl) dir = SUF(sub)(d, SUF(scale)(nn, (REAL)2 * SUF(dot)(d, nn)));
            else dir = SUF(add)(SUF(scale)(d, eta), SUF(scale)(nn, eta * ci - SQRT((REAL)1.0 - s2)));
            Ray scattered; scattered.origin = rec.position;
            scattered.direction = SUF(normalize)(dir);
            V3 in = SUF(ray_color_x)(w, scattered, depth - 1, pixel, sample, segments, trace, basis);
            return SUF(add)(emitted, SUF(mulv)(in, m.diffuse));
        }
        if (m.specular > (REAL)0.0) {                                             /* :342 */
            V3 reflected = SUF(reflect)(ray.direction, rec.normal);               /* :344 */
            if (m.roughness > (REAL)0.0)                                          /* :346 */
                reflected = SUF(add)(reflected, SUF(scale)(SUF(random_in_unit_sphere)(k), m.roughness)); /* :347 */
            Ray scattered; scattered.origin = rec.position;
            scattered.direction = SUF(normalize)(reflected);                      /* :349 */
            V3 specular_color = SUF(ray_color_x)(w, scattered, depth - 1, pixel, sample, segments, trace, basis); /* :352 */
            return SUF(add)(emitted, SUF(mulv)(SUF(scale)(specular_color, m.specular), m.diffuse));      /* :353 */
        } else {
            V3 target = SUF(add)(SUF(add)(rec.position, rec.normal), SUF(random_in_unit_sphere)(k));    /* :356 */
            Ray scattered; scattered.origin = rec.position;
            scattered.direction = SUF(normalize)(SUF(sub)(target, rec.position)); /* :357 */
            V3 in = SUF(ray_color_x)(w, scattered, depth - 1, pixel, sample, segments, trace, basis);
            return SUF(add)(emitted, SUF(mulv)(SUF(scale)(in, (REAL)0.5), m.diffuse));                  /* :360 */
        }
    }
    REAL t = (REAL)0.5 * (ray.direction.y + (REAL)1.0);                           /* :365 */
    V3 sky = SUF(add)(SUF(scale)(SUF(v3)(1.0, 1.0, 1.0), (REAL)1.0 - t),
                      SUF(scale)(SUF(v3)((REAL)0.5, (REAL)0.7, (REAL)1.0), t));   /* :366 */
    if (basis) { REAL ss = SUF(uplift)(basis, sky); return SUF(v3)(ss, ss, ss); }
    return sky;
}

/* to_acescg: examples/julia-raytracer.jl:370-384 (clamp, no gamma) */
static inline REAL SUF(aces1)(REAL x) {
    const REAL a = (REAL)2.51, b = (REAL)0.03, c = (REAL)2.43, d = (REAL)0.59, e = (REAL)0.14;
    REAL v = (x * (a * x + b)) / (x * (c * x + d) + e);                           /* :379 */
    return v < 0 ? 0 : (v > 1 ? 1 : v);
}

static inline REAL SUF(post1)(REAL x, uint32_t post) {
    switch (post) {
    case SPIRA_POST_ACES: return SUF(aces1)(x);
    case SPIRA_POST_ACES_GAMMA: return SQRT(SUF(aces1)(x));                       /* src/spira-metal-optimized.jl:1136-1142 */
    case SPIRA_POST_CLAMP_GAMMA: { REAL v = x < 0 ? 0 : (x > 1 ? 1 : x); return SQRT(v); } /* :1441-1442 */
    default: return x;
    }
}

static void SUF(world_init)(World *w, const REAL *spheres5, const REAL *materials8, const REAL *triangles10,
                            const REAL *camera12, const spira_params *p) {
    w->spheres5 = spheres5; w->materials8 = materials8; w->triangles10 = triangles10;
    w->n_spheres = p->n_spheres; w->n_materials = p->n_materials; w->n_triangles = triangles10 ? p->n_triangles : 0;
    w->cam_origin = SUF(v3)(camera12[0], camera12[1], camera12[2]);
    w->cam_llc = SUF(v3)(camera12[3], camera12[4], camera12[5]);
    w->cam_hor = SUF(v3)(camera12[6], camera12[7], camera12[8]);
    w->cam_ver = SUF(v3)(camera12[9], camera12[10], camera12[11]);
    oracle_seed_mix(p->seed, &w->sA, &w->sB);
    w->max_depth = p->max_depth;
    w->sem = p->flags & SPIRA_SEM_MASK;
    w->ext = p->flags & (SPIRA_EXT_DIELECTRIC | SPIRA_EXT_SPECTRAL);
}

/* One sample of one pixel: the body of the sample loop, examples/julia-raytracer.jl:398-401.
 * i, j are the reference's 1-based loop indices; pixel id = (j-1)*W + (i-1). */
static V3 SUF(sample_pixel)(const World *w, const spira_params *p, uint32_t i, uint32_t j, uint32_t sample,
                            uint64_t *segments, TraceSeg *trace) {
    uint32_t pixel = (j - 1) * p->width + (i - 1);
    REAL xi_u, xi_v, unused;
    SUF(rng3)(SUF(rng_key)(w, pixel, sample, 0), 0, &xi_u, &xi_v, &unused);
    REAL u = ((REAL)(i - 1) + xi_u) / (REAL)(p->width - 1);                       /* :398 */
    REAL v = ((REAL)(j - 1) + xi_v) / (REAL)(p->height - 1);                      /* :399 */
    Ray ray = SUF(get_ray)(w, u, v);                                              /* :400 */
    if (w->ext & SPIRA_EXT_SPECTRAL) {       /* the path's wavelength is the jitter try's third uniform; radiance -> linear sRGB */
        Basis basis;
        V3 resp = SUF(wavelength)(unused, &basis);
        return SUF(mulv)(resp, SUF(ray_color_x)(w, ray, (int)p->max_depth, pixel, sample, segments, trace, &basis));
    }
    return SUF(ray_color)(w, ray, (int)p->max_depth, pixel, sample, segments, trace); /* :401 */
}

/* render: examples/julia-raytracer.jl:387-421.  out_hdr / out_img are planar (3 planes of
 * rows*width), row order and tiling as documented in include/spira_hip.h. */
int SUF(oracle_render)(const REAL *spheres5, const REAL *materials8, const REAL *triangles10,
                       const REAL *camera12, const spira_params *p, REAL *out_hdr, REAL *out_img,
                       int n_threads, uint64_t *segments_out) {
    if (!spheres5 || !materials8 || !camera12 || !p) return -1;
    if ((p->flags & SPIRA_SEM_MASK) != SPIRA_SEM_A) return -5;
    World w; SUF(world_init)(&w, spheres5, materials8, triangles10, camera12, p);
    uint32_t W = p->width, H = p->height;
    uint32_t rows = p->rows ? p->rows : H;
    size_t plane = (size_t)rows * W;
    uint32_t post = p->flags & SPIRA_POST_MASK;
    uint64_t segments = 0;
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : segments)
#endif
    for (uint32_t r = 0; r < rows; ++r) {
        /* local output row r -> global output row y (row 0 = top unless ROWS_BOTTOM_UP) */
        uint32_t y = p->rows ? oracle_global_row(p, r) : r;
        /* the reference stores loop row j at hdr_data[height-j+1, i] (:408): top row = j = H */
        uint32_t j = (p->flags & SPIRA_ROWS_BOTTOM_UP) ? y + 1 : H - y;
        for (uint32_t i = 1; i <= W; ++i) {                                       /* :393 */
            V3 color = SUF(v3)(0, 0, 0);                                          /* :394 */
            for (uint32_t s = 0; s < p->spp; ++s)                                 /* :397 */
                color = SUF(add)(color, SUF(sample_pixel)(&w, p, i, j, s, &segments, NULL)); /* :401 */
            color = SUF(divs)(color, (REAL)p->spp);                               /* :405 */
            size_t o = (size_t)r * W + (i - 1);
            if (out_hdr) { out_hdr[o] = color.x; out_hdr[plane + o] = color.y; out_hdr[2 * plane + o] = color.z; } /* :408 */
            if (out_img) {                                                        /* :411 */
                out_img[o] = SUF(post1)(color.x, post);
                out_img[plane + o] = SUF(post1)(color.y, post);
                out_img[2 * plane + o] = SUF(post1)(color.z, post);
            }
        }
    }
    if (segments_out) *segments_out = segments;
    return 0;
}

/* Trace one path: per-segment primitive index (-1 = miss), t and ray direction, plus the
 * sample's radiance.  Returns the number of segments traced. */
int SUF(oracle_trace_path)(const REAL *spheres5, const REAL *materials8, const REAL *triangles10,
                           const REAL *camera12, const spira_params *p, uint32_t i, uint32_t j, uint32_t sample,
                           int *prims, REAL *ts, REAL *dirs3, REAL *radiance3) {
    World w; SUF(world_init)(&w, spheres5, materials8, triangles10, camera12, p);
    TraceSeg tr[256];
    for (uint32_t b = 0; b < 256; ++b) { tr[b].prim = -2; tr[b].t = 0; tr[b].dir[0] = tr[b].dir[1] = tr[b].dir[2] = 0; }
    uint64_t segs = 0;
    V3 c = SUF(sample_pixel)(&w, p, i, j, sample, &segs, tr);
    for (uint32_t b = 0; b < p->max_depth; ++b) {
        prims[b] = tr[b].prim; ts[b] = tr[b].t;
        dirs3[3 * b] = tr[b].dir[0]; dirs3[3 * b + 1] = tr[b].dir[1]; dirs3[3 * b + 2] = tr[b].dir[2];
    }
    radiance3[0] = c.x; radiance3[1] = c.y; radiance3[2] = c.z;
    return (int)segs;
}

/* Camera constructor: examples/julia-raytracer.jl:271-294 (and, with focus_dist = 1 and
 * Float32, src/spira-metal-optimized.jl:331-347).  out12 = origin, llc, horizontal, vertical. */
void SUF(oracle_camera)(const REAL *position, const REAL *look_at, const REAL *up, REAL fov_deg,
                        REAL aspect_ratio, REAL focus_dist, REAL *out12) {
    V3 pos = SUF(v3)(position[0], position[1], position[2]);
    V3 la = SUF(v3)(look_at[0], look_at[1], look_at[2]);
    V3 vup = SUF(v3)(up[0], up[1], up[2]);
    REAL theta = DEG2RAD(fov_deg);                                               /* :280 */
    REAL h = TAN(theta / 2);                                                     /* :281 */
    REAL viewport_height = (REAL)2.0 * h;                                        /* :282 */
    REAL viewport_width = aspect_ratio * viewport_height;                        /* :283 */
    V3 w = SUF(normalize)(SUF(sub)(pos, la));                                    /* :285 */
    V3 u = SUF(normalize)(SUF(cross)(vup, w));                                   /* :286 */
    V3 v = SUF(cross)(w, u);                                                     /* :287 */
    V3 horizontal = SUF(scale)(u, focus_dist * viewport_width);                  /* :289 */
    V3 vertical = SUF(scale)(v, focus_dist * viewport_height);                   /* :290 */
    V3 llc = SUF(sub)(SUF(sub)(SUF(sub)(pos, SUF(divs)(horizontal, 2)), SUF(divs)(vertical, 2)),
                      SUF(scale)(w, focus_dist));                                /* :291 */
    out12[0] = pos.x; out12[1] = pos.y; out12[2] = pos.z;
    out12[3] = llc.x; out12[4] = llc.y; out12[5] = llc.z;
    out12[6] = horizontal.x; out12[7] = horizontal.y; out12[8] = horizontal.z;
    out12[9] = vertical.x; out12[10] = vertical.y; out12[11] = vertical.z;
}

/* Thin wrappers for known-answer tests */
int SUF(oracle_hit_sphere)(const REAL *s5, const REAL *o3, const REAL *d3, REAL t_min, REAL t_max, REAL *t, REAL *n3) {
    Ray r; r.origin = SUF(v3)(o3[0], o3[1], o3[2]); r.direction = SUF(v3)(d3[0], d3[1], d3[2]);
    HitRecord h = SUF(hit_sphere)(s5, r, t_min, t_max);
    if (h.hit) { *t = h.t; n3[0] = h.normal.x; n3[1] = h.normal.y; n3[2] = h.normal.z; }
    return h.hit;
}
int SUF(oracle_hit_triangle)(const REAL *t10, const REAL *o3, const REAL *d3, REAL t_min, REAL t_max, REAL *t, REAL *n3) {
    Ray r; r.origin = SUF(v3)(o3[0], o3[1], o3[2]); r.direction = SUF(v3)(d3[0], d3[1], d3[2]);
    HitRecord h = SUF(hit_triangle)(t10, r, t_min, t_max);
    if (h.hit) { *t = h.t; n3[0] = h.normal.x; n3[1] = h.normal.y; n3[2] = h.normal.z; }
    return h.hit;
}
void SUF(oracle_sky)(const REAL *d3, REAL *rgb) {
    World w; memset(&w, 0, sizeof w); w.max_depth = 1;
    Ray r; r.origin = SUF(v3)(0, 0, 0); r.direction = SUF(v3)(d3[0], d3[1], d3[2]);
    V3 c = SUF(ray_color)(&w, r, 1, 0, 0, NULL, NULL);
    rgb[0] = c.x; rgb[1] = c.y; rgb[2] = c.z;
}
REAL SUF(oracle_post)(REAL x, uint32_t post) { return SUF(post1)(x, post); }
void SUF(oracle_rng_try)(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t bounce, uint32_t t, REAL *u3) {
    World w; memset(&w, 0, sizeof w); oracle_seed_mix(seed, &w.sA, &w.sB);
    SUF(rng3)(SUF(rng_key)(&w, pixel, sample, bounce), t, &u3[0], &u3[1], &u3[2]);
}

/* ======================================================================================== */
/* Variant "CPU": trace_ray of render_with_cpu, src/spira-metal-optimized.jl:1346-1450.      */
/* (What the reference's render() actually executes on an AMD machine today, :1469-1473.)    */
/* Third-party arithmetic: StaticArrays (Project.toml:21 `StaticArrays = "1"`, unvendored):  */
/* normalize(a) = inv(norm(a)) * a, norm = sqrt(sum(abs2)); restated below.  The reference    */
/* draws from Julia's unseeded global RNG: the draws here are the build's counter-based RNG  */
/* (try 0 = pixel jitter, try 1 = lobe choice, try 2 = rand(Vec3)).  PARITY UNPINNED.        */
static inline V3 SUF(sa_normalize)(V3 a) {                     /* StaticArrays: inv(norm(a)) * a */
    REAL inv = (REAL)1.0 / SQRT(SUF(dot)(a, a));
    return SUF(v3)(inv * a.x, inv * a.y, inv * a.z);
}

static V3 SUF(trace_ray_cpu)(const World *w, Ray ray, int depth, uint32_t pixel, uint32_t sample, uint64_t *segments,
                             TraceSeg *trace) {
    if (depth <= 0) return SUF(v3)(0, 0, 0);                                        /* :1352-1354 */
    uint32_t bounce = w->max_depth - (uint32_t)depth;
    if (segments) ++*segments;
    int hit_anything = 0, prim = -1, mat1 = 0;
    REAL closest_t = (REAL)1e20f;                                                   /* INF = Float32(1e20), :287,:1357 */
    V3 hit_normal = SUF(v3)(0, 0, 0);
    for (uint32_t s = 0; s < w->n_spheres; ++s) {                                   /* :1362 */
        const REAL *s5 = w->spheres5 + 5 * (size_t)s;
        V3 center = SUF(v3)(s5[0], s5[1], s5[2]);
        V3 oc = SUF(sub)(ray.origin, center);                                       /* :1363 */
        REAL a = (REAL)1.0;                                                         /* :1364 */
        REAL half_b = SUF(dot)(oc, ray.direction);                                  /* :1365 */
        REAL c = SUF(dot)(oc, oc) - s5[3] * s5[3];                                  /* :1366 */
        REAL discriminant = half_b * half_b - a * c;                                /* :1367 */
        if (discriminant > 0) {                                                     /* :1369 */
            REAL sqrtd = SQRT(discriminant);
            REAL root = (-half_b - sqrtd) / a;                                      /* :1373 */
            if (root < (REAL)0.001f) root = (-half_b + sqrtd) / a;                  /* :1374-1376 */
            if (root > (REAL)0.001f && root < closest_t) {                          /* :1378 */
                closest_t = root; hit_anything = 1; prim = (int)s;
                hit_normal = SUF(sa_normalize)(SUF(sub)(SUF(add)(ray.origin, SUF(scale)(ray.direction, closest_t)), center)); /* :1381 */
                mat1 = (int)s5[4];
            }
        }
    }
    if (trace) {
        trace[bounce].prim = prim; trace[bounce].t = hit_anything ? closest_t : (REAL)0;
        trace[bounce].dir[0] = ray.direction.x; trace[bounce].dir[1] = ray.direction.y; trace[bounce].dir[2] = ray.direction.z;
    }
    if (hit_anything) {
        V3 hit_point = SUF(add)(ray.origin, SUF(scale)(ray.direction, closest_t));  /* :1388 */
        Material m = SUF(get_material)(w, mat1);
        if (m.emission.x > 0 || m.emission.y > 0 || m.emission.z > 0) return m.emission;   /* :1392-1394: the path ends */
        RngKey k = SUF(rng_key)(w, pixel, sample, bounce);
        REAL lobe, r0, r1, r2, unused0, unused1;
        SUF(rng3)(k, 1, &lobe, &unused0, &unused1);
        SUF(rng3)(k, 2, &r0, &r1, &r2);
        V3 rv = SUF(sub)(SUF(v3)(r0, r1, r2), SUF(v3)((REAL)0.5, (REAL)0.5, (REAL)0.5));   /* rand(Vec3) - 0.5f0 */
        Ray scattered; scattered.origin = hit_point;
        if (lobe > m.specular) {                                                    /* rand(Float32) > material.metallic, :1397 */
            V3 target = SUF(add)(SUF(add)(hit_point, hit_normal), SUF(sa_normalize)(rv));   /* :1399 */
            scattered.direction = SUF(sa_normalize)(SUF(sa_normalize)(SUF(sub)(target, hit_point)));   /* :1400 + Ray ctor :297 */
            V3 in = SUF(trace_ray_cpu)(w, scattered, depth - 1, pixel, sample, segments, trace);
            return SUF(scale)(SUF(mulv)(m.diffuse, in), (REAL)0.5);                 /* albedo .* L .* 0.5, :1401 */
        } else {
            V3 reflected = SUF(sub)(ray.direction, SUF(scale)(hit_normal, (REAL)2.0 * SUF(dot)(ray.direction, hit_normal)));  /* :1404 */
            scattered.direction = SUF(sa_normalize)(SUF(sa_normalize)(SUF(add)(reflected, SUF(scale)(rv, m.roughness))));   /* :1405 + ctor */
            V3 in = SUF(trace_ray_cpu)(w, scattered, depth - 1, pixel, sample, segments, trace);
            return SUF(mulv)(m.diffuse, in);                                        /* :1406 */
        }
    }
    REAL t = (REAL)0.5 * (ray.direction.y + (REAL)1.0);                             /* :1411 */
    return SUF(add)(SUF(scale)(SUF(v3)(1, 1, 1), (REAL)1.0 - t), SUF(scale)(SUF(v3)((REAL)0.5, (REAL)0.7, (REAL)1.0), t));   /* :1412 */
}

static V3 SUF(sample_pixel_cpu)(const World *w, const spira_params *p, uint32_t i, uint32_t j, uint32_t sample,
                                uint64_t *segments, TraceSeg *trace) {
    uint32_t pixel = (j - 1) * p->width + (i - 1);
    REAL xi_u, xi_v, unused;
    SUF(rng3)(SUF(rng_key)(w, pixel, sample, 0), 0, &xi_u, &xi_v, &unused);
    REAL u = ((REAL)(i - 1) + xi_u) / (REAL)(p->width - 1);                         /* :1428 */
    REAL v = ((REAL)(j - 1) + xi_v) / (REAL)(p->height - 1);                        /* :1429 */
    V3 dir = SUF(sub)(SUF(add)(SUF(add)(w->cam_llc, SUF(scale)(w->cam_hor, u)), SUF(scale)(w->cam_ver, v)), w->cam_origin);  /* :1431 */
    Ray ray; ray.origin = w->cam_origin; ray.direction = SUF(sa_normalize)(dir);    /* Ray ctor, :1432 */
    return SUF(trace_ray_cpu)(w, ray, (int)p->max_depth, pixel, sample, segments, trace);   /* :1434 */
}

/* ======================================================================================== */
/* Variant "METAL": path_trace of src/spira_path_trace_kernel.metal:140-269 (+ helpers       */
/* :52-136).  The file is HTML-escaped and nothing loads it; this restates what  */
/* it says, in IEEE arithmetic in the written order (Metal's fast-math is unknowable).       */
/* RNG: the kernel's own LCG (:52-58), state per pixel carried from sample to sample (:268); */
/* the initial states, which the reference's host would draw at random, come from the seed.  */
/* sin/cos: a fixed polynomial evaluated in plain arithmetic (the kernels use the same one). */
static inline uint32_t SUF(lcg_uniform)(uint32_t *state, REAL *u) {
    *state = *state * 1664525u + 1013904223u;                                       /* :55-56 */
    *u = (REAL)(*state & 0x00FFFFFFu) / (REAL)0x01000000;                           /* :57 */
    return *state;
}

/* sin(2*pi*r), cos(2*pi*r) for r in [0,1): nearest quarter turn k, remainder f in [-1/2,1/2] quarter turns,
 * theta = f*(pi/2) in [-pi/4, pi/4], Taylor polynomials in theta^2 (Horner), then rotate by k quadrants. */
static inline void SUF(sincos_turn)(REAL r, REAL *sn, REAL *cs) {
    REAL t = r * (REAL)4.0;
    int k = (int)(t + (REAL)0.5);
    REAL f = t - (REAL)k;
    REAL th = f * (REAL)1.57079632679489661923;
    REAL x2 = th * th;
    REAL ps = 0, pc = 0;
    static const double SC[8] = { -1.0 / 6, 1.0 / 120, -1.0 / 5040, 1.0 / 362880, -1.0 / 39916800, 1.0 / 6227020800.0,
                                  -1.0 / 1307674368000.0, 1.0 / 355687428096000.0 };
    static const double CC[9] = { -1.0 / 2, 1.0 / 24, -1.0 / 720, 1.0 / 40320, -1.0 / 3628800, 1.0 / 479001600, -1.0 / 87178291200.0,
                                  1.0 / 20922789888000.0, -1.0 / 6402373705728000.0 };
    const int ns = sizeof(REAL) == 4 ? 4 : 8, nc = sizeof(REAL) == 4 ? 5 : 9;
    for (int i = ns - 1; i >= 0; --i) ps = (ps + (REAL)SC[i]) * x2;
    for (int i = nc - 1; i >= 0; --i) pc = (pc + (REAL)CC[i]) * x2;
    REAL s0 = th + th * ps, c0 = (REAL)1.0 + pc;
    switch (k & 3) {
    case 0: *sn = s0; *cs = c0; break;
    case 1: *sn = c0; *cs = -s0; break;
    case 2: *sn = -s0; *cs = -c0; break;
    default: *sn = -c0; *cs = s0; break;
    }
}

static V3 SUF(sample_pixel_metal)(const World *w, const spira_params *p, uint32_t x, uint32_t y, uint32_t *rng_state,
                                  uint64_t *segments, TraceSeg *trace) {
    const REAL EPSILON = (REAL)0.0001f, INF_ = (REAL)1e20f;                         /* :6-7 */
    uint32_t st = *rng_state;                                                       /* :155 */
    REAL xi;
    SUF(lcg_uniform)(&st, &xi);
    REAL u_j = ((REAL)x + xi) / (REAL)p->width;                                     /* :161 */
    SUF(lcg_uniform)(&st, &xi);
    REAL v_j = ((REAL)y + xi) / (REAL)p->height;                                    /* :162 */
    V3 o = w->cam_origin;                                                           /* :166 */
    V3 d = SUF(normalize)(SUF(sub)(SUF(add)(SUF(add)(w->cam_llc, SUF(scale)(w->cam_hor, u_j)), SUF(scale)(w->cam_ver, v_j)), o)); /* :167-170 */
    V3 acc = SUF(v3)(0, 0, 0), thr = SUF(v3)(1, 1, 1);                              /* :173-174 */
    for (uint32_t depth = 0; depth < p->max_depth; ++depth) {                       /* :176 */
        if (segments) ++*segments;
        REAL closest_t = INF_; int hit = -1; V3 n = SUF(v3)(0, 0, 0);               /* :177-179 */
        for (uint32_t s = 0; s < w->n_spheres; ++s) {                               /* :182 */
            const REAL *s5 = w->spheres5 + 5 * (size_t)s;                           /* intersect_sphere, :109-136 */
            V3 center = SUF(v3)(s5[0], s5[1], s5[2]);
            V3 oc = SUF(sub)(o, center);                                            /* :113 */
            REAL a = SUF(dot)(d, d);                                                /* :114 */
            REAL half_b = SUF(dot)(oc, d);                                          /* :115 */
            REAL c = SUF(dot)(oc, oc) - s5[3] * s5[3];                              /* :116 */
            REAL disc = half_b * half_b - a * c;                                    /* :117 */
            REAL t = INF_; V3 nn = SUF(v3)(0, 0, 0);
            if (disc > (REAL)0.0) {                                                 /* :119 */
                REAL sq = SQRT(disc);
                REAL root = (-half_b - sq) / a;                                     /* :120 */
                if (!(root > EPSILON)) root = (-half_b + sq) / a;                   /* :121,:127 */
                if (root > EPSILON) {                                               /* :128 */
                    t = root;
                    nn = SUF(normalize)(SUF(sub)(SUF(add)(o, SUF(scale)(d, t)), center));   /* :123/:130 */
                }
            }
            if (t < closest_t) { closest_t = t; hit = (int)s; n = nn; }            /* :184-188: strict <, the earlier sphere keeps a tie */
        }
        if (trace) {
            trace[depth].prim = hit; trace[depth].t = hit >= 0 ? closest_t : (REAL)0;
            trace[depth].dir[0] = d.x; trace[depth].dir[1] = d.y; trace[depth].dir[2] = d.z;
        }
        if (hit == -1) {                                                            /* :192 */
            REAL ts = (REAL)0.5 * (d.y + (REAL)1.0);                                /* :194 */
            V3 sky = SUF(add)(SUF(scale)(SUF(v3)(1, 1, 1), (REAL)1.0 - ts), SUF(scale)(SUF(v3)((REAL)0.5, (REAL)0.7, (REAL)1.0), ts)); /* :195 */
            acc = SUF(add)(acc, SUF(mulv)(thr, sky));                               /* :196 */
            break;
        }
        Material m = SUF(get_material)(w, (int)w->spheres5[5 * (size_t)hit + 4]);   /* :202 */
        V3 hit_point = SUF(add)(o, SUF(scale)(d, closest_t));                       /* :203 */
        if (SUF(dot)(d, n) > (REAL)0.0) n = SUF(v3)(-n.x, -n.y, -n.z);              /* :207-209 */
        acc = SUF(add)(acc, SUF(mulv)(thr, m.emission));                            /* :212 */
        V3 scatter_origin = SUF(add)(hit_point, SUF(scale)(n, EPSILON));            /* :215 */
        V3 nd;
        SUF(lcg_uniform)(&st, &xi);
        if (xi < m.specular) {                                                      /* random_uniform < metallic, :219 */
            nd = SUF(sub)(d, SUF(scale)(n, (REAL)2.0 * SUF(dot)(d, n)));            /* reflect, :96-98, :220 */
            if (m.roughness > (REAL)0.0) {                                          /* :221 */
                V3 pv;
                for (;;) {                                                          /* random_unit_vector, :61-70 */
                    REAL a0, a1, a2;
                    SUF(lcg_uniform)(&st, &a0); SUF(lcg_uniform)(&st, &a1); SUF(lcg_uniform)(&st, &a2);
                    pv = SUF(v3)(a0 * (REAL)2.0 - (REAL)1.0, a1 * (REAL)2.0 - (REAL)1.0, a2 * (REAL)2.0 - (REAL)1.0);
                    if (SUF(dot)(pv, pv) < (REAL)1.0) break;
                }
                nd = SUF(normalize)(SUF(add)(nd, SUF(scale)(SUF(normalize)(pv), m.roughness)));   /* :222 */
            }
        } else {                                                                    /* cosine hemisphere, :73-93 */
            REAL r1, r2, sn, cs;
            SUF(lcg_uniform)(&st, &r1); SUF(lcg_uniform)(&st, &r2);
            SUF(sincos_turn)(r1, &sn, &cs);                                         /* phi = 2*pi*r1, :77 */
            REAL sr = SQRT(r2);
            REAL hx = cs * sr, hy = sn * sr;                                        /* :83-84 */
            REAL zz = (REAL)1.0 - hx * hx - hy * hy;
            REAL hz = SQRT(zz > (REAL)0.0 ? zz : (REAL)0.0);                        /* :85 */
            V3 helper = FABS(n.x) > (REAL)0.1 ? SUF(v3)(0, 1, 0) : SUF(v3)(1, 0, 0);   /* :89 */
            V3 ua = SUF(normalize)(SUF(cross)(helper, n));                          /* :90 */
            V3 va = SUF(cross)(n, ua);                                              /* :91 */
            nd = SUF(normalize)(SUF(add)(SUF(add)(SUF(scale)(ua, hx), SUF(scale)(va, hy)), SUF(scale)(n, hz)));   /* :93 */
        }
        o = scatter_origin; d = nd;                                                 /* :230-231 */
        thr = SUF(mulv)(thr, m.diffuse);                                            /* :232 */
        if (depth > 3) {                                                            /* Russian roulette, :236-243 */
            REAL pc = thr.x > thr.y ? thr.x : thr.y; pc = pc > thr.z ? pc : thr.z;
            pc = pc < (REAL)0.95f ? pc : (REAL)0.95f;
            SUF(lcg_uniform)(&st, &xi);
            if (xi > pc) break;
            thr = SUF(divs)(thr, pc);
        }
        { REAL mx = thr.x > thr.y ? thr.x : thr.y; mx = mx > thr.z ? mx : thr.z; if (mx < (REAL)0.01f) break; }   /* :246 */
    }
    *rng_state = st;                                                                /* :268 */
    return acc;
}

/* initial per-pixel LCG state (the reference's host would fill rng_states with random UInt32s, cf.
 * src/spira-metal-optimized.jl:1258): derived from the seed */
static inline uint32_t SUF(metal_state0)(const World *w, uint32_t pixel) { return oracle_mix32(oracle_mix32(w->sA + pixel) ^ w->sB); }

/* Render with one of the secondary variants (flags & SPIRA_SEM_MASK = SPIRA_SEM_CPU | SPIRA_SEM_METAL). */
int SUF(oracle_render_variant)(const REAL *spheres5, const REAL *materials8, const REAL *camera12, const spira_params *p,
                               REAL *out_hdr, REAL *out_img, int n_threads, uint64_t *segments_out) {
    if (!spheres5 || !materials8 || !camera12 || !p) return -1;
    const uint32_t sem = p->flags & SPIRA_SEM_MASK;
    if (sem != SPIRA_SEM_CPU && sem != SPIRA_SEM_METAL) return -5;
    World w; SUF(world_init)(&w, spheres5, materials8, NULL, camera12, p);
    uint32_t W = p->width, H = p->height;
    uint32_t rows = p->rows ? p->rows : H;
    size_t plane = (size_t)rows * W;
    uint32_t post = p->flags & SPIRA_POST_MASK;
    uint64_t segments = 0;
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : segments)
#endif
    for (uint32_t r = 0; r < rows; ++r) {
        uint32_t y = p->rows ? oracle_global_row(p, r) : r;
        uint32_t j = (p->flags & SPIRA_ROWS_BOTTOM_UP) ? y + 1 : H - y;      /* j = 1 is v = 0 (gid.y = 0 in the .metal kernel) */
        for (uint32_t i = 1; i <= W; ++i) {
            V3 color = SUF(v3)(0, 0, 0);
            uint32_t pixel = (j - 1) * W + (i - 1);
            uint32_t st = SUF(metal_state0)(&w, pixel);
            for (uint32_t s = 0; s < p->spp; ++s) {
                V3 c = sem == SPIRA_SEM_CPU ? SUF(sample_pixel_cpu)(&w, p, i, j, s, &segments, NULL)
                                            : SUF(sample_pixel_metal)(&w, p, i - 1, j - 1, &st, &segments, NULL);
                color = SUF(add)(color, c);                                   /* :1434 / .metal :264 */
            }
            color = SUF(divs)(color, (REAL)p->spp);                           /* :1438 */
            size_t o = (size_t)r * W + (i - 1);
            if (out_hdr) { out_hdr[o] = color.x; out_hdr[plane + o] = color.y; out_hdr[2 * plane + o] = color.z; }
            if (out_img) {
                out_img[o] = SUF(post1)(color.x, post); out_img[plane + o] = SUF(post1)(color.y, post);
                out_img[2 * plane + o] = SUF(post1)(color.z, post);
            }
        }
    }
    if (segments_out) *segments_out = segments;
    return 0;
}

/* Trace one path of a secondary variant (METAL: the `sample`-th path of the pixel, replaying the earlier ones
 * to reach its RNG state).  Same outputs as oracle_trace_path. */
int SUF(oracle_trace_path_variant)(const REAL *spheres5, const REAL *materials8, const REAL *camera12, const spira_params *p,
                                   uint32_t i, uint32_t j, uint32_t sample, int *prims, REAL *ts, REAL *dirs3, REAL *radiance3) {
    const uint32_t sem = p->flags & SPIRA_SEM_MASK;
    World w; SUF(world_init)(&w, spheres5, materials8, NULL, camera12, p);
    TraceSeg tr[256];
    uint64_t segs = 0;
    V3 c;
    if (sem == SPIRA_SEM_CPU) {
        for (uint32_t b = 0; b < 256; ++b) { tr[b].prim = -2; tr[b].t = 0; tr[b].dir[0] = tr[b].dir[1] = tr[b].dir[2] = 0; }
        c = SUF(sample_pixel_cpu)(&w, p, i, j, sample, &segs, tr);
    } else {
        uint32_t st = SUF(metal_state0)(&w, (j - 1) * p->width + (i - 1));
        for (uint32_t s = 0; s <= sample; ++s) {
            for (uint32_t b = 0; b < 256; ++b) { tr[b].prim = -2; tr[b].t = 0; tr[b].dir[0] = tr[b].dir[1] = tr[b].dir[2] = 0; }
            segs = 0;
            c = SUF(sample_pixel_metal)(&w, p, i - 1, j - 1, &st, &segs, tr);
        }
    }
    for (uint32_t b = 0; b < p->max_depth; ++b) {
        prims[b] = tr[b].prim; ts[b] = tr[b].t;
        dirs3[3 * b] = tr[b].dir[0]; dirs3[3 * b + 1] = tr[b].dir[1]; dirs3[3 * b + 2] = tr[b].dir[2];
    }
    radiance3[0] = c.x; radiance3[1] = c.y; radiance3[2] = c.z;
    return (int)segs;
}

/* ------------------------------------------------------------------------------------------------------------------------------------
 * SPIRA_SEM_HYBRID — the estimator of render_hybrid_gpu, src/spira-metal-optimized.jl:1228-1343, AS WRITTEN (what `render()` executes on a
 * machine with a Metal / CUDA backend).  The whole image advances in lock step, one sample after the other, one depth after the other:
 *   K3  gpu_generate_initial_sample_rays_kernel!  :610-697   per-pixel xorshift32 state (:412-426), jitter of +-0.25 pixel
 *   K4  gpu_ray_sphere_intersection               :700-799   half-b test, t = t1 > 0.001 ? t1 : t2, strict t < closest: the EARLIER sphere keeps a tie
 *   --  `if sum(hit_results[:, 1]) == 0 break`    :1303      no ray of the IMAGE hit anything: the sample ends (and adds nothing if depth < max_depth)
 *   K5  gpu_scatter_kernel!                       :862-989   a ray that missed carries on from (0, 0, 0) in its old direction; a mirror reflection is not normalised
 *   --  `contribution .*= 0.5f0`                  :1328      for every ray, hit or not
 *   K6  gpu_shade_kernel!                         :1071-1105 only at depth == max_depth: albedo * contribution + emission of the LAST hit, or the sky along the NEW direction
 *   K7  gpu_tone_map_kernel!                      :1128-1144 ACES + sqrt, per sample
 *   K8  average_image_kernel! + the row flip      :1055-1068, :1157-1190
 * The reference documents none of this as intended (:1331-1338 call it a "simple model"); it is restated, not repaired.  Its per-pixel states are
 * rand(UInt32) (:1258): here derived from the seed.  The unbounded rejection loop of K5 is bounded at 64 tries like everywhere else.
 * Whole images only (the lock step needs every pixel): params->rows must be 0.  out_hdr = out_img = the reference's image (already tone-mapped per sample). */
static inline REAL SUF(xs_uniform)(uint32_t s) { return (REAL)oracle_xorshift_uniform(s); }      /* Float32(state / typemax(UInt32)), :420-426 */
static inline REAL SUF(aces_sqrt)(REAL x) {                                                        /* :1133-1143 */
    const REAL a = (REAL)2.51f, b = (REAL)0.03f, c = (REAL)2.43f, d = (REAL)0.59f, e = (REAL)0.14f;
    REAL r = (x * (a * x + b)) / (x * (c * x + d) + e);
    r = r < (REAL)0 ? (REAL)0 : (r > (REAL)1 ? (REAL)1 : r);
    return SQRT(r);
}
int SUF(oracle_render_hybrid)(const REAL *spheres5, const REAL *materials8, const REAL *camera12, const spira_params *p,
                              REAL *out_hdr, REAL *out_img, int n_threads, uint64_t *segments_out) {
    if (!materials8 || !camera12 || !p || (!spheres5 && p->n_spheres)) return -1;
    if ((p->flags & SPIRA_SEM_MASK) != SPIRA_SEM_HYBRID) return -5;
    if (p->rows != 0) return -5;
    World w; SUF(world_init)(&w, spheres5, materials8, NULL, camera12, p);
    const uint32_t W = p->width, H = p->height;
    const size_t P = (size_t)W * H;
    typedef struct { V3 o, d, pt, n; uint32_t mat, rng; V3 sum; } Px;
    Px *px = (Px *)malloc(P * sizeof(Px));
    if (!px) return -3;
    uint64_t segments = 0;
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#endif
    for (size_t k = 0; k < P; ++k) { px[k].rng = SUF(metal_state0)(&w, (uint32_t)k); px[k].sum = SUF(v3)(0, 0, 0); }      /* rand(UInt32, width*height), :1258 */
    for (uint32_t sample = 1; sample <= p->spp; ++sample) {                                          /* :1275 */
        /* ---- K3 */
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
        for (size_t k = 0; k < P; ++k) {
            uint32_t st = oracle_xorshift32(px[k].rng + sample);                                     /* :632 */
            st = oracle_xorshift32(st); const REAL rand1 = SUF(xs_uniform)(st);                      /* :635-636 */
            st = oracle_xorshift32(st); const REAL rand2 = SUF(xs_uniform)(st);                      /* :637-638 */
            px[k].rng = st;                                                                          /* :641 */
            const uint32_t col = (uint32_t)(k % W), row = (uint32_t)(k / W);                         /* :651-652 */
            const REAL u_center = (REAL)col / (REAL)(W - 1), v_center = (REAL)row / (REAL)(H - 1);   /* :656-657 */
            const REAL pw = (REAL)1.0 / (REAL)(W - 1), ph = (REAL)1.0 / (REAL)(H - 1);               /* :669-670 */
            const REAL u = u_center + (rand1 - (REAL)0.5) * (REAL)0.5 * pw;                          /* :672 (jitter_scale_factor = 0.5f0, :1286) */
            const REAL v = v_center + (rand2 - (REAL)0.5) * (REAL)0.5 * ph;                          /* :673 */
            const REAL dx = w.cam_llc.x + u * w.cam_hor.x + v * w.cam_ver.x - w.cam_origin.x;        /* :676 */
            const REAL dy = w.cam_llc.y + u * w.cam_hor.y + v * w.cam_ver.y - w.cam_origin.y;
            const REAL dz = w.cam_llc.z + u * w.cam_hor.z + v * w.cam_ver.z - w.cam_origin.z;
            const REAL len_sq = dx * dx + dy * dy + dz * dz;                                         /* :681 */
            const REAL inv_len = len_sq > (REAL)0 ? SQRT((REAL)1.0 / len_sq) : (REAL)0;              /* :682 */
            px[k].o = w.cam_origin;
            px[k].d = SUF(v3)(dx * inv_len, dy * inv_len, dz * inv_len);                             /* :689-691 */
        }
        REAL contribution = (REAL)1.0;                                                               /* Metal.ones, :1294 */
        for (uint32_t depth = 1; depth <= p->max_depth; ++depth) {                                   /* :1297 */
            /* ---- K4 */
            int any_hit = 0;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) reduction(| : any_hit)
#endif
            for (size_t k = 0; k < P; ++k) {
                const V3 o = px[k].o, d = px[k].d;
                uint32_t mat = 0; REAL best = (REAL)1e20f;                                           /* :713-715 */
                V3 pt = SUF(v3)(0, 0, 0), nn = SUF(v3)(0, 0, 0);                                     /* :718-719 */
                for (uint32_t s = 0; s < w.n_spheres; ++s) {                                         /* :724 */
                    const REAL *s5 = w.spheres5 + 5 * (size_t)s;
                    const REAL ocx = o.x - s5[0], ocy = o.y - s5[1], ocz = o.z - s5[2];              /* :735-737 */
                    const REAL a = d.x * d.x + d.y * d.y + d.z * d.z;                                /* :740 */
                    const REAL half_b = ocx * d.x + ocy * d.y + ocz * d.z;                           /* :741-743 */
                    const REAL c = ocx * ocx + ocy * ocy + ocz * ocz - s5[3] * s5[3];                /* :744 */
                    const REAL disc = half_b * half_b - a * c;                                       /* :747 */
                    if (!(disc > (REAL)0)) continue;                                                 /* :750 */
                    const REAL sq = SQRT(disc);                                                      /* :759 */
                    const REAL t1 = (-half_b - sq) / a, t2 = (-half_b + sq) / a;                     /* :760-761 */
                    const REAL t = t1 > (REAL)0.001f ? t1 : t2;                                      /* :764 */
                    if (t <= (REAL)0.001f || t >= best) continue;                                    /* :767 */
                    best = t; mat = (uint32_t)s5[4];                                                 /* :772-774 */
                    pt = SUF(v3)(o.x + t * d.x, o.y + t * d.y, o.z + t * d.z);                       /* :777-779 */
                    const REAL nx = pt.x - s5[0], ny = pt.y - s5[1], nz = pt.z - s5[2];              /* :782-784 */
                    const REAL il = (REAL)1.0 / SQRT(nx * nx + ny * ny + nz * nz);                   /* :787 */
                    nn = SUF(v3)(nx * il, ny * il, nz * il);                                         /* :788-790 */
                }
                px[k].mat = mat; px[k].pt = pt; px[k].n = nn;
                if (mat) any_hit = 1;
            }
            segments += P;
            if (!any_hit) break;                                                                     /* :1303-1310 */
            /* ---- K5 */
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
            for (size_t k = 0; k < P; ++k) {
                Px *q = &px[k];
                q->o = q->pt;                                                                        /* :883-885: (0, 0, 0) for a ray that missed */
                if (!q->mat) continue;                                                               /* :888 */
                const REAL *m = w.materials8 + 8 * (size_t)(q->mat - 1);                             /* :890-891 */
                const REAL metallic = m[6], roughness = m[7];
                const REAL nx = q->n.x, ny = q->n.y, nz = q->n.z, dx = q->d.x, dy = q->d.y, dz = q->d.z;
                const REAL dot_prod = dx * nx + dy * ny + dz * nz;                                   /* :904 */
                if (metallic > (REAL)0) {                                                            /* :907 */
                    REAL rx = dx - (REAL)2.0 * dot_prod * nx, ry = dy - (REAL)2.0 * dot_prod * ny, rz = dz - (REAL)2.0 * dot_prod * nz;   /* :908-910 */
                    if (roughness > (REAL)0) {                                                       /* :912 */
                        uint32_t st = q->rng;
                        st = oracle_xorshift32(st); REAL r1 = SUF(xs_uniform)(st) - (REAL)0.5;       /* :914-921 */
                        st = oracle_xorshift32(st); REAL r2 = SUF(xs_uniform)(st) - (REAL)0.5;
                        st = oracle_xorshift32(st); REAL r3 = SUF(xs_uniform)(st) - (REAL)0.5;
                        q->rng = st;
                        const REAL nl = SQRT(r1 * r1 + r2 * r2 + r3 * r3);                           /* :927 */
                        if (nl > (REAL)1e-5f) { const REAL il = (REAL)1.0 / nl; r1 *= il; r2 *= il; r3 *= il; }   /* :928-933 */
                        rx += roughness * r1; ry += roughness * r2; rz += roughness * r3;            /* :935-937 */
                        const REAL il = (REAL)1.0 / SQRT(rx * rx + ry * ry + rz * rz);               /* :939 */
                        rx *= il; ry *= il; rz *= il;
                    }
                    q->d = SUF(v3)(rx, ry, rz);                                                      /* :944-946 */
                } else {                                                                             /* :947 */
                    uint32_t st = q->rng;
                    REAL lx = 0, ly = 0, lz = 0;
                    for (int tries = 0; tries < 64; ++tries) {                                       /* while true, :950-963 */
                        st = oracle_xorshift32(st); const REAL r1 = SUF(xs_uniform)(st);
                        st = oracle_xorshift32(st); const REAL r2 = SUF(xs_uniform)(st);
                        st = oracle_xorshift32(st); const REAL r3 = SUF(xs_uniform)(st);
                        lx = r1 * (REAL)2.0 - (REAL)1.0; ly = r2 * (REAL)2.0 - (REAL)1.0; lz = r3 * (REAL)2.0 - (REAL)1.0;
                        if (lx * lx + ly * ly + lz * lz <= (REAL)1.0) break;
                        if (tries == 63) { lx = ly = lz = 0; }
                    }
                    q->rng = st;                                                                     /* :964 */
                    const REAL ddx = nx + lx, ddy = ny + ly, ddz = nz + lz;                          /* :966-968 */
                    const REAL ls = ddx * ddx + ddy * ddy + ddz * ddz;                               /* :970 */
                    if (ls < (REAL)1e-5f) q->d = SUF(v3)(nx, ny, nz);                                /* :971-974 */
                    else { const REAL il = (REAL)1.0 / SQRT(ls); q->d = SUF(v3)(ddx * il, ddy * il, ddz * il); }   /* :976-979 */
                }
            }
            contribution *= (REAL)0.5;                                                               /* :1328 */
            if (depth == p->max_depth) {                                                             /* :1331 */
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
                for (size_t k = 0; k < P; ++k) {
                    REAL cr, cg, cb;
                    if (px[k].mat) {                                                                 /* K6, :1084-1096 */
                        const REAL *m = w.materials8 + 8 * (size_t)(px[k].mat - 1);
                        cr = m[0] * contribution + m[3]; cg = m[1] * contribution + m[4]; cb = m[2] * contribution + m[5];
                    } else {                                                                         /* :1097-1102 */
                        const REAL t = (REAL)0.5 * (px[k].d.y + (REAL)1.0);
                        cr = ((REAL)1.0 - t) + t * (REAL)0.5; cg = ((REAL)1.0 - t) + t * (REAL)0.7f; cb = ((REAL)1.0 - t) + t * (REAL)1.0;
                    }
                    px[k].sum.x += SUF(aces_sqrt)(cr); px[k].sum.y += SUF(aces_sqrt)(cg); px[k].sum.z += SUF(aces_sqrt)(cb);   /* K7 + :1334 */
                }
            }
        }
    }
    for (uint32_t y = 0; y < H; ++y) {                                                               /* K8 and the row flip, :1055-1068, :1177-1188 */
        const uint32_t row = (p->flags & SPIRA_ROWS_BOTTOM_UP) ? y : H - 1 - y;
        for (uint32_t x = 0; x < W; ++x) {
            const Px *q = &px[(size_t)row * W + x];
            const REAL r = q->sum.x / (REAL)p->spp, g = q->sum.y / (REAL)p->spp, b = q->sum.z / (REAL)p->spp;
            const size_t o = (size_t)y * W + x;
            if (out_hdr) { out_hdr[o] = r; out_hdr[P + o] = g; out_hdr[2 * P + o] = b; }
            if (out_img) { out_img[o] = r; out_img[P + o] = g; out_img[2 * P + o] = b; }
        }
    }
    free(px);
    if (segments_out) *segments_out = segments;
    return 0;
}

void SUF(oracle_sincos_turn)(REAL r, REAL *sc2) { SUF(sincos_turn)(r, &sc2[0], &sc2[1]); }

#undef V3
#undef Material
#undef HitRecord
#undef Ray
#undef World
#undef RngKey
#undef TraceSeg
#undef Basis
