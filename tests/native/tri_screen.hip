// GPU check of the contract behind the Float64 walk's Float32 triangle screen (spira_device.h: tri_screen_f32, bvh8_step<double>):
//   class 0 ("rejects")   : the scan's own Float64 test (triangle_test<double>, examples/julia-raytracer.jl:145-187) rejects the triangle too;
//   class 2 ("hit, t_hi") : the Float64 test accepts it whenever no closer hit is known, and its distance is <= the bound the walk prunes with.
// Every sample builds a mesh frame (centre up to 1e3 units from the origin, power-of-two scale 2^-8 .. 2^8), a triangle inside the frame's
// unit box (edges 1e-5 .. 0.5 of the box, one in four a sliver), a ray that starts in the box and is aimed at a point chosen against the screen:
// inside the triangle, exactly on an edge or a vertex, outside by 1e-9 .. 1e-2 of an edge, at a grazing angle one time in four; hit distances from
// 1e-3 to the box size; the closest hit so far infinite, just behind, just in front of, or exactly at the triangle.  The Float32 record, the entry
// point, `best` and `tmin` are made exactly as spira_bvh.h and bvh8_enter / bvh8_begin make them.
// usage: tri_screen <blocks> <iterations per thread> <seed>; prints counts; exit 0 iff no accepted triangle was screened out
#include <cstdio>
#include <cstdlib>
#include "../../julia-spira_amd/csrc/spira_device.h"

using namespace spira;

__device__ __forceinline__ uint32_t mixh(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
struct Rng {
    uint32_t s;
    __device__ uint32_t next() { s = mixh(s + 0x9e3779b9u); return s; }
    __device__ double uni() { return (double)(next() >> 8) * (1.0 / 16777216.0); }                     // [0, 1)
    __device__ double sym() { return 2.0 * uni() - 1.0; }
    __device__ double logu(double lo, double hi) { return lo * exp(uni() * log(hi / lo)); }
};

struct Counts { unsigned long long n, accepted, screened_out, rejected, rejected_passed, certain, certain_wrong; };

__global__ void k_screen(uint32_t iters, uint32_t seed, Counts *out) {
    Rng g; g.s = mixh(seed ^ (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u);
    unsigned long long n = 0, acc = 0, bad = 0, rej = 0, rejp = 0, cer = 0, cerbad = 0;
    for (uint32_t it = 0; it < iters; ++it) {
        // ---- the mesh frame
        const double scale = ldexp(1.0, (int)(g.next() % 17u) - 8);
        const double cx = g.sym() * 1000.0 / scale * 0.001 * (double)(g.next() % 1000u), cy = g.sym() * 300.0, cz = g.sym() * 3.0;
        // ---- a triangle in the normalised frame, then in the caller's coordinates the way a caller would hold it (three vertices)
        double v0n[3] = {0.5 * g.sym(), 0.5 * g.sym(), 0.5 * g.sym()};
        const double len1 = g.logu(1e-5, 0.5), len2 = g.logu(1e-5, 0.5);
        double a1[3] = {g.sym(), g.sym(), g.sym()}, a2[3] = {g.sym(), g.sym(), g.sym()};
        double n1 = sqrt(a1[0] * a1[0] + a1[1] * a1[1] + a1[2] * a1[2]) + 1e-30, n2 = sqrt(a2[0] * a2[0] + a2[1] * a2[1] + a2[2] * a2[2]) + 1e-30;
        for (int k = 0; k < 3; ++k) { a1[k] *= len1 / n1; a2[k] *= len2 / n2; }
        if ((g.next() & 3u) == 0) { const double f = g.sym() * 2.0, eps = g.logu(1e-7, 1e-2); for (int k = 0; k < 3; ++k) a2[k] = f * a1[k] + eps * a2[k]; }      // a sliver
        double V0[3], V1[3], V2[3];
        const double c[3] = {cx, cy, cz};
        for (int k = 0; k < 3; ++k) {
            double p0 = v0n[k], p1 = v0n[k] + a1[k], p2 = v0n[k] + a2[k];
            p1 = fmin(0.5, fmax(-0.5, p1)); p2 = fmin(0.5, fmax(-0.5, p2));                           // keep the vertices inside the unit box
            V0[k] = p0 / scale + c[k]; V1[k] = p1 / scale + c[k]; V2[k] = p2 / scale + c[k];
        }
        Pack4<double> v0, e1, e2;
        v0.x = V0[0]; v0.y = V0[1]; v0.z = V0[2]; v0.w = 0;
        e1.x = V1[0] - V0[0]; e1.y = V1[1] - V0[1]; e1.z = V1[2] - V0[2]; e1.w = 0;                      // :149
        e2.x = V2[0] - V0[0]; e2.y = V2[1] - V0[1]; e2.z = V2[2] - V0[2]; e2.w = 0;                      // :150
        // the builder's frame centre is a Float64 value near the middle of the mesh: here c itself; the Float32 record as spira_bvh.h writes it
        const double E1[3] = {e1.x, e1.y, e1.z}, E2[3] = {e2.x, e2.y, e2.z};
        float vf[3], f1[3], f2[3], L = 0.0f;
        for (int k = 0; k < 3; ++k) {
            vf[k] = (float)((V0[k] - c[k]) * scale);
            f1[k] = (float)(E1[k] * scale); f2[k] = (float)(E2[k] * scale);
            L = fmaxf(L, fmaxf(fabsf(f1[k]), fabsf(f2[k])));
        }
        L = __uint_as_float(__float_as_uint(L) + 1u);                                                    // nextafter upward (L >= 0, finite)
        const uint4 u0 = make_uint4(__float_as_uint(vf[0]), __float_as_uint(vf[1]), __float_as_uint(vf[2]), 0u);
        const uint4 u1 = make_uint4(__float_as_uint(f1[0]), __float_as_uint(f1[1]), __float_as_uint(f1[2]), __float_as_uint(L));
        const uint4 u2 = make_uint4(__float_as_uint(f2[0]), __float_as_uint(f2[1]), __float_as_uint(f2[2]), 0u);
        // ---- the point aimed at (caller's coordinates), chosen against the screen
        double bu = g.uni(), bv = g.uni();
        if (bu + bv > 1.0) { bu = 1.0 - bu; bv = 1.0 - bv; }
        const uint32_t kind = g.next() % 12u;
        const double off = (g.next() & 1u) ? g.logu(1e-9, 1e-2) : -g.logu(1e-9, 1e-2);
        if (kind == 0) bu = 0; else if (kind == 1) bv = 0; else if (kind == 2) bv = 1.0 - bu;
        else if (kind == 3) { bu = 0; bv = 0; } else if (kind == 4) { bu = 1; bv = 0; } else if (kind == 5) { bu = 0; bv = 1; }
        else if (kind == 6) bu = off; else if (kind == 7) bv = off; else if (kind == 8) bv = 1.0 - bu + off;
        double tgt[3];
        for (int k = 0; k < 3; ++k) tgt[k] = V0[k] + bu * E1[k] + bv * E2[k];
        // ---- direction: random, one in four nearly in the triangle's plane
        double d[3] = {g.sym(), g.sym(), g.sym()};
        if ((g.next() & 3u) == 0) {
            const double nx = E1[1] * E2[2] - E1[2] * E2[1], ny = E1[2] * E2[0] - E1[0] * E2[2], nz = E1[0] * E2[1] - E1[1] * E2[0];
            const double nn = nx * nx + ny * ny + nz * nz;
            if (nn > 0) { const double dn = (d[0] * nx + d[1] * ny + d[2] * nz) / nn * (1.0 - g.logu(1e-7, 1e-1)); d[0] -= dn * nx; d[1] -= dn * ny; d[2] -= dn * nz; }
        }
        const double dl = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        if (!(dl > 1e-12)) continue;
        Vec<double> dv = mk<double>(d[0], d[1], d[2]);
        dv = normalize(dv);                                                                               // every ray of the kernels is normalised (:27-28)
        const double t_true = g.logu(1e-3, 1.0) / scale * ((g.next() & 7u) == 0 ? 1e-2 : 1.0);            // some hits closer than t_min
        Vec<double> o = mk<double>(tgt[0] - dv.x * t_true, tgt[1] - dv.y * t_true, tgt[2] - dv.z * t_true);
        // the walk starts where the ray enters the mesh's box (te >= 0): here anywhere along the first half of the way, or at the origin itself
        const double te = (g.next() & 1u) ? 0.0 : t_true * 0.5 * g.uni();
        // the origin must lie in the box for te = 0 to be what bvh8_enter returns; a ray whose origin falls outside takes its entry point instead
        Bvh8Ray r;
        const double on[3] = {((o.x + dv.x * te) - c[0]) * scale, ((o.y + dv.y * te) - c[1]) * scale, ((o.z + dv.z * te) - c[2]) * scale};
        if (fabs(on[0]) > 0.56 || fabs(on[1]) > 0.56 || fabs(on[2]) > 0.56) continue;
        r.ox = (float)on[0]; r.oy = (float)on[1]; r.oz = (float)on[2];
        r.ix = r.iy = r.iz = 1.0f; r.oct = 0;
        double closest = INFINITY;
        const uint32_t ck = g.next() % 6u;
        if (ck == 1) closest = t_true * (1.0 + g.logu(1e-12, 1e-3)); else if (ck == 2) closest = t_true * (1.0 - g.logu(1e-12, 1e-3));
        else if (ck == 3) closest = t_true; else if (ck == 4) closest = t_true * g.logu(1.0, 100.0);
        r.best = bvh8_best((float)((closest - te) * scale));
        Bvh8Walk<double> w;
        bvh8_begin<double>(w, r, te, 0.001, scale);
        double t;
        const bool exact = triangle_test<double>(v0, e1, e2, o, dv, 0.001, closest, t);
        float t_hi;
        const int cls = tri_screen_f32(u0, u1, u2, r, (float)dv.x, (float)dv.y, (float)dv.z, t_hi);
        const bool maybe = cls != 0;
        ++n;
        if (exact) { ++acc; if (!maybe) ++bad; } else { ++rej; if (maybe) ++rejp; }
        if (cls == 2) {             // a hit for sure unless something closer is known: the exact test without a closer hit accepts, at a distance within the bound
            ++cer;
            double t2;
            const bool hit = triangle_test<double>(v0, e1, e2, o, dv, 0.001, (double)INFINITY, t2);
            if (!hit || !((t2 - te) * scale <= (double)t_hi)) ++cerbad;
        }
    }
    atomicAdd(&out->n, n); atomicAdd(&out->accepted, acc); atomicAdd(&out->screened_out, bad); atomicAdd(&out->rejected, rej); atomicAdd(&out->rejected_passed, rejp);
    atomicAdd(&out->certain, cer); atomicAdd(&out->certain_wrong, cerbad);
}

int main(int argc, char **argv) {
    const uint32_t blocks = argc > 1 ? (uint32_t)atoi(argv[1]) : 1024, iters = argc > 2 ? (uint32_t)atoi(argv[2]) : 256, seed = argc > 3 ? (uint32_t)atoll(argv[3]) : 1u;
    Counts *d = nullptr, h{};
    if (hipMalloc(&d, sizeof h) != hipSuccess || hipMemset(d, 0, sizeof h) != hipSuccess) { fprintf(stderr, "no device\n"); return 2; }
    hipLaunchKernelGGL(k_screen, dim3(blocks), dim3(256), 0, 0, iters, seed, d);
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(&h, d, sizeof h, hipMemcpyDeviceToHost) != hipSuccess) { fprintf(stderr, "kernel failed\n"); return 2; }
    printf("pairs %llu: exact test accepts %llu, of those screened out %llu; exact test rejects %llu, of those the screen lets through %llu (%.1f %%); "
           "class 2 (certain hits) %llu, of those wrong %llu\n", h.n, h.accepted,
           h.screened_out, h.rejected, h.rejected_passed, h.rejected ? 100.0 * (double)h.rejected_passed / (double)h.rejected : 0.0, h.certain, h.certain_wrong);
    return h.screened_out == 0 && h.certain_wrong == 0 && h.accepted > h.n / 20 && h.certain > h.n / 100 ? 0 : 1;
}
