// ptr_kernels_ext.hip — EXTENSION modes: Whitted reflection/refraction to depth N and Philox-seeded
// multi-sample pixels.  The reference implements none of this (MAX_RAY_DEPTH is
// never read, Material.reflection/.transparency are never used, there is no sampling), so there is
// no reference parity to claim: the specification is the "weighted depth-first expansion" written out
// in include/ptr.h (extension modes), and this file performs that arithmetic in that order.
//   T = double: bit-identical to that specification (compiled with -ffp-contract=off)
//   T = float : the throughput form, compared with it within a stated tolerance
// With max_depth 0 and spp 1 the result equals the reference path (and ptr_kernels_exact.hip).
#include <hip/hip_runtime.h>
#include "ptr_device.h"
#include "ptr_kernels.h"
#include "ptr_fast_common.h"
#include "ptr_f64_roots.h"

namespace ptr {

#define PTR_EXT_MAX_DEPTH 16

template <typename T> struct V { T x, y, z; };

template <typename T> __device__ __forceinline__ T tsqrt(T x);
template <> __device__ __forceinline__ double tsqrt<double>(double x) { return __builtin_sqrt(x); }
template <> __device__ __forceinline__ float tsqrt<float>(float x) { return __builtin_sqrtf(x); }

template <typename T> __device__ __forceinline__ void normalize_v(V<T>& v)
{
    const T len2 = v.x * v.x + v.y * v.y + v.z * v.z;      // Vector3.js:23-33
    if (len2 > 0) {
        const T inv = T(1) / tsqrt<T>(len2);
        v.x *= inv; v.y *= inv; v.z *= inv;
    }
}

// Scene access: float64 reads the exact records (absolute coordinates); float32 reads the
// camera-relative records, whose origin is the camera (so rays start at 0 there).
template <typename T> struct SceneT;
template <> struct SceneT<double> {
    FrameView fv;
    __device__ V<double> center(int i) const { const SphereG64 g = fv.g64[i]; return V<double>{g.cx, g.cy, g.cz}; }
    __device__ double r2(int i) const { return fv.g64[i].r2; }
    __device__ V<double> sc(int i) const { const SphereM64& m = fv.m64[i]; return V<double>{m.sr, m.sg, m.sb}; }
    __device__ V<double> em(int i) const { const SphereM64& m = fv.m64[i]; return V<double>{m.er, m.eg, m.eb}; }
    __device__ double refl(int i) const { return fv.m64[i].refl; }
    __device__ double transp(int i) const { return fv.m64[i].transp; }
    __device__ V<double> bg() const { return V<double>{fv.h->bg64[0], fv.h->bg64[1], fv.h->bg64[2]}; }
    // planes (element n + k) and Phong terms
    __device__ V<double> pl_p(int k) const { const PlaneG64& g = fv.pl64()[k]; return V<double>{g.px, g.py, g.pz}; }
    __device__ V<double> pl_n(int k) const { const PlaneG64& g = fv.pl64()[k]; return V<double>{g.nx, g.ny, g.nz}; }
    __device__ V<double> pl_sc(int k) const { const SphereM64& m = fv.plm64()[k]; return V<double>{m.sr, m.sg, m.sb}; }
    __device__ V<double> pl_em(int k) const { const SphereM64& m = fv.plm64()[k]; return V<double>{m.er, m.eg, m.eb}; }
    __device__ double pl_refl(int k) const { return fv.plm64()[k].refl; }
    __device__ double pl_transp(int k) const { return fv.plm64()[k].transp; }
    __device__ double ks(int e) const { return fv.ph64()[e]; }
    __device__ unsigned phong_exp(int e, int ne) const { return (unsigned)fv.ph64()[ne + e]; }
};
template <> struct SceneT<float> {
    FrameView fv;
    __device__ V<float> center(int i) const { const SphereP32& p = fv.p32[i]; return V<float>{p.lx, p.ly, p.lz}; }
    __device__ float r2(int i) const { return fv.p32[i].r2; }
    __device__ V<float> sc(int i) const { const SphereM32& m = fv.m32[i]; return V<float>{m.sr, m.sg, m.sb}; }
    __device__ V<float> em(int i) const { const SphereM32& m = fv.m32[i]; return V<float>{m.er, m.eg, m.eb}; }
    __device__ float refl(int i) const { return fv.p32[i].refl; }
    __device__ float transp(int i) const { return fv.p32[i].transp; }
    __device__ V<float> bg() const { return V<float>{fv.h->bg32[0], fv.h->bg32[1], fv.h->bg32[2]}; }
    __device__ V<float> pl_p(int k) const { const PlaneP32& g = fv.pl32()[k]; return V<float>{g.px, g.py, g.pz}; }
    __device__ V<float> pl_n(int k) const { const PlaneP32& g = fv.pl32()[k]; return V<float>{g.nx, g.ny, g.nz}; }
    __device__ V<float> pl_sc(int k) const { const SphereM32& m = fv.plm32()[k]; return V<float>{m.sr, m.sg, m.sb}; }
    __device__ V<float> pl_em(int k) const { const SphereM32& m = fv.plm32()[k]; return V<float>{m.er, m.eg, m.eb}; }
    __device__ float pl_refl(int k) const { return fv.pl32()[k].refl; }
    __device__ float pl_transp(int k) const { return fv.pl32()[k].transp; }
    __device__ float ks(int e) const { return fv.ph32()[e]; }
    __device__ unsigned phong_exp(int e, int ne) const { return (unsigned)fv.ph32()[ne + e]; }
};

// per-lane work counters (PTR_FLAG_STATS), same meaning as ptr_stats in include/ptr.h
enum { XT_CAMERA = 0, XT_HITS, XT_SHADOW, XT_EXIT_TCA, XT_EXIT_D2, XT_EXIT_HIT, XT_SECONDARY, XT_N };
struct ExtCounts { unsigned c[XT_N] = {0, 0, 0, 0, 0, 0, 0}; };
template <bool STATS> __device__ __forceinline__ void xcount(ExtCounts* k, int which) { if constexpr (STATS) k->c[which]++; }

// Sphere.intersect (Sphere.js:15-33); any_hit skips the unused sqrt of the shadow test
template <typename T, bool STATS>
__device__ __forceinline__ bool intersect_t(const V<T>& o, const V<T>& d, const V<T>& c, T r2, T& t0, T& t1, bool any_hit, ExtCounts* k)
{
    const T lx = c.x - o.x, ly = c.y - o.y, lz = c.z - o.z;
    const T tca = lx * d.x + ly * d.y + lz * d.z;
    if (tca < 0) { xcount<STATS>(k, XT_EXIT_TCA); return false; }
    const T d2 = (lx * lx + ly * ly + lz * lz) - tca * tca;
    if (d2 > r2) { xcount<STATS>(k, XT_EXIT_D2); return false; }
    xcount<STATS>(k, XT_EXIT_HIT);
    if (!any_hit) {
        const T thc = tsqrt<T>(r2 - d2);
        t0 = tca - thc; t1 = tca + thc;
    }
    return true;
}

// Plane.intersect of the specification (include/ptr.h "planes and Phong"): denom = d.n; 0 -> miss; t = ((p - o).n) / denom
template <typename T, bool STATS>
__device__ __forceinline__ bool plane_t(const V<T>& o, const V<T>& d, const V<T>& p, const V<T>& n, T& t0, ExtCounts* k)
{
    const T denom = d.x * n.x + d.y * n.y + d.z * n.z;
    if (denom == 0) { xcount<STATS>(k, XT_EXIT_TCA); return false; }
    const T t = ((p.x - o.x) * n.x + (p.y - o.y) * n.y + (p.z - o.z) * n.z) / denom;
    if (t < 0) { xcount<STATS>(k, XT_EXIT_TCA); return false; }
    xcount<STATS>(k, XT_EXIT_HIT);
    t0 = t;
    return true;
}

template <typename T> struct ExtRay { V<T> o, d, w; int depth; };

// weighted depth-first expansion (include/ptr.h, "extension modes")
template <typename T, bool STATS>
__device__ V<T> trace_ext(const SceneT<T>& sc, V<T> o0, V<T> d0, int max_depth, ExtCounts* k)
{
    const int n = sc.fv.h->n, nl = sc.fv.h->n_lights;
    const int np = (int)sc.fv.h->n_planes, ne = n + np;            // elements = spheres, then planes
    const bool phong = sc.fv.h->has_phong != 0u;
    // The ray that would be popped next stays in registers (a reflection is pushed last, so it is processed first):
    // same rays in the same order, and only a second child travels through the scratch-memory stack.
    ExtRay<T> stack[PTR_EXT_MAX_DEPTH + 2];
    int top = 0;
    ExtRay<T> r{o0, d0, V<T>{1, 1, 1}, 0};
    bool have = true;
    V<T> acc{0, 0, 0};
    const V<T> bg = sc.bg();
    while (have || top > 0) {
        if (!have) r = stack[--top];
        have = false;
        T tnear = T(PTR_INFINITY);
        int e = -1;
        for (int i = 0; i < n; i++) {
            T t0 = T(PTR_INFINITY), t1 = T(PTR_INFINITY);
            if (intersect_t<T, STATS>(r.o, r.d, sc.center(i), sc.r2(i), t0, t1, false, k)) {
                if (t0 < 0) t0 = t1;
                if (t0 < tnear) { tnear = t0; e = i; }
            }
        }
        for (int q = 0; q < np; q++) {
            T t0 = T(PTR_INFINITY);
            if (plane_t<T, STATS>(r.o, r.d, sc.pl_p(q), sc.pl_n(q), t0, k) && t0 < tnear) { tnear = t0; e = n + q; }
        }
        if (e < 0) {
            acc.x += r.w.x * bg.x; acc.y += r.w.y * bg.y; acc.z += r.w.z * bg.z;
            continue;
        }
        xcount<STATS>(k, XT_HITS);
        const bool on_plane = e >= n;
        const V<T> em = on_plane ? sc.pl_em(e - n) : sc.em(e), col = on_plane ? sc.pl_sc(e - n) : sc.sc(e);
        acc.x += r.w.x * em.x; acc.y += r.w.y * em.y; acc.z += r.w.z * em.z;
        const V<T> P{r.o.x + r.d.x * tnear, r.o.y + r.d.y * tnear, r.o.z + r.d.z * tnear};
        V<T> N;
        if (on_plane) {
            N = sc.pl_n(e - n);                                   // a plane's normal is the unit vector it was given
        } else {
            const V<T> C = sc.center(e);
            N = V<T>{P.x - C.x, P.y - C.y, P.z - C.z};
            normalize_v(N);
        }
        bool inside = false;
        T dn = r.d.x * N.x + r.d.y * N.y + r.d.z * N.z;
        if (dn > 0) { N.x = -N.x; N.y = -N.y; N.z = -N.z; inside = true; dn = -dn; }
        const T transp = on_plane ? sc.pl_transp(e - n) : sc.transp(e), refl = on_plane ? sc.pl_refl(e - n) : sc.refl(e);
        if ((transp > 0 || refl > 0) && r.depth < max_depth) {
            const T facing = -dn;
            const T f = 1 - facing;
            const T fresnel = T(1) * T(0.1) + (f * f * f) * (T(1) - T(0.1));
            const V<T> wsc{r.w.x * col.x, r.w.y * col.y, r.w.z * col.z};
            if (transp > 0) {
                const T eta = inside ? T(1.1) : T(1) / T(1.1);
                const T cosi = -dn;
                const T kk2 = 1 - eta * eta * (1 - cosi * cosi);
                if (kk2 >= 0) {
                    ExtRay<T>& c = stack[top++];
                    xcount<STATS>(k, XT_SECONDARY);
                    const T g = eta * cosi - tsqrt<T>(kk2);
                    const T wt = (1 - fresnel) * transp;
                    c.o = V<T>{P.x - N.x * T(PTR_BIAS), P.y - N.y * T(PTR_BIAS), P.z - N.z * T(PTR_BIAS)};
                    c.d = V<T>{r.d.x * eta + N.x * g, r.d.y * eta + N.y * g, r.d.z * eta + N.z * g};
                    c.w = V<T>{wsc.x * wt, wsc.y * wt, wsc.z * wt};
                    normalize_v(c.d);
                    c.depth = r.depth + 1;
                }
            }
            xcount<STATS>(k, XT_SECONDARY);
            const T two = 2 * dn;
            ExtRay<T> c;
            c.o = V<T>{P.x + N.x * T(PTR_BIAS), P.y + N.y * T(PTR_BIAS), P.z + N.z * T(PTR_BIAS)};
            c.d = V<T>{r.d.x - N.x * two, r.d.y - N.y * two, r.d.z - N.z * two};
            c.w = V<T>{wsc.x * fresnel, wsc.y * fresnel, wsc.z * fresnel};
            normalize_v(c.d);
            c.depth = r.depth + 1;
            r = c;
            have = true;
        } else {
            // the reference's direct lighting (RayTracer.js:60-91), over spheres and planes, with the Phong term
            const V<T> so{P.x + N.x * T(PTR_BIAS), P.y + N.y * T(PTR_BIAS), P.z + N.z * T(PTR_BIAS)};
            V<T> dif{0, 0, 0};
            const T ks = phong ? sc.ks(e) : T(0);
            const T two_dn = 2 * dn;
            const V<T> Rv{r.d.x - N.x * two_dn, r.d.y - N.y * two_dn, r.d.z - N.z * two_dn};
            // lights: the emissive spheres in scene order (the host's list), then the emissive planes
            for (int kk = 0; kk < nl + np; kk++) {
                int i;
                V<T> Ci, ei;
                if (kk < nl) {
                    i = sc.fv.lights[kk];
                    Ci = sc.center(i); ei = sc.em(i);
                } else {
                    i = n + (kk - nl);
                    ei = sc.pl_em(kk - nl);
                    if (!(ei.x > 0 || ei.y > 0 || ei.z > 0)) continue;
                    Ci = sc.pl_p(kk - nl);
                }
                V<T> L{Ci.x - P.x, Ci.y - P.y, Ci.z - P.z};
                const T len2 = L.x * L.x + L.y * L.y + L.z * L.z;                    // a plane only blocks in front of the light
                normalize_v(L);
                xcount<STATS>(k, XT_SHADOW);
                T tr = 1;
                for (int j = 0; j < ne; j++) {
                    if (j == i) continue;
                    T t0, t1;
                    if (j < n ? intersect_t<T, STATS>(so, L, sc.center(j), sc.r2(j), t0, t1, true, k)
                              : (plane_t<T, STATS>(so, L, sc.pl_p(j - n), sc.pl_n(j - n), t0, k) && t0 * t0 < len2)) { tr = 0; break; }
                }
                T ratio = N.x * L.x + N.y * L.y + N.z * L.z;
                ratio = (ratio != ratio) ? ratio : (ratio > 0 ? ratio : T(0));       // Math.max(0, .)
                dif.x += (col.x * tr) * (ei.x * ratio);
                dif.y += (col.y * tr) * (ei.y * ratio);
                dif.z += (col.z * tr) * (ei.z * ratio);
                if (ks > 0) {
                    T sp = Rv.x * L.x + Rv.y * L.y + Rv.z * L.z;
                    sp = (sp != sp) ? sp : (sp > 0 ? sp : T(0));
                    T ph = 1, b = sp;
                    for (unsigned q = sc.phong_exp(e, ne); q > 0; q >>= 1) { if (q & 1u) ph = ph * b; b = b * b; }
                    const T kp = ks * ph;
                    dif.x += tr * (ei.x * kp);
                    dif.y += tr * (ei.y * kp);
                    dif.z += tr * (ei.z * kp);
                }
            }
            acc.x += r.w.x * dif.x; acc.y += r.w.y * dif.y; acc.z += r.w.z * dif.z;
        }
    }
    return acc;
}

// ---- float32 throughput form of the same expansion -------------------------------------------------------
// Same rays, same weights, same order as trace_ext<float>; what differs is how a wave gets through the sphere
// loops (and, within float32 rounding, the arithmetic: rsq instead of 1/sqrt, FMA contraction by hand):
//   * sphere records arrive by scalar loads (wave-uniform index, constant address space);
//   * Sphere.intersect is evaluated branch-free for all lanes (b, disc), the root only when some lane hits;
//   * the shadow loop first asks the (light, occluder) cone table of ptr_device.h (PairQ): the shadow ray runs
//     along the line through the light's centre, so an occluder whose cone that line misses is skipped by the
//     whole wave; lanes facing away from the light are not tested (their term is 0 either way); the loop ends
//     when every lane is decided.
typedef const f4 PTR_CONST* ConstRec;
__device__ __forceinline__ bool any_lane_x(bool c) { return ballot(c) != 0ull; }

// The stack of rays that wait (second children only: the reflection stays in registers).  Its entries have strictly
// increasing depths from bottom to top, so max_depth entries per lane are enough.  The lowest PTR_EXT_LDS_ENTRIES of them
// live in LDS, [entry][word][lane] (a lane's words are 256 bytes apart: every access is conflict-free), 2 560 bytes per entry
// and wave: nothing of them ever reaches HBM.  The deeper entries (a chain of internal reflections inside a transparent
// sphere pushes one refracted ray per bounce) go to a per-lane array in scratch memory.  How many entries LDS holds is an
// occupancy question: the kernel is bound by instruction issue AND by the latency of a wave, so waves per SIMD count.
// Measured (1080p, example scene, depth 5, 256 frames): in the general kernel (94 VGPRs: 5 waves per SIMD) 3 entries, a
// fourth costs a wave (14 % slower); as a kernel of its own (render_ext_f32_kernel, 70 VGPRs) 3 entries / 6 waves 14.1 ms,
// 2 entries / 7 waves 12.7 ms, 2 entries / 8 waves (3 VGPR spills) 12.7 ms, 1 entry / 7 waves 13.0 ms, 4 entries 16.1 ms.
#ifndef PTR_EXT_LDS_ENTRIES
#define PTR_EXT_LDS_ENTRIES 2
#endif
struct RayStack {
    float* base;       // this lane's first word in LDS
    ExtRay<float> deep[PTR_EXT_MAX_DEPTH + 2 - PTR_EXT_LDS_ENTRIES];
    __device__ __forceinline__ void push(int at, const ExtRay<float>& c)
    {
        if (at < PTR_EXT_LDS_ENTRIES) {
            float* p = base + at * 640;
            p[0] = c.o.x; p[64] = c.o.y; p[128] = c.o.z; p[192] = c.d.x; p[256] = c.d.y; p[320] = c.d.z;
            p[384] = c.w.x; p[448] = c.w.y; p[512] = c.w.z; p[576] = __int_as_float(c.depth);
        } else {
            deep[at - PTR_EXT_LDS_ENTRIES] = c;
        }
    }
    __device__ __forceinline__ ExtRay<float> pop(int at) const
    {
        if (at < PTR_EXT_LDS_ENTRIES) {
            const float* p = base + at * 640;
            return ExtRay<float>{V<float>{p[0], p[64], p[128]}, V<float>{p[192], p[256], p[320]}, V<float>{p[384], p[448], p[512]},
                                 __float_as_int(p[576])};
        }
        return deep[at - PTR_EXT_LDS_ENTRIES];
    }
};

// cand0: the spheres a camera ray of this wave's tile can reach at all (bit i; all ones: no such knowledge)
// GEN: the scene may have planes (elements n, n + 1, ... after the spheres) and Phong terms (include/ptr.h "planes and Phong");
// a kernel of its own, so that sphere scenes do not carry its registers.
template <bool GEN, typename Stack>
__device__ V<float> trace_ext_f32(Stack& stack, const unsigned char* block, const RenderArgs& a, V<float> o0, V<float> d0, int max_depth,
                                  unsigned long long cand0 = ~0ull)
{
    const ConstHeader hdr = (ConstHeader)(uintptr_t)block;
    const ConstF4 x4v0 = (ConstF4)(uintptr_t)(block + a.off_x4);        // lx ly lz r2 per sphere, padded with never-hit records
    const ConstF4 pairv0 = (ConstF4)(uintptr_t)(block + a.off_pairs);
    const ConstF4 lrec0 = (ConstF4)(uintptr_t)(block + a.off_lrec);
    const SphereP32* gp32 = reinterpret_cast<const SphereP32*>(block + a.off_p32);
    const SphereM32* gm32 = reinterpret_cast<const SphereM32*>(block + a.off_m32);
    const int n = a.n, nl = hdr->n_lights;
    const float bgx = hdr->bg32[0], bgy = hdr->bg32[1], bgz = hdr->bg32[2];
    constexpr float BIAS = (float)PTR_BIAS;
    // planes: px py pz refl | nx ny nz transp (camera-relative like the spheres), their materials, the Phong terms (ks per
    // element, then the exponents)
    const int np = GEN ? (int)hdr->n_planes : 0;
    const bool phong = GEN && hdr->has_phong != 0u;
    const ConstF4 plv0 = (ConstF4)(uintptr_t)(block + (GEN ? hdr->off_pl32 : 0u));
    const PlaneP32* gpl32 = reinterpret_cast<const PlaneP32*>(block + (GEN ? hdr->off_pl32 : 0u));
    const SphereM32* gplm32 = reinterpret_cast<const SphereM32*>(block + (GEN ? hdr->off_plm32 : 0u));
    const float* gph32 = reinterpret_cast<const float*>(block + (GEN ? hdr->off_ph32 : 0u));

    // Depth-first expansion with the ray that would be popped next kept in registers: a hit that spawns only a
    // reflection (the usual case) never touches the stack; the order of the rays, and with it the order of the sums, is
    // the specification's.
    int top = 0;
    ExtRay<float> r{o0, d0, V<float>{1, 1, 1}, 0};
    bool have = true;
    bool first = true;                                           // the camera ray: all lanes take the loop's first pass together
    float ax = 0.0f, ay = 0.0f, az = 0.0f;
    while (have || top > 0) {
        if (!have) r = stack.pop(--top);
        have = false;
        // nearest hit (RayTracer.js:26-41)
        float tnear = (float)PTR_INFINITY;
        int e = -1;
        {
            // The spheres come four at a time (lx ly lz r2 of each, one s_load_dwordx16 from the compact table), in two
            // buffers that swap roles: the next four are requested before these four are used, and the wave waits for scalar
            // data once per four tests instead of once per test (scalar loads return out of order: a wait is for all of them).
            // Hit / miss is decided on lane masks.
            auto test = [&](const f4 q, int i) {
                const float lx = q.x - r.o.x, ly = q.y - r.o.y, lz = q.z - r.o.z;
                const float b = fmaf(lz, r.d.z, fmaf(ly, r.d.y, lx * r.d.x));
                const float ll = fmaf(lz, lz, fmaf(ly, ly, lx * lx));
                const float disc = fmaf(b, b, q.w - ll);                      // radius2 - d2 (Sphere.js:21-27)
                // (b >= 0 and disc >= 0 as ONE compare: a compare's mask can be branched on directly, the AND of two cannot)
                const unsigned long long hit_m = ballot(fminf(b, disc) >= 0.0f);
                if (hit_m != 0ull) {
                    const float thc = disc * rsq(fmax_raw(disc, 1e-30f));
                    float t = b - thc;
                    t = t < 0.0f ? b + thc : t;                               // RayTracer.js:32-34
                    const bool nearer = lane_of(hit_m) & (t < tnear);
                    tnear = nearer ? t : tnear;
                    e = nearer ? i : e;
                }
            };
            auto test4 = [&](const P32W& q, int i) {
                test(q.g0, i);
                if (i + 1 < n) test(q.g1, i + 1);
                if (i + 2 < n) test(q.g2, i + 2);
                if (i + 3 < n) test(q.g3, i + 3);
            };
            if (cand0 != ~0ull && __builtin_amdgcn_readfirstlane((int)first) != 0) {
                // the camera ray: only the spheres the tile's cone reaches (a culled sphere is a certain miss for every lane)
                for (unsigned long long c = cand0; c != 0ull; c &= c - 1ull) {
                    const int i = __builtin_ctzll(c);
                    const f4 q = x4v0[i];
                    arrive(q);
                    test(q, i);
                }
            } else if (n > 0) {
                ConstF4 rp = x4v0;
                P32W A = load_p32w(rp), B;
                for (int i = 0;;) {
                    arrive(A); B = load_p32w(rp + 4); test4(A, i);
                    if ((i += 4) >= n) break;
                    arrive(B); A = load_p32w(rp + 8); test4(B, i);
                    if ((i += 4) >= n) break;
                    rp += 8;
                }
            }
            if constexpr (GEN) {
                // Plane.intersect of the specification: denom = d.n; 0 -> miss; t = ((p - o).n) / denom; t < 0 -> miss
                for (int q = 0; q < np; q++) {
                    const f4 p0 = plv0[2 * q], p1 = plv0[2 * q + 1];
                    arrive(p0); arrive(p1);
                    const float denom = fmaf(r.d.z, p1.z, fmaf(r.d.y, p1.y, r.d.x * p1.x));
                    const float num = fmaf(p0.z - r.o.z, p1.z, fmaf(p0.y - r.o.y, p1.y, (p0.x - r.o.x) * p1.x));
                    const float t = num * rcp(denom);
                    const bool nearer = (denom != 0.0f) & (t >= 0.0f) & (t < tnear);
                    tnear = nearer ? t : tnear;
                    e = nearer ? n + q : e;
                }
            }
            first = false;
        }
        if (e < 0) {
            ax = fmaf(r.w.x, bgx, ax); ay = fmaf(r.w.y, bgy, ay); az = fmaf(r.w.z, bgz, az);
            continue;
        }
        const bool on_plane = GEN && e >= n;
        SphereM32 m;
        float4 ce;
        float refl, transp;
        [[maybe_unused]] float inv_r = 1.0f;
        if (on_plane) {
            const PlaneP32 pl = gpl32[e - n];
            m = gplm32[e - n];
            ce = make_float4(pl.nx, pl.ny, pl.nz, 0.0f);              // (a plane's normal is the unit vector it was given)
            refl = pl.refl; transp = pl.transp;
        } else {
            m = gm32[e];
            ce = *reinterpret_cast<const float4*>(&gp32[e].lx);
            const float4 g3 = *reinterpret_cast<const float4*>(&gp32[e].inv_r);      // inv_r hed refl transp
            refl = g3.z; transp = g3.w; inv_r = g3.x;
        }
        ax = fmaf(r.w.x, m.er, ax); ay = fmaf(r.w.y, m.eg, ay); az = fmaf(r.w.z, m.eb, az);
        V<float> P{fmaf(r.d.x, tnear, r.o.x), fmaf(r.d.y, tnear, r.o.y), fmaf(r.d.z, tnear, r.o.z)};
        V<float> N{P.x - ce.x, P.y - ce.y, P.z - ce.z};
        {
            const float inv = rsq(fmaf(N.z, N.z, fmaf(N.y, N.y, N.x * N.x)));
            N.x *= inv; N.y *= inv; N.z *= inv;
        }
        if (on_plane) N = V<float>{ce.x, ce.y, ce.z};
#ifndef PTR_EXT_NO_SNAP
        else {
            // The hit point, put back ON the sphere: o + d t is off by eps (|o| + t) in every direction, and radially that can
            // exceed the 1e-4 by which the next ray's origin is moved off the surface (a small sphere far away: the refracted
            // ray then starts on the wrong side and the whole sphere comes out wrong).  C + r N is on the surface to eps (|C| + r).
            // (only where that is the smaller error: a sphere smaller than the ray is long)
            const float rr = rcp(inv_r);
            const bool snap = rr < tnear;
            P = V<float>{snap ? fmaf(N.x, rr, ce.x) : P.x, snap ? fmaf(N.y, rr, ce.y) : P.y, snap ? fmaf(N.z, rr, ce.z) : P.z};
        }
#endif
        float dn = fmaf(r.d.z, N.z, fmaf(r.d.y, N.y, r.d.x * N.x));
        const bool inside = dn > 0.0f;
        if (inside) { N.x = -N.x; N.y = -N.y; N.z = -N.z; dn = -dn; }
        if ((transp > 0.0f || refl > 0.0f) && r.depth < max_depth) {
            const float f = 1.0f + dn;                                    // 1 - facing, facing = -(d.N)
            const float fresnel = fmaf(f * f * f, 0.9f, 0.1f);
            const float wx = r.w.x * m.sr, wy = r.w.y * m.sg, wz = r.w.z * m.sb;
            if (transp > 0.0f) {
                const float eta = inside ? 1.1f : 1.0f / 1.1f;
                const float cosi = -dn;
                const float k2 = 1.0f - eta * eta * (1.0f - cosi * cosi);
                if (k2 >= 0.0f) {
                    ExtRay<float> c;
                    const float g = eta * cosi - __builtin_sqrtf(k2);
                    const float wt = (1.0f - fresnel) * transp;
                    c.o = V<float>{fmaf(-N.x, BIAS, P.x), fmaf(-N.y, BIAS, P.y), fmaf(-N.z, BIAS, P.z)};
                    V<float> cd{fmaf(r.d.x, eta, N.x * g), fmaf(r.d.y, eta, N.y * g), fmaf(r.d.z, eta, N.z * g)};
                    const float inv = rsq(fmaf(cd.z, cd.z, fmaf(cd.y, cd.y, cd.x * cd.x)));
                    c.d = V<float>{cd.x * inv, cd.y * inv, cd.z * inv};
                    c.w = V<float>{wx * wt, wy * wt, wz * wt};
                    c.depth = r.depth + 1;
                    stack.push(top++, c);
                }
            }
            // the reflection is the ray processed next (it would be pushed last and popped first): it stays in r
            const float two = 2.0f * dn;
            V<float> cd{fmaf(-N.x, two, r.d.x), fmaf(-N.y, two, r.d.y), fmaf(-N.z, two, r.d.z)};
            const float inv = rsq(fmaf(cd.z, cd.z, fmaf(cd.y, cd.y, cd.x * cd.x)));
            r.o = V<float>{fmaf(N.x, BIAS, P.x), fmaf(N.y, BIAS, P.y), fmaf(N.z, BIAS, P.z)};
            r.d = V<float>{cd.x * inv, cd.y * inv, cd.z * inv};
            r.w = V<float>{wx * fresnel, wy * fresnel, wz * fresnel};
            r.depth = r.depth + 1;
            have = true;
        } else {
            // the reference's direct lighting (RayTracer.js:60-91)
            const float sox = fmaf(N.x, BIAS, P.x), soy = fmaf(N.y, BIAS, P.y), soz = fmaf(N.z, BIAS, P.z);
            float dx = 0.0f, dy = 0.0f, dz = 0.0f;
            // Phong term of the specification: ks * max(0, R.L)^exponent per light, R = d - 2 (d.N) N, not tinted by the surface
            float ks = 0.0f, Rx = 0.0f, Ry = 0.0f, Rz = 0.0f;
            unsigned pexp = 1u;
            if constexpr (GEN) if (phong) {
                ks = gph32[e];
                pexp = (unsigned)gph32[n + np + e];
                const float two = 2.0f * dn;
                Rx = fmaf(-N.x, two, r.d.x); Ry = fmaf(-N.y, two, r.d.y); Rz = fmaf(-N.z, two, r.d.z);
            }
            // one light: element i at (lgx, lgy, lgz) with emission (ler, leg, leb); `pairs` = its row of the shadow pre-filter
            auto light = [&](int i, float lgx, float lgy, float lgz, float ler, float leg, float leb, ConstF4 pairs) {
                float Lx = lgx - P.x, Ly = lgy - P.y, Lz = lgz - P.z;
                const float len2 = fmaf(Lz, Lz, fmaf(Ly, Ly, Lx * Lx));
                const float invl = rsq(len2);
                Lx *= invl; Ly *= invl; Lz *= invl;
                const float ratio = fmaf(N.z, Lz, fmaf(N.y, Ly, N.x * Lx));
                float sp = 0.0f;
                if constexpr (GEN) sp = ks > 0.0f ? fmaf(Rz, Lz, fmaf(Ry, Ly, Rx * Lx)) : 0.0f;
                // lanes that still need a verdict (max(0, N.L) = 0 and no highlight: the terms vanish either way), as a lane mask
                unsigned long long open_m = ballot((ratio > 0.0f) | (sp > 0.0f));
                if (open_m != 0ull && n > 0) {
                    auto occluder = [&](const f4 pq, int j) {
                        if (j == i) return;                                   // RayTracer.js:76
                        const float c = fmaf(pq.z, Lz, fmaf(pq.y, Ly, pq.x * Lx));
                        if ((ballot(fabsf(c) + 1e-5f >= pq.w) & open_m) == 0ull) return;
                        const f4 q = x4v0[j];
                        arrive(q);
                        const float lx = q.x - sox, ly = q.y - soy, lz = q.z - soz;
                        const float b = fmaf(lz, Lz, fmaf(ly, Ly, lx * Lx));
                        const float ll = fmaf(lz, lz, fmaf(ly, ly, lx * lx));
                        const float disc = fmaf(b, b, q.w - ll);
                        open_m &= ~ballot(fminf(b, disc) >= 0.0f);            // any hit blocks; distance ignored (F7)
                    };
                    // pre-filter records one ahead, in two buffers that swap roles (rows are padded past n)
                    ConstF4 pp = pairs;
                    f4 A = *pp, B;
                    int j = 0;
                    for (;;) {
                        ++pp; arrive(A); B = *pp; occluder(A, j);
                        if (++j >= n || open_m == 0ull) break;
                        ++pp; arrive(B); A = *pp; occluder(B, j);
                        if (++j >= n || open_m == 0ull) break;
                    }
                }
                if constexpr (GEN) {
                    // a plane only blocks in front of the light (t^2 < |C - P|^2)
                    for (int j = 0; j < np && open_m != 0ull; j++) {
                        if (n + j == i) continue;
                        const f4 p0 = plv0[2 * j], p1 = plv0[2 * j + 1];
                        arrive(p0); arrive(p1);
                        const float denom = fmaf(Lz, p1.z, fmaf(Ly, p1.y, Lx * p1.x));
                        const float t = fmaf(p0.z - soz, p1.z, fmaf(p0.y - soy, p1.y, (p0.x - sox) * p1.x)) * rcp(denom);
                        open_m &= ~ballot((denom != 0.0f) & (t >= 0.0f) & (t * t < len2));
                    }
                }
                const bool open = lane_of(open_m);
                const float lit = (open & (ratio > 0.0f)) ? ratio : 0.0f;
                dx = fmaf(m.sr * ler, lit, dx);
                dy = fmaf(m.sg * leg, lit, dy);
                dz = fmaf(m.sb * leb, lit, dz);
                if constexpr (GEN) {
                    if (ballot(open & (sp > 0.0f)) != 0ull) {
                        float ph = 1.0f, b = sp;
                        for (unsigned q = pexp; q > 0u; q >>= 1) { ph = (q & 1u) ? ph * b : ph; b = b * b; }
                        const float kp = (open & (sp > 0.0f)) ? ks * ph : 0.0f;
                        dx = fmaf(ler, kp, dx); dy = fmaf(leg, kp, dy); dz = fmaf(leb, kp, dz);
                    }
                }
            };
            LightQ lq = load_light(lrec0);                                   // LightRec: light kk at a data-independent address
            for (int kk = 0; kk < nl; kk++) {
                arrive(lq);
                const int i = lq.index;
                const float lgx = lq.g.x, lgy = lq.g.y, lgz = lq.g.z, ler = lq.e.x, leg = lq.e.y, leb = lq.e.z;
                lq = load_light(lrec0 + 3u * (unsigned)(kk + 1));            // (the table ends with a padding record)
                const int prow = kk < PTR_MAX_PAIR_LIGHTS ? kk : (n < PTR_MAX_PAIR_LIGHTS ? n : PTR_MAX_PAIR_LIGHTS);
                light(i, lgx, lgy, lgz, ler, leg, leb, pairv0 + (unsigned)prow * (unsigned)pair_stride(n));
            }
            if constexpr (GEN) {
                // the emissive planes are lights too (at their point p), after the spheres; no pre-filter row for them
                const ConstF4 norow = pairv0 + (unsigned)(n < PTR_MAX_PAIR_LIGHTS ? n : PTR_MAX_PAIR_LIGHTS) * (unsigned)pair_stride(n);
                for (int q = 0; q < np; q++) {
                    const f4 p0 = plv0[2 * q];
                    const f4 e0 = *reinterpret_cast<ConstF4>((uintptr_t)(gplm32 + q) + 16);    // er eg eb emax
                    arrive(p0); arrive(e0);
                    if (!(e0.x > 0.0f || e0.y > 0.0f || e0.z > 0.0f)) continue;
                    light(n + q, p0.x, p0.y, p0.z, e0.x, e0.y, e0.z, norow);
                }
            }
            ax = fmaf(r.w.x, dx, ax); ay = fmaf(r.w.y, dy, ay); az = fmaf(r.w.z, dz, az);
        }
    }
    return V<float>{ax, ay, az};
}

// Philox4x32-10 (Salmon et al., SC'11); counter = (x, y, sample, 0), key = (seed, 'PTR!')
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t& r0, uint32_t& r1)
{
#pragma unroll
    for (int r = 0; r < 10; r++) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    r0 = c0; r1 = c1;
}

// sample s of pixel (x, y): the camera ray's direction (RayTracer.js:171-172, jittered for spp > 1)
template <typename T>
__device__ __forceinline__ V<T> ext_camera_dir(const RenderArgs& a, int x, int y, int s)
{
    const T inv_w = T(a.cam.inv_w), inv_h = T(a.cam.inv_h), angle = T(a.cam.angle), aspect = T(a.cam.aspect);
    T px = T(x) + T(0.5), py = T(y) + T(0.5);
    if (a.spp > 1) {
        uint32_t r0, r1;
        philox4x32_10((uint32_t)x, (uint32_t)y, (uint32_t)s, 0u, a.seed, 0x50545221u, r0, r1);
        px = T(x) + T((double(r0) + 0.5) / 4294967296.0);
        py = T(y) + T((double(r1) + 0.5) / 4294967296.0);
    }
    const T xx = (2 * (px * inv_w) - 1) * angle * aspect;      // RayTracer.js:171-172
    const T yy = (1 - 2 * (py * inv_h)) * angle;
    V<T> d;
    if (a.cam.general) {
        d.x = xx * T(a.cam.right[0]) + yy * T(a.cam.up[0]) + T(a.cam.fwd[0]);
        d.y = xx * T(a.cam.right[1]) + yy * T(a.cam.up[1]) + T(a.cam.fwd[1]);
        d.z = xx * T(a.cam.right[2]) + yy * T(a.cam.up[2]) + T(a.cam.fwd[2]);
    } else {
        d = V<T>{xx, yy, T(-1)};
    }
    normalize_v(d);
    return d;
}

template <typename T>
__device__ __forceinline__ void ext_store(const RenderArgs& a, int frame, int x, int row, V<T> sum)
{
    if (a.spp > 1) { const T k = T(a.spp); sum.x /= k; sum.y /= k; sum.z /= k; }
    const uint64_t at = frame * a.out_frame_stride + (uint64_t)a.width * row + x;
    a.out8[at] = pack_js((double)sum.x, (double)sum.y, (double)sum.z);
    if (a.out32f) {
        float* p = a.out32f + 4 * at;
        p[0] = (float)sum.x; p[1] = (float)sum.y; p[2] = (float)sum.z; p[3] = 1.0f;
    }
}

// (float32 here is the general form: scenes with planes or Phong terms; spheres alone take render_ext_f32_kernel)
template <typename T, bool STATS>
__device__ __forceinline__ void render_ext_pixel(const RenderArgs& a, int x, int row, ExtCounts* k)
{
    const int frame = blockIdx.z;
    const int y = global_y(a, row);
    SceneT<T> sc;
    sc.fv = frame_view(a.scenes + (uint64_t)(a.first_frame + frame) * a.frame_stride);
    // float32 works camera-relative: the ray starts at 0 there
    const V<T> origin = sizeof(T) == 8 ? V<T>{T(a.cam.org[0]), T(a.cam.org[1]), T(a.cam.org[2])} : V<T>{0, 0, 0};
    V<T> sum{0, 0, 0};
    for (int s = 0; s < a.spp; s++) {
        const V<T> d = ext_camera_dir<T>(a, x, y, s);
        xcount<STATS>(k, XT_CAMERA);
        const V<T> c = trace_ext<T, STATS>(sc, origin, d, a.max_depth, k);
        sum.x += c.x; sum.y += c.y; sum.z += c.z;
    }
    ext_store<T>(a, frame, x, row, sum);
}

template <typename T, bool STATS>
__global__ void __launch_bounds__(PTR_BLOCK) render_ext_kernel(RenderArgs a)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x = blockIdx.x * (int)(blockDim.x >> 3) + wave * 8 + (lane & 7);      // 8 x 8 pixels per wave, the waves side by side
    const int row = blockIdx.y * PTR_TILE_H + (lane >> 3);
    ExtCounts cnt;
    if (x < a.width && row < a.scan_height) render_ext_pixel<T, STATS>(a, x, row, &cnt);
    if constexpr (STATS) {
        // stats words: primary hits shadow exit_tca exit_d2 exit_hit | retraced reshaded | secondary
        __shared__ unsigned block_cnt[XT_N];
        if (threadIdx.x < XT_N) block_cnt[threadIdx.x] = 0u;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < XT_N; i++)
            if (cnt.c[i]) atomicAdd(&block_cnt[i], cnt.c[i]);
        __syncthreads();
        if (threadIdx.x < XT_N && block_cnt[threadIdx.x])
            atomicAdd(&a.stats[threadIdx.x == XT_SECONDARY ? 8 : threadIdx.x], (unsigned long long)block_cnt[threadIdx.x]);
    }
}

// The float32 form, one instance for scenes of spheres alone and one for scenes with planes or Phong terms (which costs
// registers the first never uses; trace_ext<float>, the same arithmetic lane by lane, is what both are checked against).
#ifndef PTR_EXT_WAVES
#define PTR_EXT_WAVES 7
#endif
#if PTR_EXT_WAVES > 0
#define PTR_EXT_OCC __attribute__((amdgpu_waves_per_eu(PTR_EXT_WAVES, PTR_EXT_WAVES)))
#else
#define PTR_EXT_OCC
#endif
// how a wave of the float32 form takes its pixels' samples: the one sample there is; all of them, at most one block; ONE block
// of several (blockIdx.z = frame * blocks + block); all blocks, block by block
enum { EXT_SPP_ONE = 0, EXT_SPP_ONE_BLOCK = 1, EXT_SPP_SPLIT = 2, EXT_SPP_BLOCKS = 3 };
template <bool GEN, int SPP>
__device__ __forceinline__ void render_ext_f32_body(const RenderArgs& a)
{
    const int lane = threadIdx.x;
    const int x = blockIdx.x * 8 + (lane & 7), row = blockIdx.y * PTR_TILE_H + (lane >> 3);
    const int frame = SPP == EXT_SPP_SPLIT ? (int)(blockIdx.z / (unsigned)a.ext_blocks) : (int)blockIdx.z;
    [[maybe_unused]] const int blk = SPP == EXT_SPP_SPLIT ? (int)(blockIdx.z - (unsigned)frame * (unsigned)a.ext_blocks) : 0;
    const unsigned char* block = a.scenes + (uint64_t)(a.first_frame + frame) * a.frame_stride;
    __shared__ float ext_lds[PTR_EXT_LDS_ENTRIES * 640];
    RayStack st;
    st.base = ext_lds + lane;
    // Which spheres can a camera ray of this wave's 8 x 8 tile reach?  The tile's rays lie in a cone around its centre ray
    // (half-angle bound from the host); lane i decides for sphere i from its CullRec (see ptr_kernels_fast.hip), and the
    // camera ray's pass over the spheres visits the set bits only: a floor or sky tile tests one sphere instead of all.
    unsigned long long cand = ~0ull;
    if (a.ext_cull_on && a.n <= 64) {
        const CullRec* gcull = reinterpret_cast<const CullRec*>(block + a.off_cull);
        const int li = lane < a.n ? lane : a.n;                  // (record n: zeros)
        const float4 g0 = *reinterpret_cast<const float4*>(&gcull[li].lx);   // lx ly lz s
        const float2 g1 = *reinterpret_cast<const float2*>(&gcull[li].re);   // re c
        const float xc = fmaf((float)(blockIdx.x * 8 + 4), a.camf.sx, a.camf.ox);
        const float yc = fmaf((float)(global_y(a, blockIdx.y * PTR_TILE_H) + PTR_TILE_H / 2), a.camf.sy, a.camf.oy);   // (consecutive rows)
        V<float> dc{xc, yc, -1.0f};
        if (a.cam.general) {
            dc.x = fmaf(xc, a.camf.right[0], fmaf(yc, a.camf.up[0], a.camf.fwd[0]));
            dc.y = fmaf(xc, a.camf.right[1], fmaf(yc, a.camf.up[1], a.camf.fwd[1]));
            dc.z = fmaf(xc, a.camf.right[2], fmaf(yc, a.camf.up[2], a.camf.fwd[2]));
        }
        const float inv = rsq(fmaf(dc.z, dc.z, fmaf(dc.y, dc.y, dc.x * dc.x)));
        const float bc = fmaf(g0.z, dc.z, fmaf(g0.y, dc.y, g0.x * dc.x)) * inv;
        const float K = fmaf(a.ext_cull_cos, g0.w, -fmaf(a.ext_cull_sin, g1.x, g1.y));
        cand = ballot(bc >= K) & (a.n >= 64 ? ~0ull : (1ull << a.n) - 1ull);
    }
    const int y = global_y(a, row);
    if (x >= a.width || row >= a.scan_height) return;            // (after the culling step: every lane speaks for a sphere there)
    V<float> sum{0, 0, 0};
    if constexpr (SPP == EXT_SPP_SPLIT) {
        // one block of samples per wave (a tile on a transparent sphere at 256 samples runs for a millisecond: with every
        // sample of a tile in one wave the launch ends in a long tail of such waves; a block is an eighth of that)
        const int s0 = blk * PTR_EXT_SPP_BLOCK, s1 = min(s0 + PTR_EXT_SPP_BLOCK, a.spp);
        for (int s = s0; s < s1; s++) {
            const V<float> d = ext_camera_dir<float>(a, x, y, s);
            const V<float> c = trace_ext_f32<GEN>(st, block, a, V<float>{0, 0, 0}, d, a.max_depth, cand);
            sum.x += c.x; sum.y += c.y; sum.z += c.z;
        }
        float4* out = reinterpret_cast<float4*>(a.ext_partial) +
                      (((uint64_t)frame * (uint64_t)a.ext_blocks + (uint64_t)blk) * (uint64_t)a.scan_height + (uint64_t)row) * (uint64_t)a.width + x;
        *out = make_float4(sum.x, sum.y, sum.z, 0.0f);
        return;
    } else if constexpr (SPP == EXT_SPP_BLOCKS) {
        // the same sums in the same order, all in this wave (callers that bring no buffer for the blocks' sums)
        V<float> bs{0, 0, 0};
        for (int s = 0; s < a.spp; s++) {
            const V<float> d = ext_camera_dir<float>(a, x, y, s);
            const V<float> c = trace_ext_f32<GEN>(st, block, a, V<float>{0, 0, 0}, d, a.max_depth, cand);
            bs.x += c.x; bs.y += c.y; bs.z += c.z;
            if ((s & (PTR_EXT_SPP_BLOCK - 1)) == PTR_EXT_SPP_BLOCK - 1 || s == a.spp - 1) {
                sum.x += bs.x; sum.y += bs.y; sum.z += bs.z;
                bs = V<float>{0, 0, 0};
            }
        }
    } else {
        const int spp = SPP == EXT_SPP_ONE ? 1 : a.spp;  // (one sample per pixel: a kernel of its own, as in the float64 form)
        for (int s = 0; s < spp; s++) {
            const V<float> d = ext_camera_dir<float>(a, x, y, s);
            const V<float> c = trace_ext_f32<GEN>(st, block, a, V<float>{0, 0, 0}, d, a.max_depth, cand);
            sum.x += c.x; sum.y += c.y; sum.z += c.z;
        }
    }
    ext_store<float>(a, frame, x, row, sum);
}

__global__ void __launch_bounds__(64) PTR_EXT_OCC render_ext_f32_kernel(RenderArgs a) { render_ext_f32_body<false, EXT_SPP_ONE_BLOCK>(a); }
// (its 50 VGPRs and the 5 120 B of LDS would allow 8 waves per SIMD: measured, 2 % slower - 12.22-12.25 against 11.95-12.01 ms per
// 256 frames at depth 5)
#ifndef PTR_EXT_ONE_WAVES
#define PTR_EXT_ONE_WAVES 7
#endif
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(PTR_EXT_ONE_WAVES, PTR_EXT_ONE_WAVES))) render_ext_f32_one_kernel(RenderArgs a) { render_ext_f32_body<false, EXT_SPP_ONE>(a); }
__global__ void __launch_bounds__(64) PTR_EXT_OCC render_ext_f32_split_kernel(RenderArgs a) { render_ext_f32_body<false, EXT_SPP_SPLIT>(a); }
__global__ void __launch_bounds__(64) PTR_EXT_OCC render_ext_f32_blocks_kernel(RenderArgs a) { render_ext_f32_body<false, EXT_SPP_BLOCKS>(a); }

// the blocks' sums of a pixel, added up in order, averaged and stored (what render_ext_f32_blocks_kernel does in registers)
__global__ void __launch_bounds__(256) ext_combine_kernel(RenderArgs a)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, row = blockIdx.y, frame = blockIdx.z;
    if (x >= a.width) return;
    const uint64_t plane = (uint64_t)a.scan_height * (uint64_t)a.width;
    const float4* in = reinterpret_cast<const float4*>(a.ext_partial) + (uint64_t)frame * (uint64_t)a.ext_blocks * plane + (uint64_t)row * (uint64_t)a.width + x;
    V<float> sum{0, 0, 0};
    for (int b = 0; b < a.ext_blocks; b++) {
        const float4 v = in[(uint64_t)b * plane];
        sum.x += v.x; sum.y += v.y; sum.z += v.z;
    }
    ext_store<float>(a, frame, x, row, sum);
}
#ifndef PTR_EXT_GEN_WAVES
#define PTR_EXT_GEN_WAVES 7
#endif
#if PTR_EXT_GEN_WAVES > 0
#define PTR_EXT_GEN_OCC __attribute__((amdgpu_waves_per_eu(PTR_EXT_GEN_WAVES, PTR_EXT_GEN_WAVES)))
#else
#define PTR_EXT_GEN_OCC
#endif
__global__ void __launch_bounds__(64) PTR_EXT_GEN_OCC render_ext_f32_gen_kernel(RenderArgs a) { render_ext_f32_body<true, EXT_SPP_ONE_BLOCK>(a); }
__global__ void __launch_bounds__(64) PTR_EXT_GEN_OCC render_ext_f32_gen_split_kernel(RenderArgs a) { render_ext_f32_body<true, EXT_SPP_SPLIT>(a); }
__global__ void __launch_bounds__(64) PTR_EXT_GEN_OCC render_ext_f32_gen_blocks_kernel(RenderArgs a) { render_ext_f32_body<true, EXT_SPP_BLOCKS>(a); }

// ---- the float64 specification in the wave-level form (scenes of spheres) ---------------------------------------------
// Bit for bit what trace_ext<double> computes - every lane performs the same float64 operations in the same order - but
// the wave gets through the two sphere loops the way the float32 form does:
//   * the sphere records (SphereG64: centre, radius^2) arrive by scalar loads on a wave-uniform index, one ahead of their use;
//   * Sphere.intersect's two early exits (Sphere.js:18-25) become one lane mask, the square root runs only when some lane hits;
//   * the camera ray's pass visits only the spheres its 8 x 8 tile can reach, the shadow loop skips the occluders whose
//     (light, occluder) cone the line through the light misses (PairQ).  Both culls only remove CERTAIN misses (their margins
//     are float32 error bands, orders of magnitude above float64 rounding), lanes whose N.L is not positive are not tested
//     (their term is (colour * tr) * (emission * 0) whatever tr is), and a blocked lane stops being tested: none of this
//     changes a bit;
//   * the waiting rays' lowest entries live in LDS, [entry][word][lane].
typedef double d4v __attribute__((ext_vector_type(4)));
typedef const volatile d4v PTR_CONST* ConstD4;
__device__ __forceinline__ void arrive(const d4v& q) { asm volatile("" ::"s"(q.x), "s"(q.y), "s"(q.z), "s"(q.w)); }
// normalize_v (Vector3.js:23-33) for all lanes of a wave
__device__ __forceinline__ void normalize_w(V<double>& v)
{
    const double len2 = v.x * v.x + v.y * v.y + v.z * v.z;
    const bool ok = len2 > 0;
    double inv;
    if (ballot(ok & !mid_range(len2)) == 0ull) inv = recip_mid(sqrt_mid(len2));     // (sqrt of a mid-range value is mid-range)
    else inv = 1.0 / __builtin_sqrt(len2);
    if (ok) { v.x *= inv; v.y *= inv; v.z *= inv; }
}
// sqrt(x) for the lanes of `want` (the others: any value)
__device__ __forceinline__ double sqrt_w(double x, bool want)
{
    if (ballot(want & !mid_range(x)) == 0ull) return sqrt_mid(x);
    return __builtin_sqrt(x);
}

// Waiting rays of a lane in LDS (4 864 B per wave and entry) x waves per SIMD, final kernel, 64 frames of the example scene at
// 1080p, A/B on one box: 2 x 4 6.85-6.88 ms at depth 5 / 4.84 at depth 3; 1 x 5 6.65-6.67 / 4.57-4.60 (the fifth wave is worth
// more than the second entry; several samples per pixel: no difference); bytes identical.
#ifndef PTR_EXT64_LDS_ENTRIES
#define PTR_EXT64_LDS_ENTRIES 1
#endif
struct RayStack64 {
    double* base;      // this lane's first double in LDS: 9 doubles per entry, 512 bytes apart
    int* dbase;        // ... and its depth word
    ExtRay<double> deep[PTR_EXT_MAX_DEPTH + 2 - PTR_EXT64_LDS_ENTRIES];
    __device__ __forceinline__ void push(int at, const ExtRay<double>& c)
    {
        if (at < PTR_EXT64_LDS_ENTRIES) {
            double* p = base + at * 576;
            p[0] = c.o.x; p[64] = c.o.y; p[128] = c.o.z; p[192] = c.d.x; p[256] = c.d.y; p[320] = c.d.z;
            p[384] = c.w.x; p[448] = c.w.y; p[512] = c.w.z;
            dbase[at * 64] = c.depth;
        } else {
            deep[at - PTR_EXT64_LDS_ENTRIES] = c;
        }
    }
    __device__ __forceinline__ ExtRay<double> pop(int at) const
    {
        if (at < PTR_EXT64_LDS_ENTRIES) {
            const double* p = base + at * 576;
            return ExtRay<double>{V<double>{p[0], p[64], p[128]}, V<double>{p[192], p[256], p[320]}, V<double>{p[384], p[448], p[512]},
                                  dbase[at * 64]};
        }
        return deep[at - PTR_EXT64_LDS_ENTRIES];
    }
};

// GEN: the scene may have planes (elements n, n + 1, ...) and Phong terms; a kernel of its own, as in the float32 form
template <bool GEN>
__device__ V<double> trace_ext_f64w(RayStack64& stack, const unsigned char* block, const RenderArgs& a, V<double> o0, V<double> d0, int max_depth,
                                    unsigned long long cand0)
{
    using T = double;
    typedef const volatile double PTR_CONST* ConstD;
    const ConstHeader hdr = (ConstHeader)(uintptr_t)block;
    const ConstD4 g64v = (ConstD4)(uintptr_t)(block + hdr->off_g64);         // cx cy cz r2
    const volatile double PTR_CONST* m64s = (const volatile double PTR_CONST*)(uintptr_t)(block + hdr->off_m64);   // sr sg sb er eg eb refl transp
    const SphereG64* g64 = reinterpret_cast<const SphereG64*>(block + hdr->off_g64);      // per-lane gathers of the hit sphere
    const SphereM64* m64 = reinterpret_cast<const SphereM64*>(block + hdr->off_m64);
    const ConstI32 lights = (ConstI32)(uintptr_t)(block + hdr->off_lights);
    const ConstF4 pairv0 = (ConstF4)(uintptr_t)(block + a.off_pairs);
    const int n = a.n, nl = hdr->n_lights;
    const V<T> bg{hdr->bg64[0], hdr->bg64[1], hdr->bg64[2]};
    [[maybe_unused]] const bool pair_always = hdr->pair_always != 0u;
    // planes: px py pz nx ny nz (PlaneG64), their materials (SphereM64's layout), the Phong terms (ks per element, then the exponents)
    const int np = GEN ? (int)hdr->n_planes : 0;
    const bool phong = GEN && hdr->has_phong != 0u;
    const ConstD pl64s = (ConstD)(uintptr_t)(block + (GEN ? hdr->off_pl64 : 0u));
    const ConstD plm64s = (ConstD)(uintptr_t)(block + (GEN ? hdr->off_plm64 : 0u));
    const PlaneG64* pl64 = reinterpret_cast<const PlaneG64*>(block + (GEN ? hdr->off_pl64 : 0u));       // per-lane gathers
    const SphereM64* plm64 = reinterpret_cast<const SphereM64*>(block + (GEN ? hdr->off_plm64 : 0u));
    const double* ph64 = reinterpret_cast<const double*>(block + (GEN ? hdr->off_ph64 : 0u));

    int top = 0;
    ExtRay<T> r{o0, d0, V<T>{1, 1, 1}, 0};
    bool have = true;
    bool first = true;
    V<T> acc{0, 0, 0};
#ifdef PTR_EXT64_UTIL
    unsigned u_lane = 0u, u_wave = 0u, u_blane = 0u, u_bwave = 0u;      // ray steps of this lane / of its wave, those that bounce
#endif
    while (have || top > 0) {
        if (!have) r = stack.pop(--top);
        have = false;
        T tnear = T(PTR_INFINITY);
        int e = -1;
#ifdef PTR_EXT64_UTIL
        u_lane++;
#endif
        if (n > 0) {
            // Sphere.intersect (Sphere.js:15-33) + RayTracer.js:32-36 for one sphere and all lanes
            auto test = [&](const d4v q, int i) {
                const T lx = q.x - r.o.x, ly = q.y - r.o.y, lz = q.z - r.o.z;
                const T tca = lx * r.d.x + ly * r.d.y + lz * r.d.z;
                if (ballot(!(tca < 0)) == 0ull) return;                       // the sphere lies behind every ray of the wave (Sphere.js:18-19)
                const T d2 = (lx * lx + ly * ly + lz * lz) - tca * tca;
                const bool hit = !(tca < 0) & !(d2 > q.w);
                if (ballot(hit) != 0ull) {
                    const T thc = sqrt_w(q.w - d2, hit);
                    T t0 = tca - thc;
                    const T t1 = tca + thc;
                    t0 = t0 < 0 ? t1 : t0;
                    const bool nearer = hit & (t0 < tnear);
                    tnear = nearer ? t0 : tnear;
                    e = nearer ? i : e;
                }
            };
            if (cand0 != ~0ull && __builtin_amdgcn_readfirstlane((int)first) != 0) {
                for (unsigned long long c = cand0; c != 0ull; c &= c - 1ull) {
                    const int i = __builtin_ctzll(c);
                    const d4v q = g64v[i];
                    arrive(q);
                    test(q, i);
                }
            } else {
                d4v A = g64v[0], B;
                for (int i = 0;;) {
                    arrive(A); B = g64v[i + 1 < n ? i + 1 : i]; test(A, i);
                    if (++i >= n) break;
                    arrive(B); A = g64v[i + 1 < n ? i + 1 : i]; test(B, i);
                    if (++i >= n) break;
                }
            }
        }
        if constexpr (GEN) {
            // Plane.intersect of the specification: denom = d.n; 0 -> miss; t = ((p - o).n) / denom; t < 0 -> miss
            for (int q = 0; q < np; q++) {
                const ConstD pl = pl64s + 6 * q;
                const T ppx = pl[0], ppy = pl[1], ppz = pl[2], pnx = pl[3], pny = pl[4], pnz = pl[5];
                const T denom = r.d.x * pnx + r.d.y * pny + r.d.z * pnz;
                const T t = ((ppx - r.o.x) * pnx + (ppy - r.o.y) * pny + (ppz - r.o.z) * pnz) / denom;
                const bool nearer = !(denom == 0) & !(t < 0) & (t < tnear);
                tnear = nearer ? t : tnear;
                e = nearer ? n + q : e;
            }
        }
        first = false;
        if (e < 0) {
            acc.x += r.w.x * bg.x; acc.y += r.w.y * bg.y; acc.z += r.w.z * bg.z;
            continue;
        }
        const bool on_plane = GEN && e >= n;
        SphereM64 m;
        V<T> N;
        if (on_plane) m = plm64[e - n]; else m = m64[e];
        const V<T> em{m.er, m.eg, m.eb}, col{m.sr, m.sg, m.sb};
        acc.x += r.w.x * em.x; acc.y += r.w.y * em.y; acc.z += r.w.z * em.z;
        const V<T> P{r.o.x + r.d.x * tnear, r.o.y + r.d.y * tnear, r.o.z + r.d.z * tnear};
        if (on_plane) {
            const PlaneG64 g = pl64[e - n];
            N = V<T>{g.nx, g.ny, g.nz};                             // a plane's normal is the unit vector it was given
        } else {
            const SphereG64 ce = g64[e];
            N = V<T>{P.x - ce.cx, P.y - ce.cy, P.z - ce.cz};
            normalize_w(N);
        }
        bool inside = false;
        T dn = r.d.x * N.x + r.d.y * N.y + r.d.z * N.z;
        if (dn > 0) { N.x = -N.x; N.y = -N.y; N.z = -N.z; inside = true; dn = -dn; }
        const T transp = m.transp, refl = m.refl;
        if ((transp > 0 || refl > 0) && r.depth < max_depth) {
#ifdef PTR_EXT64_UTIL
            u_blane++;
#endif
            const T facing = -dn;
            const T f = 1 - facing;
            const T fresnel = T(1) * T(0.1) + (f * f * f) * (T(1) - T(0.1));
            const V<T> wsc{r.w.x * col.x, r.w.y * col.y, r.w.z * col.z};
            if (transp > 0) {
                const T eta = inside ? T(1.1) : T(1) / T(1.1);
                const T cosi = -dn;
                const T kk2 = 1 - eta * eta * (1 - cosi * cosi);
                if (kk2 >= 0) {
                    ExtRay<T> c;
                    const T g = eta * cosi - sqrt_w(kk2, true);
                    const T wt = (1 - fresnel) * transp;
                    c.o = V<T>{P.x - N.x * T(PTR_BIAS), P.y - N.y * T(PTR_BIAS), P.z - N.z * T(PTR_BIAS)};
                    c.d = V<T>{r.d.x * eta + N.x * g, r.d.y * eta + N.y * g, r.d.z * eta + N.z * g};
                    c.w = V<T>{wsc.x * wt, wsc.y * wt, wsc.z * wt};
                    normalize_w(c.d);
                    c.depth = r.depth + 1;
                    stack.push(top++, c);
                }
            }
            const T two = 2 * dn;
            ExtRay<T> c;
            c.o = V<T>{P.x + N.x * T(PTR_BIAS), P.y + N.y * T(PTR_BIAS), P.z + N.z * T(PTR_BIAS)};
            c.d = V<T>{r.d.x - N.x * two, r.d.y - N.y * two, r.d.z - N.z * two};
            c.w = V<T>{wsc.x * fresnel, wsc.y * fresnel, wsc.z * fresnel};
            normalize_w(c.d);
            c.depth = r.depth + 1;
            r = c;
            have = true;
        } else {
            // the reference's direct lighting (RayTracer.js:60-91)
            const V<T> so{P.x + N.x * T(PTR_BIAS), P.y + N.y * T(PTR_BIAS), P.z + N.z * T(PTR_BIAS)};
            V<T> dif{0, 0, 0};
            // The pair table may only cull for origins near enough to the camera: farther out the float64 test below rounds
            // by more than the margin the table leaves, and that rounding is part of the specification's result
            // (FrameHeader::pair_smax2; hit points on a plane can lie anywhere).
            // Scenes of spheres alone: hit points lie on the spheres, the host has decided for the whole frame (pair_always).
            [[maybe_unused]] unsigned long long pair_ok_m = ~0ull;
            if constexpr (GEN) {
                const T sox = so.x - a.cam.org[0], soy = so.y - a.cam.org[1], soz = so.z - a.cam.org[2];
                pair_ok_m = ballot((float)(sox * sox + soy * soy + soz * soz) <= hdr->pair_smax2);
            }
            // Phong term of the specification: ks * max(0, R.L)^exponent per light, R = d - 2 (d.N) N
            T ks = 0;
            V<T> Rv{0, 0, 0};
            unsigned pexp = 1u;
            if constexpr (GEN) if (phong) {
                ks = ph64[e];
                pexp = (unsigned)ph64[n + np + e];
                const T two_dn = 2 * dn;
                Rv = V<T>{r.d.x - N.x * two_dn, r.d.y - N.y * two_dn, r.d.z - N.z * two_dn};
            }
            // lights: the emissive spheres in scene order (the host's list), then the emissive planes
            for (int kk = 0; kk < nl + np; kk++) {
                int i;
                V<T> Ci, ei;
                if (!GEN || kk < nl) {
                    i = lights[kk];
                    const d4v Cq = g64v[i];
                    const ConstD mi = m64s + 8 * i;                                       // (three words of the light's record, not all eight: SGPRs)
                    ei = V<T>{mi[3], mi[4], mi[5]};
                    arrive(Cq);
                    Ci = V<T>{Cq.x, Cq.y, Cq.z};
                } else {
                    i = n + (kk - nl);
                    const ConstD mi = plm64s + 8 * (kk - nl);
                    ei = V<T>{mi[3], mi[4], mi[5]};
                    if (!(ei.x > 0 || ei.y > 0 || ei.z > 0)) continue;
                    const ConstD pl = pl64s + 6 * (kk - nl);
                    Ci = V<T>{pl[0], pl[1], pl[2]};
                }
                V<T> L{Ci.x - P.x, Ci.y - P.y, Ci.z - P.z};
                [[maybe_unused]] const T len2 = L.x * L.x + L.y * L.y + L.z * L.z;         // a plane only blocks in front of the light
                normalize_w(L);
                T ratio = N.x * L.x + N.y * L.y + N.z * L.z;
                [[maybe_unused]] T sp = 0;
                bool shines = false;                                                      // the lane has a Phong term that can be non-zero
                if constexpr (GEN) {
                    if (ballot(ks > 0) != 0ull) {
                        sp = Rv.x * L.x + Rv.y * L.y + Rv.z * L.z;
                        shines = (ks > 0) & (!(sp <= 0) | (pexp == 0u));
                    }
                }
                // lanes whose terms can be non-zero still need a verdict, as a lane mask
                unsigned long long open_m = ballot((ratio > 0) | shines);
                const unsigned long long asked_m = open_m;
                if (open_m != 0ull) {
                    const float Lx = (float)L.x, Ly = (float)L.y, Lz = (float)L.z;       // for the pre-filter alone
                    const bool cull_ok = GEN ? (open_m & ~pair_ok_m) == 0ull : pair_always;
                    const int norow = n < PTR_MAX_PAIR_LIGHTS ? n : PTR_MAX_PAIR_LIGHTS;  // (the never-cull row: plane lights, lights past the table)
                    const int prow = (kk < nl && kk < PTR_MAX_PAIR_LIGHTS) ? kk : norow;
                    const ConstF4 pairs = pairv0 + (unsigned)prow * (unsigned)pair_stride(n);
                    for (int j = 0; j < n && open_m != 0ull; j++) {
                        if (j == i) continue;                                             // RayTracer.js:76
                        const f4 pq = pairs[j];
                        const d4v q = g64v[j];
                        arrive(pq);
                        const float c = fmaf(pq.z, Lz, fmaf(pq.y, Ly, pq.x * Lx));
#ifndef PTR_EXT64_NO_PAIRCULL
                        if (cull_ok && (ballot(fabsf(c) + 1e-5f >= pq.w) & open_m) == 0ull) continue;
#endif
                        arrive(q);
                        const T lx = q.x - so.x, ly = q.y - so.y, lz = q.z - so.z;
                        const T tca = lx * L.x + ly * L.y + lz * L.z;
                        if ((ballot(!(tca < 0)) & open_m) == 0ull) continue;              // behind every open lane's ray
                        const T d2 = (lx * lx + ly * ly + lz * lz) - tca * tca;
                        open_m &= ~ballot(!(tca < 0) & !(d2 > q.w));                      // any hit blocks; distance ignored (F7)
                    }
                    if constexpr (GEN) {
                        for (int j = 0; j < np && open_m != 0ull; j++) {
                            if (n + j == i) continue;
                            const ConstD pl = pl64s + 6 * j;
                            const T ppx = pl[0], ppy = pl[1], ppz = pl[2], pnx = pl[3], pny = pl[4], pnz = pl[5];
                            const T denom = L.x * pnx + L.y * pny + L.z * pnz;
                            const T t = ((ppx - so.x) * pnx + (ppy - so.y) * pny + (ppz - so.z) * pnz) / denom;
                            open_m &= ~ballot(!(denom == 0) & !(t < 0) & (t * t < len2));
                        }
                    }
                }
                const T tr = lane_of(asked_m & ~open_m) ? T(0) : T(1);
                ratio = (ratio != ratio) ? ratio : (ratio > 0 ? ratio : T(0));           // Math.max(0, .)
                dif.x += (col.x * tr) * (ei.x * ratio);
                dif.y += (col.y * tr) * (ei.y * ratio);
                dif.z += (col.z * tr) * (ei.z * ratio);
                if constexpr (GEN) {
                    if (ks > 0) {
                        sp = (sp != sp) ? sp : (sp > 0 ? sp : T(0));
                        T ph = 1, b = sp;
                        for (unsigned q = pexp; q > 0; q >>= 1) { if (q & 1u) ph = ph * b; b = b * b; }
                        const T kp = ks * ph;
                        dif.x += tr * (ei.x * kp);
                        dif.y += tr * (ei.y * kp);
                        dif.z += tr * (ei.z * kp);
                    }
                }
            }
            acc.x += r.w.x * dif.x; acc.y += r.w.y * dif.y; acc.z += r.w.z * dif.z;
        }
    }
#ifdef PTR_EXT64_UTIL
    // measurement build (tools/ext64_util.py): the pixel's value is its counts, not its colour.  A lane takes part in every pass
    // of the loop until its own tree is done, so the wave's passes are the largest count of its lanes.
    u_wave = u_lane; u_bwave = u_blane;
    for (int sh = 1; sh < 64; sh <<= 1) {
        const unsigned o1 = (unsigned)__shfl_xor((int)u_wave, sh), o2 = (unsigned)__shfl_xor((int)u_bwave, sh);
        u_wave = o1 > u_wave ? o1 : u_wave; u_bwave = o2 > u_bwave ? o2 : u_bwave;
    }
    acc.x = (double)(u_lane | (u_wave << 8) | (u_blane << 16) | (u_bwave << 24));
#endif
    return acc;
}

#ifndef PTR_EXT64_WAVES
#define PTR_EXT64_WAVES 5
#endif
// ONE: one sample per pixel, a kernel of its own (the sample loop keeps the camera's 30 SGPRs alive through the whole ray tree:
// with it the kernel spills scalar registers)
template <bool ONE, bool GEN>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(PTR_EXT64_WAVES, PTR_EXT64_WAVES))) render_ext_f64_kernel(RenderArgs a)
{
    const int lane = threadIdx.x;
    const int x = blockIdx.x * 8 + (lane & 7), row = blockIdx.y * PTR_TILE_H + (lane >> 3), frame = blockIdx.z;
    const unsigned char* block = a.scenes + (uint64_t)(a.first_frame + frame) * a.frame_stride;
    __shared__ double ext_lds_d[PTR_EXT64_LDS_ENTRIES * 576];
    __shared__ int ext_lds_i[PTR_EXT64_LDS_ENTRIES * 64];
    RayStack64 st;
    st.base = ext_lds_d + lane;
    st.dbase = ext_lds_i + lane;
    // the camera ray's candidates: the tile culling of render_ext_f32_body (float32 records, float32 margins: certain misses)
    unsigned long long cand = ~0ull;
#ifdef PTR_EXT64_NO_TILECULL
    if (false) {
#else
    if (a.ext_cull_on && a.n <= 64) {
#endif
        const CullRec* gcull = reinterpret_cast<const CullRec*>(block + a.off_cull);
        const int li = lane < a.n ? lane : a.n;
        const float4 g0 = *reinterpret_cast<const float4*>(&gcull[li].lx);
        const float2 g1 = *reinterpret_cast<const float2*>(&gcull[li].re);
        const float xc = fmaf((float)(blockIdx.x * 8 + 4), a.camf.sx, a.camf.ox);
        const float yc = fmaf((float)(global_y(a, blockIdx.y * PTR_TILE_H) + PTR_TILE_H / 2), a.camf.sy, a.camf.oy);
        V<float> dc{xc, yc, -1.0f};
        if (a.cam.general) {
            dc.x = fmaf(xc, a.camf.right[0], fmaf(yc, a.camf.up[0], a.camf.fwd[0]));
            dc.y = fmaf(xc, a.camf.right[1], fmaf(yc, a.camf.up[1], a.camf.fwd[1]));
            dc.z = fmaf(xc, a.camf.right[2], fmaf(yc, a.camf.up[2], a.camf.fwd[2]));
        }
        const float inv = rsq(fmaf(dc.z, dc.z, fmaf(dc.y, dc.y, dc.x * dc.x)));
        const float bc = fmaf(g0.z, dc.z, fmaf(g0.y, dc.y, g0.x * dc.x)) * inv;
        const float K = fmaf(a.ext_cull_cos, g0.w, -fmaf(a.ext_cull_sin, g1.x, g1.y));
        cand = ballot(bc >= K) & (a.n >= 64 ? ~0ull : (1ull << a.n) - 1ull);
    }
    const int y = global_y(a, row);
    if (x >= a.width || row >= a.scan_height) return;
    const V<double> origin{a.cam.org[0], a.cam.org[1], a.cam.org[2]};
    V<double> sum{0, 0, 0};
    const int spp = ONE ? 1 : a.spp;
    for (int s = 0; s < spp; s++) {
        const V<double> d = ext_camera_dir<double>(a, x, y, s);
        const V<double> c = trace_ext_f64w<GEN>(st, block, a, origin, d, a.max_depth, cand);
        sum.x += c.x; sum.y += c.y; sum.z += c.z;
    }
#ifdef PTR_EXT64_UTIL
    a.out8[frame * a.out_frame_stride + (uint64_t)a.width * row + x] = (unsigned)sum.x;
#else
    ext_store<double>(a, frame, x, row, sum);
#endif
}

hipError_t launch_render_ext(const RenderArgs& a, int n_frames, bool f64, bool stats, bool spheres_only, bool lanewise, hipStream_t s)
{
    dim3 grid((a.width + PTR_TILE_W - 1) / PTR_TILE_W, (a.scan_height + PTR_TILE_H - 1) / PTR_TILE_H, n_frames);
    if (stats) hipLaunchKernelGGL((render_ext_kernel<double, true>), grid, dim3(PTR_BLOCK), 0, s, a);
    // (one wave per workgroup, as below: 4-5 % faster than four; more waves per SIMD than the 4 its 120 VGPRs allow cost
    // spills: 5 waves +15 %, 6 waves +45 %)
    else if (f64 && !lanewise) {
        const auto kernel = spheres_only ? (a.spp == 1 ? render_ext_f64_kernel<true, false> : render_ext_f64_kernel<false, false>)
                                         : (a.spp == 1 ? render_ext_f64_kernel<true, true> : render_ext_f64_kernel<false, true>);
        hipLaunchKernelGGL(kernel, dim3((a.width + 7) / 8, grid.y, n_frames), dim3(64), 0, s, a);
    }
    else if (f64) hipLaunchKernelGGL((render_ext_kernel<double, false>), dim3((a.width + 7) / 8, grid.y, n_frames), dim3(64), 0, s, a);
    else {
        // one wave per workgroup: the waves' run times differ by an order of magnitude (a tile on a transparent sphere
        // expands a tree of rays), and a one-wave workgroup starts wherever a slot is free
        const dim3 grid1((a.width + 7) / 8, grid.y, n_frames);
        if (a.spp > PTR_EXT_SPP_BLOCK && a.ext_partial) {
            // a wave per (tile, block of samples), then the blocks' sums in order
            const dim3 gridb(grid1.x, grid1.y, (unsigned)n_frames * (unsigned)a.ext_blocks);
            hipLaunchKernelGGL(spheres_only ? render_ext_f32_split_kernel : render_ext_f32_gen_split_kernel, gridb, dim3(64), 0, s, a);
            hipLaunchKernelGGL(ext_combine_kernel, dim3((a.width + 255) / 256, a.scan_height, n_frames), dim3(256), 0, s, a);
        } else if (a.spp > PTR_EXT_SPP_BLOCK) {
            hipLaunchKernelGGL(spheres_only ? render_ext_f32_blocks_kernel : render_ext_f32_gen_blocks_kernel, grid1, dim3(64), 0, s, a);
        } else if (spheres_only) {
            hipLaunchKernelGGL(a.spp == 1 ? render_ext_f32_one_kernel : render_ext_f32_kernel, grid1, dim3(64), 0, s, a);
        } else {
            hipLaunchKernelGGL(render_ext_f32_gen_kernel, grid1, dim3(64), 0, s, a);
        }
    }
    return hipGetLastError();
}

}  // namespace ptr
