// ptr_kernels_fast.hip — the float32 kernel: one camera ray per lane, 8x8 pixels per wave.
//
// What it computes is RayTracer.render's per-pixel body (RT/src/RayTracer.js:163-193 ->
// trace :18-95 -> Sphere.intersect, RT/src/Sphere.js:15-33), restructured for the machine:
//   * the sphere loops run on a wave-uniform index, so sphere data arrives through scalar
//     loads and sits in SGPRs: no vector memory, no LDS, no bank conflicts;
//   * the geometry is camera-relative and the per-sphere terms |l|^2 - r^2 etc. are computed
//     once in float64 on the host, so the primary test is 3 FMAs + 1 FMA;
//   * everything past the hit/miss test of a sphere is skipped by the whole wave when no lane
//     hits it (exec-mask branch), the shadow loop leaves as soon as every lane is decided.
//
// Every discontinuous decision (hit/miss, which sphere is nearest, shadowed or not, the
// 8-bit rounding) carries a conservative "not sure" band.  In HYBRID mode a pixel with any
// decision inside its band is appended to a queue and re-traced afterwards by the float64
// kernel (ptr_kernels_exact.hip); all other pixels already hold the reference's bytes.
//
// The bands are wide for the geometric predicates (they only catch pixels on 1-D curves:
// silhouettes, shadow edges, sphere/sphere contacts) and tight for the rounding test (which
// can catch any pixel), which is why the winner's distance is recomputed with the
// cancellation-free root t = cc / (b + thc).
#include <hip/hip_runtime.h>
#include "ptr_device.h"
#include "ptr_kernels.h"

namespace ptr {

// ---- error-bound constants (EPS = float32 unit roundoff 2^-24) ----------------------------
constexpr float EPS = 5.9604644775390625e-08f;
constexpr float FAR_UNSURE = 9.0e7f;     // anything this close to the 1e8 far clip is re-traced

struct F3 { float x, y, z; };

__device__ __forceinline__ float dot3(F3 a, F3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
__device__ __forceinline__ float rsq(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ float rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// why a pixel was queued (debug build of the kernel only: PTR_DEBUG_REASONS)
enum : unsigned { WHY_PRIMARY = 1, WHY_CC = 2, WHY_ORDER = 4, WHY_FAR = 8, WHY_SHADOW = 16, WHY_ROUND = 32 };

template <bool DBG> struct Unsure;
template <> struct Unsure<false> {
    bool any = false;
    __device__ __forceinline__ void add(bool c, unsigned) { any |= c; }
    __device__ __forceinline__ unsigned why() const { return 0u; }
};
template <> struct Unsure<true> {
    bool any = false;
    unsigned bits = 0u;
    __device__ __forceinline__ void add(bool c, unsigned bit) { any |= c; bits |= c ? bit : 0u; }
    __device__ __forceinline__ unsigned why() const { return bits; }
};

template <bool HYBRID, bool DBG>
__global__ void __launch_bounds__(PTR_BLOCK) render_fast_kernel(RenderArgs a)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x = blockIdx.x * PTR_TILE_W + wave * 8 + (lane & 7);
    const int row = blockIdx.y * PTR_TILE_H + (lane >> 3);
    if (x >= a.width || row >= a.scan_height) return;

    const int frame = blockIdx.z;
    const uint64_t frame_px = (uint64_t)a.width * (uint64_t)a.scan_height;
    uint32_t* out8 = a.out8 + frame * a.out_frame_stride;
    const bool quirk = (a.flags & 0x0001u) != 0;      // PTR_QUIRK_ROW_SHIFT
    long long idx_out = (long long)a.width * row + x;
    if (quirk) {
        if (row == 0) {                               // RayTracer.js:169: dropped row / untouched last row
            out8[(uint64_t)a.width * (a.scan_height - 1) + x] = 0u;
            if (!a.out32f) return;
        }
        idx_out -= a.width;
    }

    const unsigned char* block = a.scenes + (uint64_t)(a.first_frame + frame) * a.frame_stride;
    const FrameView fv = frame_view(block);
    const int n = fv.h->n;

    // ---- camera ray (RayTracer.js:171-174) ------------------------------------------------
    const float xx = fmaf((float)x + 0.5f, a.camf.sx, a.camf.ox);
    const float yy = fmaf((float)global_y(a, row) + 0.5f, a.camf.sy, a.camf.oy);
    F3 d;
    if (a.cam.general) {
        d.x = fmaf(xx, a.camf.right[0], fmaf(yy, a.camf.up[0], a.camf.fwd[0]));
        d.y = fmaf(xx, a.camf.right[1], fmaf(yy, a.camf.up[1], a.camf.fwd[1]));
        d.z = fmaf(xx, a.camf.right[2], fmaf(yy, a.camf.up[2], a.camf.fwd[2]));
    } else {
        d = F3{xx, yy, -1.0f};
    }
    {
        const float inv = rsq(dot3(d, d));
        d.x *= inv; d.y *= inv; d.z *= inv;
    }

    // ---- nearest hit (RayTracer.js:26-41), wave-uniform loop ---------------------------------
    // disc = b^2 - cc is cheap (cc is per sphere) but cancels; it only has to ORDER the hits here.
    Unsure<DBG> unsure;
    float tnear = 3.0e38f, t2 = 3.0e38f, etmax = 0.0f;
    int idx = -1;
    for (int i = 0; i < n; i++) {
        const SphereP32* ps = fv.p32 + i;              // wave-uniform -> scalar loads
        const float b = fmaf(ps->lz, d.z, fmaf(ps->ly, d.y, ps->lx * d.x));
        const float disc = fmaf(b, b, -ps->cc);
        const float m = fminf(b * ps->kb, disc);       // hit <=> b >= 0 && disc >= 0 <=> m >= 0
        unsure.add(fabsf(m) <= ps->md, WHY_PRIMARY);
        if (m >= 0.0f) {                               // skipped by the wave when no lane hits
            const float s = rsq(disc);
            const float t = fmaf(ps->sgn, disc * s, b);   // t0, or t1 when t0 < 0 (RayTracer.js:32-34)
            etmax = fmaxf(etmax, fmaf(ps->hmd, s, ps->eb));
            t2 = __builtin_amdgcn_fmed3f(tnear, t2, t);   // second smallest so far
            idx = t < tnear ? i : idx;
            tnear = fminf(tnear, t);
            unsure.add(ps->unsure != 0.0f, WHY_CC);
        }
    }
    // nearest-vs-second ordering, far clip (t < 1e8, RayTracer.js:19,36)
    unsure.add((t2 - tnear <= 2.0f * etmax) & (idx >= 0), WHY_ORDER);
    unsure.add((tnear >= FAR_UNSURE) & (idx >= 0), WHY_FAR);

    float cr, cg, cb;          // unclamped colour
    uint32_t packed;
    if (idx < 0) {
        // miss: the background, bit-exact by construction (RayTracer.js:43-46)
        cr = fv.h->bg32[0]; cg = fv.h->bg32[1]; cb = fv.h->bg32[2];
        packed = fv.h->bg_packed;
    } else {
        // ---- winner, recomputed accurately (per-lane gather of its record) ---------------------
        // disc' = r^2 - |l x d|^2 has no cancellation between |l|^2 and b^2, and the nearest root
        // is taken in its cancellation-free form t = cc / (b + thc).
        const float4 g0 = *reinterpret_cast<const float4*>(&fv.p32[idx].lx);    // lx ly lz cc
        const float4 g2 = *reinterpret_cast<const float4*>(&fv.p32[idx].hmd);   // hmd eb r2 lnorm
        const float2 g3 = *reinterpret_cast<const float2*>(&fv.p32[idx].inv_r); // inv_r hed
        const SphereM32 mw = fv.m32[idx];
        const F3 lw{g0.x, g0.y, g0.z};
        const float cc_w = g0.w, eb_w = g2.y, r2_w = g2.z, inv_r = g3.x, hed_w = g3.y;
        const float b_w = dot3(lw, d);
        const F3 cr3{fmaf(lw.y, d.z, -lw.z * d.y), fmaf(lw.z, d.x, -lw.x * d.z), fmaf(lw.x, d.y, -lw.y * d.x)};
        const float disc_w = fmaxf(r2_w - dot3(cr3, cr3), 1e-30f);
        const float s_w = rsq(disc_w);
        const float thc_w = disc_w * s_w;
        const float et_w = fmaf(hed_w, s_w, eb_w);     // |error of (b + thc)|
        const float bt = b_w + thc_w;
        const float rbt = rcp(bt);
        const bool outside = cc_w > 0.0f;
        const float t = outside ? cc_w * rbt : bt;
        const float eP = outside ? t * fmaf(et_w, rbt, 4.0f * EPS) : fmaf(2.0f * EPS, t, et_w);   // |error of P|

        // ---- hit frame (RayTracer.js:48-58), camera-relative ---------------------------------
        const F3 P{d.x * t, d.y * t, d.z * t};
        F3 N{P.x - lw.x, P.y - lw.y, P.z - lw.z};      // P - C (Sphere.js:47-51)
        {
            const float inv = rsq(dot3(N, N));
            const float sg = outside ? inv : -inv;     // inside: d.N > 0 -> revert (RayTracer.js:54-58)
            N.x *= sg; N.y *= sg; N.z *= sg;
        }
        const float eN = fmaf(eP + EPS * (g2.w + t), inv_r, 3.0f * EPS);   // |error of N|
        const F3 so{fmaf(N.x, (float)PTR_BIAS, P.x), fmaf(N.y, (float)PTR_BIAS, P.y), fmaf(N.z, (float)PTR_BIAS, P.z)};

        // ---- direct light + binary shadow (RayTracer.js:61-91) ---------------------------------
        float ar = 0.0f, ag = 0.0f, ab = 0.0f;
        float wsum = 0.0f, eLmax = 0.0f;
        const int nl = fv.h->n_lights;
        for (int k = 0; k < nl; k++) {
            const int i = fv.lights[k];
            const SphereP32* pl = fv.p32 + i;
            const SphereM32* ml = fv.m32 + i;
            F3 L{pl->lx - P.x, pl->ly - P.y, pl->lz - P.z};
            const float invl = rsq(dot3(L, L));
            L.x *= invl; L.y *= invl; L.z *= invl;
            const float eL = fmaf(fmaf(EPS, pl->lnorm + t, eP), invl, 3.0f * EPS);   // |error of L|
            const float ratio = dot3(N, L);
            bool open = ratio > 0.0f;                  // lanes that still need a shadow verdict
            bool maybe = false, blocked32 = false;
            const float kL = fmaf(2.0f, eL, 10.0f * EPS);
            for (int j = 0; j < n; j++) {
                if (j == i) continue;
                if (__builtin_amdgcn_ballot_w64(open) == 0ull) break;
                const SphereP32* pj = fv.p32 + j;
                const F3 lj{pj->lx - so.x, pj->ly - so.y, pj->lz - so.z};
                const float bj = dot3(lj, L);
                const float ll = dot3(lj, lj);
                const float disc = fmaf(bj, bj, pj->r2 - ll);
                // error bounds: |lj| <= A, d(lj) <= dlj, d(bj) <= ebj, d(disc) <= band
                const float A = pj->lnorm + t + 2.0e-4f;
                const float dlj = fmaf(EPS, A, eP);
                const float band = fmaf(ll, kL, 4.0f * A * dlj);
                const float ebj = fmaf(A, eL + 3.0f * EPS, dlj);
                const bool certain = (bj > ebj) & (disc > band);   // any-hit for sure (RayTracer.js:77-81)
                maybe |= open & (bj >= -ebj) & (disc >= -band) & !certain;
                blocked32 |= open & (bj >= 0.0f) & (disc >= 0.0f);  // the float32 verdict
                open &= !certain;
            }
            // A "maybe" only matters if no certain hit blocks the light anyway.
            unsure.add(open & maybe, WHY_SHADOW);
            open &= !blocked32;
            if (open) {
                ar = fmaf(mw.sr * ml->er, ratio, ar);
                ag = fmaf(mw.sg * ml->eg, ratio, ag);
                ab = fmaf(mw.sb * ml->eb, ratio, ab);
                wsum += ml->emax;
                eLmax = fmaxf(eLmax, eL);
            }
        }
        cr = ar + mw.er; cg = ag + mw.eg; cb = ab + mw.eb;   // RayTracer.js:93

        // ---- clamp, round, pack (RayTracer.js:179-192) with a rounding band ------------------
        // |error of N.L| <= eN + eL (+ rounding of the dot and of the sums, 8 eps)
        const float band = 255.0f * fmaf(mw.scmax * wsum, eN + eLmax + 8.0f * EPS, 4.0f * EPS * mw.emax) + 2e-5f;
        const float vr = fminf(cr, 1.0f) * 255.0f, vg = fminf(cg, 1.0f) * 255.0f, vb = fminf(cb, 1.0f) * 255.0f;
        const float fr = floorf(vr), fg = floorf(vg), fb = floorf(vb);
        const float qr = vr - fr, qg = vg - fg, qb = vb - fb;
        unsure.add((fabsf(qr - 0.5f) <= band) | (fabsf(qg - 0.5f) <= band) | (fabsf(qb - 0.5f) <= band), WHY_ROUND);
        const uint32_t r8 = (uint32_t)(qr >= 0.5f ? fr + 1.0f : fr);
        const uint32_t g8 = (uint32_t)(qg >= 0.5f ? fg + 1.0f : fg);
        const uint32_t b8 = (uint32_t)(qb >= 0.5f ? fb + 1.0f : fb);
        packed = (255u << 24) | (b8 << 16) | (g8 << 8) | r8;
    }

    if constexpr (DBG) packed = unsure.why();
    if (idx_out >= 0) out8[idx_out] = packed;
    if (a.out32f) {
        float4* p = reinterpret_cast<float4*>(a.out32f) + (frame * a.out_frame_stride + (uint64_t)a.width * row + x);
        *p = make_float4(cr, cg, cb, 1.0f);
    }
    if constexpr (HYBRID) {
        if (unsure.any) {
            const uint32_t q = atomicAdd(a.counter, 1u);
            if (q < a.queue_capacity)
                a.queue[q] = (uint32_t)(frame * frame_px + (uint64_t)a.width * row + x);
        }
    }
}

hipError_t launch_render_fast(const RenderArgs& a, int n_frames, bool hybrid, hipStream_t s)
{
    dim3 grid((a.width + PTR_TILE_W - 1) / PTR_TILE_W, (a.scan_height + PTR_TILE_H - 1) / PTR_TILE_H, n_frames);
    if (hybrid) hipLaunchKernelGGL((render_fast_kernel<true, false>), grid, dim3(PTR_BLOCK), 0, s, a);
    else if (a.flags & 0x8000u /* PTR_DEBUG_REASONS */)
        hipLaunchKernelGGL((render_fast_kernel<false, true>), grid, dim3(PTR_BLOCK), 0, s, a);
    else        hipLaunchKernelGGL((render_fast_kernel<false, false>), grid, dim3(PTR_BLOCK), 0, s, a);
    return hipGetLastError();
}

}  // namespace ptr
