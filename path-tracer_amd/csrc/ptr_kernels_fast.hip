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

// Scene data is read-only for the whole launch: typing it as the constant address space makes
// every wave-uniform read a scalar load (s_load_dwordx4/x8 into SGPRs, batched by the unroller)
// whatever stores the kernel has issued before.
#define PTR_CONST __attribute__((address_space(4)))
typedef const SphereP32 PTR_CONST* ConstP32;
typedef const SphereM32 PTR_CONST* ConstM32;
typedef const FrameHeader PTR_CONST* ConstHeader;
typedef const int32_t PTR_CONST* ConstI32;
typedef float f4 __attribute__((ext_vector_type(4)));
typedef const f4 PTR_CONST* ConstF4;

// the 12 leading floats of a SphereP32 record, fetched with scalar loads one sphere ahead of use
struct P32Q { f4 g0, g1, g2; };
__device__ __forceinline__ P32Q load_p32(ConstP32 base, int i)
{
    const ConstF4 q = (ConstF4)(base + i);
    return P32Q{q[0], q[1], q[2]};
}

__device__ __forceinline__ float dot3(F3 a, F3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
__device__ __forceinline__ float rsq(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ float rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// why a pixel was queued (debug build of the kernel only: PTR_DEBUG_REASONS)
enum : unsigned { WHY_PRIMARY = 1, WHY_CC = 2, WHY_ORDER = 4, WHY_FAR = 8, WHY_SHADOW = 16, WHY_ROUND = 32 };

template <bool DBG> struct Unsure;
template <> struct Unsure<false> {
    // kept as a wave-wide lane mask in SGPRs: accumulating costs scalar ORs, no vector ALU
    unsigned long long mask = 0ull;
    __device__ __forceinline__ void add(bool c, unsigned) { mask |= __builtin_amdgcn_ballot_w64(c); }
    __device__ __forceinline__ bool any() const { return __builtin_amdgcn_inverse_ballot_w64(mask); }
    __device__ __forceinline__ unsigned why() const { return 0u; }
};
template <> struct Unsure<true> {
    unsigned bits = 0u;
    __device__ __forceinline__ void add(bool c, unsigned bit) { bits |= c ? bit : 0u; }
    __device__ __forceinline__ bool any() const { return bits != 0u; }
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
    const ConstHeader hdr = (ConstHeader)(uintptr_t)block;
    const ConstP32 cp32 = (ConstP32)(uintptr_t)(block + a.off_p32);        // uniform reads
    const ConstM32 cm32 = (ConstM32)(uintptr_t)(block + a.off_m32);
    const ConstI32 clights = (ConstI32)(uintptr_t)(block + a.off_lights);
    const SphereP32* gp32 = reinterpret_cast<const SphereP32*>(block + a.off_p32);   // per-lane gathers
    const SphereM32* gm32 = reinterpret_cast<const SphereM32*>(block + a.off_m32);
    const int n = a.n;

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
    auto primary = [&](const P32Q& q, int i) {
        const float b = fmaf(q.g0.z, d.z, fmaf(q.g0.y, d.y, q.g0.x * d.x));
        const float disc = fmaf(b, b, -q.g0.w);
        const float m = fminf(b * q.g1.y, disc);       // hit <=> b >= 0 && disc >= 0 <=> m >= 0
        // NB: the lane masks behind `unsure` are wave-uniform values; they may only be updated in
        // wave-uniform control flow (never inside an exec-masked block like the one below).
        unsure.add(fabsf(m) <= q.g1.x, WHY_PRIMARY);
        unsure.add((m >= 0.0f) & (q.g2.y != 0.0f), WHY_CC);
        if (m >= 0.0f) {                               // skipped by the wave when no lane hits
            const float s = rsq(disc);
            const float t = fmaf(q.g2.x, disc * s, b);    // t0, or t1 when t0 < 0 (RayTracer.js:32-34)
            etmax = fmaxf(etmax, fmaf(q.g2.z, s, q.g2.w));
            t2 = __builtin_amdgcn_fmed3f(tnear, t2, t);   // second smallest so far
            idx = t < tnear ? i : idx;
            tnear = fminf(tnear, t);
        }
    };
    // two record buffers in SGPRs, each loaded one sphere ahead of its use (hides the scalar-load
    // latency); n == 0 reads the zeroed padding of the block and loops zero times
    {
        P32Q A = load_p32(cp32, 0);
        for (int i = 0; i < n; i += 2) {
            const P32Q B = load_p32(cp32, i + 1 < n ? i + 1 : i);
            primary(A, i);
            if (i + 1 >= n) break;
            A = load_p32(cp32, i + 2 < n ? i + 2 : i + 1);
            primary(B, i + 1);
        }
    }
    // nearest-vs-second ordering, far clip (t < 1e8, RayTracer.js:19,36)
    unsure.add((t2 - tnear <= 2.0f * etmax) & (idx >= 0), WHY_ORDER);
    unsure.add((tnear >= FAR_UNSURE) & (idx >= 0), WHY_FAR);

    // miss: the background, bit-exact by construction (RayTracer.js:43-46)
    float cr = hdr->bg32[0], cg = hdr->bg32[1], cb = hdr->bg32[2];     // unclamped colour
    uint32_t packed = hdr->bg_packed;
    const bool valid = idx >= 0;
    // Wave-uniform branch: taken when any lane hit something; lanes that missed run along on
    // sphere 0 with every flag and result masked by `valid` (keeps all control flow below scalar).
    if (__builtin_amdgcn_ballot_w64(valid) != 0ull) {
        idx = valid ? idx : 0;
        // ---- winner, recomputed accurately (per-lane gather of its record) ---------------------
        // disc' = r^2 - |l x d|^2 has no cancellation between |l|^2 and b^2, and the nearest root
        // is taken in its cancellation-free form t = cc / (b + thc).
        const float4 g0 = *reinterpret_cast<const float4*>(&gp32[idx].lx);    // lx ly lz cc
        const float2 g1 = *reinterpret_cast<const float2*>(&gp32[idx].r2);    // r2 lnorm
        const float eb_w = gp32[idx].eb;
        const float2 g3 = *reinterpret_cast<const float2*>(&gp32[idx].inv_r); // inv_r hed
        const SphereM32 mw = gm32[idx];
        const F3 lw{g0.x, g0.y, g0.z};
        const float cc_w = g0.w, r2_w = g1.x, lnorm_w = g1.y, inv_r = g3.x, hed_w = g3.y;
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
        const float eN = fmaf(eP + EPS * (lnorm_w + t), inv_r, 3.0f * EPS);   // |error of N|
        const F3 so{fmaf(N.x, (float)PTR_BIAS, P.x), fmaf(N.y, (float)PTR_BIAS, P.y), fmaf(N.z, (float)PTR_BIAS, P.z)};

        // ---- direct light + binary shadow (RayTracer.js:61-91) ---------------------------------
        float ar = 0.0f, ag = 0.0f, ab = 0.0f;
        float wsum = 0.0f, eLmax = 0.0f;
        const int nl = hdr->n_lights;
        for (int k = 0; k < nl; k++) {
            const int i = clights[k];
            const P32Q pl = load_p32(cp32, i);
            const ConstM32 ml = cm32 + i;
            F3 L{pl.g0.x - P.x, pl.g0.y - P.y, pl.g0.z - P.z};
            const float invl = rsq(dot3(L, L));
            L.x *= invl; L.y *= invl; L.z *= invl;
            const float eL = fmaf(fmaf(EPS, pl.g1.w + t, eP), invl, 3.0f * EPS);   // |error of L|
            const float ratio = dot3(N, L);
            const float kL = fmaf(2.0f, eL, 10.0f * EPS);
            // lane masks (SGPR pairs): lanes still needing a verdict / inside a band / blocked in float32
            unsigned long long open_m = __builtin_amdgcn_ballot_w64((ratio > 0.0f) & valid);
            unsigned long long maybe_m = 0ull, blocked_m = 0ull;
            auto shadow = [&](const P32Q& pj) {
                const F3 lj{pj.g0.x - so.x, pj.g0.y - so.y, pj.g0.z - so.z};
                const float bj = dot3(lj, L);
                const float ll = dot3(lj, lj);
                const float disc = fmaf(bj, bj, pj.g1.z - ll);
                // error bounds: |lj| <= A, d(lj) <= dlj, d(bj) <= ebj, d(disc) <= band
                const float A = pj.g1.w + t + 2.0e-4f;
                const float dlj = fmaf(EPS, A, eP);
                const float band = fmaf(ll, kL, 4.0f * A * dlj);
                const float ebj = fmaf(A, eL + 3.0f * EPS, dlj);
                // certain any-hit (RayTracer.js:77-81) / possible hit
                const unsigned long long certain = __builtin_amdgcn_ballot_w64((bj > ebj) & (disc > band));
                const unsigned long long possible = __builtin_amdgcn_ballot_w64((bj >= -ebj) & (disc >= -band));
                maybe_m |= open_m & possible & ~certain;
                if constexpr (!HYBRID)                 // the float32 verdict only shows in f32-only output
                    blocked_m |= open_m & __builtin_amdgcn_ballot_w64((bj >= 0.0f) & (disc >= 0.0f));
                open_m &= ~certain;
            };
            {
                P32Q A = load_p32(cp32, 0);
                for (int j = 0; j < n; j += 2) {
                    const P32Q B = load_p32(cp32, j + 1 < n ? j + 1 : j);
                    if (open_m == 0ull) break;
                    if (j != i) shadow(A);             // the light itself never blocks (RayTracer.js:76)
                    if (j + 1 >= n) break;
                    A = load_p32(cp32, j + 2 < n ? j + 2 : j + 1);
                    if (j + 1 != i) shadow(B);
                }
            }
            // A "maybe" only matters if no certain hit blocks the light anyway.
            unsure.add(__builtin_amdgcn_inverse_ballot_w64(open_m & maybe_m), WHY_SHADOW);
            const bool open = __builtin_amdgcn_inverse_ballot_w64(open_m & ~blocked_m);
            if (open) {
                ar = fmaf(mw.sr * ml->er, ratio, ar);
                ag = fmaf(mw.sg * ml->eg, ratio, ag);
                ab = fmaf(mw.sb * ml->eb, ratio, ab);
                wsum += ml->emax;
                eLmax = fmaxf(eLmax, eL);
            }
        }
        const float hr = ar + mw.er, hg = ag + mw.eg, hb = ab + mw.eb;   // RayTracer.js:93

        // ---- clamp, round, pack (RayTracer.js:179-192) with a rounding band ------------------
        // |error of N.L| <= eN + eL (+ rounding of the dot and of the sums, 8 eps)
        const float band = 255.0f * fmaf(mw.scmax * wsum, eN + eLmax + 8.0f * EPS, 4.0f * EPS * mw.emax) + 2e-5f;
        const float vr = fminf(hr, 1.0f) * 255.0f, vg = fminf(hg, 1.0f) * 255.0f, vb = fminf(hb, 1.0f) * 255.0f;
        const float fr = floorf(vr), fg = floorf(vg), fb = floorf(vb);
        const float qr = vr - fr, qg = vg - fg, qb = vb - fb;
        const bool near_half = (fabsf(qr - 0.5f) <= band) | (fabsf(qg - 0.5f) <= band) | (fabsf(qb - 0.5f) <= band);
        unsure.add(near_half & valid, WHY_ROUND);
        const uint32_t r8 = (uint32_t)(qr >= 0.5f ? fr + 1.0f : fr);
        const uint32_t g8 = (uint32_t)(qg >= 0.5f ? fg + 1.0f : fg);
        const uint32_t b8 = (uint32_t)(qb >= 0.5f ? fb + 1.0f : fb);
        if (valid) {
            packed = (255u << 24) | (b8 << 16) | (g8 << 8) | r8;
            cr = hr; cg = hg; cb = hb;
        }
    }

    if constexpr (DBG) packed = unsure.why();
    if (idx_out >= 0) out8[idx_out] = packed;
    if (a.out32f) {
        float4* p = reinterpret_cast<float4*>(a.out32f) + (frame * a.out_frame_stride + (uint64_t)a.width * row + x);
        *p = make_float4(cr, cg, cb, 1.0f);
    }
    if constexpr (HYBRID) {
        if (unsure.any()) {
            // one counter word would saturate (~88 returning atomics/us chip-wide): the queue is cut
            // into n_queues sub-queues, each with its own counter line, chosen by workgroup
            const uint32_t bid = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
            const uint32_t sub = bid % a.n_queues;
            const uint32_t q = atomicAdd(a.counter + sub * PTR_COUNTER_STRIDE, 1u);
            if (q < a.queue_capacity)
                a.queue[(uint64_t)sub * a.queue_capacity + q] = (uint32_t)(frame * frame_px + (uint64_t)a.width * row + x);
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
