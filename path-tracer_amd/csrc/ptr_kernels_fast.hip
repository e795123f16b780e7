// ptr_kernels_fast.hip — the float32 kernel: PPL camera rays per lane, 64-lane waves.
//
// What it computes is RayTracer.render's per-pixel body (RT/src/RayTracer.js:163-193 ->
// trace :18-95 -> Sphere.intersect, RT/src/Sphere.js:15-33), restructured for the machine:
//   * the sphere loops run on a wave-uniform index, so sphere records arrive through scalar
//     loads (constant address space -> s_load_dwordx8/x4) and sit in SGPRs: no vector memory,
//     no LDS, no bank conflicts.  Each record is requested one sphere ahead of its use;
//   * the geometry is camera-relative and the per-sphere terms |l|^2 - r^2 etc. are computed
//     once in float64 on the host, so the primary test is 3 FMAs + 1 FMA;
//   * everything past the hit/miss test of a sphere is skipped by the whole wave when no lane
//     hits it (exec-mask branch); the shadow loop leaves as soon as every lane is decided;
//   * all bookkeeping booleans (who is still undecided, who is "not sure") live as 64-bit lane
//     masks in SGPRs, and each lane traces PPL pixels, so the scalar unit's work (loop control,
//     addresses, mask logic) is shared by PPL * 64 pixels.
//
// Every discontinuous decision (hit/miss, which sphere is nearest, shadowed or not, the
// 8-bit rounding) carries a conservative error bound.  In HYBRID mode a pixel with any
// decision inside its bound is appended to a queue and re-traced afterwards by the float64
// kernel (ptr_kernels_exact.hip); all other pixels already hold the reference's bytes.
#include <hip/hip_runtime.h>
#include "ptr_device.h"
#include "ptr_kernels.h"
#include "ptr_fast_common.h"

namespace ptr {

// per-pixel state of a lane
struct Pixel {
    bool  live;                 // inside the band
    int   x, row;
    F3    d;                    // camera ray direction
    float tnear, t2, etmax;     // nearest / second nearest distance, largest distance error seen
    int   idx;                  // nearest element, -1 = none
};

// state of a hit pixel during shading
struct Shade {
    bool  valid;
    F3    P, N, so;             // hit point, normal, shadow-ray origin (camera-relative)
    float t, eP, eN;
    SphereM32 mw;
    float ar, ag, ab, wsum, eLmax;
    // per light
    F3    L;
    float ratio, eL, kL, cL;
    unsigned long long open_m, maybe_m, blocked_m;
};

template <bool HYBRID, bool DBG, int PPL>
__global__ void __launch_bounds__(PTR_BLOCK) render_fast_kernel(RenderArgs a)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int frame = blockIdx.z;
    const uint64_t frame_px = (uint64_t)a.width * (uint64_t)a.scan_height;
    uint32_t* out8 = a.out8 + frame * a.out_frame_stride;
    const bool quirk = (a.flags & 0x0001u) != 0;      // PTR_QUIRK_ROW_SHIFT

    const unsigned char* block = a.scenes + (uint64_t)(a.first_frame + frame) * a.frame_stride;
    const ConstHeader hdr = (ConstHeader)(uintptr_t)block;
    const ConstF4 rec0 = (ConstF4)(uintptr_t)(block + a.off_p32);          // uniform reads, 4 f4 per sphere
    const ConstM32 cm32 = (ConstM32)(uintptr_t)(block + a.off_m32);
    const ConstI32 clights = (ConstI32)(uintptr_t)(block + a.off_lights);
    const ConstF4 pairs0 = (ConstF4)(uintptr_t)(block + a.off_pairs);      // [light][n + 2] pre-filter records
    const SphereP32* gp32 = reinterpret_cast<const SphereP32*>(block + a.off_p32);   // per-lane gathers
    const SphereM32* gm32 = reinterpret_cast<const SphereM32*>(block + a.off_m32);
    const int n = a.n;

    // ---- this lane's pixels: an 8 x (8*PPL) tile per wave, 32 x (8*PPL) per workgroup ------------
    Pixel px[PPL];
    [[maybe_unused]] unsigned dbg_full_tests = 0, dbg_pre_tests = 0;   // DBG: shadow tests this wave ran
    Unsure<DBG> unsure[PPL];
#pragma unroll
    for (int p = 0; p < PPL; p++) {
        Pixel& q = px[p];
        q.x = blockIdx.x * PTR_TILE_W + wave * 8 + (lane & 7);
        q.row = blockIdx.y * (PTR_TILE_H * PPL) + p * PTR_TILE_H + (lane >> 3);
        q.live = (q.x < a.width) & (q.row < a.scan_height);
        if (quirk & q.live & (q.row == 0)) {          // RayTracer.js:169: dropped row / untouched last row
            out8[(uint64_t)a.width * (a.scan_height - 1) + q.x] = 0u;
            q.live = a.out32f != nullptr;
        }
        // camera ray (RayTracer.js:171-174)
        const float xx = fmaf((float)q.x + 0.5f, a.camf.sx, a.camf.ox);
        const float yy = fmaf((float)global_y(a, q.row) + 0.5f, a.camf.sy, a.camf.oy);
        F3 d;
        if (a.cam.general) {
            d.x = fmaf(xx, a.camf.right[0], fmaf(yy, a.camf.up[0], a.camf.fwd[0]));
            d.y = fmaf(xx, a.camf.right[1], fmaf(yy, a.camf.up[1], a.camf.fwd[1]));
            d.z = fmaf(xx, a.camf.right[2], fmaf(yy, a.camf.up[2], a.camf.fwd[2]));
        } else {
            d = F3{xx, yy, -1.0f};
        }
        const float inv = rsq(dot3(d, d));
        q.d = F3{d.x * inv, d.y * inv, d.z * inv};
        q.tnear = 3.0e38f; q.t2 = 3.0e38f; q.etmax = 0.0f; q.idx = -1;
    }
    {
        bool any_live = false;
#pragma unroll
        for (int p = 0; p < PPL; p++) any_live |= px[p].live;
        if (ballot(any_live) == 0ull) return;          // wave-uniform exit: the whole wave is outside
    }

    // ---- nearest hit (RayTracer.js:26-41) -------------------------------------------------------
    // disc = b^2 - cc is cheap (cc is per sphere) but cancels; it only has to ORDER the hits here.
    auto primary = [&](const P32Q& q, int i) {
#pragma unroll
        for (int p = 0; p < PPL; p++) {
            Pixel& r = px[p];
            const float b = fmaf(q.g0.z, r.d.z, fmaf(q.g0.y, r.d.y, q.g0.x * r.d.x));
            const float disc = fmaf(b, b, -q.g0.w);
            const float m = fminf(b * q.g1.y, disc);   // hit <=> b >= 0 && disc >= 0 <=> m >= 0
            const unsigned long long hit_m = ballot(m >= 0.0f);
            unsure[p].add_mask(ballot(fabsf(m) <= q.g1.x), WHY_PRIMARY);
            unsure[p].add_mask(q.g2.y != 0.0f ? hit_m : 0ull, WHY_CC);
            if (m >= 0.0f) {                           // skipped by the wave when no lane hits
                const float s = rsq(disc);
                const float t = fmaf(q.g2.x, disc * s, b);   // t0, or t1 when t0 < 0 (RayTracer.js:32-34)
                r.etmax = fmaxf(r.etmax, fmaf(q.g2.z, s, q.g2.w));
                r.t2 = __builtin_amdgcn_fmed3f(r.tnear, r.t2, t);   // second smallest so far
                r.idx = t < r.tnear ? i : r.idx;
                r.tnear = fminf(r.tnear, t);
            }
        }
    };
    {
        // Two record buffers in SGPRs.  Record n exists (the host pads one never-hit record and one
        // more), so the request for "the next one" needs no bounds logic.
        ConstF4 rec = rec0;
        P32Q A = load_p32(rec);
        for (int i = 0; i < n; i += 2) {
            arrive(A);
            const P32Q B = load_p32(rec + 4);
            primary(A, i);
            arrive_after(B, px[PPL - 1].tnear);
            A = load_p32(rec + 8);
            if (i + 1 < n) primary(B, i + 1);
            rec += 8;
        }
        arrive(A);
    }

    // miss: the background, bit-exact by construction (RayTracer.js:43-46)
    const float bgr = hdr->bg32[0], bgg = hdr->bg32[1], bgb = hdr->bg32[2];
    const uint32_t bg_packed = hdr->bg_packed;
    float cr[PPL], cg[PPL], cb[PPL];
    uint32_t packed[PPL];
    Shade sh[PPL];
    bool any_valid = false;
#pragma unroll
    for (int p = 0; p < PPL; p++) {
        Pixel& r = px[p];
        cr[p] = bgr; cg[p] = bgg; cb[p] = bgb; packed[p] = bg_packed;
        sh[p].valid = (r.idx >= 0) & r.live;
        // nearest-vs-second ordering, far clip (t < 1e8, RayTracer.js:19,36)
        unsure[p].add_mask(ballot((r.t2 - r.tnear <= 2.0f * r.etmax) & sh[p].valid), WHY_ORDER);
        unsure[p].add_mask(ballot((r.tnear >= FAR_UNSURE) & sh[p].valid), WHY_FAR);
        any_valid |= sh[p].valid;
    }

    // Wave-uniform branch: taken when any lane hit something; lanes that missed run along on
    // sphere 0 with every flag and result masked by `valid` (keeps all control flow below scalar).
    if (ballot(any_valid) != 0ull) {
#pragma unroll
        for (int p = 0; p < PPL; p++) {
            Pixel& r = px[p];
            Shade& s = sh[p];
            const int idx = s.valid ? r.idx : 0;
            // ---- winner, recomputed accurately (per-lane gather of its record) -----------------
            // disc' = r^2 - |l x d|^2 has no cancellation between |l|^2 and b^2, and the nearest
            // root is taken in its cancellation-free form t = cc / (b + thc).
            const float4 g0 = *reinterpret_cast<const float4*>(&gp32[idx].lx);    // lx ly lz cc
            const float2 g1 = *reinterpret_cast<const float2*>(&gp32[idx].r2);    // r2 lnorm
            const float eb_w = gp32[idx].eb;
            const float2 g3 = *reinterpret_cast<const float2*>(&gp32[idx].inv_r); // inv_r hed
            s.mw = gm32[idx];
            const F3 lw{g0.x, g0.y, g0.z};
            const float cc_w = g0.w, r2_w = g1.x, lnorm_w = g1.y, inv_r = g3.x, hed_w = g3.y;
            const F3 d = r.d;
            const float b_w = dot3(lw, d);
            const F3 cr3{fmaf(lw.y, d.z, -lw.z * d.y), fmaf(lw.z, d.x, -lw.x * d.z), fmaf(lw.x, d.y, -lw.y * d.x)};
            const float disc_w = fmaxf(r2_w - dot3(cr3, cr3), 1e-30f);
            const float s_w = rsq(disc_w);
            const float thc_w = disc_w * s_w;
            const float et_w = fmaf(hed_w, s_w, eb_w);     // |error of (b + thc)|
            const float bt = b_w + thc_w;
            const float rbt = rcp(bt);
            const bool outside = cc_w > 0.0f;
            s.t = outside ? cc_w * rbt : bt;
            s.eP = outside ? s.t * fmaf(et_w, rbt, 4.0f * EPS) : fmaf(2.0f * EPS, s.t, et_w);   // |error of P|

            // ---- hit frame (RayTracer.js:48-58), camera-relative -------------------------------
            s.P = F3{d.x * s.t, d.y * s.t, d.z * s.t};
            F3 N{s.P.x - lw.x, s.P.y - lw.y, s.P.z - lw.z};    // P - C (Sphere.js:47-51)
            const float inv = rsq(dot3(N, N));
            const float sg = outside ? inv : -inv;     // inside: d.N > 0 -> revert (RayTracer.js:54-58)
            s.N = F3{N.x * sg, N.y * sg, N.z * sg};
            s.eN = fmaf(s.eP + EPS * (lnorm_w + s.t), inv_r, 3.0f * EPS);   // |error of N|
            s.so = F3{fmaf(s.N.x, (float)PTR_BIAS, s.P.x), fmaf(s.N.y, (float)PTR_BIAS, s.P.y),
                      fmaf(s.N.z, (float)PTR_BIAS, s.P.z)};
            s.ar = 0.0f; s.ag = 0.0f; s.ab = 0.0f; s.wsum = 0.0f; s.eLmax = 0.0f;
        }

        // ---- direct light + binary shadow (RayTracer.js:61-91) ---------------------------------
        // one shadow test of sphere record pj for every pixel slot
        auto shadow = [&](const P32H& pj) {
            if constexpr (DBG) dbg_pre_tests++;
            // Pre-filter: the shadow ray leaves (within the 1e-4 bias) along the line through the
            // light's centre, so it can only reach element j if that line does: |L.a| >= thr, with
            // the error of L added on the safe side.  Most (light, element) pairs fail this for every
            // lane of the wave, and the whole test below is skipped.
            unsigned long long near_m = 0ull;
#pragma unroll
            for (int p = 0; p < PPL; p++) {
                const Shade& s = sh[p];
                const float c = fmaf(pj.pa.z, s.L.z, fmaf(pj.pa.y, s.L.y, pj.pa.x * s.L.x));
                near_m |= ballot(fabsf(c) + s.cL >= pj.pa.w) & s.open_m;   // thr carries 5 more eps of slack
            }
            if (near_m == 0ull) return;
            if constexpr (DBG) dbg_full_tests++;
#pragma unroll
            for (int p = 0; p < PPL; p++) {
                Shade& s = sh[p];
                const F3 lj{pj.g0.x - s.so.x, pj.g0.y - s.so.y, pj.g0.z - s.so.z};
                const float bj = dot3(lj, s.L);
                const float ll = dot3(lj, lj);
                const float disc = fmaf(bj, bj, pj.g1.z - ll);
                // error bounds: |lj| <= A, 4 d(lj) <= dlj4, d(bj) <= ebj, d(disc) <= band
                const float A = pj.g1.w + s.t;                       // lnorm (padded on the host) + t
                const float dlj4 = fmaf(4.0f * EPS, A, 4.0f * s.eP);
                const float band = fmaf(ll, s.kL, A * dlj4);
                const float ebj = fmaf(A, s.cL, dlj4);
                // certain any-hit (RayTracer.js:77-81) / possible hit
                const unsigned long long certain = ballot(bj > ebj) & ballot(disc > band);
                const unsigned long long possible = ballot(bj >= -ebj) & ballot(disc >= -band);
                s.maybe_m |= possible & ~certain;
                if constexpr (!HYBRID)                 // the float32 verdict only shows in f32-only output
                    s.blocked_m |= ballot(bj >= 0.0f) & ballot(disc >= 0.0f);
                s.open_m &= ~certain;
            }
        };
        auto all_closed = [&]() {
            unsigned long long m = 0ull;
#pragma unroll
            for (int p = 0; p < PPL; p++) m |= sh[p].open_m;
            return m == 0ull;
        };
        // occluders j0 .. j1-1 (the light itself is left out by the caller, RayTracer.js:76)
        auto shadow_range = [&](ConstF4 pairs, int j0, int j1) {
            if (j0 >= j1) return;
            ConstF4 rec = rec0 + 4 * j0;
            ConstF4 pr = pairs + j0;
            P32H A = load_p32h(rec, pr);
            for (int j = j0; j < j1; j += 2) {
                if (all_closed()) break;
                arrive(A);
                const P32H B = load_p32h(rec + 4, pr + 1);
                shadow(A);
                arrive_after(B, sh[PPL - 1].open_m);
                A = load_p32h(rec + 8, pr + 2);
                if (j + 1 < j1) shadow(B);
                rec += 8;
                pr += 2;
            }
            arrive(A);
        };

        const int nl = hdr->n_lights;
        for (int k = 0; k < nl; k++) {
            const int i = clights[k];
            // lights past the table's capacity share a record that never culls (thr < 0)
            const int prow = k < PTR_MAX_PAIR_LIGHTS ? k : (n < PTR_MAX_PAIR_LIGHTS ? n : PTR_MAX_PAIR_LIGHTS);
            const ConstF4 pairs = pairs0 + (size_t)prow * (size_t)(n + 2);
            const P32H pl = load_p32h(rec0 + 4 * i, pairs);
            const float ler = cm32[i].er, leg = cm32[i].eg, leb = cm32[i].eb, lemax = cm32[i].emax;
#pragma unroll
            for (int p = 0; p < PPL; p++) {
                Shade& s = sh[p];
                F3 L{pl.g0.x - s.P.x, pl.g0.y - s.P.y, pl.g0.z - s.P.z};
                const float invl = rsq(dot3(L, L));
                s.L = F3{L.x * invl, L.y * invl, L.z * invl};
                s.eL = fmaf(fmaf(EPS, pl.g1.w + s.t, s.eP), invl, 3.0f * EPS);   // |error of L|
                s.ratio = dot3(s.N, s.L);
                s.kL = fmaf(2.0f, s.eL, 10.0f * EPS);
                s.cL = s.eL + 3.0f * EPS;
                // lane masks: lanes still needing a verdict / inside a bound / blocked in float32
                s.open_m = ballot((s.ratio > 0.0f) & s.valid);
                s.maybe_m = 0ull; s.blocked_m = 0ull;
            }
            shadow_range(pairs, 0, i);
            shadow_range(pairs, i + 1, n);
#pragma unroll
            for (int p = 0; p < PPL; p++) {
                Shade& s = sh[p];
                // A "maybe" only matters if no certain hit blocks the light anyway.
                unsure[p].add_mask(s.open_m & s.maybe_m, WHY_SHADOW);
                if (lane_of(s.open_m & ~s.blocked_m)) {
                    s.ar = fmaf(s.mw.sr * ler, s.ratio, s.ar);
                    s.ag = fmaf(s.mw.sg * leg, s.ratio, s.ag);
                    s.ab = fmaf(s.mw.sb * leb, s.ratio, s.ab);
                    s.wsum += lemax;
                    s.eLmax = fmaxf(s.eLmax, s.eL);
                }
            }
        }

#pragma unroll
        for (int p = 0; p < PPL; p++) {
            Shade& s = sh[p];
            const float hr = s.ar + s.mw.er, hg = s.ag + s.mw.eg, hb = s.ab + s.mw.eb;   // RayTracer.js:93
            // ---- clamp, round, pack (RayTracer.js:179-192) with a rounding bound ----------------
            // |error of N.L| <= eN + eL (+ rounding of the dot and of the sums, 8 eps)
            const float band = 255.0f * fmaf(s.mw.scmax * s.wsum, s.eN + s.eLmax + 8.0f * EPS, 4.0f * EPS * s.mw.emax) + 2e-5f;
            const float vr = fminf(hr, 1.0f) * 255.0f, vg = fminf(hg, 1.0f) * 255.0f, vb = fminf(hb, 1.0f) * 255.0f;
            const float fr = floorf(vr), fg = floorf(vg), fb = floorf(vb);
            const float qr = vr - fr, qg = vg - fg, qb = vb - fb;
            const float nearest = fminf(fminf(fabsf(qr - 0.5f), fabsf(qg - 0.5f)), fabsf(qb - 0.5f));
            unsure[p].add_mask(ballot((nearest <= band) & s.valid), WHY_ROUND);
            const uint32_t r8 = (uint32_t)(qr >= 0.5f ? fr + 1.0f : fr);
            const uint32_t g8 = (uint32_t)(qg >= 0.5f ? fg + 1.0f : fg);
            const uint32_t b8 = (uint32_t)(qb >= 0.5f ? fb + 1.0f : fb);
            if (s.valid) {
                packed[p] = (255u << 24) | (b8 << 16) | (g8 << 8) | r8;
                cr[p] = hr; cg[p] = hg; cb[p] = hb;
            }
        }
    }

    // ---- stores ---------------------------------------------------------------------------------
    const uint32_t bid = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
#pragma unroll
    for (int p = 0; p < PPL; p++) {
        const Pixel& r = px[p];
        if (!r.live) continue;
        if constexpr (DBG) packed[p] = unsure[p].why() | (dbg_full_tests << 8) | (dbg_pre_tests << 16);
        long long idx_out = (long long)a.width * r.row + r.x;
        if (quirk) idx_out -= a.width;
        if (idx_out >= 0) out8[idx_out] = packed[p];
        if (a.out32f) {
            float4* o = reinterpret_cast<float4*>(a.out32f) + (frame * a.out_frame_stride + (uint64_t)a.width * r.row + r.x);
            *o = make_float4(cr[p], cg[p], cb[p], 1.0f);
        }
        if constexpr (HYBRID) {
            if (unsure[p].any()) {
                // one counter word would saturate (~88 returning atomics/us chip-wide): the queue is
                // cut into n_queues sub-queues, each with its own counter line, chosen by workgroup
                const uint32_t sub = bid & (a.n_queues - 1u);       // n_queues is a power of two
                const uint32_t q = atomicAdd(a.counter + sub * PTR_COUNTER_STRIDE, 1u);
                if (q < a.queue_capacity)
                    a.queue[(uint64_t)sub * a.queue_capacity + q] = (uint32_t)(frame * frame_px + (uint64_t)a.width * r.row + r.x);
            }
        }
    }
}

template <bool HYBRID, bool DBG, int PPL>
static hipError_t launch_variant(const RenderArgs& a, int n_frames, hipStream_t s)
{
    const int th = PTR_TILE_H * PPL;
    dim3 grid((a.width + PTR_TILE_W - 1) / PTR_TILE_W, (a.scan_height + th - 1) / th, n_frames);
    hipLaunchKernelGGL((render_fast_kernel<HYBRID, DBG, PPL>), grid, dim3(PTR_BLOCK), 0, s, a);
    return hipGetLastError();
}

int fast_kernel_rows_per_block() { return PTR_TILE_H * PTR_FAST_PPL; }

hipError_t launch_render_fast(const RenderArgs& a, int n_frames, bool hybrid, hipStream_t s)
{
    if (hybrid) return launch_variant<true, false, PTR_FAST_PPL>(a, n_frames, s);
    if (a.flags & 0x8000u /* PTR_DEBUG_REASONS */) return launch_variant<false, true, PTR_FAST_PPL>(a, n_frames, s);
    return launch_variant<false, false, PTR_FAST_PPL>(a, n_frames, s);
}

}  // namespace ptr
