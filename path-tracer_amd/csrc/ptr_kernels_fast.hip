// ptr_kernels_fast.hip — the float32 kernel: PPL camera rays per lane, 64-lane waves.
//
// What it computes is RayTracer.render's per-pixel body (RT/src/RayTracer.js:163-193 ->
// trace :18-95 -> Sphere.intersect, RT/src/Sphere.js:15-33), restructured for the machine:
//   * the sphere loops run on a wave-uniform index, so sphere records arrive through scalar
//     loads (constant address space -> s_load_dwordx8/x4) and sit in SGPRs: no vector memory,
//     no LDS, no bank conflicts.  Each record is requested one sphere ahead of its use;
//   * the geometry is camera-relative and the per-sphere terms |l|^2 - r^2 etc. are computed
//     once in float64 on the host, so the primary test is 3 FMAs + 1 compare;
//   * a wave's pixels are one (8 PPL) x 8 tile of the image: lane i first tests sphere i against the cone of
//     the tile's rays, and the primary loop visits only the spheres some ray of the tile can reach; the
//     shadow loop is culled per wave the same way (cone of the wave's shadow rays against the
//     light -> occluder cones of the host's table);
//   * everything past the hit/miss test of a sphere is skipped by the whole wave when no lane
//     hits it (exec-mask branch); the shadow loop leaves as soon as every lane is decided;
//   * all bookkeeping booleans (who is still undecided, who is "not sure") live as 64-bit lane
//     masks in SGPRs, and each lane traces PPL pixels, so the scalar unit's work (loop control,
//     addresses, mask logic) is shared by PPL * 64 pixels.
//
// Every discontinuous decision (hit/miss, which sphere is nearest, shadowed or not, the
// 8-bit rounding) carries a conservative error bound.  In HYBRID mode a pixel with any
// decision inside its bound is appended to a queue and re-traced afterwards by the float64
// kernel (ptr_kernels_exact.hip) - re-shaded only, when its 8-bit rounding is the one thing in
// doubt; all other pixels already hold the reference's bytes.
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
    unsigned long long valid_m; // ... as a lane mask
    int   self;                 // the element the pixel lies on, seen from outside (-1: none / seen from inside)
    F3    P, N;                 // hit point, normal (camera-relative)
    float t, eP, eN;            // eP: |error of P| + eps t (every use adds eps t, or may)
    float ar, ag, ab, werr;          // sum over the unblocked lights of emission * N.L per channel; of (largest emission
                                     //   channel) * (error bound of that light's N.L)
    // per light
    float invl;                 // 1 / |light - P| (the unit vector itself is worked out inside the few occluder tests)
    float ratio, eL;            // N.L, |error of L| (+ 3 eps: see the pre-filter)
    unsigned long long open_m, maybe_m, blocked_m;
};

// the nearest element of the first lane, in the first pixel slot, that still waits in `rem` (wave-uniform; some lane waits)
// (Scalar branches: a select between lane masks would turn them into per-lane values.  The slots are taken by a compile-time
// recursion and every branch ends in an asm comment of its own: written as a loop over the slots, the compiler merges the
// branches' v_readlane into one with a run-time slot number - and the whole px[] array then lives in scratch memory.)
template <int P, int PPL>
__device__ __forceinline__ int first_waiting_from(const unsigned long long (&rem)[PPL], const Pixel (&px)[PPL])
{
    if constexpr (P == PPL - 1) {
        return __builtin_amdgcn_readlane(px[P].idx, (int)__builtin_ctzll(rem[P]));
    } else {
        if (rem[P] != 0ull) {
            int r = __builtin_amdgcn_readlane(px[P].idx, (int)__builtin_ctzll(rem[P]));
            asm volatile("; first_waiting: slot %1" : "+s"(r) : "i"(P));
            return r;
        }
        return first_waiting_from<P + 1, PPL>(rem, px);
    }
}
template <int PPL>
__device__ __forceinline__ int first_waiting(const unsigned long long (&rem)[PPL], const Pixel (&px)[PPL])
{
    return first_waiting_from<0, PPL>(rem, px);
}

// MODE 0: float32 only.  MODE 1: hybrid, bytes equal the reference's (every doubt, the 8-bit rounding included, is re-done
// in float64).  MODE 2: hybrid within the tolerance of one unit per channel (PTR_MODE_TOLERANT): only the DISCONTINUOUS
// doubts (hit/miss, nearest-vs-second, far clip, shadow any-hit) are re-traced; the rounding doubt, the per-light sums of
// error bounds behind it and the shading-only re-do are gone, and one compare per pixel keeps the promise: the sum of the
// error bounds of N and of every L must stay below the frame's threshold (FrameHeader::tol_thr: the float32 colour is then
// within one 8-bit unit of the float64 one before rounding, so the rounded bytes differ by at most 1).
// QUIRK: PTR_QUIRK_ROW_SHIFT (a kernel of its own: the throughput paths never pay for the test).
template <int MODE, bool DBG, int PPL, bool GENERAL, bool QUIRK>
__global__ void __launch_bounds__(PTR_FAST_BLOCK) __attribute__((amdgpu_waves_per_eu(PPL >= 4 ? PTR_FAST_WAVES_WIDE : PTR_FAST_WAVES)))
render_fast_kernel(RenderArgs a)
{
    constexpr bool HYBRID = MODE != 0;
    constexpr bool TOL = MODE == 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int frame = blockIdx.z;
    const uint64_t frame_px = (uint64_t)a.width * (uint64_t)a.scan_height;
    uint32_t* out8 = a.out8 + frame * a.out_frame_stride;
    constexpr bool quirk = QUIRK;                     // PTR_QUIRK_ROW_SHIFT

    // (the host passes `scenes` at the call's first frame and vouches that a call's frame blocks span less than 4 GB: few
    // spheres, at most 65 535 frames)
    const unsigned char* block = a.scenes + (uint32_t)((uint32_t)frame * (uint32_t)a.frame_stride);
    const ConstHeader hdr = (ConstHeader)(uintptr_t)block;
    const ConstF4 rec0 = (ConstF4)(uintptr_t)(block + a.off_p32);          // uniform reads, 4 f4 per sphere
    const ConstF4 pairs0 = (ConstF4)(uintptr_t)(block + a.off_pairs);      // [light][pair_stride(n)] pre-filter records
    const SphereP32* gp32 = reinterpret_cast<const SphereP32*>(block + a.off_p32);   // per-lane gathers
    const int n = a.n;

    // lane i's view of sphere record i for the culling step below, requested first: the latency passes during set-up
    const int li = lane < n ? lane : 0;                // record 0 exists even in an empty scene (zero padding)
    const CullRec* gcull = reinterpret_cast<const CullRec*>(block + a.off_cull);
    const float4 cull_g0 = *reinterpret_cast<const float4*>(&gcull[li].lx);   // lx ly lz s
    const float2 cull_g1 = *reinterpret_cast<const float2*>(&gcull[li].re);   // re c

    // ---- this lane's pixels: an (8*PPL) x 8 tile per wave (the pixel slots side by side: a tile never straddles two 8-row strips), one wave per workgroup ------------
    Pixel px[PPL];
    [[maybe_unused]] unsigned dbg_full_tests = 0, dbg_pre_tests = 0;   // DBG: shadow tests this wave ran
    Unsure<DBG> unsure[PPL];
    [[maybe_unused]] unsigned long long ts[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // DBG: clock at the stage boundaries
    auto stamp = [&](int k) { if constexpr (DBG) ts[k] = __builtin_readcyclecounter(); };
    stamp(0);
    // Image row of this lane's pixels (the same for every slot).  With cyclic strips that is a division per lane - unless the
    // tile's 8 rows are consecutive image rows (bands, strips of a multiple of 8 rows): then it is the tile's first row, a
    // wave-uniform value worked out on the scalar side, plus the lane's row in the tile.
    const int tile_row0 = blockIdx.y * PTR_TILE_H;
    int gy0 = 0, gy;
    if (a.rows_mode == 0) {                        // a band: the common case first, two scalar instructions
        gy0 = a.start_y + tile_row0;
        gy = gy0 + (lane >> 3);
    } else if (a.rows_mode == 1) {                 // strips of a multiple of 8 rows
        asm volatile("");                          // (keeps this a branch: the compiler would compute both forms and select)
        gy0 = global_y(a, tile_row0);
        gy = gy0 + (lane >> 3);
    } else {
        asm volatile("");
        gy = global_y(a, tile_row0 + (lane >> 3));
    }
    // camera ray of a pixel slot (RayTracer.js:171-174)
    auto camera_ray = [&](Pixel& q) {
        const float xx = fmaf((float)q.x + 0.5f, a.camf.sx, a.camf.ox);
        const float yy = fmaf((float)gy + 0.5f, a.camf.sy, a.camf.oy);
        F3 d;
        if constexpr (GENERAL) {          // a kernel of its own: as a run-time test the compiler computes both forms and selects
            d.x = fmaf(xx, a.camf.right[0], fmaf(yy, a.camf.up[0], a.camf.fwd[0]));
            d.y = fmaf(xx, a.camf.right[1], fmaf(yy, a.camf.up[1], a.camf.fwd[1]));
            d.z = fmaf(xx, a.camf.right[2], fmaf(yy, a.camf.up[2], a.camf.fwd[2]));
        } else {
            d = F3{xx, yy, -1.0f};
        }
        const float inv = rsq(dot3(d, d));
        q.d = F3{d.x * inv, d.y * inv, d.z * inv};
    };
#pragma unroll
    for (int p = 0; p < PPL; p++) {
        Pixel& q = px[p];
        q.x = blockIdx.x * (PTR_FAST_BLOCK / 64 * 8 * PPL) + wave * (8 * PPL) + p * 8 + (lane & 7);
        q.row = blockIdx.y * PTR_TILE_H + (lane >> 3);
        q.live = (q.x < a.width) & (q.row < a.scan_height);
        q.tnear = 3.0e38f; q.t2 = 3.0e38f; q.etmax = 0.0f; q.idx = -1;
    }
    {
        bool any_live = false;
#pragma unroll
        for (int p = 0; p < PPL; p++) any_live |= px[p].live;
        if (ballot(any_live) == 0ull) return;          // wave-uniform exit: the whole wave is outside
    }
    if constexpr (QUIRK) {
        if (blockIdx.y == 0) {
            // PTR_QUIRK_ROW_SHIFT (RayTracer.js:169): the last row of the band is never written by a pixel; the waves of the
            // first tile row clear it.  (A wave-uniform branch that the other waves jump over; the empty asm keeps it one.)
            asm volatile("");
#pragma unroll
            for (int p = 0; p < PPL; p++)
                if (px[p].live & (px[p].row == 0)) out8[(uint64_t)a.width * (a.scan_height - 1) + px[p].x] = 0u;
        }
    }

    stamp(1);
    // ---- which spheres can a ray of this wave reach at all? ---------------------------------------
    // The wave's pixels form a (8 PPL) x 8 block of the image, so its rays lie in a narrow cone around the
    // block's centre ray dc (half-angle bound from the host, cull_sin / cull_cos).  A ray of the cone can
    // only hit sphere i if the angle between dc and l_i is at most that half-angle plus the sphere's own
    // angular radius asin(r / |l|):  l.dc >= cos(T) sqrt(cc) - sin(T) r.  Lane i evaluates this for sphere
    // i from its CullRec (radius enlarged by the error band of the float32 hit test, so a culled sphere is a
    // CERTAIN miss for every pixel, not an unsure one; the square roots come from the host); the loop below
    // then visits the set bits only.
    unsigned cand = (1u << n) - 1u;                    // n <= PTR_FAST_MAX_SPHERES
    if (a.cull_on) {
        const int x0 = blockIdx.x * (PTR_FAST_BLOCK / 64 * 8 * PPL) + wave * (8 * PPL);
        const float xc = fmaf((float)(x0 + 4 * PPL), a.camf.sx, a.camf.ox);
        const float yc = fmaf((float)(gy0 + PTR_TILE_H / 2), a.camf.sy, a.camf.oy);     // (cull_on implies consecutive rows)
        F3 dc;
        if constexpr (GENERAL) {
            dc.x = fmaf(xc, a.camf.right[0], fmaf(yc, a.camf.up[0], a.camf.fwd[0]));
            dc.y = fmaf(xc, a.camf.right[1], fmaf(yc, a.camf.up[1], a.camf.fwd[1]));
            dc.z = fmaf(xc, a.camf.right[2], fmaf(yc, a.camf.up[2], a.camf.fwd[2]));
        } else {
            dc = F3{xc, yc, -1.0f};
        }
        const float inv = rsq(dot3(dc, dc));
        const float bc = dot3(F3{cull_g0.x, cull_g0.y, cull_g0.z}, dc) * inv;
        const float K = fmaf(a.cull_cos, cull_g0.w, -fmaf(a.cull_sin, cull_g1.x, cull_g1.y));
        cand = (unsigned)ballot(bc >= K) & ((1u << n) - 1u);
    }

    stamp(2);
    // ---- nearest hit (RayTracer.js:26-41) -------------------------------------------------------
    // One compare decides the common case: a ray with b = l.d below sclo (= sqrt(cc) - error of b) misses.
    // disc = b^2 - cc is cheap (cc is per sphere) but cancels; it only has to ORDER the hits here.
    auto primary = [&](const P32Q& q, int i) {
        float bs[PPL];
        unsigned long long poss[PPL], any_m = 0ull;
#pragma unroll
        for (int p = 0; p < PPL; p++) {
            const Pixel& r = px[p];
            bs[p] = fmaf(q.g0.z, r.d.z, fmaf(q.g0.y, r.d.y, q.g0.x * r.d.x));
            poss[p] = ballot(bs[p] >= q.g2.z);
            any_m |= poss[p];
        }
        if (any_m == 0ull) return;                     // wave-uniform: ONE compare -> mask -> branch trip per sphere
        const float ccf = q.g2.y;                      // 0.0f or 1.0f: compared as bits, on the scalar side
        const bool cc_flag = __float_as_uint(ccf) != 0u;
#pragma unroll
        for (int p = 0; p < PPL; p++) {
            Pixel& r = px[p];
            const float b = bs[p];
            const unsigned long long poss_m = poss[p];
            const float disc = fmaf(b, b, -q.g0.w);
            const float m = fminf(b * q.g1.y, disc);   // hit <=> b >= 0 && disc >= 0 <=> m >= 0
            const unsigned long long hit_m = ballot(m >= 0.0f);
            unsure[p].add_mask(ballot(fabsf(m) <= q.g1.x) & poss_m, WHY_PRIMARY);
            if (cc_flag) unsure[p].add_mask(hit_m, WHY_CC);    // wave-uniform, rare: a branch instead of four selects
            // Branch-free for the lanes that miss (they carry t = "never" and add no error): switching lanes off and on
            // costs more than the two selects.  Only the copy of a new leader's record is predicated.
            const bool hit = m >= 0.0f;
            const float s = rsq(disc);                 // (NaN where disc < 0: selected away below)
            const float t = hit ? fmaf(q.g2.x, disc * s, b) : 3.0e38f;      // t0, or t1 when t0 < 0 (RayTracer.js:32-34)
            r.etmax = fmax_raw(r.etmax, hit ? fmaf(0.5f * q.g1.x, s, q.g2.w) : 0.0f);     // |dt| <= md / (2 thc) + eb
            r.t2 = __builtin_amdgcn_fmed3f(r.tnear, r.t2, t);   // second smallest so far
            r.idx = t < r.tnear ? i : r.idx;           // strict: the first of equals stays (RayTracer.js:36)
            r.tnear = fmin_raw(r.tnear, t);
        }
    };
    {
        // Two record buffers in SGPRs: the next candidate's record is requested before the current one is
        // used.  Record n exists (zero padding), so "no next candidate" needs no special case.
        if (cand != 0u) {
            unsigned todo = cand;
            const unsigned stop = 1u << n;             // record n: the zero padding
            int i = __builtin_ctz(todo);
            P32Q A = load_p32(rec0 + 4u * (unsigned)i), B;
            // the camera rays are only worked out here: a wave whose tile no sphere can reach (sky) never needs them, and
            // the first candidate's record is on its way meanwhile
#pragma unroll
            for (int p = 0; p < PPL; p++) camera_ray(px[p]);
            // (two steps per trip, the buffers swapping roles: no record is copied from one set of SGPRs to the other)
            auto step = [&](const P32Q& cur, P32Q& next) {
                todo &= todo - 1u;
                const int inext = __builtin_ctz(todo | stop);
                arrive(cur);
                next = load_p32(rec0 + 4u * (unsigned)inext);
                primary(cur, i);
                arrive_after(next, px[PPL - 1].tnear);
                i = inext;
            };
            for (;;) {
                step(A, B);
                if (!todo) break;
                step(B, A);
                if (!todo) break;
            }
        }
    }
    stamp(3);
    // miss: the background, bit-exact by construction (RayTracer.js:43-46)
    float cr[PPL], cg[PPL], cb[PPL];           // shaded colour (lanes of shaded_m; the others take the background at the store)
    unsigned long long shaded_m[PPL];
    uint32_t packed[PPL];
    Shade sh[PPL];
    bool shade_wave = false;                   // wave-uniform: some lane hit something
    if (cand != 0u) {                          // (a wave without candidates goes straight to the background)
        bool any_valid = false;
#pragma unroll
        for (int p = 0; p < PPL; p++) {
            Pixel& r = px[p];
            // (lanes outside the image trace a ray like every other and are only kept from storing: `live` is not needed here)
            sh[p].valid = r.idx >= 0;
            sh[p].valid_m = ballot(sh[p].valid);
            // nearest-vs-second ordering, far clip (t < 1e8, RayTracer.js:19,36).  (A compare writes its lane mask directly;
            // the ballot of `compare & bool` costs two more vector instructions.)
            unsure[p].add_mask(ballot(r.t2 - r.tnear <= 2.0f * r.etmax) & sh[p].valid_m, WHY_ORDER);
            unsure[p].add_mask(ballot(r.tnear >= FAR_UNSURE) & sh[p].valid_m, WHY_FAR);
            any_valid |= sh[p].valid;
        }
        shade_wave = ballot(any_valid) != 0ull;
    }

    // Wave-uniform branch: taken when any lane hit something; lanes that missed run along on
    // sphere 0 with every flag and result masked by `valid` (keeps all control flow below scalar).
    if (shade_wave) {
        // A shadow ray never hits the sphere it starts on when that sphere was hit from outside: its origin is
        // P + 1e-4 N, at distance r + 1e-4 from the centre, so tca = l.L = -(r + 1e-4) N.L <= 0 whenever the light is
        // in front (N.L > 0), and should rounding lift tca to 0, d2 = |l|^2 - tca^2 > r^2 by 2e-4 r (Sphere.js:19-25).
        // The host vouches (self_skip) that float64 rounding at the scene's coordinates is far below that margin.
        //
        // Everything the light loop needs first is requested here, in one batch, and lands while the winner is refined:
        // the header words, light 0's record (LightRec: its address does not depend on loaded data), lane j's pre-filter
        // record for (light 0, element j) and element j's centre.
        const ConstF4 lrec0 = (ConstF4)(uintptr_t)(block + a.off_lrec);
        LightQ lq = load_light(lrec0);
        const volatile uint32_t PTR_CONST* hw = (const volatile uint32_t PTR_CONST*)(uintptr_t)block;
        const int nl = (int)hw[offsetof(FrameHeader, n_lights) / 4];
        const bool self_ok = hw[offsetof(FrameHeader, self_skip) / 4] != 0u;
        [[maybe_unused]] float tol_thr = 0.0f;
        if constexpr (TOL) tol_thr = __uint_as_float(hw[offsetof(FrameHeader, tol_thr) / 4]);
        const float4* lane_pairs = reinterpret_cast<const float4*>(block + a.off_pairs) + li;
        float4 pq = load_f4(lane_pairs);
        const F3 cj = load_f3(&gp32[li].lx);
        // ---- the winners, recomputed accurately: one trip per DISTINCT nearest element of the wave -----------------
        // Nearly every wave that shades sees one element in all its pixels (95 % on the example scene): its record is
        // fetched by ONE scalar load and used straight from SGPRs by all lanes; a wave on a silhouette takes a second
        // trip for the lanes of the other element.  No per-lane copy of the record is kept anywhere.
        // disc' = r^2 - |l x d|^2 has no cancellation between |l|^2 and b^2, and the nearest
        // root is taken in its cancellation-free form t = cc / (b + thc).
#pragma unroll
        for (int p = 0; p < PPL; p++) {               // (written under a lane mask below: any value will do until then)
            Shade& s = sh[p];
            any_value(s.P.x); any_value(s.P.y); any_value(s.P.z); any_value(s.N.x); any_value(s.N.y); any_value(s.N.z);
            any_value(s.t); any_value(s.eP); any_value(s.eN);
            s.self = -1;
            s.ar = 0.0f; s.ag = 0.0f; s.ab = 0.0f; s.werr = 0.0f;
        }
        {
            unsigned long long rem[PPL];
#pragma unroll
            for (int p = 0; p < PPL; p++) rem[p] = sh[p].valid_m;
            for (;;) {
                // the element of the first lane that still waits (wave-uniform)
                int w = first_waiting(rem, px);
                w &= 15;                                   // (tells the compiler that the index is small)
                const P32W W = load_p32w(rec0 + 4u * (unsigned)w);
                unsigned long long m[PPL];
#pragma unroll
                for (int p = 0; p < PPL; p++) { m[p] = ballot(px[p].idx == w); rem[p] &= ~m[p]; }
                arrive(W);
                const float cc_w = W.g0.w, r2_w = W.g1.z, lnorm_w = W.g1.w, eb_w = W.g2.w, inv_r = W.g3.x;
                const float el_w = EPS * lnorm_w;          // (wave-uniform: once per trip)
                const bool form_b = __float_as_int(W.g3.y) < 0;        // wave-uniform (the host's flag: see SphereP32::hed)
                const float hed_w = __int_as_float(__float_as_int(W.g3.y) & 0x7fffffff);
                const bool outside = cc_w > 0.0f;          // wave-uniform: a scalar branch, not selects
#pragma unroll
                for (int p = 0; p < PPL; p++) {
                    if (m[p] == 0ull) continue;            // wave-uniform
                    if (lane_of(m[p])) {
                        Shade& s = sh[p];
                        const F3 d = px[p].d;
                        const float b_w = fmaf(W.g0.z, d.z, fmaf(W.g0.y, d.y, W.g0.x * d.x));
                        float disc_w, herr_w;                          // the discriminant and half its error bound
                        if (form_b) {
                            // b^2 - cc: |error| <= (2|b| + eb) eb (of b) + 2 eps (b^2 + |cc|) (cc's own rounding, the fma's)
                            disc_w = fmaxf(fmaf(b_w, b_w, -cc_w), 1e-30f);
                            herr_w = 0.5001f * fmaf(fmaf(2.0f, fabsf(b_w), eb_w), eb_w, (2.0f * EPS) * fmaf(b_w, b_w, fabsf(cc_w)));
                        } else {
                            const F3 cr3{fmaf(W.g0.y, d.z, -(W.g0.z * d.y)), fmaf(W.g0.z, d.x, -(W.g0.x * d.z)), fmaf(W.g0.x, d.y, -(W.g0.y * d.x))};
                            disc_w = fmaxf(r2_w - dot3(cr3, cr3), 1e-30f);
                            herr_w = hed_w;
                        }
                        const float s_w = rsq(disc_w);
                        const float thc_w = disc_w * s_w;
                        const float et_w = fmaf(herr_w, s_w, eb_w);    // |error of (b + thc)|
                        const float bt = b_w + thc_w;
                        const float rbt = rcp(bt);
                        float sg;
                        if (outside) {
                            s.self = w;
                            s.t = cc_w * rbt;
                            s.eP = s.t * fmaf(et_w, rbt, 5.0f * EPS);  // |error of P| + eps t
                            sg = inv_r;
                        } else {
                            s.t = bt;
                            s.eP = fmaf(3.0f * EPS, s.t, et_w);
                            sg = -inv_r;                               // inside: d.N > 0 -> revert (RayTracer.js:54-58)
                        }
                        // ---- hit frame (RayTracer.js:48-58), camera-relative -------------------------------
                        s.P = F3{d.x * s.t, d.y * s.t, d.z * s.t};
                        // P - C (Sphere.js:47-51), normalised by 1 / r: the true hit point lies ON the sphere, so this differs
                        // from the true normal by (error of P) / r, which eN below already counts
                        s.N = F3{(s.P.x - W.g0.x) * sg, (s.P.y - W.g0.y) * sg, (s.P.z - W.g0.z) * sg};
                        if constexpr (TOL)
                            s.eN = (s.eP + el_w) * inv_r;                               // (the constant terms are in tol_thr; the lights add their eL to this)
                        else
                            s.eN = fmaf(s.eP + el_w, inv_r, 15.0f * EPS);               // |error of N| = (|error of P| + eps (|l| + t)) / r (1 / r itself:
                                                                                        //   2 eps), + 10 eps for the rounding of N.L and of the sums over the lights
                    }
                }
                unsigned long long left = 0ull;
#pragma unroll
                for (int p = 0; p < PPL; p++) left |= rem[p];
                if (left == 0ull) break;
            }
        }
        // (after the arithmetic above: this is the first use of a word requested at the top)
#pragma unroll
        for (int p = 0; p < PPL; p++) sh[p].self = self_ok ? sh[p].self : -1;      // (lanes without a hit hold -1 already)
        // The wave's shadow-ray origins lie in a ball: centre = the first shaded lane's hit point, radius = the largest
        // distance of a shaded lane's P from it, plus the error of P and the 1e-4 bias (used by cone_prefilter).
        float pcx, pcy, pcz, prad;
        {
            int ps = PPL - 1;                                  // the first slot with a shaded lane (scalar branches)
#pragma unroll
            for (int p = PPL - 2; p >= 0; p--) ps = sh[p].valid_m != 0ull ? p : ps;
            pcx = pcy = pcz = 0.0f;
#pragma unroll
            for (int p = 0; p < PPL; p++)
                if (ps == p) {
                    const int src = (int)__builtin_ctzll(sh[p].valid_m);
                    pcx = readlane_f(sh[p].P.x, src); pcy = readlane_f(sh[p].P.y, src); pcz = readlane_f(sh[p].P.z, src);
                }
            float dev = 0.0f;
#pragma unroll
            for (int p = 0; p < PPL; p++) {
                const Shade& s = sh[p];
                const float dx = s.P.x - pcx, dy = s.P.y - pcy, dz = s.P.z - pcz;
                const float e = __builtin_amdgcn_sqrtf(fmaf(dz, dz, fmaf(dy, dy, dx * dx))) + s.eP;
                dev = fmax_raw(dev, s.valid ? e : 0.0f);
            }
            prad = fmaf(wave_max(dev), 1.000001f, 1.01e-4f);
        }
        stamp(4);

        // ---- direct light + binary shadow (RayTracer.js:61-91) ---------------------------------
        // one shadow test of sphere record pj for every pixel slot
        // What the float64 re-do of a pixel that only has its ROUNDING in doubt may take over (retrace kind 1,
        // ptr_kernels_exact.hip reshade_exact): light k "certainly blocked" = bit 8 + 2k of the lane's idx register
        // (n <= 12: the index itself fits the low byte); k is the light being worked on.
        int k = 0;
        F3 lpos_k{0.0f, 0.0f, 0.0f};                      // light k's position (wave-uniform)
        auto shadow = [&](const P32H& pj, int j) {
            if constexpr (DBG) dbg_pre_tests++;
            // Pre-filter: the shadow ray leaves (within the 1e-4 bias) along the line through the
            // light's centre, so it can only reach element j if that line does: |L.a| >= thr, with
            // the error of L added on the safe side.  Most (light, element) pairs fail this for every
            // lane of the wave, and the whole test below is skipped.
            unsigned long long near_m = 0ull, pass_m[PPL];
            // the unit vector towards the light, rebuilt per test: few waves get here, and PPL x 3 registers stay free
            auto unit_L = [&](const Shade& s) { return F3{(lpos_k.x - s.P.x) * s.invl, (lpos_k.y - s.P.y) * s.invl, (lpos_k.z - s.P.z) * s.invl}; };
#pragma unroll
            for (int p = 0; p < PPL; p++) {
                const Shade& s = sh[p];
                const F3 L = unit_L(s);
                const float c = fmaf(pj.pa.z, L.z, fmaf(pj.pa.y, L.y, pj.pa.x * L.x));
                pass_m[p] = ballot(fabsf(c) + s.eL >= pj.pa.w) & s.open_m;   // thr carries 8 more eps of slack
                near_m |= pass_m[p];
            }
            if (near_m == 0ull) return;
            // lanes standing on j need no test of j (few records get this far: cheaper here than a wave-level
            // "everybody stands on j" worked out once per wave)
            near_m = 0ull;
#pragma unroll
            for (int p = 0; p < PPL; p++) near_m |= pass_m[p] & ~ballot(sh[p].self == j);
            if (near_m == 0ull) return;
            if constexpr (DBG) dbg_full_tests++;
#pragma unroll
            for (int p = 0; p < PPL; p++) {
                Shade& s = sh[p];
                // shadow-ray origin P + 1e-4 N (RayTracer.js:74), rebuilt here: full tests are rare, registers are not
                const F3 so{fmaf(s.N.x, (float)PTR_BIAS, s.P.x), fmaf(s.N.y, (float)PTR_BIAS, s.P.y),
                            fmaf(s.N.z, (float)PTR_BIAS, s.P.z)};
                const F3 lj{pj.g0.x - so.x, pj.g0.y - so.y, pj.g0.z - so.z};
                const float bj = dot3(lj, unit_L(s));
                const float ll = dot3(lj, lj);
                const float disc = fmaf(bj, bj, pj.g1.z - ll);
                // error bounds: |lj| <= A, 4 d(lj) <= dlj4, d(bj) <= ebj, d(disc) <= band
                const float A = pj.g1.w + s.t;                       // lnorm (padded on the host) + t
                const float dlj4 = fmaf(4.0f * EPS, A, 4.0f * s.eP);
                const float band = fmaf(ll, fmaf(2.0f, s.eL, 10.0f * EPS), A * dlj4);
                const float ebj = fmaf(A, s.eL + 3.0f * EPS, dlj4);
                // certain any-hit (RayTracer.js:77-81) / possible hit
                const unsigned long long certain = ballot(bj > ebj) & ballot(disc > band);
                const unsigned long long possible = ballot(bj >= -ebj) & ballot(disc >= -band);
                s.maybe_m |= possible & ~certain;
                if constexpr (HYBRID && !TOL) px[p].idx |= lane_of(certain & s.open_m) ? (int)(0x100u << (2 * k)) : 0;
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
        // Phase 1 for one light, once per WAVE, from the ball of shadow-ray origins alone: every origin (and every hit
        // point) is within prad of pc, so every shadow ray towards a light at C points within
        //   dmax = rho / (|C - pc| - rho),  rho = prad + rounding of C - pc,
        // of the axis (C - pc) / |C - pc|  (|u/|u| - v/|v|| <= |u - v| / min(|u|, |v|)).  The line through the light
        // along L can reach element j only if |a_j . L| >= thr_j (PairQ), and |a_j . L| <= |a_j . axis| + dmax: lane j tests
        // element j against the cone, and the ballot is a wave-uniform bit per occluder, "some lane may reach it".  No
        // per-light cross-lane maximum: the ball was found once, with the hit points.
        const F3 lcj{cj.x - pcx, cj.y - pcy, cj.z - pcz};              // lane j: element j's centre from the ball's centre
        const float lcn = __builtin_amdgcn_sqrtf(dot3(lcj, lcj)) * 1.000001f + 1e-6f;
        const float pradp = prad * 1.000001f;
        auto cone_prefilter = [&](const float4 pq, const F3 lpos, const float llnorm) -> unsigned {
            const F3 u{lpos.x - pcx, lpos.y - pcy, lpos.z - pcz};
            const float dd = dot3(u, u);
            const float inv = rsq(fmax_raw(dd, 1e-30f));
            const float dist = dd * inv;
            const float rho = fmaf(8.0f * EPS, llnorm + dist, prad);       // + rounding of the light's position and of u
            const float den = dist - rho;
            // (a light inside or next to the ball: no cone worth the name, nothing is culled)
            const float dmax = den > rho ? fmaf(rho * rcp(den), 1.000001f, 1e-6f) : 4.0f;
            const F3 ax{u.x * inv, u.y * inv, u.z * inv};
            const float c = fmaf(pq.z, ax.z, fmaf(pq.y, ax.y, pq.x * ax.x));
            // ... and element j lies BEHIND every shadow ray of the wave when tca = (C_j - origin) . L < 0 for all of
            // them (Sphere.js:18-19: tca < 0 is a miss wherever the origin is):  tca <= lcj.axis + |lcj| dmax + prad
            const float tca_up = fmaf(lcn, dmax, dot3(lcj, ax)) + pradp;
            return (unsigned)(ballot(fabsf(c) + dmax >= pq.w) & ballot(tca_up >= 0.0f)) & ((1u << n) - 1u);  // thr carries 8 more eps of slack
        };
        // Phase 2: the full test for the occluders whose bit is set (next record requested before the current is used)
        auto shadow_bits = [&](ConstF4 pairs, unsigned need) {
            // (no look-ahead here: a wave rarely has a second set bit, and a second record buffer costs 12 SGPRs that
            // the register allocator would have to spill)
            while (need) {
                const int j = __builtin_ctz(need);
                need &= need - 1u;
                const P32H A = load_p32h(rec0 + 4u * (unsigned)j, pairs + j);
                arrive(A);
                shadow(A, j);
                if (all_closed()) break;
            }
        };

        static_assert(PTR_FAST_MAX_SPHERES <= 16 && PTR_FAST_MAX_SPHERES <= PTR_MAX_PAIR_LIGHTS, "light k has pre-filter row k");
        ConstF4 lrp = lrec0, pairs = pairs0;                 // light k's record / pre-filter row: stepped, not indexed
        const unsigned pstride = (unsigned)pair_stride(n);
        const float4* lane_row = lane_pairs;
        for (k = 0; k < nl; k++) {
            arrive(lq);
            const unsigned i = (unsigned)lq.index & 15u;       // (the mask tells the compiler that the index is small)
            const F3 lpos{lq.g.x, lq.g.y, lq.g.z};
            lpos_k = lpos;
            const float llnorm = lq.g.w;
            const float ler = lq.e.x, leg = lq.e.y, leb = lq.e.z, lemax = lq.e.w;
            const float el_l = lq.eps_lnorm;                   // eps |l| of the light (host)
#pragma unroll
            for (int p = 0; p < PPL; p++) {
                Shade& s = sh[p];
                const F3 L{lpos.x - s.P.x, lpos.y - s.P.y, lpos.z - s.P.z};
                s.invl = rsq(dot3(L, L));
                s.eL = fmaf(s.eP + el_l, s.invl, 3.0f * EPS);  // |error of L / |L||: (|error of P| + eps (|l| + t)) / |L|, + its own rounding
                s.ratio = dot3(s.N, L) * s.invl;               // N.L: one multiply instead of three for the unit vector
                // lane masks: lanes still needing a verdict / inside a bound / blocked in float32
                s.open_m = ballot(s.ratio > 0.0f) & s.valid_m;
                s.maybe_m = 0ull; s.blocked_m = 0ull;
            }
            if (!all_closed()) {
                // every element but the light itself (RayTracer.js:76) and the one the whole wave stands on
                unsigned need = cone_prefilter(pq, lpos, llnorm) & ~(1u << i);
                shadow_bits(pairs, need);
            }
#pragma unroll
            for (int p = 0; p < PPL; p++) {
                Shade& s = sh[p];
                // A "maybe" only matters if no certain hit blocks the light anyway.  N.L within its error of 0 is a
                // "maybe" too: the float64 value may be positive where this one is not, and then the light counts.
                unsure[p].add_mask(s.open_m & s.maybe_m, WHY_SHADOW);
                // branch-free: an unlit lane adds zeros (selects cost less than switching lanes off and on again)
                const bool lit = lane_of(s.open_m & ~s.blocked_m);
                const float wr = lit ? s.ratio : 0.0f;
                s.ar = fmaf(ler, wr, s.ar);
                s.ag = fmaf(leg, wr, s.ag);
                s.ab = fmaf(leb, wr, s.ab);
                if constexpr (TOL) {
                    // N.L is off by at most eN + eL, whether the light counts or not (a float32 N.L <= 0 whose float64 value
                    // is positive adds at most that much too): the bounds are summed, the sum is compared once per pixel
                    s.eN += s.eL;
                } else {
                    const float enl = s.eN + s.eL;             // |error of N.L| (the rounding of the dot and of the sums is in eN)
                    if constexpr (HYBRID) unsure[p].add_mask(ballot(fabsf(s.ratio) <= enl) & s.valid_m, WHY_SHADOW);
                    s.werr = fmaf(lit ? lemax : 0.0f, enl, s.werr);
                }
            }
            // the next light's record and pre-filter row (the tables end with a padding record / row)
            lrp += 3; pairs += pstride; lane_row += pstride;
            lq = load_light(lrp);
            pq = load_f4(lane_row);
        }

        if constexpr (TOL) {
            // the one check of the tolerance mode: the summed error bounds of N.L over the lights against the frame's threshold
#pragma unroll
            for (int p = 0; p < PPL; p++) unsure[p].add_mask(ballot(sh[p].eN > tol_thr) & sh[p].valid_m, WHY_ROUND);
        }
        stamp(5);
        // the background (RayTracer.js:43-46, bit-exact by construction) is read only here: four SGPRs fewer to carry
        const uint32_t bg_packed = hdr->bg_packed;
        // surfaceColor * (sum over lights) + emission (RayTracer.js:85-93), clamp, round, pack (RayTracer.js:179-192) with a
        // rounding bound: again one trip per distinct nearest element, its material straight from SGPRs
        float rdoubt[PPL];                         // distance of the nearest rounding boundary minus its error band
#pragma unroll
        for (int p = 0; p < PPL; p++) {
            packed[p] = bg_packed; rdoubt[p] = 1.0f; shaded_m[p] = sh[p].valid_m;
            any_value(cr[p]); any_value(cg[p]); any_value(cb[p]);
            // HYBRID: a pixel in doubt for any reason OTHER than its rounding is re-traced from scratch (bit 9 of its idx
            // register: one lane mask, not two, stays in SGPRs); one that is only here for its rounding gets the shading-only re-do
            if constexpr (HYBRID && !TOL) px[p].idx |= unsure[p].any() ? 0x200 : 0;
        }
        const ConstF4 mat0 = (ConstF4)(uintptr_t)(block + a.off_m32);
        {
            unsigned long long rem[PPL];
#pragma unroll
            for (int p = 0; p < PPL; p++) rem[p] = sh[p].valid_m;
            for (;;) {
                int w = first_waiting(rem, px);
                w &= 15;
                f4 mlo, mhi;                               // sr sg sb scmax | er eg eb emax
                load_f8(mat0 + 2u * (unsigned)w, mlo, mhi);
                unsigned long long m[PPL];
#pragma unroll
                for (int p = 0; p < PPL; p++) { m[p] = ballot((TOL ? px[p].idx : (px[p].idx & 0xFF)) == w); rem[p] &= ~m[p]; }
                arrive(mlo); arrive(mhi);
#pragma unroll
                for (int p = 0; p < PPL; p++) {
                    if (m[p] == 0ull) continue;            // wave-uniform
                    if (lane_of(m[p])) {
                        const Shade& s = sh[p];
                        const float hr = fmaf(mlo.x, s.ar, mhi.x), hg = fmaf(mlo.y, s.ag, mhi.y), hb = fmaf(mlo.z, s.ab, mhi.z);
                        // Math.round(v) = floor(v + 0.5); v + 0.5 is exact in float32 for 0.5 <= v < 256, and below 0.5 it can only
                        // round across 1.0 within the band that is re-traced anyway
                        if constexpr (!TOL) {
                            const float vr = fmin_raw(hr, 1.0f) * 255.0f, vg = fmin_raw(hg, 1.0f) * 255.0f, vb = fmin_raw(hb, 1.0f) * 255.0f;
                            // |error of N.L| <= eN + eL (+ rounding of the dot and of the sums, 8 eps)
                            const float band = 255.0f * fmaf(mlo.w, s.werr, 4.0f * EPS * mhi.w) + 2e-5f;
                            const float qr = __builtin_amdgcn_fractf(vr) - 0.5f, qg = __builtin_amdgcn_fractf(vg) - 0.5f,
                                        qb = __builtin_amdgcn_fractf(vb) - 0.5f;
                            rdoubt[p] = fminf(fminf(fabsf(qr), fabsf(qg)), fabsf(qb)) - band;
                        }
                        // v_cvt_pk_u8_f32: clamp to 0..255, round to nearest (ties to even), insert the byte - one instruction per
                        // channel.  It differs from Math.round (ties up) on exact ties only, by one unit: inside the tolerant
                        // form's promise (a float32 value within 0.9 of the float64 one still rounds to within one unit of the
                        // reference's byte), and in the bytes-equal form a tie is a rounding doubt by definition (distance 0 from
                        // the boundary) and is redone in float64.
                        uint32_t pk = __builtin_amdgcn_cvt_pk_u8_f32(hr * 255.0f, 0u, 0xFF000000u);
                        pk = __builtin_amdgcn_cvt_pk_u8_f32(hg * 255.0f, 1u, pk);
                        packed[p] = __builtin_amdgcn_cvt_pk_u8_f32(hb * 255.0f, 2u, pk);
                        cr[p] = hr; cg[p] = hg; cb[p] = hb;
                    }
                }
                unsigned long long left = 0ull;
#pragma unroll
                for (int p = 0; p < PPL; p++) left |= rem[p];
                if (left == 0ull) break;
            }
        }
#pragma unroll
        for (int p = 0; p < PPL; p++)
            if constexpr (!TOL) unsure[p].add_mask(ballot(rdoubt[p] <= 0.0f), WHY_ROUND);   // (lanes without a hit hold 1)
    } else {
#pragma unroll
        for (int p = 0; p < PPL; p++) { cr[p] = 0.0f; cg[p] = 0.0f; cb[p] = 0.0f; shaded_m[p] = 0ull; packed[p] = hdr->bg_packed; }
    }

    stamp(6);
    // ---- stores ---------------------------------------------------------------------------------
    // (which lanes store is worked out again here rather than carried in a register through the whole kernel)
    const int row_min = quirk ? 1 : 0;                  // PTR_QUIRK_ROW_SHIFT: row 0 has no destination, the others move up one row
    const unsigned qshift = quirk ? (unsigned)a.width : 0u;
    // The pixel coordinates are worked out again from the thread index (made opaque: otherwise the compiler keeps the
    // PPL + 1 registers of the top of the kernel alive to here, and at 4 pixels per lane those are the ones that spill).
    {
        unsigned tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const int ln = (int)(tid & 63u), wv = (int)(tid >> 6);
#pragma unroll
        for (int p = 0; p < PPL; p++) {
            px[p].x = blockIdx.x * (PTR_FAST_BLOCK / 64 * 8 * PPL) + wv * (8 * PPL) + p * 8 + (ln & 7);
            px[p].row = blockIdx.y * PTR_TILE_H + (ln >> 3);
        }
    }
    if constexpr (!DBG) {
        // Nearly every wave: its tile lies wholly inside the band, there is no float32 frame and nothing to queue - one
        // address per lane, the slots 32 bytes apart, no lane masks.
        const int x0 = blockIdx.x * (PTR_FAST_BLOCK / 64 * 8 * PPL) + wave * (8 * PPL);
        bool plain = (x0 + 8 * PPL <= a.width) & (tile_row0 + PTR_TILE_H <= a.scan_height) & (a.out32f == nullptr);
        if constexpr (QUIRK) plain &= blockIdx.y != 0;
        if constexpr (HYBRID) {
            unsigned long long um = 0ull;
#pragma unroll
            for (int p = 0; p < PPL; p++) um |= unsure[p].mask;
            plain &= um == 0ull;
        }
        if (plain) {
            uint32_t* dst = out8 + ((unsigned)__umul24((unsigned)a.width, (unsigned)px[0].row) + (unsigned)px[0].x - qshift);
#pragma unroll
            for (int p = 0; p < PPL; p++) dst[8 * p] = packed[p];
            return;
        }
    }
#pragma unroll
    for (int p = 0; p < PPL; p++) {
        const Pixel& r = px[p];
        const bool in_img = (r.x < a.width) & (r.row < a.scan_height);
        const bool st8 = in_img & (r.row >= row_min);
        if constexpr (DBG) {
            packed[p] = unsure[p].why() | (dbg_full_tests << 8) | (dbg_pre_tests << 16);
            // 0x2000: the first 6 pixels of the wave's first row carry the clocks spent in stages 0..5 instead
            if ((a.flags & 0x2000u) && p == 0 && lane < 6) packed[p] = (uint32_t)(ts[lane + 1] - ts[lane]);
        }
        // (width, rows < 2^24 and a call writes fewer than 2^32 pixels: checked by the host)
        const unsigned pix = (unsigned)__umul24((unsigned)a.width, (unsigned)r.row) + (unsigned)r.x;
        if (st8) out8[pix - qshift] = packed[p];
        if (a.out32f) {
            if (in_img) {
                float4* o = reinterpret_cast<float4*>(a.out32f) + (frame * a.out_frame_stride + (uint64_t)pix);
                const bool shaded = lane_of(shaded_m[p]);
                *o = make_float4(shaded ? cr[p] : hdr->bg32[0], shaded ? cg[p] : hdr->bg32[1], shaded ? cb[p] : hdr->bg32[2], 1.0f);
            }
        }
        if constexpr (HYBRID) {
            if (unsure[p].mask == 0ull) continue;      // wave-uniform: nearly every wave leaves here
            const bool may = st8 | (in_img & (a.out32f != nullptr));
            const bool full = TOL || (r.idx & 0x200) != 0 || r.idx < 0;      // (a miss has nothing to re-shade; the tolerance mode has no re-shade)
            if (unsure[p].any() & may) {
                // one counter word would saturate (~88 returning atomics/us chip-wide): the queue is
                // cut into n_queues sub-queues, each with its own counter line, chosen by workgroup
                uint32_t bid = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
                asm volatile("" : "+s"(bid));           // keeps the address arithmetic below behind the branch above
                const uint32_t sub = bid & (a.n_queues - 1u);       // n_queues is a power of two
                const uint32_t gid = (uint32_t)(frame * frame_px + (uint64_t)pix);
                if (full) {
                    // kind 0, from the front of the sub-queue: re-trace from scratch
                    const uint32_t q = atomicAdd(a.counter + sub * PTR_COUNTER_STRIDE, 1u);
                    if (q < a.queue_capacity) a.queue[(uint64_t)sub * a.queue_capacity + q] = gid;
                } else {
                    // kind 1, from its back (a pixel is queued once, so the two never meet): nearest element and shadow
                    // verdicts are certain, only the shading is redone in float64
                    const uint32_t q = atomicAdd(a.counter + sub * PTR_COUNTER_STRIDE + 1, 1u);
                    if (q < a.queue_capacity) {
                        const uint64_t slot = (uint64_t)sub * a.queue_capacity + (a.queue_capacity - 1u - q);
                        a.queue[slot] = gid;
                        a.qinfo[slot] = (((uint32_t)r.idx & 0xFFu) << 24) | (((uint32_t)r.idx >> 8) & 0x555555u);
                    }
                }
            }
        }
    }
}

template <int MODE, bool DBG, int PPL, bool GENERAL>
static hipError_t launch_camera(const RenderArgs& a, int n_frames, hipStream_t s)
{
    const int tw = PTR_FAST_BLOCK / 64 * 8 * PPL;
    dim3 grid((a.width + tw - 1) / tw, (a.scan_height + PTR_TILE_H - 1) / PTR_TILE_H, n_frames);
    if constexpr (DBG) {                         // (diagnostics: the reasons come in the unshifted layout whatever the flag says)
        hipLaunchKernelGGL((render_fast_kernel<MODE, DBG, PPL, GENERAL, false>), grid, dim3(PTR_FAST_BLOCK), 0, s, a);
    } else if (a.flags & 0x0001u /* PTR_QUIRK_ROW_SHIFT */) {
        hipLaunchKernelGGL((render_fast_kernel<MODE, DBG, PPL, GENERAL, true>), grid, dim3(PTR_FAST_BLOCK), 0, s, a);
    } else {
        hipLaunchKernelGGL((render_fast_kernel<MODE, DBG, PPL, GENERAL, false>), grid, dim3(PTR_FAST_BLOCK), 0, s, a);
    }
    return hipGetLastError();
}

template <int MODE, bool DBG, int PPL>
static hipError_t launch_variant(const RenderArgs& a, int n_frames, hipStream_t s)
{
    return a.cam.general ? launch_camera<MODE, DBG, PPL, true>(a, n_frames, s) : launch_camera<MODE, DBG, PPL, false>(a, n_frames, s);
}

// pixels per lane by mode: the tolerant form carries fewer values per pixel and takes a wider tile at one wave per SIMD less
int fast_kernel_ppl(int mode) { return mode == 2 ? PTR_FAST_PPL_TOL : PTR_FAST_PPL; }
int fast_kernel_rows_per_block() { return PTR_TILE_H; }
int fast_kernel_cols_per_block(int mode) { return PTR_FAST_BLOCK / 64 * 8 * fast_kernel_ppl(mode); }

hipError_t launch_render_fast(const RenderArgs& a, int n_frames, int mode, hipStream_t s)
{
    if (mode == 2) return launch_variant<2, false, PTR_FAST_PPL_TOL>(a, n_frames, s);
    if (mode == 1) return launch_variant<1, false, PTR_FAST_PPL>(a, n_frames, s);
    if (a.flags & 0x8000u /* PTR_DEBUG_REASONS */) return launch_variant<0, true, PTR_FAST_PPL>(a, n_frames, s);
    return launch_variant<0, false, PTR_FAST_PPL>(a, n_frames, s);
}

}  // namespace ptr
