// ptr_kernels_many.hip — the float32 kernel for scenes with many spheres (BASELINE.json config 5:
// 10 000 spheres, brute-force intersection, sphere list staged through LDS).
//
// Same function as ptr_kernels_fast.hip (RayTracer.render's per-pixel body, RT/src/RayTracer.js:163-193
// -> trace :18-95 -> Sphere.intersect, RT/src/Sphere.js:15-33) and the same contract: every
// discontinuous decision carries a conservative error bound, pixels with a decision inside its bound are
// queued for the float64 kernel.  What changes with the sphere count is where the time goes: almost
// every (pixel, sphere) test is a clear miss, so the loops are split in two levels.
//
//   level 1 (hot loop, per sphere and pixel 4 + 4 packed float32 FMAs for two pixels):
//     primary: disc~ = (l.d)^2 - (cc - 2 md) >= 0      cc = |l|^2 - r^2, md = error bound of the form
//     shadow:  (a.L)^2 + 2 cL - thr^2 >= 0             the (light, occluder) cone test of PairQ
//     Both are one-sided: a test that fails IS a miss in exact arithmetic.  16-byte records
//     (lx ly lz -cc~ / ax ay az -thr^2) are staged through LDS in chunks by the whole workgroup and read
//     back with broadcast ds_read_b128; each lane carries 4 pixels (two packed pairs), so one LDS read
//     feeds 4 tests.  The verdicts of 4 spheres x 4 pixels are folded with max3 into one compare.
//   level 2 (rare: only when some lane of the wave passes level 1): the sphere's full record comes in
//     through scalar loads and the test is redone in its well-conditioned form, disc' = r^2 - |l x d|^2
//     (no cancellation between |l|^2 and b^2), with its tight bound.  With 10 000 small far spheres the
//     b^2 - cc form alone would leave 40 % of the pixels undecided; this form leaves well under 1 %.
#include <hip/hip_runtime.h>
#include "ptr_device.h"
#include "ptr_kernels.h"
#include "ptr_fast_common.h"

namespace ptr {

#ifndef PTR_MANY_CHUNK
#define PTR_MANY_CHUNK 1024          // 16 KB of LDS per workgroup
#endif
#define PTR_MANY_PPL   4
#define PTR_MANY_TILE  32            // workgroup tile 32 x 32: 2 x 2 waves of 16 x 16, 2 x 2 sub-tiles of 8 x 8 per lane

typedef float f2 __attribute__((ext_vector_type(2)));
typedef const f4 PTR_CONST* ConstF4Q;                       // non-volatile: level 2 loads sit in a branch anyway
__device__ __forceinline__ f2 pk_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 splat(float v) { return f2{v, v}; }
__device__ __forceinline__ float max3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }

constexpr float NEVER = -3.0e38f;
constexpr float K2_MAX = 1.0e-3f;        // widest cone-test slack level 1 works with (see the shadow loop)

// ---- two pixels per operation: every per-pixel quantity of a lane is a float2 (one v_pk_* instruction
// ---- serves both); comparisons, selects, rsq/rcp stay per component
struct V3p { f2 x, y, z; };
__device__ __forceinline__ f2 dotp(const V3p& a, const V3p& b) { return pk_fma(a.z, b.z, pk_fma(a.y, b.y, a.x * b.x)); }
__device__ __forceinline__ f2 dots(float ax, float ay, float az, const V3p& b)      // uniform vector . per-pixel vector
{ return pk_fma(splat(az), b.z, pk_fma(splat(ay), b.y, splat(ax) * b.x)); }
__device__ __forceinline__ V3p crossp(const V3p& a, const V3p& b)
{ return V3p{pk_fma(a.y, b.z, -(a.z * b.y)), pk_fma(a.z, b.x, -(a.x * b.z)), pk_fma(a.x, b.y, -(a.y * b.x))}; }
__device__ __forceinline__ f2 rsq2(f2 v) { return f2{rsq(v.x), rsq(v.y)}; }
__device__ __forceinline__ f2 rcp2(f2 v) { return f2{rcp(v.x), rcp(v.y)}; }
__device__ __forceinline__ f2 max2(f2 a, f2 b) { return f2{fmaxf(a.x, b.x), fmaxf(a.y, b.y)}; }
__device__ __forceinline__ f2 min2(f2 a, f2 b) { return f2{fminf(a.x, b.x), fminf(a.y, b.y)}; }
__device__ __forceinline__ f2 med2(f2 a, f2 b, f2 c) { return f2{__builtin_amdgcn_fmed3f(a.x, b.x, c.x), __builtin_amdgcn_fmed3f(a.y, b.y, c.y)}; }
__device__ __forceinline__ f2 abs2(f2 v) { return f2{fabsf(v.x), fabsf(v.y)}; }
__device__ __forceinline__ f2 sel2(bool cx, bool cy, f2 a, f2 b) { return f2{cx ? a.x : b.x, cy ? a.y : b.y}; }

// The discriminant of Sphere.intersect in whichever of its two float32 forms has the smaller error
// bound for this ray:  A = r^2 - |l x d|^2  (bound 2 hed, per sphere: good when r << |l|)  or
// B = b^2 - cc  (bound from |error of b| <= eb: good for a grazing ray from close to a huge sphere, cc << |l|^2).
struct Discp { f2 b, disc, err; };
template <bool BOTH>
__device__ __forceinline__ Discp disc_pair(const V3p& l, f2 cc, f2 r2, f2 hed, f2 eb, const V3p& d)
{
    Discp o;
    o.b = dotp(l, d);
    const V3p cr = crossp(l, d);
    o.disc = r2 - dotp(cr, cr);
    o.err = 2.0f * hed;
    if constexpr (BOTH) {
        const f2 dB = pk_fma(o.b, o.b, -cc);
        const f2 eB = 1.0001f * pk_fma(pk_fma(splat(2.0f), abs2(o.b), eb), eb, (2.0f * EPS) * pk_fma(o.b, o.b, abs2(cc)));
        const bool ux = eB.x < o.err.x, uy = eB.y < o.err.y;
        o.disc = sel2(ux, uy, dB, o.disc);
        o.err = sel2(ux, uy, eB, o.err);
    }
    return o;
}

// the two leading f4 of a SphereP32 record / a PairQ record, requested one sphere ahead (SMALL)
struct P32L { f4 g0, g1; };
__device__ __forceinline__ P32L load_p32l(ConstF4 rec) { return P32L{rec[0], rec[1]}; }
__device__ __forceinline__ void arrive(const P32L& q) { asm volatile("" ::PTR_F4_IN(q.g0), PTR_F4_IN(q.g1)); }
__device__ __forceinline__ void arrive_after(const P32L& q, float dep) { asm volatile("" ::PTR_F4_IN(q.g0), PTR_F4_IN(q.g1), "v"(dep)); }
__device__ __forceinline__ void arrive1(const f4& q) { asm volatile("" ::PTR_F4_IN(q)); }
__device__ __forceinline__ void arrive1_after(const f4& q, float dep) { asm volatile("" ::PTR_F4_IN(q), "v"(dep)); }

// per-lane flag word of a pixel: bit 0 = needs the full float64 re-trace, bit 1 = needs the float64
// re-shade only; the reasons (WHY_*) sit above bit 8 for the diagnostics build
constexpr unsigned FL_FULL = 1u, FL_SHADE = 2u;
template <bool DBG> __device__ __forceinline__ unsigned flag(bool c, unsigned kind, unsigned why)
{
    return c ? (DBG ? (kind | (why << 8)) : kind) : 0u;
}
__device__ __forceinline__ bool any_lane(bool c) { return ballot(c) != 0ull; }

// SMALL: few spheres — the level-1 records come through scalar loads into SGPRs instead of LDS (no staging,
// no barriers), one sphere per level-1 compare.
#ifdef PTR_MANY_WAVES
#define PTR_MANY_OCC __attribute__((amdgpu_waves_per_eu(PTR_MANY_WAVES, PTR_MANY_WAVES)))
#else
#define PTR_MANY_OCC
#endif
template <bool HYBRID, bool DBG, bool SMALL>
__global__ void __launch_bounds__(PTR_BLOCK) PTR_MANY_OCC render_many_kernel(RenderArgs a)
{
    constexpr int PPL = PTR_MANY_PPL, NP = PPL / 2;      // pixel p = 2 * pair + component
    constexpr bool BOUNDS = HYBRID || DBG;                // the error-bound arithmetic has a consumer
    const bool tolerant = HYBRID && (a.flags & 0x0020u) != 0u;   // PTR_MODE_TOLERANT (include/ptr.h)
    __shared__ f4 tile[SMALL ? 1 : PTR_MANY_CHUNK];

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int frame = blockIdx.z;
    const uint64_t frame_px = (uint64_t)a.width * (uint64_t)a.scan_height;
    uint32_t* out8 = a.out8 + frame * a.out_frame_stride;
    const bool quirk = (a.flags & 0x0001u) != 0;      // PTR_QUIRK_ROW_SHIFT

    const unsigned char* block = a.scenes + (uint64_t)(a.first_frame + frame) * a.frame_stride;
    const ConstHeader hdr = (ConstHeader)(uintptr_t)block;
    const ConstF4Q rec0 = (ConstF4Q)(uintptr_t)(block + a.off_p32);
    const ConstM32 cm32 = (ConstM32)(uintptr_t)(block + a.off_m32);
    const ConstI32 clights = (ConstI32)(uintptr_t)(block + a.off_lights);
    const SphereP32* gp32 = reinterpret_cast<const SphereP32*>(block + a.off_p32);
    const SphereM32* gm32 = reinterpret_cast<const SphereM32*>(block + a.off_m32);
    const PairQ* gpairs = reinterpret_cast<const PairQ*>(block + a.off_pairs);
    const int n = a.n;

    // ---- this lane's 4 pixels: pair 0 = (x0, row0), (x0 + 8, row0); pair 1 = the same 8 rows below ----
    const int x0 = blockIdx.x * PTR_MANY_TILE + (wave & 1) * 16 + (lane & 7);
    const int row0 = blockIdx.y * PTR_MANY_TILE + (wave >> 1) * 16 + (lane >> 3);
    int px_x[PPL], px_row[PPL];
    bool live[PPL];
    V3p d[NP];
    f2 tnear[NP], t2[NP], etmax[NP];
    int idx[PPL];
    unsigned fl[PPL];
    [[maybe_unused]] unsigned dbg_l2_primary = 0, dbg_l2_shadow = 0;   // DBG: level 2 calls of this wave
    bool any_live = false;
#pragma unroll
    for (int p = 0; p < PPL; p++) {
        px_x[p] = x0 + (p & 1) * 8;
        px_row[p] = row0 + (p >> 1) * 8;
        live[p] = (px_x[p] < a.width) & (px_row[p] < a.scan_height);
        if (quirk & live[p] & (px_row[p] == 0)) {      // RayTracer.js:169: dropped row / untouched last row
            out8[(uint64_t)a.width * (a.scan_height - 1) + px_x[p]] = 0u;
            live[p] = a.out32f != nullptr;
        }
        idx[p] = -1; fl[p] = 0u;
        any_live |= live[p];
    }
#pragma unroll
    for (int q = 0; q < NP; q++) {
        // camera rays (RayTracer.js:171-174)
        const f2 fx{(float)px_x[2 * q] + 0.5f, (float)px_x[2 * q + 1] + 0.5f};
        const float fy = (float)global_y(a, px_row[2 * q]) + 0.5f;
        const f2 xx = pk_fma(fx, splat(a.camf.sx), splat(a.camf.ox));
        const f2 yy = splat(fmaf(fy, a.camf.sy, a.camf.oy));
        V3p v;
        if (a.cam.general) {
            v.x = pk_fma(xx, splat(a.camf.right[0]), pk_fma(yy, splat(a.camf.up[0]), splat(a.camf.fwd[0])));
            v.y = pk_fma(xx, splat(a.camf.right[1]), pk_fma(yy, splat(a.camf.up[1]), splat(a.camf.fwd[1])));
            v.z = pk_fma(xx, splat(a.camf.right[2]), pk_fma(yy, splat(a.camf.up[2]), splat(a.camf.fwd[2])));
        } else {
            v = V3p{xx, yy, splat(-1.0f)};
        }
        // a pixel outside the band gets a NaN direction: it fails every level-1 test and stays a miss
        const f2 len2 = dotp(v, v);
        const f2 inv{live[2 * q] ? rsq(len2.x) : __builtin_nanf(""), live[2 * q + 1] ? rsq(len2.y) : __builtin_nanf("")};
        d[q] = V3p{v.x * inv, v.y * inv, v.z * inv};
        tnear[q] = splat(3.0e38f); t2[q] = splat(3.0e38f); etmax[q] = splat(0.0f);
    }
    // a wave with no pixel in the band still takes part in the staging and its barriers
    const bool wave_live = any_lane(any_live);

    // ---- nearest hit (RayTracer.js:26-41) -------------------------------------------------------
    // level 2 for sphere i: the full test for every pixel pair with a lane that passed level 1 (w = the
    // level-1 record's last word)
    auto primary_l2 = [&](int i, float lx, float ly, float lz, float w) {
        const ConstF4Q rec = rec0 + 4 * (size_t)i;
        const f4 g0 = rec[0], g1 = rec[1], g2 = rec[2], g3 = rec[3];
        const float cc = g0.w, eb = g2.w;
        const bool outside = g2.x < 0.0f, cc_unsure = g2.y != 0.0f;
        const V3p l{splat(g0.x), splat(g0.y), splat(g0.z)};
#pragma unroll
        for (int q = 0; q < NP; q++) {
            const f2 b1 = dots(lx, ly, lz, d[q]);
            const f2 e1 = pk_fma(b1, b1, splat(w));
            const bool px = e1.x >= 0.0f, py = e1.y >= 0.0f;
            if (!any_lane(px | py)) continue;                // wave-uniform
            const Discp k = disc_pair<false>(l, splat(cc), splat(g1.z), splat(fabsf(g3.y)), splat(eb), d[q]);
            const bool hx = px & (k.disc.x >= 0.0f) & (k.b.x >= 0.0f), hy = py & (k.disc.y >= 0.0f) & (k.b.y >= 0.0f);
            if constexpr (BOUNDS) {
                const bool cx = (k.disc.x > k.err.x) & (k.b.x > eb), cy = (k.disc.y > k.err.y) & (k.b.y > eb);
                const bool sx = px & (k.disc.x >= -k.err.x) & (k.b.x >= -eb), sy = py & (k.disc.y >= -k.err.y) & (k.b.y >= -eb);
                fl[2 * q] |= flag<DBG>(sx & !cx, FL_FULL, WHY_PRIMARY) | flag<DBG>(cc_unsure & (sx | hx), FL_FULL, WHY_CC);
                fl[2 * q + 1] |= flag<DBG>(sy & !cy, FL_FULL, WHY_PRIMARY) | flag<DBG>(cc_unsure & (sy | hy), FL_FULL, WHY_CC);
            }
            if (any_lane(hx | hy)) {
                const f2 dm = max2(k.disc, splat(1e-30f));
                const f2 s = rsq2(dm);
                const f2 bt = pk_fma(dm, s, k.b);                              // b + thc
                // t0 = b - thc = cc / (b + thc) from outside, t1 = b + thc from inside (RayTracer.js:32-34)
                f2 t, e;
                if (outside) {                                                // wave-uniform
                    const f2 rbt = rcp2(bt);
                    t = splat(cc) * rbt;
                    if constexpr (BOUNDS) e = t * pk_fma(pk_fma(0.5f * k.err, s, splat(eb)), rbt, splat(4.0f * EPS));
                } else {
                    t = bt;
                    if constexpr (BOUNDS) e = pk_fma(splat(2.0f * EPS), t, pk_fma(0.5f * k.err, s, splat(eb)));
                }
                t = sel2(hx, hy, t, splat(3.0e38f));                           // a lane that does not hit changes nothing
                if constexpr (BOUNDS) {
                    etmax[q] = max2(etmax[q], sel2(hx, hy, e, splat(0.0f)));
                    t2[q] = med2(tnear[q], t2[q], t);
                }
                idx[2 * q] = t.x < tnear[q].x ? i : idx[2 * q];
                idx[2 * q + 1] = t.y < tnear[q].y ? i : idx[2 * q + 1];
                tnear[q] = min2(tnear[q], t);
            }
        }
    };

    if constexpr (SMALL) {
        if (!wave_live) return;                               // no barriers here: a wave outside the band leaves
        auto primary_l1 = [&](const P32L& q, int i) {
            const float w = fmaf(2.0f, q.g1.x, -q.g0.w);      // -(cc - 2 md)
            const f2 b0 = dots(q.g0.x, q.g0.y, q.g0.z, d[0]), b1 = dots(q.g0.x, q.g0.y, q.g0.z, d[1]);
            const f2 e0 = pk_fma(b0, b0, splat(w)), e1 = pk_fma(b1, b1, splat(w));
            if (any_lane(fmaxf(max3(e0.x, e0.y, e1.x), e1.y) >= 0.0f)) {
                if constexpr (DBG) dbg_l2_primary++;
                primary_l2(i, q.g0.x, q.g0.y, q.g0.z, w);
            }
        };
        ConstF4 rec = (ConstF4)(uintptr_t)(block + a.off_p32);
        P32L A = load_p32l(rec);
        for (int i = 0; i < n; i += 2) {
            arrive(A);
            const P32L B = load_p32l(rec + 4);
            primary_l1(A, i);
            arrive_after(B, tnear[NP - 1].y);
            A = load_p32l(rec + 8);
            if (i + 1 < n) primary_l1(B, i + 1);
            rec += 8;
        }
        arrive(A);
    } else
    for (int c0 = 0; c0 < n; c0 += PTR_MANY_CHUNK) {
        const int cnt = (n - c0 < PTR_MANY_CHUNK) ? n - c0 : PTR_MANY_CHUNK;
        const int cnt4 = (cnt + 3) & ~3;
        __syncthreads();                                      // the previous chunk has been consumed
        for (int t = threadIdx.x; t < cnt4; t += PTR_BLOCK) {
            f4 v{0.0f, 0.0f, 0.0f, NEVER};
            if (c0 + t < n) {
                const SphereP32& s = gp32[c0 + t];
                v = f4{s.lx, s.ly, s.lz, -(s.cc - 2.0f * s.md)};
            }
            tile[t] = v;
        }
        __syncthreads();
        if (!wave_live) continue;
        for (int j = 0; j < cnt4; j += 4) {
            f2 m0 = splat(NEVER), m1 = splat(NEVER);
#pragma unroll
            for (int u = 0; u < 4; u += 2) {
                const f4 sa = tile[j + u], sb = tile[j + u + 1];
                const f2 ba0 = dots(sa.x, sa.y, sa.z, d[0]), ba1 = dots(sa.x, sa.y, sa.z, d[1]);
                const f2 bb0 = dots(sb.x, sb.y, sb.z, d[0]), bb1 = dots(sb.x, sb.y, sb.z, d[1]);
                const f2 da0 = pk_fma(ba0, ba0, splat(sa.w)), da1 = pk_fma(ba1, ba1, splat(sa.w));
                const f2 db0 = pk_fma(bb0, bb0, splat(sb.w)), db1 = pk_fma(bb1, bb1, splat(sb.w));
                m0 = f2{max3(m0.x, da0.x, db0.x), max3(m0.y, da0.y, db0.y)};
                m1 = f2{max3(m1.x, da1.x, db1.x), max3(m1.y, da1.y, db1.y)};
            }
            if (!any_lane(fmaxf(max3(m0.x, m0.y, m1.x), m1.y) >= 0.0f)) continue;
            // some lane may hit one of these 4 spheres: redo them one at a time
#pragma unroll 1
            for (int u = 0; u < 4; u++) {
                const int i = c0 + j + u;
                if (i >= n) break;
                const f4 s = tile[j + u];
                if constexpr (DBG) dbg_l2_primary++;
                primary_l2(i, s.x, s.y, s.z, s.w);
            }
        }
    }

    // miss: the background, bit-exact by construction (RayTracer.js:43-46)
    const float bgr = hdr->bg32[0], bgg = hdr->bg32[1], bgb = hdr->bg32[2];
    const uint32_t bg_packed = hdr->bg_packed;
    float cr[PPL], cg[PPL], cb[PPL];
    uint32_t packed[PPL];
    bool valid[PPL];
    bool any_valid = false;
    // hit frame, per pair
    V3p N[NP];
    f2 ts[NP], eP[NP], eN[NP];
    f2 ar[NP], ag[NP], ab[NP], wsum[NP], eLmax[NP];
    // Pixels whose nearest element is certain but whose shading is not (a shadow verdict or the 8-bit
    // rounding inside its bound) are queued for a float64 re-shade only: it takes the element and, per
    // light, 2 bits: 0 lit or facing away for certain, 1 blocked for certain, 2 look for yourself.
    // Pixels with a "look for yourself" light go to the full re-trace instead: with many spheres one
    // occluder loop in a wave of otherwise short entries would hold the whole wave.
    uint32_t codes[PPL];
#pragma unroll
    for (int p = 0; p < PPL; p++) {
        codes[p] = 0u;
        cr[p] = bgr; cg[p] = bgg; cb[p] = bgb; packed[p] = bg_packed;
        valid[p] = (idx[p] >= 0) & live[p];
        any_valid |= valid[p];
    }
    const bool wave_valid = any_lane(any_valid);
    if constexpr (SMALL) {
        if (!wave_valid) goto store;                          // every pixel of the wave is background
    }
    {
#pragma unroll
    for (int q = 0; q < NP; q++) {
        const int p0 = 2 * q, p1 = 2 * q + 1;
        // nearest-vs-second ordering, far clip (t < 1e8, RayTracer.js:19,36)
        if constexpr (BOUNDS) {
            const f2 gap = t2[q] - tnear[q], lim = 2.0f * etmax[q];
            fl[p0] |= flag<DBG>((gap.x <= lim.x) & valid[p0], FL_FULL, WHY_ORDER) | flag<DBG>((tnear[q].x >= FAR_UNSURE) & valid[p0], FL_FULL, WHY_FAR);
            fl[p1] |= flag<DBG>((gap.y <= lim.y) & valid[p1], FL_FULL, WHY_ORDER) | flag<DBG>((tnear[q].y >= FAR_UNSURE) & valid[p1], FL_FULL, WHY_FAR);
        }
        // the winner once more, with the better of the two discriminant forms, to get its hit frame
        const int i0 = valid[p0] ? idx[p0] : 0, i1 = valid[p1] ? idx[p1] : 0;
        const float4 a0 = *reinterpret_cast<const float4*>(&gp32[i0].lx), a1 = *reinterpret_cast<const float4*>(&gp32[i1].lx);
        const float2 b0 = *reinterpret_cast<const float2*>(&gp32[i0].r2), b1 = *reinterpret_cast<const float2*>(&gp32[i1].r2);
        const f2 eb_w{gp32[i0].eb, gp32[i1].eb};
        const float2 c0 = *reinterpret_cast<const float2*>(&gp32[i0].inv_r), c1 = *reinterpret_cast<const float2*>(&gp32[i1].inv_r);
        const V3p lw{f2{a0.x, a1.x}, f2{a0.y, a1.y}, f2{a0.z, a1.z}};
        const f2 ccw{a0.w, a1.w}, r2w{b0.x, b1.x}, lnw{b0.y, b1.y}, inv_r{c0.x, c1.x}, hedw{fabsf(c0.y), fabsf(c1.y)};      // (the sign is a flag for the few-sphere kernel)
        const Discp k = disc_pair<BOUNDS>(lw, ccw, r2w, hedw, eb_w, d[q]);
        const f2 dm = max2(k.disc, splat(1e-30f));
        const f2 s_w = rsq2(dm);
        const f2 bt = pk_fma(dm, s_w, k.b);
        const f2 et_w = pk_fma(0.5f * k.err, s_w, eb_w);
        const f2 rbt = rcp2(bt);
        const bool ox = ccw.x > 0.0f, oy = ccw.y > 0.0f;
        ts[q] = sel2(ox, oy, ccw * rbt, bt);
        eP[q] = sel2(ox, oy, ts[q] * pk_fma(et_w, rbt, splat(4.0f * EPS)), pk_fma(splat(2.0f * EPS), ts[q], et_w));
        const V3p Nn{pk_fma(d[q].x, ts[q], -lw.x), pk_fma(d[q].y, ts[q], -lw.y), pk_fma(d[q].z, ts[q], -lw.z)};   // P - C (Sphere.js:47-51)
        const f2 inv = rsq2(dotp(Nn, Nn));
        const f2 sg = sel2(ox, oy, inv, -inv);                               // RayTracer.js:54-58
        N[q] = V3p{Nn.x * sg, Nn.y * sg, Nn.z * sg};
        eN[q] = pk_fma(eP[q] + EPS * (lnw + 2.0f * ts[q]), inv_r, splat(3.0f * EPS));
        ar[q] = splat(0.0f); ag[q] = splat(0.0f); ab[q] = splat(0.0f); wsum[q] = splat(0.0f); eLmax[q] = splat(0.0f);
    }

    // ---- direct light + binary shadow (RayTracer.js:61-91) -------------------------------------
    const int nl = hdr->n_lights;
    for (int k = 0; k < nl; k++) {
        const int il = clights[k];
        const int prow = k < PTR_MAX_PAIR_LIGHTS ? k : (n < PTR_MAX_PAIR_LIGHTS ? n : PTR_MAX_PAIR_LIGHTS);
        const PairQ* pairs = gpairs + (size_t)prow * (size_t)pair_stride(n);
        const f4 lg0 = rec0[4 * (size_t)il], lg1 = rec0[4 * (size_t)il + 1];
        const float ler = cm32[il].er, leg = cm32[il].eg, leb = cm32[il].eb, lemax = cm32[il].emax;
        // per pixel: k2s > NEVER <=> the lane still needs a verdict ("open"); maybe = some occluder's test was
        // inside its bound; blocked32 = the plain float32 verdict (float32-only output of the diagnostics build)
        V3p L[NP];
        f2 ratio[NP], eL[NP], cL[NP], k2s[NP];
        bool maybe[PPL], blocked32[PPL];
        bool any_open = false;
#pragma unroll
        for (int q = 0; q < NP; q++) {
            const V3p Lv{pk_fma(-d[q].x, ts[q], splat(lg0.x)), pk_fma(-d[q].y, ts[q], splat(lg0.y)), pk_fma(-d[q].z, ts[q], splat(lg0.z))};
            const f2 invl = rsq2(dotp(Lv, Lv));
            L[q] = V3p{Lv.x * invl, Lv.y * invl, Lv.z * invl};
            eL[q] = pk_fma(pk_fma(splat(EPS), splat(lg1.w) + 3.0f * ts[q], eP[q]), invl, splat(3.0f * EPS));   // |error of L|
            ratio[q] = dotp(N[q], L[q]);
            cL[q] = eL[q] + splat(3.0f * EPS);
            const f2 k2 = pk_fma(splat(2.0f), cL[q], splat(8.0f * EPS));
            // A direction this vague (a grazing hit on a huge sphere) would pass the cone test of every
            // occluder: hybrid mode hands the light to the float64 kernel at once, float32-only mode tests
            // with the widest slack level 1 works with (tolerance mode: no longer one-sided there).
            const bool vx = HYBRID & (k2.x > K2_MAX), vy = HYBRID & (k2.y > K2_MAX);
            const bool ox = (ratio[q].x > 0.0f) & valid[2 * q] & !vx, oy = (ratio[q].y > 0.0f) & valid[2 * q + 1] & !vy;
            maybe[2 * q] = false; maybe[2 * q + 1] = false; blocked32[2 * q] = false; blocked32[2 * q + 1] = false;
            // (a.L)^2 >= thr^2 - 2 cL (+ rounding of thr^2 and of the dot) <=> level 1 passes; closed lanes never do
            k2s[q] = sel2(ox, oy, min2(k2, splat(K2_MAX)), splat(NEVER));
            any_open |= ox | oy;
        }
        bool wave_open = any_lane(any_open);

        // level 2 for occluder j: the full test for every pixel pair with a lane that passed the cone test
        auto shadow_l2 = [&](int j, float ax, float ay, float az, float w) {
            const ConstF4Q rec = rec0 + 4 * (size_t)j;
            const f4 g0 = rec[0], g1 = rec[1], g3 = rec[3];
            const float rj = g1.z * g3.x;                                    // r = r^2 / r
            bool still = false;
#pragma unroll
            for (int q = 0; q < NP; q++) {
                const f2 c = dots(ax, ay, az, L[q]);
                const f2 u = pk_fma(c, c, k2s[q]) + splat(w);
                const bool px = u.x >= 0.0f, py = u.y >= 0.0f;
                if (any_lane(px | py)) {                                     // wave-uniform
                    const f2 t = ts[q];
                    const f2 bias = splat((float)PTR_BIAS);
                    // shadow-ray origin P + N * bias, then centre - origin
                    const V3p lj{splat(g0.x) - pk_fma(N[q].x, bias, d[q].x * t), splat(g0.y) - pk_fma(N[q].y, bias, d[q].y * t),
                                 splat(g0.z) - pk_fma(N[q].z, bias, d[q].z * t)};
                    const f2 bj = dotp(lj, L[q]);
                    const V3p c3 = crossp(lj, L[q]);
                    const f2 dp2 = dotp(c3, c3);
                    const f2 disc = splat(g1.z) - dp2;
                    bool cx, cy;
                    if constexpr (BOUNDS) {
                        // |lj| <= A; error of lj <= dlj4 / 4 (generously); error of L <= eL:
                        //   the distance |lj x L| of the centre from the ray is off by at most ec, bj by at most ec
                        const f2 A = splat(g1.w) + t;
                        const f2 dlj4 = pk_fma(splat(4.0f * EPS), A, 4.0f * eP[q]);
                        const f2 ec = pk_fma(A, cL[q], dlj4);
                        const f2 band = pk_fma(ec, pk_fma(splat(2.0001f), splat(rj), ec), (8.0f * EPS) * (dp2 + splat(g1.z)));
                        cx = px & (bj.x > ec.x) & (disc.x > band.x); cy = py & (bj.y > ec.y) & (disc.y > band.y);
                        maybe[2 * q] |= px & (bj.x >= -ec.x) & (disc.x >= -band.x) & !cx;
                        maybe[2 * q + 1] |= py & (bj.y >= -ec.y) & (disc.y >= -band.y) & !cy;
                        if constexpr (!HYBRID) {
                            blocked32[2 * q] |= px & (bj.x >= 0.0f) & (disc.x >= 0.0f);
                            blocked32[2 * q + 1] |= py & (bj.y >= 0.0f) & (disc.y >= 0.0f);
                        }
                    } else {
                        cx = px & (bj.x >= 0.0f) & (disc.x >= 0.0f);           // float32-only: the plain verdict
                        cy = py & (bj.y >= 0.0f) & (disc.y >= 0.0f);
                    }
                    k2s[q] = sel2(cx, cy, splat(NEVER), k2s[q]);
                }
                still |= (k2s[q].x > 0.5f * NEVER) | (k2s[q].y > 0.5f * NEVER);
            }
            wave_open = any_lane(still);
        };

        if constexpr (SMALL) {
            auto shadow_l1 = [&](const f4& q, int jj) {
                const float w = q.w < 0.0f ? 1.0f : -(q.w * q.w);
                const f2 c0 = dots(q.x, q.y, q.z, L[0]), c1 = dots(q.x, q.y, q.z, L[1]);
                const f2 u0 = pk_fma(c0, c0, k2s[0]) + splat(w), u1 = pk_fma(c1, c1, k2s[1]) + splat(w);
                if (any_lane(fmaxf(max3(u0.x, u0.y, u1.x), u1.y) >= 0.0f) && jj != il) {
                    if constexpr (DBG) dbg_l2_shadow++;
                    shadow_l2(jj, q.x, q.y, q.z, w);
                }
            };
            ConstF4 pr = (ConstF4)(uintptr_t)(block + a.off_pairs) + (size_t)prow * (size_t)pair_stride(n);
            f4 A = pr[0];
            for (int j = 0; j < n && wave_open; j += 2) {
                arrive1(A);
                const f4 B = pr[1];
                shadow_l1(A, j);
                arrive1_after(B, k2s[NP - 1].y);
                A = pr[2];
                if (j + 1 < n && wave_open) shadow_l1(B, j + 1);
                pr += 2;
            }
            arrive1(A);
        } else
        for (int c0 = 0; c0 < n; c0 += PTR_MANY_CHUNK) {
            const int cnt = (n - c0 < PTR_MANY_CHUNK) ? n - c0 : PTR_MANY_CHUNK;
            const int cnt4 = (cnt + 3) & ~3;
            __syncthreads();
            for (int t = threadIdx.x; t < cnt4; t += PTR_BLOCK) {
                f4 v{0.0f, 0.0f, 0.0f, -4.0f};                                // never passes
                if (c0 + t < n) {
                    const PairQ q = pairs[c0 + t];
                    v = f4{q.ax, q.ay, q.az, q.thr < 0.0f ? 1.0f : -(q.thr * q.thr)};
                }
                tile[t] = v;
            }
            __syncthreads();
            if (!wave_open) continue;
            for (int j = 0; j < cnt4 && wave_open; j += 4) {
                f2 m0 = splat(NEVER), m1 = splat(NEVER);
#pragma unroll
                for (int u = 0; u < 4; u += 2) {
                    const f4 sa = tile[j + u], sb = tile[j + u + 1];
                    const f2 ca0 = dots(sa.x, sa.y, sa.z, L[0]), ca1 = dots(sa.x, sa.y, sa.z, L[1]);
                    const f2 cb0 = dots(sb.x, sb.y, sb.z, L[0]), cb1 = dots(sb.x, sb.y, sb.z, L[1]);
                    const f2 ua0 = pk_fma(ca0, ca0, k2s[0]) + splat(sa.w), ua1 = pk_fma(ca1, ca1, k2s[1]) + splat(sa.w);
                    const f2 ub0 = pk_fma(cb0, cb0, k2s[0]) + splat(sb.w), ub1 = pk_fma(cb1, cb1, k2s[1]) + splat(sb.w);
                    m0 = f2{max3(m0.x, ua0.x, ub0.x), max3(m0.y, ua0.y, ub0.y)};
                    m1 = f2{max3(m1.x, ua1.x, ub1.x), max3(m1.y, ua1.y, ub1.y)};
                }
                if (!any_lane(fmaxf(max3(m0.x, m0.y, m1.x), m1.y) >= 0.0f)) continue;
#pragma unroll 1
                for (int u = 0; u < 4; u++) {
                    const int jj = c0 + j + u;
                    if (jj >= n || !wave_open) break;
                    if (jj == il) continue;                                   // RayTracer.js:76
                    const f4 s = tile[j + u];
                    if constexpr (DBG) dbg_l2_shadow++;
                    shadow_l2(jj, s.x, s.y, s.z, s.w);
                }
            }
        }

#pragma unroll
        for (int q = 0; q < NP; q++) {
            const f2 k2 = pk_fma(splat(2.0f), cL[q], splat(8.0f * EPS));
            const f2 near0 = -(eN[q] + eL[q] + splat(8.0f * EPS));           // N.L <= 0 by less than its error
            f2 w2 = splat(0.0f);                                              // this light's weight where it is lit
#pragma unroll
            for (int c = 0; c < 2; c++) {
                const int p = 2 * q + c;
                const float rt = c ? ratio[q].y : ratio[q].x, k2c = c ? k2.y : k2.x, k2o = c ? k2s[q].y : k2s[q].x;
                const bool open = k2o > 0.5f * NEVER;                         // no certain hit blocked the light
                // a "maybe" only matters if no certain hit blocks the light anyway
                const bool unsure_light = open & maybe[p];
                const bool vague = HYBRID & (k2c > K2_MAX);
                const bool vague_lit = vague & (rt > 0.0f) & valid[p];
                if constexpr (BOUNDS) fl[p] |= flag<DBG>(unsure_light | vague_lit, FL_SHADE, WHY_SHADOW);
                if (HYBRID && k < 16) {
                    // a light that was not tested, but for which the float64 kernel may find N.L > 0, is "look for
                    // yourself" too: a vague direction, or N.L <= 0 by less than its error
                    const bool was_open = (rt > 0.0f) & valid[p] & !vague;
                    const bool look = unsure_light | vague_lit | (valid[p] & (rt <= 0.0f) & (rt >= (c ? near0.y : near0.x)));
                    const uint32_t code = look ? 2u : ((was_open & !open) ? 1u : 0u);
                    codes[p] |= code << (2 * k);
                }
                const float lit = (open & !blocked32[p]) ? rt : 0.0f;
                if (c) w2.y = lit; else w2.x = lit;
            }
            // surfaceColor * emission * max(0, N.L) where lit (RayTracer.js:87-89); the surface colour is
            // multiplied in after the loop
            ar[q] = pk_fma(splat(ler), w2, ar[q]);
            ag[q] = pk_fma(splat(leg), w2, ag[q]);
            ab[q] = pk_fma(splat(leb), w2, ab[q]);
            wsum[q] += sel2(w2.x > 0.0f, w2.y > 0.0f, splat(lemax), splat(0.0f));
            eLmax[q] = max2(eLmax[q], sel2(w2.x > 0.0f, w2.y > 0.0f, eL[q], splat(0.0f)));
        }
    }

    // ---- emission, clamp, round, pack (RayTracer.js:93,179-192) with a rounding bound -----------
#pragma unroll
    for (int q = 0; q < NP; q++) {
        const SphereM32 m0 = gm32[valid[2 * q] ? idx[2 * q] : 0], m1 = gm32[valid[2 * q + 1] ? idx[2 * q + 1] : 0];
        const f2 hr = pk_fma(f2{m0.sr, m1.sr}, ar[q], f2{m0.er, m1.er});
        const f2 hg = pk_fma(f2{m0.sg, m1.sg}, ag[q], f2{m0.eg, m1.eg});
        const f2 hb = pk_fma(f2{m0.sb, m1.sb}, ab[q], f2{m0.eb, m1.eb});
        const f2 vr = min2(hr, splat(1.0f)) * 255.0f, vg = min2(hg, splat(1.0f)) * 255.0f, vb = min2(hb, splat(1.0f)) * 255.0f;
        const f2 fr{floorf(vr.x), floorf(vr.y)}, fg{floorf(vg.x), floorf(vg.y)}, fb{floorf(vb.x), floorf(vb.y)};
        const f2 qr = vr - fr, qg = vg - fg, qb = vb - fb;
        if constexpr (BOUNDS) {
            // the float32 colour is off by at most (surface colour * sum of lit emissions) * |error of N.L| (+ rounding
            // of the products and sums: 10 eps, + the float32 emission: 4 eps)
            const f2 band = 255.0f * pk_fma(f2{m0.scmax, m1.scmax} * wsum[q], eN[q] + eLmax[q] + splat(10.0f * EPS),
                                            (4.0f * EPS) * f2{m0.emax, m1.emax}) + splat(2e-5f);
            // bytes-equal form: a channel within `band` of a rounding boundary is redone.  Tolerant form (PTR_MODE_TOLERANT, a
            // wave-uniform flag: one select per pixel pair): the rounding may fall either way, the pixel is redone only when
            // the colour itself may be off by more than 0.9 units - then the rounded bytes could differ by more than one.
            const f2 nearest = tolerant ? splat(0.9f) : min2(min2(abs2(qr - splat(0.5f)), abs2(qg - splat(0.5f))), abs2(qb - splat(0.5f)));
            fl[2 * q] |= flag<DBG>((nearest.x <= band.x) & valid[2 * q], FL_SHADE, WHY_ROUND);
            fl[2 * q + 1] |= flag<DBG>((nearest.y <= band.y) & valid[2 * q + 1], FL_SHADE, WHY_ROUND);
        }
#pragma unroll
        for (int c = 0; c < 2; c++) {
            const int p = 2 * q + c;
            const float r_ = c ? fr.y : fr.x, g_ = c ? fg.y : fg.x, b_ = c ? fb.y : fb.x;
            const uint32_t r8 = (uint32_t)((c ? qr.y : qr.x) >= 0.5f ? r_ + 1.0f : r_);
            const uint32_t g8 = (uint32_t)((c ? qg.y : qg.x) >= 0.5f ? g_ + 1.0f : g_);
            const uint32_t b8 = (uint32_t)((c ? qb.y : qb.x) >= 0.5f ? b_ + 1.0f : b_);
            if (valid[p]) {
                packed[p] = (255u << 24) | (b8 << 16) | (g8 << 8) | r8;
                cr[p] = c ? hr.y : hr.x; cg[p] = c ? hg.y : hg.x; cb[p] = c ? hb.y : hb.x;
            }
        }
    }
    }

store:
    // ---- stores ---------------------------------------------------------------------------------
    const uint32_t bid = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const bool hintable = hdr->n_lights <= 16;                // light verdicts fit the queue's info word
#pragma unroll
    for (int p = 0; p < PPL; p++) {
        if (!live[p]) continue;
        if constexpr (DBG)
            packed[p] = (fl[p] >> 8) | ((dbg_l2_primary < 4095u ? dbg_l2_primary : 4095u) << 8) |
                        ((dbg_l2_shadow < 4095u ? dbg_l2_shadow : 4095u) << 20);
        long long idx_out = (long long)a.width * px_row[p] + px_x[p];
        if (quirk) idx_out -= a.width;
        if (idx_out >= 0) out8[idx_out] = packed[p];
        if (a.out32f) {
            float4* o = reinterpret_cast<float4*>(a.out32f) + (frame * a.out_frame_stride + (uint64_t)a.width * px_row[p] + px_x[p]);
            *o = make_float4(cr[p], cg[p], cb[p], 1.0f);
        }
        if constexpr (HYBRID) {
            if (fl[p] == 0u) continue;
            const uint32_t sub = bid & (a.n_queues - 1u);           // n_queues is a power of two
            const uint32_t gid = (uint32_t)(frame * frame_px + (uint64_t)a.width * px_row[p] + px_x[p]);
            bool loops = false;                                      // some light says "look for yourself"
            for (uint32_t c = codes[p]; c != 0u; c >>= 2) loops |= (c & 3u) == 2u;
            if ((fl[p] & FL_FULL) | !hintable | loops) {
                // kind 0: from the front of the sub-queue
                const uint32_t q = atomicAdd(a.counter + sub * PTR_COUNTER_STRIDE, 1u);
                if (q < a.queue_capacity) a.queue[(uint64_t)sub * a.queue_capacity + q] = gid;
            } else {
                // kind 1: from its back (a pixel is queued once, so the two never meet)
                const uint32_t q = atomicAdd(a.counter + sub * PTR_COUNTER_STRIDE + 1, 1u);
                if (q < a.queue_capacity) {
                    const uint64_t slot = (uint64_t)sub * a.queue_capacity + (a.queue_capacity - 1u - q);
                    a.queue[slot] = gid;
                    a.qinfo[2 * slot] = (uint32_t)idx[p];
                    a.qinfo[2 * slot + 1] = codes[p];
                }
            }
        }
    }
}

template <bool HYBRID, bool DBG>
static hipError_t launch_many_variant(const RenderArgs& a, int n_frames, hipStream_t s)
{
    dim3 grid((a.width + PTR_MANY_TILE - 1) / PTR_MANY_TILE, (a.scan_height + PTR_MANY_TILE - 1) / PTR_MANY_TILE, n_frames);
    if (a.n <= PTR_MANY_THRESHOLD) hipLaunchKernelGGL((render_many_kernel<HYBRID, DBG, true>), grid, dim3(PTR_BLOCK), 0, s, a);
    else                           hipLaunchKernelGGL((render_many_kernel<HYBRID, DBG, false>), grid, dim3(PTR_BLOCK), 0, s, a);
    return hipGetLastError();
}

int many_kernel_rows_per_block() { return PTR_MANY_TILE; }

hipError_t launch_render_many(const RenderArgs& a, int n_frames, bool hybrid, hipStream_t s)
{
    if (hybrid) return launch_many_variant<true, false>(a, n_frames, s);
    if (a.flags & 0x8000u /* PTR_DEBUG_REASONS */) return launch_many_variant<false, true>(a, n_frames, s);
    return launch_many_variant<false, false>(a, n_frames, s);
}

}  // namespace ptr
