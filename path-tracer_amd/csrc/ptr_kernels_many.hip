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

#define PTR_MANY_CHUNK 1024          // 16 KB of LDS per workgroup
#define PTR_MANY_PPL   4
#define PTR_MANY_TILE  32            // workgroup tile 32 x 32: 2 x 2 waves of 16 x 16, 2 x 2 sub-tiles of 8 x 8 per lane

typedef float f2 __attribute__((ext_vector_type(2)));
typedef const f4 PTR_CONST* ConstF4Q;                       // non-volatile: level 2 loads sit in a branch anyway
__device__ __forceinline__ f2 pk_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 splat(float v) { return f2{v, v}; }
__device__ __forceinline__ float max3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }

constexpr float NEVER = -3.0e38f;
constexpr float K2_MAX = 1.0e-3f;        // widest cone-test slack level 1 works with (see the shadow loop)

// The discriminant of Sphere.intersect in whichever of its two float32 forms has the smaller error
// bound for this ray:  r^2 - |l x d|^2  (bound 2 hed, per sphere: good when r << |l|)  or
// b^2 - cc  (bound from |error of b| <= eb: good when the origin is close to a big sphere, cc << |l|^2).
struct Disc { float b, disc, err; };
__device__ __forceinline__ Disc best_disc(float lx, float ly, float lz, float cc, float r2, float hed, float eb, const F3& d)
{
    Disc o;
    o.b = fmaf(lz, d.z, fmaf(ly, d.y, lx * d.x));
    const F3 cr{fmaf(ly, d.z, -lz * d.y), fmaf(lz, d.x, -lx * d.z), fmaf(lx, d.y, -ly * d.x)};
    const float dA = r2 - dot3(cr, cr), eA = 2.0f * hed;
    const float dB = fmaf(o.b, o.b, -cc);
    const float eB = 1.0001f * fmaf(fmaf(2.0f, fabsf(o.b), eb), eb, 2.0f * EPS * fmaf(o.b, o.b, fabsf(cc)));
    const bool useB = eB < eA;
    o.disc = useB ? dB : dA;
    o.err = useB ? eB : eA;
    return o;
}

template <bool HYBRID, bool DBG>
__global__ void __launch_bounds__(PTR_BLOCK) render_many_kernel(RenderArgs a)
{
    constexpr int PPL = PTR_MANY_PPL;
    __shared__ f4 tile[PTR_MANY_CHUNK];

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

    // ---- this lane's 4 pixels -------------------------------------------------------------------
    const int x0 = blockIdx.x * PTR_MANY_TILE + (wave & 1) * 16 + (lane & 7);
    const int row0 = blockIdx.y * PTR_MANY_TILE + (wave >> 1) * 16 + (lane >> 3);
    int px_x[PPL], px_row[PPL];
    bool live[PPL];
    float dxs[PPL], dys[PPL], dzs[PPL];
    float tnear[PPL], t2[PPL], etmax[PPL];
    int idx[PPL];
    unsigned long long live_m[PPL];
    Unsure<DBG> unsure[PPL];
    [[maybe_unused]] unsigned dbg_l2_primary = 0, dbg_l2_shadow = 0;   // DBG: level 2 calls of this wave
#pragma unroll
    for (int p = 0; p < PPL; p++) {
        px_x[p] = x0 + (p & 1) * 8;
        px_row[p] = row0 + (p >> 1) * 8;
        live[p] = (px_x[p] < a.width) & (px_row[p] < a.scan_height);
        if (quirk & live[p] & (px_row[p] == 0)) {      // RayTracer.js:169: dropped row / untouched last row
            out8[(uint64_t)a.width * (a.scan_height - 1) + px_x[p]] = 0u;
            live[p] = a.out32f != nullptr;
        }
        const float xx = fmaf((float)px_x[p] + 0.5f, a.camf.sx, a.camf.ox);          // RayTracer.js:171-174
        const float yy = fmaf((float)global_y(a, px_row[p]) + 0.5f, a.camf.sy, a.camf.oy);
        F3 d;
        if (a.cam.general) {
            d.x = fmaf(xx, a.camf.right[0], fmaf(yy, a.camf.up[0], a.camf.fwd[0]));
            d.y = fmaf(xx, a.camf.right[1], fmaf(yy, a.camf.up[1], a.camf.fwd[1]));
            d.z = fmaf(xx, a.camf.right[2], fmaf(yy, a.camf.up[2], a.camf.fwd[2]));
        } else {
            d = F3{xx, yy, -1.0f};
        }
        const float inv = rsq(dot3(d, d));
        dxs[p] = d.x * inv; dys[p] = d.y * inv; dzs[p] = d.z * inv;
        tnear[p] = 3.0e38f; t2[p] = 3.0e38f; etmax[p] = 0.0f; idx[p] = -1;
        live_m[p] = ballot(live[p]);
    }
    // a wave with no pixel in the band still takes part in the staging and its barriers
    const bool wave_live = (live_m[0] | live_m[1] | live_m[2] | live_m[3]) != 0ull;
    const f2 dx2[2] = {f2{dxs[0], dxs[1]}, f2{dxs[2], dxs[3]}};
    const f2 dy2[2] = {f2{dys[0], dys[1]}, f2{dys[2], dys[3]}};
    const f2 dz2[2] = {f2{dzs[0], dzs[1]}, f2{dzs[2], dzs[3]}};

    // ---- nearest hit (RayTracer.js:26-41) -------------------------------------------------------
    // level 2 for sphere i and the lanes pm[p] that passed level 1
    auto primary_l2 = [&](int i, const unsigned long long (&pm)[PPL]) {
        const ConstF4Q rec = rec0 + 4 * (size_t)i;
        const f4 g0 = rec[0], g1 = rec[1], g2 = rec[2], g3 = rec[3];
        const float cc = g0.w, eb = g2.w;
        const bool outside = g2.x < 0.0f, cc_unsure = g2.y != 0.0f;
#pragma unroll
        for (int p = 0; p < PPL; p++) {
            if (pm[p] == 0ull) continue;                     // wave-uniform
            const bool pass = lane_of(pm[p]);
            const Disc q = best_disc(g0.x, g0.y, g0.z, cc, g1.z, g3.y, eb, F3{dxs[p], dys[p], dzs[p]});
            const unsigned long long certain = ballot(q.disc > q.err) & ballot(q.b > eb);
            const unsigned long long possible = ballot(q.disc >= -q.err) & ballot(q.b >= -eb) & pm[p];
            const unsigned long long hit_m = ballot(q.disc >= 0.0f) & ballot(q.b >= 0.0f) & pm[p];
            unsure[p].add_mask(possible & ~certain, WHY_PRIMARY);
            unsure[p].add_mask(cc_unsure ? (possible | hit_m) : 0ull, WHY_CC);
            if (pass & (q.disc >= 0.0f) & (q.b >= 0.0f)) {
                const float dm = fmaxf(q.disc, 1e-30f);
                const float s = rsq(dm);
                const float bt = fmaf(dm, s, q.b);                             // b + thc
                const float et = fmaf(0.5f * q.err, s, eb);                    // |error of (b + thc)|
                const float rbt = rcp(bt);
                // t0 = b - thc = cc / (b + thc) from outside, t1 = b + thc from inside (RayTracer.js:32-34)
                const float t = outside ? cc * rbt : bt;
                const float e = outside ? t * fmaf(et, rbt, 4.0f * EPS) : fmaf(2.0f * EPS, t, et);
                etmax[p] = fmaxf(etmax[p], e);
                t2[p] = __builtin_amdgcn_fmed3f(tnear[p], t2[p], t);
                idx[p] = t < tnear[p] ? i : idx[p];
                tnear[p] = fminf(tnear[p], t);
            }
        }
    };

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
                const f2 ba0 = pk_fma(splat(sa.z), dz2[0], pk_fma(splat(sa.y), dy2[0], splat(sa.x) * dx2[0]));
                const f2 ba1 = pk_fma(splat(sa.z), dz2[1], pk_fma(splat(sa.y), dy2[1], splat(sa.x) * dx2[1]));
                const f2 bb0 = pk_fma(splat(sb.z), dz2[0], pk_fma(splat(sb.y), dy2[0], splat(sb.x) * dx2[0]));
                const f2 bb1 = pk_fma(splat(sb.z), dz2[1], pk_fma(splat(sb.y), dy2[1], splat(sb.x) * dx2[1]));
                const f2 da0 = pk_fma(ba0, ba0, splat(sa.w)), da1 = pk_fma(ba1, ba1, splat(sa.w));
                const f2 db0 = pk_fma(bb0, bb0, splat(sb.w)), db1 = pk_fma(bb1, bb1, splat(sb.w));
                m0 = f2{max3(m0.x, da0.x, db0.x), max3(m0.y, da0.y, db0.y)};
                m1 = f2{max3(m1.x, da1.x, db1.x), max3(m1.y, da1.y, db1.y)};
            }
            if (ballot(fmaxf(max3(m0.x, m0.y, m1.x), m1.y) >= 0.0f) == 0ull) continue;
            // some lane may hit one of these 4 spheres: redo them one at a time
#pragma unroll 1
            for (int u = 0; u < 4; u++) {
                const int i = c0 + j + u;
                if (i >= n) break;
                const f4 s = tile[j + u];
                unsigned long long pm[PPL];
                unsigned long long any = 0ull;
#pragma unroll
                for (int p = 0; p < PPL; p++) {
                    const float b = fmaf(s.z, dzs[p], fmaf(s.y, dys[p], s.x * dxs[p]));
                    pm[p] = ballot(fmaf(b, b, s.w) >= 0.0f) & live_m[p];
                    any |= pm[p];
                }
                if (any != 0ull) {
                    if constexpr (DBG) dbg_l2_primary++;
                    primary_l2(i, pm);
                }
            }
        }
    }

    // miss: the background, bit-exact by construction (RayTracer.js:43-46)
    const float bgr = hdr->bg32[0], bgg = hdr->bg32[1], bgb = hdr->bg32[2];
    const uint32_t bg_packed = hdr->bg_packed;
    float cr[PPL], cg[PPL], cb[PPL];
    uint32_t packed[PPL];
    bool valid[PPL];
    unsigned long long valid_m[PPL];
    // hit frame
    F3 N[PPL];
    float ts[PPL], eP[PPL], eN[PPL];
    float ar[PPL], ag[PPL], ab[PPL], wsum[PPL], eLmax[PPL];
    // Pixels whose nearest element is certain but whose shading is not (a shadow verdict or the 8-bit
    // rounding inside its bound) are queued for a float64 re-shade only: it takes the element and, per
    // light, 2 bits: 0 lit or facing away for certain, 1 blocked for certain, 2 look for yourself.
    // Pixels with a "look for yourself" light go to the full re-trace instead: with this many spheres one
    // occluder loop in a wave of otherwise short entries would hold the whole wave.
    Unsure<DBG> reshade[PPL];
    uint32_t codes[PPL];
    bool loops[PPL];
#pragma unroll
    for (int p = 0; p < PPL; p++) {
        codes[p] = 0u; loops[p] = false;
        cr[p] = bgr; cg[p] = bgg; cb[p] = bgb; packed[p] = bg_packed;
        valid[p] = (idx[p] >= 0) & live[p];
        valid_m[p] = ballot(valid[p]);
        // nearest-vs-second ordering, far clip (t < 1e8, RayTracer.js:19,36)
        unsure[p].add_mask(ballot((t2[p] - tnear[p] <= 2.0f * etmax[p]) & valid[p]), WHY_ORDER);
        unsure[p].add_mask(ballot((tnear[p] >= FAR_UNSURE) & valid[p]), WHY_FAR);
        // the winner once more, to get its hit frame (same arithmetic as level 2 above)
        const int iw = valid[p] ? idx[p] : 0;
        const float4 g0 = *reinterpret_cast<const float4*>(&gp32[iw].lx);
        const float2 g1 = *reinterpret_cast<const float2*>(&gp32[iw].r2);
        const float eb_w = gp32[iw].eb;
        const float2 g3 = *reinterpret_cast<const float2*>(&gp32[iw].inv_r);
        const F3 lw{g0.x, g0.y, g0.z}, d{dxs[p], dys[p], dzs[p]};
        const Disc qw = best_disc(g0.x, g0.y, g0.z, g0.w, g1.x, g3.y, eb_w, d);
        const float dm = fmaxf(qw.disc, 1e-30f);
        const float s_w = rsq(dm);
        const float bt = fmaf(dm, s_w, qw.b);
        const float et_w = fmaf(0.5f * qw.err, s_w, eb_w);
        const float rbt = rcp(bt);
        const bool outside = g0.w > 0.0f;
        ts[p] = outside ? g0.w * rbt : bt;
        eP[p] = outside ? ts[p] * fmaf(et_w, rbt, 4.0f * EPS) : fmaf(2.0f * EPS, ts[p], et_w);
        const F3 P{d.x * ts[p], d.y * ts[p], d.z * ts[p]};
        const F3 Nn{P.x - lw.x, P.y - lw.y, P.z - lw.z};                   // P - C (Sphere.js:47-51)
        const float inv = rsq(dot3(Nn, Nn));
        const float sg = outside ? inv : -inv;                              // RayTracer.js:54-58
        N[p] = F3{Nn.x * sg, Nn.y * sg, Nn.z * sg};
        eN[p] = fmaf(eP[p] + EPS * (g1.y + ts[p]), g3.x, 3.0f * EPS);
        ar[p] = 0.0f; ag[p] = 0.0f; ab[p] = 0.0f; wsum[p] = 0.0f; eLmax[p] = 0.0f;
    }
    const bool wave_valid = (valid_m[0] | valid_m[1] | valid_m[2] | valid_m[3]) != 0ull;

    // ---- direct light + binary shadow (RayTracer.js:61-91) -------------------------------------
    const int nl = hdr->n_lights;
    const bool hintable = nl <= 16;                       // light verdicts fit the queue's info word
    for (int k = 0; k < nl; k++) {
        const int il = clights[k];
        const int prow = k < PTR_MAX_PAIR_LIGHTS ? k : (n < PTR_MAX_PAIR_LIGHTS ? n : PTR_MAX_PAIR_LIGHTS);
        const PairQ* pairs = gpairs + (size_t)prow * (size_t)(n + 2);
        const f4 lg0 = rec0[4 * (size_t)il], lg1 = rec0[4 * (size_t)il + 1];
        const float ler = cm32[il].er, leg = cm32[il].eg, leb = cm32[il].eb, lemax = cm32[il].emax;
        F3 L[PPL];
        float ratio[PPL], eL[PPL], cL[PPL], k2s[PPL];
        unsigned long long open_m[PPL], maybe_m[PPL], blocked_m[PPL], init_m[PPL], vague_m[PPL];
#pragma unroll
        for (int p = 0; p < PPL; p++) {
            const F3 Lv{fmaf(-dxs[p], ts[p], lg0.x), fmaf(-dys[p], ts[p], lg0.y), fmaf(-dzs[p], ts[p], lg0.z)};
            const float invl = rsq(dot3(Lv, Lv));
            L[p] = F3{Lv.x * invl, Lv.y * invl, Lv.z * invl};
            eL[p] = fmaf(fmaf(EPS, lg1.w + ts[p], eP[p] + 2.0f * EPS * ts[p]), invl, 3.0f * EPS);   // |error of L|
            ratio[p] = dot3(N[p], L[p]);
            cL[p] = eL[p] + 3.0f * EPS;
            const float k2 = fmaf(2.0f, cL[p], 8.0f * EPS);
            // A direction this vague (a grazing hit on a huge sphere) would pass the cone test of every
            // occluder: hybrid mode hands the light to the float64 kernel at once, float32-only mode tests
            // with the widest slack level 1 works with (tolerance mode: no longer one-sided there).
            const bool vague = HYBRID & (k2 > K2_MAX);
            const bool open = (ratio[p] > 0.0f) & valid[p] & !vague;
            open_m[p] = ballot(open);
            init_m[p] = open_m[p];
            // not tested, but the float64 kernel may find N.L > 0: it must then look for occluders itself
            vague_m[p] = ballot(valid[p] & (vague ? ratio[p] > 0.0f : (ratio[p] <= 0.0f) & (ratio[p] >= -(eN[p] + eL[p] + 8.0f * EPS))));
            reshade[p].add_mask(ballot(vague & (ratio[p] > 0.0f) & valid[p]), WHY_SHADOW);
            maybe_m[p] = 0ull; blocked_m[p] = 0ull;
            // (a.L)^2 >= thr^2 - 2 cL (+ rounding of thr^2 and of the dot) <=> level 1 passes; closed lanes never do
            k2s[p] = open ? fminf(k2, K2_MAX) : NEVER;
        }
        bool wave_open = wave_valid && (open_m[0] | open_m[1] | open_m[2] | open_m[3]) != 0ull;

        // level 2 for occluder j and the lanes nm[p] that passed the cone test
        auto shadow_l2 = [&](int j, const unsigned long long (&nm)[PPL]) {
            const ConstF4Q rec = rec0 + 4 * (size_t)j;
            const f4 g0 = rec[0], g1 = rec[1], g3 = rec[3];
            const float rj = g1.z * g3.x;                                    // r = r^2 / r
#pragma unroll
            for (int p = 0; p < PPL; p++) {
                if (nm[p] == 0ull) continue;                                 // wave-uniform
                const float t = ts[p];
                const F3 so{fmaf(N[p].x, (float)PTR_BIAS, dxs[p] * t), fmaf(N[p].y, (float)PTR_BIAS, dys[p] * t),
                            fmaf(N[p].z, (float)PTR_BIAS, dzs[p] * t)};
                const F3 lj{g0.x - so.x, g0.y - so.y, g0.z - so.z};
                const F3 Lp = L[p];
                const float bj = dot3(lj, Lp);
                const F3 c3{fmaf(lj.y, Lp.z, -lj.z * Lp.y), fmaf(lj.z, Lp.x, -lj.x * Lp.z), fmaf(lj.x, Lp.y, -lj.y * Lp.x)};
                const float dp2 = dot3(c3, c3);
                const float disc = g1.z - dp2;
                // |lj| <= A; error of lj <= dlj4 / 4 (generously); error of L <= eL:
                //   the distance |lj x L| of the centre from the ray is off by at most ec, bj by at most ec
                const float A = g1.w + t;
                const float dlj4 = fmaf(4.0f * EPS, A, 4.0f * eP[p]);
                const float ec = fmaf(A, cL[p], dlj4);
                const float band = fmaf(ec, fmaf(2.0001f, rj, ec), 8.0f * EPS * (dp2 + g1.z));
                const unsigned long long certain = ballot(bj > ec) & ballot(disc > band) & nm[p];
                const unsigned long long possible = ballot(bj >= -ec) & ballot(disc >= -band) & nm[p];
                maybe_m[p] |= possible & ~certain;
                if constexpr (!HYBRID) blocked_m[p] |= ballot(bj >= 0.0f) & ballot(disc >= 0.0f) & nm[p];
                open_m[p] &= ~certain;
                k2s[p] = lane_of(certain) ? NEVER : k2s[p];
            }
        };

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
            const f2 Lx2[2] = {f2{L[0].x, L[1].x}, f2{L[2].x, L[3].x}};
            const f2 Ly2[2] = {f2{L[0].y, L[1].y}, f2{L[2].y, L[3].y}};
            const f2 Lz2[2] = {f2{L[0].z, L[1].z}, f2{L[2].z, L[3].z}};
            for (int j = 0; j < cnt4 && wave_open; j += 4) {
                const f2 k20{k2s[0], k2s[1]}, k21{k2s[2], k2s[3]};
                f2 m0 = splat(NEVER), m1 = splat(NEVER);
#pragma unroll
                for (int u = 0; u < 4; u += 2) {
                    const f4 sa = tile[j + u], sb = tile[j + u + 1];
                    const f2 ca0 = pk_fma(splat(sa.z), Lz2[0], pk_fma(splat(sa.y), Ly2[0], splat(sa.x) * Lx2[0]));
                    const f2 ca1 = pk_fma(splat(sa.z), Lz2[1], pk_fma(splat(sa.y), Ly2[1], splat(sa.x) * Lx2[1]));
                    const f2 cb0 = pk_fma(splat(sb.z), Lz2[0], pk_fma(splat(sb.y), Ly2[0], splat(sb.x) * Lx2[0]));
                    const f2 cb1 = pk_fma(splat(sb.z), Lz2[1], pk_fma(splat(sb.y), Ly2[1], splat(sb.x) * Lx2[1]));
                    const f2 ua0 = pk_fma(ca0, ca0, k20) + splat(sa.w), ua1 = pk_fma(ca1, ca1, k21) + splat(sa.w);
                    const f2 ub0 = pk_fma(cb0, cb0, k20) + splat(sb.w), ub1 = pk_fma(cb1, cb1, k21) + splat(sb.w);
                    m0 = f2{max3(m0.x, ua0.x, ub0.x), max3(m0.y, ua0.y, ub0.y)};
                    m1 = f2{max3(m1.x, ua1.x, ub1.x), max3(m1.y, ua1.y, ub1.y)};
                }
                if (ballot(fmaxf(max3(m0.x, m0.y, m1.x), m1.y) >= 0.0f) == 0ull) continue;
#pragma unroll 1
                for (int u = 0; u < 4; u++) {
                    const int jj = c0 + j + u;
                    if (jj >= n) break;
                    if (jj == il) continue;                                   // RayTracer.js:76
                    const f4 s = tile[j + u];
                    unsigned long long nm[PPL];
                    unsigned long long any = 0ull;
#pragma unroll
                    for (int p = 0; p < PPL; p++) {
                        const float c = fmaf(s.z, L[p].z, fmaf(s.y, L[p].y, s.x * L[p].x));
                        nm[p] = ballot(fmaf(c, c, k2s[p]) + s.w >= 0.0f) & open_m[p];
                        any |= nm[p];
                    }
                    if (any != 0ull) {
                        if constexpr (DBG) dbg_l2_shadow++;
                        shadow_l2(jj, nm);
                        wave_open = (open_m[0] | open_m[1] | open_m[2] | open_m[3]) != 0ull;
                    }
                }
            }
        }

#pragma unroll
        for (int p = 0; p < PPL; p++) {
            // a "maybe" only matters if no certain hit blocks the light anyway
            const unsigned long long unsure_light = open_m[p] & maybe_m[p];
            reshade[p].add_mask(unsure_light, WHY_SHADOW);
            if (k < 16) {
                const uint32_t code = lane_of(unsure_light | vague_m[p]) ? 2u : (lane_of(init_m[p] & ~open_m[p]) ? 1u : 0u);
                codes[p] |= code << (2 * k);
                loops[p] |= code == 2u;
            }
            if (lane_of(open_m[p] & ~blocked_m[p])) {
                const SphereM32& mw = gm32[idx[p]];
                ar[p] = fmaf(mw.sr * ler, ratio[p], ar[p]);
                ag[p] = fmaf(mw.sg * leg, ratio[p], ag[p]);
                ab[p] = fmaf(mw.sb * leb, ratio[p], ab[p]);
                wsum[p] += lemax;
                eLmax[p] = fmaxf(eLmax[p], eL[p]);
            }
        }
    }

    // ---- emission, clamp, round, pack (RayTracer.js:93,179-192) with a rounding bound -----------
#pragma unroll
    for (int p = 0; p < PPL; p++) {
        const SphereM32 mw = gm32[valid[p] ? idx[p] : 0];
        const float hr = ar[p] + mw.er, hg = ag[p] + mw.eg, hb = ab[p] + mw.eb;
        const float band = 255.0f * fmaf(mw.scmax * wsum[p], eN[p] + eLmax[p] + 8.0f * EPS, 4.0f * EPS * mw.emax) + 2e-5f;
        const float vr = fminf(hr, 1.0f) * 255.0f, vg = fminf(hg, 1.0f) * 255.0f, vb = fminf(hb, 1.0f) * 255.0f;
        const float fr = floorf(vr), fg = floorf(vg), fb = floorf(vb);
        const float qr = vr - fr, qg = vg - fg, qb = vb - fb;
        const float nearest = fminf(fminf(fabsf(qr - 0.5f), fabsf(qg - 0.5f)), fabsf(qb - 0.5f));
        reshade[p].add_mask(ballot((nearest <= band) & valid[p]), WHY_ROUND);
        const uint32_t r8 = (uint32_t)(qr >= 0.5f ? fr + 1.0f : fr);
        const uint32_t g8 = (uint32_t)(qg >= 0.5f ? fg + 1.0f : fg);
        const uint32_t b8 = (uint32_t)(qb >= 0.5f ? fb + 1.0f : fb);
        if (valid[p]) {
            packed[p] = (255u << 24) | (b8 << 16) | (g8 << 8) | r8;
            cr[p] = hr; cg[p] = hg; cb[p] = hb;
        }
    }

    // ---- stores ---------------------------------------------------------------------------------
    const uint32_t bid = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
#pragma unroll
    for (int p = 0; p < PPL; p++) {
        if (!live[p]) continue;
        if constexpr (DBG)
            packed[p] = unsure[p].why() | reshade[p].why() | ((dbg_l2_primary < 4095u ? dbg_l2_primary : 4095u) << 8) |
                        ((dbg_l2_shadow < 4095u ? dbg_l2_shadow : 4095u) << 20);
        long long idx_out = (long long)a.width * px_row[p] + px_x[p];
        if (quirk) idx_out -= a.width;
        if (idx_out >= 0) out8[idx_out] = packed[p];
        if (a.out32f) {
            float4* o = reinterpret_cast<float4*>(a.out32f) + (frame * a.out_frame_stride + (uint64_t)a.width * px_row[p] + px_x[p]);
            *o = make_float4(cr[p], cg[p], cb[p], 1.0f);
        }
        if constexpr (HYBRID) {
            const uint32_t sub = bid & (a.n_queues - 1u);           // n_queues is a power of two
            const uint32_t gid = (uint32_t)(frame * frame_px + (uint64_t)a.width * px_row[p] + px_x[p]);
            if (unsure[p].any() | (reshade[p].any() & (!hintable | loops[p]))) {
                // kind 0: from the front of the sub-queue
                const uint32_t q = atomicAdd(a.counter + sub * PTR_COUNTER_STRIDE, 1u);
                if (q < a.queue_capacity) a.queue[(uint64_t)sub * a.queue_capacity + q] = gid;
            } else if (reshade[p].any()) {
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
    hipLaunchKernelGGL((render_many_kernel<HYBRID, DBG>), grid, dim3(PTR_BLOCK), 0, s, a);
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
