// ptr_kernels_exact.hip — float64 kernels that follow the reference operation for operation.
//
// Compiled with -ffp-contract=off: every '*' and '+' below is one IEEE-754 double rounding in
// the same left-to-right order as the JavaScript source, f64 sqrt and divide are the correctly
// rounded device forms, so results are bit-identical to the reference
// (RT/src/RayTracer.js:18-95,132-197, RT/src/Sphere.js:15-51, RT/src/Vector3.js:23-37).
//
// Used for (1) PTR_MODE_EXACT_F64 whole-band renders, (2) the hybrid mode's re-trace of
// queued pixels, (3) ptr_trace_batch_f64.
#include <hip/hip_runtime.h>
#include "ptr_device.h"
#include "ptr_kernels.h"

namespace ptr {

struct V3 { double x, y, z; };

// Vector3.normalize (Vector3.js:23-33): invLen = 1/sqrt(len2), then three multiplies
__device__ __forceinline__ void normalize_js(V3& v)
{
    double len2 = v.x * v.x + v.y * v.y + v.z * v.z;
    if (len2 > 0) {
        double inv = 1.0 / __builtin_sqrt(len2);
        v.x *= inv; v.y *= inv; v.z *= inv;
    }
}

struct ExactHit { int element; double tnear; V3 P, N; };

// stats word indices (match ptr_stats in include/ptr.h)
enum { ST_PRIMARY = 0, ST_HITS, ST_SHADOW, ST_EXIT_TCA, ST_EXIT_D2, ST_EXIT_HIT, ST_RETRACED, ST_RESHADED };

// Per-lane event counters (registers); a workgroup adds them up in LDS and issues one global
// atomic per counter (per-lane global atomics on six words took 186 ms for one 1080p frame).
struct Counts { unsigned c[6] = {0, 0, 0, 0, 0, 0}; };

template <bool STATS>
__device__ __forceinline__ void count(Counts* cnt, int which)
{
    if constexpr (STATS) cnt->c[which]++;
}

// The part of RayTracer.trace after the nearest-hit search (RayTracer.js:48-93): hit frame, every
// emissive element with its binary shadow, emission.  HINT: `codes` carries a verdict the float32 kernel
// is certain of for each of the first 16 lights (2 bits: 0 lit or facing away, 1 blocked, 2 not certain);
// only "not certain" lights run their occluder loop.
template <bool STATS, bool HINT>
__device__ __forceinline__ V3 shade_exact(const FrameView& fv, V3 o, V3 d, int element, double tnear,
                                          uint32_t codes, ExactHit* hit, Counts* stats)
{
    const int n = fv.h->n;
    // hit frame (RayTracer.js:48-58, Sphere.js:47-51)
    const SphereG64 se = fv.g64[element];              // per-lane gather
    const SphereM64 me = fv.m64[element];
    V3 P{o.x + d.x * tnear, o.y + d.y * tnear, o.z + d.z * tnear};
    V3 N{P.x - se.cx, P.y - se.cy, P.z - se.cz};
    normalize_js(N);
    if (d.x * N.x + d.y * N.y + d.z * N.z > 0) { N.x = -N.x; N.y = -N.y; N.z = -N.z; }
    if (hit) { hit->P = P; hit->N = N; }

    // shadow-ray origin (RayTracer.js:77, recomputed per j there; same value)
    const V3 so{P.x + N.x * PTR_BIAS, P.y + N.y * PTR_BIAS, P.z + N.z * PTR_BIAS};

    // every emissive element, scene order (RayTracer.js:61-91)
    V3 acc{0, 0, 0};
    const int nl = fv.h->n_lights;
    for (int k = 0; k < nl; k++) {
        const int i = fv.lights[k];
        const SphereG64 li = fv.g64[i];
        const SphereM64 lm = fv.m64[i];
        count<STATS>(stats, ST_SHADOW);
        V3 L{li.cx - P.x, li.cy - P.y, li.cz - P.z};
        normalize_js(L);
        double tr = 1.0;
        const uint32_t code = !HINT ? 2u : (k < 16 ? (codes >> (2 * k)) & 3u : 2u);
        if (code == 1u) tr = 0.0;
        if (code >= 2u) {
            for (int j = 0; j < n; j++) {
                if (j == i) continue;
                const SphereG64 s = fv.g64[j];
                double lx = s.cx - so.x, ly = s.cy - so.y, lz = s.cz - so.z;
                double tca = lx * L.x + ly * L.y + lz * L.z;
                if (tca < 0) { count<STATS>(stats, ST_EXIT_TCA); continue; }
                double d2 = (lx * lx + ly * ly + lz * lz) - tca * tca;
                if (d2 > s.r2) { count<STATS>(stats, ST_EXIT_D2); continue; }
                count<STATS>(stats, ST_EXIT_HIT);
                tr = 0.0;                                  // any hit blocks; distance ignored (F7)
                break;
            }
        }
        double ratio = js_max0(N.x * L.x + N.y * L.y + N.z * L.z);
        acc.x += (me.sr * tr) * (lm.er * ratio);       // RayTracer.js:89
        acc.y += (me.sg * tr) * (lm.eg * ratio);
        acc.z += (me.sb * tr) * (lm.eb * ratio);
    }
    acc.x += me.er; acc.y += me.eg; acc.z += me.eb;    // RayTracer.js:93
    return acc;
}

// RayTracer.trace (RayTracer.js:18-95).  o, d: ray; returns unclamped colour.
template <bool STATS>
__device__ V3 trace_exact(const FrameView& fv, V3 o, V3 d, ExactHit* hit, Counts* stats)
{
    const int n = fv.h->n;
    count<STATS>(stats, ST_PRIMARY);

    // nearest hit in scene order, strict '<' (RayTracer.js:26-41, Sphere.js:15-33)
    double tnear = PTR_INFINITY;
    int element = -1;
    for (int i = 0; i < n; i++) {
        const SphereG64 s = fv.g64[i];                 // wave-uniform address -> scalar load
        double lx = s.cx - o.x, ly = s.cy - o.y, lz = s.cz - o.z;
        double tca = lx * d.x + ly * d.y + lz * d.z;
        if (tca < 0) { count<STATS>(stats, ST_EXIT_TCA); continue; }
        double d2 = (lx * lx + ly * ly + lz * lz) - tca * tca;
        if (d2 > s.r2) { count<STATS>(stats, ST_EXIT_D2); continue; }
        count<STATS>(stats, ST_EXIT_HIT);
        double thc = __builtin_sqrt(s.r2 - d2);
        double t0 = tca - thc, t1 = tca + thc;
        if (t0 < 0) t0 = t1;
        if (t0 < tnear) { tnear = t0; element = i; }
    }
    if (hit) { hit->element = element; hit->tnear = tnear; }
    if (element < 0) {                                 // RayTracer.js:43-46
        if (hit) { hit->P = V3{0, 0, 0}; hit->N = V3{0, 0, 0}; }
        return V3{fv.h->bg64[0], fv.h->bg64[1], fv.h->bg64[2]};
    }
    count<STATS>(stats, ST_HITS);
    return shade_exact<STATS, false>(fv, o, d, element, tnear, 0u, hit, stats);
}

// RayTracer.trace for a ray whose nearest element the float32 kernel is certain of: that element's
// Sphere.intersect alone gives tnear (same operations, so the same bits as the full search).
__device__ V3 reshade_exact(const FrameView& fv, V3 o, V3 d, int element, uint32_t codes)
{
    const SphereG64 s = fv.g64[element];
    double lx = s.cx - o.x, ly = s.cy - o.y, lz = s.cz - o.z;
    double tca = lx * d.x + ly * d.y + lz * d.z;
    double d2 = (lx * lx + ly * ly + lz * lz) - tca * tca;
    double thc = __builtin_sqrt(s.r2 - d2);
    double t0 = tca - thc, t1 = tca + thc;
    if (t0 < 0) t0 = t1;
    return shade_exact<false, true>(fv, o, d, element, t0, codes, nullptr, nullptr);
}

// camera ray for pixel (x, y) of the full image (RayTracer.js:171-174)
__device__ __forceinline__ V3 primary_dir_exact(const CameraArgs& c, int x, int y)
{
    double xx = (2 * ((x + 0.5) * c.inv_w) - 1) * c.angle * c.aspect;
    double yy = (1 - 2 * ((y + 0.5) * c.inv_h)) * c.angle;
    V3 d;
    if (c.general) {
        d.x = xx * c.right[0] + yy * c.up[0] + c.fwd[0];
        d.y = xx * c.right[1] + yy * c.up[1] + c.fwd[1];
        d.z = xx * c.right[2] + yy * c.up[2] + c.fwd[2];
    } else {
        d = V3{xx, yy, -1.0};
    }
    normalize_js(d);
    return d;
}

// where pixel (x, row) of the band is stored; -1 = dropped (RayTracer.js:169 under the quirk)
__device__ __forceinline__ long long store_index(const RenderArgs& a, int x, int row)
{
    if (a.flags & 0x0001u /* PTR_QUIRK_ROW_SHIFT */)
        return (long long)a.width * row - (a.width - x);
    return (long long)a.width * row + x;
}

template <bool STATS>
__device__ __forceinline__ void shade_and_store_exact(const RenderArgs& a, int frame, int x, int row, Counts* cnt)
{
    const unsigned char* block = a.scenes + (uint64_t)(a.first_frame + frame) * a.frame_stride;
    const FrameView fv = frame_view(block);
    const long long idx = store_index(a, x, row);
    const bool quirk = (a.flags & 0x0001u) != 0;
    if (quirk && row == 0) {
        // the reference drops this row and never writes the band's last stored row
        a.out8[frame * a.out_frame_stride + (uint64_t)a.width * (a.scan_height - 1) + x] = 0u;
        if (!a.out32f && !STATS) return;
    }
    V3 o{a.cam.org[0], a.cam.org[1], a.cam.org[2]};
    V3 d = primary_dir_exact(a.cam, x, global_y(a, row));
    V3 c = trace_exact<STATS>(fv, o, d, nullptr, cnt);
    if (idx >= 0) a.out8[frame * a.out_frame_stride + idx] = pack_js(c.x, c.y, c.z);
    if (a.out32f) {
        float* p = a.out32f + 4 * (frame * a.out_frame_stride + (uint64_t)a.width * row + x);
        p[0] = (float)c.x; p[1] = (float)c.y; p[2] = (float)c.z; p[3] = 1.0f;
    }
}

// one pixel per lane, 8x8 pixels per wave, 32x8 per workgroup, blockIdx.z = frame
template <bool STATS>
__global__ void __launch_bounds__(PTR_BLOCK) render_exact_kernel(RenderArgs a)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x = blockIdx.x * PTR_TILE_W + wave * 8 + (lane & 7);
    const int row = blockIdx.y * PTR_TILE_H + (lane >> 3);
    Counts cnt;
    if (x < a.width && row < a.scan_height) shade_and_store_exact<STATS>(a, blockIdx.z, x, row, &cnt);
    if constexpr (STATS) {
        __shared__ unsigned block_cnt[6];
        if (threadIdx.x < 6) block_cnt[threadIdx.x] = 0u;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 6; k++)
            if (cnt.c[k]) atomicAdd(&block_cnt[k], cnt.c[k]);
        __syncthreads();
        if (threadIdx.x < 6 && block_cnt[threadIdx.x])
            atomicAdd(&a.stats[threadIdx.x], (unsigned long long)block_cnt[threadIdx.x]);
    }
}

// RayTracer.trace by a whole wave for ONE ray (scenes with many spheres): lane l tests elements l, l + 64, ...
// Every test is the same float64 arithmetic as in trace_exact; the nearest hit is the minimum over lanes of
// (t, index) — what the reference's strict '<' scan in scene order picks — and a shadow ray is blocked when any
// lane finds a hit.  All lanes return the same colour.
__device__ V3 trace_exact_wide(const FrameView& fv, V3 o, V3 d)
{
    const int n = fv.h->n, lane = threadIdx.x & 63;
    double tnear = PTR_INFINITY;
    int element = 0x7fffffff;
    // WIDE_UNROLL records per lane and trip, all requested before the first is used: a wave spends its time waiting for
    // these loads (one record per trip: 1.0 ms for the 13 000 re-traced pixels of a 1080p frame of BASELINE config 5), and
    // nothing else hides them.  The tests run in element order as before (a lane's strict '<' scan sees increasing i).
    constexpr int WIDE_UNROLL = 4;
    for (int i0 = lane; i0 < n; i0 += 64 * WIDE_UNROLL) {
        SphereG64 rec[WIDE_UNROLL];
#pragma unroll
        for (int u = 0; u < WIDE_UNROLL; u++) { const int i = i0 + 64 * u; rec[u] = fv.g64[i < n ? i : n - 1]; }
#pragma unroll
        for (int u = 0; u < WIDE_UNROLL; u++) {
            const int i = i0 + 64 * u;
            if (i >= n) continue;
            const SphereG64 s = rec[u];
            double lx = s.cx - o.x, ly = s.cy - o.y, lz = s.cz - o.z;
            double tca = lx * d.x + ly * d.y + lz * d.z;
            if (tca < 0) continue;
            double d2 = (lx * lx + ly * ly + lz * lz) - tca * tca;
            if (d2 > s.r2) continue;
            double thc = __builtin_sqrt(s.r2 - d2);
            double t0 = tca - thc, t1 = tca + thc;
            if (t0 < 0) t0 = t1;
            if (t0 < tnear) { tnear = t0; element = i; }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double t_o = __shfl_xor(tnear, off, 64);
        const int e_o = __shfl_xor(element, off, 64);
        if (t_o < tnear || (t_o == tnear && e_o < element)) { tnear = t_o; element = e_o; }
    }
    if (element == 0x7fffffff) return V3{fv.h->bg64[0], fv.h->bg64[1], fv.h->bg64[2]};   // RayTracer.js:43-46

    const SphereG64 se = fv.g64[element];
    const SphereM64 me = fv.m64[element];
    V3 P{o.x + d.x * tnear, o.y + d.y * tnear, o.z + d.z * tnear};
    V3 N{P.x - se.cx, P.y - se.cy, P.z - se.cz};
    normalize_js(N);
    if (d.x * N.x + d.y * N.y + d.z * N.z > 0) { N.x = -N.x; N.y = -N.y; N.z = -N.z; }
    const V3 so{P.x + N.x * PTR_BIAS, P.y + N.y * PTR_BIAS, P.z + N.z * PTR_BIAS};
    V3 acc{0, 0, 0};
    const int nl = fv.h->n_lights;
    for (int k = 0; k < nl; k++) {
        const int i = fv.lights[k];
        const SphereG64 li = fv.g64[i];
        const SphereM64 lm = fv.m64[i];
        V3 L{li.cx - P.x, li.cy - P.y, li.cz - P.z};
        normalize_js(L);
        double tr = 1.0;
        for (int j0 = 0; j0 < n; j0 += 64 * WIDE_UNROLL) {   // wave-uniform trip count
            SphereG64 rec[WIDE_UNROLL];
#pragma unroll
            for (int u = 0; u < WIDE_UNROLL; u++) { const int j = j0 + 64 * u + lane; rec[u] = fv.g64[j < n ? j : n - 1]; }
            bool hit = false;                                 // (any hit blocks the light, whichever element it is: RayTracer.js:77-81)
#pragma unroll
            for (int u = 0; u < WIDE_UNROLL; u++) {
                const int j = j0 + 64 * u + lane;
                if (j < n && j != i) {
                    const SphereG64 s = rec[u];
                    double lx = s.cx - so.x, ly = s.cy - so.y, lz = s.cz - so.z;
                    double tca = lx * L.x + ly * L.y + lz * L.z;
                    if (!(tca < 0)) {
                        double d2 = (lx * lx + ly * ly + lz * lz) - tca * tca;
                        hit |= !(d2 > s.r2);
                    }
                }
            }
            if (__builtin_amdgcn_ballot_w64(hit) != 0ull) { tr = 0.0; break; }
        }
        double ratio = js_max0(N.x * L.x + N.y * L.y + N.z * L.z);
        acc.x += (me.sr * tr) * (lm.er * ratio);
        acc.y += (me.sg * tr) * (lm.eg * ratio);
        acc.z += (me.sb * tr) * (lm.eb * ratio);
    }
    acc.x += me.er; acc.y += me.eg; acc.z += me.eb;
    return acc;
}

// hybrid mode, second kernel: the pixels the float32 kernel was not sure about.
// One workgroup per (sub-queue, kind); the fill counts are read from device memory, so the host never
// synchronises between the two kernels.  kind 0 (blockIdx.y == 0): re-trace the pixel from scratch;
// kind 1: the nearest element is certain and so are some of the shadow verdicts (qinfo), only the
// shading is redone in float64 (entries stored from the back of the sub-queue).
// one kernel per kind: the shading-only one needs far fewer registers than the full re-trace; WIDE: the full re-trace of a scene
// of many spheres (a wave per pixel, several records in flight per lane: 121 VGPRs against 93), a kernel of its own so that
// the few-sphere one keeps its five waves per SIMD
template <int KIND, bool WIDE>
__global__ void __launch_bounds__(PTR_BLOCK) retrace_exact_kernel(RenderArgs a)
{
    // The grid always has PTR_MAX_QUEUES x (parts1 + parts0) workgroups: every counter of the OTHER set is zeroed for the
    // next call (which may use more sub-queues than this one); workgroups past n_queues do only that.
    // blockIdx.y < parts1: part blockIdx.y of the sub-queue's shading-only entries (kind 1); the others: a part of its
    // full re-traces (kind 0) - those take several times longer per entry, so they are spread over more workgroups.
    const uint32_t sub = blockIdx.x;
    constexpr uint32_t kind = (uint32_t)KIND;
    const uint32_t part = blockIdx.y;
    const uint32_t nparts = gridDim.y;
    // (the full re-trace always runs, the shading-only re-do only in the mode that has such entries: the first clears both of
    // the other set's words for the next call)
    if (threadIdx.x == 0 && part == 0 && KIND == 0) { a.counter_next[sub * PTR_COUNTER_STRIDE] = 0u; a.counter_next[sub * PTR_COUNTER_STRIDE + 1] = 0u; }
    if (sub >= a.n_queues) return;
    const uint32_t fill = a.counter[sub * PTR_COUNTER_STRIDE + kind];
    const uint32_t count = fill < a.queue_capacity ? fill : a.queue_capacity;
    const uint32_t* queue = a.queue + (uint64_t)sub * a.queue_capacity;
    const uint32_t* qinfo = a.qinfo + (a.qinfo_packed ? 1u : 2u) * (uint64_t)sub * a.queue_capacity;
    const uint64_t frame_px = (uint64_t)a.width * (uint64_t)a.scan_height;
    if constexpr (KIND == 0 && WIDE) {
        // many spheres: one wave per queued pixel (trace_exact_wide), waves of all the sub-queue's workgroups take turns
        const uint32_t waves = (blockDim.x >> 6) * nparts;
        for (uint32_t q = part * (blockDim.x >> 6) + (threadIdx.x >> 6); q < count; q += waves) {
            const uint64_t gid = queue[q];
            if (gid >= frame_px * (uint64_t)a.n_frames) continue;
            const int frame = (int)(gid / frame_px);
            const uint32_t pix = (uint32_t)(gid - (uint64_t)frame * frame_px);
            const int row = (int)(pix / (uint32_t)a.width), x = (int)(pix - (uint32_t)row * a.width);
            const FrameView fv = frame_view(a.scenes + (uint64_t)(a.first_frame + frame) * a.frame_stride);
            const V3 o{a.cam.org[0], a.cam.org[1], a.cam.org[2]};
            const V3 c = trace_exact_wide(fv, o, primary_dir_exact(a.cam, x, global_y(a, row)));
            if ((threadIdx.x & 63) == 0) {
                const long long idx = store_index(a, x, row);
                if (idx >= 0) a.out8[frame * a.out_frame_stride + idx] = pack_js(c.x, c.y, c.z);
                if (a.out32f) {
                    float* p = a.out32f + 4 * (frame * a.out_frame_stride + (uint64_t)a.width * row + x);
                    p[0] = (float)c.x; p[1] = (float)c.y; p[2] = (float)c.z; p[3] = 1.0f;
                }
            }
        }
        if (threadIdx.x == 0 && part == 0 && a.stats) atomicAdd(&a.stats[ST_RETRACED], (unsigned long long)count);
        return;
    }
    // nparts workgroups share a sub-queue's entries of this kind
    // (a call holds fewer than 2^32 pixels, checked by the host: ids and their divisions are 32-bit; the block's tables are
    // found from the layout the host passes, not from the header: one dependent load less on every entry's critical path)
    const uint32_t frame_px32 = (uint32_t)frame_px, total_px = frame_px32 * (uint32_t)a.n_frames;
    for (uint32_t q = part * blockDim.x + threadIdx.x; q < count; q += blockDim.x * nparts) {
        const uint32_t slot = kind ? a.queue_capacity - 1u - q : q;
        const uint32_t gid = queue[slot];
        if (gid >= total_px) continue;                             // never trust a queue word blindly
        const uint32_t frame = gid / frame_px32;
        const uint32_t pix = gid - frame * frame_px32;
        const int row = (int)(pix / (uint32_t)a.width), x = (int)(pix - (uint32_t)row * a.width);
        const unsigned char* block = a.scenes + (uint64_t)(a.first_frame + (int)frame) * a.frame_stride;
        FrameView fv;
        fv.h = reinterpret_cast<const FrameHeader*>(block);
        fv.g64 = reinterpret_cast<const SphereG64*>(block + sizeof(FrameHeader));
        fv.m64 = reinterpret_cast<const SphereM64*>(block + sizeof(FrameHeader) + sizeof(SphereG64) * (size_t)a.n);
        fv.p32 = reinterpret_cast<const SphereP32*>(block + a.off_p32);
        fv.m32 = reinterpret_cast<const SphereM32*>(block + a.off_m32);
        fv.lights = reinterpret_cast<const int32_t*>(block + a.off_lights);
        V3 o{a.cam.org[0], a.cam.org[1], a.cam.org[2]};
        V3 d = primary_dir_exact(a.cam, x, global_y(a, row));
        V3 c;
        if constexpr (KIND == 1) {
            uint32_t element, codes;
            if (a.qinfo_packed) { const uint32_t v = qinfo[slot]; element = v >> 24; codes = v & 0xffffffu; }
            else { element = qinfo[2 * slot]; codes = qinfo[2 * slot + 1]; }
            if (element >= (uint32_t)a.n) continue;
            c = reshade_exact(fv, o, d, (int)element, codes);
        } else {
            c = trace_exact<false>(fv, o, d, nullptr, nullptr);
        }
        const long long idx = store_index(a, x, row);
        if (idx >= 0) a.out8[frame * a.out_frame_stride + idx] = pack_js(c.x, c.y, c.z);
        if (a.out32f) {
            float* p = a.out32f + 4 * (frame * a.out_frame_stride + (uint64_t)a.width * row + x);
            p[0] = (float)c.x; p[1] = (float)c.y; p[2] = (float)c.z; p[3] = 1.0f;
        }
    }
    if (threadIdx.x == 0 && part == 0 && a.stats) atomicAdd(&a.stats[kind ? ST_RESHADED : ST_RETRACED], (unsigned long long)count);
}

__global__ void __launch_bounds__(PTR_BLOCK) trace_batch_exact_kernel(TraceArgs a)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= a.n) return;
    const FrameView fv = frame_view(a.scene);
    V3 o = a.origins ? V3{a.origins[3 * k], a.origins[3 * k + 1], a.origins[3 * k + 2]}
                     : V3{a.org[0], a.org[1], a.org[2]};
    V3 d{a.dirs[3 * k], a.dirs[3 * k + 1], a.dirs[3 * k + 2]};
    ExactHit h;
    V3 c = trace_exact<false>(fv, o, d, &h, nullptr);
    if (a.rc_mode) {
        // rendercache-js/src/Raytracer.js:62-76: a miss reports a point 1e5 along the ray and normalDir = -hit;
        // a hit reports the geometric normal BEFORE it is turned towards the ray, normalized a second time
        if (h.element < 0) {
            h.P = V3{o.x + d.x * 100000.0, o.y + d.y * 100000.0, o.z + d.z * 100000.0};
            h.N = V3{-h.P.x, -h.P.y, -h.P.z};
        } else {
            const SphereG64 se = fv.g64[h.element];
            V3 N{h.P.x - se.cx, h.P.y - se.cy, h.P.z - se.cz};
            normalize_js(N);
            normalize_js(N);
            h.N = N;
        }
    }
    a.out_rgb[3 * k] = c.x; a.out_rgb[3 * k + 1] = c.y; a.out_rgb[3 * k + 2] = c.z;
    if (a.out_element) a.out_element[k] = h.element;
    if (a.out_hit) { a.out_hit[3 * k] = h.P.x; a.out_hit[3 * k + 1] = h.P.y; a.out_hit[3 * k + 2] = h.P.z; }
    if (a.out_normal) { a.out_normal[3 * k] = h.N.x; a.out_normal[3 * k + 1] = h.N.y; a.out_normal[3 * k + 2] = h.N.z; }
}

// ---- launchers (called from ptr_api.hip) --------------------------------------------------
hipError_t launch_render_exact(const RenderArgs& a, int n_frames, bool stats, hipStream_t s)
{
    dim3 grid((a.width + PTR_TILE_W - 1) / PTR_TILE_W, (a.scan_height + PTR_TILE_H - 1) / PTR_TILE_H, n_frames);
    if (stats) hipLaunchKernelGGL(render_exact_kernel<true>, grid, dim3(PTR_BLOCK), 0, s, a);
    else       hipLaunchKernelGGL(render_exact_kernel<false>, grid, dim3(PTR_BLOCK), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_retrace_exact(const RenderArgs& a0, int parts1, int parts0, hipStream_t s)
{
    const RenderArgs& a = a0;
    if (a.n >= PTR_WIDE_RETRACE_MIN) hipLaunchKernelGGL((retrace_exact_kernel<0, true>), dim3(PTR_MAX_QUEUES, parts0, 1), dim3(PTR_BLOCK), 0, s, a);
    else hipLaunchKernelGGL((retrace_exact_kernel<0, false>), dim3(PTR_MAX_QUEUES, parts0, 1), dim3(PTR_BLOCK), 0, s, a);
    if (parts1 > 0) hipLaunchKernelGGL((retrace_exact_kernel<1, false>), dim3(PTR_MAX_QUEUES, parts1, 1), dim3(PTR_BLOCK), 0, s, a);   // (0: no shading-only entries in this mode)
    return hipGetLastError();
}

hipError_t launch_trace_batch_exact(const TraceArgs& a, hipStream_t s)
{
    hipLaunchKernelGGL(trace_batch_exact_kernel, dim3((a.n + PTR_BLOCK - 1) / PTR_BLOCK), dim3(PTR_BLOCK), 0, s, a);
    return hipGetLastError();
}

}  // namespace ptr
