// ptr_device.h — data layout shared by the host side of the C ABI and the gfx950 kernels.
//
// One "frame block" per scene of a sequence lives in HBM, built on the host by
// build_frame_block() (ptr_api.hip) and read by every workgroup through wave-uniform
// (scalar) loads: the sphere loops index with a wave-uniform counter, so the geometry sits
// in SGPRs and costs no vector memory traffic or LDS.  Per-lane reads happen only once per
// pixel, for the material of the winning sphere.
#pragma once
#include <stdint.h>

#define PTR_INFINITY 1e8      // RT/src/RayTracer.js:6 — finite far clip
#define PTR_BIAS     1e-4     // RT/src/RayTracer.js:52

#define PTR_TILE_W 32         // workgroup tile of the 256-thread kernels: 4 waves side by side, each an 8x8 pixel tile
#define PTR_TILE_H 8
#define PTR_BLOCK  256
#define PTR_FAST_BLOCK 64     // few-sphere kernel: one wave per workgroup (no LDS, no barrier: a wave then starts wherever a slot is free)
#ifndef PTR_FAST_PPL
#define PTR_FAST_PPL 3            // pixels per lane in the float32 kernel (wave tile 24 x 8: the slots side by side); measured 2 / 3 / 4
                                  //   pixels per lane at 7 / 6 / 4 waves per SIMD: 0.743 / 0.728 / 1.52 ms per 64 frames of 1080p
#endif
#ifndef PTR_FAST_WAVES
#define PTR_FAST_WAVES 6          // waves per SIMD the float32 kernel's register allocation aims at (80 VGPRs)
#endif
#ifndef PTR_FAST_PPL_TOL
#define PTR_FAST_PPL_TOL 4        // PTR_MODE_TOLERANT: 4 pixels per lane (wave tile 32 x 8) at 5 waves per SIMD (96 VGPRs): measured -3.4 %
#endif                            //   against 3 pixels at 6 waves (the per-wave work - culling, cones, scalar bookkeeping - is shared by more pixels)
#ifndef PTR_FAST_WAVES_WIDE
#define PTR_FAST_WAVES_WIDE 5
#endif
#define PTR_MAX_QUEUES 1024       // sub-queues of the hybrid mode's re-trace queue
#define PTR_MAX_PAIR_LIGHTS 64     // lights with a shadow pre-filter table (more lights: no culling for them)
#define PTR_FAST_MAX_SPHERES 12    // up to this many spheres: ptr_kernels_fast.hip; more: ptr_kernels_many.hip (measured
                                  //   crossover on MI355X: 7 spheres 0.35 vs 0.46 ms, 16 spheres 0.55 vs 0.40 ms per 16 frames)
#define PTR_MANY_THRESHOLD 128     // ptr_kernels_many.hip: more spheres than this are staged through LDS, fewer come by scalar loads
#define PTR_WIDE_RETRACE_MIN 256   // this many spheres or more: a whole wave re-traces one queued pixel
#define PTR_COUNTER_STRIDE 16     // uint32 words between two sub-queue counters (64 B)

namespace ptr {

// ---- per-frame scene block -------------------------------------------------------------
struct alignas(16) FrameHeader {          // 128 bytes
    int32_t  n;                           // elements
    int32_t  n_lights;                    // emissive elements, scene order
    uint32_t bg_packed;                   // pack_js(bg64): what a miss stores
    uint32_t needs_exact;                 // 1: scene has values the f32 path cannot represent
    double   bg64[3];
    float    bg32[3];
    float    tol_thr;                     // PTR_MODE_TOLERANT: a hit pixel whose summed error bounds of N.L (|error of N| + the sum
                                          //   over the lights of |error of L|) exceed this is re-traced; below it the float32
                                          //   colour is within 0.9 / 255 of the float64 one (host: build_frame_block)
    uint32_t off_g64, off_m64, off_p32, off_m32, off_lights;             // byte offsets in block
    uint32_t pair_always;                 // 1: every hit point ON A SPHERE of this scene is within sqrt(pair_smax2) of the camera (scenes without
                                          //   planes need no per-ray test)
    uint32_t self_skip;                   // 1: coordinates are small enough for the "own sphere never shadows" proof
    // planes and Phong terms (extension modes only, include/ptr.h "planes and Phong"); elements = spheres, then planes
    uint32_t n_planes;
    uint32_t has_phong;                   // 1: phong64 / phong32 hold ks[n + n_planes] then exponent[n + n_planes]
    uint32_t off_pl64, off_plm64, off_pl32, off_plm32, off_ph64, off_ph32;
    float    pair_smax2;                  // float64 bounce loop: the shadow pre-filter (PairQ) may only cull for a shadow ray whose origin lies
                                          //   within sqrt(this) of the camera: farther out the float64 test itself rounds by more than the
                                          //   margin the pre-filter leaves, and the kernel has to reproduce that rounding (< 0: never)
};
static_assert(sizeof(FrameHeader) == 128, "FrameHeader layout");

struct alignas(16) SphereG64 { double cx, cy, cz, r2; };            // exact geometry (Sphere.js:9-11)
struct alignas(16) SphereM64 { double sr, sg, sb, er, eg, eb, refl, transp; };   // exact material (Material.js:5-10)

// float32 path: everything is relative to the (wave-uniform) camera origin, computed in
// float64 on the host and rounded once, so a far-away camera costs no float32 precision.
// Read with wave-uniform indices -> scalar loads (s_load_dwordx8 / x4 per sphere); the
// winner's record is gathered per lane once per pixel.
// eps below is the float32 unit roundoff 2^-24; "scale" = max(|l|^2, r^2).
struct alignas(64) SphereP32 {
    // group 0+1 (one s_load_dwordx8): primary hit test and shadow test
    float lx, ly, lz;     // l = C - origin
    float cc;             // |l|^2 - r^2   (sign: origin outside (+) / inside (-) the sphere)
    float md;             // bound on the error of disc = b^2 - cc          (24 eps scale)
    float kb;             // md / eb, eb = bound on the error of b = l.d    (10 eps sqrt(scale))
    float r2;             // radius^2
    float lnorm;          // |l|
    // group 2: only when some lane of the wave hits the sphere
    float sgn;            // -1: outside, nearest root is b - thc;  +1: inside, root is b + thc
    float unsure;         // 1.0f: |cc| too small to trust its sign -> any hit on it is re-traced
    float sclo;           // lower bound of the b = l.d of any ray that can hit or be within the error bound of hitting:
                          //   origin outside: sqrt(cc) - eb, rounded down; inside: -eb  (one compare per ray and sphere)
    float eb;
    // group 3: winner refinement (per-lane gather)
    float inv_r;          // 1 / radius
    float hed;            // half the error bound of disc' = r^2 - |l x d|^2  (10 eps r |l| + 2 eps r^2); NEGATIVE (same magnitude)
                          //   when disc = b^2 - cc has the smaller bound for every ray (the origin close to a huge sphere)
    float refl, transp;   // Material.reflection / .transparency (extension modes only)
};
static_assert(sizeof(SphereP32) == 64, "SphereP32 layout");
// material, gathered per lane for the winning sphere; scmax/emax = largest channel
struct alignas(16) SphereM32 { float sr, sg, sb, scmax, er, eg, eb, emax; };

// planes (extension modes): point p, unit normal n; the float32 record is relative to the camera origin like SphereP32
struct alignas(16) PlaneG64 { double px, py, pz, nx, ny, nz; };
struct alignas(16) PlaneP32 { float px, py, pz, refl, nx, ny, nz, transp; };
static_assert(sizeof(PlaneG64) == 48 && sizeof(PlaneP32) == 32, "plane record layout");

// Shadow pre-filter, one record per (light k, element j), laid out [k][pair_stride(n)] (rows padded with
// never-cull records to a multiple of 4 records, so a row can be read 16 floats at a time):
// a = unit vector from the light's centre to element j's centre, thr = lower bound of |L.a| for
// which the line through the light with direction L can come within r_j (+ shadow bias) of C_j.
// A shadow ray towards light k can only hit element j if |L.a| + (error of L) >= thr.
struct alignas(16) PairQ { float ax, ay, az, thr; };
// What the few-sphere kernel needs of light k, at an address that does not depend on loaded data (the element record
// behind lights[k] costs two dependent scalar loads): [n_lights + 1], the last record is zero padding.
struct alignas(16) LightRec { float lx, ly, lz, lnorm; float er, eg, eb, emax; int32_t index; float eps_lnorm; int32_t pad[2]; };   // eps_lnorm = 2^-24 lnorm
static_assert(sizeof(LightRec) == 48, "LightRec layout");
// Tile culling of the few-sphere kernel, read by lane i for sphere i: a ray of a cone (axis dc, half-angle T) around the
// wave tile's centre ray can reach sphere i, or the error band of its float32 hit test, only if
//   l.dc >= cos(T) s - (sin(T) re + c),   s = sqrt(max(|l|^2 - r^2 - 2 md, 0)) rounded down (-3e38 when the origin is inside
// the enlarged sphere: always a candidate), re = sqrt(r^2 + 2 md) rounded up, c = 1e-5 |l| (rounding of all of this).
struct alignas(16) CullRec { float lx, ly, lz, s; float re, c, pad0, pad1; };
static_assert(sizeof(CullRec) == 32, "CullRec layout");
__host__ __device__ inline int pair_stride(int n) { return (n + 2 + 3) & ~3; }

struct FrameView {
    const FrameHeader* h;
    const SphereG64* g64;
    const SphereM64* m64;
    const SphereP32* p32;
    const SphereM32* m32;
    const int32_t*   lights;
    // planes / Phong (read through the header's offsets by the extension kernels only)
    __host__ __device__ const PlaneG64* pl64() const { return reinterpret_cast<const PlaneG64*>(reinterpret_cast<const unsigned char*>(h) + h->off_pl64); }
    __host__ __device__ const SphereM64* plm64() const { return reinterpret_cast<const SphereM64*>(reinterpret_cast<const unsigned char*>(h) + h->off_plm64); }
    __host__ __device__ const PlaneP32* pl32() const { return reinterpret_cast<const PlaneP32*>(reinterpret_cast<const unsigned char*>(h) + h->off_pl32); }
    __host__ __device__ const SphereM32* plm32() const { return reinterpret_cast<const SphereM32*>(reinterpret_cast<const unsigned char*>(h) + h->off_plm32); }
    __host__ __device__ const double* ph64() const { return reinterpret_cast<const double*>(reinterpret_cast<const unsigned char*>(h) + h->off_ph64); }
    __host__ __device__ const float* ph32() const { return reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(h) + h->off_ph32); }
};

__host__ __device__ inline FrameView frame_view(const unsigned char* block)
{
    FrameView v;
    v.h = reinterpret_cast<const FrameHeader*>(block);
    v.g64 = reinterpret_cast<const SphereG64*>(block + v.h->off_g64);
    v.m64 = reinterpret_cast<const SphereM64*>(block + v.h->off_m64);
    v.p32 = reinterpret_cast<const SphereP32*>(block + v.h->off_p32);
    v.m32 = reinterpret_cast<const SphereM32*>(block + v.h->off_m32);
    v.lights = reinterpret_cast<const int32_t*>(block + v.h->off_lights);
    return v;
}

// ---- kernel arguments (passed by value) --------------------------------------------------
struct CameraArgs {
    double org[3], right[3], up[3], fwd[3];
    double angle;          // tan(half fov)          RayTracer.js:154
    double aspect;         // width/height           RayTracer.js:153
    double inv_w, inv_h;   // 1/width, 1/height      RayTracer.js:150-151
    int32_t general;       // 0: reference camera, dir = (xx, yy, -1)
    int32_t pad;
};

// float32 camera: xx = (x+0.5)*sx + ox, yy = (y+0.5)*sy + oy  (RayTracer.js:171-172 refactored)
struct CameraF32 {
    float sx, ox, sy, oy;
    float right[3], up[3], fwd[3];
};

struct RenderArgs {
    const unsigned char* scenes;     // frame blocks, frame_stride bytes apart
    uint64_t  frame_stride;
    uint32_t  off_p32, off_m32, off_lights, off_pairs;   // byte offsets inside a frame block (same for every frame)
    uint32_t  off_lrec, off_cull, off_x4;
    int32_t   n;                     // elements per frame
    int32_t   first_frame, n_frames;
    int32_t   width, height, start_y, scan_height;
    uint32_t  flags;                 // PTR_QUIRK_ROW_SHIFT etc. (include/ptr.h)
    int32_t   strip_h, n_parts, part; // cyclic strips: local row r is image row global_y(r); n_parts 1 = band
    int32_t   rows_mode;             // few-sphere kernel: 0 = band (row = start_y + local row), 1 = strips of a multiple of 8 rows (a tile's
                                     //   rows are consecutive image rows), 2 = any strips (a division per lane)
    uint64_t  out_frame_stride;      // pixels between consecutive frames in out8 / out32f
    CameraArgs cam;
    CameraF32  camf;
    float     cull_cos, cull_sin;    // float32 kernel: every ray of a wave's (8 PPL) x 8 tile is within this angle of the
    int32_t   cull_on;               //   tile's centre ray (bound, padded); cull_on 0: tiles are not contiguous, no culling
    float     ext_cull_cos, ext_cull_sin;   // the same for the 8 x 8 tiles of the float32 bounce-loop kernel, jittered samples included
    int32_t   ext_cull_on;
    uint32_t* out8;                  // n_frames * width * scan_height
    float*    out32f;                // nullable, n_frames * width * scan_height * 4
    uint32_t* queue;                 // hybrid: n_queues sub-queues of global pixel ids: full re-traces fill a
                                     //   sub-queue from its front, shading-only entries from its back
    uint32_t* qinfo;                 // hybrid: per queue slot, nearest element and light verdict codes: one word (element << 24 | codes)
    uint32_t  qinfo_packed;          //   from the few-sphere kernel (qinfo_packed 1), two words from the many-sphere kernel
    uint32_t* counter;               // hybrid: sub-queue fills (this call's set), PTR_COUNTER_STRIDE apart;
                                     //   word 0 = full re-traces, word 1 = shading-only entries
    uint32_t* counter_next;          // hybrid: the other set, zeroed by the re-trace for the next call
    uint32_t  queue_capacity;        // entries per sub-queue
    uint32_t  n_queues;
    unsigned long long* stats;       // 8 words, see ptr_stats
    int32_t   max_depth, spp;        // extension modes (ptr_kernels_ext.hip)
    uint32_t  seed;
    // float32 bounce loop with more than PTR_EXT_SPP_BLOCK samples per pixel: a wave takes ONE block of samples of its tile and
    // leaves the block's sum here, [frame][block][row][x] as float4; ext_combine_kernel adds the blocks up in order.
    // NULL: every wave takes all samples of its pixels (same sums in the same order, in registers).
    float*    ext_partial;
    int32_t   ext_blocks;            // blocks per pixel = ceil(spp / PTR_EXT_SPP_BLOCK)
};

// The float32 forms of the extension modes add a pixel's samples up block by block: the sum of every PTR_EXT_SPP_BLOCK
// consecutive samples first, then the blocks' sums in order.  With at most that many samples it is the plain sum.  (The float64
// form is the specification: one running sum.)
// (Block size, A/B on one box with tools/config3_time.py, configs[3] on one GPU / configs[2]: 16 samples 15.33-15.38 / 0.992 ms,
// 32 samples 15.22-15.25 / 1.00 ms, 64 samples 15.73-15.75 / 1.05-1.06 ms.)
#ifndef PTR_EXT_SPP_BLOCK
#define PTR_EXT_SPP_BLOCK 32
#endif

struct TraceArgs {
    const unsigned char* scene;      // one frame block
    const double* origins;           // nullable -> cam origin
    const double* dirs;
    int32_t n;
    int32_t rc_mode;                 // 1: outputs follow rendercache-js (ptr_trace_samples_f64)
    double org[3];
    double* out_rgb;                 // 3 per ray
    int32_t* out_element;            // nullable
    double* out_hit;                 // nullable
    double* out_normal;              // nullable
};

// render cache reprojection pass (ptr_kernels_rc.hip); all pointers are device memory
struct RcArgs {
    double from[3], view[16], proj[16];      // camera.from, matrixWorldInverse, projectionMatrix (column-major)
    int32_t width, height, n_items;
    int32_t item_base;                       // a shard of a cache split over several GPUs: index of its first item in the whole cache
    double age_factor, depth_threshold;
    const double* hit_xyz;                   // 3 per item, NaN in x = no hit
    double* age;                             // in/out
    int32_t* item_pixel;                     // out: pixel (y * (width + 2) + x) the item was attached to, or -1
    int32_t* pixel_element;                  // out, (width + 2) * (height + 2), interior only
    double* pixel_depth;                     // out, same grid
    uint8_t* pixel_sample;                   // nullable: the frame loop's per-pixel sample flags, cleared with the interior
    // scratch owned by the context
    double* item_depth;
    uint32_t *sort_key, *sort_val, *sorted_key, *sorted_val;
};

// reprojection of a cache whose items are split over several GPUs in cache order: what every rank does after all ranks'
// local grids (nearest local item per pixel and its depth) have been gathered (ptr_kernels_rc.hip)
struct RcMergeArgs {
    int32_t width, height, n_items, item_base;   // this rank's items
    int32_t n_ranks, rank;
    double age_factor, depth_threshold;
    const double* all_depth;                 // n_ranks x grid: every rank's local pixel_depth
    const int32_t* all_element;              // n_ranks x grid: every rank's local pixel_element (indices in the whole cache)
    double* age; int32_t* item_pixel;        // this rank's items, as its local pass left them
    const double* item_depth;                // scratch of the local pass
    int32_t* pixel_element; double* pixel_depth;     // out: the grid of the whole cache
    double* before;                          // scratch, grid: nearest depth among the items of the ranks before this one
};

// render cache depth culling + gap filling + frames (ptr_kernels_rc.hip); all pointers are device memory
struct RcShadeArgs {
    int32_t width, height, n_items;
    double age_factor, interpolation_zero, interpolation_random, priority_max;
    const int32_t* pixel_element;            // (width + 2) * (height + 2)
    const double* pixel_depth;
    double* item_age;                        // in/out
    const uint32_t* item_rgb;                // r | g << 8 | b << 16 per item
    uint32_t* pixel_rgb;                     // out, grid
    int32_t* pixel_priority;                 // out, grid
    uint32_t* hist;                          // out, 3 x 256
    long long* totals;                       // out: pixels with an element, statistics.totalPriority, this.totalPriority increment
    uint32_t *color_frame, *reprojection_frame, *priority_frame;   // out, width * height each, nullable
    double* pixel_age;                       // scratch owned by the context, grid: the element's age after culling
};

// render cache frame loop on a cache that stays in HBM (ptr_kernels_rcloop.hip).  RcLoopWords is the small block of
// device words the passes of one frame hand to each other and the host reads back; it follows the 3 x 256 histogram
// bins and the 3 totals of the shading pass in one allocation (RC_WORDS_* offsets).
struct RcLoopWords {
    double  samples;                 // Driver.directSamples: priorities summed down to the threshold (Driver.js:690-706)
    int32_t threshold;               // -1 = null (no level exceeded the budget)
    int32_t thr_value;               // what `>= threshold` and `- threshold` see: null compares and subtracts as 0
    int32_t candidates;              // pixels at or above the threshold
    int32_t k_a, k_b;                // numbers of the picks whose diffusion reaches the walk's last pixel (INT_MAX: none)
    int32_t half[3];                 // diffused amount before k_a / up to k_b / after (Driver.js:829-839 with the stale pixel)
    int32_t n_need;                  // picks that had no cache item
    int32_t page_next;               // page the allocator looks at next (Driver.cachePointer / 8, wrapped)
    int32_t used;                    // cache items attached to a pixel (getCacheUsage)
    int32_t pad;
};
constexpr size_t RC_WORDS_HIST = 0, RC_WORDS_TOTALS = 3 * 256 * sizeof(uint32_t),
                 RC_WORDS_LOOP = RC_WORDS_TOTALS + 3 * sizeof(long long), RC_WORDS_BYTES = RC_WORDS_LOOP + sizeof(RcLoopWords);

// The picks of Driver.directSamples (:795-822) in closed form.  `index += requestFactor` in float64 moves by a constant
// number of units in the last place while the sum stays within one power of two, so the running sum is piecewise
// S0 + j * D (integers below 2^53) times 2^e: the host finds the pieces (one or two per power of two), a lane evaluates
// its own step.  Step k of piece `seg` is s = (S0 + (k - k0) * D) * 2^e, the pick is Math.round(s).
struct RcPickSegments {
    int32_t n, pad;
    int32_t k0[96];                  // first step of the piece
    int32_t e[96];
    long long S0[96], D[96];
};

struct RcLoopArgs {
    int32_t width, height, n_items, n_picks;
    int32_t pages, priority_max, rank_lo, rank_hi;
    double  age_factor, max_samples;
    double  from[3], world[16], proj_inv[16];      // camera.from, matrixWorld, projectionMatrixInverse (column-major)
    // the cache and the pixel grid
    int32_t* item_pixel; double* item_hit; double* item_age; uint32_t* item_rgb;
    int32_t* pixel_element; uint32_t* pixel_rgb; int32_t* pixel_priority; uint8_t* pixel_sample;
    const uint32_t* hist; RcLoopWords* words;
    // this frame's requests
    int32_t* cand;                   // candidate pixels in walk order
    int32_t* pick_index;             // which candidates are requested
    int32_t *pick_item, *need, *rank;
    double* dirs; const double* tr_rgb; const double* tr_hit;
    uint32_t* frames[4];             // colour, reprojection, priority, sampling (nullable), width * height each
};

// multi-GPU gather, copy step on the root (ptr_kernels_gather.hip); device pointers
struct GatherArgs {
    const uint32_t* src;             // one part's rows, n_frames frames, src_frame_stride_px apart
    uint32_t* full;                  // n_frames * height * width
    uint64_t src_frame_stride_px;
    int32_t width, height, n_frames;
    int32_t strip_h, n_parts, part, rows;   // rows = local rows of this part
};

// image row of local row `row` (band: start_y + row; cyclic strips: strip k belongs to part k % n_parts)
__host__ __device__ inline int global_y(const RenderArgs& a, int row)
{
    if (a.n_parts <= 1) return a.start_y + row;
    const int ls = row / a.strip_h;
    return a.start_y + (ls * a.n_parts + a.part) * a.strip_h + (row - ls * a.strip_h);
}

// ---- JS number semantics of the final pack (RayTracer.js:179-192) -------------------------
__host__ __device__ inline double js_min1(double c) { return (c != c) ? c : (c < 1.0 ? c : 1.0); }
__host__ __device__ inline double js_max0(double x) { return (x != x) ? x : (x > 0.0 ? x : 0.0); }

__host__ __device__ inline double js_round(double x)
{
    // Math.round: nearest integer, ties towards +infinity; NaN and +-Infinity pass through
    if (!(x - x == 0.0)) return x;
    double f = __builtin_floor(x);
    return (x - f >= 0.5) ? f + 1.0 : f;
}

__host__ __device__ inline uint32_t js_to_uint32(double x)
{
    // ECMAScript ToInt32/ToUint32 bit pattern
    if (!(x - x == 0.0)) return 0u;
    double t = __builtin_trunc(x);
    if (t >= -2147483648.0 && t < 4294967296.0)
        return t < 0 ? (uint32_t)(int32_t)t : (uint32_t)t;
    double m = __builtin_fmod(t, 4294967296.0);
    if (m < 0) m += 4294967296.0;
    return (uint32_t)m;
}

__host__ __device__ inline uint32_t pack_js(double r, double g, double b)
{
    double rr = js_round(js_min1(r) * 255.0);
    double gg = js_round(js_min1(g) * 255.0);
    double bb = js_round(js_min1(b) * 255.0);
    return (255u << 24) | (js_to_uint32(bb) << 16) | (js_to_uint32(gg) << 8) | js_to_uint32(rr);
}

}  // namespace ptr
