// ptr_api.hip — host side of the C ABI declared in include/ptr.h.
//
// Owns all device memory of a renderer context: the frame blocks (scene sequence flattened to
// the layouts of ptr_device.h), the re-trace queue and its two counter words, the stats words
// and staging buffers for the host-pointer entry points.  There is no CPU implementation of
// the path in here: without a GPU ptr_create fails.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <cmath>
#include <cstdarg>
#include <cstddef>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/ptr.h"
#include "ptr_device.h"
#include "ptr_kernels.h"

using namespace ptr;

namespace {

thread_local char g_create_error[512] = "";      // fixed buffers: reporting an error must never throw

struct DeviceBuffer {
    void*  ptr = nullptr;
    size_t bytes = 0;
};

// The render cache when it lives on the GPU (ptr_rc_cache_*, ptr_rc_next_frame): Driver.cache and Driver.buffer as flat
// arrays, the words and lists one frame's passes hand to each other, and two pinned host buffers for what crosses PCIe.
struct RcCache {
    int width = 0, height = 0, n_items = 0;
    bool loaded = false;
    DeviceBuffer item_pixel, item_hit, item_age, item_rgb;                                  // n_items
    DeviceBuffer pixel_element, pixel_depth, pixel_rgb, pixel_priority, pixel_sample;       // (width + 2) * (height + 2)
    DeviceBuffer words;                                                                     // RC_WORDS_*
    DeviceBuffer cand, picks, pick_item, need, rank, dirs, tr_rgb, tr_hit;                  // up to width * height requests
    DeviceBuffer frames, temp;
    unsigned char* host_words = nullptr;
    int32_t* host_picks = nullptr;
};

// layout of ptr_ctx::words (device): two queue counters on lines of their own, then the stats
constexpr size_t WORDS_COUNTERS  = (size_t)PTR_MAX_QUEUES * PTR_COUNTER_STRIDE * sizeof(uint32_t);
constexpr size_t WORDS_COUNTER_A = 0;
constexpr size_t WORDS_COUNTER_B = WORDS_COUNTERS;
constexpr size_t WORDS_STATS     = 2 * WORDS_COUNTERS;     // 8 x uint64
constexpr size_t WORDS_BYTES     = WORDS_STATS + 256;

// ptr_render_device_noalloc: the same words and the re-trace queue, in an array that belongs to the code object instead of
// coming from hipMalloc.  Every GPU the library runs on holds its own copy (32 MiB) from the moment the library is loaded;
// all contexts on one GPU share that copy.  Host code never uses the array's name as an address: ptr_create asks for this
// GPU's address once (hipGetSymbolAddress) and everything below works with that pointer.
constexpr size_t NOALLOC_SCRATCH_BYTES = (size_t)32 << 20;
constexpr size_t NOALLOC_QUEUE_OFFSET  = (WORDS_BYTES + 255) / 256 * 256;
static __device__ __align__(256) unsigned char scratch[NOALLOC_SCRATCH_BYTES];

}  // namespace

struct ptr_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::mutex mu;
    char err[512] = "";

    // host copy of the scene sequence (float64, C-ABI plane order)
    std::vector<double> soa;
    int n_spheres = 0;
    int n_frames = 0;
    bool have_scene = false;
    double bg[3] = {0, 0, 0};
    ptr_camera cam;
    bool cam_general = false;
    // extension scene data (ptr_set_extension_f64): planes (14 rows of n_planes), Phong terms (ks, exponent per element)
    std::vector<double> planes, phong;
    int n_planes = 0, n_phong = 0;     // n_phong: elements the Phong arrays were given for (0 = none)
    bool dirty = true;                 // frame blocks need a rebuild + upload
    bool needs_exact = false;          // some frame holds values the float32 path cannot take

    DeviceBuffer scenes;               // n_frames frame blocks
    size_t frame_stride = 0;
    DeviceBuffer queue;                // uint32 per pixel of the largest hybrid call so far
    DeviceBuffer words;                // see WORDS_*
    unsigned hybrid_calls = 0;         // picks the counter word of the next hybrid call
    bool words_dirty = true;           // counters/stats must be zeroed before the next use
    DeviceBuffer out8, out32f;         // staging for ptr_render
    DeviceBuffer tr_in_o, tr_in_d, tr_rgb, tr_el, tr_hit, tr_nrm;   // staging for ptr_trace_batch*
    DeviceBuffer rc_depth, rc_keys, rc_temp;                         // scratch of ptr_rc_reproject*
    DeviceBuffer rc_hit, rc_age, rc_item, rc_pel, rc_pdep;           // staging for ptr_rc_reproject (host buffers)
    DeviceBuffer rc_rgb, rc_prgb, rc_pprio, rc_hist, rc_frames;      // staging for ptr_rc_shade (host buffers)
    DeviceBuffer rc_page;                                            // scratch of ptr_rc_shade*: per-pixel element age
    RcCache* rcc = nullptr;                                          // ptr_rc_cache_create
    DeviceBuffer rcm_before, rcm_local_el, rcm_local_dep, rcm_all_el, rcm_all_dep;   // ptr_rc_merge_shards_device / _multi_device
    std::vector<unsigned char> host_blocks;
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};   // PTR_FLAG_TIMING
    int timed_kernels = 0;             // kernels bracketed by ev[] in the last timed call
    // The device-pointer entry points only enqueue, on the caller's stream: last_ev marks the end of the most recent
    // call's kernels, so that a later call on another stream, a scene upload or ptr_get_stats can order itself behind them.
    hipEvent_t last_ev = nullptr;
    hipStream_t last_stream = nullptr;
    bool have_last = false;
    // multi-GPU (ptr_comm_*, ptr_gather_strips_*): this context's membership of an RCCL communicator
    void* comm = nullptr;                   // ncclComm_t
    int comm_rank = 0, comm_n = 0;          // comm_n 0: no communicator
    DeviceBuffer gather_stage;              // root: the peers' local buffers as they arrive
    DeviceBuffer multi_local, multi_full;   // ptr_render_multi: this GPU's part / the whole frame on the root
    unsigned char* stats_words = nullptr;   // where the last call's kernels put their stats words
    unsigned char* scratch_dev = nullptr;   // this GPU's address of the static array `scratch`
};

namespace {

int fail(ptr_ctx* c, int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(c ? c->err : g_create_error, 512, fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(c, call)                                                                      \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail((c), PTR_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_));     \
    } while (0)

#define PTR_TRY(expr)                                                                         \
    do { int rc_ = (expr); if (rc_ != PTR_OK) return rc_; } while (0)

int grow(ptr_ctx* c, DeviceBuffer& b, size_t bytes)
{
    if (bytes <= b.bytes) return PTR_OK;
    if (b.ptr) { HIP_TRY(c, hipFree(b.ptr)); b.ptr = nullptr; b.bytes = 0; }
    HIP_TRY(c, hipMalloc(&b.ptr, bytes));
    b.bytes = bytes;
    return PTR_OK;
}

void rc_cache_free(ptr_ctx* c)
{
    RcCache* r = c->rcc;
    if (!r) return;
    for (DeviceBuffer* b : {&r->item_pixel, &r->item_hit, &r->item_age, &r->item_rgb, &r->pixel_element, &r->pixel_depth, &r->pixel_rgb,
                            &r->pixel_priority, &r->pixel_sample, &r->words, &r->cand, &r->picks, &r->pick_item, &r->need, &r->rank,
                            &r->dirs, &r->tr_rgb, &r->tr_hit, &r->frames, &r->temp})
        if (b->ptr) (void)hipFree(b->ptr);
    if (r->host_words) (void)hipHostFree(r->host_words);
    if (r->host_picks) (void)hipHostFree(r->host_picks);
    delete r;
    c->rcc = nullptr;
}

void default_camera(ptr_camera& cam)
{
    // RT/src/RayTracer.js:150-155,171-174: origin 0, fov 30, looking down -Z
    std::memset(&cam, 0, sizeof cam);
    cam.right[0] = 1.0; cam.up[1] = 1.0; cam.forward[2] = -1.0;
    cam.tan_half_fov = std::tan(3.141592653589793 * 0.5 * 30.0 / 180.0);
}

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct BlockLayout {
    uint32_t off_g64, off_m64, off_p32, off_m32, off_lights, off_pairs, off_lrec, off_cull, off_x4;
    uint32_t off_pl64, off_plm64, off_pl32, off_plm32, off_ph64, off_ph32;
    size_t   bytes;
};

BlockLayout block_layout(int n, int np = 0, bool phong = false)
{
    BlockLayout L;
    size_t o = sizeof(FrameHeader);
    L.off_g64 = (uint32_t)o;    o += align_up(sizeof(SphereG64) * (size_t)n, 16);
    L.off_m64 = (uint32_t)o;    o += align_up(sizeof(SphereM64) * (size_t)n, 16);
    o = align_up(o, 64);
    L.off_p32 = (uint32_t)o;    o += sizeof(SphereP32) * ((size_t)n + 2);   // + 2 zero records: look-ahead loads
    L.off_m32 = (uint32_t)o;    o += align_up(sizeof(SphereM32) * (size_t)n, 16);
    L.off_lights = (uint32_t)o; o += align_up(sizeof(int32_t) * (size_t)(n > 0 ? n : 1), 16);
    // room for every element being a light: [n][n + 2] pre-filter records would be quadratic, so the
    // table is sized for at most PTR_MAX_PAIR_LIGHTS lights; further lights are never culled
    // (+ 1 row of never-cull records, used by the lights past the cap)
    o = align_up(o, 64);
    L.off_pairs = (uint32_t)o;
    o += sizeof(PairQ) * ((size_t)(n < PTR_MAX_PAIR_LIGHTS ? n : PTR_MAX_PAIR_LIGHTS) + 1) * (size_t)pair_stride(n);
    o = align_up(o, 64);
    L.off_lrec = (uint32_t)o;   o += sizeof(LightRec) * ((size_t)n + 1);
    o = align_up(o, 64);
    L.off_cull = (uint32_t)o;   o += sizeof(CullRec) * ((size_t)n + 1);
    // the bounce-loop kernel's view of the spheres: lx ly lz r2, 16 bytes each, read four at a time; padded with records no
    // ray hits (r2 = -1) to a multiple of four + four for the look-ahead
    o = align_up(o, 64);
    L.off_x4 = (uint32_t)o;     o += 16 * (align_up((size_t)n, 4) + 4);
    // planes and Phong terms (extension modes; empty otherwise)
    o = align_up(o, 64);
    L.off_pl64 = (uint32_t)o;   o += sizeof(PlaneG64) * (size_t)np;
    L.off_plm64 = (uint32_t)o;  o += sizeof(SphereM64) * (size_t)np;
    L.off_pl32 = (uint32_t)o;   o += sizeof(PlaneP32) * (size_t)np;
    L.off_plm32 = (uint32_t)o;  o += sizeof(SphereM32) * (size_t)np;
    o = align_up(o, 16);
    L.off_ph64 = (uint32_t)o;   o += phong ? sizeof(double) * 2 * ((size_t)n + (size_t)np) : 0;
    L.off_ph32 = (uint32_t)o;   o += phong ? sizeof(float) * 2 * ((size_t)n + (size_t)np) : 0;
    L.bytes = align_up(o, 256);
    return L;
}

// error bounds of the float32 primary test (see SphereP32); EPS32 = 2^-24
constexpr double EPS32   = 5.9604644775390625e-08;
constexpr double PR_RELD = 24.0 * EPS32;   // disc = b^2 - cc:  |err| <= PR_RELD * scale
constexpr double PR_RELB = 10.0 * EPS32;   // b = l.d:          |err| <= PR_RELB * sqrt(scale)

// Flatten one scene (12 planes of n doubles) into a frame block.  Returns needs_exact.
bool build_frame_block(unsigned char* dst, const BlockLayout& L, const double* soa, int n,
                       const double bg[3], const double org[3],
                       const double* planes = nullptr, int np = 0, const double* phong = nullptr)
{
    std::memset(dst, 0, L.bytes);
    FrameHeader* h = reinterpret_cast<FrameHeader*>(dst);
    SphereG64* g64 = reinterpret_cast<SphereG64*>(dst + L.off_g64);
    SphereM64* m64 = reinterpret_cast<SphereM64*>(dst + L.off_m64);
    SphereP32* p32 = reinterpret_cast<SphereP32*>(dst + L.off_p32);
    SphereM32* m32 = reinterpret_cast<SphereM32*>(dst + L.off_m32);
    int32_t* lights = reinterpret_cast<int32_t*>(dst + L.off_lights);
    auto pl = [&](int k, int i) { return soa[(size_t)k * (size_t)n + (size_t)i]; };

    bool needs_exact = false, self_skip = true;
    int nl = 0;
    for (int i = 0; i < n; i++) {
        const double cx = pl(0, i), cy = pl(1, i), cz = pl(2, i), r = pl(3, i);
        const double sr = pl(4, i), sg = pl(5, i), sb = pl(6, i);
        const double er = pl(9, i), eg = pl(10, i), eb = pl(11, i);
        for (double v : {cx, cy, cz, r, sr, sg, sb, er, eg, eb})
            if (!(std::fabs(v) <= 1e15)) needs_exact = true;          // NaN, Inf, huge
        if (sr < 0 || sg < 0 || sb < 0 || er < 0 || eg < 0 || eb < 0) needs_exact = true;

        g64[i] = SphereG64{cx, cy, cz, r * r};          // radius2 = radius*radius, Sphere.js:11
        m64[i] = SphereM64{sr, sg, sb, er, eg, eb, pl(7, i), pl(8, i)};
        if (er > 0 || eg > 0 || eb > 0) lights[nl++] = i;   // RayTracer.js:65-66

        const double lx = cx - org[0], ly = cy - org[1], lz = cz - org[2];
        const double ll = lx * lx + ly * ly + lz * lz, r2 = r * r;
        const double cc = ll - r2;
        const double scale = std::fmax(std::fmax(ll, r2), 1e-30);
        const double lnorm = std::sqrt(ll), rad = std::fabs(r);
        SphereP32 p;
        std::memset(&p, 0, sizeof p);
        p.lx = (float)lx; p.ly = (float)ly; p.lz = (float)lz; p.cc = (float)cc;
        p.md = (float)(PR_RELD * scale);
        p.kb = (float)(PR_RELD * scale / (PR_RELB * std::sqrt(scale)));
        p.sgn = cc > 0 ? -1.0f : 1.0f;
        p.unsure = (std::fabs(cc) <= 4.0 * PR_RELD * scale) ? 1.0f : 0.0f;
        p.eb = (float)(PR_RELB * std::sqrt(scale));
        {
            // a ray whose float32 b = l.d is below this cannot hit, nor come within the error bound of hitting
            const double lo = (cc > 0 ? std::sqrt(cc) : 0.0) - PR_RELB * std::sqrt(scale) * (1.0 + 4 * EPS32);
            float f = (float)lo;
            if ((double)f > lo) f = std::nextafterf(f, -INFINITY);
            p.sclo = std::isfinite(lo) ? f : -3.0e38f;
        }
        // "a shadow ray never hits the sphere it starts on (from outside)" holds in the reference's float64 arithmetic
        // while its rounding (2^-52 of the coordinates and of the 1e8 far clip) is far below the 1e-4 shadow bias
        if (!(std::fabs(cx) + std::fabs(cy) + std::fabs(cz) + rad + std::fabs(org[0]) + std::fabs(org[1]) +
              std::fabs(org[2]) < 5.0e8) || !(rad > 0))
            self_skip = false;
        p.r2 = (float)r2;
        p.lnorm = (float)(lnorm * (1.0 + 2 * EPS32) + 2.1e-4);   // upper bound of |l| + shadow bias
        p.inv_r = (float)(1.0 / std::fmax(rad, 1e-30));
        {
            // half the error bound of disc' = r^2 - |l x d|^2: 10 eps r |l| + 2 eps r^2, with |l| as padded above, half an
            // eps more on each term for the rounding of the record's own values, rounded up
            const double hed = 10.5 * EPS32 * rad * (double)p.lnorm + 2.5 * EPS32 * r2;
            float f = (float)hed;
            if ((double)f < hed) f = std::nextafterf(f, INFINITY);
            // The other form, disc = b^2 - cc, errs by at most (2|b| + eb) eb + 2 eps (b^2 + |cc|) with |b| <= |l|: where that is
            // below 2 hed for EVERY ray (the floor of the example scene: r = 10^4, the eye 4 above it) the few-sphere kernel
            // refines the winner with it - three to four times less error in the hit point, and so in everything shaded from it.
            const double ebd = PR_RELB * std::sqrt(scale), ln = (double)p.lnorm;
            const bool form_b = 1.001 * ((2.0 * ln + ebd) * ebd + 2.0 * EPS32 * (ln * ln + std::fabs(cc))) < 2.0 * hed;
            p.hed = form_b ? -f : f;
        }
        p.refl = (float)pl(7, i); p.transp = (float)pl(8, i);
        p32[i] = p;
        SphereM32 m;
        m.sr = (float)sr; m.sg = (float)sg; m.sb = (float)sb;
        m.scmax = (float)std::fmax(sr, std::fmax(sg, sb)) * 1.0000002f;
        m.er = (float)er; m.eg = (float)eg; m.eb = (float)eb;
        m.emax = (float)std::fmax(er, std::fmax(eg, eb)) * 1.0000002f;
        m32[i] = m;
    }
    // shadow pre-filter table (see PairQ)
    PairQ* pairs = reinterpret_cast<PairQ*>(dst + L.off_pairs);
    for (int k = 0; k < nl && k < PTR_MAX_PAIR_LIGHTS; k++) {
        const int i = lights[k];
        for (int j = 0; j < n; j++) {
            PairQ q{0.0f, 0.0f, 0.0f, -1.0f};                        // thr < 0: never culled
            const double ax = pl(0, j) - pl(0, i), ay = pl(1, j) - pl(1, i), az = pl(2, j) - pl(2, i);
            const double D = std::sqrt(ax * ax + ay * ay + az * az);
            const double reach = std::fabs(pl(3, j)) + 1.0e-4 + 1.0e-6 * (std::fabs(pl(3, j)) + D);   // r + bias + slop
            if (D > 0 && reach < D && std::isfinite(D)) {
                const double x = reach / D;
                const double thr = std::sqrt(1.0 - x * x);
                q.ax = (float)(ax / D); q.ay = (float)(ay / D); q.az = (float)(az / D);
                q.thr = (float)(thr * (1.0 - 4 * EPS32) - 13 * EPS32);   // rounding of a, of the dot, of thr itself
            }
            pairs[(size_t)k * (size_t)pair_stride(n) + (size_t)j] = q;
        }
        for (int j = n; j < pair_stride(n); j++) pairs[(size_t)k * (size_t)pair_stride(n) + (size_t)j] = PairQ{0, 0, 0, -1.0f};
    }
    {
        const size_t row = (size_t)(n < PTR_MAX_PAIR_LIGHTS ? n : PTR_MAX_PAIR_LIGHTS);
        for (int j = 0; j < pair_stride(n); j++) pairs[row * (size_t)pair_stride(n) + (size_t)j] = PairQ{0, 0, 0, -1.0f};
    }
    {
        float* x4 = reinterpret_cast<float*>(dst + L.off_x4);
        const size_t padded = align_up((size_t)n, 4) + 4;
        for (size_t i = 0; i < padded; i++) {
            const bool real = i < (size_t)n;
            x4[4 * i] = real ? p32[i].lx : 0.0f; x4[4 * i + 1] = real ? p32[i].ly : 0.0f; x4[4 * i + 2] = real ? p32[i].lz : 0.0f;
            x4[4 * i + 3] = real ? p32[i].r2 : -1.0f;
        }
    }
    CullRec* cull = reinterpret_cast<CullRec*>(dst + L.off_cull);        // (record n: zeros, read by the lanes past n)
    for (int i = 0; i < n; i++) {
        const SphereP32& p = p32[i];
        // from the ROUNDED record values, the ones the kernel's hit test works with
        const double cce = (double)p.cc - 2.0 * (double)p.md, re2 = (double)p.r2 + 2.0 * (double)p.md;
        float s = -3.0e38f, re = (float)std::sqrt(re2);
        if (cce > 0 && std::isfinite(cce)) { s = (float)std::sqrt(cce); if ((double)s * (double)s > cce) s = std::nextafterf(s, -INFINITY); }
        if ((double)re * (double)re < re2) re = std::nextafterf(re, INFINITY);
        if (!std::isfinite(re2)) s = -3.0e38f, re = 0.0f;
        cull[i] = CullRec{p.lx, p.ly, p.lz, s, re, 1e-5f * p.lnorm, 0.0f, 0.0f};
    }
    LightRec* lrec = reinterpret_cast<LightRec*>(dst + L.off_lrec);      // (the block is zeroed: record nl is the padding)
    for (int k = 0; k < nl; k++) {
        const SphereP32& p = p32[lights[k]];
        const SphereM32& m = m32[lights[k]];
        lrec[k] = LightRec{p.lx, p.ly, p.lz, p.lnorm, m.er, m.eg, m.eb, m.emax, lights[k], (float)EPS32 * p.lnorm * 1.0000002f, {0, 0}};
    }
    for (int k = 0; k < 3; k++) {
        h->bg64[k] = bg[k];
        h->bg32[k] = (float)bg[k];
    }
    {
        // PTR_MODE_TOLERANT: |float32 colour - float64 colour| <= scmax * sum_k emax_k * (|error of N.L_k| + 15 eps) (+ the
        // roundings of the sums, far below); N.L_k is off by at most eN + eL_k, which the kernel sums over the lights (S).
        // 255 * scmax * (sum_k emax_k) * (S + 15 eps) <= 0.9 keeps the two colours within one 8-bit unit BEFORE rounding, so the
        // rounded bytes differ by at most 1.  No lights: nothing is shaded from N.L, any S will do.
        double scm = 0.0, esum = 0.0;
        for (int i = 0; i < n; i++) scm = std::fmax(scm, (double)m32[i].scmax);
        for (int k = 0; k < nl; k++) esum += (double)m32[lights[k]].emax;
        const double den = 255.0 * scm * esum;
        const double thr = den > 0 ? 0.9 / den - 16.0 * EPS32 : 3.0e38;
        h->tol_thr = std::isfinite(thr) ? (float)std::fmin(thr, 3.0e38) : -1.0f;     // (<= 0: every hit pixel is re-traced)
    }
    h->n = n;
    h->n_lights = nl;
    h->bg_packed = pack_js(bg[0], bg[1], bg[2]);
    h->needs_exact = needs_exact ? 1u : 0u;
    h->self_skip = self_skip ? 1u : 0u;
    h->off_g64 = L.off_g64; h->off_m64 = L.off_m64; h->off_p32 = L.off_p32;
    h->off_m32 = L.off_m32; h->off_lights = L.off_lights;
    h->n_planes = (uint32_t)np; h->has_phong = phong ? 1u : 0u;
    h->off_pl64 = L.off_pl64; h->off_plm64 = L.off_plm64; h->off_pl32 = L.off_pl32; h->off_plm32 = L.off_plm32;
    h->off_ph64 = L.off_ph64; h->off_ph32 = L.off_ph32;
    {
        PlaneG64* g = reinterpret_cast<PlaneG64*>(dst + L.off_pl64);
        SphereM64* m = reinterpret_cast<SphereM64*>(dst + L.off_plm64);
        PlaneP32* g32 = reinterpret_cast<PlaneP32*>(dst + L.off_pl32);
        SphereM32* mm = reinterpret_cast<SphereM32*>(dst + L.off_plm32);
        auto xp = [&](int row, int k) { return planes[(size_t)row * (size_t)np + (size_t)k]; };
        for (int k = 0; k < np; k++) {
            g[k] = PlaneG64{xp(0, k), xp(1, k), xp(2, k), xp(3, k), xp(4, k), xp(5, k)};
            m[k] = SphereM64{xp(6, k), xp(7, k), xp(8, k), xp(11, k), xp(12, k), xp(13, k), xp(9, k), xp(10, k)};
            g32[k] = PlaneP32{(float)(xp(0, k) - org[0]), (float)(xp(1, k) - org[1]), (float)(xp(2, k) - org[2]), (float)xp(9, k),
                              (float)xp(3, k), (float)xp(4, k), (float)xp(5, k), (float)xp(10, k)};
            mm[k] = SphereM32{(float)xp(6, k), (float)xp(7, k), (float)xp(8, k), 0.0f, (float)xp(11, k), (float)xp(12, k), (float)xp(13, k), 0.0f};
        }
        if (phong) {
            double* p64 = reinterpret_cast<double*>(dst + L.off_ph64);
            float* p32f = reinterpret_cast<float*>(dst + L.off_ph32);
            for (int k = 0; k < 2 * (n + np); k++) { p64[k] = phong[k]; p32f[k] = (float)phong[k]; }
        }
    }
    return needs_exact;
}

// make stream s wait for the kernels of the context's most recent call, if that call went to another stream
int order_after_last(ptr_ctx* c, hipStream_t s)
{
    if (c->have_last && c->last_stream != s) HIP_TRY(c, hipStreamWaitEvent(s, c->last_ev, 0));
    return PTR_OK;
}

int mark_launch(ptr_ctx* c, hipStream_t s)
{
    HIP_TRY(c, hipEventRecord(c->last_ev, s));
    c->last_stream = s;
    c->have_last = true;
    return PTR_OK;
}

// Rebuild + upload the frame blocks if scene, background or camera changed.  The copy goes onto the stream the kernels
// will be launched on, so it is ordered behind the kernels of earlier calls that may still be reading the old blocks
// (the device-pointer entry points do not wait for their kernels), and the host waits for it: host_blocks is rebuilt in
// place by the next upload.
int ensure_uploaded(ptr_ctx* c, hipStream_t s)
{
    if (!c->dirty) return PTR_OK;
    const bool phong = c->n_phong > 0;
    if (phong && c->n_phong != c->n_spheres + c->n_planes)
        return fail(c, PTR_ERR_INVALID, "the Phong terms were given for %d elements, the scene has %d spheres + %d planes "
                                        "(ptr_set_extension_f64 after ptr_set_scene*)", c->n_phong, c->n_spheres, c->n_planes);
    const BlockLayout L = block_layout(c->n_spheres, c->n_planes, phong);
    try {
        c->host_blocks.resize(L.bytes * (size_t)c->n_frames);
    } catch (const std::bad_alloc&) {
        return fail(c, PTR_ERR_INVALID, "out of host memory building the frame blocks");
    }
    bool ne = false;
    for (int f = 0; f < c->n_frames; f++)
        ne |= build_frame_block(c->host_blocks.data() + L.bytes * (size_t)f, L,
                                c->soa.data() + (size_t)f * 12u * (size_t)c->n_spheres, c->n_spheres,
                                c->bg, c->cam.origin, c->planes.data(), c->n_planes, phong ? c->phong.data() : nullptr);
    PTR_TRY(order_after_last(c, s));
    PTR_TRY(grow(c, c->scenes, c->host_blocks.size()));          // (a hipFree waits for the whole device)
    HIP_TRY(c, hipMemcpyAsync(c->scenes.ptr, c->host_blocks.data(), c->host_blocks.size(),
                              hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    c->frame_stride = L.bytes;
    c->needs_exact = ne;
    c->dirty = false;
    return PTR_OK;
}

int check_params(ptr_ctx* c, const ptr_params* p, int first_frame, int n_frames)
{
    if (!p) return fail(c, PTR_ERR_INVALID, "params is NULL");
    if (!c->have_scene) return fail(c, PTR_ERR_NO_SCENE, "no scene uploaded (call ptr_set_scene* first)");
    if (p->width < 1 || p->height < 1 || p->scan_height < 1)
        return fail(c, PTR_ERR_INVALID, "width, height and scan_height must be >= 1 (got %d, %d, %d)",
                    p->width, p->height, p->scan_height);
    if (p->width >= (1 << 24))
        return fail(c, PTR_ERR_INVALID, "width must be < 2^24 (got %d)", p->width);
    if (p->start_y < -(1 << 30) || p->start_y > (1 << 30))
        return fail(c, PTR_ERR_INVALID, "start_y out of range (%d)", p->start_y);
    if (n_frames < 1 || first_frame < 0 || first_frame + n_frames > c->n_frames)
        return fail(c, PTR_ERR_INVALID, "frames [%d, %d) outside the uploaded sequence of %d",
                    first_frame, first_frame + n_frames, c->n_frames);
    if (n_frames > 65535)
        return fail(c, PTR_ERR_INVALID, "at most 65535 frames per call (got %d)", n_frames);
    const uint64_t px = (uint64_t)p->width * (uint64_t)p->scan_height * (uint64_t)n_frames;
    if (px >= (1ull << 32))
        return fail(c, PTR_ERR_INVALID, "width*scan_height*n_frames must be < 2^32 (got %llu)",
                    (unsigned long long)px);
    if ((p->scan_height + PTR_TILE_H - 1) / PTR_TILE_H > 65535)
        return fail(c, PTR_ERR_INVALID, "scan_height too large (%d)", p->scan_height);
    if (p->max_depth < 0 || p->max_depth > PTR_MAX_DEPTH_LIMIT || p->spp < 1 || p->spp > PTR_MAX_SPP)
        return fail(c, PTR_ERR_INVALID, "max_depth must be in [0, %d] and spp in [1, %d] (got %d, %d)",
                    PTR_MAX_DEPTH_LIMIT, PTR_MAX_SPP, p->max_depth, p->spp);
    if (p->max_depth != 0 || p->spp != 1 || c->n_planes > 0 || c->n_phong > 0) {
        // extension modes (ptr_kernels_ext.hip): not reference behaviour, so no reference quirk and no
        // reference work counters exist for them
        if (p->flags & PTR_QUIRK_ROW_SHIFT)
            return fail(c, PTR_ERR_INVALID, "PTR_QUIRK_ROW_SHIFT applies to the reference semantics only "
                                            "(max_depth 0, spp 1, no planes, no Phong terms); got max_depth=%d spp=%d planes=%d",
                        p->max_depth, p->spp, c->n_planes);
    }
    if (((p->flags & PTR_MODE_EXACT_F64) != 0) + ((p->flags & PTR_MODE_F32_ONLY) != 0) + ((p->flags & PTR_MODE_TOLERANT) != 0) > 1)
        return fail(c, PTR_ERR_INVALID, "PTR_MODE_EXACT_F64, PTR_MODE_F32_ONLY and PTR_MODE_TOLERANT are exclusive");
    return PTR_OK;
}

int ensure_words(ptr_ctx* c, hipStream_t s)
{
    PTR_TRY(grow(c, c->words, WORDS_BYTES));
    if (c->words_dirty) {
        HIP_TRY(c, hipMemsetAsync(c->words.ptr, 0, WORDS_BYTES, s));
        c->words_dirty = false;
        c->hybrid_calls = 0;
    }
    return PTR_OK;
}

// Enqueue the kernels of one render call; no synchronisation.  fixed_ws: the kernels' words and re-trace queue come from
// the static array (ptr_render_device_noalloc) and nothing in here allocates; otherwise from the context's own buffers,
// grown on demand.
int enqueue_render(ptr_ctx* c, const ptr_params* p, int first_frame, int n_frames,
                   uint32_t* d_out8, float* d_out32f, hipStream_t s,
                   const ptr_strips* strips = nullptr, uint64_t frame_stride_px = 0, bool fixed_ws = false)
{
    PTR_TRY(order_after_last(c, s));          // queue, counters and stats words are shared by all calls on the context
    if (fixed_ws) {
        if (c->dirty)
            return fail(c, PTR_ERR_INVALID, "ptr_render_device_noalloc: the scene is not resident on the GPU yet; call "
                                            "ptr_sync() after the last ptr_set_* call (the upload allocates)");
        if (p->flags & PTR_FLAG_TIMING)
            return fail(c, PTR_ERR_INVALID, "ptr_render_device_noalloc: PTR_FLAG_TIMING is not available (it creates events)");
        // every call on this GPU shares the array and finds the last call's contents in it: clear what must start at zero
        HIP_TRY(c, hipMemsetAsync(c->scratch_dev, 0, WORDS_BYTES, s));
    } else {
        PTR_TRY(ensure_uploaded(c, s));
        PTR_TRY(ensure_words(c, s));
    }

    RenderArgs a;
    std::memset(&a, 0, sizeof a);
    // (the kernels see the call's first frame as frame 0; the few-sphere kernel counts on a call's blocks spanning < 4 GB)
    a.scenes = static_cast<const unsigned char*>(c->scenes.ptr) + (size_t)first_frame * c->frame_stride;
    a.frame_stride = c->frame_stride;
    {
        const BlockLayout L = block_layout(c->n_spheres);
        a.off_p32 = L.off_p32; a.off_m32 = L.off_m32; a.off_lights = L.off_lights; a.off_pairs = L.off_pairs; a.off_lrec = L.off_lrec; a.off_cull = L.off_cull; a.off_x4 = L.off_x4;
        a.n = c->n_spheres;
    }
    a.first_frame = 0;
    a.n_frames = n_frames;
    a.width = p->width; a.height = p->height; a.start_y = p->start_y; a.scan_height = p->scan_height;
    a.flags = p->flags;
    a.strip_h = strips ? strips->strip_height : 1;
    a.n_parts = strips ? strips->n_parts : 1;
    a.part = strips ? strips->part : 0;
    a.rows_mode = a.n_parts <= 1 ? 0 : (a.strip_h % PTR_TILE_H == 0 ? 1 : 2);
    a.out_frame_stride = frame_stride_px ? frame_stride_px : (uint64_t)p->width * (uint64_t)p->scan_height;
    for (int k = 0; k < 3; k++) {
        a.cam.org[k] = c->cam.origin[k];
        a.cam.right[k] = c->cam.right[k]; a.cam.up[k] = c->cam.up[k]; a.cam.fwd[k] = c->cam.forward[k];
        a.camf.right[k] = (float)c->cam.right[k]; a.camf.up[k] = (float)c->cam.up[k];
        a.camf.fwd[k] = (float)c->cam.forward[k];
    }
    a.cam.angle = c->cam.tan_half_fov;
    a.cam.aspect = (double)p->width / (double)p->height;      // RayTracer.js:153
    a.cam.inv_w = 1.0 / (double)p->width;                     // RayTracer.js:150
    a.cam.inv_h = 1.0 / (double)p->height;                    // RayTracer.js:151
    a.cam.general = c->cam_general ? 1 : 0;
    a.camf.sx = (float)(2.0 * a.cam.inv_w * a.cam.angle * a.cam.aspect);
    a.camf.ox = (float)(-a.cam.angle * a.cam.aspect);
    a.camf.sy = (float)(-2.0 * a.cam.inv_h * a.cam.angle);
    a.camf.oy = (float)(a.cam.angle);
    bool cam_exact = false;
    {
        // Tile culling of the float32 kernel: a wave's pixels are a (8 PPL) x 8 block of the image when its 8 local rows
        // are consecutive image rows.  Unnormalised directions D = xx R + yy U + F differ from the tile centre's by at
        // most rho, and |D| >= dmin (distance of the image plane from the eye), so sin(angle) <= rho / dmin.
        const double* R = c->cam.right; const double* U = c->cam.up; const double* F = c->cam.forward;
        const double nx = R[1] * U[2] - R[2] * U[1], ny = R[2] * U[0] - R[0] * U[2], nz = R[0] * U[1] - R[1] * U[0];
        const double nn = std::sqrt(nx * nx + ny * ny + nz * nz);
        const double dmin = nn > 0 ? std::fabs(F[0] * nx + F[1] * ny + F[2] * nz) / nn : 0.0;
        const int fast_mode = (p->flags & PTR_MODE_F32_ONLY) ? 0 : ((p->flags & PTR_MODE_TOLERANT) ? 2 : 1);
        const double rho = (4.0 * fast_kernel_ppl(fast_mode) - 0.5) * std::fabs((double)a.camf.sx) * std::sqrt(R[0] * R[0] + R[1] * R[1] + R[2] * R[2]) +
                           (0.5 * PTR_TILE_H - 0.5) * std::fabs((double)a.camf.sy) * std::sqrt(U[0] * U[0] + U[1] * U[1] + U[2] * U[2]);
        const double sn = dmin > 0 ? 1.02 * rho / dmin + 1e-6 : 2.0;
        const bool rows_consecutive = a.n_parts <= 1 || a.strip_h % PTR_TILE_H == 0;
        a.cull_on = (rows_consecutive && sn < 0.5) ? 1 : 0;
        a.cull_sin = (float)(sn * (1.0 + 1e-6));
        a.cull_cos = (float)(std::sqrt(std::fmax(0.0, 1.0 - sn * sn)) * (1.0 - 1e-6));
        // (the bounce-loop kernel's waves cover 8 x 8 pixels, and a jittered sample lies anywhere in its pixel: 4 pixels each way)
        const double rho_x = 4.0 * std::fabs((double)a.camf.sx) * std::sqrt(R[0] * R[0] + R[1] * R[1] + R[2] * R[2]) +
                             4.0 * std::fabs((double)a.camf.sy) * std::sqrt(U[0] * U[0] + U[1] * U[1] + U[2] * U[2]);
        const double snx = dmin > 0 ? 1.02 * rho_x / dmin + 1e-6 : 2.0;
        a.ext_cull_on = (rows_consecutive && snx < 0.5) ? 1 : 0;
        a.ext_cull_sin = (float)(snx * (1.0 + 1e-6));
        a.ext_cull_cos = (float)(std::sqrt(std::fmax(0.0, 1.0 - snx * snx)) * (1.0 - 1e-6));
        // The float32 error budget of the hit tests (PR_RELB: 10 eps for b = l.d) counts on a camera ray that is good to
        // about one eps, which is what xx R + yy U + F gives while its terms are of the size of the sum: the direction's
        // relative error grows like g = (|xx|max |R| + |yy|max |U| + |F|) / dmin (1.7 for the reference camera at 16:9;
        // measured 1.1 eps there, 3.3 eps at g = 10-11: fov 150 degrees, or aspect 32).  Wide or skewed cameras beyond
        // g = 4 are rendered in float64 rather than trusted to the budget.
        const double g = dmin > 0 ? (std::fabs(a.cam.angle * a.cam.aspect) * std::sqrt(R[0] * R[0] + R[1] * R[1] + R[2] * R[2]) +
                                     std::fabs(a.cam.angle) * std::sqrt(U[0] * U[0] + U[1] * U[1] + U[2] * U[2]) +
                                     std::sqrt(F[0] * F[0] + F[1] * F[1] + F[2] * F[2])) / dmin : 1e300;
        cam_exact = !(g <= 4.0);
    }
    a.out8 = d_out8;
    a.out32f = d_out32f;
    unsigned char* words = fixed_ws ? c->scratch_dev : static_cast<unsigned char*>(c->words.ptr);
    a.stats = reinterpret_cast<unsigned long long*>(words + WORDS_STATS);
    c->stats_words = words + WORDS_STATS;

    const bool stats = (p->flags & PTR_FLAG_STATS) != 0;
    // (PTR_MODE_F32_ONLY is the caller's explicit choice of float32 whatever the camera)
    const bool exact = stats || (p->flags & PTR_MODE_EXACT_F64) || c->needs_exact ||
                       (cam_exact && !(p->flags & PTR_MODE_F32_ONLY) && p->max_depth == 0 && p->spp == 1);
    const bool timing = (p->flags & PTR_FLAG_TIMING) != 0;
    if (stats && !fixed_ws) HIP_TRY(c, hipMemsetAsync(words + WORDS_STATS, 0, 256, s));
    if (timing) {
        for (auto& e : c->ev)
            if (!e) HIP_TRY(c, hipEventCreate(&e));
        c->timed_kernels = 0;
        HIP_TRY(c, hipEventRecord(c->ev[0], s));
    }

    if (p->max_depth != 0 || p->spp != 1 || c->n_planes > 0 || c->n_phong > 0) {
        // Extension modes (bounces, samples, planes, Phong terms): float64 = the specification bit for bit, and that is the DEFAULT: a float32 ray tree has no
        // error bounds yet that would tell which pixels to re-do (errors grow with every bounce off a curved surface), and
        // the contract is <= 1/255 on every pixel.  PTR_MODE_F32_ONLY asks for the float32 throughput form explicitly.
        a.max_depth = p->max_depth; a.spp = p->spp; a.seed = p->seed;
        HIP_TRY(c, launch_render_ext(a, n_frames, exact || !(p->flags & PTR_MODE_F32_ONLY), stats, c->n_planes == 0 && c->n_phong == 0, s));
        if (timing) { HIP_TRY(c, hipEventRecord(c->ev[1], s)); c->timed_kernels = 1; }
        return mark_launch(c, s);
    }
    if (exact) {
        HIP_TRY(c, launch_render_exact(a, n_frames, stats, s));
        if (timing) { HIP_TRY(c, hipEventRecord(c->ev[1], s)); c->timed_kernels = 1; }
        return mark_launch(c, s);
    }
    const bool many = c->n_spheres > PTR_FAST_MAX_SPHERES || (p->flags & PTR_DEBUG_FORCE_MANY);
    if (!many && (uint64_t)n_frames * (uint64_t)c->frame_stride >= (1ull << 32))
        return fail(c, PTR_ERR_INVALID, "internal: %d frame blocks of %zu bytes span 4 GB or more", n_frames, c->frame_stride);
    if (p->flags & PTR_MODE_F32_ONLY) {
        HIP_TRY(c, many ? launch_render_many(a, n_frames, false, s) : launch_render_fast(a, n_frames, 0, s));
        if (timing) { HIP_TRY(c, hipEventRecord(c->ev[1], s)); c->timed_kernels = 1; }
        return mark_launch(c, s);
    }
    // hybrid: float32 kernel fills the sub-queues, float64 kernel drains them
    const int rpb = many ? many_kernel_rows_per_block() : fast_kernel_rows_per_block();
    const int cpb = many ? PTR_TILE_W : fast_kernel_cols_per_block((p->flags & PTR_MODE_TOLERANT) ? 2 : 1);
    const uint64_t n_blocks = (uint64_t)((p->width + cpb - 1) / cpb) *
                              (uint64_t)((p->scan_height + rpb - 1) / rpb) * (uint64_t)n_frames;
    uint64_t n_queues = 1;                                     // a power of two (the kernel masks)
    while (n_queues * 2 <= PTR_MAX_QUEUES && n_queues * 8 <= n_blocks) n_queues *= 2;
    // a workgroup can queue at most PTR_BLOCK pixels; sub-queue s serves the workgroups b with (b & (n_queues - 1)) == s
    const uint64_t sub_capacity = (n_blocks + n_queues - 1) / n_queues * (uint64_t)(cpb * rpb);
    const size_t queue_bytes = (size_t)(n_queues * sub_capacity) * (many ? 3 : 2) * sizeof(uint32_t);   // ids + 1 or 2 info words
    if (fixed_ws) {
        if (queue_bytes > NOALLOC_SCRATCH_BYTES - NOALLOC_QUEUE_OFFSET)
            return fail(c, PTR_ERR_INVALID, "ptr_render_device_noalloc: this call needs a re-trace queue of %zu bytes (8 per "
                                            "pixel of its tiles, 12 with more than 12 spheres), the static array holds %zu: render fewer pixels per call "
                                            "(about 3.9 M at most), or use ptr_render_device",
                        queue_bytes, NOALLOC_SCRATCH_BYTES - NOALLOC_QUEUE_OFFSET);
    } else {
        PTR_TRY(grow(c, c->queue, queue_bytes));
    }
    // (the static array was cleared above: its calls always use counter set A)
    const bool odd = !fixed_ws && (c->hybrid_calls & 1u) != 0;
    a.queue = fixed_ws ? reinterpret_cast<uint32_t*>(c->scratch_dev + NOALLOC_QUEUE_OFFSET) : static_cast<uint32_t*>(c->queue.ptr);
    a.qinfo = a.queue + n_queues * sub_capacity;
    a.qinfo_packed = many ? 0u : 1u;
    a.queue_capacity = (uint32_t)sub_capacity;
    a.n_queues = (uint32_t)n_queues;
    a.counter = reinterpret_cast<uint32_t*>(words + (odd ? WORDS_COUNTER_B : WORDS_COUNTER_A));
    a.counter_next = reinterpret_cast<uint32_t*>(words + (odd ? WORDS_COUNTER_A : WORDS_COUNTER_B));
    if (!fixed_ws) {
        c->hybrid_calls++;
        HIP_TRY(c, hipMemsetAsync(words + WORDS_STATS, 0, 256, s));
    }
    // (PTR_MODE_TOLERANT: the few-sphere kernel has a form of its own; the many-sphere kernel keeps every doubt)
    hipError_t e = many ? launch_render_many(a, n_frames, true, s)
                        : launch_render_fast(a, n_frames, (p->flags & PTR_MODE_TOLERANT) ? 2 : 1, s);
    if (e == hipSuccess && timing) e = hipEventRecord(c->ev[1], s);
    if (e == hipSuccess) {
        e = many ? launch_retrace_exact(a, 8, 8, s) : launch_retrace_exact(a, 1, 4, s);
        if (e == hipSuccess && timing) { e = hipEventRecord(c->ev[2], s); c->timed_kernels = 2; }
    }
    if (e != hipSuccess) {
        c->words_dirty = true;
        return fail(c, PTR_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(e));
    }
    return mark_launch(c, s);
}

int set_scene_locked(ptr_ctx* c, const double* soa, int n, int n_frames)
{
    if (n < 0 || n_frames < 1) return fail(c, PTR_ERR_INVALID, "n_spheres must be >= 0 and n_frames >= 1");
    if (n > 0 && !soa) return fail(c, PTR_ERR_INVALID, "soa is NULL");
    if ((uint64_t)n * (uint64_t)n_frames > (1ull << 28))
        return fail(c, PTR_ERR_INVALID, "scene sequence too large (%d spheres x %d frames)", n, n_frames);
    try {
        c->soa.assign(soa, soa + (size_t)12 * (size_t)n * (size_t)n_frames);
    } catch (const std::bad_alloc&) {
        return fail(c, PTR_ERR_INVALID, "out of host memory copying the scene");
    }
    c->n_spheres = n;
    c->n_frames = n_frames;
    c->have_scene = true;
    c->dirty = true;
    return PTR_OK;
}

}  // namespace

// =============================================================================================
extern "C" {

int ptr_abi_version(void) { return PTR_ABI_VERSION; }

const char* ptr_last_error(ptr_ctx* ctx)
{
    return ctx ? ctx->err : g_create_error;
}

int ptr_create(int device, ptr_ctx** out)
{
    if (!out) return fail(nullptr, PTR_ERR_INVALID, "out is NULL");
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count < 1) {
        (void)hipGetLastError();
        return fail(nullptr, PTR_ERR_NO_DEVICE,
                    "no HIP device available (%s); this library has no CPU fallback",
                    e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
    }
    if (device < 0 || device >= count)
        return fail(nullptr, PTR_ERR_INVALID, "device %d out of range (0..%d)", device, count - 1);
    e = hipSetDevice(device);
    if (e != hipSuccess) return fail(nullptr, PTR_ERR_NO_DEVICE, "hipSetDevice(%d): %s", device, hipGetErrorString(e));
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) return fail(nullptr, PTR_ERR_NO_DEVICE, "hipGetDeviceProperties: %s", hipGetErrorString(e));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, PTR_ERR_NO_DEVICE, "device %d is %s; this library is built for gfx950 only",
                    device, prop.gcnArchName);
    ptr_ctx* c = new (std::nothrow) ptr_ctx();
    if (!c) return fail(nullptr, PTR_ERR_INVALID, "out of host memory");
    c->device = device;
    default_camera(c->cam);
    e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete c;
        return fail(nullptr, PTR_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e));
    }
    e = hipEventCreateWithFlags(&c->last_ev, hipEventDisableTiming);
    // this GPU's copy of the static array (the device is current): the address differs from GPU to GPU
    if (e == hipSuccess) e = hipGetSymbolAddress(reinterpret_cast<void**>(&c->scratch_dev), HIP_SYMBOL(scratch));
    if (e != hipSuccess || !c->scratch_dev) {
        if (c->last_ev) (void)hipEventDestroy(c->last_ev);
        (void)hipStreamDestroy(c->stream);
        delete c;
        return fail(nullptr, PTR_ERR_HIP, "ptr_create: event / hipGetSymbolAddress(scratch): %s", hipGetErrorString(e));
    }
    *out = c;
    return PTR_OK;
}

void ptr_destroy(ptr_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) { (void)hipStreamSynchronize(c->stream); }
    if (c->have_last) (void)hipEventSynchronize(c->last_ev);     // kernels enqueued on a caller's stream still read our buffers
    if (c->comm) (void)ptr_comm_destroy(c);
    rc_cache_free(c);
    for (DeviceBuffer* b : {&c->gather_stage, &c->multi_local, &c->multi_full, &c->scenes, &c->queue, &c->words, &c->out8, &c->out32f, &c->tr_in_o,
                            &c->tr_in_d, &c->tr_rgb, &c->tr_el, &c->tr_hit, &c->tr_nrm, &c->rc_depth, &c->rc_keys,
                            &c->rc_temp, &c->rc_hit, &c->rc_age, &c->rc_item, &c->rc_pel, &c->rc_pdep, &c->rc_rgb,
                            &c->rc_prgb, &c->rc_pprio, &c->rc_hist, &c->rc_frames, &c->rc_page, &c->rcm_before, &c->rcm_local_el,
                            &c->rcm_local_dep, &c->rcm_all_el, &c->rcm_all_dep})
        if (b->ptr) (void)hipFree(b->ptr);
    for (auto& e : c->ev) if (e) (void)hipEventDestroy(e);
    if (c->last_ev) (void)hipEventDestroy(c->last_ev);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int ptr_set_scene_sequence_f64(ptr_ctx* c, const double* soa, int n_spheres, int n_frames)
{
    if (!c) return PTR_ERR_INVALID;
    std::lock_guard<std::mutex> g(c->mu);
    return set_scene_locked(c, soa, n_spheres, n_frames);
}

int ptr_set_scene_f64(ptr_ctx* c, const double* soa, int n_spheres)
{
    return ptr_set_scene_sequence_f64(c, soa, n_spheres, 1);
}

int ptr_set_scene(ptr_ctx* c, const float* soa, int n_spheres)
{
    if (!c) return PTR_ERR_INVALID;
    if (n_spheres < 0) { std::lock_guard<std::mutex> g(c->mu); return fail(c, PTR_ERR_INVALID, "n_spheres < 0"); }
    if (n_spheres > 0 && !soa) { std::lock_guard<std::mutex> g(c->mu); return fail(c, PTR_ERR_INVALID, "soa is NULL"); }
    std::vector<double> tmp;
    try {
        tmp.resize((size_t)12 * (size_t)n_spheres);
    } catch (const std::bad_alloc&) {
        std::lock_guard<std::mutex> g(c->mu);
        return fail(c, PTR_ERR_INVALID, "out of host memory copying the scene");
    }
    for (size_t k = 0; k < tmp.size(); k++) tmp[k] = (double)soa[k];
    return ptr_set_scene_sequence_f64(c, tmp.data(), n_spheres, 1);
}

int ptr_set_background_f64(ptr_ctx* c, const double rgb[3])
{
    if (!c) return PTR_ERR_INVALID;
    std::lock_guard<std::mutex> g(c->mu);
    if (!rgb) return fail(c, PTR_ERR_INVALID, "rgb is NULL");
    for (int k = 0; k < 3; k++) c->bg[k] = rgb[k];
    c->dirty = true;
    return PTR_OK;
}

int ptr_set_background(ptr_ctx* c, const float rgb[3])
{
    if (!c) return PTR_ERR_INVALID;
    if (!rgb) { std::lock_guard<std::mutex> g(c->mu); return fail(c, PTR_ERR_INVALID, "rgb is NULL"); }
    const double d[3] = {(double)rgb[0], (double)rgb[1], (double)rgb[2]};
    return ptr_set_background_f64(c, d);
}

int ptr_set_camera(ptr_ctx* c, const ptr_camera* cam)
{
    if (!c) return PTR_ERR_INVALID;
    std::lock_guard<std::mutex> g(c->mu);
    if (!cam) {
        default_camera(c->cam);
        c->cam_general = false;
    } else {
        const double* v = cam->origin;
        for (int k = 0; k < 13; k++)
            if (!(std::fabs(v[k]) <= 1e15)) return fail(c, PTR_ERR_INVALID, "camera field %d is not finite", k);
        if (!(cam->tan_half_fov > 0)) return fail(c, PTR_ERR_INVALID, "tan_half_fov must be > 0");
        c->cam = *cam;
        // the reference layout of the basis keeps the reference formula dir = (xx, yy, -1)
        c->cam_general = !(cam->right[0] == 1 && cam->right[1] == 0 && cam->right[2] == 0 &&
                           cam->up[0] == 0 && cam->up[1] == 1 && cam->up[2] == 0 &&
                           cam->forward[0] == 0 && cam->forward[1] == 0 && cam->forward[2] == -1);
    }
    c->dirty = true;     // camera-relative sphere terms depend on the origin
    return PTR_OK;
}

int ptr_set_extension_f64(ptr_ctx* c, const double* planes_soa, int n_planes, const double* phong, int n_elements)
{
    if (!c) return PTR_ERR_INVALID;
    std::lock_guard<std::mutex> g(c->mu);
    if (n_planes < 0 || n_planes > 4096 || (n_planes > 0 && !planes_soa)) return fail(c, PTR_ERR_INVALID, "bad plane list (%d planes)", n_planes);
    if (phong && n_elements < 1) return fail(c, PTR_ERR_INVALID, "Phong terms need n_elements >= 1");
    for (int k = 0; k < n_planes; k++) {
        const double nx = planes_soa[3 * (size_t)n_planes + k], ny = planes_soa[4 * (size_t)n_planes + k], nz = planes_soa[5 * (size_t)n_planes + k];
        if (!(std::fabs(nx * nx + ny * ny + nz * nz - 1.0) <= 1e-9))
            return fail(c, PTR_ERR_INVALID, "plane %d: the normal must be a unit vector", k);
        for (int row = 0; row < 14; row++)
            if (!(std::fabs(planes_soa[(size_t)row * (size_t)n_planes + k]) <= 1e15)) return fail(c, PTR_ERR_INVALID, "plane %d: value %d is not finite", k, row);
    }
    for (int k = 0; phong && k < n_elements; k++) {
        const double ks = phong[k], m = phong[(size_t)n_elements + k];
        if (!(ks >= 0 && ks <= 1e15) || !(m >= 1 && m <= 65536 && m == std::floor(m)))
            return fail(c, PTR_ERR_INVALID, "element %d: ks must be >= 0 and the exponent an integer in 1..65536 (got %g, %g)", k, ks, m);
    }
    try {
        c->planes.assign(planes_soa, planes_soa + (size_t)14 * (size_t)n_planes);
        if (phong) c->phong.assign(phong, phong + 2 * (size_t)n_elements); else c->phong.clear();
    } catch (const std::bad_alloc&) {
        return fail(c, PTR_ERR_INVALID, "out of host memory");
    }
    c->n_planes = n_planes;
    c->n_phong = phong ? n_elements : 0;
    c->dirty = true;
    return PTR_OK;
}

int ptr_render_device(ptr_ctx* c, const ptr_params* p, int first_frame, int n_frames,
                      void* d_rgba8, void* d_rgba32f, void* stream)
{
    if (!c) return PTR_ERR_INVALID;
    std::lock_guard<std::mutex> g(c->mu);
    PTR_TRY(check_params(c, p, first_frame, n_frames));
    if (!d_rgba8) return fail(c, PTR_ERR_INVALID, "d_rgba8 is NULL");
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : c->stream;
    return enqueue_render(c, p, first_frame, n_frames, static_cast<uint32_t*>(d_rgba8),
                          static_cast<float*>(d_rgba32f), s);
}

int ptr_render_device_noalloc(ptr_ctx* c, const ptr_params* p, int first_frame, int n_frames,
                              void* d_rgba8, void* d_rgba32f, void* stream)
{
    if (!c) return PTR_ERR_INVALID;
    std::lock_guard<std::mutex> g(c->mu);
    PTR_TRY(check_params(c, p, first_frame, n_frames));
    if (!d_rgba8) return fail(c, PTR_ERR_INVALID, "d_rgba8 is NULL");
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : c->stream;
    return enqueue_render(c, p, first_frame, n_frames, static_cast<uint32_t*>(d_rgba8),
                          static_cast<float*>(d_rgba32f), s, nullptr, 0, /*fixed_ws=*/true);
}

int ptr_strip_rows(const ptr_strips* s, int height)
{
    if (!s || s->n_parts < 1 || s->part < 0 || s->part >= s->n_parts || s->strip_height < 1 || height < 0)
        return PTR_ERR_INVALID;
    const int sh = s->strip_height;
    const int n_strips = (height + sh - 1) / sh;
    int rows = 0;
    for (int k = s->part; k < n_strips; k += s->n_parts) {
        const int y0 = k * sh, y1 = y0 + sh < height ? y0 + sh : height;
        rows += y1 - y0;
    }
    return rows;
}

int ptr_strip_image_row(const ptr_strips* s, int local_row)
{
    if (!s || s->n_parts < 1 || s->part < 0 || s->part >= s->n_parts || s->strip_height < 1 || local_row < 0)
        return PTR_ERR_INVALID;
    const int ls = local_row / s->strip_height;
    const long long y = ((long long)ls * s->n_parts + s->part) * s->strip_height + (local_row - ls * s->strip_height);
    return y > 0x7fffffffLL ? PTR_ERR_INVALID : (int)y;
}

int ptr_render_strips_device(ptr_ctx* c, const ptr_params* p, const ptr_strips* st,
                             int first_frame, int n_frames, void* d_rgba8, void* d_rgba32f,
                             uint64_t frame_stride_px, void* stream)
{
    if (!c) return PTR_ERR_INVALID;
    std::lock_guard<std::mutex> g(c->mu);
    PTR_TRY(check_params(c, p, first_frame, n_frames));
    if (!d_rgba8) return fail(c, PTR_ERR_INVALID, "d_rgba8 is NULL");
    if (!st) return fail(c, PTR_ERR_INVALID, "strips is NULL");
    const int rows = ptr_strip_rows(st, p->height);
    if (rows < 0) return fail(c, PTR_ERR_INVALID, "bad strip description");
    if (p->scan_height != rows || p->start_y != 0)
        return fail(c, PTR_ERR_INVALID, "strips: scan_height must be %d (got %d) and start_y 0 (got %d)",
                    rows, p->scan_height, p->start_y);
    if (p->flags & PTR_QUIRK_ROW_SHIFT)
        return fail(c, PTR_ERR_INVALID, "PTR_QUIRK_ROW_SHIFT cannot be combined with strips");
    if (frame_stride_px && frame_stride_px < (uint64_t)p->width * (uint64_t)rows)
        return fail(c, PTR_ERR_INVALID, "frame_stride_px smaller than one frame");
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : c->stream;
    return enqueue_render(c, p, first_frame, n_frames, static_cast<uint32_t*>(d_rgba8),
                          static_cast<float*>(d_rgba32f), s, st, frame_stride_px);
}

int ptr_sync(ptr_ctx* c)
{
    if (!c) return PTR_ERR_INVALID;
    std::lock_guard<std::mutex> g(c->mu);
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->have_scene) PTR_TRY(ensure_uploaded(c, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return PTR_OK;
}

int ptr_render(ptr_ctx* c, const ptr_params* p, uint32_t* out_rgba8, float* out_rgba32f)
{
    if (!c) return PTR_ERR_INVALID;
    std::lock_guard<std::mutex> g(c->mu);
    PTR_TRY(check_params(c, p, 0, 1));
    if (!out_rgba8) return fail(c, PTR_ERR_INVALID, "out_rgba8 is NULL");
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t px = (size_t)p->width * (size_t)p->scan_height;
    PTR_TRY(grow(c, c->out8, px * 4));
    if (out_rgba32f) PTR_TRY(grow(c, c->out32f, px * 16));
    PTR_TRY(enqueue_render(c, p, 0, 1, static_cast<uint32_t*>(c->out8.ptr),
                           out_rgba32f ? static_cast<float*>(c->out32f.ptr) : nullptr, c->stream));
    HIP_TRY(c, hipMemcpyAsync(out_rgba8, c->out8.ptr, px * 4, hipMemcpyDeviceToHost, c->stream));
    if (out_rgba32f)
        HIP_TRY(c, hipMemcpyAsync(out_rgba32f, c->out32f.ptr, px * 16, hipMemcpyDeviceToHost, c->stream));
    hipError_t e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) {
        c->words_dirty = true;
        return fail(c, PTR_ERR_HIP, "render failed: %s", hipGetErrorString(e));
    }
    return PTR_OK;
}

// shared body of the batched single-ray entries; org = origin of every ray when origins == NULL;
// rc = the render cache's output convention (ptr_trace_samples_f64).  Caller holds c->mu.
static int trace_batch_locked(ptr_ctx* c, const double* origins, const double org[3], const double* dirs, int n,
                              double* out_rgb, int32_t* out_element, double* out_hit, double* out_normal, bool rc)
{
    if (!c->have_scene) return fail(c, PTR_ERR_NO_SCENE, "no scene uploaded (call ptr_set_scene* first)");
    if (c->n_planes > 0 || c->n_phong > 0)
        return fail(c, PTR_ERR_UNSUPPORTED, "the single-ray entries have the reference's semantics (spheres, Lambert): clear the planes / "
                                            "Phong terms (ptr_set_extension_f64(ctx, NULL, 0, NULL, 0)) first");
    if (n < 0) return fail(c, PTR_ERR_INVALID, "n < 0");
    if (n == 0) return PTR_OK;
    if (!dirs || !out_rgb) return fail(c, PTR_ERR_INVALID, "dirs_xyz / out_rgb64 is NULL");
    HIP_TRY(c, hipSetDevice(c->device));
    PTR_TRY(ensure_uploaded(c, c->stream));
    const size_t v3 = (size_t)n * 3 * sizeof(double);
    PTR_TRY(grow(c, c->tr_in_d, v3));
    PTR_TRY(grow(c, c->tr_rgb, v3));
    if (origins) PTR_TRY(grow(c, c->tr_in_o, v3));
    if (out_element) PTR_TRY(grow(c, c->tr_el, (size_t)n * sizeof(int32_t)));
    if (out_hit) PTR_TRY(grow(c, c->tr_hit, v3));
    if (out_normal) PTR_TRY(grow(c, c->tr_nrm, v3));
    hipStream_t s = c->stream;
    HIP_TRY(c, hipMemcpyAsync(c->tr_in_d.ptr, dirs, v3, hipMemcpyHostToDevice, s));
    if (origins) HIP_TRY(c, hipMemcpyAsync(c->tr_in_o.ptr, origins, v3, hipMemcpyHostToDevice, s));
    TraceArgs a;
    std::memset(&a, 0, sizeof a);
    a.scene = static_cast<const unsigned char*>(c->scenes.ptr);
    a.origins = origins ? static_cast<const double*>(c->tr_in_o.ptr) : nullptr;
    a.dirs = static_cast<const double*>(c->tr_in_d.ptr);
    a.n = n;
    for (int k = 0; k < 3; k++) a.org[k] = org[k];
    a.rc_mode = rc ? 1 : 0;
    a.out_rgb = static_cast<double*>(c->tr_rgb.ptr);
    a.out_element = out_element ? static_cast<int32_t*>(c->tr_el.ptr) : nullptr;
    a.out_hit = out_hit ? static_cast<double*>(c->tr_hit.ptr) : nullptr;
    a.out_normal = out_normal ? static_cast<double*>(c->tr_nrm.ptr) : nullptr;
    HIP_TRY(c, launch_trace_batch_exact(a, s));
    HIP_TRY(c, hipMemcpyAsync(out_rgb, c->tr_rgb.ptr, v3, hipMemcpyDeviceToHost, s));
    if (out_element) HIP_TRY(c, hipMemcpyAsync(out_element, c->tr_el.ptr, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    if (out_hit) HIP_TRY(c, hipMemcpyAsync(out_hit, c->tr_hit.ptr, v3, hipMemcpyDeviceToHost, s));
    if (out_normal) HIP_TRY(c, hipMemcpyAsync(out_normal, c->tr_nrm.ptr, v3, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    return PTR_OK;
}

int ptr_trace_batch_f64(ptr_ctx* c, const double* origins, const double* dirs, int n,
                        double* out_rgb, int32_t* out_element, double* out_hit, double* out_normal)
{
    if (!c) return PTR_ERR_INVALID;
    std::lock_guard<std::mutex> g(c->mu);
    return trace_batch_locked(c, origins, c->cam.origin, dirs, n, out_rgb, out_element, out_hit, out_normal, false);
}

int ptr_trace_samples_f64(ptr_ctx* c, const double origin_xyz[3], const double* dirs_xyz, int n,
                          double* out_rgb64, uint8_t* out_rgb8, double* out_hit_xyz, double* out_normal_xyz)
{
    if (!c) return PTR_ERR_INVALID;
    std::lock_guard<std::mutex> g(c->mu);
    if (!origin_xyz) return fail(c, PTR_ERR_INVALID, "origin_xyz is NULL");
    if (n > 0 && !out_rgb64 && !out_rgb8) return fail(c, PTR_ERR_INVALID, "out_rgb64 and out_rgb8 are both NULL");
    std::vector<double> tmp;
    double* rgb = out_rgb64;
    if (!rgb && n > 0) {
        try { tmp.resize((size_t)n * 3); } catch (const std::bad_alloc&) { return fail(c, PTR_ERR_INVALID, "out of host memory"); }
        rgb = tmp.data();
    }
    PTR_TRY(trace_batch_locked(c, nullptr, origin_xyz, dirs_xyz, n, rgb, nullptr, out_hit_xyz, out_normal_xyz, true));
    if (out_rgb8)                                   // Sample.js:57-65: Math.round(Math.min(1, c) * 255)
        for (size_t k = 0; k < (size_t)n * 3; k++)
            out_rgb8[k] = (uint8_t)(js_to_uint32(js_round(js_min1(rgb[k]) * 255.0)) & 0xffu);
    return PTR_OK;
}

int ptr_trace_batch(ptr_ctx* c, const float* dirs_xyz, int n,
                    uint8_t* out_rgb, float* out_hit_xyz, float* out_normal_xyz)
{
    if (!c) return PTR_ERR_INVALID;
    if (n < 0 || (n > 0 && (!dirs_xyz || !out_rgb))) {
        std::lock_guard<std::mutex> g(c->mu);
        return fail(c, PTR_ERR_INVALID, "bad arguments to ptr_trace_batch");
    }
    if (n == 0) return PTR_OK;
    std::vector<double> d, rgb, hit, nrm;
    try {
        d.resize((size_t)n * 3); rgb.resize((size_t)n * 3);
        if (out_hit_xyz) hit.resize((size_t)n * 3);
        if (out_normal_xyz) nrm.resize((size_t)n * 3);
    } catch (const std::bad_alloc&) {
        std::lock_guard<std::mutex> g(c->mu);
        return fail(c, PTR_ERR_INVALID, "out of host memory in ptr_trace_batch");
    }
    for (size_t k = 0; k < d.size(); k++) d[k] = (double)dirs_xyz[k];
    PTR_TRY(ptr_trace_batch_f64(c, nullptr, d.data(), n, rgb.data(), nullptr,
                                out_hit_xyz ? hit.data() : nullptr, out_normal_xyz ? nrm.data() : nullptr));
    for (size_t k = 0; k < rgb.size(); k++)     // RayTracer.js:105-112
        out_rgb[k] = (uint8_t)(js_to_uint32(js_round(js_min1(rgb[k]) * 255.0)) & 0xffu);
    for (size_t k = 0; k < hit.size(); k++) out_hit_xyz[k] = (float)hit[k];
    for (size_t k = 0; k < nrm.size(); k++) out_normal_xyz[k] = (float)nrm[k];
    return PTR_OK;
}

namespace {
int rc_check(ptr_ctx* c, const ptr_rc_camera* cam, int n_items, const void* hit, const void* age, const void* item_pixel,
             const void* pel, const void* pdep)
{
    if (!cam) return fail(c, PTR_ERR_INVALID, "camera is NULL");
    if (cam->width < 1 || cam->height < 1 || (long long)(cam->width + 2) * (cam->height + 2) >= (1ll << 31))
        return fail(c, PTR_ERR_INVALID, "bad frame size %d x %d", cam->width, cam->height);
    if (n_items < 0) return fail(c, PTR_ERR_INVALID, "n_items < 0");
    if (!pel || !pdep || (n_items > 0 && (!hit || !age || !item_pixel)))
        return fail(c, PTR_ERR_INVALID, "a required buffer is NULL");
    return PTR_OK;
}

// enqueue the pass on stream s; all data pointers are device memory.  Caller holds c->mu.
int rc_enqueue(ptr_ctx* c, const ptr_rc_camera* cam, int n_items, const double* d_hit, double* d_age, double age_factor,
               double depth_threshold, int32_t* d_item_pixel, int32_t* d_pel, double* d_pdep, hipStream_t s, int item_base = 0,
               uint8_t* d_pixel_sample = nullptr)      // (the frame loop's sample flags: cleared with the grid's interior)
{
    const size_t n = (size_t)(n_items > 0 ? n_items : 1);
    PTR_TRY(grow(c, c->rc_depth, n * sizeof(double)));
    PTR_TRY(grow(c, c->rc_keys, 4 * n * sizeof(uint32_t)));
    const size_t temp = rc_sort_temp_bytes(n_items > 0 ? n_items : 1, cam->width, cam->height);
    PTR_TRY(grow(c, c->rc_temp, temp > 0 ? temp : 16));
    RcArgs a;
    std::memset(&a, 0, sizeof a);
    std::memcpy(a.from, cam->from, sizeof a.from);
    std::memcpy(a.view, cam->view, sizeof a.view);
    std::memcpy(a.proj, cam->proj, sizeof a.proj);
    a.width = cam->width; a.height = cam->height; a.n_items = n_items; a.item_base = item_base;
    a.age_factor = age_factor; a.depth_threshold = depth_threshold;
    a.hit_xyz = d_hit; a.age = d_age; a.item_pixel = d_item_pixel; a.pixel_element = d_pel; a.pixel_depth = d_pdep;
    a.pixel_sample = d_pixel_sample;
    a.item_depth = static_cast<double*>(c->rc_depth.ptr);
    uint32_t* k = static_cast<uint32_t*>(c->rc_keys.ptr);
    a.sort_key = k; a.sort_val = k + n; a.sorted_key = k + 2 * n; a.sorted_val = k + 3 * n;
    HIP_TRY(c, launch_rc_reproject(a, c->rc_temp.ptr, c->rc_temp.bytes, s));
    return PTR_OK;
}
}  // namespace

namespace {
int rc_shade_check(ptr_ctx* c, const ptr_rc_shade_params* p, int n_items, const void* pel, const void* pdep, const void* age,
                   const void* rgb, const void* prgb, const void* pprio, const void* hist, const void* totals)
{
    if (!p) return fail(c, PTR_ERR_INVALID, "params is NULL");
    if (p->width < 1 || p->height < 1 || (long long)(p->width + 2) * (p->height + 2) >= (1ll << 31))
        return fail(c, PTR_ERR_INVALID, "bad frame size %d x %d", p->width, p->height);
    if (n_items < 0) return fail(c, PTR_ERR_INVALID, "n_items < 0");
    if (!pel || !pdep || !prgb || !pprio || !hist || !totals || (n_items > 0 && (!age || !rgb)))
        return fail(c, PTR_ERR_INVALID, "a required buffer is NULL");
    return PTR_OK;
}

RcShadeArgs rc_shade_args(const ptr_rc_shade_params* p, int n_items)
{
    RcShadeArgs a;
    std::memset(&a, 0, sizeof a);
    a.width = p->width; a.height = p->height; a.n_items = n_items;
    a.age_factor = p->age_factor; a.interpolation_zero = p->interpolation_zero;
    a.interpolation_random = p->interpolation_random; a.priority_max = p->priority_max;
    return a;
}
}  // namespace

int ptr_rc_shade_device(ptr_ctx* c, const ptr_rc_shade_params* p, int n_items, const void* d_pixel_element,
                        const void* d_pixel_depth, void* d_item_age, const void* d_item_rgb, void* d_pixel_rgb,
                        void* d_pixel_priority, void* d_hist, void* d_totals, void* d_color_frame,
                        void* d_reprojection_frame, void* d_priority_frame, void* stream)
{
    if (!c) return PTR_ERR_INVALID;
    std::lock_guard<std::mutex> g(c->mu);
    PTR_TRY(rc_shade_check(c, p, n_items, d_pixel_element, d_pixel_depth, d_item_age, d_item_rgb, d_pixel_rgb, d_pixel_priority,
                           d_hist, d_totals));
    HIP_TRY(c, hipSetDevice(c->device));
    RcShadeArgs a = rc_shade_args(p, n_items);
    a.pixel_element = static_cast<const int32_t*>(d_pixel_element); a.pixel_depth = static_cast<const double*>(d_pixel_depth);
    a.item_age = static_cast<double*>(d_item_age); a.item_rgb = static_cast<const uint32_t*>(d_item_rgb);
    a.pixel_rgb = static_cast<uint32_t*>(d_pixel_rgb); a.pixel_priority = static_cast<int32_t*>(d_pixel_priority);
    a.hist = static_cast<uint32_t*>(d_hist); a.totals = static_cast<long long*>(d_totals);
    a.color_frame = static_cast<uint32_t*>(d_color_frame); a.reprojection_frame = static_cast<uint32_t*>(d_reprojection_frame);
    a.priority_frame = static_cast<uint32_t*>(d_priority_frame);
    PTR_TRY(grow(c, c->rc_page, (size_t)(p->width + 2) * (size_t)(p->height + 2) * sizeof(double)));
    a.pixel_age = static_cast<double*>(c->rc_page.ptr);
    HIP_TRY(c, launch_rc_shade(a, stream ? static_cast<hipStream_t>(stream) : c->stream));
    return PTR_OK;
}

int ptr_rc_shade(ptr_ctx* c, const ptr_rc_shade_params* p, int n_items, const int32_t* pixel_element, const double* pixel_depth,
                 double* item_age, const uint32_t* item_rgb, uint32_t* pixel_rgb, int32_t* pixel_priority, uint32_t* hist,
                 double stats[3], uint32_t* color_frame, uint32_t* reprojection_frame, uint32_t* priority_frame)
{
    if (!c) return PTR_ERR_INVALID;
    std::lock_guard<std::mutex> g(c->mu);
    PTR_TRY(rc_shade_check(c, p, n_items, pixel_element, pixel_depth, item_age, item_rgb, pixel_rgb, pixel_priority, hist, stats));
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t n = (size_t)n_items, cells = (size_t)(p->width + 2) * (size_t)(p->height + 2), px = (size_t)p->width * (size_t)p->height;
    PTR_TRY(grow(c, c->rc_pel, cells * sizeof(int32_t)));
    PTR_TRY(grow(c, c->rc_pdep, cells * sizeof(double)));
    PTR_TRY(grow(c, c->rc_age, (n ? n : 1) * sizeof(double)));
    PTR_TRY(grow(c, c->rc_rgb, (n ? n : 1) * sizeof(uint32_t)));
    PTR_TRY(grow(c, c->rc_prgb, cells * sizeof(uint32_t)));
    PTR_TRY(grow(c, c->rc_pprio, cells * sizeof(int32_t)));
    PTR_TRY(grow(c, c->rc_hist, 3 * 256 * sizeof(uint32_t) + 3 * sizeof(long long)));
    PTR_TRY(grow(c, c->rc_frames, 3 * px * sizeof(uint32_t)));
    hipStream_t s = c->stream;
    HIP_TRY(c, hipMemcpyAsync(c->rc_pel.ptr, pixel_element, cells * sizeof(int32_t), hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipMemcpyAsync(c->rc_pdep.ptr, pixel_depth, cells * sizeof(double), hipMemcpyHostToDevice, s));
    // grid borders are not written by the pass: they go up and come back as the caller has them
    HIP_TRY(c, hipMemcpyAsync(c->rc_prgb.ptr, pixel_rgb, cells * sizeof(uint32_t), hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipMemcpyAsync(c->rc_pprio.ptr, pixel_priority, cells * sizeof(int32_t), hipMemcpyHostToDevice, s));
    if (n) {
        HIP_TRY(c, hipMemcpyAsync(c->rc_age.ptr, item_age, n * sizeof(double), hipMemcpyHostToDevice, s));
        HIP_TRY(c, hipMemcpyAsync(c->rc_rgb.ptr, item_rgb, n * sizeof(uint32_t), hipMemcpyHostToDevice, s));
    }
    RcShadeArgs a = rc_shade_args(p, n_items);
    a.pixel_element = static_cast<const int32_t*>(c->rc_pel.ptr); a.pixel_depth = static_cast<const double*>(c->rc_pdep.ptr);
    a.item_age = static_cast<double*>(c->rc_age.ptr); a.item_rgb = static_cast<const uint32_t*>(c->rc_rgb.ptr);
    a.pixel_rgb = static_cast<uint32_t*>(c->rc_prgb.ptr); a.pixel_priority = static_cast<int32_t*>(c->rc_pprio.ptr);
    a.hist = static_cast<uint32_t*>(c->rc_hist.ptr);
    a.totals = reinterpret_cast<long long*>(static_cast<unsigned char*>(c->rc_hist.ptr) + 3 * 256 * sizeof(uint32_t));
    uint32_t* fr = static_cast<uint32_t*>(c->rc_frames.ptr);
    a.color_frame = color_frame ? fr : nullptr;
    a.reprojection_frame = reprojection_frame ? fr + px : nullptr;
    a.priority_frame = priority_frame ? fr + 2 * px : nullptr;
    PTR_TRY(grow(c, c->rc_page, cells * sizeof(double)));
    a.pixel_age = static_cast<double*>(c->rc_page.ptr);
    HIP_TRY(c, launch_rc_shade(a, s));
    long long totals[3];
    if (n) HIP_TRY(c, hipMemcpyAsync(item_age, c->rc_age.ptr, n * sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipMemcpyAsync(pixel_rgb, c->rc_prgb.ptr, cells * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipMemcpyAsync(pixel_priority, c->rc_pprio.ptr, cells * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipMemcpyAsync(hist, a.hist, 3 * 256 * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipMemcpyAsync(totals, a.totals, sizeof totals, hipMemcpyDeviceToHost, s));
    if (color_frame) HIP_TRY(c, hipMemcpyAsync(color_frame, fr, px * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    if (reprojection_frame) HIP_TRY(c, hipMemcpyAsync(reprojection_frame, fr + px, px * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    if (priority_frame) HIP_TRY(c, hipMemcpyAsync(priority_frame, fr + 2 * px, px * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    stats[0] = 100.0 * ((double)totals[0] / ((double)p->width * (double)p->height));     // Driver.js:676
    stats[1] = (double)totals[1];
    stats[2] = (double)totals[2];
    return PTR_OK;
}

int ptr_rc_reproject_device(ptr_ctx* c, const ptr_rc_camera* cam, int n_items, const void* d_hit_xyz, void* d_age,
                            double age_factor, double depth_threshold, void* d_item_pixel, void* d_pixel_element,
                            void* d_pixel_depth, void* stream)
{
    if (!c) return PTR_ERR_INVALID;
    std::lock_guard<std::mutex> g(c->mu);
    PTR_TRY(rc_check(c, cam, n_items, d_hit_xyz, d_age, d_item_pixel, d_pixel_element, d_pixel_depth));
    HIP_TRY(c, hipSetDevice(c->device));
    return rc_enqueue(c, cam, n_items, static_cast<const double*>(d_hit_xyz), static_cast<double*>(d_age), age_factor,
                      depth_threshold, static_cast<int32_t*>(d_item_pixel), static_cast<int32_t*>(d_pixel_element),
                      static_cast<double*>(d_pixel_depth), stream ? static_cast<hipStream_t>(stream) : c->stream);
}

int ptr_rc_reproject(ptr_ctx* c, const ptr_rc_camera* cam, int n_items, const double* hit_xyz, double* age,
                     double age_factor, double depth_threshold, int32_t* item_pixel, int32_t* pixel_element, double* pixel_depth)
{
    if (!c) return PTR_ERR_INVALID;
    std::lock_guard<std::mutex> g(c->mu);
    PTR_TRY(rc_check(c, cam, n_items, hit_xyz, age, item_pixel, pixel_element, pixel_depth));
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t n = (size_t)n_items, cells = (size_t)(cam->width + 2) * (size_t)(cam->height + 2);
    PTR_TRY(grow(c, c->rc_hit, (n ? n : 1) * 3 * sizeof(double)));
    PTR_TRY(grow(c, c->rc_age, (n ? n : 1) * sizeof(double)));
    PTR_TRY(grow(c, c->rc_item, (n ? n : 1) * sizeof(int32_t)));
    PTR_TRY(grow(c, c->rc_pel, cells * sizeof(int32_t)));
    PTR_TRY(grow(c, c->rc_pdep, cells * sizeof(double)));
    hipStream_t s = c->stream;
    if (n) {
        HIP_TRY(c, hipMemcpyAsync(c->rc_hit.ptr, hit_xyz, n * 3 * sizeof(double), hipMemcpyHostToDevice, s));
        HIP_TRY(c, hipMemcpyAsync(c->rc_age.ptr, age, n * sizeof(double), hipMemcpyHostToDevice, s));
    }
    // the border of the padded grid is not touched by the pass: it goes up and comes back as the caller has it
    HIP_TRY(c, hipMemcpyAsync(c->rc_pel.ptr, pixel_element, cells * sizeof(int32_t), hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipMemcpyAsync(c->rc_pdep.ptr, pixel_depth, cells * sizeof(double), hipMemcpyHostToDevice, s));
    PTR_TRY(rc_enqueue(c, cam, n_items, static_cast<const double*>(c->rc_hit.ptr), static_cast<double*>(c->rc_age.ptr),
                       age_factor, depth_threshold, static_cast<int32_t*>(c->rc_item.ptr),
                       static_cast<int32_t*>(c->rc_pel.ptr), static_cast<double*>(c->rc_pdep.ptr), s));
    if (n) {
        HIP_TRY(c, hipMemcpyAsync(age, c->rc_age.ptr, n * sizeof(double), hipMemcpyDeviceToHost, s));
        HIP_TRY(c, hipMemcpyAsync(item_pixel, c->rc_item.ptr, n * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    }
    HIP_TRY(c, hipMemcpyAsync(pixel_element, c->rc_pel.ptr, cells * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipMemcpyAsync(pixel_depth, c->rc_pdep.ptr, cells * sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    return PTR_OK;
}

// ---- render cache: the frame loop on a resident cache ------------------------------------------------------------
namespace {
int rc_cache_ready(ptr_ctx* c, bool need_state)
{
    if (!c->rcc) return fail(c, PTR_ERR_INVALID, "no render cache on this context (ptr_rc_cache_create)");
    if (need_state && !c->rcc->loaded) return fail(c, PTR_ERR_INVALID, "the render cache has no state yet (ptr_rc_cache_upload)");
    return PTR_OK;
}

// Driver.directSamples' pick loop (Driver.js:795-822) as pieces of RcPickSegments.  n_cand >= 1, rf > 0.  Returns false when
// the table would overflow (the caller then adds the sum up step by step).  *n_picks = how many candidates are requested.
bool rc_pick_segments(double rf, int n_cand, RcPickSegments* g, long long* n_picks)
{
    g->n = 0; g->pad = 0;
    auto push = [&](long long k, long long S, long long D, int e) {
        if (g->n >= 96 || k > 0x7fffffffll) return false;
        g->k0[g->n] = (int32_t)k; g->S0[g->n] = S; g->D[g->n] = D; g->e[g->n] = e;
        g->n++;
        return true;
    };
    if (rf <= 1) {                  // the rounded sum moves by 0 or 1 per step: every candidate is requested, once
        *n_picks = n_cand;
        return push(0, 0, 1, 0);
    }
    const double C = (double)n_cand;
    if (!push(0, 0, 0, 0)) return false;                // step 0: the sum is 0
    long long k = 1;
    double s = rf, last = 0;                            // 0 + rf is exact
    while (s < C) {
        int ex;
        (void)std::frexp(s, &ex);                       // 2^(ex-1) <= s < 2^ex: one unit in the last place is 2^(ex-53)
        const int e = ex - 53;
        const double top = std::ldexp(1.0, ex);
        const long long S = (long long)std::ldexp(s, -e);
        const double s1 = s + rf, s2 = s1 + rf;
        long long D = 0, cnt = 1;
        // a constant step while the sum stays below `top` (a tie of the rounding may take one step to settle: checked on two)
        if (s1 < top && (s2 >= top || std::ldexp(s2 - s1, -e) == std::ldexp(s1 - s, -e))) {
            D = (long long)std::ldexp(s1 - s, -e);
            const long long L = (long long)std::ceil(std::ldexp(top < C ? top : C, -e));
            cnt = (L - S + D - 1) / D;
        }
        if (!push(k, S, D, e)) return false;
        k += cnt;
        last = std::ldexp((double)(S + (cnt - 1) * D), e);
        s = last + rf;
    }
    *n_picks = js_round(last) < C ? k : k - 1;          // `if (i < candidates.length)` can only fail for the last step
    return true;
}

int32_t rc_pick_at(const RcPickSegments& g, long long k)       // what rc_pick_index_kernel computes for step k
{
    int seg = 0;
    while (seg + 1 < g.n && k >= g.k0[seg + 1]) seg++;
    return (int32_t)js_round(std::ldexp((double)(g.S0[seg] + (k - g.k0[seg]) * g.D[seg]), g.e[seg]));
}

RcLoopArgs rc_loop_args(const RcCache& r)
{
    RcLoopArgs a;
    std::memset(&a, 0, sizeof a);
    a.width = r.width; a.height = r.height; a.n_items = r.n_items;
    a.pages = r.n_items / 8;
    a.item_pixel = static_cast<int32_t*>(r.item_pixel.ptr); a.item_hit = static_cast<double*>(r.item_hit.ptr);
    a.item_age = static_cast<double*>(r.item_age.ptr); a.item_rgb = static_cast<uint32_t*>(r.item_rgb.ptr);
    a.pixel_element = static_cast<int32_t*>(r.pixel_element.ptr); a.pixel_rgb = static_cast<uint32_t*>(r.pixel_rgb.ptr);
    a.pixel_priority = static_cast<int32_t*>(r.pixel_priority.ptr); a.pixel_sample = static_cast<uint8_t*>(r.pixel_sample.ptr);
    unsigned char* w = static_cast<unsigned char*>(r.words.ptr);
    a.hist = reinterpret_cast<const uint32_t*>(w + RC_WORDS_HIST);
    a.words = reinterpret_cast<RcLoopWords*>(w + RC_WORDS_LOOP);
    a.cand = static_cast<int32_t*>(r.cand.ptr); a.pick_index = static_cast<int32_t*>(r.picks.ptr);
    a.pick_item = static_cast<int32_t*>(r.pick_item.ptr); a.need = static_cast<int32_t*>(r.need.ptr); a.rank = static_cast<int32_t*>(r.rank.ptr);
    a.dirs = static_cast<double*>(r.dirs.ptr); a.tr_rgb = static_cast<const double*>(r.tr_rgb.ptr); a.tr_hit = static_cast<const double*>(r.tr_hit.ptr);
    return a;
}
}  // namespace

int ptr_rc_cache_create(ptr_ctx* c, int width, int height, int n_items)
{
    if (!c) return PTR_ERR_INVALID;
    std::lock_guard<std::mutex> g(c->mu);
    if (width < 1 || height < 1 || (long long)(width + 2) * (height + 2) >= (1ll << 31))
        return fail(c, PTR_ERR_INVALID, "bad frame size %d x %d", width, height);
    if (n_items < 8) return fail(c, PTR_ERR_INVALID, "a render cache needs at least one page of 8 items (n_items = %d)", n_items);
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    rc_cache_free(c);
    RcCache* r = new (std::nothrow) RcCache();
    if (!r) return fail(c, PTR_ERR_INVALID, "out of host memory");
    c->rcc = r;
    r->width = width; r->height = height; r->n_items = n_items;
    const size_t n = (size_t)n_items, cells = (size_t)(width + 2) * (size_t)(height + 2), px = (size_t)width * (size_t)height;
    const size_t temp = rc_loop_temp_bytes(width, height);
    int rc = PTR_OK;
    const struct { DeviceBuffer* b; size_t bytes; } want[] = {
        {&r->item_pixel, n * 4}, {&r->item_hit, n * 24}, {&r->item_age, n * 8}, {&r->item_rgb, n * 4},
        {&r->pixel_element, cells * 4}, {&r->pixel_depth, cells * 8}, {&r->pixel_rgb, cells * 4}, {&r->pixel_priority, cells * 4},
        {&r->pixel_sample, cells}, {&r->words, RC_WORDS_BYTES}, {&r->cand, px * 4}, {&r->picks, px * 4}, {&r->pick_item, px * 4},
        {&r->need, px * 4}, {&r->rank, px * 4}, {&r->dirs, px * 24}, {&r->tr_rgb, px * 24}, {&r->tr_hit, px * 24},
        {&r->frames, 4 * px * 4}, {&r->temp, temp > 0 ? temp : 16}};
    for (const auto& w : want)
        if (rc == PTR_OK) rc = grow(c, *w.b, w.bytes);
    if (rc == PTR_OK && hipHostMalloc(reinterpret_cast<void**>(&r->host_words), RC_WORDS_BYTES, hipHostMallocDefault) != hipSuccess)
        rc = fail(c, PTR_ERR_HIP, "hipHostMalloc failed for the render cache's words");
    if (rc == PTR_OK && hipHostMalloc(reinterpret_cast<void**>(&r->host_picks), px * 4, hipHostMallocDefault) != hipSuccess)
        rc = fail(c, PTR_ERR_HIP, "hipHostMalloc failed for the render cache's pick list");
    if (rc == PTR_OK && temp == 0) rc = fail(c, PTR_ERR_HIP, "hipCUB did not report its temporary storage size");
    if (rc != PTR_OK) { (void)hipGetLastError(); rc_cache_free(c); }
    return rc;
}

void ptr_rc_cache_destroy(ptr_ctx* c)
{
    if (!c) return;
    std::lock_guard<std::mutex> g(c->mu);
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    rc_cache_free(c);
}

int ptr_rc_cache_shape(ptr_ctx* c, int* width, int* height, int* n_items)
{
    if (!c) return PTR_ERR_INVALID;
    std::lock_guard<std::mutex> g(c->mu);
    PTR_TRY(rc_cache_ready(c, false));
    if (!width || !height || !n_items) return fail(c, PTR_ERR_INVALID, "output pointer is NULL");
    *width = c->rcc->width; *height = c->rcc->height; *n_items = c->rcc->n_items;
    return PTR_OK;
}

int ptr_rc_cache_upload(ptr_ctx* c, const int32_t* item_pixel, const double* item_hit, const double* item_age, const uint32_t* item_rgb,
                        const int32_t* pixel_element, const double* pixel_depth, const uint32_t* pixel_rgb, const int32_t* pixel_priority,
                        const uint8_t* pixel_sample, int64_t cache_pointer)
{
    if (!c) return PTR_ERR_INVALID;
    std::lock_guard<std::mutex> g(c->mu);
    PTR_TRY(rc_cache_ready(c, false));
    RcCache& r = *c->rcc;
    if (!item_pixel || !item_hit || !item_age || !item_rgb || !pixel_element || !pixel_depth || !pixel_rgb || !pixel_priority || !pixel_sample)
        return fail(c, PTR_ERR_INVALID, "a required buffer is NULL");
    if (cache_pointer < 0 || cache_pointer % 8 != 0 || cache_pointer > (int64_t)r.n_items)
        return fail(c, PTR_ERR_INVALID, "cache_pointer must be a multiple of 8 in 0..n_items");
    const size_t n = (size_t)r.n_items, cells = (size_t)(r.width + 2) * (size_t)(r.height + 2);
    for (size_t i = 0; i < n; i++)
        if (item_pixel[i] < -1 || (size_t)(item_pixel[i] + 1) > cells)
            return fail(c, PTR_ERR_INVALID, "item_pixel[%zu] = %d is not a pixel of the grid", i, item_pixel[i]);
    for (size_t p = 0; p < cells; p++)
        if (pixel_element[p] < -1 || (size_t)(pixel_element[p] + 1) > n)
            return fail(c, PTR_ERR_INVALID, "pixel_element[%zu] = %d is not a cache item", p, pixel_element[p]);
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    const struct { void* dst; const void* src; size_t bytes; } up[] = {
        {r.item_pixel.ptr, item_pixel, n * 4}, {r.item_hit.ptr, item_hit, n * 24}, {r.item_age.ptr, item_age, n * 8},
        {r.item_rgb.ptr, item_rgb, n * 4}, {r.pixel_element.ptr, pixel_element, cells * 4}, {r.pixel_depth.ptr, pixel_depth, cells * 8},
        {r.pixel_rgb.ptr, pixel_rgb, cells * 4}, {r.pixel_priority.ptr, pixel_priority, cells * 4}, {r.pixel_sample.ptr, pixel_sample, cells}};
    for (const auto& u : up) HIP_TRY(c, hipMemcpyAsync(u.dst, u.src, u.bytes, hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipMemsetAsync(r.words.ptr, 0, RC_WORDS_BYTES, s));
    const int32_t page_next = (int32_t)((cache_pointer / 8) % (int64_t)(r.n_items / 8));
    HIP_TRY(c, hipMemcpyAsync(static_cast<unsigned char*>(r.words.ptr) + RC_WORDS_LOOP + offsetof(RcLoopWords, page_next), &page_next,
                              sizeof page_next, hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    r.loaded = true;
    return PTR_OK;
}

int ptr_rc_cache_download(ptr_ctx* c, int32_t* item_pixel, double* item_hit, double* item_age, uint32_t* item_rgb, int32_t* pixel_element,
                          double* pixel_depth, uint32_t* pixel_rgb, int32_t* pixel_priority, uint8_t* pixel_sample, int64_t* cache_pointer)
{
    if (!c) return PTR_ERR_INVALID;
    std::lock_guard<std::mutex> g(c->mu);
    PTR_TRY(rc_cache_ready(c, true));
    RcCache& r = *c->rcc;
    if (!item_pixel || !item_hit || !item_age || !item_rgb || !pixel_element || !pixel_depth || !pixel_rgb || !pixel_priority ||
        !pixel_sample || !cache_pointer)
        return fail(c, PTR_ERR_INVALID, "a required buffer is NULL");
    const size_t n = (size_t)r.n_items, cells = (size_t)(r.width + 2) * (size_t)(r.height + 2);
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    const struct { void* dst; const void* src; size_t bytes; } down[] = {
        {item_pixel, r.item_pixel.ptr, n * 4}, {item_hit, r.item_hit.ptr, n * 24}, {item_age, r.item_age.ptr, n * 8},
        {item_rgb, r.item_rgb.ptr, n * 4}, {pixel_element, r.pixel_element.ptr, cells * 4}, {pixel_depth, r.pixel_depth.ptr, cells * 8},
        {pixel_rgb, r.pixel_rgb.ptr, cells * 4}, {pixel_priority, r.pixel_priority.ptr, cells * 4}, {pixel_sample, r.pixel_sample.ptr, cells}};
    for (const auto& d : down) HIP_TRY(c, hipMemcpyAsync(d.dst, d.src, d.bytes, hipMemcpyDeviceToHost, s));
    int32_t page_next = 0;
    HIP_TRY(c, hipMemcpyAsync(&page_next, static_cast<unsigned char*>(r.words.ptr) + RC_WORDS_LOOP + offsetof(RcLoopWords, page_next),
                              sizeof page_next, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    *cache_pointer = (int64_t)page_next * 8;        // (the pointer after the last page and 0 name the same next page)
    return PTR_OK;
}

int ptr_rc_next_frame(ptr_ctx* c, const ptr_rc_view* view, const ptr_rc_loop_params* par, ptr_rc_frame_stats* out)
{
    if (!c) return PTR_ERR_INVALID;
    std::lock_guard<std::mutex> g(c->mu);
    PTR_TRY(rc_cache_ready(c, true));
    if (!view || !par || !out) return fail(c, PTR_ERR_INVALID, "view / params / out is NULL");
    if (!c->have_scene) return fail(c, PTR_ERR_NO_SCENE, "no scene uploaded (call ptr_set_scene* first)");
    if (c->n_planes > 0 || c->n_phong > 0)
        return fail(c, PTR_ERR_UNSUPPORTED, "the render cache traces with the reference's semantics (spheres, Lambert): clear the planes / "
                                            "Phong terms (ptr_set_extension_f64(ctx, NULL, 0, NULL, 0)) first");
    if (!(par->priority_max >= 0 && par->priority_max <= 255) || par->priority_max != std::floor(par->priority_max))
        return fail(c, PTR_ERR_INVALID, "priority_max must be an integer in 0..255");
    if (!(par->max_samples_per_frame > 0)) return fail(c, PTR_ERR_INVALID, "max_samples_per_frame must be > 0");
    RcCache& r = *c->rcc;
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    PTR_TRY(order_after_last(c, s));
    PTR_TRY(ensure_uploaded(c, s));
    const int w = r.width, h = r.height, st = w + 2;

    // reprojectFrame (with detachFrameFromCache: the interior's sample flags)
    ptr_rc_camera cam;
    std::memcpy(cam.from, view->from, sizeof cam.from);
    std::memcpy(cam.view, view->world_inv, sizeof cam.view);
    std::memcpy(cam.proj, view->proj, sizeof cam.proj);
    cam.width = w; cam.height = h;
    PTR_TRY(rc_enqueue(c, &cam, r.n_items, static_cast<const double*>(r.item_hit.ptr), static_cast<double*>(r.item_age.ptr),
                       par->age_factor, par->depth_threshold, static_cast<int32_t*>(r.item_pixel.ptr),
                       static_cast<int32_t*>(r.pixel_element.ptr), static_cast<double*>(r.pixel_depth.ptr), s, 0,
                       static_cast<uint8_t*>(r.pixel_sample.ptr)));

    // depthCulling + fillGaps
    ptr_rc_shade_params sp = {w, h, par->age_factor, par->interpolation_zero, par->interpolation_random, par->priority_max};
    RcShadeArgs sa = rc_shade_args(&sp, r.n_items);
    unsigned char* words = static_cast<unsigned char*>(r.words.ptr);
    sa.pixel_element = static_cast<const int32_t*>(r.pixel_element.ptr); sa.pixel_depth = static_cast<const double*>(r.pixel_depth.ptr);
    sa.item_age = static_cast<double*>(r.item_age.ptr); sa.item_rgb = static_cast<const uint32_t*>(r.item_rgb.ptr);
    sa.pixel_rgb = static_cast<uint32_t*>(r.pixel_rgb.ptr); sa.pixel_priority = static_cast<int32_t*>(r.pixel_priority.ptr);
    sa.hist = reinterpret_cast<uint32_t*>(words + RC_WORDS_HIST); sa.totals = reinterpret_cast<long long*>(words + RC_WORDS_TOTALS);
    PTR_TRY(grow(c, c->rc_page, (size_t)st * (size_t)(h + 2) * sizeof(double)));
    sa.pixel_age = static_cast<double*>(c->rc_page.ptr);
    HIP_TRY(c, launch_rc_shade(sa, s));

    // directSamples: threshold and candidates on the GPU, the running sum of the picks here
    RcLoopArgs a = rc_loop_args(r);
    a.priority_max = (int)par->priority_max;
    a.age_factor = par->age_factor; a.max_samples = par->max_samples_per_frame;
    std::memcpy(a.from, view->from, sizeof a.from);
    std::memcpy(a.world, view->world, sizeof a.world);
    std::memcpy(a.proj_inv, view->proj_inv, sizeof a.proj_inv);
    HIP_TRY(c, launch_rc_candidates(a, r.temp.ptr, r.temp.bytes, s));
    HIP_TRY(c, hipMemcpyAsync(r.host_words, words, RC_WORDS_BYTES, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    RcLoopWords lw;
    std::memcpy(&lw, r.host_words + RC_WORDS_LOOP, sizeof lw);
    long long totals[3];
    std::memcpy(totals, r.host_words + RC_WORDS_TOTALS, sizeof totals);
    const double rf = lw.samples / par->max_samples_per_frame;          // :708
    const double n_cand = (double)lw.candidates;
    int n_picks = 0;
    RcPickSegments segs;
    bool closed_form = false;
    if (lw.candidates > 0) {
        if (!(rf > 0) || n_cand / rf > 4.0e9)
            return fail(c, PTR_ERR_INVALID, "request factor %g with %d candidates: the reference's pick loop (Driver.js:795-822) would "
                                            "not end", rf, lw.candidates);
        long long np = 0;
        closed_form = rc_pick_segments(rf, lw.candidates, &segs, &np);
        if (closed_form) {
            n_picks = (int)np;
        } else {                                                        // (more pieces than the table holds: add the sum up here)
            double index = 0, last_i = -1;                              // :795-822; lastI === null is the -1 here
            while (index < n_cand) {
                const double i = js_round(index);
                if (i != last_i && i < n_cand) r.host_picks[n_picks++] = (int32_t)i;
                last_i = i;
                index += rf;
            }
            segs.n = 0;
        }
    }
    a.n_picks = n_picks;
    if (n_picks > 0) {
        if (!closed_form) HIP_TRY(c, hipMemcpyAsync(r.picks.ptr, r.host_picks, (size_t)n_picks * 4, hipMemcpyHostToDevice, s));
        HIP_TRY(c, launch_rc_requests(a, segs, r.temp.ptr, r.temp.bytes, s));
        TraceArgs t;                                                    // requestSamples: Sample.doRaytracing for every request
        std::memset(&t, 0, sizeof t);
        t.scene = static_cast<const unsigned char*>(c->scenes.ptr);
        t.dirs = a.dirs; t.n = n_picks; t.rc_mode = 1;
        for (int k = 0; k < 3; k++) t.org[k] = view->from[k];
        t.out_rgb = static_cast<double*>(r.tr_rgb.ptr); t.out_hit = static_cast<double*>(r.tr_hit.ptr);
        HIP_TRY(c, launch_trace_batch_exact(t, s));
    }
    HIP_TRY(c, launch_rc_store(a, s));
    HIP_TRY(c, hipMemcpyAsync(r.host_words + RC_WORDS_LOOP, words + RC_WORDS_LOOP, sizeof(RcLoopWords), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    std::memcpy(&lw, r.host_words + RC_WORDS_LOOP, sizeof lw);
    out->threshold = lw.threshold < 0 ? std::nan("") : (double)lw.threshold;
    out->request_factor = rf;
    out->candidates = n_cand;
    out->requests = (double)n_picks;
    out->completeness = 100.0 * ((double)totals[0] / ((double)w * (double)h));        // :676
    out->total_priority = (double)totals[1];
    out->total_priority_increment = (double)totals[2];
    out->cache_usage = (double)lw.used / (double)r.n_items;
    return PTR_OK;
}

int ptr_rc_pick_sequence(double request_factor, int n_candidates, int32_t* out, int capacity)
{
    if (n_candidates < 0 || capacity < 0 || (capacity > 0 && !out)) return PTR_ERR_INVALID;
    if (n_candidates == 0) return 0;
    if (!(request_factor > 0) || (double)n_candidates / request_factor > 4.0e9) return PTR_ERR_INVALID;
    RcPickSegments g;
    long long n = 0;
    if (!rc_pick_segments(request_factor, n_candidates, &g, &n)) return PTR_ERR_UNSUPPORTED;
    if (n > capacity) return PTR_ERR_INVALID;
    for (long long k = 0; k < n; k++) out[k] = rc_pick_at(g, k);
    return (int)n;
}

int ptr_rc_get_frames(ptr_ctx* c, uint32_t* color_frame, uint32_t* reprojection_frame, uint32_t* priority_frame, uint32_t* sampling_frame)
{
    if (!c) return PTR_ERR_INVALID;
    std::lock_guard<std::mutex> g(c->mu);
    PTR_TRY(rc_cache_ready(c, true));
    RcCache& r = *c->rcc;
    uint32_t* host[4] = {color_frame, reprojection_frame, priority_frame, sampling_frame};
    if (!host[0] && !host[1] && !host[2] && !host[3]) return PTR_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    RcLoopArgs a = rc_loop_args(r);
    const size_t px = (size_t)r.width * (size_t)r.height;
    for (int k = 0; k < 4; k++) a.frames[k] = host[k] ? static_cast<uint32_t*>(r.frames.ptr) + (size_t)k * px : nullptr;
    HIP_TRY(c, launch_rc_frames(a, s));
    for (int k = 0; k < 4; k++)
        if (host[k]) HIP_TRY(c, hipMemcpyAsync(host[k], a.frames[k], px * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    return PTR_OK;
}

int ptr_get_kernel_ms(ptr_ctx* c, float* main_ms, float* retrace_ms)
{
    if (!c) return PTR_ERR_INVALID;
    std::lock_guard<std::mutex> g(c->mu);
    if (!main_ms || !retrace_ms) return fail(c, PTR_ERR_INVALID, "output pointer is NULL");
    *main_ms = 0.0f; *retrace_ms = 0.0f;
    if (c->timed_kernels < 1) return fail(c, PTR_ERR_INVALID, "no call with PTR_FLAG_TIMING yet");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipEventSynchronize(c->ev[c->timed_kernels]));
    HIP_TRY(c, hipEventElapsedTime(main_ms, c->ev[0], c->ev[1]));
    if (c->timed_kernels == 2) HIP_TRY(c, hipEventElapsedTime(retrace_ms, c->ev[1], c->ev[2]));
    return PTR_OK;
}

int ptr_get_stats(ptr_ctx* c, ptr_stats* out)
{
    if (!c) return PTR_ERR_INVALID;
    std::lock_guard<std::mutex> g(c->mu);
    if (!out) return fail(c, PTR_ERR_INVALID, "out is NULL");
    std::memset(out, 0, sizeof *out);
    if (!c->stats_words) return PTR_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    // the last call's kernels may have gone to a caller's stream without anybody waiting for them yet
    if (c->have_last) HIP_TRY(c, hipEventSynchronize(c->last_ev));
    unsigned long long w[9];
    HIP_TRY(c, hipMemcpy(w, c->stats_words, sizeof w, hipMemcpyDeviceToHost));
    out->primary = w[0]; out->hits = w[1]; out->shadow_rays = w[2];
    out->exit_tca = w[3]; out->exit_d2 = w[4]; out->exit_hit = w[5]; out->retraced = w[6]; out->reshaded = w[7]; out->secondary = w[8];
    return PTR_OK;
}


// ---- multi-GPU gather over RCCL ---------------------------------------------------------------------------------
extern "C++" {
namespace {

// The few RCCL entry points the gather needs, resolved at run time from librccl.so.1 (the process may already have one
// loaded - PyTorch brings its own - and then that instance is the one used; a host without RCCL can still render on one GPU).
// Types as in <rccl/rccl.h>: ncclComm_t is a pointer, ncclUniqueId 128 bytes passed by value, results and data types ints.
struct RcclId { char internal[PTR_COMM_ID_BYTES]; };
struct Rccl {
    int (*GetUniqueId)(RcclId*) = nullptr;
    int (*CommInitRank)(void**, int, RcclId, int) = nullptr;
    int (*CommInitAll)(void**, int, const int*) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*Send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool ok = false;
    char why[256] = "";
};
constexpr int RCCL_UINT32 = 3;      // ncclUint32 (rccl.h: ncclInt8 0, ncclUint8 1, ncclInt32 2, ncclUint32 3)
constexpr int RCCL_INT32 = 2, RCCL_FLOAT64 = 8;

Rccl& rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!h) { snprintf(r.why, sizeof r.why, "cannot load librccl.so.1: %s", dlerror()); return; }
        bool all = true;
        auto sym = [&](const char* name) { void* p = dlsym(h, name); if (!p) { all = false; snprintf(r.why, sizeof r.why, "librccl.so.1 lacks %s", name); } return p; };
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
        r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(sym("ncclCommInitAll"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.Send = reinterpret_cast<decltype(r.Send)>(sym("ncclSend"));
        r.Recv = reinterpret_cast<decltype(r.Recv)>(sym("ncclRecv"));
        r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
        r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
        r.ok = all;
    });
    return r;
}

#define RCCL_TRY(c, call)                                                                       \
    do {                                                                                        \
        int r_ = (call);                                                                        \
        if (r_ != 0) return fail((c), PTR_ERR_HIP, "%s failed: %s", #call, rccl().GetErrorString(r_)); \
    } while (0)

struct GatherShape { int rows_max, rows; uint64_t stride; size_t count; };

// sizes of one gather; fails on arguments that do not describe one.  part / n_parts = the context's rank / ranks.
int gather_shape(ptr_ctx* c, int width, int height, int strip_height, int n_frames, uint64_t frame_stride_px, int part, int n_parts,
                 GatherShape* g)
{
    if (width < 1 || height < 1 || strip_height < 1 || n_frames < 1 || n_frames > 65535 || width >= (1 << 24))
        return fail(c, PTR_ERR_INVALID, "gather: bad sizes (width %d, height %d, strip_height %d, n_frames %d)", width, height, strip_height, n_frames);
    const ptr_strips s0{n_parts, 0, strip_height, 0}, sp{n_parts, part, strip_height, 0};
    g->rows_max = ptr_strip_rows(&s0, height);
    g->rows = ptr_strip_rows(&sp, height);
    if (g->rows_max < 0 || g->rows < 0 || g->rows_max > 65535) return fail(c, PTR_ERR_INVALID, "gather: bad strip partition");
    g->stride = frame_stride_px ? frame_stride_px : (uint64_t)g->rows_max * (uint64_t)width;
    if (g->stride < (uint64_t)g->rows_max * (uint64_t)width)
        return fail(c, PTR_ERR_INVALID, "gather: frame_stride_px %llu is smaller than the largest part (%d rows of %d pixels)",
                    (unsigned long long)g->stride, g->rows_max, width);
    g->count = (size_t)g->stride * (size_t)n_frames;
    if ((uint64_t)width * (uint64_t)height * (uint64_t)n_frames >= (1ull << 40)) return fail(c, PTR_ERR_INVALID, "gather: too many pixels");
    return PTR_OK;
}

// the root's copy step for the rows of `part` that lie at src
int gather_place(ptr_ctx* root, const uint32_t* src, uint32_t* full, int width, int height, int strip_height, int n_frames,
                 const GatherShape& g, int part, int n_parts, hipStream_t s)
{
    GatherArgs a;
    a.src = src; a.full = full; a.src_frame_stride_px = g.stride;
    a.width = width; a.height = height; a.n_frames = n_frames;
    a.strip_h = strip_height; a.n_parts = n_parts; a.part = part;
    const ptr_strips sp{n_parts, part, strip_height, 0};
    a.rows = ptr_strip_rows(&sp, height);
    HIP_TRY(root, launch_gather_rows(a, s));
    return PTR_OK;
}

}  // namespace
}  // extern "C++"

int ptr_comm_unique_id(unsigned char id[PTR_COMM_ID_BYTES])
{
    if (!id) return fail(nullptr, PTR_ERR_INVALID, "id is NULL");
    Rccl& r = rccl();
    if (!r.ok) return fail(nullptr, PTR_ERR_UNSUPPORTED, "%s", r.why);
    RcclId u;
    RCCL_TRY(nullptr, r.GetUniqueId(&u));
    std::memcpy(id, u.internal, PTR_COMM_ID_BYTES);
    return PTR_OK;
}

int ptr_comm_init_rank(ptr_ctx* c, const unsigned char id[PTR_COMM_ID_BYTES], int rank, int n_ranks)
{
    if (!c) return PTR_ERR_INVALID;
    std::lock_guard<std::mutex> g(c->mu);
    if (n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(c, PTR_ERR_INVALID, "rank %d of %d", rank, n_ranks);
    if (c->comm_n) return fail(c, PTR_ERR_INVALID, "the context already belongs to a communicator (ptr_comm_destroy first)");
    if (n_ranks > 1) {
        if (!id) return fail(c, PTR_ERR_INVALID, "id is NULL");
        Rccl& r = rccl();
        if (!r.ok) return fail(c, PTR_ERR_UNSUPPORTED, "%s", r.why);
        HIP_TRY(c, hipSetDevice(c->device));
        RcclId u;
        std::memcpy(u.internal, id, PTR_COMM_ID_BYTES);
        RCCL_TRY(c, r.CommInitRank(&c->comm, n_ranks, u, rank));
    }
    c->comm_rank = rank; c->comm_n = n_ranks;
    return PTR_OK;
}

int ptr_comm_init_all(ptr_ctx* const* ctxs, int n)
{
    if (!ctxs || n < 1 || n > 64 || !ctxs[0]) return fail(nullptr, PTR_ERR_INVALID, "ptr_comm_init_all: bad context list");
    for (int i = 0; i < n; i++) {                      // (messages go to ctxs[0], the context a host asks afterwards)
        if (!ctxs[i]) return fail(ctxs[0], PTR_ERR_INVALID, "ptr_comm_init_all: context %d is NULL", i);
        if (ctxs[i]->comm_n) return fail(ctxs[0], PTR_ERR_INVALID, "context %d already belongs to a communicator", i);
        for (int j = 0; j < i; j++)
            if (ctxs[j]->device == ctxs[i]->device)
                return fail(ctxs[0], PTR_ERR_INVALID, "contexts %d and %d are on the same GPU (%d): one rank per GPU", j, i, ctxs[i]->device);
    }
    if (n > 1) {
        Rccl& r = rccl();
        if (!r.ok) return fail(ctxs[0], PTR_ERR_UNSUPPORTED, "%s", r.why);
        void* comms[64];
        int devs[64];
        for (int i = 0; i < n; i++) devs[i] = ctxs[i]->device;
        RCCL_TRY(ctxs[0], r.CommInitAll(comms, n, devs));
        for (int i = 0; i < n; i++) ctxs[i]->comm = comms[i];
    }
    for (int i = 0; i < n; i++) { ctxs[i]->comm_rank = i; ctxs[i]->comm_n = n; }
    return PTR_OK;
}

int ptr_comm_destroy(ptr_ctx* c)
{
    if (!c) return PTR_ERR_INVALID;
    if (c->comm) {
        (void)hipSetDevice(c->device);
        (void)rccl().CommDestroy(c->comm);
        c->comm = nullptr;
    }
    c->comm_n = 0; c->comm_rank = 0;
    return PTR_OK;
}

// ---- reprojection of a cache split over several GPUs ---------------------------------------------------------------
extern "C++" {
namespace {
int rc_shard_check(ptr_ctx* c, int n_local, int item_base)
{
    if (n_local < 0 || item_base < 0 || (long long)item_base + n_local >= (1ll << 31))
        return fail(c, PTR_ERR_INVALID, "bad shard: %d items from %d", n_local, item_base);
    return PTR_OK;
}

int rc_merge_enqueue(ptr_ctx* c, int width, int height, int n_ranks, int rank, const void* d_all_el, const void* d_all_dep, int n_local,
                     int item_base, void* d_age, double age_factor, double depth_threshold, void* d_item_pixel, void* d_pel, void* d_pdep,
                     hipStream_t s)
{
    const size_t cells = (size_t)(width + 2) * (size_t)(height + 2);
    PTR_TRY(grow(c, c->rcm_before, cells * sizeof(double)));
    if (n_local > 0 && c->rc_depth.bytes < (size_t)n_local * sizeof(double))
        return fail(c, PTR_ERR_INVALID, "merge: no local pass of %d items ran on this context (ptr_rc_reproject_shard_device first)", n_local);
    RcMergeArgs a;
    std::memset(&a, 0, sizeof a);
    a.width = width; a.height = height; a.n_items = n_local; a.item_base = item_base; a.n_ranks = n_ranks; a.rank = rank;
    a.age_factor = age_factor; a.depth_threshold = depth_threshold;
    a.all_depth = static_cast<const double*>(d_all_dep); a.all_element = static_cast<const int32_t*>(d_all_el);
    a.age = static_cast<double*>(d_age); a.item_pixel = static_cast<int32_t*>(d_item_pixel);
    a.item_depth = static_cast<const double*>(c->rc_depth.ptr);
    a.pixel_element = static_cast<int32_t*>(d_pel); a.pixel_depth = static_cast<double*>(d_pdep);
    a.before = static_cast<double*>(c->rcm_before.ptr);
    HIP_TRY(c, launch_rc_merge(a, s));
    return PTR_OK;
}
}  // namespace
}  // extern "C++"

int ptr_rc_reproject_shard_device(ptr_ctx* c, const ptr_rc_camera* cam, int n_local, int item_base, const void* d_hit_xyz, void* d_age,
                                  double age_factor, double depth_threshold, void* d_item_pixel, void* d_local_element,
                                  void* d_local_depth, void* stream)
{
    if (!c) return PTR_ERR_INVALID;
    std::lock_guard<std::mutex> g(c->mu);
    PTR_TRY(rc_check(c, cam, n_local, d_hit_xyz, d_age, d_item_pixel, d_local_element, d_local_depth));
    PTR_TRY(rc_shard_check(c, n_local, item_base));
    HIP_TRY(c, hipSetDevice(c->device));
    return rc_enqueue(c, cam, n_local, static_cast<const double*>(d_hit_xyz), static_cast<double*>(d_age), age_factor, depth_threshold,
                      static_cast<int32_t*>(d_item_pixel), static_cast<int32_t*>(d_local_element), static_cast<double*>(d_local_depth),
                      stream ? static_cast<hipStream_t>(stream) : c->stream, item_base);
}

int ptr_rc_merge_shards_device(ptr_ctx* c, int width, int height, int n_ranks, int rank, const void* d_all_element, const void* d_all_depth,
                               int n_local, int item_base, void* d_age, double age_factor, double depth_threshold, void* d_item_pixel,
                               void* d_pixel_element, void* d_pixel_depth, void* stream)
{
    if (!c) return PTR_ERR_INVALID;
    std::lock_guard<std::mutex> g(c->mu);
    if (width < 1 || height < 1 || (long long)(width + 2) * (height + 2) >= (1ll << 31))
        return fail(c, PTR_ERR_INVALID, "bad frame size %d x %d", width, height);
    if (n_ranks < 1 || n_ranks > 64 || rank < 0 || rank >= n_ranks) return fail(c, PTR_ERR_INVALID, "merge: rank %d of %d", rank, n_ranks);
    if (!d_all_element || !d_all_depth || !d_pixel_element || !d_pixel_depth || (n_local > 0 && (!d_age || !d_item_pixel)))
        return fail(c, PTR_ERR_INVALID, "a required buffer is NULL");
    PTR_TRY(rc_shard_check(c, n_local, item_base));
    HIP_TRY(c, hipSetDevice(c->device));
    return rc_merge_enqueue(c, width, height, n_ranks, rank, d_all_element, d_all_depth, n_local, item_base, d_age, age_factor,
                            depth_threshold, d_item_pixel, d_pixel_element, d_pixel_depth, stream ? static_cast<hipStream_t>(stream) : c->stream);
}

int ptr_rc_reproject_multi_device(ptr_ctx* c, const ptr_rc_camera* cam, int n_local, int item_base, const void* d_hit_xyz, void* d_age,
                                  double age_factor, double depth_threshold, void* d_item_pixel, void* d_pixel_element,
                                  void* d_pixel_depth, void* stream)
{
    if (!c) return PTR_ERR_INVALID;
    std::lock_guard<std::mutex> g(c->mu);
    if (c->comm_n < 1) return fail(c, PTR_ERR_INVALID, "the context belongs to no communicator (ptr_comm_init_rank / _all)");
    PTR_TRY(rc_check(c, cam, n_local, d_hit_xyz, d_age, d_item_pixel, d_pixel_element, d_pixel_depth));
    PTR_TRY(rc_shard_check(c, n_local, item_base));
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : c->stream;
    const int n = c->comm_n, me = c->comm_rank;
    const size_t cells = (size_t)(cam->width + 2) * (size_t)(cam->height + 2);
    PTR_TRY(grow(c, c->rcm_all_el, (size_t)n * cells * sizeof(int32_t)));
    PTR_TRY(grow(c, c->rcm_all_dep, (size_t)n * cells * sizeof(double)));
    int32_t* all_el = static_cast<int32_t*>(c->rcm_all_el.ptr);
    double* all_dep = static_cast<double*>(c->rcm_all_dep.ptr);
    // the local pass writes straight into this rank's place of the gathered grids when it is alone; otherwise into a
    // buffer of its own (ncclAllGather's send buffer must not lie inside the receive buffer except at its own offset)
    int32_t* loc_el = all_el + (size_t)me * cells;
    double* loc_dep = all_dep + (size_t)me * cells;
    PTR_TRY(rc_enqueue(c, cam, n_local, static_cast<const double*>(d_hit_xyz), static_cast<double*>(d_age), age_factor, depth_threshold,
                       static_cast<int32_t*>(d_item_pixel), loc_el, loc_dep, s, item_base));
    if (n > 1) {
        Rccl& r = rccl();
        if (!r.ok) return fail(c, PTR_ERR_UNSUPPORTED, "%s", r.why);
        RCCL_TRY(c, r.GroupStart());                                    // in place: each rank's grid already sits at its offset
        const int rc1 = r.AllGather(loc_el, all_el, cells, RCCL_INT32, c->comm, s);
        const int rc2 = rc1 == 0 ? r.AllGather(loc_dep, all_dep, cells, RCCL_FLOAT64, c->comm, s) : 0;
        const int rc3 = r.GroupEnd();
        if (rc1 != 0 || rc2 != 0 || rc3 != 0)
            return fail(c, PTR_ERR_HIP, "ncclAllGather failed: %s", r.GetErrorString(rc1 ? rc1 : rc2 ? rc2 : rc3));
    }
    return rc_merge_enqueue(c, cam->width, cam->height, n, me, all_el, all_dep, n_local, item_base, d_age, age_factor, depth_threshold,
                            d_item_pixel, d_pixel_element, d_pixel_depth, s);
}

int ptr_gather_strips_device(ptr_ctx* c, int width, int height, int strip_height, int n_frames,
                             const void* d_local, uint64_t frame_stride_px, void* d_full, int root, void* stream)
{
    if (!c) return PTR_ERR_INVALID;
    std::lock_guard<std::mutex> lk(c->mu);
    if (c->comm_n < 1) return fail(c, PTR_ERR_INVALID, "gather: the context belongs to no communicator (ptr_comm_init_rank / _all)");
    const int n = c->comm_n, me = c->comm_rank;
    if (root < 0 || root >= n) return fail(c, PTR_ERR_INVALID, "gather: root %d of %d ranks", root, n);
    if (!d_local || (me == root && !d_full)) return fail(c, PTR_ERR_INVALID, "gather: a required buffer is NULL");
    GatherShape g;
    PTR_TRY(gather_shape(c, width, height, strip_height, n_frames, frame_stride_px, me, n, &g));
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : c->stream;
    uint32_t* stage = nullptr;
    if (me == root && n > 1) {
        PTR_TRY(grow(c, c->gather_stage, (size_t)(n - 1) * g.count * sizeof(uint32_t)));
        stage = static_cast<uint32_t*>(c->gather_stage.ptr);
    }
    if (n > 1) {
        // a direct gather: every peer sends over its own xGMI link to the root; one group, so the transfers run side by side
        Rccl& r = rccl();
        RCCL_TRY(c, r.GroupStart());
        int rc = 0;
        if (me == root) {
            int slot = 0;
            for (int p = 0; p < n && rc == 0; p++)
                if (p != root) rc = r.Recv(stage + (size_t)(slot++) * g.count, g.count, RCCL_UINT32, p, c->comm, s);
        } else {
            rc = r.Send(d_local, g.count, RCCL_UINT32, root, c->comm, s);
        }
        const int rc2 = r.GroupEnd();
        if (rc != 0 || rc2 != 0) return fail(c, PTR_ERR_HIP, "gather: ncclSend/ncclRecv failed: %s", r.GetErrorString(rc ? rc : rc2));
    }
    if (me == root) {
        int slot = 0;
        for (int p = 0; p < n; p++) {
            const uint32_t* src = p == root ? static_cast<const uint32_t*>(d_local) : stage + (size_t)(slot++) * g.count;
            PTR_TRY(gather_place(c, src, static_cast<uint32_t*>(d_full), width, height, strip_height, n_frames, g, p, n, s));
        }
    }
    return PTR_OK;
}

int ptr_place_strips_device(ptr_ctx* c, const ptr_strips* st, int width, int height, int n_frames,
                            const void* d_part, uint64_t frame_stride_px, void* d_full, void* stream)
{
    if (!c) return PTR_ERR_INVALID;
    std::lock_guard<std::mutex> lk(c->mu);
    if (!st || st->n_parts < 1 || st->part < 0 || st->part >= st->n_parts) return fail(c, PTR_ERR_INVALID, "place: bad strip description");
    if (!d_part || !d_full) return fail(c, PTR_ERR_INVALID, "place: a required buffer is NULL");
    GatherShape g;
    PTR_TRY(gather_shape(c, width, height, st->strip_height, n_frames, 0, st->part, st->n_parts, &g));
    g.stride = frame_stride_px ? frame_stride_px : (uint64_t)g.rows * (uint64_t)width;
    if (g.stride < (uint64_t)g.rows * (uint64_t)width) return fail(c, PTR_ERR_INVALID, "place: frame_stride_px smaller than the part");
    HIP_TRY(c, hipSetDevice(c->device));
    return gather_place(c, static_cast<const uint32_t*>(d_part), static_cast<uint32_t*>(d_full), width, height, st->strip_height,
                        n_frames, g, st->part, st->n_parts, stream ? static_cast<hipStream_t>(stream) : c->stream);
}

int ptr_gather_strips_all(ptr_ctx* const* ctxs, int n, int width, int height, int strip_height, int n_frames,
                          const void* const* d_locals, uint64_t frame_stride_px, void* d_full)
{
    if (!ctxs || n < 1 || !ctxs[0]) return PTR_ERR_INVALID;
    ptr_ctx* root = ctxs[0];
    if (!d_locals || !d_full) return fail(root, PTR_ERR_INVALID, "gather: a required buffer is NULL");
    for (int i = 0; i < n; i++)
        if (!ctxs[i] || ctxs[i]->comm_n != n || ctxs[i]->comm_rank != i || !d_locals[i])
            return fail(root, PTR_ERR_INVALID, "gather: context %d is not rank %d of a communicator of %d (ptr_comm_init_all), or its buffer is NULL", i, i, n);
    GatherShape g;
    PTR_TRY(gather_shape(root, width, height, strip_height, n_frames, frame_stride_px, 0, n, &g));
    uint32_t* stage = nullptr;
    if (n > 1) {
        HIP_TRY(root, hipSetDevice(root->device));
        PTR_TRY(grow(root, root->gather_stage, (size_t)(n - 1) * g.count * sizeof(uint32_t)));
        stage = static_cast<uint32_t*>(root->gather_stage.ptr);
        Rccl& r = rccl();
        RCCL_TRY(root, r.GroupStart());
        int rc = 0;
        for (int i = 1; i < n && rc == 0; i++) {
            rc = r.Recv(stage + (size_t)(i - 1) * g.count, g.count, RCCL_UINT32, i, root->comm, root->stream);
            if (rc == 0) rc = r.Send(d_locals[i], g.count, RCCL_UINT32, 0, ctxs[i]->comm, ctxs[i]->stream);
        }
        const int rc2 = r.GroupEnd();
        if (rc != 0 || rc2 != 0) return fail(root, PTR_ERR_HIP, "gather: ncclSend/ncclRecv failed: %s", r.GetErrorString(rc ? rc : rc2));
    }
    HIP_TRY(root, hipSetDevice(root->device));
    for (int i = 0; i < n; i++) {
        const uint32_t* src = i == 0 ? static_cast<const uint32_t*>(d_locals[0]) : stage + (size_t)(i - 1) * g.count;
        PTR_TRY(gather_place(root, src, static_cast<uint32_t*>(d_full), width, height, strip_height, n_frames, g, i, n, root->stream));
    }
    return PTR_OK;
}

int ptr_render_multi(ptr_ctx* const* ctxs, int n, const ptr_params* p, uint32_t* out_rgba8)
{
    if (!ctxs || n < 1 || !ctxs[0]) return PTR_ERR_INVALID;
    ptr_ctx* root = ctxs[0];
    if (!p || !out_rgba8) return fail(root, PTR_ERR_INVALID, "ptr_render_multi: params / out_rgba8 is NULL");
    if (p->start_y != 0 || p->scan_height != p->height || (p->flags & PTR_QUIRK_ROW_SHIFT))
        return fail(root, PTR_ERR_INVALID, "ptr_render_multi renders whole frames: start_y 0, scan_height == height, no PTR_QUIRK_ROW_SHIFT");
    for (int i = 0; i < n; i++)
        if (!ctxs[i] || ctxs[i]->comm_n != n || ctxs[i]->comm_rank != i)
            return fail(root, PTR_ERR_INVALID, "ptr_render_multi: context %d is not rank %d of a communicator of %d (ptr_comm_init_all)", i, i, n);
    constexpr int STRIP = 8;
    const ptr_strips s0{n, 0, STRIP, 0};
    const int rows_max = ptr_strip_rows(&s0, p->height);
    if (rows_max < 1) return fail(root, PTR_ERR_INVALID, "ptr_render_multi: bad height %d", p->height);
    const uint64_t stride = (uint64_t)rows_max * (uint64_t)p->width;
    const void* locals[64];
    for (int i = 0; i < n; i++) {
        ptr_ctx* c = ctxs[i];
        std::lock_guard<std::mutex> lk(c->mu);
        const ptr_strips sp{n, i, STRIP, 0};
        ptr_params q = *p;
        q.scan_height = ptr_strip_rows(&sp, p->height);
        HIP_TRY(c, hipSetDevice(c->device));
        PTR_TRY(grow(c, c->multi_local, (size_t)stride * sizeof(uint32_t)));
        locals[i] = c->multi_local.ptr;
        if (q.scan_height < 1) continue;                       // more GPUs than strips: nothing to render here
        int rc = check_params(c, &q, 0, 1);
        if (rc == PTR_OK) rc = enqueue_render(c, &q, 0, 1, static_cast<uint32_t*>(c->multi_local.ptr), nullptr, c->stream, &sp, stride);
        if (rc != PTR_OK) { if (c != root) fail(root, rc, "ptr_render_multi, GPU %d: %s", c->device, c->err); return rc; }
    }
    const size_t px = (size_t)p->width * (size_t)p->height;
    {
        std::lock_guard<std::mutex> lk(root->mu);
        HIP_TRY(root, hipSetDevice(root->device));
        PTR_TRY(grow(root, root->multi_full, px * sizeof(uint32_t)));
    }
    PTR_TRY(ptr_gather_strips_all(ctxs, n, p->width, p->height, STRIP, 1, locals, stride, root->multi_full.ptr));
    std::lock_guard<std::mutex> lk(root->mu);
    HIP_TRY(root, hipSetDevice(root->device));
    HIP_TRY(root, hipMemcpyAsync(out_rgba8, root->multi_full.ptr, px * sizeof(uint32_t), hipMemcpyDeviceToHost, root->stream));
    HIP_TRY(root, hipStreamSynchronize(root->stream));
    for (int i = 1; i < n; i++) {                              // the peers' sends are done once the root has everything; leave no work behind
        HIP_TRY(ctxs[i], hipSetDevice(ctxs[i]->device));
        HIP_TRY(ctxs[i], hipStreamSynchronize(ctxs[i]->stream));
    }
    return PTR_OK;
}

}  // extern "C"
