// ptr_api.hip — host side of the C ABI declared in include/ptr.h: context, scene, render and trace entry points.
// (ptr_api_rc.hip: the render cache; ptr_api_comm.hip: RCCL, the strip gather, the multi-GPU forms; ptr_host.h: what they share.)
//
// Owns all device memory of a renderer context: the frame blocks (scene sequence flattened to
// the layouts of ptr_device.h), the re-trace queue and its two counter words, the stats words
// and staging buffers for the host-pointer entry points.  There is no CPU implementation of
// the path in here: without a GPU ptr_create fails.
#include "ptr_host.h"

namespace ptrh {
thread_local char g_create_error[512] = "";
static __device__ __align__(256) unsigned char scratch[NOALLOC_SCRATCH_BYTES];
}  // namespace ptrh

namespace ptrh {

int fail(ptr_ctx* c, int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(c ? c->err : g_create_error, 512, fmt, ap);
    va_end(ap);
    return code;
}


int grow(ptr_ctx* c, DeviceBuffer& b, size_t bytes)
{
    if (bytes <= b.bytes) return PTR_OK;
    if (b.ptr) { HIP_TRY(c, hipFree(b.ptr)); b.ptr = nullptr; b.bytes = 0; }
    HIP_TRY(c, hipMalloc(&b.ptr, bytes));
    b.bytes = bytes;
    return PTR_OK;
}

int copy_to_host(ptr_ctx* c, void* dst, const void* d_src, size_t bytes, hipStream_t s)
{
    if (bytes == 0) return PTR_OK;
    bool direct = bytes <= HOST_STAGE_MIN;                 // small: the set-up below costs more than it saves
    if (!direct) {
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, dst) == hipSuccess) direct = at.type == hipMemoryTypeHost;      // page-locked already
        else (void)hipGetLastError();                      // (an address the runtime does not know: ordinary memory)
    }
    // Pieces of at most 1 MiB, at least two of them (measured on one box, 1080p frame, ptr_render as a whole: pieces of 128 KiB
    // 1.08 ms, 256 KiB 0.68, 512 KiB 0.45, 1 MiB 0.40-0.41, 2 MiB 0.43, 4 MiB 0.44; the driver's own pageable path 0.54; a
    // page-locked destination 0.22: every piece costs its start-up, larger pieces overlap less).
    size_t half = ((bytes + 1) / 2 + 65535) & ~(size_t)65535;
    if (half > HOST_STAGE_MAX_HALF) half = HOST_STAGE_MAX_HALF;
    if (!direct && c->host_stage_half < half) {
        // (should the host refuse to lock pages, the plain copy still works)
        if (c->host_stage) { (void)hipHostFree(c->host_stage); c->host_stage = nullptr; c->host_stage_half = 0; }
        bool ok = hipHostMalloc(reinterpret_cast<void**>(&c->host_stage), 2 * half, hipHostMallocDefault) == hipSuccess;
        for (auto& e : c->stage_ev)
            if (ok && !e) ok = hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
        if (ok) c->host_stage_half = half;
        else {
            (void)hipGetLastError();
            if (c->host_stage) { (void)hipHostFree(c->host_stage); c->host_stage = nullptr; }
            direct = true;
        }
    }
    if (direct) {
        HIP_TRY(c, hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, s));
        HIP_TRY(c, hipStreamSynchronize(s));
        return PTR_OK;
    }
    const size_t n_chunks = (bytes + half - 1) / half;
    auto chunk_bytes = [&](size_t i) { return i + 1 < n_chunks ? half : bytes - i * half; };
    for (size_t i = 0; i <= n_chunks; i++) {
        if (i < n_chunks) {                                // chunk i into half i & 1 (emptied one trip ago)
            HIP_TRY(c, hipMemcpyAsync(c->host_stage + (i & 1) * c->host_stage_half, static_cast<const unsigned char*>(d_src) + i * half,
                                      chunk_bytes(i), hipMemcpyDeviceToHost, s));
            HIP_TRY(c, hipEventRecord(c->stage_ev[i & 1], s));
        }
        if (i > 0) {                                       // chunk i - 1 out of the other half, while the copy engine works
            // (polled: hipEventSynchronize gives the thread up and comes back tens of microseconds after the copy is done)
            hipError_t q;
            while ((q = hipEventQuery(c->stage_ev[(i - 1) & 1])) == hipErrorNotReady) {}
            (void)hipGetLastError();                       // ("not ready" must not show up as a later launch's error)
            if (q != hipSuccess) return fail(c, PTR_ERR_HIP, "render failed: %s", hipGetErrorString(q));
            std::memcpy(static_cast<unsigned char*>(dst) + (i - 1) * half, c->host_stage + ((i - 1) & 1) * c->host_stage_half,
                        chunk_bytes(i - 1));
        }
    }
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
    uint32_t off_g64, off_m64, off_p32, off_m32, off_lights, off_pairs, off_pairs_x, off_lrec, off_cull, off_x4;
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
    // ... and the same table for the bounce loop, whose shadow rays start anywhere on the spheres (build_frame_block)
    L.off_pairs_x = (uint32_t)o;
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
    // A culled pair is a miss in EXACT arithmetic: the shadow ray's line stays farther than r + 1e-6 (r + D) from the occluder's
    // centre (`reach` below minus the bias), i.e. d2 - r^2 > 2e-6 r (r + D).  What every kernel has to agree with, though, is the
    // reference's FLOAT64 test, which computes d2 = |l|^2 - tca^2 with an error of up to 14 eps |l|^2 (eps = 2^-53): where that
    // exceeds the margin its outcome is rounding noise, and the noise is the reference's result.  A pair is therefore cullable
    // only if 8 x 14 eps |l|^2 stays below its margin for every shadow origin the kernels that use the table can produce
    // (|l| <= |origin - camera| + |centre - camera|).  Two tables: camera rays (reference semantics) end within the tangent
    // length sqrt(cc) of a sphere seen from outside, within |l| + r of one seen from inside; the bounce loop's rays end
    // anywhere on the spheres, up to a sphere's far side.
    double reach_cam = 0.0, reach_max = 0.0;
    for (int j = 0; j < n; j++) {
        const double ln = (double)p32[j].lnorm * 1.000001, r = std::fabs(pl(3, j)), cc = ln * ln - r * r;
        reach_cam = std::fmax(reach_cam, (cc > 0 ? std::sqrt(cc) : ln + r) + 2.0e-4);
        reach_max = std::fmax(reach_max, ln + r + 2.0e-4);
    }
    const double noise = 8.0 * 14.0 * 1.1102230246251565e-16;
    auto fill_pairs = [&](PairQ* pairs, double reach_bound) {
    for (int k = 0; k < nl && k < PTR_MAX_PAIR_LIGHTS; k++) {
        const int i = lights[k];
        for (int j = 0; j < n; j++) {
            PairQ q{0.0f, 0.0f, 0.0f, -1.0f};                        // thr < 0: never culled
            const double ax = pl(0, j) - pl(0, i), ay = pl(1, j) - pl(1, i), az = pl(2, j) - pl(2, i);
            const double D = std::sqrt(ax * ax + ay * ay + az * az);
            const double reach = std::fabs(pl(3, j)) + 1.0e-4 + 1.0e-6 * (std::fabs(pl(3, j)) + D);   // r + bias + slop
            const double l_max = reach_bound + (double)p32[j].lnorm * 1.000001;
            const bool conditioned = noise * l_max * l_max < 2.0e-6 * std::fabs(pl(3, j)) * (std::fabs(pl(3, j)) + D);
            if (D > 0 && reach < D && std::isfinite(D) && conditioned) {
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
    };
    fill_pairs(reinterpret_cast<PairQ*>(dst + L.off_pairs), reach_cam);
    PairQ* pairs = reinterpret_cast<PairQ*>(dst + L.off_pairs_x);
    fill_pairs(pairs, reach_max);
    {
        // The float64 bounce loop with planes: hit points on a plane lie anywhere, so the conditioning argument above needs a
        // bound per ray: the cull is allowed only while the shadow origin is within sqrt(pair_smax2) of the camera (the
        // smallest margin of any cullable pair decides).  Without planes the table is safe as built (pair_always).
        double worst = INFINITY, far = 0.0;
        for (int j = 0; j < n; j++) far = std::fmax(far, (double)p32[j].lnorm);
        for (int k = 0; k < nl && k < PTR_MAX_PAIR_LIGHTS; k++)
            for (int j = 0; j < n; j++) {
                if (!(pairs[(size_t)k * (size_t)pair_stride(n) + (size_t)j].thr > 0.0f)) continue;       // never culled anyway
                const int i = lights[k];
                const double ax = pl(0, j) - pl(0, i), ay = pl(1, j) - pl(1, i), az = pl(2, j) - pl(2, i);
                const double r = std::fabs(pl(3, j));
                worst = std::fmin(worst, r * (r + std::sqrt(ax * ax + ay * ay + az * az)));
            }
        const double ll_safe = 2.0e-6 * worst / noise;
        const double smax = std::sqrt(ll_safe) - far * 1.000001;
        h->pair_smax2 = (std::isfinite(smax) && smax > 0) ? (float)(smax * smax * 0.999) : (worst == INFINITY ? 3.0e38f : -1.0f);
        h->pair_always = 1u;              // (every pair left in the table is well conditioned for hit points on spheres)
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

// Is stream s recording a HIP graph?  Then nothing the context keeps may be tied to it: an event recorded on a capturing
// stream becomes a captured event, and ptr_get_stats' hipEventSynchronize or another stream's wait on it would fail
// (hipErrorCapturedEvent) or invalidate the capture.  Ordering inside a graph is the graph's own.
bool stream_is_capturing(hipStream_t s)
{
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess) { (void)hipGetLastError(); return false; }
    return st == hipStreamCaptureStatusActive;
}

// make stream s wait for the kernels of the context's most recent call, if that call went to another stream
int order_after_last(ptr_ctx* c, hipStream_t s)
{
    if (stream_is_capturing(s)) return PTR_OK;      // (the caller orders a graph against earlier work when it launches it)
    if (c->have_last && c->last_stream != s) HIP_TRY(c, hipStreamWaitEvent(s, c->last_ev, 0));
    return PTR_OK;
}

int mark_launch(ptr_ctx* c, hipStream_t s)
{
    if (stream_is_capturing(s)) return PTR_OK;      // nothing was enqueued: the kernels run when the graph is launched
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

// The float32 error budget of the hit tests (PR_RELB: 10 eps for b = l.d) counts on a camera ray that is good to about one
// eps, which is what xx R + yy U + F gives while its terms are of the size of the sum: the direction's relative error grows
// like g = (|xx|max |R| + |yy|max |U| + |F|) / dmin (1.7 for the reference camera at 16:9; measured 1.1 eps there, 3.3 eps at
// g = 10-11: fov 150 degrees, or aspect 32).  Wide or skewed cameras beyond g = 4 are rendered in float64 rather than trusted
// to the budget.  dmin = distance of the image plane from the eye.
bool camera_needs_exact(const ptr_ctx* c, const ptr_params* p, double* dmin_out = nullptr)
{
    const double* R = c->cam.right; const double* U = c->cam.up; const double* F = c->cam.forward;
    const double nx = R[1] * U[2] - R[2] * U[1], ny = R[2] * U[0] - R[0] * U[2], nz = R[0] * U[1] - R[1] * U[0];
    const double nn = std::sqrt(nx * nx + ny * ny + nz * nz);
    const double dmin = nn > 0 ? std::fabs(F[0] * nx + F[1] * ny + F[2] * nz) / nn : 0.0;
    if (dmin_out) *dmin_out = dmin;
    const double angle = c->cam.tan_half_fov, aspect = (double)p->width / (double)p->height;
    const double g = dmin > 0 ? (std::fabs(angle * aspect) * std::sqrt(R[0] * R[0] + R[1] * R[1] + R[2] * R[2]) +
                                 std::fabs(angle) * std::sqrt(U[0] * U[0] + U[1] * U[1] + U[2] * U[2]) +
                                 std::sqrt(F[0] * F[0] + F[1] * F[1] + F[2] * F[2])) / dmin : 1e300;
    return !(g <= 4.0);
}

// Which kernels a call takes, and the re-trace queue a hybrid call needs (ids + 1 or 2 info words per slot).
struct RenderPlan {
    bool extension, exact, many, hybrid;
    uint64_t n_queues, sub_capacity;
    size_t queue_bytes;
};
RenderPlan plan_render(const ptr_ctx* c, const ptr_params* p, int n_frames)
{
    RenderPlan r{};
    const bool stats = (p->flags & PTR_FLAG_STATS) != 0;
    r.extension = p->max_depth != 0 || p->spp != 1 || c->n_planes > 0 || c->n_phong > 0;
    // (PTR_MODE_F32_ONLY is the caller's explicit choice of float32 whatever the camera)
    r.exact = stats || (p->flags & PTR_MODE_EXACT_F64) || c->needs_exact ||
              (!(p->flags & PTR_MODE_F32_ONLY) && !r.extension && camera_needs_exact(c, p));
    r.many = c->n_spheres > PTR_FAST_MAX_SPHERES || (p->flags & PTR_DEBUG_FORCE_MANY);
    r.hybrid = !r.extension && !r.exact && !(p->flags & PTR_MODE_F32_ONLY);
    if (!r.hybrid) return r;
    const int rpb = r.many ? many_kernel_rows_per_block() : fast_kernel_rows_per_block();
    const int cpb = r.many ? PTR_TILE_W : fast_kernel_cols_per_block((p->flags & PTR_MODE_TOLERANT) ? 2 : 1);
    const uint64_t n_blocks = (uint64_t)((p->width + cpb - 1) / cpb) *
                              (uint64_t)((p->scan_height + rpb - 1) / rpb) * (uint64_t)n_frames;
    r.n_queues = 1;                                              // a power of two (the kernel masks)
    while (r.n_queues * 2 <= PTR_MAX_QUEUES && r.n_queues * 8 <= n_blocks) r.n_queues *= 2;
    // a workgroup can queue at most its tile's pixels; sub-queue s serves the workgroups b with (b & (n_queues - 1)) == s
    r.sub_capacity = (n_blocks + r.n_queues - 1) / r.n_queues * (uint64_t)(cpb * rpb);
    r.queue_bytes = (size_t)(r.n_queues * r.sub_capacity) * (r.many ? 3 : 2) * sizeof(uint32_t);
    return r;
}

// Enqueue the kernels of one render call; no synchronisation.  fixed_ws: the kernels' words and re-trace queue come from
// the static array (ptr_render_device_noalloc) and nothing in here allocates; otherwise from the context's own buffers,
// grown on demand.
int enqueue_render(ptr_ctx* c, const ptr_params* p, int first_frame, int n_frames,
                   uint32_t* d_out8, float* d_out32f, hipStream_t s,
                   const ptr_strips* strips, uint64_t frame_stride_px, bool fixed_ws)
{
    // (the plan reads c->needs_exact, which the upload of a changed scene sets: upload first, where uploading is allowed)
    if (!fixed_ws) {
        PTR_TRY(order_after_last(c, s));      // queue, counters and stats words are shared by all calls on the context
        PTR_TRY(ensure_uploaded(c, s));
    }
    const RenderPlan plan = plan_render(c, p, n_frames);
    if (fixed_ws) {
        // every reason to refuse the call comes first: a refused call enqueues nothing and leaves the shared array, and
        // what ptr_get_stats reads, as the last successful call left them
        if (c->dirty)
            return fail(c, PTR_ERR_INVALID, "ptr_render_device_noalloc: the scene is not resident on the GPU yet; call "
                                            "ptr_sync() after the last ptr_set_* call (the upload allocates)");
        if (p->flags & PTR_FLAG_TIMING)
            return fail(c, PTR_ERR_INVALID, "ptr_render_device_noalloc: PTR_FLAG_TIMING is not available (it creates events)");
        const size_t need = plan.queue_bytes;
        if (plan.hybrid && need > NOALLOC_SCRATCH_BYTES - NOALLOC_QUEUE_OFFSET)
            return fail(c, PTR_ERR_INVALID, "ptr_render_device_noalloc: this call needs a re-trace queue of %zu bytes (8 per "
                                            "pixel of its tiles, 12 with more than 12 spheres), the static array holds %zu: render fewer pixels per call "
                                            "(about 3.9 M at most), or use ptr_render_device",
                        need, NOALLOC_SCRATCH_BYTES - NOALLOC_QUEUE_OFFSET);
    }
    if (fixed_ws) {
        PTR_TRY(order_after_last(c, s));      // queue, counters and stats words are shared by all calls on the context
        // every call on this GPU shares the array and finds the last call's contents in it: clear what must start at zero
        // (by a kernel, not hipMemsetAsync: as a HIP graph's memset node the clear came out wrong from the second replay on)
        HIP_TRY(c, launch_clear_words(c->scratch_dev, WORDS_BYTES, s));
    } else {
        PTR_TRY(ensure_words(c, s));
    }

    RenderArgs a;
    std::memset(&a, 0, sizeof a);
    // (the kernels see the call's first frame as frame 0; the few-sphere kernel counts on a call's blocks spanning < 4 GB)
    a.scenes = static_cast<const unsigned char*>(c->scenes.ptr) + (size_t)first_frame * c->frame_stride;
    a.frame_stride = c->frame_stride;
    {
        const BlockLayout L = block_layout(c->n_spheres);
        a.off_p32 = L.off_p32; a.off_m32 = L.off_m32; a.off_lights = L.off_lights; a.off_pairs = plan.extension ? L.off_pairs_x : L.off_pairs; a.off_lrec = L.off_lrec; a.off_cull = L.off_cull; a.off_x4 = L.off_x4;
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
    }
    a.out8 = d_out8;
    a.out32f = d_out32f;
    unsigned char* words = fixed_ws ? c->scratch_dev : static_cast<unsigned char*>(c->words.ptr);
    a.stats = reinterpret_cast<unsigned long long*>(words + WORDS_STATS);
    c->stats_words = words + WORDS_STATS;       // (no rejection below this line)

    const bool stats = (p->flags & PTR_FLAG_STATS) != 0;
    const bool exact = plan.exact;
    const bool timing = (p->flags & PTR_FLAG_TIMING) != 0;
    if (stats && !fixed_ws) HIP_TRY(c, hipMemsetAsync(words + WORDS_STATS, 0, 256, s));
    if (timing) {
        for (auto& e : c->ev)
            if (!e) HIP_TRY(c, hipEventCreate(&e));
        c->timed_kernels = 0;
        HIP_TRY(c, hipEventRecord(c->ev[0], s));
    }

    if (plan.extension) {
        // Extension modes (bounces, samples, planes, Phong terms): float64 = the specification bit for bit, and that is the DEFAULT: a float32 ray tree has no
        // error bounds yet that would tell which pixels to re-do (errors grow with every bounce off a curved surface), and
        // the contract is <= 1/255 on every pixel.  PTR_MODE_F32_ONLY asks for the float32 throughput form explicitly.
        a.max_depth = p->max_depth; a.spp = p->spp; a.seed = p->seed;
        a.ext_blocks = (p->spp + PTR_EXT_SPP_BLOCK - 1) / PTR_EXT_SPP_BLOCK;
        const bool f32_form = !exact && (p->flags & PTR_MODE_F32_ONLY) && !stats;
        if (f32_form && a.ext_blocks > 1 && !fixed_ws && (uint64_t)n_frames * (uint64_t)a.ext_blocks <= 65535u) {
            // a wave per block of samples (ptr_kernels_ext.hip): room for the blocks' sums, unless that is out of proportion
            // (then, and for the caller that cannot allocate, every wave takes all blocks of its pixels: same bytes)
            const uint64_t bytes = (uint64_t)n_frames * (uint64_t)a.ext_blocks * (uint64_t)p->width * (uint64_t)p->scan_height * 16u;
            if (bytes <= EXT_PARTIAL_MAX_BYTES) {
                PTR_TRY(order_after_last(c, s));
                PTR_TRY(grow(c, c->ext_partial, (size_t)bytes));
                a.ext_partial = static_cast<float*>(c->ext_partial.ptr);
            }
        }
        HIP_TRY(c, launch_render_ext(a, n_frames, exact || !(p->flags & PTR_MODE_F32_ONLY), stats, c->n_planes == 0 && c->n_phong == 0,
                                     (p->flags & PTR_DEBUG_EXT_LANEWISE) != 0, s));
        if (timing) { HIP_TRY(c, hipEventRecord(c->ev[1], s)); c->timed_kernels = 1; }
        return mark_launch(c, s);
    }
    if (exact) {
        HIP_TRY(c, launch_render_exact(a, n_frames, stats, s));
        if (timing) { HIP_TRY(c, hipEventRecord(c->ev[1], s)); c->timed_kernels = 1; }
        return mark_launch(c, s);
    }
    const bool many = plan.many;
    if (!many && (uint64_t)n_frames * (uint64_t)c->frame_stride >= (1ull << 32))
        return fail(c, PTR_ERR_INVALID, "internal: %d frame blocks of %zu bytes span 4 GB or more", n_frames, c->frame_stride);
    if (p->flags & PTR_MODE_F32_ONLY) {
        HIP_TRY(c, many ? launch_render_many(a, n_frames, false, s) : launch_render_fast(a, n_frames, 0, s));
        if (timing) { HIP_TRY(c, hipEventRecord(c->ev[1], s)); c->timed_kernels = 1; }
        return mark_launch(c, s);
    }
    // hybrid: float32 kernel fills the sub-queues, float64 kernel drains them
    const uint64_t n_queues = plan.n_queues, sub_capacity = plan.sub_capacity;
    const size_t queue_bytes = plan.queue_bytes;
    if (!fixed_ws) PTR_TRY(grow(c, c->queue, queue_bytes));       // (the static array's size was checked before anything was enqueued)
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
    // (PTR_MODE_TOLERANT: the few-sphere kernel has a form of its own; the many-sphere kernel reads the flag and trades its
    // rounding doubt for one test of the colour's error bound, its queue keeps both kinds)
    hipError_t e = many ? launch_render_many(a, n_frames, true, s)
                        : launch_render_fast(a, n_frames, (p->flags & PTR_MODE_TOLERANT) ? 2 : 1, s);
    if (e == hipSuccess && timing) e = hipEventRecord(c->ev[1], s);
    if (e == hipSuccess) {
        const bool tol = !many && (p->flags & PTR_MODE_TOLERANT);       // (no shading-only entries: one launch fewer)
        e = many ? launch_retrace_exact(a, 8, 8, s) : launch_retrace_exact(a, tol ? 0 : 1, 4, s);
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

}  // namespace ptrh

// =============================================================================================

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
    for (DeviceBuffer* b : {&c->gather_stage, &c->multi_local, &c->multi_full, &c->scenes, &c->queue, &c->words, &c->ext_partial, &c->out8, &c->out32f, &c->tr_in_o,
                            &c->tr_in_d, &c->tr_rgb, &c->tr_el, &c->tr_hit, &c->tr_nrm, &c->rc_depth, &c->rc_keys,
                            &c->rc_temp, &c->rc_hit, &c->rc_age, &c->rc_item, &c->rc_pel, &c->rc_pdep, &c->rc_rgb,
                            &c->rc_prgb, &c->rc_pprio, &c->rc_hist, &c->rc_frames, &c->rc_page, &c->rcm_before, &c->rcm_local_el,
                            &c->rcm_local_dep, &c->rcm_all_el, &c->rcm_all_dep})
        if (b->ptr) (void)hipFree(b->ptr);
    if (c->host_stage) (void)hipHostFree(c->host_stage);
    for (auto& e : c->stage_ev) if (e) (void)hipEventDestroy(e);
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
        // both lists are built aside and swapped in together with their counts: a failed allocation leaves the context as it was
        std::vector<double> new_planes(planes_soa, planes_soa + (size_t)14 * (size_t)n_planes), new_phong;
        if (phong) new_phong.assign(phong, phong + 2 * (size_t)n_elements);
        c->planes.swap(new_planes);
        c->phong.swap(new_phong);
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
    int rc = copy_to_host(c, out_rgba8, c->out8.ptr, px * 4, c->stream);
    if (rc == PTR_OK && out_rgba32f) rc = copy_to_host(c, out_rgba32f, c->out32f.ptr, px * 16, c->stream);
    if (rc != PTR_OK) c->words_dirty = true;               // (a failed kernel shows here: the copies wait for the stream)
    return rc;
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

}  // extern "C"
