// ptr_api_rc.hip — host side of the C ABI for the render cache (rendercache-js): the reprojection pass, depth culling + gap
// filling + frames, the frame loop on a cache that stays in HBM, and the per-rank halves of the reprojection of a cache
// split over several GPUs.  Declarations and reference sites: include/ptr.h.
#include "ptr_host.h"

namespace ptrh {
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
               double depth_threshold, int32_t* d_item_pixel, int32_t* d_pel, double* d_pdep, hipStream_t s, int item_base,
               uint8_t* d_pixel_sample)      // (the frame loop's sample flags: cleared with the grid's interior)
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
}  // namespace ptrh

namespace ptrh {
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
}  // namespace ptrh

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
namespace ptrh {
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
}  // namespace ptrh

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
        if (host[k]) PTR_TRY(copy_to_host(c, host[k], a.frames[k], px * 4, s));
    return PTR_OK;
}

namespace ptrh {
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
}  // namespace ptrh

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

