// ptr_host.h — what the translation units of the C ABI's host side share: the context, its buffers, the error macros and
// the few helpers that cross file boundaries.  ptr_api.hip: context, scene, render and trace entry points;
// ptr_api_rc.hip: the render cache (reprojection, shading, the resident frame loop); ptr_api_comm.hip: RCCL, the strip
// gather and the multi-GPU forms.  Not part of the public interface (that is include/ptr.h).
#pragma once
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

namespace ptrh {


extern thread_local char g_create_error[512];      // fixed buffers: reporting an error must never throw (defined in ptr_api.hip)

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
constexpr uint64_t EXT_PARTIAL_MAX_BYTES = (uint64_t)2 << 30;     // the float32 bounce loop's per-block sums (16 B per pixel and block)
constexpr size_t NOALLOC_SCRATCH_BYTES = (size_t)32 << 20;
constexpr size_t NOALLOC_QUEUE_OFFSET  = (WORDS_BYTES + 255) / 256 * 256;


}  // namespace ptrh
using namespace ptrh;

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
    DeviceBuffer ext_partial;          // float32 bounce loop, more than one block of samples per pixel: the blocks' sums
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
    // frames that go to PAGEABLE host memory (ptr_render, ptr_render_multi, ptr_rc_get_frames): two page-locked halves that the
    // copy engine fills in turn while the CPU empties the other into the caller's buffer (copy_to_host)
    unsigned char* host_stage = nullptr;
    size_t host_stage_half = 0;
    hipEvent_t stage_ev[2] = {nullptr, nullptr};
    unsigned char* stats_words = nullptr;   // where the last call's kernels put their stats words
    unsigned char* scratch_dev = nullptr;   // this GPU's address of the static array `scratch`
};

namespace ptrh {

int fail(ptr_ctx* c, int code, const char* fmt, ...);

#define HIP_TRY(c, call)                                                                      \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail((c), PTR_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_));     \
    } while (0)

#define PTR_TRY(expr)                                                                         \
    do { int rc_ = (expr); if (rc_ != PTR_OK) return rc_; } while (0)

int grow(ptr_ctx* c, DeviceBuffer& b, size_t bytes);
// device -> host on stream s, complete on return.  A page-locked destination takes one DMA; a pageable one is filled through
// the context's two page-locked halves (the driver's own path for pageable memory runs at a third of the link's rate).
int copy_to_host(ptr_ctx* c, void* dst, const void* d_src, size_t bytes, hipStream_t s);
constexpr size_t HOST_STAGE_MIN = (size_t)256 << 10, HOST_STAGE_MAX_HALF = (size_t)1 << 20;
void rc_cache_free(ptr_ctx* c);
bool stream_is_capturing(hipStream_t s);
int order_after_last(ptr_ctx* c, hipStream_t s);      // make s wait for the context's most recent call on another stream
int mark_launch(ptr_ctx* c, hipStream_t s);
int ensure_uploaded(ptr_ctx* c, hipStream_t s);
int check_params(ptr_ctx* c, const ptr_params* p, int first_frame, int n_frames);
// Enqueue the kernels of one render call; no synchronisation (ptr_api.hip).
int enqueue_render(ptr_ctx* c, const ptr_params* p, int first_frame, int n_frames, uint32_t* d_out8, float* d_out32f, hipStream_t s,
                   const ptr_strips* strips = nullptr, uint64_t frame_stride_px = 0, bool fixed_ws = false);

// render cache (ptr_api_rc.hip)
int rc_check(ptr_ctx* c, const ptr_rc_camera* cam, int n_items, const void* hit, const void* age, const void* item_pixel,
             const void* pel, const void* pdep);
int rc_enqueue(ptr_ctx* c, const ptr_rc_camera* cam, int n_items, const double* d_hit, double* d_age, double age_factor,
               double depth_threshold, int32_t* d_item_pixel, int32_t* d_pel, double* d_pdep, hipStream_t s, int item_base = 0,
               uint8_t* d_pixel_sample = nullptr);
int rc_shard_check(ptr_ctx* c, int n_local, int item_base);
int rc_merge_enqueue(ptr_ctx* c, int width, int height, int n_ranks, int rank, const void* d_all_el, const void* d_all_dep, int n_local,
                     int item_base, void* d_age, double age_factor, double depth_threshold, void* d_item_pixel, void* d_pel, void* d_pdep,
                     hipStream_t s);

}  // namespace ptrh
