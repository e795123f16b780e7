// ptr_kernels.h — launch functions the C-ABI host code (ptr_api.hip) calls.
#pragma once
#include <hip/hip_runtime.h>
#include "ptr_device.h"

namespace ptr {

// float64, op-for-op the reference (ptr_kernels_exact.hip)
hipError_t launch_render_exact(const RenderArgs& a, int n_frames, bool stats, hipStream_t s);
hipError_t launch_retrace_exact(const RenderArgs& a, int parts1, int parts0, hipStream_t s);   // workgroups per sub-queue: shading-only entries / full re-traces
hipError_t launch_trace_batch_exact(const TraceArgs& a, hipStream_t s);

// float32 with uncertainty filter (ptr_kernels_fast.hip); mode 0: float32 only, 1: hybrid (queue every doubtful pixel: bytes
// equal the reference's), 2: hybrid within one unit per channel (PTR_MODE_TOLERANT: queue the discontinuous doubts only)
hipError_t launch_render_fast(const RenderArgs& a, int n_frames, int mode, hipStream_t s);
int fast_kernel_ppl(int mode);      // pixels per lane of the float32 kernel in that mode
int fast_kernel_rows_per_block();   // image rows / columns one workgroup of the float32 kernel covers
int fast_kernel_cols_per_block(int mode);

// float32 for many spheres: sphere list staged through LDS, two-level tests (ptr_kernels_many.hip)
hipError_t launch_render_many(const RenderArgs& a, int n_frames, bool hybrid, hipStream_t s);
int many_kernel_rows_per_block();

// extension modes: reflection/refraction to max_depth, Philox-jittered spp (ptr_kernels_ext.hip)
hipError_t launch_render_ext(const RenderArgs& a, int n_frames, bool f64, bool stats, bool spheres_only, bool lanewise, hipStream_t s);   // spheres_only: no planes, no Phong terms; lanewise: PTR_DEBUG_EXT_LANEWISE

// render cache reprojection pass (ptr_kernels_rc.hip)
hipError_t launch_rc_shade(const RcShadeArgs& a, hipStream_t s);
size_t rc_sort_temp_bytes(int n_items, int width, int height);
hipError_t launch_rc_reproject(const RcArgs& a, void* sort_temp, size_t sort_temp_bytes, hipStream_t s);
hipError_t launch_rc_merge(const RcMergeArgs& a, hipStream_t s);      // the cache split over several GPUs: after the gather

// render cache frame loop on a resident cache (ptr_kernels_rcloop.hip): threshold + candidate walk; picks, diffusion,
// allocator, ray directions; results into the cache + ageing; the four frames
size_t rc_loop_temp_bytes(int width, int height);
hipError_t launch_rc_candidates(const RcLoopArgs& a, void* temp, size_t temp_bytes, hipStream_t s);
hipError_t launch_rc_requests(const RcLoopArgs& a, const RcPickSegments& picks, void* temp, size_t temp_bytes, hipStream_t s);
hipError_t launch_rc_store(const RcLoopArgs& a, hipStream_t s);
hipError_t launch_rc_frames(const RcLoopArgs& a, hipStream_t s);

// multi-GPU gather: one part's compact rows -> their places in the whole frames (ptr_kernels_gather.hip)
hipError_t launch_gather_rows(const GatherArgs& a, hipStream_t s);
// zero a run of 32-bit words by a kernel (graph-capture safe where a memset node is not; ptr_kernels_gather.hip)
hipError_t launch_clear_words(void* p, size_t bytes, hipStream_t s);

}  // namespace ptr
