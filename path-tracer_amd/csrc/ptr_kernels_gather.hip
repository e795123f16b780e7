// ptr_kernels_gather.hip — the copy kernel that ends a multi-GPU gather: the rows of one part of the cyclic strip
// partition, as a rank rendered them (compact, frame after frame), go to their places in the whole frames on the root GPU.
// HBM-bound by nature: 4 bytes read + 4 written per pixel, 16 bytes per lane and access where the width allows it.
#include <hip/hip_runtime.h>
#include "ptr_device.h"
#include "ptr_kernels.h"

namespace ptr {

template <typename T>      // uint4: four pixels per lane (width a multiple of 4, buffers 16-byte aligned); uint32_t: one
__global__ void __launch_bounds__(256) gather_rows_kernel(GatherArgs a)
{
    const int row = blockIdx.y, frame = blockIdx.z;               // local row of this part, frame
    const int per = (int)(sizeof(T) / sizeof(uint32_t));
    const int x = (blockIdx.x * 256 + threadIdx.x) * per;
    if (x >= a.width) return;
    const int ls = row / a.strip_h;
    const int y = (ls * a.n_parts + a.part) * a.strip_h + (row - ls * a.strip_h);    // global_y of ptr_device.h, start_y 0
    if (y >= a.height) return;
    const T* src = reinterpret_cast<const T*>(a.src + (uint64_t)frame * a.src_frame_stride_px + (uint64_t)row * a.width + x);
    T* dst = reinterpret_cast<T*>(a.full + ((uint64_t)frame * a.height + y) * (uint64_t)a.width + x);
    *dst = *src;
}

hipError_t launch_gather_rows(const GatherArgs& a, hipStream_t s)
{
    if (a.rows <= 0 || a.n_frames <= 0) return hipSuccess;
    const bool wide = a.width % 4 == 0 && a.src_frame_stride_px % 4 == 0 &&
                      (reinterpret_cast<uintptr_t>(a.src) | reinterpret_cast<uintptr_t>(a.full)) % 16 == 0;
    const int per = wide ? 4 : 1;
    dim3 grid((a.width + 256 * per - 1) / (256 * per), a.rows, a.n_frames);
    if (wide) hipLaunchKernelGGL(gather_rows_kernel<uint4>, grid, dim3(256), 0, s, a);
    else      hipLaunchKernelGGL(gather_rows_kernel<uint32_t>, grid, dim3(256), 0, s, a);
    return hipGetLastError();
}

// Zeroes n_words 32-bit words (16 bytes per lane where the count allows it).  ptr_render_device_noalloc clears its counter
// and statistics words with this kernel rather than with hipMemsetAsync: recorded into a HIP graph, a memset node of this
// size left garbage in the words from the second replay on (ROCm 7.0 / 7.2, measured: tools/diag_capture.py); a kernel node
// replays as it was recorded.
__global__ void __launch_bounds__(256) clear_words_kernel(uint32_t* p, uint32_t n_words)
{
    const uint32_t k = (blockIdx.x * 256u + threadIdx.x) * 4u;
    if (k + 4u <= n_words) *reinterpret_cast<uint4*>(p + k) = make_uint4(0u, 0u, 0u, 0u);
    else for (uint32_t j = k; j < n_words; j++) p[j] = 0u;
}

hipError_t launch_clear_words(void* p, size_t bytes, hipStream_t s)          // p 16-byte aligned, bytes a multiple of 4
{
    const uint32_t n_words = (uint32_t)(bytes / 4);
    if (n_words == 0) return hipSuccess;
    hipLaunchKernelGGL(clear_words_kernel, dim3((n_words + 1023u) / 1024u), dim3(256), 0, s, static_cast<uint32_t*>(p), n_words);
    return hipGetLastError();
}

}  // namespace ptr
