// ptr_kernels_rc.hip — the render cache's reprojection pass:
// Driver.reprojectFrame (rendercache-js/src/Driver.js:417-486) with Camera.reprojectPixel
// (rendercache-js/src/Camera.js:52-74) and the three.js Vector3.project arithmetic underneath it.
//
// The reference walks the cache sequentially and its result depends on that order: the item that ends up
// attached to a pixel is the nearest one (first in cache order among equals), every other item that lands on
// the pixel is aged once, and an item keeps a (stale) pixel pointer exactly when it was the nearest of all
// items BEFORE it on that pixel.  To reproduce that bit for bit in parallel:
//   1. rc_project_kernel — one item per lane, float64, operation for operation: depth = |hit - from|,
//      hit -> view -> clip with a perspective divide after each matrix, x/y = Math.round(ndc * half + half);
//      items that fail (depth 0) or leave the frame are aged here (+4 / +2 age factors) and get no pixel;
//   2. a stable radix sort of (pixel, item index) pairs by pixel (hipCUB): within a pixel the items stay in
//      cache order;
//   3. rc_resolve_kernel — the first lane of every pixel's run replays the reference's few steps for that pixel
//      in cache order (runs are ~cacheFactor = 4 items long).
// HBM-bound: 24 B hit + 16 B age read/write + 4 B pixel pointer per item, 12 B per pixel; no scene access.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include "ptr_device.h"
#include "ptr_kernels.h"

namespace ptr {

constexpr uint32_t RC_NO_PIXEL = 0xffffffffu;

// detachFrameFromCache (Driver.js:405-416): interior pixels only
__global__ void __launch_bounds__(256) rc_clear_kernel(RcArgs a)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x + 1, y = blockIdx.y + 1;
    if (x > a.width) return;
    const size_t p = (size_t)y * (a.width + 2) + x;
    a.pixel_element[p] = -1;
    a.pixel_depth[p] = a.depth_threshold;
    if (a.pixel_sample) a.pixel_sample[p] = 0;
}

// THREE.Vector3.applyMatrix4 (three.js r139): column-major elements, perspective divide
__device__ __forceinline__ void apply_matrix4(double& x, double& y, double& z, const double* e)
{
    const double w = 1 / (e[3] * x + e[7] * y + e[11] * z + e[15]);
    const double nx = (e[0] * x + e[4] * y + e[8] * z + e[12]) * w;
    const double ny = (e[1] * x + e[5] * y + e[9] * z + e[13]) * w;
    const double nz = (e[2] * x + e[6] * y + e[10] * z + e[14]) * w;
    x = nx; y = ny; z = nz;
}

__global__ void __launch_bounds__(256) rc_project_kernel(RcArgs a)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n_items) return;
    a.item_pixel[i] = -1;                                   // cacheItem.pixel = null (Driver.js:430)
    a.sort_val[i] = (uint32_t)i;
    uint32_t key = RC_NO_PIXEL;
    const double hx = a.hit_xyz[3 * (size_t)i], hy = a.hit_xyz[3 * (size_t)i + 1], hz = a.hit_xyz[3 * (size_t)i + 2];
    if (hx == hx) {                                         // cacheItem.hit !== null
        // Camera.reprojectPixel (Camera.js:52-74)
        const double dx = hx - a.from[0], dy = hy - a.from[1], dz = hz - a.from[2];
        const double depth = __builtin_sqrt(dx * dx + dy * dy + dz * dz);
        double x = hx, y = hy, z = hz;
        apply_matrix4(x, y, z, a.view);
        apply_matrix4(x, y, z, a.proj);
        const double half_x = 0.5 * a.width, half_y = 0.5 * a.height;
        const double rx = js_round(x * half_x + half_x), ry = js_round(y * half_y + half_y);
        if (depth == 0) {
            a.age[i] += a.age_factor * 4;                   // Driver.js:481
        } else if (rx >= 1 && rx <= a.width && ry >= 1 && ry <= a.height) {
            key = (uint32_t)((int)ry * (a.width + 2) + (int)rx);
            a.item_depth[i] = depth;
        } else {
            a.age[i] += a.age_factor * 2;                   // Driver.js:474
        }
    }
    a.sort_key[i] = key;
}

// one run of equal keys = the items of one pixel, in cache order (the sort is stable)
__global__ void __launch_bounds__(256) rc_resolve_kernel(RcArgs a)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= a.n_items) return;
    const uint32_t p = a.sorted_key[j];
    if (p == RC_NO_PIXEL || (j > 0 && a.sorted_key[j - 1] == p)) return;
    double depth = a.depth_threshold;                       // pixel.depth after detachFrameFromCache
    int element = -1;
    for (int k = j; k < a.n_items && a.sorted_key[k] == p; k++) {
        const int i = (int)a.sorted_val[k];
        const double d = a.item_depth[i];
        if (d < depth) {                                    // Driver.js:446-462
            if (element >= 0) a.age[element] += a.age_factor;
            a.item_pixel[i] = (int32_t)p;                   // Sample.attach: the pointer stays even if displaced later
            element = i;
            depth = d;
        } else {
            a.age[i] += a.age_factor;                       // Driver.js:468
        }
    }
    a.pixel_element[p] = element < 0 ? -1 : element + a.item_base;
    a.pixel_depth[p] = depth;
}

// ---- the same pass for a cache split over several GPUs in cache order (multi-GPU) ---------
// Rank r holds items [base_r, base_r + n_r) and runs the pass above on them alone.  What the sequential walk over the
// WHOLE cache would have left follows from the ranks' local grids (gathered, 12 B per pixel and rank):
//   * the pixel's element is the local winner of the first rank that reaches the smallest local depth (within a rank
//     the local winner already is the first item of that depth);
//   * every item on the pixel except that one is aged once: the local pass aged all local losers, so the only item
//     left to age is a local winner that loses against another rank;
//   * an item keeps its pixel pointer when it was nearer than everything before it in cache order: nearer than the
//     earlier items of its own rank (the local pass decided that) AND than the nearest item of all earlier ranks.
__global__ void __launch_bounds__(256) rc_merge_pixels_kernel(RcMergeArgs a)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x + 1, y = blockIdx.y + 1;
    if (x > a.width) return;
    const size_t cells = (size_t)(a.width + 2) * (a.height + 2), p = (size_t)y * (a.width + 2) + x;
    double best = a.depth_threshold, before = a.depth_threshold;
    int element = -1, winner = -1;
    for (int r = 0; r < a.n_ranks; r++) {
        if (r == a.rank) before = best;
        const int e = a.all_element[r * cells + p];
        const double d = a.all_depth[r * cells + p];
        if (e >= 0 && d < best) { best = d; element = e; winner = r; }
    }
    const int mine = a.all_element[a.rank * cells + p];
    if (mine >= 0 && winner != a.rank) a.age[mine - a.item_base] += a.age_factor;
    a.pixel_element[p] = element;
    a.pixel_depth[p] = best;
    a.before[p] = before;
}

__global__ void __launch_bounds__(256) rc_merge_items_kernel(RcMergeArgs a)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n_items) return;
    const int p = a.item_pixel[i];
    if (p >= 0 && !(a.item_depth[i] < a.before[p])) a.item_pixel[i] = -1;
}

// Keys are pixel indices of the padded grid, or all ones: with 2^bits > cells, all ones stays the largest key
// in its low `bits` bits, so the sort only has to look at those.
static int rc_key_bits(int width, int height)
{
    const long long cells = (long long)(width + 2) * (height + 2);
    int bits = 1;
    while ((1ll << bits) <= cells) bits++;
    return bits;
}

// ---- rank 3: depth culling, gap filling, frames (Driver.js:488-679, 960-1018) ---------------------------------
// Both passes are 3x3 stencils over the padded pixel grid and independent per pixel (a pixel's colour is only read
// when it has an element, and only written when it has none), so each is one kernel, one pixel per lane.
// Colours are bytes (what Sample.doRaytracing leaves) packed r | g << 8 | b << 16; priorities are integers.
__device__ __forceinline__ bool rc_border(const RcShadeArgs& a, int x, int y)
{
    return x == 0 || x == a.width + 1 || y == 0 || y == a.height + 1;
}

// Driver.depthCulling: compare a pixel's depth with the mean depth of its 4 DIAGONAL neighbours that have an element.
// This is also where the element -> item indirection is paid, once per pixel: the item's (possibly just raised) age
// and its colour are copied into per-pixel arrays, so that the gap filling below reads its 3x3 neighbourhood from
// contiguous rows instead of gathering 2 x 9 random cache items per empty pixel.
__global__ void __launch_bounds__(256) rc_cull_kernel(RcShadeArgs a)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x + 1, y = blockIdx.y + 1;
    if (x > a.width) return;
    const int stride = a.width + 2;
    const int p = y * stride + x, e = a.pixel_element[p];
    if (e < 0) return;
    double age = a.item_age[e];
    double items = 0, sum = 0;
#pragma unroll
    for (int i = -1; i <= 1; i += 2)
#pragma unroll
        for (int j = -1; j <= 1; j += 2) {
            const int q = (y + j) * stride + (x + i);
            // (a border cell may carry an element: Driver.initializeCache draws x, y from 0, and nothing detaches the border)
            if (a.pixel_element[q] >= 0) { sum += a.pixel_depth[q]; items++; }
        }
    if (items != 0) {
        const double ratio = sum / (items * a.pixel_depth[p]);
        if (ratio < 0.9 || ratio > 1.1) {                                            // Driver.js:536
            age += (a.age_factor * 10);
            a.item_age[e] = age;
        }
    }
    a.pixel_age[p] = age;
    a.pixel_rgb[p] = a.item_rgb[e];                                                  // pixel.color = cacheItem.color (:547)
}

// Driver.fillGaps + the three frame getters
__global__ void __launch_bounds__(256) rc_fill_kernel(RcShadeArgs a)
{
    __shared__ unsigned lh[3 * 256];
    __shared__ long long ltot[3];
    for (int k = threadIdx.x; k < 3 * 256; k += blockDim.x) lh[k] = 0u;
    if (threadIdx.x < 3) ltot[threadIdx.x] = 0;
    __syncthreads();
    // A workgroup walks rows y, y + gridDim.y, ...: its histogram and sums go to global memory once, and with at most
    // 128 workgroups per column of the grid the shared bins see ~1000 atomics each instead of one per 256 pixels
    // (one address takes ~88 atomics per microsecond: 8640 workgroups cost 100 us there).
    const int x = blockIdx.x * blockDim.x + threadIdx.x + 1;
    long long t_elem = 0, t_total = 0, t_member = 0;
    for (int y = blockIdx.y + 1; y <= a.height; y += gridDim.y)
    if (x <= a.width) {
        const int stride = a.width + 2;
        const int p = y * stride + x, e = a.pixel_element[p];
        uint32_t colour;
        long long prio;
        if (e < 0) {
            double age = 0, weight = 0, items = 0, cw = 0, r = 0, g = 0, b = 0;
#pragma unroll
            for (int i = -1; i <= 1; i++)
#pragma unroll
                for (int j = -1; j <= 1; j++) {
                    const int q = (y + j) * stride + (x + i);
                    const double iw = (i != 0 && j != 0) ? 1 : 2;                   // corner 1, edge and centre 2
                    if (a.pixel_element[q] >= 0) {
                        // interior neighbours were staged by the cull kernel; a border cell keeps the colour the caller gave
                        // it (pixel.color of a border pixel only changes when it is sampled) and its element's current age
                        const uint32_t c = a.pixel_rgb[q];
                        r += (double)(c & 255u) * iw; g += (double)((c >> 8) & 255u) * iw; b += (double)((c >> 16) & 255u) * iw;
                        age += rc_border(a, x + i, y + j) ? a.item_age[a.pixel_element[q]] : a.pixel_age[q];
                        cw += iw;
                        items++;
                    }
                }
            if (items > 0) {
                const uint32_t fr = (uint32_t)__builtin_floor(r / cw), fg = (uint32_t)__builtin_floor(g / cw), fb = (uint32_t)__builtin_floor(b / cw);
                colour = fr | (fg << 8) | (fb << 16);
                double pr = age / items;
                pr += (a.interpolation_zero + (weight - items) * a.interpolation_random);   // pixel.weight is 0 everywhere
                prio = (long long)js_round(pr);
                if (prio >= 0 && prio < 256) atomicAdd(&lh[prio], 1u);
                t_member += prio;                                                            // this.totalPriority
            } else {
                colour = 0u;
                prio = (long long)a.priority_max;
                if (prio >= 0 && prio < 256) atomicAdd(&lh[256 + prio], 1u);
                t_total += prio;
            }
            a.pixel_rgb[p] = colour;            // an empty pixel's colour is never read by a neighbour: no race with the reads above
        } else {
            colour = a.pixel_rgb[p];
            prio = (long long)((int32_t)js_to_uint32(a.pixel_age[p]) >> 1);                   // age >> 1
            if (prio > 0) t_total += prio;
            if (prio >= 0 && prio < 256) atomicAdd(&lh[512 + prio], 1u);
            t_elem += 1;
        }
        a.pixel_priority[p] = (int32_t)prio;
        // frames are stored bottom-up (Driver.js:965): grid row y -> frame row height - y
        const size_t at = (size_t)(a.height - y) * a.width + (x - 1);
        const uint32_t alpha = 255u << 24;
        if (a.color_frame) a.color_frame[at] = alpha | (colour & 0xffffffu);
        if (a.reprojection_frame) a.reprojection_frame[at] = e >= 0 ? (alpha | (colour & 0xffffffu)) : alpha;
        if (a.priority_frame) { const uint32_t pr = (uint32_t)(int32_t)prio; a.priority_frame[at] = alpha | (pr << 16) | (pr << 8) | pr; }
    }
    // the three sums: wave reduction, then one LDS add per wave
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        t_elem += __shfl_xor(t_elem, off, 64); t_total += __shfl_xor(t_total, off, 64); t_member += __shfl_xor(t_member, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd((unsigned long long*)&ltot[0], (unsigned long long)t_elem);
        atomicAdd((unsigned long long*)&ltot[1], (unsigned long long)t_total);
        atomicAdd((unsigned long long*)&ltot[2], (unsigned long long)t_member);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < 3 * 256; k += blockDim.x)
        if (lh[k]) atomicAdd(&a.hist[k], lh[k]);
    if (threadIdx.x < 3 && ltot[threadIdx.x]) atomicAdd((unsigned long long*)&a.totals[threadIdx.x], (unsigned long long)ltot[threadIdx.x]);
}

hipError_t launch_rc_shade(const RcShadeArgs& a, hipStream_t s)
{
    hipError_t e;
    if (reinterpret_cast<unsigned char*>(a.totals) == reinterpret_cast<unsigned char*>(a.hist) + 3 * 256 * sizeof(uint32_t)) {
        e = hipMemsetAsync(a.hist, 0, 3 * 256 * sizeof(uint32_t) + 3 * sizeof(long long), s);      // (one small fill, not two)
        if (e != hipSuccess) return e;
    } else {
        e = hipMemsetAsync(a.hist, 0, 3 * 256 * sizeof(uint32_t), s);
        if (e != hipSuccess) return e;
        e = hipMemsetAsync(a.totals, 0, 3 * sizeof(long long), s);
        if (e != hipSuccess) return e;
    }
    dim3 grid((a.width + 255) / 256, a.height);
    hipLaunchKernelGGL(rc_cull_kernel, grid, dim3(256), 0, s, a);
    dim3 fill_grid(grid.x, a.height < 128 ? a.height : 128);
    hipLaunchKernelGGL(rc_fill_kernel, fill_grid, dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_rc_merge(const RcMergeArgs& a, hipStream_t s)
{
    hipLaunchKernelGGL(rc_merge_pixels_kernel, dim3((a.width + 255) / 256, a.height), dim3(256), 0, s, a);
    if (a.n_items > 0) hipLaunchKernelGGL(rc_merge_items_kernel, dim3((a.n_items + 255) / 256), dim3(256), 0, s, a);
    return hipGetLastError();
}

size_t rc_sort_temp_bytes(int n_items, int width, int height)
{
    size_t bytes = 0;
    // (a size query: no work is launched; on an error the caller's allocation of 0 bytes makes the sort itself fail loudly)
    if (hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const uint32_t*)nullptr,
                                           (uint32_t*)nullptr, n_items, 0, rc_key_bits(width, height)) != hipSuccess)
        return 0;
    return bytes;
}

hipError_t launch_rc_reproject(const RcArgs& a, void* sort_temp, size_t sort_temp_bytes, hipStream_t s)
{
    if (a.n_items <= 0) {
        hipLaunchKernelGGL(rc_clear_kernel, dim3((a.width + 255) / 256, a.height), dim3(256), 0, s, a);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(rc_clear_kernel, dim3((a.width + 255) / 256, a.height), dim3(256), 0, s, a);
    const int blocks = (a.n_items + 255) / 256;
    hipLaunchKernelGGL(rc_project_kernel, dim3(blocks), dim3(256), 0, s, a);
    hipError_t e = hipcub::DeviceRadixSort::SortPairs(sort_temp, sort_temp_bytes, a.sort_key, a.sorted_key, a.sort_val, a.sorted_val,
                                                      a.n_items, 0, rc_key_bits(a.width, a.height), s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(rc_resolve_kernel, dim3(blocks), dim3(256), 0, s, a);
    return hipGetLastError();
}

}  // namespace ptr
