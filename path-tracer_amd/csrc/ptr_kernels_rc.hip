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
    a.pixel_element[p] = element;
    a.pixel_depth[p] = depth;
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

size_t rc_sort_temp_bytes(int n_items, int width, int height)
{
    size_t bytes = 0;
    hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr,
                                       (const uint32_t*)nullptr, (uint32_t*)nullptr, n_items, 0, rc_key_bits(width, height));
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
