// ptr_kernels_rcloop.hip — what the render cache's frame loop (rendercache-js/src/Driver.js:95-115) does between the
// reprojection / shading passes (ptr_kernels_rc.hip) and the trace of the requested samples, on a cache and a pixel grid
// that stay in HBM:
//   Driver.directSamples (:682-827)   threshold over the three priority histograms, the serpentine candidate walk,
//                                     the picks at a fixed stride, the error diffusion of distributeExceedingPriority
//                                     (:829-839) with the stale pixel it is handed (:813-814)
//   Driver.addRequest / allocCacheItem (:252-326)   the paged allocator: 8 items per page, first empty one or the oldest
//   Driver.requestSamples (:842-863)  ray directions (Camera.computeDirToXY, Camera.js:39-50), results into the cache
//   Driver.age (:936-942), getCacheUsage, the four frame getters (:960-1046)
// The reference does all of this sequentially; the order only matters in three places, and each has a parallel form
// that gives the same state:
//   * the candidate list is a stable compaction of the pixels in walk order: count per 2 048 walk steps, one workgroup's
//     scan of the counts, write (the library's select did the same with two small fills and a threshold kernel before it);
//   * the picks are `Math.round` of a running float64 sum, which is piecewise linear in integers (RcPickSegments): the
//     host finds the few pieces, every lane evaluates its own step;
//   * the diffused amount depends on the priority of ONE pixel, the last of the walk, which only two picks can change
//     (its upper and its inner neighbour): three amounts, computed by one lane, cover every pick;
//   * the allocator hands the k-th request that needs an item the k-th page after the pointer: an exclusive sum of the
//     need flags gives k, every page is then looked at by one lane (launches of at most `pages` requests each, so that a
//     page is never looked at twice in one launch).
// All of it is integer / byte / float64 work on a few words per request; nothing here is near a roofline of its own.
#include <hip/hip_runtime.h>
#include <climits>

#include "ptr_device.h"
#include "ptr_kernels.h"

namespace ptr {

namespace {

__device__ __forceinline__ int rc_last_pixel(const RcLoopArgs& a)      // where the walk ends: row `height`
{
    return a.height * (a.width + 2) + ((a.height & 1) ? 1 : a.width);
}

constexpr int RC_CHUNK = 2048;                                  // walk steps per workgroup of the compaction (256 lanes x 8)

// exclusive prefix sum of one int per lane over a 256-lane workgroup (4 waves); *total = the workgroup's sum
__device__ __forceinline__ int rc_block_exclusive(int v, int* wave_sums /* 4, shared */, int* total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int u = __shfl_up(inc, off, 64); if (lane >= off) inc += u; }
    if (lane == 63) wave_sums[wave] = inc;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; w++) base += wave_sums[w];
    *total = wave_sums[0] + wave_sums[1] + wave_sums[2] + wave_sums[3];
    __syncthreads();                                            // (wave_sums may be written again by the caller's next scan)
    return base + inc - v;
}

// k-th pixel of the walk (Driver.js:713-760): rows 1..height, even rows left to right, odd rows right to left
__device__ __forceinline__ int rc_walk_pixel(int k, int width)
{
    const int y = k / width + 1, j = k - (y - 1) * width;
    return y * (width + 2) + ((y & 1) ? width - j : 1 + j);
}

// Driver.js:690-706 by a whole workgroup: lane i holds priority level i (URGENT + INTERPOLATED + SAMPLED).  The reference
// walks the levels downwards and stops at the first whose running sum exceeds the budget (`samples === null` only makes the
// first level its own start); the sums are of integers, exact in float64 in any order, so: C_i = sum of the levels >= i,
// threshold = the largest i with C_i > budget (null if none), samples = C_threshold (C_0 if none).
__device__ __forceinline__ void rc_find_threshold(const RcLoopArgs& a, double* lev /* 256, shared */, int* s_thr /* shared */,
                                                  int& threshold, double& samples)
{
    const int i = threadIdx.x;
    lev[i] = i <= a.priority_max ? (double)a.hist[256 + i] + (double)a.hist[i] + (double)a.hist[512 + i] : 0.0;
    if (i == 0) *s_thr = -1;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {                   // suffix sums
        const double u = i + off < 256 ? lev[i + off] : 0.0;
        __syncthreads();
        lev[i] += u;
        __syncthreads();
    }
    if (i <= a.priority_max && lev[i] > a.max_samples) atomicMax(s_thr, i);
    __syncthreads();
    threshold = *s_thr;
    samples = lev[threshold < 0 ? 0 : threshold];
}

// candidates, pass 1: every workgroup finds the threshold for itself (the first one records it) and counts its 2 048 walk steps
__global__ void __launch_bounds__(256) rc_cand_count_kernel(RcLoopArgs a, int32_t* block_count)
{
    __shared__ double lev[256];
    __shared__ int s_thr, wave_sums[4];
    int threshold; double samples;
    rc_find_threshold(a, lev, &s_thr, threshold, samples);
    const int thr_value = threshold < 0 ? 0 : threshold;        // what `>= threshold` sees: null compares as 0
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        RcLoopWords& w = *a.words;
        w.samples = samples;
        w.threshold = threshold;
        w.thr_value = thr_value;
        w.candidates = 0;
        w.k_a = INT_MAX; w.k_b = INT_MAX;
        w.half[0] = w.half[1] = w.half[2] = -1;
        w.n_need = 0;
        w.used = 0;
    }
    const int n = a.width * a.height, k0 = (blockIdx.x * 256 + threadIdx.x) * 8;
    int c = 0;
    for (int j = 0; j < 8; j++)
        if (k0 + j < n) c += a.pixel_priority[rc_walk_pixel(k0 + j, a.width)] >= thr_value ? 1 : 0;
    int total;
    (void)rc_block_exclusive(c, wave_sums, &total);
    if (threadIdx.x == 0) block_count[blockIdx.x] = total;
}

// pass 2 (one workgroup): counts -> offsets, in place; the total is the number of candidates
__global__ void __launch_bounds__(256) rc_cand_offsets_kernel(int32_t* block_count, int n_blocks, RcLoopWords* words)
{
    __shared__ int wave_sums[4];
    int carry = 0;
    for (int b0 = 0; b0 < n_blocks; b0 += 256) {
        const int b = b0 + threadIdx.x;
        const int v = b < n_blocks ? block_count[b] : 0;
        int total;
        const int ex = rc_block_exclusive(v, wave_sums, &total);
        if (b < n_blocks) block_count[b] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) words->candidates = carry;
}

// pass 3: the candidates, in walk order
__global__ void __launch_bounds__(256) rc_cand_write_kernel(RcLoopArgs a, const int32_t* block_off)
{
    __shared__ int wave_sums[4];
    const int thr_value = a.words->thr_value;
    const int n = a.width * a.height, k0 = (blockIdx.x * 256 + threadIdx.x) * 8;
    int px[8], c = 0;
    unsigned flags = 0u;
    for (int j = 0; j < 8; j++) {
        px[j] = k0 + j < n ? rc_walk_pixel(k0 + j, a.width) : 0;
        if (k0 + j < n && a.pixel_priority[px[j]] >= thr_value) { flags |= 1u << j; c++; }
    }
    int total;
    int at = block_off[blockIdx.x] + rc_block_exclusive(c, wave_sums, &total);
    for (int j = 0; j < 8; j++)
        if (flags >> j & 1u) a.cand[at++] = px[j];
}

// which candidate step k requests: Math.round of the running sum (see RcPickSegments)
__global__ void __launch_bounds__(256) rc_pick_index_kernel(int32_t* pick_index, int n_picks, RcPickSegments g)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_picks) return;
    int seg = 0;
    while (seg + 1 < g.n && k >= g.k0[seg + 1]) seg++;
    const long long S = g.S0[seg] + (long long)(k - g.k0[seg]) * g.D[seg];
    pick_index[k] = (int32_t)js_round(__builtin_scalbn((double)S, g.e[seg]));
}

__global__ void __launch_bounds__(256) rc_pick_kernel(RcLoopArgs a, int32_t* block_need)
{
    __shared__ int wave_sums[4];
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    int need = 0;
    if (k < a.n_picks) {
        const int st = a.width + 2, p = a.cand[a.pick_index[k]];
        a.pixel_sample[p] = 1;                                  // addRequest (:312)
        need = a.pixel_element[p] < 0 ? 1 : 0;
        a.need[k] = need;
        const int last = rc_last_pixel(a), dx_last = (a.height & 1) ? -1 : 1;
        if (p == last - st) a.words->k_a = k;                   // its lower neighbour is the walk's last pixel
        if (p == last - dx_last) a.words->k_b = k;              // its neighbour in walk direction is
    }
    int total;
    (void)rc_block_exclusive(need, wave_sums, &total);          // this workgroup's requests without an item
    if (threadIdx.x == 0) block_need[blockIdx.x] = total;
}

// one workgroup: the workgroups' need counts -> offsets (in place), their total = the requests without an item; then the
// three diffused amounts
__global__ void __launch_bounds__(256) rc_amounts_kernel(RcLoopArgs a, int32_t* block_need, int n_blocks)
{
    __shared__ int wave_sums[4];
    int carry = 0;
    for (int b0 = 0; b0 < n_blocks; b0 += 256) {
        const int b = b0 + threadIdx.x;
        const int v = b < n_blocks ? block_need[b] : 0;
        int total;
        const int ex = rc_block_exclusive(v, wave_sums, &total);
        if (b < n_blocks) block_need[b] = carry + ex;
        carry += total;
    }
    if (threadIdx.x != 0) return;
    RcLoopWords& w = *a.words;
    int pr = a.pixel_priority[rc_last_pixel(a)];
    const int thr = w.thr_value;
    const int h0 = (pr - thr) >> 1;                             // Math.floor((priority - threshold) / 2.0), integers
    if (w.k_a != INT_MAX && h0 >= 0) pr += h0;
    const int h1 = (pr - thr) >> 1;
    if (w.k_b != INT_MAX && h1 >= 0) pr += h1;
    const int h2 = (pr - thr) >> 1;
    w.half[0] = h0; w.half[1] = h1; w.half[2] = h2;
    w.n_need = carry;
}

// ... and, with the same lane-to-request mapping as rc_pick_kernel, every request's number among those without an item
__global__ void __launch_bounds__(256) rc_diffuse_kernel(RcLoopArgs a, const int32_t* block_need_off)
{
    __shared__ int wave_sums[4];
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = k < a.n_picks;
    int total;
    const int ex = rc_block_exclusive(live ? a.need[k] : 0, wave_sums, &total);
    if (!live) return;
    a.rank[k] = block_need_off[blockIdx.x] + ex;
    const RcLoopWords& w = *a.words;
    const int st = a.width + 2, p = a.cand[a.pick_index[k]];
    const int half = w.half[k > w.k_b ? 2 : k > w.k_a ? 1 : 0];
    if (half < 0) return;                                       // :831
    const int dx = ((p / st) & 1) ? -1 : 1;
    atomicAdd(&a.pixel_priority[p + dx], half);
    atomicAdd(&a.pixel_priority[p + st], half);
}

// Driver.allocCacheItem (:252-305) for the requests whose number among the needy ones is in [rank_lo, rank_hi)
__global__ void __launch_bounds__(256) rc_alloc_kernel(RcLoopArgs a)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= a.n_picks || !a.need[k]) return;
    const int r = a.rank[k];
    if (r < a.rank_lo || r >= a.rank_hi) return;
    const int p = a.cand[a.pick_index[k]];
    const int base = (int)(((long long)a.words->page_next + r) % a.pages) * 8;
    int take = -1, oldest = -1;
    double max_age = -1;
    for (int i = base; i < base + 8; i++) {
        if (a.item_pixel[i] < 0) { take = i; break; }           // an empty item: take it
        const double age = a.item_age[i];
        if (age > max_age) { max_age = age; oldest = i; }
    }
    if (take < 0) {
        if (oldest < 0) return;                                 // (no attached item is younger than 0; the reference would store at index -1)
        take = oldest;                                          // Sample.clear(): the pixel that had it keeps pointing at it
        a.item_hit[3 * (size_t)take] = __builtin_nan("");
        a.item_age[take] = -1;
        a.item_rgb[take] = 0u;
    }
    a.item_pixel[take] = p;
    a.pixel_element[p] = take;
}

// Sample.doRaytracing's direction (Camera.computeDirToXY, Camera.js:39-50, THREE.Raycaster.setFromCamera underneath)
__global__ void __launch_bounds__(256) rc_dirs_kernel(RcLoopArgs a)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= a.n_picks) return;
    if (k == 0) a.words->page_next = (int)(((long long)a.words->page_next + a.words->n_need) % a.pages);
    const int st = a.width + 2, p = a.cand[a.pick_index[k]];
    const int it = a.pixel_element[p], pp = it >= 0 ? a.item_pixel[it] : -1;
    double vx = 0, vy = 0, vz = -1;
    if (pp >= 0) {
        const int x = pp % st, y = pp / st;
        vx = ((double)x / (double)a.width) * 2 - 1; vy = ((double)y / (double)a.height) * 2 - 1; vz = 0.5;
#pragma unroll
        for (int m = 0; m < 2; m++) {                           // Vector3.applyMatrix4: projectionMatrixInverse, then matrixWorld
            const double* e = m == 0 ? a.proj_inv : a.world;
            const double w = 1 / (e[3] * vx + e[7] * vy + e[11] * vz + e[15]);
            const double nx = (e[0] * vx + e[4] * vy + e[8] * vz + e[12]) * w;
            const double ny = (e[1] * vx + e[5] * vy + e[9] * vz + e[13]) * w;
            const double nz = (e[2] * vx + e[6] * vy + e[10] * vz + e[14]) * w;
            vx = nx; vy = ny; vz = nz;
        }
        vx -= a.world[12]; vy -= a.world[13]; vz -= a.world[14];
        double len = __builtin_sqrt(vx * vx + vy * vy + vz * vz);
        if (!(len != 0)) len = 1;                               // `Math.sqrt(..) || 1`: 0 and NaN are falsy
        const double s = 1 / len;
        vx *= s; vy *= s; vz *= s;
        const double len2 = vx * vx + vy * vy + vz * vz;        // the cache's own Vector3.normalize (Vector3.js:23-33)
        if (len2 > 0) { const double inv = 1 / __builtin_sqrt(len2); vx *= inv; vy *= inv; vz *= inv; }
    }
    a.pick_item[k] = pp >= 0 ? it : -1;                         // requestSamples only traces items that have a pixel
    a.dirs[3 * (size_t)k] = vx; a.dirs[3 * (size_t)k + 1] = vy; a.dirs[3 * (size_t)k + 2] = vz;
}

// Driver.requestSamples (:842-863): the traced colour as bytes, the hit point, and the colour of the item's pixel.
// An item requested twice (it was taken from a pixel requested earlier in the frame) gets the same values twice.
__global__ void __launch_bounds__(256) rc_store_kernel(RcLoopArgs a)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= a.n_picks) return;
    const int it = a.pick_item[k];
    if (it < 0) return;
    uint32_t v = 0;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        v |= (js_to_uint32(js_round(js_min1(a.tr_rgb[3 * (size_t)k + c]) * 255.0)) & 255u) << (8 * c);
        a.item_hit[3 * (size_t)it + c] = a.tr_hit[3 * (size_t)k + c];
    }
    a.item_rgb[it] = v;
    a.pixel_rgb[a.item_pixel[it]] = v;
}

// age(ageFactor, cache) (:113, :936-942) and getCacheUsage's count
__global__ void __launch_bounds__(256) rc_age_kernel(RcLoopArgs a)
{
    __shared__ int block_used;
    if (threadIdx.x == 0) block_used = 0;
    __syncthreads();
    int used = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < a.n_items; i += gridDim.x * blockDim.x) {
        a.item_age[i] += a.age_factor;
        used += a.item_pixel[i] >= 0 ? 1 : 0;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) used += __shfl_xor(used, off, 64);
    if ((threadIdx.x & 63) == 0 && used) atomicAdd(&block_used, used);
    __syncthreads();
    if (threadIdx.x == 0 && block_used) atomicAdd(&a.words->used, block_used);      // one address: ~88 atomics per microsecond
}

// getColorFrame / getReprojectionFrame / getPriorityFrame / getSamplingFrame (:960-1046) from the live state, rows bottom-up
__global__ void __launch_bounds__(256) rc_frames_kernel(RcLoopArgs a)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x + 1, y = blockIdx.y + 1;
    if (x > a.width) return;
    const int p = y * (a.width + 2) + x;
    const size_t at = (size_t)(a.height - y) * a.width + (x - 1);
    const uint32_t alpha = 255u << 24;
    if (a.frames[0]) a.frames[0][at] = alpha | a.pixel_rgb[p];
    if (a.frames[1]) { const int e = a.pixel_element[p]; a.frames[1][at] = alpha | (e >= 0 ? a.item_rgb[e] : 0u); }
    if (a.frames[2]) { const uint32_t v = (uint32_t)a.pixel_priority[p]; a.frames[2][at] = alpha | (v << 16) | (v << 8) | v; }
    if (a.frames[3]) a.frames[3][at] = a.pixel_sample[p] ? 0xffffffffu : 0xff000000u;
}

inline dim3 blocks_for(int n) { return dim3((unsigned)((n + 255) / 256)); }

}  // namespace

// scratch of the candidate compaction (a count per 2 048 walk steps) and of the requests' ranks (a count per 256 requests)
size_t rc_loop_temp_bytes(int width, int height)
{
    const size_t px = (size_t)width * (size_t)height;
    return ((px + RC_CHUNK - 1) / RC_CHUNK + (px + 255) / 256 + 64) * sizeof(int32_t);
}

hipError_t launch_rc_candidates(const RcLoopArgs& a, void* temp, size_t temp_bytes, hipStream_t s)
{
    const int n = a.width * a.height, blocks = (n + RC_CHUNK - 1) / RC_CHUNK;
    if (temp_bytes < rc_loop_temp_bytes(a.width, a.height)) return hipErrorInvalidValue;
    int32_t* counts = static_cast<int32_t*>(temp);
    hipLaunchKernelGGL(rc_cand_count_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a, counts);
    hipLaunchKernelGGL(rc_cand_offsets_kernel, dim3(1), dim3(256), 0, s, counts, blocks, a.words);
    hipLaunchKernelGGL(rc_cand_write_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a, (const int32_t*)counts);
    return hipGetLastError();
}

hipError_t launch_rc_requests(const RcLoopArgs& a, const RcPickSegments& picks, void* temp, size_t temp_bytes, hipStream_t s)
{
    if (a.n_picks <= 0) return hipSuccess;
    if (temp_bytes < rc_loop_temp_bytes(a.width, a.height)) return hipErrorInvalidValue;
    const dim3 grid = blocks_for(a.n_picks);
    int32_t* block_need = static_cast<int32_t*>(temp) + (a.width * a.height + RC_CHUNK - 1) / RC_CHUNK;
    if (picks.n > 0) hipLaunchKernelGGL(rc_pick_index_kernel, grid, dim3(256), 0, s, a.pick_index, a.n_picks, picks);   // else: the host sent them
    hipLaunchKernelGGL(rc_pick_kernel, grid, dim3(256), 0, s, a, block_need);
    hipLaunchKernelGGL(rc_amounts_kernel, dim3(1), dim3(256), 0, s, a, block_need, (int)grid.x);
    hipLaunchKernelGGL(rc_diffuse_kernel, grid, dim3(256), 0, s, a, (const int32_t*)block_need);
    for (long long lo = 0; lo < a.n_picks; lo += a.pages) {     // one launch unless a frame requests more items than there are pages
        RcLoopArgs b = a;
        b.rank_lo = (int)lo;
        b.rank_hi = (int)(lo + a.pages < (long long)a.n_picks ? lo + a.pages : (long long)a.n_picks);
        hipLaunchKernelGGL(rc_alloc_kernel, grid, dim3(256), 0, s, b);
    }
    hipLaunchKernelGGL(rc_dirs_kernel, grid, dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_rc_store(const RcLoopArgs& a, hipStream_t s)
{
    if (a.n_picks > 0) hipLaunchKernelGGL(rc_store_kernel, blocks_for(a.n_picks), dim3(256), 0, s, a);
    const int blocks = (a.n_items + 255) / 256;
    hipLaunchKernelGGL(rc_age_kernel, dim3((unsigned)(blocks < 1024 ? blocks : 1024)), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_rc_frames(const RcLoopArgs& a, hipStream_t s)
{
    hipLaunchKernelGGL(rc_frames_kernel, dim3((unsigned)((a.width + 255) / 256), (unsigned)a.height), dim3(256), 0, s, a);
    return hipGetLastError();
}

}  // namespace ptr
