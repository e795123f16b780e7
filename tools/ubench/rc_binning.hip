// rc_binning.hip — a working prototype of the OTHER design for the render cache's reprojection pass: no
// global sort; a coarse pass bins the items by screen tile (32 x 16 pixels), one workgroup per tile then orders its items
// in LDS and replays the reference's per-pixel steps (Driver.reprojectFrame, rendercache-js/src/Driver.js:417-486).
// Measured here, kernel by kernel, on the bench's cache (1920 x 1080 x 4 items, 80 % with a hit, random positions) and
// checked against a sequential model of the reference's walk; the product keeps the sort-based pass
// (ptr_kernels_rc.hip) — profiles/r03_rc_reproject.md has the table and the reasons.
//   hipcc -O3 --offload-arch=gfx950 -o tools/ubench/rc_binning tools/ubench/rc_binning.hip && tools/ubench/rc_binning
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int W = 1920, H = 1080, FACTOR = 4;
#ifndef RC_TILE_W
#define RC_TILE_W 32
#define RC_TILE_H 16
#define RC_CAP 3072
#endif
#ifndef RC_BIN_ITEMS
#define RC_BIN_ITEMS 8
#endif
constexpr int TILE_W = RC_TILE_W, TILE_H = RC_TILE_H, TILE_PX = TILE_W * TILE_H;
constexpr int TILES_X = (W + TILE_W - 1) / TILE_W, TILES_Y = (H + TILE_H - 1) / TILE_H, N_TILES = TILES_X * TILES_Y;
constexpr int CAP = RC_CAP;                       // items of a tile that fit in LDS (16 B each + list links)
constexpr int BIN_THREADS = 1024, BIN_ITEMS = RC_BIN_ITEMS;
constexpr double THRESHOLD = 1e6, AGE_FACTOR = 1.0;

struct Cam { double tx, ty; };                  // pinhole at the origin looking down -z: ndc = (x / -z) / tx, (y / -z) / ty
struct Rec { uint32_t idx, lp; double depth; };

// the projection of one item: -> pixel (1-based, as the reference's grid with its 1-pixel frame) or -1 / -2 (no hit / outside)
__host__ __device__ inline int project(const double* hit, int i, Cam c, int& px, int& py, double& depth)
{
    const double hx = hit[3 * (size_t)i], hy = hit[3 * (size_t)i + 1], hz = hit[3 * (size_t)i + 2];
    if (!(hx == hx)) return -1;
    depth = sqrt(hx * hx + hy * hy + hz * hz);
    const double w = 1.0 / -hz;
    const double rx = floor((hx * w / c.tx) * (0.5 * W) + 0.5 * W + 0.5), ry = floor((hy * w / c.ty) * (0.5 * H) + 0.5 * H + 0.5);
    if (!(rx >= 1 && rx <= W && ry >= 1 && ry <= H)) return -2;
    px = (int)rx; py = (int)ry;
    return 0;
}

struct Item { int tile; uint32_t lp; double depth; };
__device__ inline Item bin_item(const double* hit, int i, int n, Cam c, double* age, int32_t* item_pixel, bool write)
{
    Item it{-1, 0u, 0.0};
    if (i >= n) return it;
    int px, py;
    const int r = project(hit, i, c, px, py, it.depth);
    if (r == 0) {
        it.tile = ((py - 1) / TILE_H) * TILES_X + (px - 1) / TILE_W;
        it.lp = (uint32_t)(((py - 1) % TILE_H) * TILE_W + (px - 1) % TILE_W);
        if (write) item_pixel[i] = py * (W + 2) + px;        // the pointer every item starts with; the tile pass takes it back
    } else if (write) {
        item_pixel[i] = -1;
        if (r == -2) age[i] += AGE_FACTOR * 2;
    }
    return it;
}

// pass 1: items per tile (LDS histogram per workgroup, one global add per non-empty bin)
__global__ void __launch_bounds__(BIN_THREADS) count_kernel(const double* hit, int n, Cam c, uint32_t* tile_count)
{
    __shared__ uint32_t hist[N_TILES];
    for (int b = threadIdx.x; b < N_TILES; b += BIN_THREADS) hist[b] = 0u;
    __syncthreads();
    const int base = blockIdx.x * (BIN_THREADS * BIN_ITEMS);
#pragma unroll
    for (int k = 0; k < BIN_ITEMS; k++) {
        const Item it = bin_item(hit, base + k * BIN_THREADS + threadIdx.x, n, c, nullptr, nullptr, false);
        if (it.tile >= 0) atomicAdd(&hist[it.tile], 1u);
    }
    __syncthreads();
    for (int b = threadIdx.x; b < N_TILES; b += BIN_THREADS)
        if (hist[b]) atomicAdd(&tile_count[b], hist[b]);
}

// exclusive scan of the tile counts (one workgroup), the cursors start at the tile starts
__global__ void __launch_bounds__(1024) scan_kernel(const uint32_t* tile_count, uint32_t* tile_start, uint32_t* cursor, uint32_t* overflow)
{
    __shared__ uint32_t part[1024];
    constexpr int PER = (N_TILES + 1023) / 1024;
    uint32_t local[PER], sum = 0u;
    for (int k = 0; k < PER; k++) {
        const int b = threadIdx.x * PER + k;
        local[k] = b < N_TILES ? tile_count[b] : 0u;
        if (local[k] > (uint32_t)CAP) atomicAdd(overflow, 1u);
        sum += local[k];
    }
    part[threadIdx.x] = sum;
    __syncthreads();
    for (int d = 1; d < 1024; d *= 2) {
        const uint32_t v = threadIdx.x >= (unsigned)d ? part[threadIdx.x - d] : 0u;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = part[threadIdx.x] - sum;
    for (int k = 0; k < PER; k++) {
        const int b = threadIdx.x * PER + k;
        if (b < N_TILES) { tile_start[b] = run; cursor[b] = run; }
        run += local[k];
    }
}

// pass 2: the records go to their tiles (order inside a tile: arbitrary; the tile pass orders them by cache index)
__global__ void __launch_bounds__(BIN_THREADS) scatter_kernel(const double* hit, int n, Cam c, double* age, int32_t* item_pixel,
                                                              uint32_t* cursor, Rec* rec)
{
    __shared__ uint32_t hist[N_TILES];
    for (int b = threadIdx.x; b < N_TILES; b += BIN_THREADS) hist[b] = 0u;
    __syncthreads();
    const int base = blockIdx.x * (BIN_THREADS * BIN_ITEMS);
    Item it[BIN_ITEMS];
    uint32_t rank[BIN_ITEMS];
#pragma unroll
    for (int k = 0; k < BIN_ITEMS; k++) {
        it[k] = bin_item(hit, base + k * BIN_THREADS + threadIdx.x, n, c, age, item_pixel, true);
        rank[k] = it[k].tile >= 0 ? atomicAdd(&hist[it[k].tile], 1u) : 0u;
    }
    __syncthreads();
    for (int b = threadIdx.x; b < N_TILES; b += BIN_THREADS)
        if (hist[b]) hist[b] = atomicAdd(&cursor[b], hist[b]);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < BIN_ITEMS; k++)
        if (it[k].tile >= 0)
            rec[hist[it[k].tile] + rank[k]] = Rec{(uint32_t)(base + k * BIN_THREADS + threadIdx.x), it[k].lp, it[k].depth};
}

// pass 3: one workgroup per tile.  Per-pixel lists in LDS; every item walks its pixel's list (about cacheFactor items):
//   loser  <=> some item of the pixel is nearer, or as near and earlier in the cache   -> aged once (Driver.js:446-468)
//   keeps its pointer <=> no EARLIER item of the pixel is as near or nearer             (Sample.attach, a prefix minimum)
__global__ void __launch_bounds__(256) tile_kernel(const uint32_t* tile_start, const uint32_t* tile_count, const Rec* rec, double* age,
                                                   int32_t* item_pixel, int32_t* pixel_element, double* pixel_depth)
{
    __shared__ double dep[CAP];
    __shared__ uint32_t idx[CAP];
    __shared__ uint16_t lpx[CAP], next[CAP];
    __shared__ uint32_t head[TILE_PX];
    const int tile = blockIdx.x;
    const uint32_t m = tile_count[tile], s0 = tile_start[tile];
    if (m > (uint32_t)CAP) return;                              // (the prototype has no path for a tile that does not fit)
    for (int p = threadIdx.x; p < TILE_PX; p += 256) head[p] = 0xffffu;
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < m; k += 256) {
        const Rec r = rec[s0 + k];
        dep[k] = r.depth; idx[k] = r.idx; lpx[k] = (uint16_t)r.lp;
        next[k] = (uint16_t)atomicExch(&head[r.lp], k);
    }
    __syncthreads();
    const int tx0 = (tile % TILES_X) * TILE_W + 1, ty0 = (tile / TILES_X) * TILE_H + 1;
    // (decisions first, for all of this lane's items; the updates at random cache indices afterwards, back to back)
    constexpr int MAXK = (CAP + 255) / 256;
    uint32_t flags = 0u;
#pragma unroll
    for (int q = 0; q < MAXK; q++) {
        const uint32_t k = threadIdx.x + 256u * q;
        if (k >= m) break;
        const double d = dep[k];
        const uint32_t i = idx[k];
        bool loser = false, keeps = true;
        for (uint32_t j = head[lpx[k]]; j != 0xffffu; j = next[j]) {
            if (j == k) continue;
            const double dj = dep[j];
            const bool earlier = idx[j] < i;
            loser |= dj < d || (dj == d && earlier);
            keeps &= !(earlier && dj <= d);
        }
        if (d >= THRESHOLD) { loser = true; keeps = false; }   // (never nearer than the cleared pixel: Driver.js:446)
        flags |= ((loser ? 1u : 0u) | (keeps ? 2u : 0u)) << (2 * q);
    }
#pragma unroll
    for (int q = 0; q < MAXK; q++) {
        const uint32_t k = threadIdx.x + 256u * q;
        if (k >= m) break;
        const uint32_t i = idx[k], f = flags >> (2 * q);
        if (f & 1u) age[i] += AGE_FACTOR;
        else {
            const int p = (ty0 + lpx[k] / TILE_W) * (W + 2) + tx0 + lpx[k] % TILE_W;
            pixel_element[p] = (int32_t)i; pixel_depth[p] = dep[k];
        }
        if (!(f & 2u)) item_pixel[i] = -1;
    }
    // (pixels without an item keep the cleared values: the clear pass stays as it is in the product)
}

__global__ void __launch_bounds__(256) clear_kernel(int32_t* pixel_element, double* pixel_depth)
{
    const int x = blockIdx.x * 256 + threadIdx.x + 1, y = blockIdx.y + 1;
    if (x > W) return;
    pixel_element[(size_t)y * (W + 2) + x] = -1;
    pixel_depth[(size_t)y * (W + 2) + x] = THRESHOLD;
}

int main()
{
    const int n = W * H * FACTOR;
    const size_t cells = (size_t)(W + 2) * (H + 2);
    const Cam cam{0.6, 0.6 * H / W};
    // hit points: a random pixel of the frame, a random depth, projected back (so that every hit lands inside, spread evenly)
    std::vector<double> hit((size_t)n * 3), age0(n);
    uint64_t s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (double)(s >> 11) / 9007199254740992.0; };
    for (int i = 0; i < n; i++) {
        const double z = -(8.0 + 52.0 * rnd()), u = 2.0 * rnd() - 1.0, v = 2.0 * rnd() - 1.0;
        hit[3 * (size_t)i] = rnd() < 0.2 ? NAN : u * cam.tx * -z;
        hit[3 * (size_t)i + 1] = v * cam.ty * -z;
        hit[3 * (size_t)i + 2] = z;
        age0[i] = (double)(int)(30.0 * rnd());
    }
    // the sequential walk of the reference
    std::vector<double> age_w(age0), pdep_w(cells, THRESHOLD);
    std::vector<int32_t> ip_w(n, -1), pel_w(cells, -1);
    size_t on_screen = 0;
    for (int i = 0; i < n; i++) {
        int px, py; double d;
        const int r = project(hit.data(), i, cam, px, py, d);
        if (r == -2) age_w[i] += AGE_FACTOR * 2;
        if (r != 0) continue;
        on_screen++;
        const size_t p = (size_t)py * (W + 2) + px;
        if (d < pdep_w[p]) {
            if (pel_w[p] >= 0) age_w[pel_w[p]] += AGE_FACTOR;
            ip_w[i] = (int32_t)p; pel_w[p] = i; pdep_w[p] = d;
        } else {
            age_w[i] += AGE_FACTOR;
        }
    }
    double *d_hit, *d_age, *d_pdep; int32_t *d_ip, *d_pel; uint32_t *d_cnt, *d_start, *d_cur, *d_over; Rec* d_rec;
    CHECK(hipMalloc(&d_hit, (size_t)n * 24)); CHECK(hipMalloc(&d_age, (size_t)n * 8)); CHECK(hipMalloc(&d_pdep, cells * 8));
    CHECK(hipMalloc(&d_ip, (size_t)n * 4)); CHECK(hipMalloc(&d_pel, cells * 4));
    CHECK(hipMalloc(&d_cnt, N_TILES * 4)); CHECK(hipMalloc(&d_start, N_TILES * 4)); CHECK(hipMalloc(&d_cur, N_TILES * 4));
    CHECK(hipMalloc(&d_over, 4)); CHECK(hipMalloc(&d_rec, (size_t)n * sizeof(Rec)));
    CHECK(hipMemcpy(d_hit, hit.data(), (size_t)n * 24, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_age, age0.data(), (size_t)n * 8, hipMemcpyHostToDevice));
    CHECK(hipMemset(d_over, 0, 4));
    const dim3 bin_grid((n + BIN_THREADS * BIN_ITEMS - 1) / (BIN_THREADS * BIN_ITEMS));
    hipEvent_t ev[7];
    for (auto& e : ev) CHECK(hipEventCreate(&e));
    float us[6] = {0, 0, 0, 0, 0, 0};
    const int reps = 11;
    std::vector<double> age_g(n), pdep_g(cells);
    std::vector<int32_t> ip_g(n), pel_g(cells);
    for (int r = 0; r < reps; r++) {
        CHECK(hipEventRecord(ev[0]));
        hipLaunchKernelGGL(clear_kernel, dim3((W + 255) / 256, H), dim3(256), 0, 0, d_pel, d_pdep);
        CHECK(hipMemsetAsync(d_cnt, 0, N_TILES * 4));
        CHECK(hipEventRecord(ev[1]));
        hipLaunchKernelGGL(count_kernel, bin_grid, dim3(BIN_THREADS), 0, 0, d_hit, n, cam, d_cnt);
        CHECK(hipEventRecord(ev[2]));
        hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(1024), 0, 0, d_cnt, d_start, d_cur, d_over);
        CHECK(hipEventRecord(ev[3]));
        hipLaunchKernelGGL(scatter_kernel, bin_grid, dim3(BIN_THREADS), 0, 0, d_hit, n, cam, d_age, d_ip, d_cur, d_rec);
        CHECK(hipEventRecord(ev[4]));
        hipLaunchKernelGGL(tile_kernel, dim3(N_TILES), dim3(256), 0, 0, d_start, d_cnt, d_rec, d_age, d_ip, d_pel, d_pdep);
        CHECK(hipEventRecord(ev[5]));
        CHECK(hipGetLastError());
        CHECK(hipEventSynchronize(ev[5]));
        if (r == 0) {                                   // the first pass is the checked one (the ages go on growing after it)
            uint32_t over; CHECK(hipMemcpy(&over, d_over, 4, hipMemcpyDeviceToHost));
            CHECK(hipMemcpy(age_g.data(), d_age, (size_t)n * 8, hipMemcpyDeviceToHost));
            CHECK(hipMemcpy(ip_g.data(), d_ip, (size_t)n * 4, hipMemcpyDeviceToHost));
            CHECK(hipMemcpy(pel_g.data(), d_pel, cells * 4, hipMemcpyDeviceToHost));
            CHECK(hipMemcpy(pdep_g.data(), d_pdep, cells * 8, hipMemcpyDeviceToHost));
            bool ok = over == 0 && age_g == age_w && ip_g == ip_w;
            for (int y = 1; y <= H && ok; y++)
                for (int x = 1; x <= W; x++) {
                    const size_t p = (size_t)y * (W + 2) + x;
                    if (pel_g[p] != pel_w[p] || pdep_g[p] != pdep_w[p]) { ok = false; break; }
                }
            printf("%d items, %zu land in the frame, %d tiles of %d x %d, %u tiles over the LDS capacity of %d: %s\n", n, on_screen, N_TILES,
                   TILE_W, TILE_H, over, CAP, ok ? "ages, pointers and grid equal the sequential walk" : "MISMATCH against the sequential walk");
            if (!ok) return 1;
            continue;
        }
        for (int k = 0; k < 5; k++) { float ms; CHECK(hipEventElapsedTime(&ms, ev[k], ev[k + 1])); us[k] += ms * 1e3f / (reps - 1); }
        float ms; CHECK(hipEventElapsedTime(&ms, ev[0], ev[5])); us[5] += ms * 1e3f / (reps - 1);
    }
    const double alg = (double)n * (24 + 8 + 8 + 4) + (double)W * H * 12;
    printf("clear + zero the counts   %7.1f us\ncount per tile            %7.1f us\nscan                      %7.1f us\n"
           "scatter into the tiles    %7.1f us\ntile pass (order, replay) %7.1f us\nwhole pass                %7.1f us = %.0f GB/s "
           "algorithmic = %.3f of 8 TB/s\n", us[0], us[1], us[2], us[3], us[4], us[5], alg / us[5] / 1e3, alg / us[5] / 1e3 / 8000.0);
    return 0;
}
