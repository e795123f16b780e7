// Microbenchmark for the render cache's reprojection pass: how fast are 8.3 M returning atomics at random addresses of a
// 2 M-entry int32 array (a per-pixel list head), against plain random 16-byte gathers and 8-byte read-modify-writes of
// the same count?  Decides whether per-pixel linked lists (atomicExch on the head) can replace the radix sort.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

// mode 0: atomicExch with return + 16-byte node store (the list build); 1: atomicMin on uint64; 2: random 16-byte gather;
// 3: random 8-byte read-modify-write (non-atomic); 4: coherent addresses (neighbouring items -> neighbouring pixels)
template <int MODE>
__global__ void __launch_bounds__(256) k(int n, uint32_t cells, int32_t* head, unsigned long long* z, int4* node, double* age, int4* sink)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t p = mix((uint32_t)i) % cells;
    if (MODE == 4) p = ((uint32_t)i / 4u + (mix((uint32_t)i) & 63u)) % cells;
    if (MODE == 0 || MODE == 4) {
        const int old = atomicExch(&head[p], i);
        node[i] = int4{old, (int)p, i, 0};
    } else if (MODE == 1) {
        atomicMin(&z[p], ((unsigned long long)mix(i * 7u) << 32) | (unsigned)i);
    } else if (MODE == 2) {
        const int4 v = node[mix((uint32_t)i * 3u) % (uint32_t)n];
        if (v.x == 0x7fffffff) sink[0] = v;
    } else {
        age[mix((uint32_t)i * 5u) % (uint32_t)n] += 1.0;
    }
}

template <int MODE> void run(const char* what, int n, uint32_t cells, int32_t* head, unsigned long long* z, int4* node, double* age, int4* sink)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const dim3 grid((n + 255) / 256);
    for (int r = 0; r < 3; r++) hipLaunchKernelGGL(k<MODE>, grid, dim3(256), 0, 0, n, cells, head, z, node, age, sink);
    hipEventRecord(e0);
    for (int r = 0; r < 10; r++) hipLaunchKernelGGL(k<MODE>, grid, dim3(256), 0, 0, n, cells, head, z, node, age, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    printf("%-58s %8.1f us  %7.1f G/s\n", what, ms * 1e3, n / ms / 1e6);
}

int main()
{
    const int n = 1920 * 1080 * 4; const uint32_t cells = 1922 * 1082;
    int32_t* head; unsigned long long* z; int4* node; double* age; int4* sink;
    hipMalloc(&head, cells * 4); hipMalloc(&z, cells * 8ull); hipMalloc(&node, (size_t)n * 16); hipMalloc(&age, (size_t)n * 8); hipMalloc(&sink, 64);
    hipMemset(head, 0xff, cells * 4); hipMemset(z, 0xff, cells * 8ull); hipMemset(node, 0, (size_t)n * 16); hipMemset(age, 0, (size_t)n * 8);
    run<0>("atomicExch (returning) on random heads + 16 B node store", n, cells, head, z, node, age, sink);
    run<4>("the same, neighbouring items on neighbouring pixels", n, cells, head, z, node, age, sink);
    run<1>("atomicMin u64 (no return) on a random z-buffer", n, cells, head, z, node, age, sink);
    run<2>("random 16 B gathers", n, cells, head, z, node, age, sink);
    run<3>("random 8 B read-modify-writes", n, cells, head, z, node, age, sink);
    return 0;
}
