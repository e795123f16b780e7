// Microbenchmark: how fast can the chip START waves?  The few-sphere kernel launches 1 036 800 waves of 64 lanes per
// step (64 frames of 1920 x 1080, 128 pixels per wave); a kernel that does nothing but exist on the same grid gives the
// floor that dispatch alone puts under a step.  Variants: empty body; one 4-byte store per lane; a body that holds
// VGPRs / LDS like the real kernel (occupancy 7) and spins for a given number of dependent FMAs.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int WORK>
__global__ void __launch_bounds__(256) k(uint32_t* out, int width, float s0, int spin)
{
    if (WORK == 0) return;
    const unsigned x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    float a = threadIdx.x;
    if (WORK >= 2)
        for (int i = 0; i < spin; i++) a = fmaf(a, s0, 1.0f);
    out[(blockIdx.z * gridDim.y * 4 + y) * width + x] = (uint32_t)a;
}

template <int WORK> float run(uint32_t* d, dim3 grid, int width, int spin)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL(k<WORK>, grid, dim3(256), 0, 0, d, width, 1.0001f, spin);
    hipEventRecord(e0);
    for (int i = 0; i < 10; i++) hipLaunchKernelGGL(k<WORK>, grid, dim3(256), 0, 0, d, width, 1.0001f, spin);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / 10;
}

int main()
{
    const int W = 1920, H = 1080, F = 64;
    uint32_t* d; hipMalloc(&d, (size_t)W * H * F * 4);
    // grid A: one lane per pixel (4 147 200 waves); grid B: the real kernel's wave count (one lane per 2 pixels)
    for (int halves = 0; halves < 3; halves++) {
        const int f = F >> halves;
        dim3 grid(W / 64, H / 4, f);
        const double waves = (double)grid.x * grid.y * grid.z * 4;
        printf("%.0f waves in workgroups of 4:\n", waves);
        float t = run<0>(d, grid, W, 0);   printf("  empty body          %.3f ms  = %.2f waves/ns\n", t, waves / t * 1e-6);
        t = run<1>(d, grid, W, 0);         printf("  one store per lane  %.3f ms  = %.2f waves/ns\n", t, waves / t * 1e-6);
        for (int spin : {100, 400, 1000}) { t = run<2>(d, grid, W, spin); printf("  %4d dependent FMAs + store  %.3f ms  = %.2f waves/ns\n", spin, t, waves / t * 1e-6); }
    }
    return 0;
}
