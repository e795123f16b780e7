// Calibration of "waves per SIMD from counters" (VERDICT r2 item 4): kernels that hold a KNOWN number of waves on every SIMD
// for about a millisecond.  k workgroups of 256 threads (4 waves: one per SIMD) per CU; the kernel's register allocation is
// padded so that no more than k waves fit a SIMD (amdgpu_waves_per_eu(k, k)): k = 1, 2, 3, 4, 5, 6, 8 waves per SIMD.  (A first
// version limited residency by LDS: from k = 3 on fewer workgroups than computed were resident - the run times doubled - and
// the counters showed exactly that.)  Run under rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES
// SQ_BUSY_CU_CYCLES SQ_LEVEL_WAVES SQ_ACCUM_PREV_HIRES GRBM_GUI_ACTIVE and compare the dispatches:
//   waves per SIMD = SQ_WAVE_CYCLES x u / (SQ_BUSY_CYCLES x s)   with u, s the counters' units - which is what this finds.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int K>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(K, K))) hold(float* out, int iters)
{
    extern __shared__ float lds[];
    float a = threadIdx.x * 1e-3f, b = 1.0000001f;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++) a = fmaf(a, b, 1e-9f);
    }
    if (a == 123.456f) lds[threadIdx.x] = a;      // (keeps the LDS allocation and the loop alive)
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

template <int K> void run(float* d, int iters)
{
    const size_t lds = 0;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(hold<K>, dim3(256 * K), dim3(256), lds, 0, d, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("hold<%d>: %d workgroups, %zu B LDS each, %.3f ms\n", K, 256 * K, lds, ms);
}

int main()
{
    float* d; (void)hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
    const int iters = 20000;
    run<1>(d, 100);                 // warm-up
    run<1>(d, iters); run<2>(d, iters); run<3>(d, iters); run<4>(d, iters); run<5>(d, iters); run<6>(d, iters); run<8>(d, iters);
    (void)hipDeviceSynchronize();
    return 0;
}
