// Microbenchmark: how fast does ONE wave issue?  Chains of dependent v_fma_f32 (1, 2, 4, 8 independent chains per
// wave) and a compare -> lane mask -> scalar branch loop, at 1, 2, 4, 8 waves per SIMD (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>

template <int CHAINS, int MODE>
__global__ void __launch_bounds__(256) k(float* out, float s0, float s1, int iters)
{
    float a[8];
#pragma unroll
    for (int u = 0; u < 8; u++) a[u] = threadIdx.x * 0.001f + u + 1.0f;
    unsigned long long acc = 0;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < 64 / CHAINS; r++) {
#pragma unroll
            for (int u = 0; u < CHAINS; u++) {
                if (MODE == 0) a[u] = fmaf(a[u], s0, s1);
                else if (MODE == 1) a[u] = fminf(a[u] * s0, a[(u + 1) & 7]);     // 2-cycle + 4-cycle op
                else {                                                           // fma, compare, mask test, branch
                    a[u] = fmaf(a[u], s0, s1);
                    const unsigned long long m = __builtin_amdgcn_ballot_w64(a[u] > 1e30f);
                    if (m != 0ull) { a[u] = 0.5f; acc += m; }
                }
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int u = 0; u < 8; u++) s += a[u];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + (float)(acc & 1);
}

template <int CHAINS, int MODE> void run(float* d, int wpc, const char* name)
{
    const int iters = 2000, blocks = 256 * wpc;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<CHAINS, MODE>), dim3(blocks), dim3(256), 0, 0, d, 1.0000001f, 1e-9f, 10);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<CHAINS, MODE>), dim3(blocks), dim3(256), 0, 0, d, 1.0000001f, 1e-9f, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double per_wave = (double)iters * 64;                 // source-level steps per wave
    printf("%-28s chains %d  waves/SIMD %d: %.2f ns per step per wave, %.2f ns per step per SIMD\n", name, CHAINS, wpc,
           ms * 1e6 / per_wave, ms * 1e6 / (per_wave * wpc));
}

int main()
{
    float* d; (void)hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
    for (int wpc : {1, 2, 4, 8}) {
        run<1, 0>(d, wpc, "v_fma dependent");
        run<2, 0>(d, wpc, "v_fma");
        run<4, 0>(d, wpc, "v_fma");
        run<8, 0>(d, wpc, "v_fma");
        run<1, 1>(d, wpc, "v_mul + v_min dependent");
        run<8, 1>(d, wpc, "v_mul + v_min");
        run<1, 2>(d, wpc, "fma+cmp+branch dependent");
        run<8, 2>(d, wpc, "fma+cmp+branch");
    }
    return 0;
}
