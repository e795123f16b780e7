// Microbenchmark: what do the scalar-side idioms of the ray kernels cost per SIMD (gfx950)?  Every step is one
// v_fma_f32 plus the idiom; 8 independent chains per wave; 1 / 4 / 8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>

static const char* names[] = {"v_fma only", "+ v_cmp -> vcc", "+ v_cmp, s_cbranch_vccnz (not taken)", "+ v_cmp -> sgpr, s_or_b64",
                              "+ s_cmp, s_cbranch_scc1 (scalar only)", "+ v_cmp, s_and_saveexec, s_or exec", "+ 2 x s_and_b64",
                              "+ v_cmp, s_cmp_lg_u64 vcc, s_cbranch_scc1", "+ v_cndmask (vcc fixed)"};
template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, float s0, float s1, int iters, int zero)
{
    float a[8];
#pragma unroll
    for (int u = 0; u < 8; u++) a[u] = threadIdx.x * 0.001f + u + 1.0f;
    const float big = 1e30f + s1;
    unsigned long long acc = 0, m1 = 0x5555555555555555ull, m2 = ~0ull;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                a[u] = fmaf(a[u], s0, s1);
                if (MODE == 1) asm volatile("v_cmp_gt_f32 vcc, %1, %0" ::"v"(a[u]), "v"(big) : "vcc");
                if (MODE == 2) asm volatile("v_cmp_gt_f32 vcc, %1, %0\n s_cbranch_vccnz .L%=\n .L%=:" ::"v"(a[u]), "v"(big) : "vcc");
                if (MODE == 3) { unsigned long long t; asm volatile("v_cmp_gt_f32 %0, %2, %1" : "=s"(t) : "v"(a[u]), "v"(big)); acc |= t; }
                if (MODE == 4) asm volatile("s_cmp_eq_u32 %0, 1\n s_cbranch_scc1 .L%=\n .L%=:" ::"s"(zero) : "scc");
                if (MODE == 5) { unsigned long long t; asm volatile("v_cmp_gt_f32 vcc, %2, %1\n s_and_saveexec_b64 %0, vcc\n s_or_b64 exec, exec, %0" : "=s"(t) : "v"(a[u]), "v"(big) : "vcc", "scc"); }
                if (MODE == 6) asm volatile("s_and_b64 %0, %0, %1\n s_and_b64 %1, %1, %0" : "+s"(m1), "+s"(m2) : : "scc");
                if (MODE == 7) asm volatile("v_cmp_gt_f32 vcc, %1, %0\n s_cmp_lg_u64 vcc, 0\n s_cbranch_scc1 .L%=\n .L%=:" ::"v"(a[u]), "v"(big) : "vcc", "scc");
                if (MODE == 8) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[u]) : "v"(a[(u + 1) & 7]));
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int u = 0; u < 8; u++) s += a[u];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + (float)((acc ^ m1 ^ m2) & 1);
}

template <int MODE> void run(float* d)
{
    for (int wpc : {1, 4, 8}) {
        const int iters = 2000, blocks = 256 * wpc;
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0000001f, 1e-9f, 10, 0);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0000001f, 1e-9f, iters, 0);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double per_wave = (double)iters * 64;
        printf("%-44s waves/SIMD %d: %6.2f ns per step per wave, %5.2f ns per step per SIMD\n", names[MODE], wpc,
               ms * 1e6 / per_wave, ms * 1e6 / (per_wave * wpc));
    }
    if constexpr (MODE < 8) run<MODE + 1>(d);
}

int main()
{
    float* d; (void)hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
    run<0>(d);
    return 0;
}
