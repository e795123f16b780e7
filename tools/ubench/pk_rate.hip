// Microbenchmark: issue rate of v_pk_fma_f32 against v_fma_f32 on gfx950 (one wave64 instruction = how many cycles?)
// build: hipcc -O3 --offload-arch=gfx950 -o pk_rate pk_rate.hip ; run: ./pk_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, float s0, float s1, int iters)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    f2 p0{a0, a1}, p1{a2, a3}, p2{a4, a5}, p3{a6, a7}, p4{a1, a0}, p5{a3, a2}, p6{a5, a4}, p7{a7, a6};
    f2 m{s0, s1}, c{s1, s0};
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {            // 8 independent scalar FMAs
#pragma unroll
            for (int u = 0; u < 8; u++) {
                a0 = fmaf(a0, s0, s1); a1 = fmaf(a1, s0, s1); a2 = fmaf(a2, s0, s1); a3 = fmaf(a3, s0, s1);
                a4 = fmaf(a4, s0, s1); a5 = fmaf(a5, s0, s1); a6 = fmaf(a6, s0, s1); a7 = fmaf(a7, s0, s1);
            }
        } else if (MODE == 1) {     // 8 independent packed FMAs, all-VGPR operands
#pragma unroll
            for (int u = 0; u < 8; u++) {
                p0 = __builtin_elementwise_fma(p0, m, c); p1 = __builtin_elementwise_fma(p1, m, c);
                p2 = __builtin_elementwise_fma(p2, m, c); p3 = __builtin_elementwise_fma(p3, m, c);
                p4 = __builtin_elementwise_fma(p4, m, c); p5 = __builtin_elementwise_fma(p5, m, c);
                p6 = __builtin_elementwise_fma(p6, m, c); p7 = __builtin_elementwise_fma(p7, m, c);
            }
        } else {                    // packed FMAs with a broadcast (op_sel) VGPR operand, as the kernels use
#pragma unroll
            for (int u = 0; u < 8; u++) {
                p0 = __builtin_elementwise_fma(f2{p4.x, p4.x}, p0, p1); p1 = __builtin_elementwise_fma(f2{p4.y, p4.y}, p1, p2);
                p2 = __builtin_elementwise_fma(f2{p5.x, p5.x}, p2, p3); p3 = __builtin_elementwise_fma(f2{p5.y, p5.y}, p3, p0);
                p6 = __builtin_elementwise_fma(f2{p4.x, p4.x}, p6, p7); p7 = __builtin_elementwise_fma(f2{p4.y, p4.y}, p7, p6);
                p0 = __builtin_elementwise_fma(f2{p5.x, p5.x}, p0, p2); p2 = __builtin_elementwise_fma(f2{p5.y, p5.y}, p2, p1);
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y +
                                                 p3.x + p3.y + p4.x + p4.y + p5.x + p5.y + p6.x + p6.y + p7.x + p7.y;
}

template <int MODE> void run(const char* name, float* d, int wpc)
{
    const int iters = 2000, blocks = 256 * wpc;          // wpc workgroups (4 waves each) per CU
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0000001f, 1e-9f, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0000001f, 1e-9f, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = (double)iters * 64 * wpc;              // wave-instructions each SIMD issues (wpc waves per SIMD)
    printf("%-28s waves/SIMD %d: %.3f ms -> %.2f ns per wave-instruction per SIMD (%.2f cycles at 2.4 GHz)\n", name, wpc, ms,
           ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
}

int main()
{
    float* d; hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
    for (int wpc : {1, 2, 4}) {
        run<0>("v_fma_f32", d, wpc);
        run<1>("v_pk_fma_f32", d, wpc);
        run<2>("v_pk_fma_f32 op_sel bcast", d, wpc);
    }
    return 0;
}
