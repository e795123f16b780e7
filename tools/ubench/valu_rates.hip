// Microbenchmark: cycles per wave64 instruction of the VALU operations the ray kernels are made of (gfx950).
// Each mode runs 8 independent dependency chains of one operation; 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>

enum { FMA, MUL, ADD, MINF, MAX3, MED3, RSQ, RCP, SQRT, CMP_SGPR, CNDMASK, CMP_CND, FLOOR, CVT, FMA_SGPR, AND, OR3, ADDU, LSHL, CMP_ONLY, PKFMA, PKMUL, PKADD, MAXF, SUB, MODES };
static const char* names[] = {"v_fma_f32", "v_mul_f32", "v_add_f32", "v_min_f32", "v_max3_f32", "v_med3_f32", "v_rsq_f32", "v_rcp_f32",
                              "v_sqrt_f32", "v_cmp (ballot) + s_or", "v_cndmask (fixed mask)", "v_cmp + v_cndmask", "v_floor_f32",
                              "v_cvt_u32_f32", "v_fma_f32 2 SGPR-free/1 sgpr", "v_and_b32", "v_or3_b32", "v_add_u32", "v_lshlrev_b32",
                              "v_cmp only (s_or fold)", "v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_max_f32", "v_sub_f32"};

template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, float s0, float s1, int iters)
{
    float a[8];
    unsigned b[8];
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p[8];
#pragma unroll
    for (int u = 0; u < 8; u++) { a[u] = threadIdx.x * 0.001f + u + 1.0f; b[u] = threadIdx.x * 2654435761u + u; p[u] = f2{a[u], a[u] + 0.5f}; }
    const f2 m2{s0, s0}, c2{s1, s1};
    unsigned long long acc = 0;
    const bool fixed = (threadIdx.x & 3) == 1;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                float& x = a[u];
                if (MODE == FMA) x = fmaf(x, s0, s1);
                else if (MODE == MUL) x = x * s0;
                else if (MODE == ADD) x = x + s1;
                else if (MODE == MINF) x = fminf(x, a[(u + 1) & 7]);
                else if (MODE == MAX3) x = fmaxf(fmaxf(x, a[(u + 1) & 7]), a[(u + 2) & 7]);
                else if (MODE == MED3) x = __builtin_amdgcn_fmed3f(x, a[(u + 1) & 7], a[(u + 2) & 7]);
                else if (MODE == RSQ) x = __builtin_amdgcn_rsqf(x);
                else if (MODE == RCP) x = __builtin_amdgcn_rcpf(x);
                else if (MODE == SQRT) x = __builtin_amdgcn_sqrtf(x);
                else if (MODE == CMP_SGPR) acc |= __builtin_amdgcn_ballot_w64(x > a[(u + r) & 7] + s1);   // add keeps operands changing
                else if (MODE == CNDMASK) x = fixed ? a[(u + 1) & 7] : x;
                else if (MODE == CMP_CND) x = x > a[(u + 3) & 7] ? a[(u + 1) & 7] : x;
                else if (MODE == FLOOR) x = __builtin_floorf(x) + 0.0f;
                else if (MODE == CVT) x = (float)(unsigned)(x);
                else if (MODE == FMA_SGPR) x = fmaf(x, x, s1);
                else if (MODE == AND) b[u] = b[u] & b[(u + 1) & 7];
                else if (MODE == OR3) b[u] = b[u] | b[(u + 1) & 7] | b[(u + 2) & 7];
                else if (MODE == ADDU) b[u] = b[u] + b[(u + 1) & 7];
                else if (MODE == LSHL) b[u] = (b[u] << 1) ^ 0u;
                else if (MODE == CMP_ONLY) acc |= __builtin_amdgcn_ballot_w64(a[u] > a[(u + r + 1) & 7]);
                else if (MODE == PKFMA) p[u] = __builtin_elementwise_fma(p[u], m2, c2);
                else if (MODE == PKMUL) p[u] = p[u] * m2;
                else if (MODE == PKADD) p[u] = p[u] + c2;
                else if (MODE == MAXF) x = fmaxf(x, a[(u + 1) & 7]);
                else if (MODE == SUB) x = x - s1;
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int u = 0; u < 8; u++) s += a[u] + (float)b[u] + p[u].x + p[u].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + (float)(acc & 1);
}

template <int MODE> void run(float* d)
{
    const int iters = 1000, wpc = 4, blocks = 256 * wpc;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0000001f, 1e-9f, 10);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0000001f, 1e-9f, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double ops = (double)iters * 64 * wpc;
    printf("%-32s %.3f ms -> %.2f ns per source-level op per SIMD\n", names[MODE], ms, ms * 1e6 / ops);
    if constexpr (MODE + 1 < MODES) run<MODE + 1>(d);
}

int main()
{
    float* d; (void)hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
    run<0>(d);
    return 0;
}
