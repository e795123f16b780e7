// f64_cmp_rate.hip — issue cost of the float64 compare / min / max / class instructions on gfx950, by inline assembly
// (16 back-to-back instructions on independent registers per loop pass; 4 waves per SIMD).
//   hipcc -O3 --offload-arch=gfx950 -o tools/ubench/f64_cmp_rate tools/ubench/f64_cmp_rate.hip && tools/ubench/f64_cmp_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

#define REP16(S) S S S S S S S S S S S S S S S S
enum { CMP_F64_VCC, CMP_F64_SGPR, CMP_F32_VCC, CMP_I32_VCC, CMP_U64_VCC, MAX_F64, MIN_F64, CLASS_F64, FMA_F64, ADD_F64, CNDMASK, CMP_F64_CND, CMP_I32_CND, MODES };
static const char* names[] = {"v_cmp_gt_f64 vcc", "v_cmp_gt_f64 s[10:11]", "v_cmp_gt_f32 vcc", "v_cmp_gt_i32 vcc", "v_cmp_gt_u64 vcc", "v_max_f64", "v_min_f64",
                              "v_cmp_class_f64 vcc", "v_fma_f64", "v_add_f64", "v_cndmask_b32", "v_cmp_gt_f64 + 2 v_cndmask (dependent)", "v_cmp_gt_i32 + 2 v_cndmask (dependent)"};

template <int MODE>
__global__ void __launch_bounds__(256) k(double* out, int iters)
{
    double a = 1.0 + threadIdx.x * 0.001, b = 2.0 - threadIdx.x * 0.002, c = 0.5;
    double d = 3.0;
    for (int i = 0; i < iters; i++) {
        if (MODE == CMP_F64_VCC) asm volatile(REP16("v_cmp_gt_f64 vcc, %0, %1\n") :: "v"(a), "v"(b) : "vcc");
        else if (MODE == CMP_F64_SGPR) asm volatile(REP16("v_cmp_gt_f64 s[10:11], %0, %1\n") :: "v"(a), "v"(b) : "s10", "s11");
        else if (MODE == CMP_F32_VCC) asm volatile(REP16("v_cmp_gt_f32 vcc, %0, %1\n") :: "v"((float)a), "v"((float)b) : "vcc");
        else if (MODE == CMP_I32_VCC) asm volatile(REP16("v_cmp_gt_i32 vcc, %0, %1\n") :: "v"((int)a), "v"((int)b) : "vcc");
        else if (MODE == CMP_U64_VCC) asm volatile(REP16("v_cmp_gt_u64 vcc, %0, %1\n") :: "v"(a), "v"(b) : "vcc");
        else if (MODE == MAX_F64) asm volatile(REP16("v_max_f64 %0, %1, %2\n") : "=v"(d) : "v"(a), "v"(b));
        else if (MODE == MIN_F64) asm volatile(REP16("v_min_f64 %0, %1, %2\n") : "=v"(d) : "v"(a), "v"(b));
        else if (MODE == CLASS_F64) asm volatile(REP16("v_cmp_class_f64 vcc, %0, %1\n") :: "v"(a), "v"(0x260) : "vcc");
        else if (MODE == FMA_F64) asm volatile(REP16("v_fma_f64 %0, %1, %2, %3\n") : "=v"(d) : "v"(a), "v"(b), "v"(c));
        else if (MODE == ADD_F64) asm volatile(REP16("v_add_f64 %0, %1, %2\n") : "=v"(d) : "v"(a), "v"(b));
        else if (MODE == CNDMASK) { int lo; asm volatile(REP16("v_cndmask_b32 %0, %1, %2, vcc\n") : "=v"(lo) : "v"((int)a), "v"((int)b) : "vcc"); d += lo; }
        else if (MODE == CMP_F64_CND) {
            int lo, hi;
            asm volatile(REP16("v_cmp_gt_f64 vcc, %2, %3\n s_nop 1\n v_cndmask_b32 %0, %4, %5, vcc\n v_cndmask_b32 %1, %5, %4, vcc\n")
                         : "=v"(lo), "=v"(hi) : "v"(a), "v"(b), "v"((int)a), "v"((int)b) : "vcc");
            d += lo + hi;
        }
        else if (MODE == CMP_I32_CND) {
            int lo, hi;
            asm volatile(REP16("v_cmp_gt_i32 vcc, %2, %3\n s_nop 1\n v_cndmask_b32 %0, %2, %3, vcc\n v_cndmask_b32 %1, %3, %2, vcc\n")
                         : "=v"(lo), "=v"(hi) : "v"((int)a), "v"((int)b) : "vcc");
            d += lo + hi;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = d;
}
template <int MODE> void run(double* d)
{
    const int iters = 2000, wpc = 4, blocks = 256 * wpc;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 10);
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double ops = (double)iters * 16 * wpc;
    printf("%-44s %.3f ms -> %.2f ns per instruction (group) per SIMD\n", names[MODE], ms, ms * 1e6 / ops);
    if constexpr (MODE + 1 < MODES) run<MODE + 1>(d);
}
int main()
{
    double* d; CHECK(hipMalloc(&d, 256 * 4 * 256 * sizeof(double)));
    run<0>(d);
    return 0;
}
