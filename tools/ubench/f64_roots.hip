// f64_roots.hip — (1) are the float64 square root / reciprocal sequences of the wave-level bounce loop (ptr_f64_roots.h) and the
// compiler's own `1.0 / x`, `sqrt(x)` the IEEE results on gfx950?  Checked against the host's (x86 SSE2: correctly rounded) on
// the hard cases (significands of all ones, neighbours of powers of two and of 1, squares +- an ulp) and on random operands.
// (2) cycles per wave64 instruction of the float64 operations the loop is made of.
//   hipcc -O3 --offload-arch=gfx950 -o tools/ubench/f64_roots tools/ubench/f64_roots.hip && tools/ubench/f64_roots [millions of random operands]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../path-tracer_amd/csrc/ptr_f64_roots.h"

namespace ptr {
// A candidate that is NOT in the product: RN(1 / RN(sqrt(x))), the two roundings of Vector3.normalize's `1 / Math.sqrt(len2)`,
// with the reciprocal of the rounded root started from the root's own h ~ 1 / (2 sqrt(x)) instead of v_rcp_f64 (one Newton step
// and the same last residual correction as recip_mid: 14 instructions instead of 16, none of them a v_rcp_f64).  It is wrong
// exactly where the last correction needs a correctly rounded estimate: roots whose significand is all ones (x = 1 - 1..3 ulps
// and their like at every odd exponent), where 1 / s lies 2^-106 above a tie and a Newton step from below rounds it down.
// v_rcp_f64 returns the upper neighbour there, which is why the compiler's sequence (and recip_mid) get it right.  len2 of a
// vector that is a unit vector up to rounding IS such an operand all the time, so the candidate is dropped.
__device__ __forceinline__ double inv_sqrt_mid(double x)
{
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = y * 0.5;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    double d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, x);
    const double s = __builtin_fma(d, h, g);
    double q = h + h;
    double e = __builtin_fma(-s, q, 1.0);
    q = __builtin_fma(q, e, q);
    e = __builtin_fma(-s, q, 1.0);
    return __builtin_fma(e, q, q);
}
}  // namespace ptr

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// out[0..5][i]: builtin 1/x, recip_mid, builtin sqrt, sqrt_mid, builtin 1/sqrt(x) in two roundings, inv_sqrt_mid
__global__ void eval_kernel(const double* x, double* out, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double v = x[i];
    out[i] = 1.0 / v;
    out[n + i] = ptr::recip_mid(v);
    out[2 * n + i] = __builtin_sqrt(v);
    out[3 * n + i] = ptr::sqrt_mid(v);
    out[4 * n + i] = 1.0 / __builtin_sqrt(v);
    out[5 * n + i] = ptr::inv_sqrt_mid(v);
}

static double from_bits(uint64_t b) { double d; memcpy(&d, &b, 8); return d; }
static uint64_t bits_of(double d) { uint64_t b; memcpy(&b, &d, 8); return b; }

static size_t check(const std::vector<double>& xs, const char* what)
{
    const size_t n = xs.size();
    double *dx, *dout;
    CHECK(hipMalloc(&dx, n * 8)); CHECK(hipMalloc(&dout, 6 * n * 8));
    CHECK(hipMemcpy(dx, xs.data(), n * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(eval_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, dx, dout, n);
    CHECK(hipGetLastError());
    std::vector<double> got(6 * n);
    CHECK(hipMemcpy(got.data(), dout, 6 * n * 8, hipMemcpyDeviceToHost));
    CHECK(hipFree(dx)); CHECK(hipFree(dout));
    static const char* names[6] = {"1.0 / x (compiler)", "recip_mid", "sqrt (compiler)", "sqrt_mid", "1.0 / sqrt(x) (compiler)", "inv_sqrt_mid"};
    size_t bad[6] = {0, 0, 0, 0, 0, 0};
    for (size_t i = 0; i < n; i++) {
        volatile double v = xs[i];
        volatile double s = std::sqrt(v);
        const double want[6] = {1.0 / v, 1.0 / v, s, s, 1.0 / s, 1.0 / s};
        for (int k = 0; k < 6; k++)
            if (bits_of(got[k * n + i]) != bits_of(want[k])) {
                if (bad[k]++ < 3) printf("    %s(%a) = %a, host %a\n", names[k], xs[i], got[k * n + i], want[k]);
            }
    }
    size_t total = 0;
    printf("%-44s %10zu operands:", what, n);
    for (int k = 0; k < 6; k++) { printf(" %zu", bad[k]); if (k < 5) total += bad[k]; }      // (the dropped candidate is reported, not counted)
    printf("  (differ from the host: 1/x, recip_mid, sqrt, sqrt_mid, 1/sqrt, inv_sqrt_mid)\n");
    return total;
}

// ---- rates ----
enum { FMA64, MUL64, ADD64, RCP64, RSQ64, SQRT64, DIV64, SQRTMID, RECIPMID, INVSQRTMID, RECIP_OF_SQRT_MID, CMP64, CND64, CVT64_32, MODES };
static const char* rate_names[] = {"v_fma_f64", "v_mul_f64", "v_add_f64", "v_rcp_f64", "v_rsq_f64", "sqrt (compiler, 18 instr.)", "1.0 / x (compiler, 11 instr.)",
                                   "sqrt_mid (9)", "recip_mid (7)", "inv_sqrt_mid (14)", "recip_mid(sqrt_mid) (16)", "v_cmp_f64 + v_cndmask x 2",
                                   "v_cndmask x 2 (fixed mask)", "v_cvt_f32_f64"};
template <int MODE>
__global__ void __launch_bounds__(256) rate_kernel(double* out, double s0, double s1, int iters)
{
    double a[4];
#pragma unroll
    for (int u = 0; u < 4; u++) a[u] = 1.0 + threadIdx.x * 0.001 + u;
    const bool fixed = (threadIdx.x & 3) == 1;
    float facc = 0;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                double& x = a[u];
                if (MODE == FMA64) x = __builtin_fma(x, s0, s1);
                else if (MODE == MUL64) x = x * s0;
                else if (MODE == ADD64) x = x + s1;
                else if (MODE == RCP64) x = __builtin_amdgcn_rcp(x);
                else if (MODE == RSQ64) x = __builtin_amdgcn_rsq(x);
                else if (MODE == SQRT64) x = __builtin_sqrt(x) + s0;
                else if (MODE == DIV64) x = 1.0 / x + s0;
                else if (MODE == SQRTMID) x = ptr::sqrt_mid(x) + s0;
                else if (MODE == RECIPMID) x = ptr::recip_mid(x) + s0;
                else if (MODE == INVSQRTMID) x = ptr::inv_sqrt_mid(x) + s0;
                else if (MODE == RECIP_OF_SQRT_MID) x = ptr::recip_mid(ptr::sqrt_mid(x)) + s0;
                else if (MODE == CMP64) x = x > a[(u + 1) & 3] ? a[(u + 2) & 3] : x;
                else if (MODE == CND64) x = fixed ? a[(u + 1) & 3] : x;
                else if (MODE == CVT64_32) { facc += (float)x; }
            }
        }
    }
    double s = facc;
#pragma unroll
    for (int u = 0; u < 4; u++) s += a[u];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> void rates(double* d)
{
    const int iters = 500, wpc = 4, blocks = 256 * wpc;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(rate_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0000001, 1e-9, 10);
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(rate_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0000001, 1e-9, iters);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double ops = (double)iters * 32 * wpc;      // source-level operations per SIMD (4 waves of each of the 4 workgroups per CU / 4 SIMDs)
    printf("%-34s %.3f ms -> %.2f ns per source-level op per SIMD\n", rate_names[MODE], ms, ms * 1e6 / ops);
    if constexpr (MODE + 1 < MODES) rates<MODE + 1>(d);
}

int main(int argc, char** argv)
{
    const size_t millions = argc > 1 ? (size_t)atol(argv[1]) : 64;
    size_t bad = 0;
    {   // hard cases
        std::vector<double> xs;
        for (int e = -400; e <= 400; e += 1) {
            const uint64_t eb = (uint64_t)(1023 + e) << 52;
            for (uint64_t k = 0; k < 64; k++) {
                xs.push_back(from_bits(eb | (0x000FFFFFFFFFFFFFull - k)));     // all ones and below
                xs.push_back(from_bits(eb | k));                               // the power of two and above
                xs.push_back(from_bits(eb | (0x0008000000000000ull + k)));
                xs.push_back(from_bits(eb | (0x0008000000000000ull - 1 - k)));
            }
        }
        bad += check(xs, "hard cases (all ones, powers of two +-)");
        xs.clear();
        // squares of numbers with few bits, +- some ulps: roots that land on or beside a rounding boundary
        uint64_t s = 0x9E3779B97F4A7C15ull;
        for (int i = 0; i < 2000000; i++) {
            s ^= s << 13; s ^= s >> 7; s ^= s << 17;
            const double a = 1.0 + (double)(s >> 38) * 0x1p-26;          // 26 significant bits: a * a is exact
            const double sq = a * a;
            const int64_t off = (int64_t)((s >> 8) & 7) - 3;
            xs.push_back(from_bits(bits_of(sq) + (uint64_t)off));
            // (a + half an ulp)^2 = the numbers whose roots lie next to a tie
            const double b = 1.0 + (double)(s >> 12) * 0x1p-52;
            const double mid2 = b * b + b * 0x1p-52;                       // ~ (b + ulp / 2)^2
            xs.push_back(from_bits(bits_of(mid2) + (uint64_t)off));
        }
        bad += check(xs, "squares and near-tie squares");
        xs.clear();
        // len2 of vectors that are unit vectors up to rounding: 1 +- a few ulps (what the bounce loop normalises all the time)
        for (int64_t k = -4096; k <= 4096; k++) xs.push_back(from_bits(bits_of(1.0) + (uint64_t)k));
        bad += check(xs, "1 +- 4096 ulps");
    }
    {   // random operands, mid-range exponents
        uint64_t s = 88172645463325252ull;
        const size_t chunk = 16u << 20;
        std::vector<double> xs(chunk);
        for (size_t done = 0; done < millions << 20; done += chunk) {
            for (size_t i = 0; i < chunk; i++) {
                s ^= s << 13; s ^= s >> 7; s ^= s << 17;
                const uint64_t mant = s & 0x000FFFFFFFFFFFFFull;
                const int e = (int)((s >> 52) % 200) - 100;
                xs[i] = from_bits(((uint64_t)(1023 + e) << 52) | mant);
            }
            char what[64]; snprintf(what, sizeof what, "random, exponents -100..99 (%zu M done)", done >> 20);
            bad += check(xs, what);
        }
    }
    printf("total differing results of the product's sequences and the compiler's: %zu\n", bad);
    double* d; CHECK(hipMalloc(&d, 256 * 4 * 256 * sizeof(double)));
    rates<0>(d);
    return bad == 0 ? 0 : 1;
}
