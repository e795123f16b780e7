// ptr_f64_roots.h - float64 square root and reciprocal of the wave-level bounce loop (ptr_kernels_ext.hip), in a header of
// their own so that tools/ubench/f64_roots.hip checks exactly this code against the host's IEEE results.
#pragma once
#include <hip/hip_runtime.h>

namespace ptr {

// Correctly rounded float64 square root and reciprocal for operands in 2^-500 .. 2^500: the compiler's own sequences
// (v_rsq_f64 / v_rcp_f64 + Newton steps in FMAs + a last residual correction) without the parts that only rescale operands
// near the ends of the exponent range and patch zero / infinity (5-7 of 18 and 4 of 11 instructions).  Every call site asks
// mid_range() for all its lanes first (two integer instructions on the high word) and takes the builtin otherwise, so the
// results are the IEEE ones for every operand - which is what JavaScript's Math.sqrt and `/` return.
__device__ __forceinline__ bool mid_range(double x)
{
#ifdef PTR_EXT64_NO_MID
    return false;
#endif
    return (unsigned)__double2hiint(x) - 0x20B00000u < 0x3E800000u;      // exponent field in [1023 - 500, 1023 + 500): positive, normal
}
__device__ __forceinline__ double sqrt_mid(double x)
{
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = y * 0.5;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    double d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, x);
    return __builtin_fma(d, h, g);
}
__device__ __forceinline__ double recip_mid(double s)
{
    double r = __builtin_amdgcn_rcp(s);
    double e = __builtin_fma(-s, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-s, r, 1.0);
    r = __builtin_fma(r, e, r);
    const double rem = __builtin_fma(-s, r, 1.0);        // (the quotient estimate of 1 / s is r itself)
    return __builtin_fma(rem, r, r);
}

}  // namespace ptr
