// ptr_fast_common.h — pieces shared by the float32 kernels (ptr_kernels_fast.hip: few spheres, records in
// SGPRs; ptr_kernels_many.hip: many spheres, records staged through LDS).
#pragma once
#include <hip/hip_runtime.h>
#include "ptr_device.h"

namespace ptr {

// ---- error-bound constants (EPS = float32 unit roundoff 2^-24) ----------------------------
constexpr float EPS = 5.9604644775390625e-08f;
constexpr float FAR_UNSURE = 9.0e7f;     // anything this close to the 1e8 far clip is re-traced

struct F3 { float x, y, z; };

// Scene data is read-only for the whole launch: typing it as the constant address space makes
// every wave-uniform read a scalar load whatever stores the kernel has issued before.  The
// record loads are volatile so that the compiler keeps them where they are written (it otherwise
// sinks them next to their first use, which serialises load latency and compute).
#define PTR_CONST __attribute__((address_space(4)))
typedef const SphereM32 PTR_CONST* ConstM32;
typedef const FrameHeader PTR_CONST* ConstHeader;
typedef const int32_t PTR_CONST* ConstI32;
typedef float f4 __attribute__((ext_vector_type(4)));
typedef const volatile f4 PTR_CONST* ConstF4;      // a SphereP32 record = 4 x f4

// the 12 leading floats of a SphereP32 record (see ptr_device.h for the field names)
typedef float f8 __attribute__((ext_vector_type(8)));
// 8 floats by one s_load_dwordx8 (the record tables are 32-byte aligned)
__device__ __forceinline__ void load_f8(ConstF4 rec, f4& lo, f4& hi)
{
    const f8 v = *reinterpret_cast<const volatile f8 PTR_CONST*>(rec);
    lo = f4{v[0], v[1], v[2], v[3]};
    hi = f4{v[4], v[5], v[6], v[7]};
}
struct P32Q { f4 g0, g1, g2; };
__device__ __forceinline__ P32Q load_p32(ConstF4 rec) { P32Q q; load_f8(rec, q.g0, q.g1); q.g2 = rec[2]; return q; }
// the whole 64-byte record by one s_load_dwordx16 (the record table is 64-byte aligned)
typedef float f16v __attribute__((ext_vector_type(16)));
struct P32W { f4 g0, g1, g2, g3; };
__device__ __forceinline__ P32W load_p32w(ConstF4 rec)
{
    const f16v v = *reinterpret_cast<const volatile f16v PTR_CONST*>(rec);
    P32W q;
    q.g0 = f4{v[0], v[1], v[2], v[3]};   q.g1 = f4{v[4], v[5], v[6], v[7]};
    q.g2 = f4{v[8], v[9], v[10], v[11]}; q.g3 = f4{v[12], v[13], v[14], v[15]};
    return q;
}
struct P32G { f4 g0, g1; };
// the 8 leading floats (shadow test) + the (light, element) pre-filter record
struct P32H { f4 g0, g1, pa; };
__device__ __forceinline__ P32H load_p32h(ConstF4 rec, ConstF4 pair) { P32H q; load_f8(rec, q.g0, q.g1); q.pa = pair[0]; return q; }
// Scalar loads return out of order, so the only wait is "all of them": make the wave wait for the
// record it is about to use BEFORE it requests the next one.  arrive() is an empty asm that
// consumes every word of the record: that forces the wait here, and it keeps words the compute
// never reads (each loop uses a subset of the record) allocated until the load has landed —
// otherwise the register allocator hands them out while the load is in flight and the hardware
// hazard check stalls the wave right behind the load.
// arrive_after() also takes a value the preceding compute produces, which pins the wait AFTER that
// compute (an asm is otherwise free to float up to just behind the loads, which hides nothing).
// the 8 leading floats alone
__device__ __forceinline__ P32G load_p32g(ConstF4 rec) { P32G q; load_f8(rec, q.g0, q.g1); return q; }
// light k's record (LightRec)
struct LightQ { f4 g, e; int32_t index; float eps_lnorm; };
typedef float f2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ LightQ load_light(ConstF4 rec)
{
    LightQ q; load_f8(rec, q.g, q.e);
    const f2v t = *reinterpret_cast<const volatile f2v PTR_CONST*>(rec + 2);      // index (bits), eps |l|
    q.index = __builtin_bit_cast(int32_t, t[0]); q.eps_lnorm = t[1];
    return q;
}
#define PTR_F4_IN(v) "s"((v).x), "s"((v).y), "s"((v).z), "s"((v).w)
__device__ __forceinline__ void arrive(const P32Q& q)
{ asm volatile("" ::PTR_F4_IN(q.g0), PTR_F4_IN(q.g1), PTR_F4_IN(q.g2)); }
__device__ __forceinline__ void arrive_after(const P32Q& q, float dep)
{ asm volatile("" ::PTR_F4_IN(q.g0), PTR_F4_IN(q.g1), PTR_F4_IN(q.g2), "v"(dep)); }
__device__ __forceinline__ void arrive(const P32W& q)
{ asm volatile("" ::PTR_F4_IN(q.g0), PTR_F4_IN(q.g1), PTR_F4_IN(q.g2), PTR_F4_IN(q.g3)); }
__device__ __forceinline__ void arrive(const P32G& q)
{ asm volatile("" ::PTR_F4_IN(q.g0), PTR_F4_IN(q.g1)); }
__device__ __forceinline__ void arrive(const f4& q)
{ asm volatile("" ::PTR_F4_IN(q)); }
__device__ __forceinline__ void arrive(const LightQ& q)
{ asm volatile("" ::PTR_F4_IN(q.g), PTR_F4_IN(q.e), "s"(q.index), "s"(q.eps_lnorm)); }
__device__ __forceinline__ void arrive(const P32H& q)
{ asm volatile("" ::PTR_F4_IN(q.g0), PTR_F4_IN(q.g1), PTR_F4_IN(q.pa)); }
__device__ __forceinline__ void arrive_after(const P32H& q, unsigned long long dep)
{ asm volatile("" ::PTR_F4_IN(q.g0), PTR_F4_IN(q.g1), PTR_F4_IN(q.pa), "s"(dep)); }

__device__ __forceinline__ float dot3(F3 a, F3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
__device__ __forceinline__ float rsq(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ unsigned long long ballot(bool c) { return __builtin_amdgcn_ballot_w64(c); }
__device__ __forceinline__ bool lane_of(unsigned long long m) { return __builtin_amdgcn_inverse_ballot_w64(m); }

// A per-lane 16-byte load that is issued where it is written (the compiler otherwise sinks a plain load next to its first
// use, inside a branch) and waited for only at its first use.  Not a volatile load: on gfx950 that is a cache-bypassing
// flat load followed by an immediate wait.  A relaxed atomic load at wavefront scope is an ordinary global_load that the
// optimiser neither sinks nor duplicates.
__device__ __forceinline__ float4 load_f4(const float4* p)
{
    const uint64_t* q = reinterpret_cast<const uint64_t*>(p);
    const uint64_t lo = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    const uint64_t hi = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    return make_float4(__uint_as_float((uint32_t)lo), __uint_as_float((uint32_t)(lo >> 32)),
                       __uint_as_float((uint32_t)hi), __uint_as_float((uint32_t)(hi >> 32)));
}

// ... and 12 bytes (no fourth word: a destination register nobody reads is reused at once, and the reuse waits for the load)
__device__ __forceinline__ F3 load_f3(const float* p)
{
    const uint64_t lo = __hip_atomic_load(reinterpret_cast<const uint64_t*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    const uint32_t z = __hip_atomic_load(reinterpret_cast<const uint32_t*>(p) + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    return F3{__uint_as_float((uint32_t)lo), __uint_as_float((uint32_t)(lo >> 32)), __uint_as_float(z)};
}

// lane `l`'s value of v as a wave-uniform value
__device__ __forceinline__ float readlane_f(float v, int l)
{ return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); }

// an unspecified (but, for the compiler, defined) value: for registers whose every use is masked until they are written
// (volatile: one register per use, not one shared value that then has to be copied)
__device__ __forceinline__ void any_value(float& v) { asm volatile("" : "=v"(v)); }

// lane-wise m ? a : b for a lane mask m, as one v_cndmask
__device__ __forceinline__ uint32_t select_u32(unsigned long long m, uint32_t a, uint32_t b)
{
    uint32_t r;
    asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(m));
    return r;
}

// max of two values that are known not to be NaN: fmaxf() spends an extra instruction per operand on quieting NaNs
__device__ __forceinline__ float fmax_raw(float a, float b)
{
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

__device__ __forceinline__ float fmin_raw(float a, float b)
{
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// largest value of v (>= 0, no NaN) over the wave's 64 lanes, as a wave-uniform value: DPP steps inside rows of 16
// lanes, then across rows; lane 63 ends up with the maximum.  (Each step reads what the previous one wrote: the
// two wait states a DPP read needs after a VALU write are spelled out, the compiler does not look inside.)
__device__ __forceinline__ float wave_max(float v)
{
    asm volatile("s_nop 1\n"
                 "v_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                 "s_nop 1\n"
                 "v_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
                 "s_nop 1\n"
                 "v_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n"
                 "s_nop 1\n"
                 "v_max_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n"
                 "s_nop 1\n"
                 "v_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n"
                 "s_nop 1\n"
                 "v_max_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n"
                 "s_nop 1"
                 : "+v"(v));
    return readlane_f(v, 63);
}

// why a pixel was queued (debug build of the kernel only: PTR_DEBUG_REASONS)
enum : unsigned { WHY_PRIMARY = 1, WHY_CC = 2, WHY_ORDER = 4, WHY_FAR = 8, WHY_SHADOW = 16, WHY_ROUND = 32 };

// NB: lane masks are wave-uniform values; they may only be updated in wave-uniform control flow
// (never inside an exec-masked block), or the compiler turns them into per-lane values.
template <bool DBG> struct Unsure;
template <> struct Unsure<false> {
    unsigned long long mask = 0ull;
    __device__ __forceinline__ void add_mask(unsigned long long m, unsigned) { mask |= m; }
    __device__ __forceinline__ bool any() const { return lane_of(mask); }
    __device__ __forceinline__ unsigned why() const { return 0u; }
};
template <> struct Unsure<true> {
    unsigned bits = 0u;
    __device__ __forceinline__ void add_mask(unsigned long long m, unsigned bit) { bits |= lane_of(m) ? bit : 0u; }
    __device__ __forceinline__ bool any() const { return bits != 0u; }
    __device__ __forceinline__ unsigned why() const { return bits; }
};

}  // namespace ptr
