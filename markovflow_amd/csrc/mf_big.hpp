// Large state dimension (fp32: 10 <= d <= 64, fp64: 10 <= d <= 32): the Kalman log-likelihood with ONE WORKGROUP per (series, time-chunk),
// every d x d matrix resident in LDS and every GEMM-shaped product on the matrix cores
// (v_mfma_f32_16x16x4_f32: f32 in, f32 accumulate - exact f32, BASELINE config 5: d = 64, T = 2048).
//
// Same algorithm as the small-d kernels (mf_kernels.hpp: partitioned block elimination with a spike towards the
// chunk's left neighbour, reduced system handed to the next level), re-expressed on 16 x 16 tiles:
//   * matrices live in LDS as DP x DP row-major images (DP = d rounded up to 16, row stride DP + 4 floats); a state
//     dimension that is not a multiple of 16 is padded with an identity block (independent N(0,1) states that add
//     nothing to the log-determinant or the quadratic form);
//   * triangular solves are products with explicit inverses: the 16 x 16 diagonal tiles are factored/inverted by one
//     wavefront in registers (row per lane, pivots broadcast with v_readlane), everything off the diagonal is MFMA;
//   * seven tiles (119 KB at d = 64) + vectors: one workgroup of 256 threads per CU.
// Notation follows SURVEY.md Appendix B / mf_kernels.hpp.
#pragma once
#include "mf_kernels.hpp"
#include "mf_wave_api.hpp"

#include "mf_env.hpp"
#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace mf {
namespace bigcommon {
typedef float v4f __attribute__((ext_vector_type(4)));
typedef double v4d __attribute__((ext_vector_type(4)));
// D = A B + C on one 16 x 16 x 4 matrix-core instruction; accumulator register e of lane (r, q) holds column r of row
// 4 q + e (f32) or q + 4 e (v_mfma_f64_16x16x4_f64)
__device__ __forceinline__ v4f mfma(float a, float b, v4f c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ v4d mfma(double a, double b, v4d c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
__device__ __forceinline__ float bcast(float v, int src) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}
__device__ __forceinline__ double bcast(double v, int src) {
    const long long b = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, src);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), src);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
// The same broadcast WITHIN each row of 16 lanes (DPP row_newbcast): for data that is replicated across the four rows of a
// wavefront (the diagonal-tile kernels: lane l works on row l & 15).  One v_mov_b32_dpp per dword instead of a v_readlane into an
// SGPR (whose consumer has to wait out the VALU-writes-SGPR hazard); `src` must be a compile-time constant after unrolling.
template <int SRC> __device__ __forceinline__ int dpp_row_bcast(int v) {
    return __builtin_amdgcn_update_dpp(0, v, 0x150 + SRC, 0xF, 0xF, false);
}
__device__ __forceinline__ int bcast16_bits(int v, int src) {
    switch (src & 15) {
        case 0: return dpp_row_bcast<0>(v);   case 1: return dpp_row_bcast<1>(v);   case 2: return dpp_row_bcast<2>(v);
        case 3: return dpp_row_bcast<3>(v);   case 4: return dpp_row_bcast<4>(v);   case 5: return dpp_row_bcast<5>(v);
        case 6: return dpp_row_bcast<6>(v);   case 7: return dpp_row_bcast<7>(v);   case 8: return dpp_row_bcast<8>(v);
        case 9: return dpp_row_bcast<9>(v);   case 10: return dpp_row_bcast<10>(v); case 11: return dpp_row_bcast<11>(v);
        case 12: return dpp_row_bcast<12>(v); case 13: return dpp_row_bcast<13>(v); case 14: return dpp_row_bcast<14>(v);
        default: return dpp_row_bcast<15>(v);
    }
}
__device__ __forceinline__ float bcast16(float v, int src) { return __int_as_float(bcast16_bits(__float_as_int(v), src)); }
__device__ __forceinline__ double bcast16(double v, int src) {
    const long long b = __double_as_longlong(v);
    const unsigned lo = (unsigned)bcast16_bits((int)(unsigned)b, src);
    const unsigned hi = (unsigned)bcast16_bits((int)(unsigned)(b >> 32), src);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
// The broadcast as the DPP operand of the consuming instruction itself (fp32): no SGPR (with v_readlane the sixteen steps of a
// diagonal tile keep > 100 broadcast values in SGPRs and the compiler spills them to VGPR lanes: v_writelane / v_readlane per
// value) and no extra VGPR.  Inline asm is opaque to the hazard recogniser, so the two wait states a DPP read needs after a VALU
// write of the same VGPR (the producing instruction, or a copy / AGPR reload the register allocator put in front) are part of
// every asm statement: one s_nop 1 in front of the DPP instructions of a statement, none of which writes a DPP source.
template <int K> __device__ __forceinline__ float mov_bcast16(float s) {
    float d;
    asm("s_nop 1\n\tv_mov_b32_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(d) : "v"(s), "n"(K));
    return d;
}
// d -= bcast16(s, K) * t
template <int K> __device__ __forceinline__ void fnma_bcast16(float& d, float s, float t) {
    asm("s_nop 1\n\tv_fmac_f32_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(d) : "v"(s), "v"(t), "n"(K));
}
// v[k] -= bcast16(s, k) * t for k = K0 ... 15 as ONE statement (one s_nop for up to fifteen DPP instructions).  Generated:
//   for K0 in 1..15: outputs %0.. = v[K0..15], then s, t
template <int K0> __device__ __forceinline__ void fnma_bcast16_from(float (&v)[16], float s, float t) {
#define MF_DPP_FMA(i, k) "\n\tv_fmac_f32_dpp %" #i ", -%[s], %[t] row_newbcast:" #k " row_mask:0xf bank_mask:0xf"
    if constexpr (K0 == 1)
        asm("s_nop 1" MF_DPP_FMA(0, 1) MF_DPP_FMA(1, 2) MF_DPP_FMA(2, 3) MF_DPP_FMA(3, 4) MF_DPP_FMA(4, 5) MF_DPP_FMA(5, 6) MF_DPP_FMA(6, 7) MF_DPP_FMA(7, 8) MF_DPP_FMA(8, 9) MF_DPP_FMA(9, 10) MF_DPP_FMA(10, 11) MF_DPP_FMA(11, 12) MF_DPP_FMA(12, 13) MF_DPP_FMA(13, 14) MF_DPP_FMA(14, 15)
            : "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]) : [s] "v"(s), [t] "v"(t));
    else if constexpr (K0 == 2)
        asm("s_nop 1" MF_DPP_FMA(0, 2) MF_DPP_FMA(1, 3) MF_DPP_FMA(2, 4) MF_DPP_FMA(3, 5) MF_DPP_FMA(4, 6) MF_DPP_FMA(5, 7) MF_DPP_FMA(6, 8) MF_DPP_FMA(7, 9) MF_DPP_FMA(8, 10) MF_DPP_FMA(9, 11) MF_DPP_FMA(10, 12) MF_DPP_FMA(11, 13) MF_DPP_FMA(12, 14) MF_DPP_FMA(13, 15)
            : "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]) : [s] "v"(s), [t] "v"(t));
    else if constexpr (K0 == 3)
        asm("s_nop 1" MF_DPP_FMA(0, 3) MF_DPP_FMA(1, 4) MF_DPP_FMA(2, 5) MF_DPP_FMA(3, 6) MF_DPP_FMA(4, 7) MF_DPP_FMA(5, 8) MF_DPP_FMA(6, 9) MF_DPP_FMA(7, 10) MF_DPP_FMA(8, 11) MF_DPP_FMA(9, 12) MF_DPP_FMA(10, 13) MF_DPP_FMA(11, 14) MF_DPP_FMA(12, 15)
            : "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]) : [s] "v"(s), [t] "v"(t));
    else if constexpr (K0 == 4)
        asm("s_nop 1" MF_DPP_FMA(0, 4) MF_DPP_FMA(1, 5) MF_DPP_FMA(2, 6) MF_DPP_FMA(3, 7) MF_DPP_FMA(4, 8) MF_DPP_FMA(5, 9) MF_DPP_FMA(6, 10) MF_DPP_FMA(7, 11) MF_DPP_FMA(8, 12) MF_DPP_FMA(9, 13) MF_DPP_FMA(10, 14) MF_DPP_FMA(11, 15)
            : "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]) : [s] "v"(s), [t] "v"(t));
    else if constexpr (K0 == 5)
        asm("s_nop 1" MF_DPP_FMA(0, 5) MF_DPP_FMA(1, 6) MF_DPP_FMA(2, 7) MF_DPP_FMA(3, 8) MF_DPP_FMA(4, 9) MF_DPP_FMA(5, 10) MF_DPP_FMA(6, 11) MF_DPP_FMA(7, 12) MF_DPP_FMA(8, 13) MF_DPP_FMA(9, 14) MF_DPP_FMA(10, 15)
            : "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]) : [s] "v"(s), [t] "v"(t));
    else if constexpr (K0 == 6)
        asm("s_nop 1" MF_DPP_FMA(0, 6) MF_DPP_FMA(1, 7) MF_DPP_FMA(2, 8) MF_DPP_FMA(3, 9) MF_DPP_FMA(4, 10) MF_DPP_FMA(5, 11) MF_DPP_FMA(6, 12) MF_DPP_FMA(7, 13) MF_DPP_FMA(8, 14) MF_DPP_FMA(9, 15)
            : "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]) : [s] "v"(s), [t] "v"(t));
    else if constexpr (K0 == 7)
        asm("s_nop 1" MF_DPP_FMA(0, 7) MF_DPP_FMA(1, 8) MF_DPP_FMA(2, 9) MF_DPP_FMA(3, 10) MF_DPP_FMA(4, 11) MF_DPP_FMA(5, 12) MF_DPP_FMA(6, 13) MF_DPP_FMA(7, 14) MF_DPP_FMA(8, 15)
            : "+v"(v[7]), "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]) : [s] "v"(s), [t] "v"(t));
    else if constexpr (K0 == 8)
        asm("s_nop 1" MF_DPP_FMA(0, 8) MF_DPP_FMA(1, 9) MF_DPP_FMA(2, 10) MF_DPP_FMA(3, 11) MF_DPP_FMA(4, 12) MF_DPP_FMA(5, 13) MF_DPP_FMA(6, 14) MF_DPP_FMA(7, 15)
            : "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]) : [s] "v"(s), [t] "v"(t));
    else if constexpr (K0 == 9)
        asm("s_nop 1" MF_DPP_FMA(0, 9) MF_DPP_FMA(1, 10) MF_DPP_FMA(2, 11) MF_DPP_FMA(3, 12) MF_DPP_FMA(4, 13) MF_DPP_FMA(5, 14) MF_DPP_FMA(6, 15)
            : "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]) : [s] "v"(s), [t] "v"(t));
    else if constexpr (K0 == 10)
        asm("s_nop 1" MF_DPP_FMA(0, 10) MF_DPP_FMA(1, 11) MF_DPP_FMA(2, 12) MF_DPP_FMA(3, 13) MF_DPP_FMA(4, 14) MF_DPP_FMA(5, 15)
            : "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]) : [s] "v"(s), [t] "v"(t));
    else if constexpr (K0 == 11)
        asm("s_nop 1" MF_DPP_FMA(0, 11) MF_DPP_FMA(1, 12) MF_DPP_FMA(2, 13) MF_DPP_FMA(3, 14) MF_DPP_FMA(4, 15)
            : "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]) : [s] "v"(s), [t] "v"(t));
    else if constexpr (K0 == 12)
        asm("s_nop 1" MF_DPP_FMA(0, 12) MF_DPP_FMA(1, 13) MF_DPP_FMA(2, 14) MF_DPP_FMA(3, 15)
            : "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]) : [s] "v"(s), [t] "v"(t));
    else if constexpr (K0 == 13)
        asm("s_nop 1" MF_DPP_FMA(0, 13) MF_DPP_FMA(1, 14) MF_DPP_FMA(2, 15)
            : "+v"(v[13]), "+v"(v[14]), "+v"(v[15]) : [s] "v"(s), [t] "v"(t));
    else if constexpr (K0 == 14)
        asm("s_nop 1" MF_DPP_FMA(0, 14) MF_DPP_FMA(1, 15)
            : "+v"(v[14]), "+v"(v[15]) : [s] "v"(s), [t] "v"(t));
    else if constexpr (K0 == 15)
        asm("s_nop 1" MF_DPP_FMA(0, 15)
            : "+v"(v[15]) : [s] "v"(s), [t] "v"(t));
#undef MF_DPP_FMA
}
// compile-time loop: f(std::integral_constant<int, I>) for I = BEGIN ... END-1 (the DPP lane select is an immediate)
template <int BEGIN, int END, typename F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (BEGIN < END) {
        f(std::integral_constant<int, BEGIN>{});
        static_for<BEGIN + 1, END>(f);
    }
}
// natural logarithm for the running log-determinant: the hardware log2 (one instruction, ~1 ulp of fp32) in fp32
__device__ __forceinline__ float mf_log(float x) { return __log2f(x) * 0.6931471805599453094f; }
__device__ __forceinline__ double mf_log(double x) { return log(x); }
}  // namespace bigcommon
}  // namespace mf

#define MF_BIG_T float
#define MF_BIG_NS big
#include "mf_big_impl.hpp"
#include "mf_bigops_impl.hpp"
#include "mf_bigpar_impl.hpp"
#include "mf_biggrad_impl.hpp"
#undef MF_BIG_T
#undef MF_BIG_NS
// fp64: v_mfma_f64_16x16x4_f64 (same operand maps, accumulator rows q + 4 e); seven tiles fit up to DP = 32
#define MF_BIG_T double
#define MF_BIG_NS bigd
#include "mf_big_impl.hpp"
#include "mf_bigops_impl.hpp"
#include "mf_bigpar_impl.hpp"
#include "mf_biggrad_impl.hpp"
#undef MF_BIG_T
#undef MF_BIG_NS
