// Split-operand MFMA helpers of the graph-conv kernels (modes "f32_split_gcn" / step_gcn_*_split): an fp32 value v is
// hi = bf16(v) plus lo = bf16(v - hi) (16 mantissa bits together) and a product a*b is taken as
// a_hi*b_hi + a_lo*b_hi + a_hi*b_lo on the bf16 matrix cores with fp32 accumulation (the dropped a_lo*b_lo is 2^-16
// relative).
#pragma once
#include "common.h"
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
// A-role operand: (hi[4], lo[4]) along k; B-role operand: (hi, hi) and (lo, 0).  With K = 32 per instruction
// (v_mfma_f32_16x16x32_bf16, the full-rate shape; lane (q, .) holds k = 8q .. 8q+7 = the four k-steps x {hi, lo}):
//   (a_hi, a_lo) . (b_hi, b_hi) = a_hi*b_hi + a_lo*b_hi      (a_hi, a_lo) . (b_lo, 0) = a_hi*b_lo
struct SplitA { bf16x8_t hl; };
struct SplitB { bf16x8_t hh, l0; };
static __device__ __forceinline__ SplitA split_a(const f32x4 v) {
    const bf16x4_t h = __builtin_convertvector(v, bf16x4_t);
    const bf16x4_t l = __builtin_convertvector(v - __builtin_convertvector(h, f32x4), bf16x4_t);
    return SplitA{__builtin_shufflevector(h, l, 0, 1, 2, 3, 4, 5, 6, 7)};
}
static __device__ __forceinline__ SplitB split_b(const f32x4 v) {
    const bf16x4_t h = __builtin_convertvector(v, bf16x4_t);
    const bf16x4_t l = __builtin_convertvector(v - __builtin_convertvector(h, f32x4), bf16x4_t);
    const bf16x4_t z = {(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
    return SplitB{__builtin_shufflevector(h, h, 0, 1, 2, 3, 0, 1, 2, 3), __builtin_shufflevector(l, z, 0, 1, 2, 3, 4, 5, 6, 7)};
}
static __device__ __forceinline__ f32x4 mfma_split(const SplitA& a, const SplitB& b, f32x4 c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.hl, b.l0, c, 0, 0, 0);        // the small term first
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.hl, b.hh, c, 0, 0, 0);
}

