// MFMA operand types and conversions shared by the field kernels (field.hip) and the stage-1 weight-gradient kernel (stage1.hip).
#pragma once

#include "common.h"

namespace nsig {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// A B operand of one K-step (8 values per lane) as split bf16: hi = bf16(v), lo = bf16(v - hi), two values per dword.
struct Split8 {
    uint32_t hi[4], lo[4];
};
// ... and as plain fp16, two values per dword.
struct Half8 {
    uint32_t v[4];
};
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ inline uint32_t cvt_pk_bf16(float a, float b) {   // v_cvt_pk_bf16_f32: round to nearest even, a in the low half
    const f32x2 v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}
__device__ inline uint32_t cvt_pk_f16(float a, float b) {    // round to nearest even, a in the low half
    const f32x2 v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2));
}
__device__ inline bf16x8 operand(const uint32_t (&w)[4]) {
    const uint4 u = {w[0], w[1], w[2], w[3]};
    return __builtin_bit_cast(bf16x8, u);
}
__device__ inline f16x8 operand_h(const uint32_t (&w)[4]) {
    const uint4 u = {w[0], w[1], w[2], w[3]};
    return __builtin_bit_cast(f16x8, u);
}

// eight fp32 values -> one split-bf16 operand; c += a . b as three MFMAs, lo parts first, hi * hi last (the order of Bf16x3::mac, fieldmlp.h)
__device__ inline void split8(const float (&v)[8], Split8 &s) {
#pragma unroll
    for (int jp = 0; jp < 4; ++jp) {
        const uint32_t hi = cvt_pk_bf16(v[2 * jp], v[2 * jp + 1]);
        s.hi[jp] = hi;
        s.lo[jp] = cvt_pk_bf16(v[2 * jp] - __uint_as_float(hi << 16), v[2 * jp + 1] - __uint_as_float(hi & 0xffff0000u));
    }
}

__device__ inline f32x16 mac3(const Split8 &a, const Split8 &b, f32x16 c) {      // lo parts first, hi * hi last (as Bf16x3::mac)
    const bf16x8 a_hi = operand(a.hi), a_lo = operand(a.lo), b_hi = operand(b.hi), b_lo = operand(b.lo);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_lo, b_hi, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, b_lo, c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, b_hi, c, 0, 0, 0);
}

// stage1.hip: the slab sums of the weight-gradient kernels (k_field_wgrad, k_field_bwd_wgrad) in workgroup order -> tcnn's parameter layout
int wgrad_reduce_launch(const float *slabs, uint32_t n_wg, float *grad_sigma_params, float *grad_color_params, hipStream_t st, const char *what);

// row of a 32-row accumulator block held in register r (0..15) of lane half h (v_mfma_f32_32x32x16: column = lane & 31)
__host__ __device__ inline int row_of_reg16(int h, int r) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

}  // namespace nsig
