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

// row of a 32-row accumulator block held in register r (0..15) of lane half h (v_mfma_f32_32x32x16: column = lane & 31)
__host__ __device__ inline int row_of_reg16(int h, int r) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

}  // namespace nsig
