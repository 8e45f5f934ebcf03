// Building blocks of the field MLP kernels shared by field.hip and stage1_fused.hip: the packed weight layout, the two MFMA precisions,
// the wave-level layer / ReLU / mask helpers, the SH encoding and the stage-1 trace records.  (Moved out of field.hip unchanged.)
#pragma once

#include <type_traits>

#include "hashgrid.h"
#include "mfma.h"

namespace nsig {

// ----------------------------------------------------------------------------- packed weight layout

// Forward A fragments (W as stored by tcnn: [out,in]) and backward A fragments (W^T).
constexpr int F0 = 0;   // sigma L1   W1s[64x32]      2 row blocks x 2 k-steps
constexpr int F1 = 4;   // sigma L2   W2s[16x64]      1 x 4
constexpr int F2 = 8;   // color L1   Wc1[64x32]      2 x 2
constexpr int F3 = 12;  // color L2   Wc2[64x64]      2 x 4
constexpr int F4 = 20;  // color L3   Wc3[16x64]      1 x 4
constexpr int kFwdFrags = 24;
constexpr int B0 = 0;   // Wc3^T [64x16]              2 x 1
constexpr int B1 = 2;   // Wc2^T [64x64]              2 x 4
constexpr int B2 = 10;  // Wc1^T rows -> sigma-out    1 x 4
constexpr int B3 = 14;  // W2s^T [64x16]              2 x 1
constexpr int B4 = 16;  // W1s[:,30:32]^T             1 x 4
constexpr int B4F = 20; // W1s^T, all 32 features     1 x 4   (stage-1 training: gradients of the base tables)
constexpr int kBwdFrags = 24;
constexpr int kFragBytes = 64 * 16;  // 64 lanes x 8 bf16
// packed = [fwd hi | fwd lo | bwd hi | bwd lo] (bf16) | [fwd | bwd] (fp16)
constexpr size_t kFwdBytes = (size_t)kFwdFrags * kFragBytes, kBwdBytes = (size_t)kBwdFrags * kFragBytes;
constexpr size_t kPackedBf16Bytes = 2 * kFwdBytes + 2 * kBwdBytes;
constexpr size_t kPackedBytes = kPackedBf16Bytes + kFwdBytes + kBwdBytes;

constexpr int kSigmaW1 = 0, kSigmaW2 = 2048;                 // offsets in sigma_params (3072)
constexpr int kColorW1 = 0, kColorW2 = 2048, kColorW3 = 6144;  // offsets in color_params (7168)

// K index carried by element j of lane half h in k-step ks when the B operand is the previous layer's
// accumulator (registers 8s..8s+7 of row block rb, ks = 2 rb + s).
__host__ __device__ inline int k_from_acc(int ks, int h, int j) { return 32 * (ks >> 1) + 16 * (ks & 1) + 8 * (j >> 2) + 4 * h + (j & 3); }
// Row of a 32-row accumulator block held in register r (< 8) of lane half h.
__host__ __device__ inline int row_of_reg(int h, int r) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// ----------------------------------------------------------------------------- wave-level building blocks

// The two precisions of the MFMA kernels.  A tag names the operand type, how a pair of fp32 values enters it, where the A
// fragments of the forward / backward weights sit in the packed image and how many LDS bytes they take.
struct Bf16x3 {
    typedef Split8 Op;
    static constexpr int kMfmaPerProduct = 3;
    static constexpr size_t kFwdOffset = 0, kBwdOffset = 2 * kFwdBytes, kFwdLds = 2 * kFwdBytes, kBwdLds = 2 * kBwdBytes;
    // elements 2*jp, 2*jp + 1 of the operand: 6 VALU operations per pair, results already packed
    __device__ static inline void put2(Split8 &s, int jp, float a, float b) {
        const uint32_t hi = cvt_pk_bf16(a, b);
        s.hi[jp] = hi;
        s.lo[jp] = cvt_pk_bf16(a - __uint_as_float(hi << 16), b - __uint_as_float(hi & 0xffff0000u));
    }
    // c += A(fragment at byte offset off) . B: lo parts first, hi*hi last
    __device__ static inline f32x16 mac(const char *__restrict__ lds, size_t half_bytes, int off, const Split8 &b, f32x16 c) {
        const bf16x8 a_hi = *reinterpret_cast<const bf16x8 *>(lds + off);
        const bf16x8 a_lo = *reinterpret_cast<const bf16x8 *>(lds + half_bytes + off);
        const bf16x8 b_hi = operand(b.hi), b_lo = operand(b.lo);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_lo, b_hi, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, b_lo, c, 0, 0, 0);
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, b_hi, c, 0, 0, 0);
    }
};
struct F16 {
    typedef Half8 Op;
    static constexpr int kMfmaPerProduct = 1;
    static constexpr size_t kFwdOffset = kPackedBf16Bytes, kBwdOffset = kPackedBf16Bytes + kFwdBytes, kFwdLds = kFwdBytes, kBwdLds = kBwdBytes;
    __device__ static inline void put2(Half8 &s, int jp, float a, float b) { s.v[jp] = cvt_pk_f16(a, b); }
    __device__ static inline f32x16 mac(const char *__restrict__ lds, size_t, int off, const Half8 &b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const f16x8 *>(lds + off), operand_h(b.v), c, 0, 0, 0);
    }
};

// acc[rb] = sum over k-steps of A(frag0 + rb*KS + ks) . B[ks].  lds: the staged fragments; half_bytes: distance from the hi to the
// lo fragments (Bf16x3 only).
template <typename P, int RB, int KS>
__device__ inline void mfma_layer(const char *__restrict__ lds, size_t half_bytes, int frag0, int lane, const typename P::Op (&b)[KS], f32x16 (&acc)[RB]) {
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        f32x16 c = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) c = P::mac(lds, half_bytes, (frag0 + rb * KS + ks) * kFragBytes + lane * 16, b[ks], c);
        acc[rb] = c;
    }
}

// Where the "was positive" flag of activation i = 16*rb + r (row block rb, accumulator register r) sits in the 32-bit mask word.
// Bf16x3: bit i.  F16: the flags are the inverted fp16 SIGN bits of the packed operand dwords, collected pairwise -- the pair
// k = 8*rb + r/2 lands with its even element in bit k and its odd element in bit 16 + k (relu_to_operand<F16>).
template <typename P>
__host__ __device__ constexpr int mask_bit(int i) {
    return P::kMfmaPerProduct == 1 ? ((i & 1) ? 16 + ((i >> 4) * 8 + ((i & 15) >> 1)) : ((i >> 4) * 8 + ((i & 15) >> 1))) : i;
}

// fp16: a layer's A fragments fetched from LDS into registers in one burst, and the layer evaluated from them (see field_fwd_pipelined)
template <int N>
__device__ inline void load_frags(const char *__restrict__ lds, int frag0, int lane, f16x8 (&a)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) a[i] = *reinterpret_cast<const f16x8 *>(lds + (frag0 + i) * kFragBytes + lane * 16);
}
template <int RB, int KS>
__device__ inline void mfma_regs(const f16x8 (&a)[RB * KS], const Half8 (&b)[KS], f32x16 (&acc)[RB]) {
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        f32x16 c = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[rb * KS + ks], operand_h(b[ks].v), c, 0, 0, 0);
        acc[rb] = c;
    }
}

// ReLU a 64-row activation held in two accumulators, return the "was positive" bits (bit mask_bit<P>(16*rb + r)), emit the next B operand.
// No compares (they would hold 32 lane masks in SGPRs): the value is max(v, 0); the flag is the sign bit of 0 - bits(v) --
// set exactly when v > 0, because an accumulator that starts at +0 never holds -0 -- shifted in with one v_alignbit.
template <typename P>
__device__ inline uint32_t relu_to_operand(const f32x16 (&acc)[2], typename P::Op (&b)[4]) {
    if constexpr (P::kMfmaPerProduct == 1) {
        // fp16 operands: round the PAIR first (one v_cvt_pk), clamp it packed (one v_pk_max_f16), and take the flags from the packed
        // value's two sign bits -- shifted into a running word with one v_lshrrev + one v_bfi: four vector instructions per pair
        // instead of seven (two subtractions, two v_alignbit, two integer max, one pack).  Rounding is monotonic, so the operand is the
        // same value as max(v, 0) rounded; the flag of an activation that is exactly +0 reads "positive" here (its upstream
        // gradient then passes a ReLU whose output was 0 either way: only all-zero padding rows have exact zeros, and their
        // gradients are zero).
        uint32_t neg = 0;
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const uint32_t x = cvt_pk_f16(acc[rb][r], acc[rb][r + 1]);
                const f16x2 zero = {(_Float16)0.0f, (_Float16)0.0f};
                b[2 * rb + (r >> 3)].v[(r & 7) >> 1] = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(f16x2, x), zero));
                neg = (x & 0x80008000u) | ((neg >> 1) & 0x7fff7fffu);      // v_bfi_b32: after the 16th pair, pair k sits in bits k and 16 + k
            }
        return ~neg;
    } else {
        uint32_t bits = 0;   // filled most-significant-first, reversed at the end: index i ends in bit i
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                uint32_t u0 = __float_as_uint(acc[rb][r]), u1 = __float_as_uint(acc[rb][r + 1]);
                asm("" : "+v"(u0), "+v"(u1));   // one copy out of the accumulator registers, used twice
                bits = __builtin_amdgcn_alignbit(bits, 0u - u0, 31);
                bits = __builtin_amdgcn_alignbit(bits, 0u - u1, 31);
                // max(v, 0) as an integer max on the bit patterns (negative floats are negative integers): no NaN canonicalisation
                P::put2(b[2 * rb + (r >> 3)], (r & 7) >> 1, __uint_as_float((uint32_t)max((int)u0, 0)), __uint_as_float((uint32_t)max((int)u1, 0)));
            }
        return __builtin_bitreverse32(bits);
    }
}

// Backward through a ReLU: zero the rows whose forward activation was clamped, emit the next B operand.
__device__ inline float masked(float v, uint32_t bits, int i) {   // v where bit i is set, else +0: sign-extended 1-bit field as AND mask
    uint32_t m;   // inline asm: the compiler would turn the builtin into a compare + select (and a lane mask in SGPRs) again
    asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(m) : "v"(bits), "n"(i));
    return __uint_as_float(__float_as_uint(v) & m);
}
template <typename P>
__device__ inline void mask_to_operand(const f32x16 (&acc)[2], uint32_t bits, typename P::Op (&b)[4]) {
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int r = 0; r < 16; r += 2)
            P::put2(b[2 * rb + (r >> 3)], (r & 7) >> 1, masked(acc[rb][r], bits, mask_bit<P>(rb * 16 + r)), masked(acc[rb][r + 1], bits, mask_bit<P>(rb * 16 + r + 1)));
}

// Degree-4 real spherical harmonics (the 16 components of hash_encoding.py:157-183) of d in [-1,1]^3.
__device__ inline void sh16(float x, float y, float z, float (&o)[16]) {
    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    o[0] = 0.28209479177387814f;
    o[1] = -0.4886025119029199f * y;
    o[2] = 0.4886025119029199f * z;
    o[3] = -0.4886025119029199f * x;
    o[4] = 1.0925484305920792f * xy;
    o[5] = -1.0925484305920792f * yz;
    o[6] = 0.31539156525252005f * (2.0f * zz - xx - yy);
    o[7] = -1.0925484305920792f * xz;
    o[8] = 0.5462742152960396f * (xx - yy);
    o[9] = -0.5900435899266435f * y * (3.0f * xx - yy);
    o[10] = 2.890611442640554f * xy * z;
    o[11] = -0.4570457994644658f * y * (4.0f * zz - xx - yy);
    o[12] = 0.3731763325901154f * z * (2.0f * zz - 3.0f * xx - 3.0f * yy);
    o[13] = -0.4570457994644658f * x * (4.0f * zz - xx - yy);
    o[14] = 1.445305721320277f * z * (xx - yy);
    o[15] = -0.5900435899266435f * x * (xx - 3.0f * yy);
}

__device__ inline void stage_weights(char *lds, const char *__restrict__ src, int bytes) {
    for (int o = threadIdx.x * 16; o < bytes; o += blockDim.x * 16)
        *reinterpret_cast<uint4 *>(lds + o) = *reinterpret_cast<const uint4 *>(src + o);
    __syncthreads();
}

// Stage-1 (clean model) training needs the gradients of every weight matrix and of all 32 encoder features.  The weight
// gradients are reductions over all points of (pre-activation gradient) x (layer input) -- plain GEMMs -- so the forward
// optionally saves each layer's input and the backward each layer's pre-activation gradient, feature-major
// ([width][stride] fp32: a wave stores 128 contiguous bytes per row), and the host reduces them with library GEMMs.
struct ActTrace {    // written by the forward
    float *hs;       // [64][stride] sigma hidden layer, post-ReLU
    float *cin;      // [32][stride] colour input: 16 SH, 15 geometry features, the padded 1.0
    float *h1, *h2;  // [64][stride] colour hidden layers, post-ReLU
};
struct GradTrace {   // written by the backward
    float *d_hs, *d_h1, *d_h2;  // [64][stride] pre-activation gradients of the hidden layers
    float *d_so, *d_out;        // [16][stride] gradients of the two heads' outputs (sigma head rows 0..15; colour rows 0..2)
    float2 *d_planes;           // [16][stride] gradient of the 32 encoder features, level-major like the forward's planes
};

// base[uniform + lane]: the wave-uniform part of the index pinned into SGPRs (readfirstlane), the lane part a 32-bit BYTE offset -- the load or store is then
// `global_load v, v_off, s[base:base+1]`: ONE VGPR of address for all accesses of a request.  Left to itself the compiler folds the lane offset into each
// access's row base and keeps one 64-bit lane address per access live over the whole tile loop as a loop invariant (the 112 trace stores of the stage-1 forward: 224 VGPRs).  `base` stays the kernel
// argument it is (a pointer rebuilt from an integer would be a FLAT one).
template <typename T>
__device__ inline T *at_uniform(T *base, size_t uniform_elems, uint32_t lane_bytes) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)uniform_elems);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uint64_t)uniform_elems >> 32));
    typedef std::conditional_t<std::is_const<T>::value, const char, char> Byte;
    return reinterpret_cast<T *>(reinterpret_cast<Byte *>(base + (((size_t)hi << 32) | lo)) + lane_bytes);
}

// rows 32 rb + row_of_reg16(h, r) of a feature-major trace, column s: the row's uniform part (h = 0) + the tile's first point as the SGPR base, the lane's half and
// point as a 32-bit byte offset (stride < 2^27: the entry points check).
// TT: the element type the trace is kept in -- float, or _Float16 (field_fwd_trace_f16: half the bytes; the reference's MLPs keep fp16 activations for their
// backward, tinycudann FullyFusedMLP): the ActTrace pointers then address _Float16 rows of the same [width][stride] shape.
// fp16 rows go out two POINTS per store: the forward is bound by the NUMBER of its trace stores (~100 per tile: a store per row and lane), so neighbouring lanes trade
// values -- of a register pair (rows R, R + 1) the even lane ends up with row R of points (p, p + 1), the odd lane with row R + 1 of points (p - 1, p) -- and each issues
// ONE 4-byte store per pair: 16 per layer instead of 32 (quad_perm [1, 0, 3, 2]: the lanes of a pair swap).  139 -> 129 us with the bytes halved, -> 115 us with the stores
// halved; the same elements in the same places.  (measured, rejected) the same pairing for fp32 rows (8-byte stores): the forward 132 -> 124 us, but the BACKWARD that reads
// the rows 181 -> 190 us and the step +2.5 %, twice on one box (profiles/r06_stage1_f16_traces_ab3_fp32_paired.txt): fp32 rows keep one 4-byte store per row and lane.
template <typename TT = float, typename F>
__device__ inline void store_rows64(float *__restrict__ dst_f32, uint32_t stride, uint32_t s, int h, const f32x16 (&acc)[2], F f) {
    TT *__restrict__ dst = reinterpret_cast<TT *>(dst_f32);
    const uint32_t s0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)s);      // lane 0 holds the tile's first point
    if constexpr (std::is_same<TT, _Float16>::value) {
        typedef TT pair_t __attribute__((ext_vector_type(2)));
        const uint32_t p = s - s0, odd = p & 1u;
        const uint32_t lane_bytes = ((4u * (uint32_t)h + odd) * stride + (p & ~1u)) * (uint32_t)sizeof(TT);
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const float a = f(acc[rb][r], rb * 16 + r), b = f(acc[rb][r + 1], rb * 16 + r + 1);
                const float got = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, odd ? a : b), 0xB1, 0xF, 0xF, false));
                const pair_t v = odd ? pair_t{(TT)got, (TT)b} : pair_t{(TT)a, (TT)got};
                __builtin_nontemporal_store(v, reinterpret_cast<pair_t *>(at_uniform(dst, (size_t)(32 * rb + row_of_reg16(0, r)) * stride + s0, lane_bytes)));
            }
    } else {
        const uint32_t lane_bytes = (4u * (uint32_t)h * stride + (s - s0)) * (uint32_t)sizeof(TT);
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int r = 0; r < 16; ++r)      // streaming stores: a trace is written once and read once, a kernel or more later, and is far larger than the L2 (same box: 148 -> 124 us)
                __builtin_nontemporal_store((TT)f(acc[rb][r], rb * 16 + r), at_uniform(dst, (size_t)(32 * rb + row_of_reg16(0, r)) * stride + s0, lane_bytes));
    }
}

template <typename P, typename TT = float>
__device__ inline void color_branch(const char *lds, int lane, int h, float dx, float dy, float dz,
                                    const float (&geo8)[8], uint32_t (&mask)[2], float (&rgb)[3], const ActTrace *trace = nullptr,
                                    uint32_t stride = 0, uint32_t s = 0) {
    constexpr size_t kHalf = kFwdBytes;
    // tcnn's SH encoding takes inputs in [0,1] and maps them back (network_wtmk_tcnn.py:114-115)
    const float ux = (dx + 1.0f) / 2.0f, uy = (dy + 1.0f) / 2.0f, uz = (dz + 1.0f) / 2.0f;
    float sh[16];
    sh16(ux * 2.0f - 1.0f, uy * 2.0f - 1.0f, uz * 2.0f - 1.0f, sh);
    typename P::Op cin[2];
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        P::put2(cin[0], j >> 1, h ? sh[8 + j] : sh[j], h ? sh[9 + j] : sh[j + 1]);
        P::put2(cin[1], j >> 1, geo8[j], geo8[j + 1]);
    }
    auto relu = [](float v, int) { return v > 0.0f ? v : 0.0f; };
    if (trace != nullptr) {
        constexpr uint32_t kEl = (uint32_t)sizeof(TT);
        TT *cin_rows = reinterpret_cast<TT *>(trace->cin);
        const uint32_t s0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)s), col = (s - s0) * kEl;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            __builtin_nontemporal_store((TT)(h ? sh[8 + j] : sh[j]), at_uniform(cin_rows, (size_t)j * stride + s0, 8u * (uint32_t)h * stride * kEl + col));      // row 8 h + j
            // slot (h, j) of the second K-step carries sigma-head row rho = row_of_reg(h, j) -> colour input 15 + rho; rho == 0 (h = 0, j = 0) -> the padded 1.0, row 31
            const uint32_t lane_rows = j == 0 ? (h ? 4u : 16u) : 4u * (uint32_t)h;
            __builtin_nontemporal_store((TT)geo8[j], at_uniform(cin_rows, (size_t)(15 + row_of_reg(0, j)) * stride + s0, lane_rows * stride * kEl + col));
        }
    }
    f32x16 hid[2];
    typename P::Op b4[4];
    mfma_layer<P, 2, 2>(lds, kHalf, F2, lane, cin, hid);
    mask[0] = relu_to_operand<P>(hid, b4);
    if (trace != nullptr) store_rows64<TT>(trace->h1, stride, s, h, hid, relu);
    mfma_layer<P, 2, 4>(lds, kHalf, F3, lane, b4, hid);
    mask[1] = relu_to_operand<P>(hid, b4);
    if (trace != nullptr) store_rows64<TT>(trace->h2, stride, s, h, hid, relu);
    f32x16 out[1];
    mfma_layer<P, 1, 4>(lds, kHalf, F4, lane, b4, out);
#pragma unroll
    for (int c = 0; c < 3; ++c) rgb[c] = 1.0f / (1.0f + expf(-out[0][c]));  // rows 0..2 live in lane half 0
}

}  // namespace nsig
