// Stage-1 (clean model) training, SURVEY.md 8(f) N3: the MLP backward with the five weight-gradient reductions INSIDE it.
//
// Behavioural reference: the autograd backward of /root/reference/nerf/network_hash.py:98-152 (sigma_net, color_net: every weight trainable, :154-166).
//
// field_bwd_trace + field_wgrad (field.hip, stage1.hip) write every layer's pre-activation gradient feature-major to memory (896 B per point) and read
// it back (with the layer inputs: 1955 B per point) for five split-K products: at 620 k points 158 + 298 us of a 1.43 ms step, both passes bound by
// those bytes.  Here a wave keeps its tile's pre-activation gradients on the chip:
//   * the backward chain of k_field_bwd<Bf16x3, kFull> as it is (same instructions, same bits in d_planes);
//   * each layer's gradient block -- in the accumulators column = point, rows in registers -- is written to a wave-private LDS scratch as
//     [row][point] and read back as MFMA A operands whose K dimension runs over the POINTS (lane (r, h): row r, points 8h..8h+7: two ds_read_b128
//     from rows padded to 36 floats, conflict-free); LDS instructions of one wave execute in program order, so the transposition needs no barrier;
//   * the layer's INPUT (field_fwd_trace's feature-major rows) comes from memory into the same scratch -- 8 lanes per 128-byte line, requested one
//     layer ahead into registers -- and is read back the same way as B operands;
//   * dW += A . B^T as split bf16 (hi + lo, three v_mfma_f32_32x32x16_bf16, fp32 accumulate: the gradients of an unscaled MSE sit around 1e-6, below
//     fp16's range), into TWELVE 32 x 32 accumulator blocks (192 registers) that a wave keeps over all its tiles.  That is why this kernel lives in a
//     translation unit of its own: it is compiled WITHOUT -amdgpu-mfma-vgpr-form (build.py), so that the persistent accumulators sit in the AGPR half
//     of the unified register file (launch bounds 256 threads = one wave per SIMD = 512 registers) and the chain's working set in the VGPR half.
//   * at the end the four waves of a workgroup add their blocks through LDS in a fixed order and store ONE slab in k_field_wgrad's layout;
//     k_wgrad_reduce (stage1.hip) adds the slabs in workgroup order and writes tcnn's parameter layout.  Fixed tile -> wave assignment: bit-reproducible.
// One workgroup per compute unit (48 KiB of split-bf16 backward weights + 4 x 18 KiB of scratch); nothing but d_planes (128 B per point, the table
// scatter's input) is written per point.
#include "fieldmlp.h"

namespace nsig {

constexpr uint32_t kFusedRowFloats = 36;                         // 32 points + 4 floats: consecutive rows start 4 banks apart (as k_field_wgrad's staging area)
constexpr uint32_t kFusedRowsX = 64, kFusedRowsDY = 64;          // a wave's scratch: one layer input | one pre-activation gradient
constexpr uint32_t kFusedWaveFloats = (kFusedRowsX + kFusedRowsDY) * kFusedRowFloats;      // 18 KiB
constexpr uint32_t kFusedMaxWGs = kCUs;
constexpr uint32_t kFusedBlocks = 12, kFusedSlab = kFusedBlocks * 1024u;      // = k_field_wgrad's 3 roles x 4 products x 16 registers x 64 lanes

struct FusedArgs {
    const float *g_sigma, *g_rgb, *sigmas, *rgbs;
    const uint32_t *masks;
    const char *packed;
    const float2 *planes;                   // [16][stride] float2 (the forward's encoder features)
    const float *hs, *cin, *h1, *h2;        // field_fwd_trace's layer inputs, [width][stride]
    float2 *d_planes;                       // out: gradient of the 32 encoder features, level-major
    float *slabs;                           // out: [workgroups][12][16][64] partial sums
};

// ---- a layer input from memory into a wave's scratch: ROWS8 x 8 rows of 32 points; lane 8 g + c takes points 4c..4c+3 of row 8 i + g
// TT: the trace's element type (float: 16 bytes per lane and row; _Float16: 8 bytes, widened when the rows are staged)
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
template <typename TT> struct RowVec { typedef float4 type; };
template <> struct RowVec<_Float16> { typedef f16x4 type; };
__device__ inline float4 widen(const float4 &v) { return v; }
__device__ inline float4 widen(const f16x4 &v) { return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]); }
template <int ROWS8, typename TT = float>
struct RowRegs {
    typename RowVec<TT>::type v[ROWS8];
};
template <int ROWS8, typename TT>
__device__ inline void rows_request(const float *__restrict__ base_f32, uint32_t stride, uint32_t tile, int lane, RowRegs<ROWS8, TT> &x) {
    typedef typename RowVec<TT>::type Vec;
#ifdef NSIG_FUSED_NOLOAD      // (diagnostic build, tools/_ab_fused.py: the kernel without its layer-input loads)
    for (int i = 0; i < ROWS8; ++i) x.v[i] = Vec{(TT)1.0f, (TT)2.0f, (TT)3.0f, (TT)4.0f};
    return;
#endif
    const TT *__restrict__ base = reinterpret_cast<const TT *>(base_f32);
    const uint32_t off = (uint32_t)(lane >> 3) * stride + 4u * (uint32_t)(lane & 7);
#pragma unroll
    for (int i = 0; i < ROWS8; ++i) x.v[i] = *reinterpret_cast<const Vec *>(at_uniform(base, (size_t)(8 * i) * stride + (size_t)tile * 32u, off * (uint32_t)sizeof(TT)));
}
// points at or beyond `live` (of this tile's 32) enter as zeros: stale rows of buffers sized for more points may hold anything, and 0 x NaN is NaN
template <int ROWS8, typename TT>
__device__ inline void rows_stage(float *__restrict__ X, int lane, uint32_t live, const RowRegs<ROWS8, TT> &x) {
    const uint32_t g = (uint32_t)lane >> 3, c4 = 4u * ((uint32_t)lane & 7u);
#pragma unroll
    for (int i = 0; i < ROWS8; ++i) {
        float4 v = widen(x.v[i]);
        if (live < 32u) {      // (uniform: only the last tile of all can be partial)
            v.x = c4 < live ? v.x : 0.0f; v.y = c4 + 1u < live ? v.y : 0.0f; v.z = c4 + 2u < live ? v.z : 0.0f; v.w = c4 + 3u < live ? v.w : 0.0f;
        }
        *reinterpret_cast<float4 *>(X + (8u * i + g) * kFusedRowFloats + c4) = v;
    }
}
// the encoder planes are float2 pairs (features 2l, 2l+1 of level l): lane 16 q + c2 takes points 2 c2, 2 c2 + 1 of level 4 i + q
struct PlaneRegs {
    float4 v[4];
};
__device__ inline void planes_request(const float2 *__restrict__ planes, uint32_t stride, uint32_t tile, int lane, PlaneRegs &x) {
    const uint32_t off = (uint32_t)(lane >> 4) * stride + 2u * (uint32_t)(lane & 15);
#pragma unroll
    for (int i = 0; i < 4; ++i) x.v[i] = *reinterpret_cast<const float4 *>(at_uniform(planes, (size_t)(4 * i) * stride + (size_t)tile * 32u, off * 8u));
}
__device__ inline void planes_stage(float *__restrict__ X, int lane, uint32_t live, const PlaneRegs &x) {
    const uint32_t q = (uint32_t)lane >> 4, c2 = 2u * ((uint32_t)lane & 15u);
    const bool in0 = c2 < live, in1 = c2 + 1u < live;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float4 v = x.v[i];
        const uint32_t l = 4u * i + q;
        *reinterpret_cast<float2 *>(X + (2u * l) * kFusedRowFloats + c2) = make_float2(in0 ? v.x : 0.0f, in1 ? v.z : 0.0f);
        *reinterpret_cast<float2 *>(X + (2u * l + 1u) * kFusedRowFloats + c2) = make_float2(in0 ? v.y : 0.0f, in1 ? v.w : 0.0f);
    }
}

// ---- transposition: an accumulator block (column = point p, register r = row row_of_reg16(h, r)) into rows row0.. of the scratch
__device__ inline void put_block(float *__restrict__ DY, uint32_t row0, const f32x16 &v, int p, int h) {
#pragma unroll
    for (int r = 0; r < 16; ++r) DY[(row0 + (uint32_t)row_of_reg16(h, r)) * kFusedRowFloats + p] = v[r];
}
// ... and back as K-step u of an MFMA operand: lane (r, h) holds row `row`, points 16 u + 8 h .. + 7.  Every lane reads (a row that does not exist is some
// other row of the scratch) and selects afterwards: no divergent branch inside the tile's body.
__device__ inline void fetch_operand(const float *__restrict__ rows, uint32_t row, bool exists, int h, int u, Split8 &o) {
    const float *at = rows + row * kFusedRowFloats + 16u * u + 8u * h;
    const float4 x = *reinterpret_cast<const float4 *>(at), y = *reinterpret_cast<const float4 *>(at + 4);
    const float v[8] = {exists ? x.x : 0.0f, exists ? x.y : 0.0f, exists ? x.z : 0.0f, exists ? x.w : 0.0f,
                        exists ? y.x : 0.0f, exists ? y.y : 0.0f, exists ? y.z : 0.0f, exists ? y.w : 0.0f};
    split8(v, o);
}
// acc[rb][cb] += A(rows 32 rb + r of DY) . B(rows 32 cb + r of X)^T over the tile's 32 points, one K-step (16 points) at a time
template <int RB, int CB>
__device__ inline void products(const float *__restrict__ DY, const float *__restrict__ X, uint32_t r, bool a_exists, int h, f32x16 *const (&acc)[RB][CB]) {
#ifdef NSIG_FUSED_NOPROD      // (diagnostic build: the kernel without its weight-gradient products)
    return;
#endif
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        Split8 A[RB], B[CB];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) fetch_operand(DY, 32u * rb + r, a_exists, h, u, A[rb]);
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) fetch_operand(X, 32u * cb + r, true, h, u, B[cb]);
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) *acc[rb][cb] = mac3(A[rb], B[cb], *acc[rb][cb]);
    }
}

// the compiler must keep a wave's LDS writes in front of the reads of OTHER lanes' values (it only knows this lane's addresses)
__device__ inline void scratch_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// zero the rows of a 64-row gradient whose forward activation was clamped (bit mask_bit<P>(i) of `bits` clear): the values field_bwd_trace stores
template <typename P>
__device__ inline void apply_mask(f32x16 (&v)[2], uint32_t bits) {
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r) v[rb][r] = masked(v[rb][r], bits, mask_bit<P>(rb * 16 + r));
}
template <typename P>
__device__ inline void to_operand(const f32x16 (&v)[2], typename P::Op (&b)[4]) {
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int r = 0; r < 16; r += 2) P::put2(b[2 * rb + (r >> 3)], (r & 7) >> 1, v[rb][r], v[rb][r + 1]);
}

typedef float f32x3u __attribute__((ext_vector_type(3), aligned(4)));
struct TileIn {      // what a tile's backward starts from
    uint32_t mask_s, mask_c0, mask_c1;
    f32x3u rgb, grgb;
    float gs, sig;
};

template <typename TT>
__global__ void __launch_bounds__(256) k_field_bwd_wgrad(FusedArgs a, uint32_t stride, uint32_t M, const uint32_t *__restrict__ rows_dev) {
    typedef Bf16x3 P;
    // one object, the weights FIRST: their fragments are read at immediate offsets from one lane address, and a DS immediate reaches 64 KiB (behind the 72 KiB
    // scratch every fragment would need an address register of its own)
    struct Lds {
        char w[P::kBwdLds];
        float scratch[4 * kFusedWaveFloats];
    };
    __shared__ __attribute__((aligned(16))) Lds lds_all;
    char *const wlds = lds_all.w;
    float *const scratch = lds_all.scratch;
    const uint32_t n = rows_dev != nullptr ? min(M, *rows_dev) : M;
    stage_weights(wlds, a.packed + P::kBwdOffset, (int)P::kBwdLds);
    constexpr size_t kHalf = kBwdBytes;
    const int lane = threadIdx.x & 63, p = lane & 31, h = lane >> 5;
    const uint32_t wid = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    float *const X = scratch + wid * kFusedWaveFloats, *const DY = X + kFusedRowsX * kFusedRowFloats;
    const uint32_t n_tiles = ceil_div(n, 32u), step = gridDim.x * 4u;
    const float e_lo = expf(-15.0f), e_hi = expf(15.0f);

    // block b = 4 * role + q of k_field_wgrad's slab: 0,1 W1s row blocks | 2,3 Wc1 row blocks | 4,5 W2s column blocks | 6,7 Wc3 column blocks | 8..11 Wc2 (row block, column block)
    f32x16 acc[kFusedBlocks];
#pragma unroll
    for (int b = 0; b < (int)kFusedBlocks; ++b)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[b][e] = 0.0f;

    auto request_in = [&](uint32_t tile, TileIn &in) {
        const uint32_t sl = min(tile * 32u + (uint32_t)p, n - 1u);
        const uint32_t *mrow = a.masks + (size_t)tile * 192 + lane;
        in.mask_s = mrow[0]; in.mask_c0 = mrow[64]; in.mask_c1 = mrow[128];
        in.rgb = *reinterpret_cast<const f32x3u *>(a.rgbs + 3 * (size_t)sl);
        in.grgb = *reinterpret_cast<const f32x3u *>(a.g_rgb + 3 * (size_t)sl);
        in.gs = a.g_sigma[sl];
        in.sig = a.sigmas[sl];
    };

    uint32_t tile = blockIdx.x * 4u + wid;
    // requests are unconditional (past the wave's last tile: the last tile of all again, never used): see k_field_wgrad on the compiler's wait counts
    TileIn in{};
    RowRegs<8, TT> x_h2{}, x_h1{};
    if (n_tiles != 0) {
        request_in(min(tile, n_tiles - 1u), in);
        rows_request<8, TT>(a.h2, stride, min(tile, n_tiles - 1u), lane, x_h2);
        rows_request<8, TT>(a.h1, stride, min(tile, n_tiles - 1u), lane, x_h1);
    }
    for (; tile < n_tiles; tile += step) {
        const uint32_t s = tile * 32u + (uint32_t)p;
        const bool live_pt = s < n;
        const uint32_t live = min(32u, n - tile * 32u);      // points of this tile that exist
        const uint32_t upcoming = min(tile + step, n_tiles - 1u);

        // ---- the tile's inputs: d(pre-sigmoid colour) -- only lane half 0, rows 0..2 -- and d log-density = g * exp(clamp(h0, -15, 15)) (activation.py:14)
        float dv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) dv[j] = 0.0f;
        float head0 = 0.0f;
        if (h == 0 && live_pt) {
            dv[0] = in.grgb.x * (in.rgb.x * (1.0f - in.rgb.x));
            dv[1] = in.grgb.y * (in.rgb.y * (1.0f - in.rgb.y));
            dv[2] = in.grgb.z * (in.rgb.z * (1.0f - in.rgb.z));
            head0 = in.gs * fminf(fmaxf(in.sig, e_lo), e_hi);
        }
        const uint32_t mask_s = in.mask_s, mask_c0 = in.mask_c0, mask_c1 = in.mask_c1;
        rows_stage<8, TT>(X, lane, live, x_h2);                  // h2: the input of the colour head
        RowRegs<4, TT> x_cin;
        rows_request<4, TT>(a.cin, stride, tile, lane, x_cin);   // (every layer input is requested TWO layers ahead: one layer's arithmetic is shorter than a trip to memory)

        // ---- colour head: dWc3 = d_out x h2^T (rows 0..2 of 16)
        if (h == 0) {
#pragma unroll
            for (int j = 0; j < 3; ++j) DY[(uint32_t)j * kFusedRowFloats + p] = dv[j];
        }
        typename P::Op dout[1];
#pragma unroll
        for (int j = 0; j < 8; j += 2) P::put2(dout[0], j >> 1, dv[j], dv[j + 1]);
        f32x16 hid[2];
        typename P::Op b4[4];
        mfma_layer<P, 2, 1>(wlds, kHalf, B0, lane, dout, hid);
        scratch_fence();
        products<1, 2>(DY, X, (uint32_t)p, p < 3, h, {{&acc[6], &acc[7]}});

        // ---- colour layer 2: dWc2 = d_h2 x h1^T
        apply_mask<P>(hid, mask_c1);
        to_operand<P>(hid, b4);
        scratch_fence();
        put_block(DY, 0u, hid[0], p, h);
        put_block(DY, 32u, hid[1], p, h);
        rows_stage<8, TT>(X, lane, live, x_h1);
        RowRegs<8, TT> x_hs;
        rows_request<8, TT>(a.hs, stride, tile, lane, x_hs);
        mfma_layer<P, 2, 4>(wlds, kHalf, B1, lane, b4, hid);
        scratch_fence();
        products<2, 2>(DY, X, (uint32_t)p, true, h, {{&acc[8], &acc[9]}, {&acc[10], &acc[11]}});

        // ---- colour layer 1: dWc1 = d_h1 x cin^T
        apply_mask<P>(hid, mask_c0);
        to_operand<P>(hid, b4);
        scratch_fence();
        put_block(DY, 0u, hid[0], p, h);
        put_block(DY, 32u, hid[1], p, h);
        rows_stage<4, TT>(X, lane, live, x_cin);
        PlaneRegs x_pl;
        planes_request(a.planes, stride, tile, lane, x_pl);
        f32x16 dso[1];
        mfma_layer<P, 1, 4>(wlds, kHalf, B2, lane, b4, dso);      // rows 1..15 = d geo_feat
        scratch_fence();
        products<2, 1>(DY, X, (uint32_t)p, true, h, {{&acc[2]}, {&acc[3]}});

        // ---- sigma head: dW2s = d_so x hs^T (16 rows)
        typename P::Op dhead[1];
        float head8[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) head8[r] = dso[0][r];
        if (h == 0) head8[0] = head0;      // row 0: d log-density
#pragma unroll
        for (int r = 0; r < 8; r += 2) P::put2(dhead[0], r >> 1, head8[r], head8[r + 1]);
        scratch_fence();
#pragma unroll
        for (int r = 0; r < 8; ++r) DY[(uint32_t)row_of_reg(h, r) * kFusedRowFloats + p] = head8[r];
        rows_stage<8, TT>(X, lane, live, x_hs);
        rows_request<8, TT>(a.h2, stride, upcoming, lane, x_h2);      // the next tile's first layer input and inputs
        request_in(upcoming, in);
        mfma_layer<P, 2, 1>(wlds, kHalf, B3, lane, dhead, hid);
        scratch_fence();
        products<1, 2>(DY, X, (uint32_t)p, p < 16, h, {{&acc[4], &acc[5]}});

        // ---- sigma layer 1: dW1s = d_hs x feat^T; d feature = W1s^T d_hs
        apply_mask<P>(hid, mask_s);
        to_operand<P>(hid, b4);
        scratch_fence();
        put_block(DY, 0u, hid[0], p, h);
        put_block(DY, 32u, hid[1], p, h);
        planes_stage(X, lane, live, x_pl);
        rows_request<8, TT>(a.h1, stride, upcoming, lane, x_h1);
        f32x16 dall[1];
        mfma_layer<P, 1, 4>(wlds, kHalf, B4F, lane, b4, dall);      // row f = d feature[f]; registers (r, r+1), r even, hold one level's pair
        scratch_fence();
        products<2, 1>(DY, X, (uint32_t)p, true, h, {{&acc[0]}, {&acc[1]}});
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            float2 v;
            v.x = dall[0][r]; v.y = dall[0][r + 1];
            // level = (row_of_reg16(h, r) >> 1) = (row_of_reg16(0, r) >> 1) + 2 h: uniform part | lane part
            *at_uniform(a.d_planes, (size_t)(row_of_reg16(0, r) >> 1) * stride + (size_t)tile * 32u, (2u * (uint32_t)h * stride + (uint32_t)p) * 8u) = v;
        }
        scratch_fence();      // the next tile's staging overwrites what this tile's last operands were read from
    }

    // ---- the workgroup's four waves add their blocks in wave order (four blocks per round through the scratch) and store one slab
    __syncthreads();
#pragma unroll
    for (int round = 0; round < 3; ++round) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 16; ++e) scratch[(wid * 4u + q) * 1024u + e * 64 + lane] = acc[round * 4 + q][e];
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < 4096u; i += 256u)
            a.slabs[(size_t)blockIdx.x * kFusedSlab + round * 4096u + i] = ((scratch[i] + scratch[4096u + i]) + scratch[8192u + i]) + scratch[12288u + i];
        __syncthreads();
    }
}

}  // namespace nsig

using namespace nsig;

static uint32_t fused_workgroups(uint32_t M) {      // four tiles of 32 points per workgroup where there is work; at most one workgroup per compute unit
    const uint32_t want = ceil_div(ceil_div(M, 32u), 4u);
    return want < 1u ? 1u : (want > kFusedMaxWGs ? kFusedMaxWGs : want);
}

NSIG_EXPORT size_t field_bwd_wgrad_scratch_bytes(uint32_t M) { return (size_t)fused_workgroups(M) * kFusedSlab * sizeof(float); }

static int bwd_wgrad_impl(const char *who, bool trace_f16, uint32_t M, const uint32_t *rows_dev, const float *grad_sigmas, const float *grad_rgbs, const float *sigmas,
                          const float *rgbs, const uint32_t *masks, const void *packed, const void *planes, const void *act_hs, const void *act_cin, const void *act_h1,
                          const void *act_h2, void *d_planes, void *scratch, float *grad_sigma_params, float *grad_color_params, nsig_stream_t stream) {
    NSIG_REQUIRE(grad_sigmas && grad_rgbs && sigmas && rgbs && masks && packed && planes && act_hs && act_cin && act_h1 && act_h2 && d_planes && scratch &&
                 grad_sigma_params && grad_color_params, "%s: null pointer", who);
    NSIG_REQUIRE(M >= 1 && M <= (1u << 26), "%s: M=%u out of range (1 .. 2^26, as field_fwd_trace: the lane part of an address is a 32-bit byte offset of up to 32 x stride)", who, M);
    const void *all[] = {packed, planes, act_hs, act_cin, act_h1, act_h2, d_planes, scratch};
    for (const void *q : all) NSIG_REQUIRE((reinterpret_cast<uintptr_t>(q) & 15) == 0, "%s: packed, planes, the layer inputs, d_planes and scratch must be 16-byte aligned", who);
    const uint32_t stride = ceil_div(M, 32u) * 32u, n_wg = fused_workgroups(M);
    FusedArgs a{grad_sigmas, grad_rgbs, sigmas, rgbs, masks, reinterpret_cast<const char *>(packed), reinterpret_cast<const float2 *>(planes),
                reinterpret_cast<const float *>(act_hs), reinterpret_cast<const float *>(act_cin), reinterpret_cast<const float *>(act_h1), reinterpret_cast<const float *>(act_h2),
                reinterpret_cast<float2 *>(d_planes), reinterpret_cast<float *>(scratch)};
    hipStream_t st = as_stream(stream);
    if (trace_f16) k_field_bwd_wgrad<_Float16><<<n_wg, 256, 0, st>>>(a, stride, M, rows_dev);
    else k_field_bwd_wgrad<float><<<n_wg, 256, 0, st>>>(a, stride, M, rows_dev);
    if (int e = check_launch(who)) return e;
    return wgrad_reduce_launch(reinterpret_cast<const float *>(scratch), n_wg, grad_sigma_params, grad_color_params, st, "field_bwd_wgrad (reduce)");
}

NSIG_EXPORT int field_bwd_wgrad(uint32_t M, const uint32_t *rows_dev, const float *grad_sigmas, const float *grad_rgbs, const float *sigmas, const float *rgbs,
                                const uint32_t *masks, const void *packed, const void *planes, const float *act_hs, const float *act_cin, const float *act_h1,
                                const float *act_h2, void *d_planes, void *scratch, float *grad_sigma_params, float *grad_color_params, nsig_stream_t stream) {
    return bwd_wgrad_impl("field_bwd_wgrad", false, M, rows_dev, grad_sigmas, grad_rgbs, sigmas, rgbs, masks, packed, planes, act_hs, act_cin, act_h1, act_h2, d_planes, scratch,
                          grad_sigma_params, grad_color_params, stream);
}

NSIG_EXPORT int field_bwd_wgrad_f16(uint32_t M, const uint32_t *rows_dev, const float *grad_sigmas, const float *grad_rgbs, const float *sigmas, const float *rgbs,
                                    const uint32_t *masks, const void *packed, const void *planes, const void *act_hs, const void *act_cin, const void *act_h1,
                                    const void *act_h2, void *d_planes, void *scratch, float *grad_sigma_params, float *grad_color_params, nsig_stream_t stream) {
    return bwd_wgrad_impl("field_bwd_wgrad_f16", true, M, rows_dev, grad_sigmas, grad_rgbs, sigmas, rgbs, masks, packed, planes, act_hs, act_cin, act_h1, act_h2, d_planes, scratch,
                          grad_sigma_params, grad_color_params, stream);
}
