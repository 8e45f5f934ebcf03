// Fused field network for gfx950: hash encode (16 base levels + summed codebook table) -> sigma MLP ->
// trunc_exp | SH(deg 4) + geo features -> color MLP -> sigmoid, forward and input-gradient backward.
//
// Behavioural reference: /root/reference/nerf/network_wtmk_tcnn.py:97-176 (NeRFNetwork.forward / density /
// color), with the MLP semantics of tiny-cuda-nn's FullyFusedMLP restated in oracle/field_ref.py (the
// reference delegates them to tinycudann, which is not part of its tree: parity unpinned).
//
// Mapping onto CDNA4
//   * one wave = 32 points.  The point index sits on the MFMA column (lane & 31); the two lane halves
//     split the K dimension, so lane (p, h) gathers exactly the 8 hash levels whose 16 features are its
//     share of the B operand of v_mfma_f32_32x32x16_bf16 -- the encoder output never leaves registers
//     and needs no cross-lane movement before the first layer.
//   * layers are evaluated transposed (H^T = W . X^T) so that the 32x32 accumulator of one layer (column
//     = point, rows in registers) is already the B operand of the next one; the K order this implies is
//     baked into the packed weights (mlp_pack_weights), the activations never touch LDS or HBM.
//   * weights are frozen and tiny: they are staged once per workgroup into LDS as ready-made A fragments
//     (16 B per lane, conflict-free ds_read_b128).
//   * precision, two selectable variants of every MFMA kernel (mlp_set_precision / NERFSIG_MLP):
//       Bf16x3: every product is evaluated as split-bf16 (hi*hi + hi*lo + lo*hi, fp32 accumulate), ~2^-16 relative error per
//               product, 3 x 24 MFMAs per 32 points forward and ~6 VALU operations per operand pair for the split;
//       F16   : operands rounded once to fp16 (2^-11), ONE v_mfma_f32_32x32x16_f16 per product with fp32 accumulate (the
//               reference's tcnn FullyFusedMLP computes in fp16 WITH fp16 accumulate, network_wtmk_tcnn.py:52-88): 24 MFMAs
//               forward / 20 backward per 32 points, one v_cvt_pk per operand pair, half the LDS.  The backward normalises each
//               point's upstream gradient by a power of two first (exact), so loss-scaled gradients (GradScaler's 65536x) can
//               neither overflow nor underflow fp16.
//   * backward needs only the input gradient of the codebook channels (all network weights are frozen,
//     network_wtmk_tcnn.py:90-95), so the forward saves just the ReLU sign bits (6 words per point).
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <unordered_set>
#include "fieldmlp.h"

#include <stdlib.h>

namespace nsig {

// (packed weight layout, precision tags and the wave-level building blocks: fieldmlp.h)

__device__ inline float fwd_weight(int frag, int lane, int j, const float *__restrict__ sp, const float *__restrict__ cp) {
    const int r = lane & 31, h = lane >> 5;
    if (frag < F1) {
        const int rb = (frag - F0) >> 1, ks = (frag - F0) & 1;
        return sp[kSigmaW1 + (32 * rb + r) * 32 + 16 * ks + 8 * h + j];
    }
    if (frag < F2) return r < 16 ? sp[kSigmaW2 + r * 64 + k_from_acc(frag - F1, h, j)] : 0.0f;
    if (frag < F3) {
        const int rb = (frag - F2) >> 1, ks = (frag - F2) & 1;
        int cin;
        if (ks == 0) cin = 8 * h + j;  // SH component
        else { const int rho = row_of_reg(h, j); cin = rho == 0 ? 31 : 15 + rho; }  // geo feature rho-1, or the padded 1.0 input
        return cp[kColorW1 + (32 * rb + r) * 32 + cin];
    }
    if (frag < F4) {
        const int rb = (frag - F3) >> 2, ks = (frag - F3) & 3;
        return cp[kColorW2 + (32 * rb + r) * 64 + k_from_acc(ks, h, j)];
    }
    return r < 16 ? cp[kColorW3 + r * 64 + k_from_acc(frag - F4, h, j)] : 0.0f;
}

__device__ inline float bwd_weight(int frag, int lane, int j, const float *__restrict__ sp, const float *__restrict__ cp) {
    const int r = lane & 31, h = lane >> 5;
    if (frag < B1) return cp[kColorW3 + (8 * h + j) * 64 + 32 * (frag - B0) + r];
    if (frag < B2) {
        const int rb = (frag - B1) >> 2, ks = (frag - B1) & 3;
        return cp[kColorW2 + k_from_acc(ks, h, j) * 64 + 32 * rb + r];
    }
    if (frag < B3) return (r >= 1 && r < 16) ? cp[kColorW1 + k_from_acc(frag - B2, h, j) * 32 + 15 + r] : 0.0f;
    if (frag < B4) return sp[kSigmaW2 + row_of_reg(h, j) * 64 + 32 * (frag - B3) + r];
    if (frag < B4F) return r < 2 ? sp[kSigmaW1 + k_from_acc(frag - B4, h, j) * 32 + 30 + r] : 0.0f;
    return sp[kSigmaW1 + k_from_acc(frag - B4F, h, j) * 32 + r];
}

__global__ void __launch_bounds__(256) k_pack_weights(const float *__restrict__ sp, const float *__restrict__ cp, __bf16 *__restrict__ packed) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;  // (frag, lane, j)
    const int total = (kFwdFrags + kBwdFrags) * 512;
    if (e >= total) return;
    const int frag = e >> 9, lane = (e >> 3) & 63, j = e & 7;
    const bool fwd = frag < kFwdFrags;
    const float w = fwd ? fwd_weight(frag, lane, j, sp, cp) : bwd_weight(frag - kFwdFrags, lane, j, sp, cp);
    const __bf16 hi = (__bf16)w;
    const __bf16 lo = (__bf16)(w - (float)hi);
    __bf16 *base_hi = packed + (fwd ? 0 : 2 * kFwdBytes / 2);
    const int f = fwd ? frag : frag - kFwdFrags;
    const size_t lo_off = (fwd ? kFwdBytes : kBwdBytes) / 2;
    base_hi[(size_t)f * 512 + lane * 8 + j] = hi;
    base_hi[lo_off + (size_t)f * 512 + lane * 8 + j] = lo;
    _Float16 *half_base = reinterpret_cast<_Float16 *>(reinterpret_cast<char *>(packed) + kPackedBf16Bytes) + (fwd ? 0 : kFwdBytes / 2);
    half_base[(size_t)f * 512 + lane * 8 + j] = (_Float16)w;
}

// ----------------------------------------------------------------------------- forward

// XCD-partitioned, level-major encoder.  The expensive part of the encoder is the 5-6 finest levels: consecutive
// points of a ray fall into different cells there, so every point costs ~4 distinct cache lines per level, served from
// beyond L2 when all 17 tables (68 MiB) compete for each XCD's 4 MiB L2.  Workgroups are dealt round-robin over the 8
// XCDs (blockIdx % 8 labels the XCD group; a speed assumption only), so workgroup (tile, slot = blockIdx % 8) encodes
// for its tile of 256 points only the levels assigned to that slot: each XCD then gathers from ONE fine table (4 MiB,
// L2-sized) plus a few coarse ones.  Features are written level-major, planes[level][point] (float2), so the stores of
// a wave are 512 contiguous bytes; streaming loads/stores are non-temporal to leave L2 to the tables.
struct SlotTable {
    uint8_t n[8];
    uint8_t level[8][8];      // 0..15 base levels, 16 = pre-summed codebook; every level belongs to exactly one slot
};

// Lane pairs cooperate on the gathers.  A gather costs ~2.4 clk per distinct 128-byte line per instruction plus ~1 clk per lane
// (tools/micro/gather_rate.hip), and the two x-neighbours of a (dy, dz) pair share a line in 15 of 16 cells.  With one lane per
// point they sit in two different instructions (64 lines each: one misses L1, the next hits it); here lanes 2i and 2i+1 first
// fetch the x = 0 / x = 1 sides of point 2i, then of point 2i+1, so every instruction touches 32 lines instead of 64 and none
// relies on L1 to keep a line until its partner instruction arrives.  Same number of gather instructions, same values; the
// cell hashes travel from the owner to its neighbour by DPP, the fetched corners back the same way.
__device__ inline uint32_t dpp_u(uint32_t v, int ctrl) {
    switch (ctrl) {   // quad_perm selectors must be immediates
        case 0: return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xA0, 0xF, 0xF, true);    // [0,0,2,2]: the even lane's value
        case 1: return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xF5, 0xF, 0xF, true);    // [1,1,3,3]: the odd lane's value
        default: return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true);   // [1,0,3,2]: swap with the neighbour
    }
}

// One level of one 256-point tile: lane pairs fetch the two x sides (see above), trilinear interpolation, one streaming store.
// Two layouts of a plane set (17 feature planes of `stride` points):
//   fp32   planes[level][point] float2 -- every consumer reads it;
//   mixed  levels 0..14 as fp16 pairs (4 bytes per point), then level 15 and the codebook level as float2 -- written by hg_encode_planes_mixed for the fp16
//          MLP, whose first-layer operand is exactly those fp16 pairs (rounded to nearest even here instead of at the MLP's load: the same bits).  Level 15
//          stays fp32 because the codebook is added to it BEFORE the rounding (network_wtmk_tcnn.py:106).  15 x 4 + 2 x 8 = 76 instead of 136 bytes per point
//          each way between the encoder and the MLP.
constexpr int kHalfLevels = NSIG_BASE_LEVELS - 1;
__device__ __host__ inline float2 *mixed_f32_plane(void *planes, uint32_t stride, int level) {      // level 15 or 16 of a mixed set
    return reinterpret_cast<float2 *>(reinterpret_cast<uint32_t *>(planes) + (size_t)kHalfLevels * stride) + (size_t)(level - kHalfLevels) * stride;
}
__device__ inline float2 load_plane(const float2 *__restrict__ planes, uint32_t stride, int level, uint32_t s, bool mixed) {
    if (!mixed) return planes[(size_t)level * stride + s];
    if (level >= kHalfLevels) return mixed_f32_plane(const_cast<float2 *>(planes), stride, level)[s];
    const f32x2 w = __builtin_convertvector(__builtin_bit_cast(f16x2, reinterpret_cast<const uint32_t *>(planes)[(size_t)level * stride + s]), f32x2);
    return make_float2(w[0], w[1]);
}

__device__ inline void encode_tile_level(const float2 *__restrict__ table, float cell, uint32_t xs, float x, float y, float z,
                                         float2 *__restrict__ out, bool half_out = false, bool streaming = true) {
    uint32_t ix, iy, iz;
    float wx, wy, wz;
    axis_cell(x, cell, ix, wx);
    axis_cell(y, cell, iy, wy);
    axis_cell(z, cell, iz, wz);
    const uint32_t hy0 = iy * kPrimeY, hy1 = (iy + 1u) * kPrimeY, hz0 = iz * kPrimeZ, hz1 = (iz + 1u) * kPrimeZ;   // corner_rows()
    float2 v[2][4];   // v[P][q]: this lane's x side of point P of the pair, corner (dy, dz) = (q >> 1, q & 1)
#pragma unroll
    for (int P = 0; P < 2; ++P) {
        const uint32_t hx = dpp_u(ix, P) + xs;
        const uint32_t a0 = dpp_u(hy0, P), a1 = dpp_u(hy1, P), b0 = dpp_u(hz0, P), b1 = dpp_u(hz1, P);
        v[P][0] = table[(hx ^ a0 ^ b0) & kRowMask];
        v[P][1] = table[(hx ^ a0 ^ b1) & kRowMask];
        v[P][2] = table[(hx ^ a1 ^ b0) & kRowMask];
        v[P][3] = table[(hx ^ a1 ^ b1) & kRowMask];
    }
    float2 e[8];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float2 mine = xs ? v[1][q] : v[0][q];      // my own point, my x side
        const float2 give = xs ? v[0][q] : v[1][q];      // the neighbour's point, my x side
        float2 got;                                       // my own point, the other x side (fetched by the neighbour)
        got.x = __uint_as_float(dpp_u(__float_as_uint(give.x), 2));
        got.y = __uint_as_float(dpp_u(__float_as_uint(give.y), 2));
        e[q] = xs ? got : mine;          // corner k = 4*dx + q
        e[4 + q] = xs ? mine : got;
    }
    const float2 val = trilerp(e, wx, wy, wz);
    if (half_out) {      // (a level of a mixed plane set: `out` addresses 4-byte elements)
        __builtin_nontemporal_store(cvt_pk_f16(val.x, val.y), reinterpret_cast<uint32_t *>(out));
        return;
    }
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    const f32x2_t vv = {val.x, val.y};
    // the all-fp32 plane set (stage 1, the split-bf16 MLP) goes through the caches: its consumer runs right behind this launch and found the planes in memory when
    // they were streamed out (stage-1 forward 166 -> 133 us, this launch +2.5, same box, two rounds); the mixed set of the headline step keeps its streaming
    // stores (LABNOTES section 12: fewer store transactions on the fabric the line fills compete for)
    if (streaming) __builtin_nontemporal_store(vv, reinterpret_cast<f32x2_t *>(out));
    else *reinterpret_cast<f32x2_t *>(out) = vv;
}

// Reads the tables of a slot through that slot's XCD (blockIdx % 8): its L2 then holds the slot's fine table when the encoder starts.
__global__ void __launch_bounds__(256) k_warm_tables(TablePtrs base, const float *__restrict__ S, SlotTable tab, float *__restrict__ sink) {
    const uint32_t slot = blockIdx.x & 7u, wg = blockIdx.x >> 3, n_wg = gridDim.x >> 3;
    float acc = 0.0f;
    for (int i = 0; i < tab.n[slot]; ++i) {
        const int l = tab.level[slot][i];
        const float4 *t = reinterpret_cast<const float4 *>(l == NSIG_BASE_LEVELS ? S : base.p[l]);
        if (t == nullptr) continue;
        for (uint32_t e = wg * 256 + threadIdx.x; e < NSIG_TABLE_ROWS / 2; e += n_wg * 256) {
            const float4 v = t[e];
            acc += v.x + v.y + v.z + v.w;
        }
    }
    if (acc == 1.2345e-33f) *sink = acc;      // (never: keeps the loads)
}

// Workgroup (tile, slot) encodes, for its 256 points, the levels (or the part of a level's tile range) assigned to its XCD slot.
template <bool mixed>      // (the plane layout as a compile-time constant: the destination of a level's features is then one address computation, not a chain of branches per level)
__global__ void __launch_bounds__(256) k_encode_planes(const float *__restrict__ xyzs, uint32_t M, float bound, TablePtrs base, LevelGeom geom,
                                                       const float *__restrict__ S, float2 *__restrict__ planes, uint32_t stride, SlotTable tab,
                                                       const uint32_t *__restrict__ rows_dev = nullptr) {
    // rows_dev (the eval loop's bursts, hg_encode_planes_rows): the number of rows that exist is known on the device only; the launch is sized for `M`
    // (the buffers' capacity, which also fixes the plane stride) and rows beyond the count are skipped
    uint32_t lim = stride;
    if (rows_dev != nullptr) {
        const uint32_t r = *rows_dev;
        if (r == 0) return;
        M = min(M, r);
        lim = min(stride, ceil_div(M, 32u) * 32u);
    }
    const uint32_t slot = blockIdx.x & 7u;
    const uint32_t n_tiles = ceil_div(stride, 256u);
    const int n_levels = tab.n[slot];
    const float two_b = 2.0f * bound;
    const uint32_t xs = threadIdx.x & 1u;   // which x side of the cell this lane fetches
    for (uint32_t tile = blockIdx.x >> 3; tile < n_tiles; tile += gridDim.x >> 3) {
        const uint32_t m = tile * 256 + threadIdx.x;
        if (m >= lim) continue;             // a multiple of 32: lane pairs (and DPP quads) are in or out together
        const uint32_t ml = min(m, M - 1);  // rows in [M, stride) replicate the last point (never consumed)
        // one 12-byte load and one 8-byte streaming store per level: the kernel is bound by the address path, every instruction counts
        const float3 pt = *reinterpret_cast<const float3 *>(xyzs + 3 * (size_t)ml);
        const float x = (pt.x + bound) / two_b, y = (pt.y + bound) / two_b, z = (pt.z + bound) / two_b;
        for (int i = 0; i < n_levels; ++i) {
            const int l = tab.level[slot][i];
            const bool half_out = mixed && l < kHalfLevels;
            float2 *out = !mixed ? planes + (size_t)l * stride + m
                                 : (half_out ? reinterpret_cast<float2 *>(reinterpret_cast<uint32_t *>(planes) + (size_t)l * stride + m) : mixed_f32_plane(planes, stride, l) + m);
            encode_tile_level(reinterpret_cast<const float2 *>(l == NSIG_BASE_LEVELS ? S : base.p[l]), geom.cell[l], xs, x, y, z, out, half_out, /*streaming=*/mixed);
        }
    }
}

// The codebook level alone, into plane 16 of a plane set whose base planes are already there (hg_encode_codebook_plane: rays that
// do not change between steps).  One tile per workgroup, all XCDs: S (4 MiB) is the only table, every L2 holds its own copy.
// reset: the header of a scatter plan kept across steps -- its largest-gradient word starts every step at zero (k_plan_dest does
// that for a plan computed per step).
__global__ void __launch_bounds__(256) k_encode_codebook_plane(const float *__restrict__ xyzs, uint32_t M, float bound, float cell, const float *__restrict__ S,
                                                               float2 *__restrict__ plane, uint32_t stride, BinHeader *__restrict__ reset) {
    if (reset != nullptr && blockIdx.x == 0 && threadIdx.x == 0) reset->gmax_bits = 0;
    const uint32_t m = blockIdx.x * 256 + threadIdx.x;
    if (m >= stride) return;                // stride is a multiple of 32: lane pairs (and DPP quads) are in or out together
    const uint32_t ml = min(m, M - 1);
    const float two_b = 2.0f * bound;
    const float3 pt = *reinterpret_cast<const float3 *>(xyzs + 3 * (size_t)ml);
    const float x = (pt.x + bound) / two_b, y = (pt.y + bound) / two_b, z = (pt.z + bound) / two_b;
    encode_tile_level(reinterpret_cast<const float2 *>(S), cell, threadIdx.x & 1u, x, y, z, plane + m);
}

// (k_field_fwd's kPlanes = 0: gather the features in-kernel (fused); 1: read all 17 from the level-major planes.)
// The training render's forward launch (all 17 feature planes in memory, sigma + rgb + ReLU masks out), software-pipelined over a wave's tiles.
// The plain loop -- load, evaluate, store, next tile -- spends most of a tile's ~4.9 us waiting, three times: for the planes at its head, for the
// view directions in front of the colour branch (a load issued there drains, in order, everything requested before it), and at the head of the
// next tile for the acknowledgement of its own stores (on gfx9 stores count in vmcnt like loads) -- with three waves per SIMD the arithmetic
// (~0.9 us of a tile) cannot cover that.  Here every wait is for something issued a whole evaluation earlier:
//     head of tile i:  wait for planes + directions of tile i (requested at the head of tile i-1)
//                      -> issue the STORES of tile i-1's results (held in 7 registers) -> request planes + directions of tile i+1 -> evaluate tile i.
template <typename P, bool kMixed>
__device__ inline void field_fwd_pipelined(const char *lds, int lane, const float *__restrict__ dirs, uint32_t M, bool add_codebook,
                                           const float2 *__restrict__ planes, uint32_t stride, float *__restrict__ sigmas, float *__restrict__ rgbs,
                                           uint32_t *__restrict__ masks) {
    static_assert(!kMixed || P::kMfmaPerProduct == 1, "mixed plane sets carry fp16 operands: the fp16 MLP only");
    constexpr size_t kHalf = kFwdBytes;
    const int p = lane & 31, h = lane >> 5;
    const uint32_t n_tiles = ceil_div(M, 32u);
    const uint32_t first = blockIdx.x * 4 + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), step = gridDim.x * 4;
    // fp32 set: nf[q] = level 8 (q >> 2) + (q & 3) + 4 h.  mixed set: nh[q] holds the level's fp16 pair -- already the operand word -- except for
    // level 15 (q = 7 of lane half 1), which arrives as float2 in nf[7] and takes the codebook before it is rounded.
    float2 nf[8] = {}, nc = make_float2(0.0f, 0.0f);
    uint32_t nh[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
    typedef float f32x3u __attribute__((ext_vector_type(3), aligned(4)));
    f32x3u nd = {0.0f, 0.0f, 0.0f};      // one 12-byte value across the loop: as three scalars the loaded triple is copied into their registers right behind the load -- a wait
    auto request = [&](uint32_t tile) {
        const uint32_t s = tile * 32 + p, sl = min(s, M - 1);
        if constexpr (kMixed) {
            const uint32_t *hp = reinterpret_cast<const uint32_t *>(planes);
#pragma unroll
            // 128 contiguous bytes per half-wave; streaming loads (the planes are read exactly once: step -0.4 % over three same-box rounds, at the edge of resolution)
            for (int q = 0; q < 7; ++q) nh[q] = __builtin_nontemporal_load(hp + (size_t)(8 * (q >> 2) + (q & 3) + 4 * h) * stride + s);
            if (h) nf[7] = mixed_f32_plane(const_cast<float2 *>(planes), stride, NSIG_BASE_LEVELS - 1)[s];
            else nh[7] = hp[(size_t)11 * stride + s];
            if (add_codebook && h) nc = mixed_f32_plane(const_cast<float2 *>(planes), stride, NSIG_BASE_LEVELS)[s];
        } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) nf[q] = planes[(size_t)(8 * (q >> 2) + (q & 3) + 4 * h) * stride + s];  // 256 contiguous bytes per half-wave
            if (add_codebook && h) nc = planes[(size_t)NSIG_BASE_LEVELS * stride + s];
        }
        nd = *reinterpret_cast<const f32x3u *>(dirs + 3 * (size_t)sl);
    };
    if (first >= n_tiles) return;
    request(first);
    uint32_t ptile = 0;
    bool have = false;
    float psigma = 0.0f, prgb[3] = {0.0f, 0.0f, 0.0f};
    uint32_t pmask[3] = {0u, 0u, 0u};
    auto store_prev = [&]() {
        const uint32_t s = ptile * 32 + p;
        if (s < M && h == 0) {
            sigmas[s] = psigma;
            rgbs[3 * (size_t)s] = prgb[0]; rgbs[3 * (size_t)s + 1] = prgb[1]; rgbs[3 * (size_t)s + 2] = prgb[2];
        }
        if (masks != nullptr) {      // (a render without gradients keeps no ReLU masks)
            uint32_t *mrow = masks + (size_t)ptile * 192 + lane;
            mrow[0] = pmask[0]; mrow[64] = pmask[1]; mrow[128] = pmask[2];
        }
    };
    for (uint32_t tile = first; tile < n_tiles; tile += step) {
        typename P::Op feat[2];
        if (add_codebook && h) {  // codebook added into channels 30:32 (network_wtmk_tcnn.py:106)
            nf[7].x = nf[7].x + nc.x;
            nf[7].y = nf[7].y + nc.y;
        }
        if constexpr (kMixed) {
#pragma unroll
            for (int q = 0; q < 7; ++q) feat[q >> 2].v[q & 3] = nh[q];
            const uint32_t top = cvt_pk_f16(nf[7].x, nf[7].y);
            feat[1].v[3] = h ? top : nh[7];
        } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) P::put2(feat[q >> 2], q & 3, nf[q].x, nf[q].y);
        }
        float dx = nd.x, dy = nd.y, dz = nd.z;
        // A compiler barrier that consumes this tile's inputs: the wait for them (vmcnt counts in order) is placed HERE, where only they and
        // long-acknowledged stores are outstanding -- left free, the compiler issues the next tile's requests first and then has to drain them too.
        if constexpr (P::kMfmaPerProduct == 1)
            asm volatile("" : "+v"(feat[0].v[0]), "+v"(feat[0].v[1]), "+v"(feat[0].v[2]), "+v"(feat[0].v[3]), "+v"(feat[1].v[0]), "+v"(feat[1].v[1]),
                         "+v"(feat[1].v[2]), "+v"(feat[1].v[3]), "+v"(dx), "+v"(dy), "+v"(dz) :: "memory");
        else
            asm volatile("" : "+v"(dx), "+v"(dy), "+v"(dz) :: "memory");
        if (have) store_prev();
        if (tile + step < n_tiles) request(tile + step);
        asm volatile("" ::: "memory");

        f32x16 hid[2];
        typename P::Op b4[4];
        f32x16 so[1];
        if constexpr (P::kMfmaPerProduct == 1) {
            // fp16: every layer's weight fragments are fetched from LDS as ONE burst, issued in front of the vector work that precedes the layer
            // (the ReLU / packing of the layer before: ~260 cycles) -- fetched one by one, each right in front of its MFMA, a fragment's LDS
            // latency (~100 cycles) was paid 24 times per tile: half of a wave's cycles were spent parked on lgkmcnt (SQ_WAIT_ANY).
            f16x8 a0[4], a1[4];
            load_frags(lds, F0, lane, a0);
            load_frags(lds, F1, lane, a1);
            __builtin_amdgcn_sched_barrier(0);
            mfma_regs<2, 2>(a0, feat, hid);
            pmask[0] = relu_to_operand<P>(hid, b4);
            mfma_regs<1, 4>(a1, b4, so);
        } else {
            mfma_layer<P, 2, 2>(lds, kHalf, F0, lane, feat, hid);
            pmask[0] = relu_to_operand<P>(hid, b4);
            mfma_layer<P, 1, 4>(lds, kHalf, F1, lane, b4, so);
        }
        psigma = expf(so[0][0]);  // trunc_exp forward (activation.py:9); row 0 of the sigma head lives in lane half 0
        float geo8[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) geo8[r] = so[0][r];
        if (h == 0) geo8[0] = 1.0f;  // the slot of row 0 carries the padded constant input (weight column 31)
        if constexpr (P::kMfmaPerProduct == 1) {     // color_branch() with the fragment bursts in front of the vector work
            f16x8 a2[4], a3[8], a4[4];
            load_frags(lds, F2, lane, a2);
            load_frags(lds, F3, lane, a3);
            __builtin_amdgcn_sched_barrier(0);
            const float ux = (dx + 1.0f) / 2.0f, uy = (dy + 1.0f) / 2.0f, uz = (dz + 1.0f) / 2.0f;   // (network_wtmk_tcnn.py:114-115)
            float sh[16];
            sh16(ux * 2.0f - 1.0f, uy * 2.0f - 1.0f, uz * 2.0f - 1.0f, sh);
            typename P::Op cin[2];
            const uint32_t hm = 0u - (uint32_t)h;     // a bit select (v_bfi): written as `h ? sh[8 + j] : sh[j]` the compiler indexes a scratch copy of sh[]
            auto pick = [&](int j) { return __uint_as_float((__float_as_uint(sh[j]) & ~hm) | (__float_as_uint(sh[8 + j]) & hm)); };
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
                P::put2(cin[0], j >> 1, pick(j), pick(j + 1));
                P::put2(cin[1], j >> 1, geo8[j], geo8[j + 1]);
            }
            mfma_regs<2, 2>(a2, cin, hid);
            load_frags(lds, F4, lane, a4);
            __builtin_amdgcn_sched_barrier(0);
            pmask[1] = relu_to_operand<P>(hid, b4);
            mfma_regs<2, 4>(a3, b4, hid);
            pmask[2] = relu_to_operand<P>(hid, b4);
            mfma_regs<1, 4>(a4, b4, so);
#pragma unroll
            for (int c = 0; c < 3; ++c) prgb[c] = 1.0f / (1.0f + expf(-so[0][c]));  // rows 0..2 live in lane half 0
        } else {
            uint32_t mask_c[2] = {0u, 0u};
            color_branch<P>(lds, lane, h, dx, dy, dz, geo8, mask_c, prgb);
            pmask[1] = mask_c[0];
            pmask[2] = mask_c[1];
        }
        ptile = tile;
        have = true;
    }
    store_prev();
}

// 164 registers: three waves per SIMD would fit, but the launch uses TWO workgroups per CU (field_grid(.., 2)): the kernel takes 57 us with 512, 768 or
// 1024 workgroups (71 us for the plain loop it replaces), and the step is shortest with 512 -- the content render's kernels run beside this launch
// (same box, three rounds: 1.051-1.063 ms against 1.064-1.070 with 768 and 1.062-1.074 with 1024).
template <typename P, bool kMixed = false>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3)))
k_field_fwd_train(const float *__restrict__ dirs, uint32_t M, bool add_codebook, const float2 *__restrict__ planes, uint32_t stride,
                  const char *__restrict__ packed, float *__restrict__ sigmas, float *__restrict__ rgbs, uint32_t *__restrict__ masks,
                  const uint32_t *__restrict__ rows_dev = nullptr) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    if (rows_dev != nullptr) {      // (field_fwd_rows: the row count lives on the device; M is the capacity the launch and the plane stride were sized for)
        const uint32_t r = *rows_dev;
        if (r == 0) return;
        M = min(M, r);
    }
    stage_weights(lds, packed + P::kFwdOffset, (int)P::kFwdLds);
    field_fwd_pipelined<P, kMixed>(lds, threadIdx.x & 63, dirs, M, add_codebook, planes, stride, sigmas, rgbs, masks);
}

template <typename P, int kPlanes, bool kTrace = false>
__global__ void __launch_bounds__(256) k_field_fwd(const float *__restrict__ xyzs, const float *__restrict__ dirs, uint32_t M, float bound,
                                                   TablePtrs base, LevelGeom geom, const float *__restrict__ S,
                                                   const float2 *__restrict__ planes, uint32_t stride,
                                                   const char *__restrict__ packed, float *__restrict__ sigmas, float *__restrict__ rgbs,
                                                   float *__restrict__ geo_out, uint32_t *__restrict__ masks, ActTrace trace = ActTrace{},
                                                   const uint32_t *__restrict__ rows_dev = nullptr, bool mixed = false) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    if (rows_dev != nullptr) {      // (field_fwd_rows)
        const uint32_t r = *rows_dev;
        if (r == 0) return;
        M = min(M, r);
    }
    stage_weights(lds, packed + P::kFwdOffset, (int)P::kFwdLds);
    constexpr size_t kHalf = kFwdBytes;

    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int p = lane & 31, h = lane >> 5;
    const uint32_t n_tiles = ceil_div(M, 32u);
    for (uint32_t tile = blockIdx.x * 4 + wid; tile < n_tiles; tile += gridDim.x * 4) {
        const uint32_t s = tile * 32 + p;
        const uint32_t sl = min(s, M - 1);
        // lane half 0 owns levels {0..3, 8..11}, half 1 owns {4..7, 12..15}: its 16 features are exactly
        // its elements of the two K-steps of the first layer's B operand.
        typename P::Op feat[2];
        if (kPlanes) {
            float2 f[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) f[q] = load_plane(planes, stride, 8 * (q >> 2) + (q & 3) + 4 * h, s, mixed);  // 256 (mixed: 128) contiguous bytes per half-wave
            if (S != nullptr && h) {  // codebook added into channels 30:32 (network_wtmk_tcnn.py:106)
                const float2 c = load_plane(planes, stride, NSIG_BASE_LEVELS, s, mixed);
                f[7].x = f[7].x + c.x;
                f[7].y = f[7].y + c.y;
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) P::put2(feat[q >> 2], q & 3, f[q].x, f[q].y);
        } else {
            const float two_b = 2.0f * bound;
            const float x = (xyzs[3 * (size_t)sl] + bound) / two_b;       // network_wtmk_tcnn.py:101
            const float y = (xyzs[3 * (size_t)sl + 1] + bound) / two_b;
            const float z = (xyzs[3 * (size_t)sl + 2] + bound) / two_b;
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int l0 = 8 * g + i, l1 = l0 + 4;
                    float2 f = encode_level(h ? base.p[l1] : base.p[l0], x, y, z, h ? geom.cell[l1] : geom.cell[l0]);
                    if (g == 1 && i == 3 && S != nullptr && h) {  // codebook added into channels 30:32 (:106)
                        const float2 c = encode_level(S, x, y, z, geom.cell[NSIG_BASE_LEVELS]);
                        f.x = f.x + c.x;
                        f.y = f.y + c.y;
                    }
                    P::put2(feat[g], i, f.x, f.y);
                }
        }

        f32x16 hid[2];
        typename P::Op b4[4];
        mfma_layer<P, 2, 2>(lds, kHalf, F0, lane, feat, hid);
        const uint32_t mask_s = relu_to_operand<P>(hid, b4);
        if (kTrace) store_rows64(trace.hs, stride, s, h, hid, [](float v, int) { return v > 0.0f ? v : 0.0f; });
        f32x16 so[1];
        mfma_layer<P, 1, 4>(lds, kHalf, F1, lane, b4, so);

        // rows 0..15 of the sigma head: register r (< 8) of half h is row_of_reg(h, r); row 0 is log-density
        const bool live = s < M;
        if (live && h == 0) sigmas[s] = expf(so[0][0]);  // trunc_exp forward (activation.py:9)
        if (geo_out != nullptr && live) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int rho = row_of_reg(h, r);
                if (rho >= 1) geo_out[15 * (size_t)s + rho - 1] = so[0][r];
            }
        }
        uint32_t mask_c[2] = {0u, 0u};
        if (rgbs != nullptr) {
            float geo8[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) geo8[r] = so[0][r];
            if (h == 0) geo8[0] = 1.0f;  // the slot of row 0 carries the padded constant input (weight column 31)
            float rgb[3];
            color_branch<P>(lds, lane, h, dirs[3 * (size_t)sl], dirs[3 * (size_t)sl + 1], dirs[3 * (size_t)sl + 2], geo8, mask_c, rgb,
                            kTrace ? &trace : nullptr, stride, s);
            if (live && h == 0) { rgbs[3 * (size_t)s] = rgb[0]; rgbs[3 * (size_t)s + 1] = rgb[1]; rgbs[3 * (size_t)s + 2] = rgb[2]; }
        }
        if (masks != nullptr) {
            uint32_t *mrow = masks + (size_t)tile * 192 + lane;
            mrow[0] = mask_s; mrow[64] = mask_c[0]; mrow[128] = mask_c[1];
        }
    }
}

// field_fwd_trace's launch: k_field_fwd<Bf16x3, 1, true> -- the same arithmetic, instruction for instruction: the same bits in every output -- with its inputs
// software-pipelined.  The plain loop asks for a tile's 8 feature planes at the tile's head and for its direction in the middle of the tile; loads return in
// order, so the planes wait for the ~100 trace stores of the previous tile to be acknowledged and the direction for the 32 stores of the sigma layer in front of it
// (counters, 718 k points: 54 % of the wave cycles in s_waitcnt).  Here the NEXT tile's planes and direction are requested at a tile's head, in front of all of the
// tile's stores: they are there when the next tile begins.  Requests are unconditional (past a wave's last tile: the launch's last tile again, never used).
template <typename TT>
__global__ void __launch_bounds__(256) k_field_fwd_trace(const float *__restrict__ dirs, uint32_t M, const float2 *__restrict__ planes, uint32_t stride,
                                                         const char *__restrict__ packed, float *__restrict__ sigmas, float *__restrict__ rgbs,
                                                         uint32_t *__restrict__ masks, ActTrace trace, const uint32_t *__restrict__ rows_dev) {
    typedef Bf16x3 P;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    if (rows_dev != nullptr) {      // (field_fwd_trace_rows)
        const uint32_t r = *rows_dev;
        if (r == 0) return;
        M = min(M, r);
    }
    stage_weights(lds, packed + P::kFwdOffset, (int)P::kFwdLds);
    constexpr size_t kHalf = kFwdBytes;
    const int lane = threadIdx.x & 63, p = lane & 31, h = lane >> 5;
    const uint32_t wid = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t n_tiles = ceil_div(M, 32u), step = gridDim.x * 4u;
    typedef float f32x3u __attribute__((ext_vector_type(3), aligned(4)));
    float2 nf[8];
    f32x3u nd;
    auto request = [&](uint32_t tl) {
        const uint32_t s = tl * 32 + p, sl = min(s, M - 1);
#pragma unroll
        for (int q = 0; q < 8; ++q) nf[q] = planes[(size_t)(8 * (q >> 2) + (q & 3) + 4 * h) * stride + s];      // lane half 0 owns levels {0..3, 8..11}, half 1 {4..7, 12..15}
        nd = *reinterpret_cast<const f32x3u *>(dirs + 3 * (size_t)sl);
    };
    uint32_t tile = blockIdx.x * 4u + wid;
    request(min(tile, n_tiles - 1u));
    for (; tile < n_tiles; tile += step) {
        const uint32_t s = tile * 32 + p;
        const bool live = s < M;
        float fx[8], fy[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) { fx[q] = nf[q].x; fy[q] = nf[q].y; }
        float dx = nd.x, dy = nd.y, dz = nd.z;
        // compiler barrier that consumes the inputs: the wait for them stands here, in front of the request and the stores below
        asm volatile("" : "+v"(fx[0]), "+v"(fx[1]), "+v"(fx[2]), "+v"(fx[3]), "+v"(fx[4]), "+v"(fx[5]), "+v"(fx[6]), "+v"(fx[7]), "+v"(fy[0]), "+v"(fy[1]), "+v"(fy[2]),
                     "+v"(fy[3]), "+v"(fy[4]), "+v"(fy[5]), "+v"(fy[6]), "+v"(fy[7]), "+v"(dx), "+v"(dy), "+v"(dz) :: "memory");
        request(min(tile + step, n_tiles - 1u));
        asm volatile("" ::: "memory");

        typename P::Op feat[2];
#pragma unroll
        for (int q = 0; q < 8; ++q) P::put2(feat[q >> 2], q & 3, fx[q], fy[q]);
        f32x16 hid[2];
        typename P::Op b4[4];
        mfma_layer<P, 2, 2>(lds, kHalf, F0, lane, feat, hid);
        const uint32_t mask_s = relu_to_operand<P>(hid, b4);
        store_rows64<TT>(trace.hs, stride, s, h, hid, [](float v, int) { return v > 0.0f ? v : 0.0f; });
        f32x16 so[1];
        mfma_layer<P, 1, 4>(lds, kHalf, F1, lane, b4, so);
        if (live && h == 0) sigmas[s] = expf(so[0][0]);  // trunc_exp forward (activation.py:9)
        uint32_t mask_c[2] = {0u, 0u};
        float geo8[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) geo8[r] = so[0][r];
        if (h == 0) geo8[0] = 1.0f;  // the slot of row 0 carries the padded constant input (weight column 31)
        float rgb[3];
        color_branch<P, TT>(lds, lane, h, dx, dy, dz, geo8, mask_c, rgb, &trace, stride, s);
        if (live && h == 0) { rgbs[3 * (size_t)s] = rgb[0]; rgbs[3 * (size_t)s + 1] = rgb[1]; rgbs[3 * (size_t)s + 2] = rgb[2]; }
        uint32_t *mrow = masks + (size_t)tile * 192 + lane;
        mrow[0] = mask_s; mrow[64] = mask_c[0]; mrow[128] = mask_c[1];
    }
}

// NeRFNetwork.color: geo features come from memory instead of the sigma head.
template <typename P>
__global__ void __launch_bounds__(256) k_field_color(const float *__restrict__ dirs, const float *__restrict__ geo, uint32_t M,
                                                     const char *__restrict__ packed, float *__restrict__ rgbs) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    stage_weights(lds, packed + P::kFwdOffset, (int)P::kFwdLds);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int p = lane & 31, h = lane >> 5;
    const uint32_t n_tiles = ceil_div(M, 32u);
    for (uint32_t tile = blockIdx.x * 4 + wid; tile < n_tiles; tile += gridDim.x * 4) {
        const uint32_t s = tile * 32 + p;
        const uint32_t sl = min(s, M - 1);
        float geo8[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int rho = row_of_reg(h, r);
            geo8[r] = rho == 0 ? 1.0f : geo[15 * (size_t)sl + rho - 1];
        }
        uint32_t mask_c[2];
        float rgb[3];
        color_branch<P>(lds, lane, h, dirs[3 * (size_t)sl], dirs[3 * (size_t)sl + 1], dirs[3 * (size_t)sl + 2], geo8, mask_c, rgb);
        if (s < M && h == 0) { rgbs[3 * (size_t)s] = rgb[0]; rgbs[3 * (size_t)s + 1] = rgb[1]; rgbs[3 * (size_t)s + 2] = rgb[2]; }
    }
}

// ----------------------------------------------------------------------------- backward

template <typename P, bool kFull = false>
__global__ void __launch_bounds__(256) k_field_bwd(const float *__restrict__ xyzs, uint32_t M, float bound, float cb_cell,
                                                   const float *__restrict__ g_sigma, const float *__restrict__ g_rgb,
                                                   const float *__restrict__ sigmas, const float *__restrict__ rgbs,
                                                   const uint32_t *__restrict__ masks, const char *__restrict__ packed,
                                                   float *__restrict__ G, float *__restrict__ dfeat_out, float *__restrict__ rec_out,
                                                   GradTrace gt = GradTrace{}, uint32_t stride = 0, ScatterPlan plan = ScatterPlan{},
                                                   const uint32_t *__restrict__ rows_dev = nullptr) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    if (rows_dev != nullptr) {      // (field_bwd_trace_rows: the point count is a device value; `stride` keeps the buffers' layout)
        const uint32_t r = *rows_dev;
        if (r == 0) return;
        M = min(M, r);
    }
    stage_weights(lds, packed + P::kBwdOffset, (int)P::kBwdLds);
    constexpr size_t kHalf = kBwdBytes;
    // F16: the backward is linear in the point's upstream gradient, so that gradient is first scaled by a power of two that brings
    // its largest component into [1, 2) (exact), and the result scaled back: fp16's 5-bit exponent then never overflows or
    // underflows, whatever loss scale the caller's GradScaler applies.  (The stage-1 trace path stores intermediates: Bf16x3 only.)
    constexpr bool kNormalise = P::kMfmaPerProduct == 1 && !kFull;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int p = lane & 31, h = lane >> 5;
    const uint32_t n_tiles = ceil_div(M, 32u);
    const float e_lo = expf(-15.0f), e_hi = expf(15.0f);
    const bool planned = !kFull && plan.hd != nullptr;
    uint32_t gbits = 0;   // planned scatter: running max |gradient| of this lane's points, as a bit pattern
    // Tile order: a 1024-point chunk (32 tiles, the unit the scatter plan sorts by) is handled by 8 workgroups of ONE XCD at the
    // same time (workgroup b runs on XCD b % 8), so the 16-byte queue entries of a (chunk, slice) run -- written by many waves --
    // merge into whole lines in that XCD's L2 instead of leaving eight L2s with partial lines each.  gridDim.x % 8 == 0.
    const uint32_t xcd = blockIdx.x & 7u, per_xcd = gridDim.x >> 3, n_chunks = ceil_div(n_tiles, 32u);
    for (uint32_t lq = blockIdx.x >> 3;; lq += per_xcd) {
        const uint32_t chunk = (lq >> 3) * 8u + xcd;
        if (chunk >= n_chunks) break;
        const uint32_t tile = (chunk * 8u + (lq & 7u)) * 4u + wid;
        if (tile >= n_tiles) continue;
        const uint32_t s = tile * 32 + p;
        const bool live = s < M;
        const uint32_t sl = min(s, M - 1);
        const uint32_t *mrow = masks + (size_t)tile * 192 + lane;
        const uint32_t mask_s = mrow[0], mask_c0 = mrow[64], mask_c1 = mrow[128];
        uint4 dst = make_uint4(0u, 0u, 0u, 0u);
        if (planned) dst = plan.dest[sl];   // issued with the masks: long done when the gradients are

        // d(pre-sigmoid color): only lane half 0, elements 0..2 of the 16-wide K-step are non-zero
        typename P::Op dout[1];
        float dv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v = 0.0f;
            if (j < 3 && h == 0 && live) {
                const float c = rgbs[3 * (size_t)sl + j];
                v = g_rgb[3 * (size_t)sl + j] * (c * (1.0f - c));
            }
            dv[j] = v;
            if (kFull) gt.d_out[(size_t)(8 * h + j) * stride + s] = v;   // rows 3..15 are zero
        }
        // d log-density = g * exp(clamp(h0, -15, 15)) (activation.py:14), with exp(h0) = sigma: enters at the sigma head, row 0 (half 0)
        float head0 = 0.0f;
        if (h == 0 && live) head0 = g_sigma[sl] * fminf(fmaxf(sigmas[sl], e_lo), e_hi);
        float unscale = 1.0f;
        if (kNormalise) {
            float amp = fmaxf(fmaxf(fabsf(dv[0]), fabsf(dv[1])), fmaxf(fabsf(dv[2]), fabsf(head0)));   // (half 1 holds zeros)
            amp = fmaxf(amp, __shfl_xor(amp, 32, 64));
            const uint32_t e = (__float_as_uint(amp) >> 23) & 0xffu;
            if (e >= 1u && e <= 253u) {          // zero, subnormal and non-finite amplitudes pass through unscaled
                const float sc = __uint_as_float((254u - e) << 23);
                unscale = __uint_as_float(e << 23);
                dv[0] *= sc; dv[1] *= sc; dv[2] *= sc; head0 *= sc;
            }
        }
#pragma unroll
        for (int j = 0; j < 8; j += 2) P::put2(dout[0], j >> 1, dv[j], dv[j + 1]);
        f32x16 hid[2];
        typename P::Op b4[4];
        mfma_layer<P, 2, 1>(lds, kHalf, B0, lane, dout, hid);
        mask_to_operand<P>(hid, mask_c1, b4);
        if (kFull) store_rows64(gt.d_h2, stride, s, h, hid, [=](float v, int i) { return ((mask_c1 >> mask_bit<P>(i)) & 1u) ? v : 0.0f; });
        mfma_layer<P, 2, 4>(lds, kHalf, B1, lane, b4, hid);
        mask_to_operand<P>(hid, mask_c0, b4);
        if (kFull) store_rows64(gt.d_h1, stride, s, h, hid, [=](float v, int i) { return ((mask_c0 >> mask_bit<P>(i)) & 1u) ? v : 0.0f; });
        f32x16 dso[1];
        mfma_layer<P, 1, 4>(lds, kHalf, B2, lane, b4, dso);  // rows 1..15 = d geo_feat

        typename P::Op dhead[1];
        float head8[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) head8[r] = dso[0][r];
        if (h == 0) head8[0] = head0;  // row 0: d log-density
#pragma unroll
        for (int r = 0; r < 8; ++r)
            if (kFull) gt.d_so[(size_t)row_of_reg(h, r) * stride + s] = head8[r];
#pragma unroll
        for (int r = 0; r < 8; r += 2) P::put2(dhead[0], r >> 1, head8[r], head8[r + 1]);
        mfma_layer<P, 2, 1>(lds, kHalf, B3, lane, dhead, hid);
        mask_to_operand<P>(hid, mask_s, b4);
        if (kFull) {
            store_rows64(gt.d_hs, stride, s, h, hid, [=](float v, int i) { return ((mask_s >> mask_bit<P>(i)) & 1u) ? v : 0.0f; });
            f32x16 dall[1];
            mfma_layer<P, 1, 4>(lds, kHalf, B4F, lane, b4, dall);  // row f = d feature[f]; registers (r, r+1), r even, hold one level's pair
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                float2 v;
                v.x = dall[0][r]; v.y = dall[0][r + 1];
                gt.d_planes[(size_t)(row_of_reg16(h, r) >> 1) * stride + s] = v;
            }
            continue;
        }
        f32x16 dfe[1];
        mfma_layer<P, 1, 4>(lds, kHalf, B4, lane, b4, dfe);  // rows 0,1 (lane half 0) = d feature[30], d feature[31]

        // both halves of a point share the scatter: half h handles corners 4h..4h+3
        const float g0 = __shfl(dfe[0][0], p, 64) * unscale, g1 = __shfl(dfe[0][1], p, 64) * unscale;
        if (!live) continue;
        if (dfeat_out != nullptr && h == 0) { dfeat_out[2 * (size_t)s] = g0; dfeat_out[2 * (size_t)s + 1] = g1; }
        const float two_b = 2.0f * bound;
        const float x01 = (xyzs[3 * (size_t)s] + bound) / two_b, y01 = (xyzs[3 * (size_t)s + 1] + bound) / two_b, z01 = (xyzs[3 * (size_t)s + 2] + bound) / two_b;
        if (planned) {
            // The entries' places in the slice-sorted queue were fixed from the positions alone (hg_scatter_plan): lane half h
            // writes the two (dy, dz) pairs with dy = h.  Nothing else stands between this kernel and the slice owners.
            uint32_t ix, iy, iz;
            float wx, wy, wz;
            codebook_axis(x01, ix, wx);
            codebook_axis(y01, iy, wy);
            codebook_axis(z01, iz, wz);
            const uint32_t q0 = 2u * (uint32_t)h;
            plan.queue[h ? dst.z : dst.x] = pair_entry(ix, pair_hash(iy, iz, q0), wx, wy, wz, g0, g1, q0);
            plan.queue[h ? dst.w : dst.y] = pair_entry(ix, pair_hash(iy, iz, q0 + 1u), wx, wy, wz, g0, g1, q0 + 1u);
            gbits = max(gbits, max(__float_as_uint(g0) & 0x7fffffffu, __float_as_uint(g1) & 0x7fffffffu));   // |x| as bit patterns: ordered like the values, NaN above +inf (fmaxf would drop a NaN)
            continue;
        }
        if (rec_out != nullptr && h == 0) {
            // The scatter's workgroups re-read this record, so what is computed here once is not recomputed there: integer
            // cell, interpolation weights, gradients.  32 bytes per point, two 16-byte stores.
            uint32_t ix, iy, iz;
            float wx, wy, wz;
            codebook_axis(x01, ix, wx);
            codebook_axis(y01, iy, wy);
            codebook_axis(z01, iz, wz);
            uint4 *r4 = reinterpret_cast<uint4 *>(rec_out) + 2 * (size_t)s;
            r4[0] = make_uint4(ix | (iy << 16), iz, __float_as_uint(wx), __float_as_uint(wy));
            r4[1] = make_uint4(__float_as_uint(wz), __float_as_uint(g0), __float_as_uint(g1), 0u);
        }
        if (G == nullptr || (g0 == 0.0f && g1 == 0.0f)) continue;
        Corner8 c;
        corner_rows(x01, y01, z01, cb_cell, c);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if ((k >> 2) != h) continue;
            float *row = G + 2 * (size_t)c.row[k];
            atomicAdd(row, corner_weight(c, k, g0));
            atomicAdd(row + 1, corner_weight(c, k, g1));
        }
    }
    if (planned) {   // the owners' fixed-point scale comes from the launch's largest |gradient|
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) gbits = max(gbits, (uint32_t)__shfl_xor((int)gbits, d, 64));
        // 3072 waves finish together: atomics on one address serialise (~10 ns each), so only a wave that would raise the
        // maximum issues one -- after the first few, almost none does
        if (lane == 0 && gbits > __hip_atomic_load(&plan.hd->gmax_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&plan.hd->gmax_bits, gbits);
    }
}

// The training render's backward launch (planned scatter: gradients written straight into the slice-sorted queue), software-pipelined over a wave's
// tiles like field_fwd_pipelined: at the head of tile i the wave waits for tile i's inputs (requested at the head of tile i-1), computes and stores
// tile i-1's two queue entries per lane (held back as 7 registers: position, the two gradients, the two queue slots), requests tile i+1's inputs, and
// only then evaluates tile i.  The plain loop waited three times per tile -- for the inputs at its head, for the position in front of the queue
// entries, and at the next head for the acknowledgement of its own stores: 72 % of a wave's cycles parked (SQ_WAIT_ANY) at six waves per SIMD.
template <typename P>
__device__ inline void field_bwd_pipelined(const char *lds, int lane, const float *__restrict__ xyzs, uint32_t M, float bound,
                                           const float *__restrict__ g_sigma, const float *__restrict__ g_rgb, const float *__restrict__ sigmas,
                                           const float *__restrict__ rgbs, const uint32_t *__restrict__ masks, ScatterPlan plan) {
    constexpr size_t kHalf = kBwdBytes;
    constexpr bool kNormalise = P::kMfmaPerProduct == 1;
    const int p = lane & 31, h = lane >> 5;
    const uint32_t wid = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t n_tiles = ceil_div(M, 32u);
    const float e_lo = expf(-15.0f), e_hi = expf(15.0f), two_b = 2.0f * bound;
    uint32_t gbits = 0;   // running max |gradient| of this lane's points, as a bit pattern
    // tile order: see k_field_bwd (a 1024-point chunk is handled by 8 workgroups of one XCD at the same time)
    const uint32_t xcd = blockIdx.x & 7u, per_xcd = gridDim.x >> 3, n_chunks = ceil_div(n_tiles, 32u);
    uint32_t lq = blockIdx.x >> 3;
    auto next_tile = [&]() -> uint32_t {      // the wave's next tile, or ~0u when it has none left
        for (;; lq += per_xcd) {
            const uint32_t chunk = (lq >> 3) * 8u + xcd;
            if (chunk >= n_chunks) return ~0u;
            const uint32_t tile = (chunk * 8u + (lq & 7u)) * 4u + wid;
            if (tile < n_tiles) {
                lq += per_xcd;
                return tile;
            }
        }
    };
    typedef float f32x3u __attribute__((ext_vector_type(3), aligned(4)));
    // requested inputs of the next tile (one value per load instruction: see field_fwd_pipelined on loop-carried triples)
    uint32_t n_mask[3] = {0u, 0u, 0u};
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4 n_dst = {0u, 0u, 0u, 0u};
    f32x3u n_rgb = {0.0f, 0.0f, 0.0f}, n_grgb = {0.0f, 0.0f, 0.0f}, n_xyz = {0.0f, 0.0f, 0.0f};
    float n_gs = 0.0f, n_sig = 0.0f;
    auto request = [&](uint32_t tile) {
        const uint32_t s = tile * 32 + p, sl = min(s, M - 1);
        const uint32_t *mrow = masks + (size_t)tile * 192 + lane;
        n_mask[0] = mrow[0]; n_mask[1] = mrow[64]; n_mask[2] = mrow[128];
        n_dst = *reinterpret_cast<const u32x4 *>(plan.dest + sl);
        n_rgb = *reinterpret_cast<const f32x3u *>(rgbs + 3 * (size_t)sl);
        n_grgb = *reinterpret_cast<const f32x3u *>(g_rgb + 3 * (size_t)sl);
        n_xyz = *reinterpret_cast<const f32x3u *>(xyzs + 3 * (size_t)sl);
        n_gs = g_sigma[sl];
        n_sig = sigmas[sl];
    };
    // results of the previous tile, not stored yet
    bool have = false;
    float p_x = 0.0f, p_y = 0.0f, p_z = 0.0f, p_g0 = 0.0f, p_g1 = 0.0f;
    uint32_t p_d0 = 0u, p_d1 = 0u;
    bool p_live = false;
    auto store_prev = [&]() {
        if (!p_live) return;
        // The entries' places in the slice-sorted queue were fixed from the positions alone (hg_scatter_plan): lane half h
        // writes the two (dy, dz) pairs with dy = h.  Nothing else stands between this kernel and the slice owners.
        uint32_t ix, iy, iz;
        float wx, wy, wz;
        codebook_axis((p_x + bound) / two_b, ix, wx);
        codebook_axis((p_y + bound) / two_b, iy, wy);
        codebook_axis((p_z + bound) / two_b, iz, wz);
        const uint32_t q0 = 2u * (uint32_t)h;
        plan.queue[p_d0] = pair_entry(ix, pair_hash(iy, iz, q0), wx, wy, wz, p_g0, p_g1, q0);
        plan.queue[p_d1] = pair_entry(ix, pair_hash(iy, iz, q0 + 1u), wx, wy, wz, p_g0, p_g1, q0 + 1u);
        gbits = max(gbits, max(__float_as_uint(p_g0) & 0x7fffffffu, __float_as_uint(p_g1) & 0x7fffffffu));   // |x| as bit patterns: ordered like the values, NaN above +inf
    };
    uint32_t tile = next_tile();
    if (tile != ~0u) request(tile);
    while (tile != ~0u) {
        const uint32_t s = tile * 32 + p;
        const bool live = s < M;
        // consume the inputs: d(pre-sigmoid color) -- only lane half 0, elements 0..2 of the 16-wide K-step are non-zero -- and
        // d log-density = g * exp(clamp(h0, -15, 15)) (activation.py:14), with exp(h0) = sigma: enters at the sigma head, row 0 (half 0)
        float dv[3] = {0.0f, 0.0f, 0.0f}, head0 = 0.0f;
        if (h == 0 && live) {
            dv[0] = n_grgb.x * (n_rgb.x * (1.0f - n_rgb.x));
            dv[1] = n_grgb.y * (n_rgb.y * (1.0f - n_rgb.y));
            dv[2] = n_grgb.z * (n_rgb.z * (1.0f - n_rgb.z));
            head0 = n_gs * fminf(fmaxf(n_sig, e_lo), e_hi);
        }
        uint32_t mask_s = n_mask[0], mask_c0 = n_mask[1], mask_c1 = n_mask[2];
        uint32_t d0 = h ? n_dst.z : n_dst.x, d1 = h ? n_dst.w : n_dst.y;
        float cx = n_xyz.x, cy = n_xyz.y, cz = n_xyz.z;
        // compiler barrier that consumes the inputs: the wait for them stands here, in front of the stores and requests below
        asm volatile("" : "+v"(dv[0]), "+v"(dv[1]), "+v"(dv[2]), "+v"(head0), "+v"(mask_s), "+v"(mask_c0), "+v"(mask_c1), "+v"(d0), "+v"(d1),
                     "+v"(cx), "+v"(cy), "+v"(cz) :: "memory");
        if (have) store_prev();
        const uint32_t upcoming = next_tile();
        if (upcoming != ~0u) request(upcoming);
        asm volatile("" ::: "memory");

        float unscale = 1.0f;
        if (kNormalise) {    // (see k_field_bwd: the upstream gradient scaled into [1, 2) by a power of two, the result scaled back)
            float amp = fmaxf(fmaxf(fabsf(dv[0]), fabsf(dv[1])), fmaxf(fabsf(dv[2]), fabsf(head0)));   // (half 1 holds zeros)
            amp = fmaxf(amp, __shfl_xor(amp, 32, 64));
            const uint32_t e = (__float_as_uint(amp) >> 23) & 0xffu;
            if (e >= 1u && e <= 253u) {          // zero, subnormal and non-finite amplitudes pass through unscaled
                const float sc = __uint_as_float((254u - e) << 23);
                unscale = __uint_as_float(e << 23);
                dv[0] *= sc; dv[1] *= sc; dv[2] *= sc; head0 *= sc;
            }
        }
        typename P::Op dout[1];
        P::put2(dout[0], 0, dv[0], dv[1]);
        P::put2(dout[0], 1, dv[2], 0.0f);
        P::put2(dout[0], 2, 0.0f, 0.0f);
        P::put2(dout[0], 3, 0.0f, 0.0f);
        f32x16 hid[2];
        typename P::Op b4[4];
        mfma_layer<P, 2, 1>(lds, kHalf, B0, lane, dout, hid);
        mask_to_operand<P>(hid, mask_c1, b4);
        mfma_layer<P, 2, 4>(lds, kHalf, B1, lane, b4, hid);
        mask_to_operand<P>(hid, mask_c0, b4);
        f32x16 dso[1];
        mfma_layer<P, 1, 4>(lds, kHalf, B2, lane, b4, dso);  // rows 1..15 = d geo_feat
        typename P::Op dhead[1];
        float head8[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) head8[r] = dso[0][r];
        if (h == 0) head8[0] = head0;  // row 0: d log-density
#pragma unroll
        for (int r = 0; r < 8; r += 2) P::put2(dhead[0], r >> 1, head8[r], head8[r + 1]);
        mfma_layer<P, 2, 1>(lds, kHalf, B3, lane, dhead, hid);
        mask_to_operand<P>(hid, mask_s, b4);
        f32x16 dfe[1];
        mfma_layer<P, 1, 4>(lds, kHalf, B4, lane, b4, dfe);  // rows 0,1 (lane half 0) = d feature[30], d feature[31]
        // both halves of a point share the scatter: half h handles corners 4h..4h+3
        p_g0 = __shfl(dfe[0][0], p, 64) * unscale;
        p_g1 = __shfl(dfe[0][1], p, 64) * unscale;
        p_x = cx; p_y = cy; p_z = cz; p_d0 = d0; p_d1 = d1; p_live = live;
        have = true;
        tile = upcoming;
    }
    if (have) store_prev();
    // the owners' fixed-point scale comes from the launch's largest |gradient|
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) gbits = max(gbits, (uint32_t)__shfl_xor((int)gbits, d, 64));
    if (lane == 0 && gbits > __hip_atomic_load(&plan.hd->gmax_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&plan.hd->gmax_bits, gbits);
}

template <typename P>
__global__ void __launch_bounds__(256) k_field_bwd_train(const float *__restrict__ xyzs, uint32_t M, float bound, const float *__restrict__ g_sigma,
                                                         const float *__restrict__ g_rgb, const float *__restrict__ sigmas, const float *__restrict__ rgbs,
                                                         const uint32_t *__restrict__ masks, const char *__restrict__ packed, ScatterPlan plan) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    stage_weights(lds, packed + P::kBwdOffset, (int)P::kBwdLds);
    field_bwd_pipelined<P>(lds, threadIdx.x & 63, xyzs, M, bound, g_sigma, g_rgb, sigmas, rgbs, masks, plan);
}

}  // namespace nsig

using namespace nsig;

NSIG_EXPORT size_t mlp_packed_bytes(void) { return kPackedBytes; }

// 0 = Bf16x3 (split-bf16, three MFMAs per product, fp32-level accuracy), 1 = F16 (one fp16 MFMA per product, fp32 accumulate).
// Process-wide; default from NERFSIG_MLP ("bf16x3" | "f16").  Both operand images live in every packed buffer.
static int g_mlp_precision = -1;
static int mlp_precision() {
    if (g_mlp_precision < 0) {
        const char *e = getenv("NERFSIG_MLP");
        g_mlp_precision = (e != nullptr && (e[0] == 'b' || e[0] == 'B')) ? 0 : 1;
    }
    return g_mlp_precision;
}
NSIG_EXPORT int mlp_get_precision(void) { return mlp_precision(); }
NSIG_EXPORT int mlp_set_precision(int mode) {
    NSIG_REQUIRE(mode == 0 || mode == 1, "mlp_set_precision: 0 (bf16x3) or 1 (f16)");
    g_mlp_precision = mode;
    return NSIG_OK;
}

NSIG_EXPORT int mlp_pack_weights(const float *sigma_params, const float *color_params, void *packed, nsig_stream_t stream) {
    NSIG_REQUIRE(sigma_params && color_params && packed, "mlp_pack_weights: null pointer");
    NSIG_REQUIRE((reinterpret_cast<uintptr_t>(packed) & 15) == 0, "mlp_pack_weights: packed must be 16-byte aligned");
    const int total = (kFwdFrags + kBwdFrags) * 512;
    k_pack_weights<<<ceil_div(total, 256), 256, 0, as_stream(stream)>>>(sigma_params, color_params, reinterpret_cast<__bf16 *>(packed));
    return check_launch("mlp_pack_weights");
}

// Bit 0: the training render's forward goes through k_field_fwd_train, bit 1: its planned backward through k_field_bwd_train (the software-pipelined
// launches; results bit-identical to the plain loops k_field_fwd<F16, 1> / k_field_bwd<F16>); bit 0 also selects k_field_fwd_trace for field_fwd_trace(_rows)
// (plain loop: k_field_fwd<Bf16x3, 1, true>, the same bits).  Default both; mlp_set_pipelined() selects the plain loops (the bit-identity tests do).
static int g_mlp_pipelined = 3;
static int mlp_pipelined() {
    return g_mlp_pipelined;
}
static bool fwd_pipelined() { return (mlp_pipelined() & 1) != 0; }
static bool bwd_pipelined() { return (mlp_pipelined() & 2) != 0; }
NSIG_EXPORT int mlp_get_pipelined(void) { return mlp_pipelined(); }
NSIG_EXPORT int mlp_set_pipelined(int mask) {
    NSIG_REQUIRE(mask >= 0 && mask <= 3, "mlp_set_pipelined: bit 0 = forward, bit 1 = backward");
    g_mlp_pipelined = mask;
    return NSIG_OK;
}
static uint32_t field_grid(uint32_t M, bool forward = false, uint32_t per_cu = 0) {
    const uint32_t blocks = ceil_div(ceil_div(ceil_div(M, 32u), 4u), 8u) * 8u;   // a multiple of 8: k_field_bwd's XCD-aware tile order
    // persistent workgroups (the packed weights are staged once per workgroup): 3 per CU fit the LDS budget.  Same-box sweeps
    // (profiles/r01_k_field_grid_sweep.txt): the backward is fastest with exactly the resident 768 (113-117 us; 128 with 512 or 1024),
    // the forward with 512 or 1024 (117-120 us against 124-125 with 768) and the step with 1024 (1.118-1.123 ms against 1.132-1.133).
    const uint32_t cap = (uint32_t)kCUs * (per_cu ? per_cu : (forward ? 4u : 3u));
    return blocks < cap ? blocks : cap;
}

static int fill_base_tables(const float *const *host, TablePtrs &base, const char *who) {
    NSIG_REQUIRE(host, "%s: null base table list", who);
    for (int l = 0; l < NSIG_BASE_LEVELS; ++l) {
        NSIG_REQUIRE(host[l] != nullptr, "%s: base table %d is null", who, l);
        base.p[l] = host[l];
    }
    return NSIG_OK;
}

// Level -> XCD-slot assignment of k_encode_planes: the six finest levels each get a slot of their own or share it only with coarse
// (cache-resident) levels; whole levels per slot.  The assignment that won round 1's same-box sweeps (profiles/r01_k_encoder_slots_sweep.txt); a
// cost-balanced split with levels cut across slots was measured slower (287-292 against 283 us, profiles/r02_encoder_experiments.txt: with all eight
// XCDs busy the launch is bound chip-wide by L2->L1 line fills, not by its fullest slot) and is gone.
static SlotTable default_slots(bool with_codebook) {
    static const uint8_t kWith[8][4] = {{16, 255, 255, 255}, {15, 255, 255, 255}, {14, 0, 255, 255}, {13, 1, 255, 255}, {12, 2, 3, 255}, {11, 4, 5, 255}, {10, 9, 255, 255}, {8, 7, 6, 255}};
    static const uint8_t kWithout[8][4] = {{15, 255, 255, 255}, {14, 255, 255, 255}, {13, 255, 255, 255}, {12, 255, 255, 255}, {11, 0, 1, 255}, {10, 2, 3, 255}, {9, 8, 4, 255}, {7, 6, 5, 255}};
    SlotTable t{};
    for (int slot = 0; slot < 8; ++slot)
        for (int k = 0; k < 4; ++k) {
            const uint8_t l = (with_codebook ? kWith : kWithout)[slot][k];
            if (l != 255) t.level[slot][t.n[slot]++] = l;
        }
    return t;
}

NSIG_EXPORT int hg_warm_tables(const float *const *base_tables_host, const float *S, float *sink, nsig_stream_t stream) {
    NSIG_REQUIRE(sink != nullptr, "hg_warm_tables: sink is one writable float (never written)");
    TablePtrs base{};
    if (int e = fill_base_tables(base_tables_host, base, "hg_warm_tables")) return e;
    k_warm_tables<<<8 * 128, 256, 0, as_stream(stream)>>>(base, S, default_slots(S != nullptr), sink);
    return check_launch("hg_warm_tables");
}

NSIG_EXPORT size_t hg_planes_bytes(uint32_t M) { return (size_t)(NSIG_BASE_LEVELS + 1) * ceil_div(M, 32u) * 32u * sizeof(float2); }

// A plane set is in one of two layouts, NSIG_PLANES_F32 ([17][stride] float2) or NSIG_PLANES_MIXED (hg_encode_planes_mixed); the OWNER of the buffer says which when
// it hands the set to an entry point that reads or completes it (field_fwd / field_fwd_rows / hg_encode_codebook_plane: `planes_layout`).
static int check_layout(int layout, const char *who) {
    NSIG_REQUIRE(layout == NSIG_PLANES_F32 || layout == NSIG_PLANES_MIXED, "%s: planes_layout must be NSIG_PLANES_F32 (0) or NSIG_PLANES_MIXED (1)", who);
    return NSIG_OK;
}

static int encode_planes_impl(const float *xyzs, uint32_t M, float bound, const float *const *base_tables_host, const float *S, void *planes,
                              const uint32_t *rows_dev, nsig_stream_t stream, bool mixed = false) {
    if (M == 0) return NSIG_OK;
    NSIG_REQUIRE(xyzs && planes, "hg_encode_planes: null pointer");
    NSIG_REQUIRE(bound > 0.0f, "hg_encode_planes: bound must be positive");
    NSIG_REQUIRE((reinterpret_cast<uintptr_t>(planes) & 7) == 0, "hg_encode_planes: planes must be 8-byte aligned");
    TablePtrs base{};
    if (int e = fill_base_tables(base_tables_host, base, "hg_encode_planes")) return e;
    const uint32_t stride = ceil_div(M, 32u) * 32u;
    const SlotTable tab = default_slots(S != nullptr);
    int covered[NSIG_BASE_LEVELS + 1] = {};     // slots that encode the level: exactly one
    for (int s = 0; s < 8; ++s)
        for (int i = 0; i < tab.n[s]; ++i) covered[tab.level[s][i]] += 1;
    for (int l = 0; l < NSIG_BASE_LEVELS + (S != nullptr ? 1 : 0); ++l)
        NSIG_REQUIRE(covered[l] == 1, "hg_encode_planes: level %d is assigned to %d slots", l, covered[l]);
    NSIG_REQUIRE(S != nullptr || covered[NSIG_BASE_LEVELS] == 0, "hg_encode_planes: slot table names the codebook level but S is NULL");
    const uint32_t tiles = ceil_div(stride, 256u);
    // tiles per XCD slot handled by distinct workgroups before they start looping: with one tile per workgroup (cap >= tiles) the block
    // render's launch takes 244-247 us against 258-261 us with 1024 looping workgroups per slot (same-box sweep, profiles/r01_k_encoder_grid_sweep.txt)
    const uint32_t per_slot = tiles < 8192u ? tiles : 8192u;
    if (mixed) k_encode_planes<true><<<per_slot * 8, 256, 0, as_stream(stream)>>>(xyzs, M, bound, base, make_level_geom(), S, reinterpret_cast<float2 *>(planes), stride, tab, rows_dev);
    else k_encode_planes<false><<<per_slot * 8, 256, 0, as_stream(stream)>>>(xyzs, M, bound, base, make_level_geom(), S, reinterpret_cast<float2 *>(planes), stride, tab, rows_dev);
    return check_launch("hg_encode_planes");
}

NSIG_EXPORT int hg_encode_planes_mixed(const float *xyzs, uint32_t M_capacity, const uint32_t *rows_dev, float bound, const float *const *base_tables_host,
                                       const float *S, void *planes, nsig_stream_t stream) {
    NSIG_REQUIRE(mlp_precision() == 1, "hg_encode_planes_mixed: the mixed layout carries the fp16 MLP's operands (mlp_set_precision(1))");
    return encode_planes_impl(xyzs, M_capacity, bound, base_tables_host, S, planes, rows_dev, stream, true);
}

NSIG_EXPORT int hg_encode_planes(const float *xyzs, uint32_t M, float bound, const float *const *base_tables_host, const float *S, void *planes,
                                 nsig_stream_t stream) {
    return encode_planes_impl(xyzs, M, bound, base_tables_host, S, planes, nullptr, stream);
}

NSIG_EXPORT int hg_encode_planes_rows(const float *xyzs, uint32_t M_capacity, const uint32_t *rows_dev, float bound, const float *const *base_tables_host,
                                      const float *S, void *planes, nsig_stream_t stream) {
    NSIG_REQUIRE(rows_dev != nullptr, "hg_encode_planes_rows: null row count");
    return encode_planes_impl(xyzs, M_capacity, bound, base_tables_host, S, planes, rows_dev, stream);
}

NSIG_EXPORT int hg_encode_codebook_plane(const float *xyzs, uint32_t M, float bound, const float *S, void *planes, int planes_layout, void *plan_to_reset,
                                         nsig_stream_t stream) {
    if (M == 0) return NSIG_OK;
    if (int e = check_layout(planes_layout, "hg_encode_codebook_plane")) return e;
    NSIG_REQUIRE(xyzs && S && planes, "hg_encode_codebook_plane: null pointer");
    NSIG_REQUIRE(bound > 0.0f, "hg_encode_codebook_plane: bound must be positive");
    NSIG_REQUIRE((reinterpret_cast<uintptr_t>(planes) & 7) == 0, "hg_encode_codebook_plane: planes must be 8-byte aligned");
    NSIG_REQUIRE(plan_to_reset == nullptr || (reinterpret_cast<uintptr_t>(plan_to_reset) & 15) == 0, "hg_encode_codebook_plane: plan must be 16-byte aligned");
    const uint32_t stride = ceil_div(M, 32u) * 32u;
    float2 *plane = planes_layout == NSIG_PLANES_MIXED ? mixed_f32_plane(planes, stride, NSIG_BASE_LEVELS) : reinterpret_cast<float2 *>(planes) + (size_t)NSIG_BASE_LEVELS * stride;
    k_encode_codebook_plane<<<ceil_div(stride, 256u), 256, 0, as_stream(stream)>>>(xyzs, M, bound, make_level_geom().cell[NSIG_BASE_LEVELS], S, plane, stride,
                                                                                  reinterpret_cast<BinHeader *>(plan_to_reset));
    return check_launch("hg_encode_codebook_plane");
}

static int field_fwd_impl(const float *xyzs, const float *dirs, uint32_t M, float bound, const float *const *base_tables_host,
                          const float *S, const void *packed, float *sigmas, float *rgbs, float *geo_feat, uint32_t *masks,
                          const void *planes, int planes_layout, const uint32_t *rows_dev, nsig_stream_t stream) {
    if (M == 0) return NSIG_OK;
    NSIG_REQUIRE(xyzs && packed && sigmas, "field_fwd: null pointer");
    NSIG_REQUIRE(rgbs == nullptr || dirs != nullptr, "field_fwd: dirs is required when rgbs is requested");
    NSIG_REQUIRE(bound > 0.0f, "field_fwd: bound must be positive");
    NSIG_REQUIRE((reinterpret_cast<uintptr_t>(packed) & 15) == 0, "field_fwd: packed must be 16-byte aligned");
    TablePtrs base{};
    if (int e = fill_base_tables(base_tables_host, base, "field_fwd")) return e;
    const char *pk = reinterpret_cast<const char *>(packed);
    hipStream_t st = as_stream(stream);
    const bool f16 = mlp_precision() == 1;
    if (planes == nullptr) {  // fused: gather inside the MLP kernel (small batches)
        if (f16) k_field_fwd<F16, 0><<<field_grid(M, true), 256, F16::kFwdLds, st>>>(xyzs, dirs, M, bound, base, make_level_geom(), S, nullptr, 0, pk, sigmas, rgbs, geo_feat, masks, ActTrace{}, rows_dev);
        else k_field_fwd<Bf16x3, 0><<<field_grid(M, true), 256, Bf16x3::kFwdLds, st>>>(xyzs, dirs, M, bound, base, make_level_geom(), S, nullptr, 0, pk, sigmas, rgbs, geo_feat, masks, ActTrace{}, rows_dev);
        return check_launch("field_fwd");
    }
    NSIG_REQUIRE((reinterpret_cast<uintptr_t>(planes) & 7) == 0, "field_fwd: planes must be 8-byte aligned");
    const uint32_t stride = ceil_div(M, 32u) * 32u;
    const float2 *pl = reinterpret_cast<const float2 *>(planes);
    if (int e = check_layout(planes_layout, "field_fwd")) return e;
    const bool mixed = planes_layout == NSIG_PLANES_MIXED;
    NSIG_REQUIRE(!mixed || f16, "field_fwd: this plane set was written in the mixed (fp16) layout; the split-bf16 MLP needs hg_encode_planes");
    if (f16 && fwd_pipelined() && dirs != nullptr && rgbs != nullptr && geo_feat == nullptr) {    // the training render's launch (masks) and staged no-grad renders
        if (mixed) k_field_fwd_train<F16, true><<<field_grid(M, true, 2), 256, F16::kFwdLds, st>>>(dirs, M, S != nullptr, pl, stride, pk, sigmas, rgbs, masks, rows_dev);
        else k_field_fwd_train<F16, false><<<field_grid(M, true, 2), 256, F16::kFwdLds, st>>>(dirs, M, S != nullptr, pl, stride, pk, sigmas, rgbs, masks, rows_dev);
        return check_launch("field_fwd");
    }
    if (f16) k_field_fwd<F16, 1><<<field_grid(M, true), 256, F16::kFwdLds, st>>>(xyzs, dirs, M, bound, base, make_level_geom(), S, pl, stride, pk, sigmas, rgbs, geo_feat, masks, ActTrace{}, rows_dev, mixed);
    else k_field_fwd<Bf16x3, 1><<<field_grid(M, true), 256, Bf16x3::kFwdLds, st>>>(xyzs, dirs, M, bound, base, make_level_geom(), S, pl, stride, pk, sigmas, rgbs, geo_feat, masks, ActTrace{}, rows_dev);
    return check_launch("field_fwd");
}

NSIG_EXPORT int field_fwd(const float *xyzs, const float *dirs, uint32_t M, float bound, const float *const *base_tables_host,
                          const float *S, const void *packed, float *sigmas, float *rgbs, float *geo_feat, uint32_t *masks,
                          const void *planes, int planes_layout, nsig_stream_t stream) {
    return field_fwd_impl(xyzs, dirs, M, bound, base_tables_host, S, packed, sigmas, rgbs, geo_feat, masks, planes, planes_layout, nullptr, stream);
}

NSIG_EXPORT int field_fwd_rows(const float *xyzs, const float *dirs, uint32_t M_capacity, const uint32_t *rows_dev, float bound,
                               const float *const *base_tables_host, const float *S, const void *packed, float *sigmas, float *rgbs, const void *planes,
                               int planes_layout, nsig_stream_t stream) {
    NSIG_REQUIRE(rows_dev != nullptr, "field_fwd_rows: null row count");
    return field_fwd_impl(xyzs, dirs, M_capacity, bound, base_tables_host, S, packed, sigmas, rgbs, nullptr, nullptr, planes, planes_layout, rows_dev, stream);
}

NSIG_EXPORT int field_color_fwd(const float *dirs, const float *geo_feat, uint32_t M, const void *packed, float *rgbs, nsig_stream_t stream) {
    if (M == 0) return NSIG_OK;
    NSIG_REQUIRE(dirs && geo_feat && packed && rgbs, "field_color_fwd: null pointer");
    if (mlp_precision() == 1) k_field_color<F16><<<field_grid(M), 256, F16::kFwdLds, as_stream(stream)>>>(dirs, geo_feat, M, reinterpret_cast<const char *>(packed), rgbs);
    else k_field_color<Bf16x3><<<field_grid(M), 256, Bf16x3::kFwdLds, as_stream(stream)>>>(dirs, geo_feat, M, reinterpret_cast<const char *>(packed), rgbs);
    return check_launch("field_color_fwd");
}

NSIG_EXPORT int field_bwd(const float *xyzs, uint32_t M, float bound, const float *grad_sigmas, const float *grad_rgbs, const float *sigmas,
                          const float *rgbs, const uint32_t *masks, const void *packed, float *G, float *dfeat_out, float *rec_out,
                          nsig_stream_t stream) {
    if (M == 0) return NSIG_OK;
    NSIG_REQUIRE(xyzs && grad_sigmas && grad_rgbs && sigmas && rgbs && masks && packed, "field_bwd: null pointer");
    NSIG_REQUIRE(G || dfeat_out || rec_out, "field_bwd: at least one of G / dfeat_out / rec_out must be given");
    NSIG_REQUIRE(bound > 0.0f, "field_bwd: bound must be positive");
    if (mlp_precision() == 1)
        k_field_bwd<F16, false><<<field_grid(M), 256, F16::kBwdLds, as_stream(stream)>>>(xyzs, M, bound, 1.0f / kCodebookResolution, grad_sigmas, grad_rgbs, sigmas, rgbs, masks,
                                                                                 reinterpret_cast<const char *>(packed), G, dfeat_out, rec_out);
    else
        k_field_bwd<Bf16x3, false><<<field_grid(M), 256, Bf16x3::kBwdLds, as_stream(stream)>>>(xyzs, M, bound, 1.0f / kCodebookResolution, grad_sigmas, grad_rgbs, sigmas, rgbs,
                                                                                       masks, reinterpret_cast<const char *>(packed), G, dfeat_out, rec_out);
    return check_launch("field_bwd");
}

// field_bwd with the codebook scatter's queue as its output: `plan` was prepared by hg_scatter_plan from the same xyzs; follow with
// hg_scatter_planned.  (reference: the autograd backward of network_wtmk_tcnn.py:97-124 down to the index_add of hash_encoding.py)
NSIG_EXPORT int field_bwd_planned(const float *xyzs, uint32_t M, float bound, const float *grad_sigmas, const float *grad_rgbs, const float *sigmas,
                                  const float *rgbs, const uint32_t *masks, const void *packed, void *plan, nsig_stream_t stream) {
    if (M == 0) return NSIG_OK;
    NSIG_REQUIRE(xyzs && grad_sigmas && grad_rgbs && sigmas && rgbs && masks && packed && plan, "field_bwd_planned: null pointer");
    NSIG_REQUIRE(bound > 0.0f && (reinterpret_cast<uintptr_t>(plan) & 15) == 0 && M < (1u << 28), "field_bwd_planned: bound must be positive, plan 16-byte aligned, M < 2^28");
    if (mlp_precision() == 1 && bwd_pipelined())
        k_field_bwd_train<F16><<<field_grid(M), 256, F16::kBwdLds, as_stream(stream)>>>(xyzs, M, bound, grad_sigmas, grad_rgbs, sigmas, rgbs, masks,
                                                                                reinterpret_cast<const char *>(packed), scatter_plan_view(plan, M));
    else if (mlp_precision() == 1)
        k_field_bwd<F16, false><<<field_grid(M), 256, F16::kBwdLds, as_stream(stream)>>>(xyzs, M, bound, 1.0f / kCodebookResolution, grad_sigmas, grad_rgbs, sigmas, rgbs, masks,
                                                                                 reinterpret_cast<const char *>(packed), nullptr, nullptr, nullptr, GradTrace{}, 0,
                                                                                 scatter_plan_view(plan, M));
    else
        k_field_bwd<Bf16x3, false><<<field_grid(M), 256, Bf16x3::kBwdLds, as_stream(stream)>>>(xyzs, M, bound, 1.0f / kCodebookResolution, grad_sigmas, grad_rgbs, sigmas, rgbs,
                                                                                       masks, reinterpret_cast<const char *>(packed), nullptr, nullptr, nullptr, GradTrace{}, 0,
                                                                                       scatter_plan_view(plan, M));
    return check_launch("field_bwd_planned");
}

// ----------------------------------------------------------------------------- stage-1 (clean model) training entry points

static int fwd_trace_impl(const float *xyzs, const float *dirs, uint32_t M, const uint32_t *rows_dev, float bound, const float *const *base_tables_host,
                          const void *packed, const void *planes, float *sigmas, float *rgbs, uint32_t *masks, float *act_hs,
                          float *act_cin, float *act_h1, float *act_h2, nsig_stream_t stream, bool trace_f16 = false) {
    if (M == 0) return NSIG_OK;
    NSIG_REQUIRE(xyzs && dirs && packed && planes && sigmas && rgbs && masks && act_hs && act_cin && act_h1 && act_h2, "field_fwd_trace: null pointer");
    NSIG_REQUIRE(bound > 0.0f, "field_fwd_trace: bound must be positive");
    NSIG_REQUIRE(M <= (1u << 26), "field_fwd_trace: M=%u out of range (<= 2^26: the lane part of a trace address is a 32-bit byte offset of up to 32 x stride)", M);
    TablePtrs base{};
    if (int e = fill_base_tables(base_tables_host, base, "field_fwd_trace")) return e;
    const uint32_t stride = ceil_div(M, 32u) * 32u;
    ActTrace tr{act_hs, act_cin, act_h1, act_h2};
    if (trace_f16)      // (the ActTrace pointers address _Float16 rows)
        k_field_fwd_trace<_Float16><<<field_grid(M), 256, Bf16x3::kFwdLds, as_stream(stream)>>>(dirs, M, reinterpret_cast<const float2 *>(planes), stride,
                                                                                        reinterpret_cast<const char *>(packed), sigmas, rgbs, masks, tr, rows_dev);
    else if (!fwd_pipelined())      // (mlp_set_pipelined bit 0 clear: the generic kernel's trace variant, the cross-check of tests/test_gpu_stage1.py)
        k_field_fwd<Bf16x3, 1, true><<<field_grid(M), 256, Bf16x3::kFwdLds, as_stream(stream)>>>(xyzs, dirs, M, bound, base, make_level_geom(), nullptr,
                                                                                         reinterpret_cast<const float2 *>(planes), stride,
                                                                                         reinterpret_cast<const char *>(packed), sigmas, rgbs, nullptr, masks, tr,
                                                                                         rows_dev);
    else
        k_field_fwd_trace<float><<<field_grid(M), 256, Bf16x3::kFwdLds, as_stream(stream)>>>(dirs, M, reinterpret_cast<const float2 *>(planes), stride,
                                                                                     reinterpret_cast<const char *>(packed), sigmas, rgbs, masks, tr, rows_dev);
    return check_launch("field_fwd_trace");
}

NSIG_EXPORT int field_fwd_trace(const float *xyzs, const float *dirs, uint32_t M, float bound, const float *const *base_tables_host,
                                const void *packed, const void *planes, float *sigmas, float *rgbs, uint32_t *masks, float *act_hs,
                                float *act_cin, float *act_h1, float *act_h2, nsig_stream_t stream) {
    return fwd_trace_impl(xyzs, dirs, M, nullptr, bound, base_tables_host, packed, planes, sigmas, rgbs, masks, act_hs, act_cin, act_h1, act_h2, stream);
}

NSIG_EXPORT int field_fwd_trace_f16(const float *xyzs, const float *dirs, uint32_t M_capacity, const uint32_t *rows_dev, float bound,
                                    const float *const *base_tables_host, const void *packed, const void *planes, float *sigmas, float *rgbs, uint32_t *masks,
                                    void *act_hs, void *act_cin, void *act_h1, void *act_h2, nsig_stream_t stream) {
    return fwd_trace_impl(xyzs, dirs, M_capacity, rows_dev, bound, base_tables_host, packed, planes, sigmas, rgbs, masks, reinterpret_cast<float *>(act_hs),
                          reinterpret_cast<float *>(act_cin), reinterpret_cast<float *>(act_h1), reinterpret_cast<float *>(act_h2), stream, true);
}

NSIG_EXPORT int field_fwd_trace_rows(const float *xyzs, const float *dirs, uint32_t M_capacity, const uint32_t *rows_dev, float bound,
                                     const float *const *base_tables_host, const void *packed, const void *planes, float *sigmas, float *rgbs,
                                     uint32_t *masks, float *act_hs, float *act_cin, float *act_h1, float *act_h2, nsig_stream_t stream) {
    NSIG_REQUIRE(rows_dev != nullptr, "field_fwd_trace_rows: null row count");
    return fwd_trace_impl(xyzs, dirs, M_capacity, rows_dev, bound, base_tables_host, packed, planes, sigmas, rgbs, masks, act_hs, act_cin, act_h1, act_h2, stream);
}

static int bwd_trace_impl(uint32_t M, const uint32_t *rows_dev, const float *grad_sigmas, const float *grad_rgbs, const float *sigmas, const float *rgbs,
                          const uint32_t *masks, const void *packed, float *d_hs, float *d_so, float *d_h1, float *d_h2, float *d_out,
                          void *d_planes, nsig_stream_t stream) {
    if (M == 0) return NSIG_OK;
    NSIG_REQUIRE(grad_sigmas && grad_rgbs && sigmas && rgbs && masks && packed && d_hs && d_so && d_h1 && d_h2 && d_out && d_planes,
                 "field_bwd_trace: null pointer");
    NSIG_REQUIRE(M <= (1u << 26), "field_bwd_trace: M=%u out of range (<= 2^26: the lane part of a trace address is a 32-bit byte offset of up to 32 x stride)", M);
    const uint32_t stride = ceil_div(M, 32u) * 32u;
    GradTrace gt{d_hs, d_h1, d_h2, d_so, d_out, reinterpret_cast<float2 *>(d_planes)};
    k_field_bwd<Bf16x3, true><<<field_grid(M), 256, Bf16x3::kBwdLds, as_stream(stream)>>>(nullptr, M, 1.0f, 1.0f / kCodebookResolution, grad_sigmas, grad_rgbs, sigmas,
                                                                               rgbs, masks, reinterpret_cast<const char *>(packed), nullptr, nullptr,
                                                                               nullptr, gt, stride, ScatterPlan{}, rows_dev);
    return check_launch("field_bwd_trace");
}

NSIG_EXPORT int field_bwd_trace(uint32_t M, const float *grad_sigmas, const float *grad_rgbs, const float *sigmas, const float *rgbs,
                                const uint32_t *masks, const void *packed, float *d_hs, float *d_so, float *d_h1, float *d_h2, float *d_out,
                                void *d_planes, nsig_stream_t stream) {
    return bwd_trace_impl(M, nullptr, grad_sigmas, grad_rgbs, sigmas, rgbs, masks, packed, d_hs, d_so, d_h1, d_h2, d_out, d_planes, stream);
}

NSIG_EXPORT int field_bwd_trace_rows(uint32_t M_capacity, const uint32_t *rows_dev, const float *grad_sigmas, const float *grad_rgbs, const float *sigmas,
                                     const float *rgbs, const uint32_t *masks, const void *packed, float *d_hs, float *d_so, float *d_h1, float *d_h2,
                                     float *d_out, void *d_planes, nsig_stream_t stream) {
    NSIG_REQUIRE(rows_dev != nullptr, "field_bwd_trace_rows: null row count");
    return bwd_trace_impl(M_capacity, rows_dev, grad_sigmas, grad_rgbs, sigmas, rgbs, masks, packed, d_hs, d_so, d_h1, d_h2, d_out, d_planes, stream);
}
