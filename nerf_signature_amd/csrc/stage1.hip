// Stage-1 (clean model) training kernels for gfx950, SURVEY.md 8(f) N3: what the watermark stage never needs because its MLPs and base
// tables are frozen (nerf/network_wtmk_tcnn.py:90-95) and what the clean model of /root/reference/nerf/network_hash.py trains (:154-166):
//
//   * field_wgrad    the five weight-gradient reductions dW = sum over points of (pre-activation gradient) x (layer input)^T.  The reference
//                    leaves them to tiny-cuda-nn's CUTLASS GEMMs; a BLAS library handles their shape (64 x 64 outputs, K = 10^5..10^6
//                    points) with one or two workgroups: 5 x 145 us of a 4096-ray step at 140 k points, more than everything else together.
//                    Here: split-K over the points on MFMA, tiles staged through LDS, slabs of partial sums, one fixed-order reduction --
//                    bit-reproducible.
//   * clean_loss     the MSE of nerf/utils.py:503 and its gradient in one launch.
//
// The level scatter of the base-table gradients lives beside its siblings in hashgrid.hip (hg_levels_plan / hg_levels_scatter).
#include "hashgrid.h"
#include "mfma.h"

namespace nsig {

// ----------------------------------------------------------------------------- weight gradients
//
// Both factors are stored feature-major, [width][stride] fp32 (field_fwd_trace / field_bwd_trace): a row is one feature over all points, and the
// K dimension of the products runs over the POINTS.  An operand of v_mfma_f32_32x32x16 wants lane (r, h) to hold row r, k = 8h..8h+7: 32
// contiguous bytes of row r.  Loaded straight from memory that way (the first version of this kernel), every 16-byte load instruction of a wave
// touches 32 different 128-byte lines -- 32 rows x 2 halves -- and the kernel ran at 2.3-2.7 TB/s whatever its occupancy (84 -> 336 workgroups per
// role: 546 -> 483 us at 675 k points, profiles/r05_stage1_wgrad_ab.txt): bound by the address path, not by latency.  Here a workgroup stages a TILE
// of 32 points x all rows of its role through LDS: 8 consecutive lanes read one whole 128-byte line (8 lines per instruction instead of 32), rows
// are padded to 36 floats so that the operand reads (two ds_read_b128 per lane) are conflict-free, and the loads of the next two tiles are in flight
// (in registers) while the current one is multiplied.
// Operands enter as split bf16 (hi + lo, three MFMAs per product, fp32 accumulate): the pre-activation gradients of an unscaled MSE loss sit
// around 1e-6 and would need a loss scale in fp16; bf16 has fp32's exponent range.
//
// Work split: three ROLES (blockIdx.y), four 32 x 32 products each -- ONE PER WAVE, accumulated over all tiles of the workgroup:
//   role 0: dW1s[64x32] = d_hs x feat^T (waves 0, 1), dWc1[64x32] = d_h1 x cin^T (2, 3)
//   role 1: dW2s[16x64] = d_so x hs^T   (0, 1),       dWc3[16x64] = d_out x h2^T (2, 3)      (the A rows 16..31 are zero)
//   role 2: dWc2[64x64] = d_h2 x h1^T   (0..3)
// A workgroup stores ONE slab of 4096 partial sums in accumulator order (wave q = product q); k_wgrad_reduce adds the slabs in workgroup order
// and writes tcnn's layout ([out][in] row-major).  Fixed tile -> workgroup assignment, fixed order: bit-reproducible.
constexpr uint32_t kWgradRoles = 3, kWgradSlab = 4u * 16u * 64u;   // floats per (workgroup, role): 4 products x 16 registers x 64 lanes
#ifndef NSIG_WGRAD_WGS
#define NSIG_WGRAD_WGS 84
#endif
constexpr uint32_t kWgradMaxWGs = NSIG_WGRAD_WGS;      // x 3 roles = 252 workgroups of 4 waves: ONE per compute unit, 27 KiB of LDS -- the kernel runs beside the table
                                                       // scatter (1024-thread workgroups with 64-128 KiB of LDS), which must still fit on the same units
constexpr uint32_t kWgradRowFloats = 36;               // 32 points + 4 floats of padding: consecutive rows start 4 banks apart (16 bytes), 8 lanes x 16 B cover all 32 banks
constexpr uint32_t kWgradRows = 192;                   // role 0 stages the most rows (64 + 64 + 32 + 32)
#ifndef NSIG_WGRAD_AHEAD
#define NSIG_WGRAD_AHEAD 4
#endif
constexpr int kWgradAhead = NSIG_WGRAD_AHEAD;             // tiles whose loads are in flight (registers) while one is multiplied

struct WgradArgs {
    const float2 *planes;                               // [16][stride] float2: encoder features 2l, 2l+1 of level l
    const float *hs, *cin, *h1, *h2;                    // layer inputs
    const float *d_hs, *d_so, *d_h1, *d_h2, *d_out;     // pre-activation gradients
};

// (split8 / mac3: mfma.h)

// What one thread fetches of a tile: up to six 16-byte pieces.  Thread t = 8 g + c takes chunk c (points 4c..4c+3 of the tile) of rows g and g + 32 of every
// 64-row factor and of row g of every 32-row one; the feature planes are float2 pairs (feature 2l, 2l+1 of level l): thread t = 16 l + c2 takes points
// 2 c2, 2 c2 + 1 of level l.  LDS rows per role:  0: d_hs 0..63 | d_h1 64..127 | cin 128..159 | feat 160..191;  1: hs 0..63 | h2 64..127 | d_so 128..143 |
// d_out 144..159;  2: d_h2 0..63 | h1 64..127.
struct WgradTile {
    float4 v[6];
};

template <int role>      // (a compile-time role: with the three cases as run-time branches inside the tile loop the compiler's wait-count bookkeeping drained every load
                         //  -- the tiles requested ahead included -- in front of each staging pass)
__device__ inline void wgrad_request(const WgradArgs &a, uint32_t stride, uint32_t tile, WgradTile &t) {
    const uint32_t g = threadIdx.x >> 3, c = threadIdx.x & 7u;
    const size_t at = (size_t)tile * 32u + 4u * c;
    auto row = [&](const float *base, uint32_t r) { return *reinterpret_cast<const float4 *>(base + (size_t)r * stride + at); };
    if (role == 0) {
        t.v[0] = row(a.d_hs, g); t.v[1] = row(a.d_hs, g + 32u); t.v[2] = row(a.d_h1, g); t.v[3] = row(a.d_h1, g + 32u); t.v[4] = row(a.cin, g);
        t.v[5] = *reinterpret_cast<const float4 *>(a.planes + (size_t)(threadIdx.x >> 4) * stride + (size_t)tile * 32u + 2u * (threadIdx.x & 15u));
    } else if (role == 1) {
        t.v[0] = row(a.hs, g); t.v[1] = row(a.hs, g + 32u); t.v[2] = row(a.h2, g); t.v[3] = row(a.h2, g + 32u);
        t.v[4] = g < 16u ? row(a.d_so, g) : row(a.d_out, g - 16u);
    } else {
        t.v[0] = row(a.d_h2, g); t.v[1] = row(a.d_h2, g + 32u); t.v[2] = row(a.h1, g); t.v[3] = row(a.h1, g + 32u);
    }
}

// ... into the staging area; points at or beyond n (stale rows of buffers sized for more points) enter as zeros
template <int role>
__device__ inline void wgrad_stage(float *__restrict__ lds, uint32_t tile, uint32_t n, const WgradTile &t) {
    const uint32_t g = threadIdx.x >> 3, c = threadIdx.x & 7u;
    const uint32_t p = tile * 32u + 4u * c;
    auto put = [&](uint32_t r, float4 v) {
        v.x = p < n ? v.x : 0.0f; v.y = p + 1u < n ? v.y : 0.0f; v.z = p + 2u < n ? v.z : 0.0f; v.w = p + 3u < n ? v.w : 0.0f;
        *reinterpret_cast<float4 *>(lds + r * kWgradRowFloats + 4u * c) = v;
    };
    put(g, t.v[0]); put(g + 32u, t.v[1]); put(g + 64u, t.v[2]); put(g + 96u, t.v[3]);
    if (role == 0) {
        put(g + 128u, t.v[4]);
        const uint32_t l = threadIdx.x >> 4, c2 = threadIdx.x & 15u, q = tile * 32u + 2u * c2;
        const float4 v = t.v[5];
        const bool in0 = q < n, in1 = q + 1u < n;
        *reinterpret_cast<float2 *>(lds + (160u + 2u * l) * kWgradRowFloats + 2u * c2) = make_float2(in0 ? v.x : 0.0f, in1 ? v.z : 0.0f);
        *reinterpret_cast<float2 *>(lds + (161u + 2u * l) * kWgradRowFloats + 2u * c2) = make_float2(in0 ? v.y : 0.0f, in1 ? v.w : 0.0f);
    } else if (role == 1) {
        put(g + 128u, t.v[4]);
    }
}

__device__ inline void wgrad_operand(const float *__restrict__ lds, uint32_t row, bool exists, uint32_t col, Split8 &s) {
    float v[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    if (exists) {
        const float4 x = *reinterpret_cast<const float4 *>(lds + row * kWgradRowFloats + col), y = *reinterpret_cast<const float4 *>(lds + row * kWgradRowFloats + col + 4u);
        v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w; v[4] = y.x; v[5] = y.y; v[6] = y.z; v[7] = y.w;
    }
    split8(v, s);
}

template <int role>
__device__ inline void wgrad_role(float *__restrict__ stage, const WgradArgs &a, uint32_t stride, uint32_t n, float *__restrict__ slabs) {
    const uint32_t lane = threadIdx.x & 63u, q = threadIdx.x >> 6, r = lane & 31u, h = lane >> 5;
    // this wave's product: rows of its A and B factor in the staging area
    uint32_t a_row, b_row;
    bool a_exists = true;
    if (role == 0) { a_row = 32u * q + r; b_row = (q < 2u ? 160u : 128u) + r; }
    else if (role == 1) { a_row = (q < 2u ? 128u : 144u) + r; a_exists = r < 16u; b_row = 32u * q + r; }
    else { a_row = 32u * (q >> 1) + r; b_row = 64u + 32u * (q & 1u) + r; }
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
    const uint32_t n_tiles = ceil_div(n, 32u), step = gridDim.x;
    // Requests are UNCONDITIONAL (past the last tile: the last tile again, never staged): a request skipped on some path leaves the compiler's wait counts with
    // a path on which fewer loads are outstanding, and it then drains everything in front of each staging pass.
    WgradTile ahead[kWgradAhead];
    if (n_tiles != 0) {
#pragma unroll
        for (int k = 0; k < kWgradAhead; ++k) wgrad_request<role>(a, stride, min(blockIdx.x + k * step, n_tiles - 1u), ahead[k]);
    }
    for (uint32_t t0 = blockIdx.x; t0 < n_tiles; t0 += kWgradAhead * step) {
#pragma unroll
        for (int k = 0; k < kWgradAhead; ++k) {      // (unrolled: which register set holds which tile is known at compile time)
            const uint32_t tile = t0 + k * step;
            if (tile >= n_tiles) break;              // (uniform)
            __syncthreads();                         // every wave is done with the previous tile's operands
            wgrad_stage<role>(stage, tile, n, ahead[k]);
            wgrad_request<role>(a, stride, min(tile + kWgradAhead * step, n_tiles - 1u), ahead[k]);
            __syncthreads();
#pragma unroll
            for (uint32_t u = 0; u < 2u; ++u) {
                Split8 A, B;
                wgrad_operand(stage, a_row, a_exists, 16u * u + 8u * h, A);
                wgrad_operand(stage, b_row, true, 16u * u + 8u * h, B);
                acc = mac3(A, B, acc);
            }
        }
    }
    float *__restrict__ out = slabs + ((size_t)blockIdx.x * kWgradRoles + role) * kWgradSlab + (size_t)q * 1024u;
#pragma unroll
    for (int e = 0; e < 16; ++e) out[e * 64 + lane] = acc[e];
}

__global__ void __launch_bounds__(256) k_field_wgrad(WgradArgs a, uint32_t stride, uint32_t M, const uint32_t *__restrict__ rows_dev, float *__restrict__ slabs) {
    __shared__ __attribute__((aligned(16))) float stage[kWgradRows * kWgradRowFloats];      // 27 KiB
    const uint32_t n = rows_dev != nullptr ? min(M, *rows_dev) : M;
    if (blockIdx.y == 0) wgrad_role<0>(stage, a, stride, n, slabs);
    else if (blockIdx.y == 1) wgrad_role<1>(stage, a, stride, n, slabs);
    else wgrad_role<2>(stage, a, stride, n, slabs);
}

// slab sums in workgroup order -> sigma_net.params' gradient [3072] = [W1s 64x32 | W2s 16x64], color_net.params' [7168] = [Wc1 64x32 | Wc2 64x64 | Wc3 16x64]
__global__ void __launch_bounds__(256) k_wgrad_reduce(const float *__restrict__ slabs, uint32_t n_wg, float *__restrict__ g_sigma, float *__restrict__ g_color) {
    // 64 elements per workgroup; its four waves each add a quarter of the slabs (sixteen in flight per thread: a chain of one dependent load per slab took 60 us
    // for 128 slabs, one wave walking all 256 still 13 us -- this launch sits between the backward and the scatter), then the four partial sums are combined through
    // LDS in wave order.  Every order fixed: the same bits every run.
    __shared__ float quarter[4][64];
    const uint32_t grp = threadIdx.x >> 6, j = threadIdx.x & 63u;
    const uint32_t e = blockIdx.x * 64u + j;      // (role, product, register, lane); the grid covers kWgradRoles * kWgradSlab exactly
    const uint32_t role = e / kWgradSlab, i = e % kWgradSlab;
    const uint32_t per = ceil_div(n_wg, 4u), w_begin = min(n_wg, grp * per), w_end = min(n_wg, w_begin + per);
    float part[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) part[u] = 0.0f;
    for (uint32_t w0 = w_begin; w0 < w_end; w0 += 16u) {
        float v[16];
#pragma unroll
        for (uint32_t u = 0; u < 16u; ++u) v[u] = w0 + u < w_end ? slabs[((size_t)(w0 + u) * kWgradRoles + role) * kWgradSlab + i] : 0.0f;
#pragma unroll
        for (int u = 0; u < 16; ++u) part[u] += v[u];
    }
    float s = 0.0f;
#pragma unroll
    for (int u = 0; u < 16; ++u) s += part[u];
    quarter[grp][j] = s;
    __syncthreads();
    if (grp != 0) return;
    s = ((quarter[0][j] + quarter[1][j]) + quarter[2][j]) + quarter[3][j];
    const uint32_t q = i >> 10, reg = (i >> 6) & 15u, lane = i & 63u;
    const uint32_t row = (uint32_t)row_of_reg16((int)(lane >> 5), (int)reg), col = lane & 31u;
    if (role == 0) {
        float *dst = q < 2 ? g_sigma : g_color;              // W1s / Wc1: [64][32]
        dst[(32u * (q & 1u) + row) * 32u + col] = s;
    } else if (role == 1) {
        if (row >= 16u) return;
        float *dst = q < 2 ? g_sigma + 2048 : g_color + 6144;   // W2s / Wc3: [16][64]
        dst[row * 64u + 32u * (q & 1u) + col] = s;
    } else {
        g_color[2048u + (32u * (q >> 1) + row) * 64u + 32u * (q & 1u) + col] = s;   // Wc2: [64][64]
    }
}

// ----------------------------------------------------------------------------- loss
// nerf/utils.py:503: loss = MSE(pred_rgb, gt_rgb).mean(-1).mean() = the mean over all N * 3 elements; g_image = grad_scale * dloss/dimage.
// The same single workgroup keeps a captured loop's books (every pointer optional): the march's (points, rays) totals into row step % 16 of a ring
// -- what NeRFRenderer.step_counter holds for update_extra_state's mean_count (renderer_wtmk.py:282-284,533-536) --, the loss into a ring
// the host reads whenever it likes, and the step count itself, advanced here: nothing of a step's tail reads it, the next replay's head does.
__device__ inline uint64_t mix64(uint64_t z) {      // splitmix64's finaliser
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ void __launch_bounds__(1024) k_clean_loss(const float *__restrict__ image, const float *__restrict__ gt, uint32_t n, float grad_scale,
                                                     float *__restrict__ loss, float *__restrict__ g_image, uint32_t *__restrict__ step_dev,
                                                     const int32_t *__restrict__ march_counter, int32_t *__restrict__ count_ring,
                                                     float *__restrict__ loss_ring, uint32_t loss_ring_len, float *__restrict__ noise_next,
                                                     uint32_t n_noise, uint64_t seed) {
    __shared__ float scratch[16];
    // the NEXT step's per-ray start offsets (perturb=True: `noises = torch.rand(N)`, raymarching.py:213): U[0, 1) as a pure function of
    // (seed, step + 1, ray) -- a replayed graph draws fresh values without a host-side generator (torch.rand under capture costs two
    // fill launches per replay for its seed / offset tensors).  This step's march has long consumed its own values.
    if (noise_next != nullptr) {
        const uint64_t key = mix64(seed ^ (0x9E3779B97F4A7C15ull * ((uint64_t)(step_dev != nullptr ? *step_dev : 0u) + 2ull)));
        for (uint32_t i = threadIdx.x; i < n_noise; i += 1024u)
            noise_next[i] = (float)(uint32_t)(mix64(key + 0xD1B54A32D192ED03ull * ((uint64_t)i + 1ull)) >> 40) * (1.0f / 16777216.0f);
    }
    __syncthreads();      // (every thread has read the step count before thread 0 advances it)
    float s = 0.0f;
    const float k = grad_scale * 2.0f / (float)n;
    for (uint32_t i = threadIdx.x; i < n; i += 1024u) {
        const float d = image[i] - gt[i];
        s += d * d;
        g_image[i] = k * d;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
    if ((threadIdx.x & 63u) == 0) scratch[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.0f;
        for (int w = 0; w < 16; ++w) t += scratch[w];
        t = t / (float)n;
        loss[0] = t;
        const uint32_t step = step_dev != nullptr ? *step_dev : 0u;
        if (count_ring != nullptr && march_counter != nullptr) {
            count_ring[2u * (step % 16u)] = march_counter[0];
            count_ring[2u * (step % 16u) + 1u] = march_counter[1];
        }
        if (loss_ring != nullptr && loss_ring_len > 0) loss_ring[step % loss_ring_len] = t;
        if (step_dev != nullptr) *step_dev = step + 1u;
    }
}

// ----------------------------------------------------------------------------- EMA of the parameters
// The stage-1 trainer keeps an exponential moving average of every parameter (main_nerf.py:130 `ema_decay=0.95`; utils.py:389-390 torch_ema's
// ExponentialMovingAverage, updated after every optimiser step, :761-762/:892-893) and evaluates / checkpoints with it (:801-811).  torch_ema 0.3's update(), with
// its default warm-up (use_num_updates): num_updates += 1; decay = min(decay, (1 + num_updates) / (10 + num_updates)); then for every parameter
//     tmp = shadow - param;  tmp *= (1 - decay);  shadow -= tmp
// -- here one launch over all tensors, the update count read from the captured loop's device step counter (which the step's loss kernel has already advanced).
constexpr int kEmaMax = 32;
constexpr uint32_t kEmaChunk = 4096;      // elements per workgroup
struct DenseEma {
    const float *p[kEmaMax];
    float *s[kEmaMax];
    uint32_t numel[kEmaMax], chunk0[kEmaMax + 1];
};
__global__ void __launch_bounds__(256) k_ema_dense(DenseEma a, uint32_t n, const uint32_t *__restrict__ num_updates, double decay) {
    uint32_t t = 0;
    while (t + 1 < n && blockIdx.x >= a.chunk0[t + 1]) ++t;      // (uniform)
    const double nu = (double)*num_updates;
    const float w = (float)(1.0 - fmin(decay, (1.0 + nu) / (10.0 + nu)));      // (python: a double, handed to mul_ as a float32 scalar)
    const uint32_t first = (blockIdx.x - a.chunk0[t]) * kEmaChunk, numel = a.numel[t];
    const float *__restrict__ p = a.p[t];
    float *__restrict__ s = a.s[t];
    if (numel % 4u == 0 && ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(s)) & 15) == 0) {      // (uniform)
        for (uint32_t i = first + 4u * threadIdx.x; i < min(numel, first + kEmaChunk); i += 1024u) {
            const float4 pv = *reinterpret_cast<const float4 *>(p + i);
            float4 sv = *reinterpret_cast<const float4 *>(s + i);
            sv.x = sv.x - (sv.x - pv.x) * w; sv.y = sv.y - (sv.y - pv.y) * w; sv.z = sv.z - (sv.z - pv.z) * w; sv.w = sv.w - (sv.w - pv.w) * w;
            *reinterpret_cast<float4 *>(s + i) = sv;
        }
    } else {
        for (uint32_t i = first + threadIdx.x; i < min(numel, first + kEmaChunk); i += 256u) s[i] = s[i] - (s[i] - p[i]) * w;
    }
}

}  // namespace nsig

using namespace nsig;

NSIG_EXPORT int opt_ema_update(uint32_t n, const float *const *params_host, float *const *shadow_host, const uint32_t *numel_host, const uint32_t *num_updates,
                               double decay, nsig_stream_t stream) {
    NSIG_REQUIRE(params_host && shadow_host && numel_host && num_updates, "opt_ema_update: null pointer");
    NSIG_REQUIRE(n >= 1 && n <= (uint32_t)kEmaMax && decay >= 0.0 && decay <= 1.0, "opt_ema_update: 1 .. %d tensors, decay in [0, 1]", kEmaMax);
    DenseEma a{};
    uint32_t chunks = 0;
    for (uint32_t i = 0; i < n; ++i) {
        NSIG_REQUIRE(params_host[i] && shadow_host[i] && numel_host[i] > 0, "opt_ema_update: tensor %u has a null pointer or no elements", i);
        a.p[i] = params_host[i]; a.s[i] = shadow_host[i]; a.numel[i] = numel_host[i];
        a.chunk0[i] = chunks;
        chunks += ceil_div(numel_host[i], kEmaChunk);
    }
    a.chunk0[n] = chunks;
    k_ema_dense<<<chunks, 256, 0, as_stream(stream)>>>(a, n, num_updates, decay);
    return check_launch("opt_ema_update");
}

// slabs[n_wg][3 roles][4 products][16 registers][64 lanes] -> the two parameter gradients (also the tail of field_bwd_wgrad, stage1_fused.hip)
int nsig::wgrad_reduce_launch(const float *slabs, uint32_t n_wg, float *grad_sigma_params, float *grad_color_params, hipStream_t st, const char *what) {
    static_assert((kWgradRoles * kWgradSlab) % 64u == 0, "k_wgrad_reduce: 64 elements per workgroup");
    k_wgrad_reduce<<<kWgradRoles * kWgradSlab / 64u, 256, 0, st>>>(slabs, n_wg, grad_sigma_params, grad_color_params);
    return check_launch(what);
}

static uint32_t wgrad_workgroups(uint32_t M) {      // pairs of K-steps (32 points), 4 waves per workgroup, >= 2 pairs per wave where there is work
    const uint32_t want = ceil_div(ceil_div(M, 32u), 8u);
    return want < 32u ? 32u : (want > kWgradMaxWGs ? kWgradMaxWGs : want);
}

NSIG_EXPORT size_t field_wgrad_scratch_bytes(uint32_t M) { return (size_t)wgrad_workgroups(M) * kWgradRoles * kWgradSlab * sizeof(float); }

NSIG_EXPORT int field_wgrad(uint32_t M, const uint32_t *rows_dev, const void *planes, const float *act_hs, const float *act_cin, const float *act_h1,
                            const float *act_h2, const float *d_hs, const float *d_so, const float *d_h1, const float *d_h2, const float *d_out,
                            void *scratch, float *grad_sigma_params, float *grad_color_params, nsig_stream_t stream) {
    NSIG_REQUIRE(planes && act_hs && act_cin && act_h1 && act_h2 && d_hs && d_so && d_h1 && d_h2 && d_out && scratch && grad_sigma_params && grad_color_params,
                 "field_wgrad: null pointer");
    NSIG_REQUIRE(M >= 1 && M < (1u << 28), "field_wgrad: M=%u out of range", M);
    const void *all[] = {planes, act_hs, act_cin, act_h1, act_h2, d_hs, d_so, d_h1, d_h2, d_out, scratch};
    for (const void *p : all) NSIG_REQUIRE((reinterpret_cast<uintptr_t>(p) & 15) == 0, "field_wgrad: every buffer must be 16-byte aligned");
    const uint32_t stride = ceil_div(M, 32u) * 32u, n_wg = wgrad_workgroups(M);
    WgradArgs a{reinterpret_cast<const float2 *>(planes), act_hs, act_cin, act_h1, act_h2, d_hs, d_so, d_h1, d_h2, d_out};
    hipStream_t st = as_stream(stream);
    k_field_wgrad<<<dim3(n_wg, kWgradRoles), 256, 0, st>>>(a, stride, M, rows_dev, reinterpret_cast<float *>(scratch));
    if (int e = check_launch("field_wgrad")) return e;
    return wgrad_reduce_launch(reinterpret_cast<const float *>(scratch), n_wg, grad_sigma_params, grad_color_params, st, "field_wgrad (reduce)");
}

NSIG_EXPORT int clean_loss(const float *image, const float *gt, uint32_t n_values, float grad_scale, float *loss, float *grad_image, uint32_t *step_dev,
                           const int32_t *march_counter, int32_t *count_ring, float *loss_ring, uint32_t loss_ring_len, float *noise_next,
                           uint32_t n_noise, uint64_t seed, nsig_stream_t stream) {
    NSIG_REQUIRE(image && gt && loss && grad_image, "clean_loss: null pointer");
    NSIG_REQUIRE(n_values >= 1, "clean_loss: empty input");
    k_clean_loss<<<1, 1024, 0, as_stream(stream)>>>(image, gt, n_values, grad_scale, loss, grad_image, step_dev, march_counter, count_ring, loss_ring, loss_ring_len,
                                                    noise_next, n_noise, seed);
    return check_launch("clean_loss");
}
