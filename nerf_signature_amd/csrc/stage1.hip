// Stage-1 (clean model) training kernels for gfx950, SURVEY.md 8(f) N3: what the watermark stage never needs because its MLPs and base
// tables are frozen (nerf/network_wtmk_tcnn.py:90-95) and what the clean model of /root/reference/nerf/network_hash.py trains (:154-166):
//
//   * field_wgrad    the five weight-gradient reductions dW = sum over points of (pre-activation gradient) x (layer input)^T.  The reference
//                    leaves them to tiny-cuda-nn's CUTLASS GEMMs; a BLAS library handles their shape (64 x 64 outputs, K = 10^5..10^6
//                    points) with one or two workgroups: 5 x 145 us of a 4096-ray step, more than everything else together
//                    (profiles/r05_stage1_baseline_kernels.txt).  Here: split-K over the points on MFMA, slabs of partial sums, one
//                    fixed-order reduction -- bit-reproducible.
//   * clean_loss     the MSE of nerf/utils.py:503 and its gradient in one launch.
//
// The level scatter of the base-table gradients lives beside its siblings in hashgrid.hip (hg_levels_plan / hg_levels_scatter).
#include "hashgrid.h"
#include "mfma.h"

namespace nsig {

// ----------------------------------------------------------------------------- weight gradients
//
// Both factors are stored feature-major, [width][stride] fp32 (field_fwd_trace / field_bwd_trace), so 8 consecutive points of one row are
// 32 contiguous bytes: exactly one lane's share of an A or B operand of v_mfma_f32_32x32x16 when the K dimension runs over the POINTS
// (lane (r, h) holds row r, k = 8h..8h+7).  A 16-point K-step of a 32 x 32 output block is therefore two plain 32-byte loads per lane and no
// transposition.  Operands enter as split bf16 (hi + lo, three MFMAs per product, fp32 accumulate): the pre-activation gradients of an
// unscaled MSE loss sit around 1e-6 and would need a loss scale in fp16; bf16 has fp32's exponent range.
//
// Work split: three ROLES (blockIdx.y), four 32 x 32 products each, so that a wave keeps 64 accumulator registers and every stored row is
// read by exactly one role:
//   role 0: dW1s[64x32] = d_hs x feat^T (2 products), dWc1[64x32] = d_h1 x cin^T (2)
//   role 1: dW2s[16x64] = d_so x hs^T   (2),          dWc3[16x64] = d_out x h2^T (2)      (the A rows 16..31 are zero)
//   role 2: dWc2[64x64] = d_h2 x h1^T   (4)
// Waves stride over the K-steps; a workgroup's four waves are summed through LDS and the workgroup stores ONE slab of 4096 partial sums
// in accumulator order; k_wgrad_reduce adds the slabs in workgroup order and writes tcnn's layout ([out][in] row-major).
constexpr uint32_t kWgradRoles = 3, kWgradSlab = 4u * 16u * 64u;   // floats per (workgroup, role): 4 products x 16 registers x 64 lanes
#ifndef NSIG_WGRAD_WGS
#define NSIG_WGRAD_WGS 84
#endif
constexpr uint32_t kWgradMaxWGs = NSIG_WGRAD_WGS;      // x 3 roles = 252 workgroups of 4 waves: ONE per compute unit, one wave per SIMD, 16 KiB of LDS -- the kernel runs
                                           // beside the table scatter (1024-thread workgroups with 64-128 KiB of LDS), which must still fit on the same units

struct WgradArgs {
    const float2 *planes;                               // [16][stride] float2: encoder features 2l, 2l+1 of level l
    const float *hs, *cin, *h1, *h2;                    // layer inputs
    const float *d_hs, *d_so, *d_h1, *d_h2, *d_out;     // pre-activation gradients
};

__device__ inline void split8(const float (&v)[8], Split8 &s) {
#pragma unroll
    for (int jp = 0; jp < 4; ++jp) {
        const uint32_t hi = cvt_pk_bf16(v[2 * jp], v[2 * jp + 1]);
        s.hi[jp] = hi;
        s.lo[jp] = cvt_pk_bf16(v[2 * jp] - __uint_as_float(hi << 16), v[2 * jp + 1] - __uint_as_float(hi & 0xffff0000u));
    }
}

// points p0..p0+7 of one stored row; points >= n (stale rows of a buffer sized for more points) and rows that do not exist read as zero
struct Raw8 {
    float v[8];
};
__device__ inline void fetch_row8(const float *__restrict__ base, uint32_t stride, uint32_t row, uint32_t p0, uint32_t n, bool row_exists, Raw8 &r) {
#pragma unroll
    for (int j = 0; j < 8; ++j) r.v[j] = 0.0f;
    if (row_exists && p0 < n) {
        const float4 a = *reinterpret_cast<const float4 *>(base + (size_t)row * stride + p0);
        const float4 b = *reinterpret_cast<const float4 *>(base + (size_t)row * stride + p0 + 4);
        const float w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) r.v[j] = p0 + j < n ? w[j] : 0.0f;
    }
}

// ... of encoder feature i (component i & 1 of level i >> 1)
__device__ inline void fetch_feat8(const float2 *__restrict__ planes, uint32_t stride, uint32_t i, uint32_t p0, uint32_t n, Raw8 &r) {
#pragma unroll
    for (int j = 0; j < 8; ++j) r.v[j] = 0.0f;
    if (p0 < n) {
        const float4 *src = reinterpret_cast<const float4 *>(planes + (size_t)(i >> 1) * stride + p0);
        const float4 q[4] = {src[0], src[1], src[2], src[3]};
        const bool odd = i & 1u;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            r.v[2 * t] = p0 + 2 * t < n ? (odd ? q[t].y : q[t].x) : 0.0f;
            r.v[2 * t + 1] = p0 + 2 * t + 1 < n ? (odd ? q[t].w : q[t].z) : 0.0f;
        }
    }
}

__device__ inline f32x16 mac3(const Split8 &a, const Split8 &b, f32x16 c) {      // lo parts first, hi * hi last (as Bf16x3::mac)
    const bf16x8 a_hi = operand(a.hi), a_lo = operand(a.lo), b_hi = operand(b.hi), b_lo = operand(b.lo);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_lo, b_hi, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, b_lo, c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, b_hi, c, 0, 0, 0);
}

// One wave takes PAIRS of adjacent K-steps (32 points: each row's whole 128-byte line) and requests all their operands -- up to 24 32-byte
// loads per lane -- before it converts the first: the kernel runs at one wave per SIMD beside the table scatter, so the loads in flight per
// wave are what hides the memory latency.
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2)))
k_field_wgrad(WgradArgs a, uint32_t stride, uint32_t M, const uint32_t *__restrict__ rows_dev, float *__restrict__ slabs) {
    __shared__ float red[kWgradSlab];        // 16 KiB: waves 1..3 hand their sums to wave 0 one after the other
    const uint32_t n = rows_dev != nullptr ? min(M, *rows_dev) : M;
    const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6, r = lane & 31u, h = lane >> 5;
    const uint32_t role = blockIdx.y;
    f32x16 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[q][e] = 0.0f;
    const uint32_t n_pairs = ceil_div(n, 32u);
    for (uint32_t kp = blockIdx.x * 4u + wid; kp < n_pairs; kp += gridDim.x * 4u) {
        Raw8 ra[2][4], rb[2][4];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const uint32_t p0 = kp * 32u + 16u * u + 8u * h;
            if (role == 0) {
                fetch_row8(a.d_hs, stride, r, p0, n, true, ra[u][0]);
                fetch_row8(a.d_hs, stride, 32u + r, p0, n, true, ra[u][1]);
                fetch_row8(a.d_h1, stride, r, p0, n, true, ra[u][2]);
                fetch_row8(a.d_h1, stride, 32u + r, p0, n, true, ra[u][3]);
                fetch_feat8(a.planes, stride, r, p0, n, rb[u][0]);
                fetch_row8(a.cin, stride, r, p0, n, true, rb[u][1]);
            } else if (role == 1) {
                fetch_row8(a.d_so, stride, r, p0, n, r < 16u, ra[u][0]);
                fetch_row8(a.d_out, stride, r, p0, n, r < 16u, ra[u][1]);
                fetch_row8(a.hs, stride, r, p0, n, true, rb[u][0]);
                fetch_row8(a.hs, stride, 32u + r, p0, n, true, rb[u][1]);
                fetch_row8(a.h2, stride, r, p0, n, true, rb[u][2]);
                fetch_row8(a.h2, stride, 32u + r, p0, n, true, rb[u][3]);
            } else {
                fetch_row8(a.d_h2, stride, r, p0, n, true, ra[u][0]);
                fetch_row8(a.d_h2, stride, 32u + r, p0, n, true, ra[u][1]);
                fetch_row8(a.h1, stride, r, p0, n, true, rb[u][0]);
                fetch_row8(a.h1, stride, 32u + r, p0, n, true, rb[u][1]);
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (role == 0) {
                Split8 A0, A1, A2, A3, B0, B1;
                split8(ra[u][0].v, A0); split8(ra[u][1].v, A1); split8(ra[u][2].v, A2); split8(ra[u][3].v, A3);
                split8(rb[u][0].v, B0); split8(rb[u][1].v, B1);
                acc[0] = mac3(A0, B0, acc[0]);
                acc[1] = mac3(A1, B0, acc[1]);
                acc[2] = mac3(A2, B1, acc[2]);
                acc[3] = mac3(A3, B1, acc[3]);
            } else if (role == 1) {
                Split8 A0, A1, B0, B1, B2, B3;
                split8(ra[u][0].v, A0); split8(ra[u][1].v, A1);
                split8(rb[u][0].v, B0); split8(rb[u][1].v, B1); split8(rb[u][2].v, B2); split8(rb[u][3].v, B3);
                acc[0] = mac3(A0, B0, acc[0]);
                acc[1] = mac3(A0, B1, acc[1]);
                acc[2] = mac3(A1, B2, acc[2]);
                acc[3] = mac3(A1, B3, acc[3]);
            } else {
                Split8 A0, A1, B0, B1;
                split8(ra[u][0].v, A0); split8(ra[u][1].v, A1);
                split8(rb[u][0].v, B0); split8(rb[u][1].v, B1);
                acc[0] = mac3(A0, B0, acc[0]);
                acc[1] = mac3(A0, B1, acc[1]);
                acc[2] = mac3(A1, B0, acc[2]);
                acc[3] = mac3(A1, B1, acc[3]);
            }
        }
    }
    for (uint32_t w = 1; w < 4; ++w) {       // (three rounds at the end of a kernel that streams for tens of microseconds)
        if (wid == w) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int e = 0; e < 16; ++e) red[(q * 16 + e) * 64 + lane] = acc[q][e];
        }
        __syncthreads();
        if (wid == 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[q][e] += red[(q * 16 + e) * 64 + lane];
        }
        __syncthreads();
    }
    if (wid == 0) {
        float *__restrict__ out = slabs + ((size_t)blockIdx.x * kWgradRoles + role) * kWgradSlab;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 16; ++e) out[(q * 16 + e) * 64 + lane] = acc[q][e];
    }
}

// slab sums in workgroup order -> sigma_net.params' gradient [3072] = [W1s 64x32 | W2s 16x64], color_net.params' [7168] = [Wc1 64x32 | Wc2 64x64 | Wc3 16x64]
__global__ void __launch_bounds__(256) k_wgrad_reduce(const float *__restrict__ slabs, uint32_t n_wg, float *__restrict__ g_sigma, float *__restrict__ g_color) {
    const uint32_t e = blockIdx.x * 256u + threadIdx.x;      // (role, product, register, lane)
    if (e >= kWgradRoles * kWgradSlab) return;
    const uint32_t role = e / kWgradSlab, i = e % kWgradSlab;
    // sixteen slabs in flight per thread (a chain of one dependent load per slab took 60 us for 128 slabs); the partial sums are combined in
    // a fixed order: the same bits every run
    float part[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) part[u] = 0.0f;
    for (uint32_t w0 = 0; w0 < n_wg; w0 += 16u) {
        float v[16];
#pragma unroll
        for (uint32_t u = 0; u < 16u; ++u) v[u] = w0 + u < n_wg ? slabs[((size_t)(w0 + u) * kWgradRoles + role) * kWgradSlab + i] : 0.0f;
#pragma unroll
        for (int u = 0; u < 16; ++u) part[u] += v[u];
    }
    float s = 0.0f;
#pragma unroll
    for (int u = 0; u < 16; ++u) s += part[u];
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

}  // namespace nsig

using namespace nsig;

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
    k_wgrad_reduce<<<ceil_div(kWgradRoles * kWgradSlab, 256u), 256, 0, st>>>(reinterpret_cast<const float *>(scratch), n_wg, grad_sigma_params, grad_color_params);
    return check_launch("field_wgrad (reduce)");
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
