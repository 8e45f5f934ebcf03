// The HiDDeN message decoder (/root/reference/nerf/hidden_models.py:104-137: ConvBNRelu(Cin,64), 7 x ConvBNRelu(64,64),
// ConvBNRelu(64,1), global average pool, Linear(1,1)) as a chain of fused kernels, forward and backward.
//
// Why: the decoder sees D images of ~12x12 pixels.  Every stock operator on such a tensor (1.2 MB) is launch-latency
// bound, and a training step runs ~120 of them (conv, BN statistics, BN apply, GELU, their backward kernels, MIOpen's
// workspace fills): ~0.9 ms of a 2.25 ms step.  Here a layer is ONE kernel each way:
//
//   forward  layer l : prologue  a_{l-1} = GELU(BN(x_{l-1}))      (batch statistics combined from the producer's partials)
//                      body      x_l = conv3x3(a_{l-1})           (implicit GEMM on MFMA, split-bf16 = fp32-grade products)
//                      epilogue  store x_l, per-(image, tile) partial statistics (sum, M2) of x_l
//   backward layer l : prologue  dx_l = BN-backward(dz_l)         (needs the batch sums of dz_l, dz_l*xhat_l: partials again)
//                      body      G_{l-1} = conv3x3^T(dx_l)  and, in other workgroups of the same launch,
//                                dW_l   = sum_pixels dx_l (x) a_{l-1}   (K = pixels, shifted-window operand)
//                      epilogue  dz_{l-1} = G_{l-1} * GELU'(z_{l-1}), partial sums of it
//
// The kernel boundary is the batch-wide synchronisation BatchNorm needs; nothing else is exchanged between workgroups.
// conv biases are not applied: BatchNorm's mean subtraction cancels them exactly (their gradient is identically zero).
//
// Latency discipline: the tensors are tiny, so a kernel's time is the number of dependent L2 round trips on its critical
// path.  Every loop that reads global memory issues its loads in batches before using them; the 72 A fragments of a
// (rb, tile) unit are all requested before the prologue starts; batch statistics are exchanged as one partial per
// (image, tile pair).
//
// Data layout: activations [B][P][C] fp32 (pixel-major, channels contiguous), P = H*W.  MFMA operands are staged in LDS
// as bf16 hi/lo planes.  Weights are re-packed once per step into A-fragment order (k_dec_pack).
#include <algorithm>
#include <cstring>

#include "common.h"

namespace nsig {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kC = 64;          // hidden channels (hidden_models.py:181)
constexpr int kLayers = 9;      // ConvBNRelu blocks
constexpr int kKS = 36;         // k-steps of a 64->64 3x3 conv: 9 taps x 4 chunks of 16 channels
constexpr int kPitch = 144;     // LDS bytes per halo pixel per plane: 64 bf16 + 16 pad (conflict-free b128 reads)
constexpr int kSets = 15;       // packed fragment sets: layers 1..7 x {forward, dgrad}, + layer 0 dgrad (towards the image)
constexpr size_t kSetU4 = (size_t)2 * kKS * 2 * 64;   // uint4 per set: [rb][ks][plane][lane]
constexpr uint32_t kMaxCin = 32, kMaxP = 1024;

struct DecGeom {
    uint32_t B, H, W, P, Cin, ntile, npair, rows_max;   // rows_max: most halo rows any tile pair stages
    uint32_t RS, nband, R;                               // wgrad: halo row stride (multiple of 8), row bands, rows per band
    uint32_t pd, pa, nks;                                // wgrad: LDS row pitches in bytes (dx, a), k-steps per band
    uint32_t magic_w2;                                   // (pos * magic_w2) >> 16 == pos / (W + 2) for every staged position
    float eps;
};

struct DecWs {
    float *x[kLayers], *xhat[kLayers], *gprime[kLayers], *dz[kLayers], *act[kLayers - 1];
    float *stat[kLayers], *bsum[kLayers], *minv[kLayers];
    float *pool;
    float *wpart;    // [7][B*nband][9][64][64]
    float *wpart0;   // [B*npair][64][9*Cin]
    float *wpart8;   // [B*npair][64][9]
    uint4 *packed;   // [kSets] fragment sets
};

// How the image reaches layer 0.  mode 0: [B][Cin][H][W], used as it is.  mode 1: the rendered blocks [B][H][W][Cin] straight
// from the compositor; clamp to [0,1] and (x - mean_c)/std_c (utils_wtmk_disen.py:599-601: torch.clamp, permute, normalize_img)
// are applied on load, and their backward in the image-gradient epilogue.
struct DecInput {
    uint32_t mode;
    float mean[8], istd[8];
    // The distortion layer of the training step (Trainer.distortion_layer, utils_wtmk_disen.py:551-577, applied at :594 between the clamp and the
    // normalisation; mode 1 only).  0 none; 1 noise: x + noise[b][y][x][c] (the reference draws N(0, 0.1) per element); 2 brightness: clamp(f * x, 0, 1)
    // (torchvision ColorJitter(brightness=0.5): ONE factor f in [0.5, 1.5] per call, blended against black); 3 blurring: the 3x3 Gaussian of
    // torchvision GaussianBlur(3, sigma in [0.01, 0.5]) -- taps exp(-0.5 (d / sigma)^2), d in {-1, 0, 1}, normalised, reflect padding.
    // f / sigma are read from dparam[0] ON THE DEVICE (so a captured step can draw them itself), the noise from dnoise.
    uint32_t distort, H, W;
    const float *dparam, *dnoise;
};
enum : uint32_t { kDistNone = 0, kDistNoise = 1, kDistBrightness = 2, kDistBlur = 3, kDistRotation = 4, kDistScaling = 5 };

__device__ inline float clamp01(float v) { return fminf(fmaxf(v, 0.0f), 1.0f); }
__device__ inline int reflect1(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }   // torch 'reflect' padding by one (n >= 2)
// side weight a and centre weight b of the normalised 1-d kernel [a, b, a]
__device__ inline void blur_taps(float sigma, float &a, float &b) {
    const float e = __expf(-0.5f / (sigma * sigma));
    b = 1.0f / (1.0f + 2.0f * e);
    a = e * b;
}
// the distorted value of channel c at pixel pix of image im, from the rendered blocks [B][H][W][Cin] (before the normalisation)
__device__ inline float distorted_value(const float *__restrict__ img, const DecInput &in, uint32_t im, uint32_t c, uint32_t pix, uint32_t Cin, uint32_t P) {
    const float *base = img + (size_t)im * P * Cin + c;
    if (in.distort == kDistBlur) {
        float a, b;
        blur_taps(in.dparam[0], a, b);
        const int y = (int)(pix / in.W), x = (int)(pix - (uint32_t)y * in.W);
        float acc = 0.0f;
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy) {
            const int yy = reflect1(y + dy, (int)in.H);
            float row = 0.0f;
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) row += (dx == 0 ? b : a) * clamp01(base[(size_t)(yy * (int)in.W + reflect1(x + dx, (int)in.W)) * Cin]);
            acc += (dy == 0 ? b : a) * row;
        }
        return acc;
    }
    const float x = clamp01(base[(size_t)pix * Cin]);
    if (in.distort == kDistNoise) return x + in.dnoise[((size_t)im * P + pix) * Cin + c];
    if (in.distort == kDistBrightness) return clamp01(in.dparam[0] * x);
    return x;
}
__device__ inline float input_value(const float *__restrict__ img, const DecInput &in, uint32_t im, uint32_t c, uint32_t pix, uint32_t Cin, uint32_t P) {
    if (in.mode == 0) return img[((size_t)im * Cin + c) * P + pix];
    return (distorted_value(img, in, im, c, pix, Cin, P) - in.mean[c]) * in.istd[c];
}

struct DecParams {
    const float *w[kLayers], *gamma[kLayers], *beta[kLayers], *lin_w, *lin_b;
};
struct DecGrads {
    float *w[kLayers], *gamma[kLayers], *beta[kLayers], *lin_w, *lin_b;
};

__host__ __device__ inline uint32_t tile_count(uint32_t tile, uint32_t P) { return P - 32u * tile < 32u ? P - 32u * tile : 32u; }
__host__ __device__ inline uint32_t pair_count(uint32_t pair, uint32_t P) { return P - 64u * pair < 64u ? P - 64u * pair : 64u; }

// Chan's update: fold a group (n_b values, sum_b, M2_b about its own mean) into a running (cnt, mean, M2).
__device__ inline void chan_merge(float &cnt, float &mean, float &M2, float n_b, float sum_b, float M2_b) {
    if (n_b <= 0.0f) return;
    const float tot = cnt + n_b, delta = sum_b / n_b - mean;
    mean += delta * (n_b / tot);
    M2 += M2_b + delta * delta * (cnt * n_b / tot);
    cnt = tot;
}

// GELU (erf form) and its derivative from one exponential: erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, i.e. at the
// fp32 rounding level of Phi), sharing exp(-z^2/2) with the Gaussian density.  ~25 VALU operations instead of ~90 for erff + expf.
__device__ inline void gelu_parts(float z, float &a, float &gp) {
    const float e = __expf(-0.5f * z * z);
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * 0.70710678118654752f * fabsf(z));
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float tail = 0.5f * poly * e;              // 1 - Phi(|z|)
    const float Phi = z >= 0.0f ? 1.0f - tail : tail;
    a = z * Phi;
    gp = Phi + z * (0.39894228040143268f * e);
}

__device__ inline float half_sum(float v) {   // sum over the 32 lanes that share lane >> 5
#pragma unroll
    for (int d = 16; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}
__device__ inline float wave_sum64(float v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}
__device__ inline float block_sum256(float v, float *scratch) {   // scratch: 4 floats
    v = wave_sum64(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    return scratch[0] + scratch[1] + scratch[2] + scratch[3];
}

__device__ inline void split_bf16(float v, __bf16 &hi, __bf16 &lo) {
    hi = (__bf16)v;
    lo = (__bf16)(v - (float)hi);
}

// ----------------------------------------------------------------------------- weight packing

// forward set (kind 0):  A[row = co][k = (tap, ci)]   = W[co][ci][tap]
// dgrad set   (kind 1):  A[row = ci][k = (tap', co)]  = W[co][ci][8 - tap']      (transposed, taps flipped)
// set 14: layer 0's dgrad, rows = input channel (< Cin, rest zero), W0 is [64][Cin][3][3].
// fragment (rb, ks = tap*4 + c4), lane (row = lane & 31, h = lane >> 5), element j: k-channel 16*c4 + 8*h + j.
__global__ void __launch_bounds__(64) k_dec_pack(DecParams prm, uint4 *__restrict__ packed, uint32_t Cin) {
    const int f = blockIdx.x, ks = f % kKS, rb = (f / kKS) & 1, set = f / (2 * kKS);
    const int lane = threadIdx.x, row = 32 * rb + (lane & 31), h = lane >> 5, tap = ks >> 2, c4 = ks & 3;
    const int layer = set < 14 ? 1 + (set >> 1) : 0, kind = set < 14 ? (set & 1) : 1;
    const float *__restrict__ w = prm.w[layer];
    bf16x8 hi, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int kc = 16 * c4 + 8 * h + j;
        float v;
        if (layer == 0)
            v = row < (int)Cin ? w[(kc * Cin + row) * 9 + (8 - tap)] : 0.0f;
        else
            v = kind == 0 ? w[(row * kC + kc) * 9 + tap] : w[(kc * kC + row) * 9 + (8 - tap)];
        __bf16 vh, vl;
        split_bf16(v, vh, vl);
        hi[j] = vh;
        lo[j] = vl;
    }
    uint4 *dst = packed + (size_t)set * kSetU4 + ((size_t)rb * kKS + ks) * 128;
    dst[lane] = *reinterpret_cast<uint4 *>(&hi);
    dst[64 + lane] = *reinterpret_cast<uint4 *>(&lo);
}

// ----------------------------------------------------------------------------- batch statistics from partials

// Batch statistics travel between kernels as one partial per (pair, image) and channel: index i = pair*B + image, so the
// partials of the (possibly shorter) last pair are the contiguous tail and a group's size needs no division.
constexpr int kPartMax = 24;   // float4 partials per thread held in registers: B*npair <= 8*kPartMax

// kT threads = 32 channel pairs x kT/32 groups; thread (cp, grp) holds partials i = grp + (kT/32) k of channels 2cp, 2cp+1.
struct Partials {
    float4 v[kPartMax];   // (a, b) of channel 2cp, (a, b) of channel 2cp + 1
};
template <int K0, int K1, int kT>
__device__ inline void load_partial_range(const float4 *__restrict__ p4, uint32_t n, Partials &pv) {
    const uint32_t cp = threadIdx.x & 31, grp = threadIdx.x >> 5;
#pragma unroll
    for (int k = K0; k < K1; ++k) pv.v[k] = p4[min(grp + (uint32_t)(kT / 32) * k, n - 1) * 32 + cp];   // clamped, not predicated: no branches between the loads
}
// Entries past the end hold a copy of the last partial; the combine functions mask them by index.
template <int kT = 256>
__device__ inline void load_partials(const float *__restrict__ part, uint32_t n, Partials &pv) {
    constexpr int kHeld = kPartMax * 256 / kT;   // partials per thread: 24 with 8 groups, 12 with 16
    const float4 *__restrict__ p4 = reinterpret_cast<const float4 *>(part);
    load_partial_range<0, kHeld / 2, kT>(p4, n, pv);
    if (n > (kT / 32) * (kHeld / 2)) {
        load_partial_range<kHeld / 2, kHeld, kT>(p4, n, pv);
    } else {
#pragma unroll
        for (int k = kHeld / 2; k < kHeld; ++k) pv.v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

// Forward: partials (sum, M2 about the pair mean) -> batch mean and 1/sqrt(var + eps): two passes over the registers,
// M2 = sum_i M2_i + n_i (mean_i - mean)^2 (no E[x^2] - E[x]^2 cancellation, no divisions).  red: 2 * 64 * kT/32 floats (the group
// sums of the two passes side by side: three barriers in all -- every thread adds up the group sums of its own two channels
// instead of waiting for 64 threads to publish the means).
template <int kT = 256>
__device__ inline void combine_fwd_stats(const Partials &pv, const DecGeom &g, float *red, float *s_mean, float *s_inv) {
    constexpr int kG = kT / 32, kHeld = kPartMax * 256 / kT;
    const uint32_t cp = threadIdx.x & 31, grp = threadIdx.x >> 5, c = threadIdx.x & 63;
    const uint32_t n = g.B * g.npair, last0 = (g.npair - 1) * g.B;
    const float N = (float)(g.B * g.P), n_last = (float)pair_count(g.npair - 1, g.P), rn_last = 1.0f / n_last;
    float *red2 = red + 64 * kG;
    float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
    for (int k = 0; k < kHeld; ++k) {
        const bool in = grp + (uint32_t)kG * k < n;
        s0 += in ? pv.v[k].x : 0.0f;
        s1 += in ? pv.v[k].z : 0.0f;
    }
    *reinterpret_cast<float2 *>(red + grp * 64 + 2 * cp) = make_float2(s0, s1);
    __syncthreads();
    float m0 = 0.0f, m1 = 0.0f;
#pragma unroll
    for (int k = 0; k < kG; ++k) {
        const float2 v = *reinterpret_cast<const float2 *>(red + k * 64 + 2 * cp);
        m0 += v.x;
        m1 += v.y;
    }
    m0 = m0 / N;   // the batch means of channels 2cp, 2cp + 1 (the same operations in every thread that shares them)
    m1 = m1 / N;
    float q0 = 0.0f, q1 = 0.0f;
#pragma unroll
    for (int k = 0; k < kHeld; ++k) {
        const uint32_t i = grp + kG * k;
        const bool last = i >= last0;
        const float ni = last ? n_last : 64.0f, rni = last ? rn_last : 1.0f / 64.0f;
        const float d0 = pv.v[k].x - ni * m0, d1 = pv.v[k].z - ni * m1;     // n_i * (mean_i - mean)
        q0 += i < n ? pv.v[k].y + d0 * d0 * rni : 0.0f;
        q1 += i < n ? pv.v[k].w + d1 * d1 * rni : 0.0f;
    }
    *reinterpret_cast<float2 *>(red2 + grp * 64 + 2 * cp) = make_float2(q0, q1);
    if (grp == 0) *reinterpret_cast<float2 *>(s_mean + 2 * cp) = make_float2(m0, m1);
    __syncthreads();
    if (threadIdx.x < 64) {
        float M2 = 0.0f;
#pragma unroll
        for (int k = 0; k < kG; ++k) M2 += red2[k * 64 + c];
        s_inv[c] = 1.0f / sqrtf(M2 / N + g.eps);
    }
    __syncthreads();
}

// Backward: k = gamma*inv/N, S1 = sum dz, S2 = sum dz*xhat over the batch.  red: 2 * 64 * kT/32 floats; tab: [3][64].
template <int kT = 256>
__device__ inline void combine_bwd_sums(const Partials &pv, const float *__restrict__ gamma, const float *__restrict__ minv, const DecGeom &g,
                                        float *red, float *tab) {
    constexpr int kG = kT / 32, kHeld = kPartMax * 256 / kT;
    const uint32_t cp = threadIdx.x & 31, grp = threadIdx.x >> 5, c = threadIdx.x & 63;
    const uint32_t n = g.B * g.npair;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < kHeld; ++k) {
        const bool in = grp + (uint32_t)kG * k < n;
        s.x += in ? pv.v[k].x : 0.0f;
        s.y += in ? pv.v[k].y : 0.0f;
        s.z += in ? pv.v[k].z : 0.0f;
        s.w += in ? pv.v[k].w : 0.0f;
    }
    red[grp * 64 + 2 * cp] = s.x;
    red[grp * 64 + 2 * cp + 1] = s.z;
    red[64 * kG + grp * 64 + 2 * cp] = s.y;
    red[64 * kG + grp * 64 + 2 * cp + 1] = s.w;
    __syncthreads();
    if (threadIdx.x < 64) {
        float S1 = 0.0f, S2 = 0.0f;
#pragma unroll
        for (int k = 0; k < kG; ++k) {
            S1 += red[k * 64 + c];
            S2 += red[64 * kG + k * 64 + c];
        }
        tab[c] = gamma[c] * minv[64 + c] / (float)(g.B * g.P);
        tab[64 + c] = S1;
        tab[128 + c] = S2;
    }
    __syncthreads();
}

// ----------------------------------------------------------------------------- staging: what a tile pair reads

// Rows of the halo image an (image, pair) workgroup stages: staged row r <-> image row py0 + r - 1.
struct PairRows {
    uint32_t q0, q1, py0, nrow;
};
__host__ __device__ inline PairRows pair_rows(uint32_t pair, uint32_t P, uint32_t W) {
    PairRows r;
    r.q0 = pair * 64;
    r.q1 = (r.q0 + 64 < P ? r.q0 + 64 : P) - 1;
    r.py0 = r.q0 / W;
    r.nrow = r.q1 / W - r.py0 + 3;
    return r;
}

// Staged position -> pixel index, or -1 for the zero padding.
__device__ inline int staged_pixel(uint32_t pos, uint32_t W2, uint32_t py0, const DecGeom &g) {
    const uint32_t hr = (pos * g.magic_w2) >> 16, hc = pos - hr * W2;   // pos / W2 (magic verified by make_geom for every staged pos)
    const int iy = (int)(py0 + hr) - 1, ix = (int)hc - 1;
    return (iy >= 0 && iy < (int)g.H && ix >= 0 && ix < (int)g.W) ? iy * (int)g.W + ix : -1;
}

// ----------------------------------------------------------------------------- layer 0 forward: Cin -> 64 on the VALU

// grid (npair, B), 256 threads: thread = (co, quarter of the pair's 64 pixels).  K = 9*Cin is too short for MFMA to matter.
// Only the rows the pair touches are staged (rows_max x (W+2) positions per channel); kCin > 0: the thread's 9*kCin weights
// live in registers (the watermark path has kCin = 3), kCin = 0: any Cin, weights read from LDS.
template <int kCin>
__global__ void __launch_bounds__(256) k_dec_l0_fwd(const float *__restrict__ img, DecInput inp, float *__restrict__ clamped_out, DecParams prm, DecWs ws, DecGeom g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const uint32_t pair = blockIdx.x, im = blockIdx.y, t = threadIdx.x;
    const uint32_t W2 = g.W + 2, Cin = kCin ? kCin : g.Cin, K = 9 * Cin;
    const PairRows pr = pair_rows(pair, g.P, g.W);
    const uint32_t npos = pr.nrow * W2, cap = g.rows_max * W2;
    float *s_img = smem;                    // [Cin][rows_max*(W+2)] zero-padded rows of the pair
    float *s_red = s_img + Cin * cap;       // [4][64]
    float *s_w = s_red + 256;               // [64][K] (kCin == 0 only)
    const uint32_t co = t & 63, sub = t >> 6, q0 = pair * 64 + sub * 16;
    float wreg[kCin ? 9 * kCin : 1];
    if (kCin) {
#pragma unroll
        for (int k = 0; k < 9 * kCin; ++k) wreg[k] = prm.w[0][co * K + k];   // [co][ci][3][3] row-major = [co][K]
    } else {
        for (uint32_t i = t; i < kC * K; i += 256) s_w[i] = prm.w[0][i];
    }
    for (uint32_t c = 0; c < Cin; ++c)
        for (uint32_t pos = t; pos < npos; pos += 256) {
            const int q = staged_pixel(pos, W2, pr.py0, g);
            s_img[c * cap + pos] = q >= 0 ? input_value(img, inp, im, c, q, Cin, g.P) : 0.0f;
        }
    if (clamped_out && inp.mode == 1)   // the clamped render the caller reports (pred_rgb), own pixels only
        for (uint32_t i = t; i < 64 * Cin; i += 256) {
            const uint32_t e = pair * 64 * Cin + i;
            if (e < g.P * Cin) clamped_out[(size_t)im * g.P * Cin + e] = fminf(fmaxf(img[(size_t)im * g.P * Cin + e], 0.0f), 1.0f);
        }
    __syncthreads();
    float v[16];
    float s = 0.0f;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const uint32_t q = q0 + j;
        v[j] = 0.0f;
        if (q < g.P) {
            const uint32_t py = q / g.W, px = q - py * g.W, hp0 = (py - pr.py0) * W2 + px;
            float acc = 0.0f;
            if (kCin) {
#pragma unroll
                for (int c = 0; c < (kCin ? kCin : 1); ++c)
#pragma unroll
                    for (int tap = 0; tap < 9; ++tap) acc += wreg[c * 9 + tap] * s_img[c * cap + hp0 + (tap / 3) * W2 + tap % 3];
            } else {
                for (uint32_t c = 0; c < Cin; ++c)
#pragma unroll
                    for (int tap = 0; tap < 9; ++tap) acc += s_w[co * K + c * 9 + tap] * s_img[c * cap + hp0 + (tap / 3) * W2 + tap % 3];
            }
            v[j] = acc;
            ws.x[0][((size_t)im * g.P + q) * kC + co] = acc;
            s += acc;
        }
    }
    // pair statistics, two passes inside the workgroup: sum -> mean, then M2 about it
    s_red[sub * 64 + co] = s;
    __syncthreads();
    const float ts = s_red[co] + s_red[64 + co] + s_red[128 + co] + s_red[192 + co];
    const float mean = ts / (float)pair_count(pair, g.P);
    float m2 = 0.0f;
#pragma unroll
    for (int j = 0; j < 16; ++j)
        if (q0 + j < g.P) m2 += (v[j] - mean) * (v[j] - mean);
    __syncthreads();
    s_red[sub * 64 + co] = m2;
    __syncthreads();
    if (sub == 0) {
        float *po = ws.stat[0] + (((size_t)pair * g.B + im) * kC + co) * 2;
        po[0] = ts;
        po[1] = s_red[co] + s_red[64 + co] + s_red[128 + co] + s_red[192 + co];
    }
}

// a = GELU(BN(x)) for 4 channels of one pixel; tab = [mean | inv | gamma | beta].
__device__ inline void bn_gelu4(float4 xv, uint32_t cg, const float *tab, float (&xh)[4], float (&a)[4], float (&gp)[4]) {
    const float xin[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = cg * 4 + j;
        xh[j] = (xin[j] - tab[c]) * tab[64 + c];
        gelu_parts(xh[j] * tab[128 + c] + tab[192 + c], a[j], gp[j]);
    }
}

// dx = k * (N*dz - S1 - xhat*S2) for 4 channels of one pixel (BatchNorm backward, batch statistics); tab = [k | S1 | S2].
__device__ inline float4 bn_bwd4(float4 d, float4 h, uint32_t cg, const float *tab, float N) {
    const float dv[4] = {d.x, d.y, d.z, d.w}, hv[4] = {h.x, h.y, h.z, h.w};
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = cg * 4 + j;
        o[j] = tab[c] * (N * dv[j] - tab[64 + c] - hv[j] * tab[128 + c]);
    }
    return make_float4(o[0], o[1], o[2], o[3]);
}

__device__ inline void split4_to_lds(char *lds_hi, char *lds_lo, uint32_t pos, uint32_t cg, const float (&v)[4]) {
    bf16x4 hi, lo;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        __bf16 vh, vl;
        split_bf16(v[j], vh, vl);
        hi[j] = vh;
        lo[j] = vl;
    }
    *reinterpret_cast<bf16x4 *>(lds_hi + pos * kPitch + cg * 8) = hi;
    *reinterpret_cast<bf16x4 *>(lds_lo + pos * kPitch + cg * 8) = lo;
}

constexpr int kBatch = 4;   // prologue items whose loads are in flight together

// Forward prologue shared by the MFMA layers (bf16 hi/lo planes) and layer 8 (fp32 rows): stage GELU(BN(x_prev)) for every
// position the pair touches; the owner of a pixel also records xhat, GELU' and a for the backward pass.
template <bool kF32, int kT = 256>
__device__ inline void stage_forward(const float *__restrict__ x, float *__restrict__ xhat, float *__restrict__ gprime, float *__restrict__ act,
                                     uint32_t im, const PairRows &pr, const DecGeom &g, const float *tab, char *lds_hi, char *lds_lo, float *lds_f32, uint32_t first = 0, bool record = true) {
    const uint32_t W2 = g.W + 2, total = pr.nrow * W2 * 16;
    for (uint32_t base = first + threadIdx.x; base < total; base += kT * kBatch) {
        float4 xv[kBatch];
        int q[kBatch];
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
            const uint32_t i = base + kT * u;
            q[u] = i < total ? staged_pixel(i >> 4, W2, pr.py0, g) : -1;
            xv[u] = q[u] >= 0 ? *reinterpret_cast<const float4 *>(x + ((size_t)im * g.P + q[u]) * kC + (i & 15) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
            const uint32_t i = base + kT * u, pos = i >> 4, cg = i & 15;
            if (i >= total) break;
            float xh[4], a[4] = {0.f, 0.f, 0.f, 0.f}, gp[4];
            if (q[u] >= 0) {
                bn_gelu4(xv[u], cg, tab, xh, a, gp);
                if (record && (uint32_t)q[u] >= pr.q0 && (uint32_t)q[u] <= pr.q1) {
                    const size_t e = ((size_t)im * g.P + q[u]) * kC + cg * 4;
                    *reinterpret_cast<float4 *>(xhat + e) = make_float4(xh[0], xh[1], xh[2], xh[3]);
                    *reinterpret_cast<float4 *>(gprime + e) = make_float4(gp[0], gp[1], gp[2], gp[3]);
                    *reinterpret_cast<float4 *>(act + e) = make_float4(a[0], a[1], a[2], a[3]);
                }
            }
            if (kF32)
                *reinterpret_cast<float4 *>(lds_f32 + (size_t)pos * kC + cg * 4) = make_float4(a[0], a[1], a[2], a[3]);
            else
                split4_to_lds(lds_hi, lds_lo, pos, cg, a);
        }
    }
}

// Backward prologue: stage dx = BN-backward(dz) for every position the pair touches.
template <int kT = 256>
__device__ inline void stage_backward(const float *__restrict__ dz, const float *__restrict__ xhat, uint32_t im, const PairRows &pr, const DecGeom &g,
                                      const float *tab, char *lds_hi, char *lds_lo, uint32_t first = 0) {
    const uint32_t W2 = g.W + 2, total = pr.nrow * W2 * 16;
    const float N = (float)(g.B * g.P);
    for (uint32_t base = first + threadIdx.x; base < total; base += kT * kBatch) {
        float4 dv[kBatch], hv[kBatch];
        int q[kBatch];
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
            const uint32_t i = base + kT * u;
            q[u] = i < total ? staged_pixel(i >> 4, W2, pr.py0, g) : -1;
            const size_t e = ((size_t)im * g.P + (q[u] >= 0 ? q[u] : 0)) * kC + (i & 15) * 4;
            dv[u] = q[u] >= 0 ? *reinterpret_cast<const float4 *>(dz + e) : make_float4(0.f, 0.f, 0.f, 0.f);
            hv[u] = q[u] >= 0 ? *reinterpret_cast<const float4 *>(xhat + e) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
            const uint32_t i = base + kT * u;
            if (i >= total) break;
            const float4 d = q[u] >= 0 ? bn_bwd4(dv[u], hv[u], i & 15, tab, N) : make_float4(0.f, 0.f, 0.f, 0.f);
            const float v[4] = {d.x, d.y, d.z, d.w};
            split4_to_lds(lds_hi, lds_lo, i >> 4, i & 15, v);
        }
    }
}

// ----------------------------------------------------------------------------- the 64 -> 64 conv kernel (forward / dgrad)

// k-steps per wave: the 36 k-steps of a (rb, tile) unit are split over the waves that share the tile -- two in a 256-thread
// workgroup (18 each), four in a 512-thread one (9 each)
template <int K>
struct AFrags {
    uint4 hi[K], lo[K];   // the A operand of one (rb, tile, k-part), requested up front (8 VGPRs per k-step)
};
template <int K>
__device__ inline void load_afrags(const uint4 *__restrict__ A, int lane, AFrags<K> &f) {
#pragma unroll
    for (int ks = 0; ks < K; ++ks) {
        f.hi[ks] = A[ks * 128 + lane];
        f.lo[ks] = A[ks * 128 + 64 + lane];
    }
}

// One k-part of a (rb, tile) unit: acc[32 rows x 32 pixels] += K k-steps starting at ks0 (wave-uniform); B fragments are 16-byte
// LDS reads of the lane's pixel at the tap's offset.
template <int K>
__device__ inline f32x16 conv_part(AFrags<K> &f, const char *lds_hi, const char *lds_lo, uint32_t hp0, uint32_t W2, int lane, int ks0) {
    f32x16 c = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int h = lane >> 5;
#pragma unroll
    for (int ks = 0; ks < K; ++ks) {
        const int kg = ks0 + ks, tap = kg >> 2, c4 = kg & 3, ty = tap / 3, tx = tap - 3 * ty;   // scalar
        const uint32_t off = (hp0 + ty * W2 + tx) * kPitch + (16 * c4 + 8 * h) * 2;
        const bf16x8 b_hi = *reinterpret_cast<const bf16x8 *>(lds_hi + off);
        const bf16x8 b_lo = *reinterpret_cast<const bf16x8 *>(lds_lo + off);
        const bf16x8 ah = *reinterpret_cast<bf16x8 *>(&f.hi[ks]), al = *reinterpret_cast<bf16x8 *>(&f.lo[ks]);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, b_hi, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b_lo, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b_hi, c, 0, 0, 0);
    }
    return c;
}

enum ConvMode { kFwd = 0, kDgrad = 1, kDgradImg = 2 };

#ifdef NSIG_DEC_TIMING
// Phase stamps of workgroup (0,0), wave 0 (tools/dec_timing.py builds a private copy of the library with this enabled).
__device__ unsigned long long g_dec_stamps[3][16];
#define DEC_STAMP(k)                                                                               \
    do {                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) g_dec_stamps[MODE][k] = wall_clock64(); \
        __builtin_amdgcn_sched_barrier(0);                                                         \
    } while (0)
#else
#define DEC_STAMP(k)
#endif

__device__ inline float bcast_lane(float v, int src_lane) {   // src_lane: compile-time constant
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src_lane));
}

// Per-channel sums over the 32 pixels of a wave's accumulator tile, two quantities at once, through an LDS transpose
// (pitch 33: conflict-free both ways) instead of 160 cross-lane shuffles.  va/vb: the lane's 16 values (rows 8*g4 + 4*h + j).
// Returns in lanes (ch = lane & 31) the sums for channel ch, identical in both halves.  scr: 2*32*33 floats per wave.
__device__ inline float2 tile_channel_sums(const float (&va)[16], const float (&vb)[16], float *scr, int lane) {
    const int p = lane & 31, h = lane >> 5;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int ch = 8 * (r >> 2) + 4 * h + (r & 3);
        scr[ch * 33 + p] = va[r];
        scr[32 * 33 + ch * 33 + p] = vb[r];
    }
    float sa = 0.0f, sb = 0.0f;   // same wave, same array: program order is enough
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        sa += scr[p * 33 + 16 * h + k];
        sb += scr[32 * 33 + p * 33 + 16 * h + k];
    }
    sa += __shfl_xor(sa, 32, 64);
    sb += __shfl_xor(sb, 32, 64);
    return make_float2(sa, sb);
}

// grid (npair, B, 2 row blocks), kT threads.  A workgroup owns 32 output rows (rb = blockIdx.z) of one tile pair; its waves are
// (tile of the pair) x (k-part) -- k-halves with 256 threads, k-quarters with 512 -- so every wave streams only its share of the
// 72 A fragments, and the k-parts of a tile meet through LDS.  All waves build the B operand (the prologue is the longest phase:
// that, not the MFMA count, is why the workgroup has eight waves); the rb = 1 workgroup repeats that staging on another CU -- the
// chip has more CUs (256) than this layer has tile pairs (96).
//   kFwd      (layer 1..7): consumes x[l-1], stat[l-1]; writes xhat/gprime/act[l-1] (own pixels, rb 0), minv[l-1], x[l], stat[l].
//   kDgrad    (layer 7..1): consumes dz[l], xhat[l], bsum[l]; writes dz[l-1], bsum[l-1].
//   kDgradImg (layer 0)   : consumes dz[0], xhat[0], bsum[0]; writes the gradient of the input image [B][Cin][H][W]  (grid z = 1).
// Order of global requests (loads return in order): batch partials, the prologue inputs, then the A fragments -- which land
// while the statistics are combined and the prologue computes.
template <int MODE, int kT>
__global__ void __launch_bounds__(kT) k_dec_conv(int layer, DecParams prm, DecWs ws, DecGeom g, float *__restrict__ grad_img,
                                                  const float *__restrict__ img, DecInput inp) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const uint32_t pair = blockIdx.x, im = blockIdx.y, rb = blockIdx.z, t = threadIdx.x;
    const uint32_t W2 = g.W + 2;
    const PairRows pr = pair_rows(pair, g.P, g.W);
    const uint32_t plane = g.rows_max * W2 * kPitch, total = pr.nrow * W2 * 16;
    char *lds_hi = smem_raw, *lds_lo = smem_raw + plane;
    float *s_tab = reinterpret_cast<float *>(smem_raw + 2 * plane);   // [4][64] per-channel constants
    float *s_red = s_tab + 4 * 64;                                    // [1024] scratch
    float *s_scr = s_red + 1024;                                      // [2 tiles][2*32*33] epilogue transposes
    float *s_acc = s_scr + 2 * (2 * 32 * 33);                         // [2 tiles][3][16][64] k-part exchange
    float *s_ref = s_acc + 2 * 3 * 16 * 64;                           // [2 tiles][32]
    const float N = (float)(g.B * g.P);

    // waves = (tile of the pair) x (k-part): kT / 128 waves share a tile; the one with kp == 0 owns its epilogue.  The request,
    // statistics and prologue phases -- the longest ones, all per-thread element counts -- are spread over every thread.
    constexpr int kParts = kT / 128, kK = kKS / kParts;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63, tp = wave & 1, kp = wave >> 1, p = lane & 31, h = lane >> 5;
    const uint32_t tile = 2 * pair + tp;
    const bool active = tile < g.ntile;
    const int set = MODE == kFwd ? (layer - 1) * 2 : (MODE == kDgrad ? (layer - 1) * 2 + 1 : 14);
    const uint32_t img_off = im * g.P * kC;   // element offsets fit 32 bits (checked by make_geom)
    DEC_STAMP(0);

    // ---- requests
    Partials pv;
    load_partials<kT>(MODE == kFwd ? ws.stat[layer - 1] : ws.bsum[layer], g.B * g.npair, pv);
    DEC_STAMP(7);
    const float *__restrict__ in0 = MODE == kFwd ? ws.x[layer - 1] : ws.dz[layer];
    const float *__restrict__ in1 = ws.xhat[layer];   // backward only
    constexpr int kPre = 2048 / kT;
    float4 v0[kPre], v1[kPre];
    int q[kPre];
#pragma unroll
    for (int u = 0; u < kPre; ++u) {
        const uint32_t i = t + kT * u;
        q[u] = i < total ? staged_pixel(i >> 4, W2, pr.py0, g) : -1;
        const uint32_t e = img_off + (q[u] >= 0 ? q[u] : 0) * kC + (i & 15) * 4;
        v0[u] = q[u] >= 0 ? *reinterpret_cast<const float4 *>(in0 + e) : make_float4(0.f, 0.f, 0.f, 0.f);
        if (MODE != kFwd) v1[u] = q[u] >= 0 ? *reinterpret_cast<const float4 *>(in1 + e) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    DEC_STAMP(8);
    AFrags<kK> af;
    if (active) load_afrags(ws.packed + (size_t)set * kSetU4 + ((size_t)rb * kKS + kp * kK) * 128, lane, af);
    DEC_STAMP(1);

    // ---- batch statistics -> per-channel constants
    if (MODE == kFwd) {
        combine_fwd_stats<kT>(pv, g, s_red, s_tab, s_tab + 64);
        if (t < 64) {
            s_tab[128 + t] = prm.gamma[layer - 1][t];
            s_tab[192 + t] = prm.beta[layer - 1][t];
            if (pair == 0 && im == 0 && rb == 0) {
                ws.minv[layer - 1][t] = s_tab[t];
                ws.minv[layer - 1][64 + t] = s_tab[64 + t];
            }
        }
        __syncthreads();
    } else {
        combine_bwd_sums<kT>(pv, prm.gamma[layer], ws.minv[layer], g, kT == 256 ? s_red : s_scr, s_tab);   // (s_scr: 4224 floats, idle until the epilogue)
    }
    DEC_STAMP(2);

    // ---- prologue: the B operand of every position this pair touches, halo included
#pragma unroll
    for (int u = 0; u < kPre; ++u) {
        const uint32_t i = t + kT * u, pos = i >> 4, cg = i & 15;
        if (i >= total) break;
        float a[4] = {0.f, 0.f, 0.f, 0.f};
        if (q[u] >= 0) {
            if (MODE == kFwd) {
                float xh[4], gp[4];
                bn_gelu4(v0[u], cg, s_tab, xh, a, gp);
                if (rb == 0 && (uint32_t)q[u] >= pr.q0 && (uint32_t)q[u] <= pr.q1) {
                    const uint32_t e = img_off + q[u] * kC + cg * 4;
                    *reinterpret_cast<float4 *>(ws.xhat[layer - 1] + e) = make_float4(xh[0], xh[1], xh[2], xh[3]);
                    *reinterpret_cast<float4 *>(ws.gprime[layer - 1] + e) = make_float4(gp[0], gp[1], gp[2], gp[3]);
                    *reinterpret_cast<float4 *>(ws.act[layer - 1] + e) = make_float4(a[0], a[1], a[2], a[3]);
                }
            } else {
                const float4 d = bn_bwd4(v0[u], v1[u], cg, s_tab, N);
                a[0] = d.x; a[1] = d.y; a[2] = d.z; a[3] = d.w;
            }
        }
        split4_to_lds(lds_hi, lds_lo, pos, cg, a);
    }
    if (total > kT * kPre) {   // pairs that stage more than the preloaded batch: the rest, batched
        if (MODE == kFwd)
            stage_forward<false, kT>(ws.x[layer - 1], ws.xhat[layer - 1], ws.gprime[layer - 1], ws.act[layer - 1], im, pr, g, s_tab, lds_hi, lds_lo, nullptr, kT * kPre, rb == 0);
        else
            stage_backward<kT>(ws.dz[layer], ws.xhat[layer], im, pr, g, s_tab, lds_hi, lds_lo, kT * kPre);
    }
    DEC_STAMP(3);

    // ---- body: lane holds rows 32*rb + 8*g4 + 4*h + {0..3} (g4 = 0..3) of pixel qo
    const uint32_t qo = 32 * tile + p;
    const bool valid = active && qo < g.P;
    const uint32_t qq = valid ? qo : pr.q0, py = qq / g.W, px = qq - py * g.W;
    const uint32_t eo = img_off + qq * kC + 32 * rb + 4 * h;
    float4 gp4[4], xh4[4];
    if (MODE == kDgrad && kp == 0) {   // epilogue operands, requested before the MFMA loop
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            gp4[g4] = valid ? *reinterpret_cast<const float4 *>(ws.gprime[layer - 1] + eo + 8 * g4) : make_float4(0.f, 0.f, 0.f, 0.f);
            xh4[g4] = valid ? *reinterpret_cast<const float4 *>(ws.xhat[layer - 1] + eo + 8 * g4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    __syncthreads();
    DEC_STAMP(4);
    f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (active) acc = conv_part(af, lds_hi, lds_lo, (py - pr.py0) * W2 + px, W2, lane, kp * kK);
    if (kp != 0)
#pragma unroll
        for (int r = 0; r < 16; ++r) s_acc[((tp * 3 + kp - 1) * 16 + r) * 64 + lane] = acc[r];
    __syncthreads();
    if (kp == 0)
#pragma unroll
        for (int q = 0; q < kParts - 1; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] += s_acc[((tp * 3 + q) * 16 + r) * 64 + lane];
#ifdef NSIG_DEC_TIMING
    if (acc[0] == 12345.678f) DEC_STAMP(15);   // make the stamp below wait for the accumulators
#endif
    DEC_STAMP(5);

    // ---- epilogue (waves kp == 0 hold the tiles; the others only keep the barriers company)
    float *scr = s_scr + tp * (2 * 32 * 33), *s_ep = s_red;   // s_ep: [2 tiles][32 channels][2]
    if (MODE == kFwd) {
        if (kp == 0) {
            if (valid) {
                float *xo = ws.x[layer] + eo;
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) *reinterpret_cast<float4 *>(xo + 8 * g4) = make_float4(acc[4 * g4], acc[4 * g4 + 1], acc[4 * g4 + 2], acc[4 * g4 + 3]);
            }
            // sum and sum of squares about a reference sample of the same channel (the tile's first pixel):
            // M2 = sum d^2 - (sum d)^2/n cancels only relative to the spread inside the tile.
            float d[16], d2[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float r0 = bcast_lane(acc[r], 0), r1 = bcast_lane(acc[r], 32);
                const float ref = h ? r1 : r0;
                d[r] = valid ? acc[r] - ref : 0.0f;
                d2[r] = d[r] * d[r];
                if (p == 0) s_ref[tp * 32 + 8 * (r >> 2) + 4 * h + (r & 3)] = ref;
            }
            const float2 sums = tile_channel_sums(d, d2, scr, lane);   // lane -> channel p of this row block
            if (h == 0) {
                const float n_t = active ? (float)tile_count(tile, g.P) : 0.0f;
                s_ep[(tp * 32 + p) * 2] = active ? sums.x + n_t * s_ref[tp * 32 + p] : 0.0f;                 // sum of x
                s_ep[(tp * 32 + p) * 2 + 1] = active ? sums.y - sums.x * sums.x / fmaxf(n_t, 1.0f) : 0.0f;   // M2 about the tile mean
            }
        }
        __syncthreads();
        if (t < 32) {   // merge the pair's two tiles
            float cnt = (float)tile_count(2 * pair, g.P), mean = s_ep[t * 2] / cnt, M2 = s_ep[t * 2 + 1];
            if (2 * pair + 1 < g.ntile) chan_merge(cnt, mean, M2, (float)tile_count(2 * pair + 1, g.P), s_ep[(32 + t) * 2], s_ep[(32 + t) * 2 + 1]);
            float *po = ws.stat[layer] + ((pair * g.B + im) * kC + 32 * rb + t) * 2;
            po[0] = mean * cnt;
            po[1] = M2;
        }
        DEC_STAMP(6);
    } else if (MODE == kDgrad) {
        if (kp == 0) {
            float dzv[16], dzx[16];
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const float gpv[4] = {gp4[g4].x, gp4[g4].y, gp4[g4].z, gp4[g4].w}, xhv[4] = {xh4[g4].x, xh4[g4].y, xh4[g4].z, xh4[g4].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    dzv[4 * g4 + j] = valid ? acc[4 * g4 + j] * gpv[j] : 0.0f;
                    dzx[4 * g4 + j] = dzv[4 * g4 + j] * xhv[j];
                }
                if (valid) *reinterpret_cast<float4 *>(ws.dz[layer - 1] + eo + 8 * g4) = make_float4(dzv[4 * g4], dzv[4 * g4 + 1], dzv[4 * g4 + 2], dzv[4 * g4 + 3]);
            }
            const float2 sums = tile_channel_sums(dzv, dzx, scr, lane);
            if (h == 0) {
                s_ep[(tp * 32 + p) * 2] = active ? sums.x : 0.0f;
                s_ep[(tp * 32 + p) * 2 + 1] = active ? sums.y : 0.0f;
            }
        }
        __syncthreads();
        if (t < 32) {
            float *po = ws.bsum[layer - 1] + ((pair * g.B + im) * kC + 32 * rb + t) * 2;
            po[0] = s_ep[t * 2] + s_ep[(32 + t) * 2];
            po[1] = s_ep[t * 2 + 1] + s_ep[(32 + t) * 2 + 1];
        }
        DEC_STAMP(6);
    } else {
        // rows = input channels c = 8*g4 + 4*h + j < Cin
        if (kp == 0 && valid)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const uint32_t c = 8 * (r >> 2) + 4 * h + (r & 3);
                if (c < g.Cin) {
                    if (inp.mode == 0) {
                        grad_img[((size_t)im * g.Cin + c) * g.P + qo] = acc[r];
                    } else {   // through the normalisation and the clamp (gradient passes where 0 <= x <= 1, as torch.clamp)
                        const size_t e = ((size_t)im * g.P + qo) * g.Cin + c;
                        const float x = img[e];
                        float gy = acc[r] * inp.istd[c];          // d loss / d (distorted value)
                        if (inp.distort == kDistBlur) {           // the blur's adjoint needs the neighbours' gradients: k_dec_blur_adjoint finishes the job
                            grad_img[e] = gy;
                        } else {
                            if (inp.distort == kDistBrightness) {
                                const float f = inp.dparam[0], fx = f * clamp01(x);
                                gy = (fx >= 0.0f && fx <= 1.0f) ? gy * f : 0.0f;
                            }
                            grad_img[e] = (x >= 0.0f && x <= 1.0f) ? gy : 0.0f;
                        }
                    }
                }
            }
    }
}

// ----------------------------------------------------------------------------- layer 8 (64 -> 1) and the head

// grid (npair, B), 256 threads.  Prologue as kFwd (a_7 kept fp32 in LDS, [pos][64]); wave = 16 pixels, lane = input channel;
// the 64-channel sums of the wave's 16 pixels go through one LDS transpose instead of 16 x 6 dependent shuffles.
__global__ void __launch_bounds__(256) k_dec_l8_fwd(DecParams prm, DecWs ws, DecGeom g) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const uint32_t pair = blockIdx.x, im = blockIdx.y, t = threadIdx.x, W2 = g.W + 2;
    const PairRows pr = pair_rows(pair, g.P, g.W);
    const uint32_t total = pr.nrow * W2 * 16, img_off = im * g.P * kC;
    float *s_a = reinterpret_cast<float *>(smem_raw);          // [rows_max*W2][64]
    float *s_tab = s_a + (size_t)g.rows_max * W2 * kC;          // [4][64]
    float *s_red = s_tab + 4 * 64;                              // [1024]
    float *s_t = s_red + 1024;                                  // [4 waves][16][65]
    const int wave = t >> 6, lane = t & 63;

    // ---- requests: partials, the first batch of x_7, the layer's 9 weights
    Partials pv;
    load_partials(ws.stat[7], g.B * g.npair, pv);
    constexpr int kPre = 8;
    float4 v0[kPre];
    int q[kPre];
#pragma unroll
    for (int u = 0; u < kPre; ++u) {
        const uint32_t i = t + 256 * u;
        q[u] = i < total ? staged_pixel(i >> 4, W2, pr.py0, g) : -1;
        v0[u] = q[u] >= 0 ? *reinterpret_cast<const float4 *>(ws.x[7] + img_off + q[u] * kC + (i & 15) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float w8[9];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) w8[tap] = prm.w[8][lane * 9 + tap];   // [1][64][3][3]

    combine_fwd_stats(pv, g, s_red, s_tab, s_tab + 64);
    if (t < 64) {
        s_tab[128 + t] = prm.gamma[7][t];
        s_tab[192 + t] = prm.beta[7][t];
        if (pair == 0 && im == 0) {
            ws.minv[7][t] = s_tab[t];
            ws.minv[7][64 + t] = s_tab[64 + t];
        }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < kPre; ++u) {
        const uint32_t i = t + 256 * u, pos = i >> 4, cg = i & 15;
        if (i >= total) break;
        float xh[4], a[4] = {0.f, 0.f, 0.f, 0.f}, gp[4];
        if (q[u] >= 0) {
            bn_gelu4(v0[u], cg, s_tab, xh, a, gp);
            if ((uint32_t)q[u] >= pr.q0 && (uint32_t)q[u] <= pr.q1) {
                const uint32_t e = img_off + q[u] * kC + cg * 4;
                *reinterpret_cast<float4 *>(ws.xhat[7] + e) = make_float4(xh[0], xh[1], xh[2], xh[3]);
                *reinterpret_cast<float4 *>(ws.gprime[7] + e) = make_float4(gp[0], gp[1], gp[2], gp[3]);
                *reinterpret_cast<float4 *>(ws.act[7] + e) = make_float4(a[0], a[1], a[2], a[3]);
            }
        }
        *reinterpret_cast<float4 *>(s_a + (size_t)pos * kC + cg * 4) = make_float4(a[0], a[1], a[2], a[3]);
    }
    if (total > 256 * kPre) stage_forward<true>(ws.x[7], ws.xhat[7], ws.gprime[7], ws.act[7], im, pr, g, s_tab, nullptr, nullptr, s_a, 256 * kPre);
    __syncthreads();

    // ---- lane = input channel: its contribution to each of the wave's 16 pixels, then the channel sum by transpose
    float *tw = s_t + wave * 16 * 65;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const uint32_t qo = pr.q0 + 16 * wave + j;
        float v = 0.0f;
        if (qo < g.P) {
            const uint32_t py = qo / g.W, px = qo - py * g.W, hp0 = (py - pr.py0) * W2 + px;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) v += w8[tap] * s_a[(size_t)(hp0 + (tap / 3) * W2 + tap % 3) * kC + lane];
        }
        tw[j * 65 + lane] = v;
    }
    const int j = lane & 15, part = lane >> 4;
    float sum = 0.0f;
#pragma unroll
    for (int k = 0; k < 16; ++k) sum += tw[j * 65 + part * 16 + k];
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    float *s_x = s_red;   // 64 outputs of the pair
    if (part == 0) {
        const uint32_t qo = pr.q0 + 16 * wave + j;
        s_x[16 * wave + j] = sum;
        if (qo < g.P) ws.x[8][(size_t)im * g.P + qo] = sum;
    }
    __syncthreads();
    if (t < 64) {   // one wave: the pair's (sum, M2)
        const uint32_t n = pair_count(pair, g.P);
        const float v = t < n ? s_x[t] : 0.0f, s = wave_sum64(v);
        const float d = t < n ? v - s / (float)n : 0.0f, m2 = wave_sum64(d * d);
        if (t == 0) {
            ws.stat[8][((size_t)pair * g.B + im) * 2] = s;
            ws.stat[8][((size_t)pair * g.B + im) * 2 + 1] = m2;
        }
    }
}

// grid (B), 256 threads: BN + GELU of the single-channel x_8, average pool, Linear(1,1) (hidden_models.py:121-123,131-135).
// bce_seed (optional): the watermark loss's gradient with respect to this image's logit, written here so that the backward chain can start without a
// loss kernel in between: BCE-with-logits(temp * decoded, message) averaged over the B images, d/d decoded[im] = scale * (sigmoid(temp * decoded[im]) -
// message[im]) with scale = lambda_w * temp / B folded in by the caller (utils_wtmk_disen.py:441,641-644).
__global__ void __launch_bounds__(256) k_dec_head_fwd(DecParams prm, DecWs ws, DecGeom g, float *__restrict__ decoded, const float *__restrict__ bce_message,
                                                      float bce_temp, float bce_scale, float *__restrict__ bce_seed) {
    __shared__ float scratch[4];
    const uint32_t im = blockIdx.x, t = threadIdx.x, n = g.B * g.npair;
    const float N = (float)(g.B * g.P);
    float s = 0.0f;
    for (uint32_t i = t; i < n; i += 256) s += ws.stat[8][(size_t)i * 2];
    const float mean = block_sum256(s, scratch) / N;
    float m2 = 0.0f;
    for (uint32_t i = t; i < n; i += 256) {
        const float ni = (float)pair_count(i / g.B, g.P), d = ws.stat[8][(size_t)i * 2] / ni - mean;
        m2 += ws.stat[8][(size_t)i * 2 + 1] + ni * d * d;
    }
    const float inv = 1.0f / sqrtf(block_sum256(m2, scratch) / N + g.eps);
    const float gamma = prm.gamma[8][0], beta = prm.beta[8][0];
    float ps = 0.0f;
    for (uint32_t q = t; q < g.P; q += 256) {
        const size_t e = (size_t)im * g.P + q;
        const float xh = (ws.x[8][e] - mean) * inv;
        float a, gp;
        gelu_parts(xh * gamma + beta, a, gp);
        ws.xhat[8][e] = xh;
        ws.gprime[8][e] = gp;
        ps += a;
    }
    const float pool = block_sum256(ps, scratch) / (float)g.P;
    if (t == 0) {
        ws.pool[im] = pool;
        const float logit = pool * prm.lin_w[0] + prm.lin_b[0];
        decoded[im] = logit;
        if (bce_seed) bce_seed[im] = bce_scale * (1.0f / (1.0f + expf(-bce_temp * logit)) - bce_message[im]);
        if (im == 0) {
            ws.minv[8][0] = mean;
            ws.minv[8][1] = inv;
        }
    }
}

// grid (B), 256 threads: d(decoded) -> dz_8 = d(a_8) * GELU'(z_8) and its per-pair sums.
__global__ void __launch_bounds__(256) k_dec_head_bwd(const float *__restrict__ grad_dec, DecParams prm, DecWs ws, DecGeom g) {
    __shared__ float s_dz[kMaxP], s_xh[kMaxP];
    const uint32_t im = blockIdx.x, t = threadIdx.x;
    const float ga = grad_dec[im] * prm.lin_w[0] / (float)g.P;
    for (uint32_t q = t; q < g.P; q += 256) {
        const size_t e = (size_t)im * g.P + q;
        const float dz = ga * ws.gprime[8][e];
        ws.dz[8][e] = dz;
        s_dz[q] = dz;
        s_xh[q] = ws.xhat[8][e];
    }
    __syncthreads();
    for (uint32_t pair = t >> 6; pair < g.npair; pair += 4) {   // one wave per pair
        const uint32_t lane = t & 63, q = 64 * pair + lane;
        const float dz = q < g.P ? s_dz[q] : 0.0f, xh = q < g.P ? s_xh[q] : 0.0f;
        const float s1 = wave_sum64(dz), s2 = wave_sum64(dz * xh);
        if (lane == 0) {
            ws.bsum[8][((size_t)pair * g.B + im) * 2] = s1;
            ws.bsum[8][((size_t)pair * g.B + im) * 2 + 1] = s2;
        }
    }
}

// grid (npair, B), 256 threads: layer 8 backward.  dx_8 (one channel) and a_7 (the pair's rows, halo included) in LDS;
// wave = 16 pixels, lane = channel of layer 7:
//   G_7[q][ci] = sum_tap W8[ci][tap] dx_8[q - (tap - 1)]  ->  dz_7 = G_7 * GELU'(z_7) and its pair sums,
//   dW8[ci][tap] partial = sum_q dx_8[q] a_7[q + (tap - 1)][ci].
__global__ void __launch_bounds__(256) k_dec_l8_bwd(DecParams prm, DecWs ws, DecGeom g) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    __shared__ float scratch[4];
    const uint32_t pair = blockIdx.x, im = blockIdx.y, t = threadIdx.x, W2 = g.W + 2;
    const PairRows pr = pair_rows(pair, g.P, g.W);
    const uint32_t npos = pr.nrow * W2, n = g.B * g.npair, total = npos * 16, img_off = im * g.P * kC;
    float *s_a = reinterpret_cast<float *>(smem_raw);           // [rows_max*W2][64]
    float *s_dx = s_a + (size_t)g.rows_max * W2 * kC;            // [rows_max*W2]
    float *s_red = s_dx + g.rows_max * W2;                       // [4][64][11]
    const float N = (float)(g.B * g.P);
    const int wave = t >> 6, lane = t & 63;

    // ---- requests: sums of layer 8, dz_8/xhat_8 of the staged positions, a_7 rows, the wave's epilogue operands
    float s1 = 0.0f, s2 = 0.0f;
    for (uint32_t i = t; i < n; i += 256) {
        const float2 v = reinterpret_cast<const float2 *>(ws.bsum[8])[i];
        s1 += v.x;
        s2 += v.y;
    }
    float dzv[2], xhv[2];
    int qd[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {   // npos <= 512 covers every supported shape (checked on the host)
        const uint32_t pos = t + 256 * u;
        qd[u] = pos < npos ? staged_pixel(pos, W2, pr.py0, g) : -1;
        dzv[u] = qd[u] >= 0 ? ws.dz[8][(size_t)im * g.P + qd[u]] : 0.0f;
        xhv[u] = qd[u] >= 0 ? ws.xhat[8][(size_t)im * g.P + qd[u]] : 0.0f;
    }
    for (uint32_t base = t; base < total; base += 256 * kBatch) {
        float4 v[kBatch];
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
            const uint32_t i = base + 256 * u;
            const int q = i < total ? staged_pixel(i >> 4, W2, pr.py0, g) : -1;
            v[u] = q >= 0 ? *reinterpret_cast<const float4 *>(ws.act[7] + img_off + q * kC + (i & 15) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
            const uint32_t i = base + 256 * u;
            if (i < total) *reinterpret_cast<float4 *>(s_a + (size_t)(i >> 4) * kC + (i & 15) * 4) = v[u];
        }
    }
    float gp[16], xh[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const uint32_t q = pr.q0 + 16 * wave + j;
        gp[j] = q < g.P ? ws.gprime[7][img_off + q * kC + lane] : 0.0f;
        xh[j] = q < g.P ? ws.xhat[7][img_off + q * kC + lane] : 0.0f;
    }
    float w8[9], dw[9];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        w8[tap] = prm.w[8][lane * 9 + tap];
        dw[tap] = 0.0f;
    }
    const float S1 = block_sum256(s1, scratch), S2 = block_sum256(s2, scratch);
    const float k8 = prm.gamma[8][0] * ws.minv[8][1] / N;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const uint32_t pos = t + 256 * u;
        if (pos < npos) s_dx[pos] = qd[u] >= 0 ? k8 * (N * dzv[u] - S1 - xhv[u] * S2) : 0.0f;
    }
    __syncthreads();

    float b1 = 0.0f, b2 = 0.0f;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const uint32_t q = pr.q0 + 16 * wave + j;
        if (q >= g.P) break;
        const uint32_t py = q / g.W, px = q - py * g.W, hp0 = (py - pr.py0) * W2 + px, hp = hp0 + W2 + 1;   // hp: the pixel itself
        float G = 0.0f;
        const float dxq = s_dx[hp];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            G += w8[tap] * s_dx[hp + (1 - tap / 3) * (int)W2 + (1 - tap % 3)];
            dw[tap] += dxq * s_a[(size_t)(hp0 + (tap / 3) * W2 + tap % 3) * kC + lane];
        }
        const float dz = G * gp[j];
        ws.dz[7][img_off + q * kC + lane] = dz;
        b1 += dz;
        b2 += dz * xh[j];
    }
    float *r = s_red + (wave * 64 + lane) * 11;
    r[0] = b1;
    r[1] = b2;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) r[2 + tap] = dw[tap];
    __syncthreads();
    if (wave == 0) {
        float *po = ws.bsum[7] + (((size_t)pair * g.B + im) * kC + lane) * 2;
        po[0] = s_red[lane * 11] + s_red[(64 + lane) * 11] + s_red[(128 + lane) * 11] + s_red[(192 + lane) * 11];
        po[1] = s_red[lane * 11 + 1] + s_red[(64 + lane) * 11 + 1] + s_red[(128 + lane) * 11 + 1] + s_red[(192 + lane) * 11 + 1];
    }
    if (wave == 1) {
        float *po = ws.wpart8 + (((size_t)im * g.npair + pair) * kC + lane) * 9;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            float sacc = 0.0f;
            for (int w = 0; w < 4; ++w) sacc += s_red[(w * 64 + lane) * 11 + 2 + tap];
            po[tap] = sacc;
        }
    }
}

// ----------------------------------------------------------------------------- weight gradients of the 64 -> 64 layers

// One launch for all seven layers, after the dgrad chain (dz_l, the batch sums and a_{l-1} are all in the workspace by then):
// grid (nband, B, 7), 256 threads = 4 waves = the 4 (rb, cb) 32x32 blocks of dW[co][ci], each for all 9 taps.
// K runs over halo-linear pixel positions r = hr*RS + hc (RS a multiple of 8, so a tap's row shift keeps 16-byte alignment);
// dx^T [co][r] and a^T [ci][r] live in LDS as bf16 hi/lo planes.  The +-1 column shift of a tap is applied in registers:
// an aligned 8-element group plus one dword from each neighbour, funnel-shifted by one element (v_alignbyte).
__device__ inline bf16x8 shift_window(uint4 gq, uint32_t prev, uint32_t next, int tx) {
    uint4 o = gq;
    if (tx == 0) {
        o.x = __builtin_amdgcn_alignbyte(gq.x, prev, 2);
        o.y = __builtin_amdgcn_alignbyte(gq.y, gq.x, 2);
        o.z = __builtin_amdgcn_alignbyte(gq.z, gq.y, 2);
        o.w = __builtin_amdgcn_alignbyte(gq.w, gq.z, 2);
    } else if (tx == 2) {
        o.x = __builtin_amdgcn_alignbyte(gq.y, gq.x, 2);
        o.y = __builtin_amdgcn_alignbyte(gq.z, gq.y, 2);
        o.z = __builtin_amdgcn_alignbyte(gq.w, gq.z, 2);
        o.w = __builtin_amdgcn_alignbyte(next, gq.w, 2);
    }
    return *reinterpret_cast<bf16x8 *>(&o);
}

__device__ inline void put_transposed(char *hi, char *lo, uint32_t pitch, uint32_t cg, uint32_t k, float4 v) {
    const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        __bf16 vh, vl;
        split_bf16(vv[j], vh, vl);
        *reinterpret_cast<__bf16 *>(hi + (cg * 4 + j) * pitch + k * 2) = vh;
        *reinterpret_cast<__bf16 *>(lo + (cg * 4 + j) * pitch + k * 2) = vl;
    }
}

__global__ void __launch_bounds__(256) k_dec_wgrad(DecParams prm, DecWs ws, DecGeom g) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const uint32_t band = blockIdx.x, im = blockIdx.y, layer = 1 + blockIdx.z, t = threadIdx.x;
    const uint32_t Kpad = g.nks * 16, La = Kpad + 2 * g.RS + 16;
    char *dx_hi = smem_raw, *dx_lo = dx_hi + 64 * g.pd, *a_hi = dx_lo + 64 * g.pd, *a_lo = a_hi + 64 * g.pa;
    float *s_tab = reinterpret_cast<float *>(a_lo + 64 * g.pa);   // [3][64]
    float *s_red = s_tab + 3 * 64;                                 // [1024]
    const float N = (float)(g.B * g.P);
    const uint32_t row_lo = band * g.R, row_hi = min(g.H, row_lo + g.R);   // image rows of this band
    const int RS = (int)g.RS, rbase = (int)((row_lo + 1) * g.RS), abase = rbase - RS - 8;

    Partials pv;
    load_partials(ws.bsum[layer], g.B * g.npair, pv);
    // a^T does not depend on the batch sums
    for (uint32_t base = t; base < La * 16; base += 256 * kBatch) {
        float4 v[kBatch];
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
            const uint32_t i = base + 256 * u;
            const int r = abase + (int)(i >> 4), hr = r / RS, hc = r - hr * RS, iy = hr - 1, ix = hc - 1;
            const bool ok = i < La * 16 && r >= 0 && iy >= 0 && iy < (int)g.H && ix >= 0 && ix < (int)g.W;
            v[u] = ok ? *reinterpret_cast<const float4 *>(ws.act[layer - 1] + ((size_t)im * g.P + iy * g.W + ix) * kC + (i & 15) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
            const uint32_t i = base + 256 * u;
            if (i < La * 16) put_transposed(a_hi, a_lo, g.pa, i & 15, i >> 4, v[u]);
        }
    }
    combine_bwd_sums(pv, prm.gamma[layer], ws.minv[layer], g, s_red, s_tab);
    for (uint32_t base = t; base < Kpad * 16; base += 256 * kBatch) {
        float4 dv[kBatch], hv[kBatch];
        bool ok[kBatch];
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
            const uint32_t i = base + 256 * u;
            const int r = rbase + (int)(i >> 4), hr = r / RS, hc = r - hr * RS, iy = hr - 1, ix = hc - 1;
            ok[u] = i < Kpad * 16 && iy >= (int)row_lo && iy < (int)row_hi && ix >= 0 && ix < (int)g.W;
            const size_t e = ((size_t)im * g.P + (ok[u] ? iy * g.W + ix : 0)) * kC + (i & 15) * 4;
            dv[u] = ok[u] ? *reinterpret_cast<const float4 *>(ws.dz[layer] + e) : make_float4(0.f, 0.f, 0.f, 0.f);
            hv[u] = ok[u] ? *reinterpret_cast<const float4 *>(ws.xhat[layer] + e) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
            const uint32_t i = base + 256 * u;
            if (i < Kpad * 16) put_transposed(dx_hi, dx_lo, g.pd, i & 15, i >> 4, ok[u] ? bn_bwd4(dv[u], hv[u], i & 15, s_tab, N) : make_float4(0.f, 0.f, 0.f, 0.f));
        }
    }
    __syncthreads();

    const int wave = t >> 6, lane = t & 63, rb = wave & 1, cb = wave >> 1, l31 = lane & 31, h = lane >> 5;
    f32x16 acc[9];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[tap][r] = 0.0f;
    const char *pa_hi = a_hi + (32 * cb + l31) * g.pa, *pa_lo = a_lo + (32 * cb + l31) * g.pa;
    const char *pd_hi = dx_hi + (32 * rb + l31) * g.pd, *pd_lo = dx_lo + (32 * rb + l31) * g.pd;
    for (uint32_t ks = 0; ks < g.nks; ++ks) {
        const uint32_t k0 = 16 * ks + 8 * h;
        const bf16x8 dh = *reinterpret_cast<const bf16x8 *>(pd_hi + k0 * 2), dl = *reinterpret_cast<const bf16x8 *>(pd_lo + k0 * 2);
#pragma unroll
        for (int ty = 0; ty < 3; ++ty) {
            const uint32_t j0 = (k0 + 8 + ty * g.RS) * 2;   // byte offset of the aligned group for column shift 0
            const uint4 gh = *reinterpret_cast<const uint4 *>(pa_hi + j0), gl = *reinterpret_cast<const uint4 *>(pa_lo + j0);
            const uint32_t ph = *reinterpret_cast<const uint32_t *>(pa_hi + j0 - 4), pl = *reinterpret_cast<const uint32_t *>(pa_lo + j0 - 4);
            const uint32_t nh = *reinterpret_cast<const uint32_t *>(pa_hi + j0 + 16), nl = *reinterpret_cast<const uint32_t *>(pa_lo + j0 + 16);
#pragma unroll
            for (int tx = 0; tx < 3; ++tx) {
                const bf16x8 bh = shift_window(gh, ph, nh, tx), bl = shift_window(gl, pl, nl, tx);
                f32x16 c = acc[ty * 3 + tx];
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dl, bh, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dh, bl, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dh, bh, c, 0, 0, 0);
                acc[ty * 3 + tx] = c;
            }
        }
    }
    // lane (col = ci, h) holds rows co = 32*rb + 8*(r>>2) + 4*h + (r&3)
    float *po = ws.wpart + ((size_t)(layer - 1) * g.B * g.nband + (size_t)im * g.nband + band) * 9 * kC * kC;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int r = 0; r < 16; ++r) po[((size_t)tap * kC + 32 * rb + 8 * (r >> 2) + 4 * h + (r & 3)) * kC + 32 * cb + l31] = acc[tap][r];
}

// ----------------------------------------------------------------------------- layer 0 weight gradient (VALU)

// grid (npair, B), 256 threads: thread = (co, quarter); dW0[co][c][tap] partial over the pair's pixels.
__global__ void __launch_bounds__(256) k_dec_l0_wgrad(const float *__restrict__ img, DecInput inp, DecParams prm, DecWs ws, DecGeom g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const uint32_t pair = blockIdx.x, im = blockIdx.y, t = threadIdx.x;
    const uint32_t W2 = g.W + 2, HW2 = (g.H + 2) * W2;
    float *s_img = smem;                      // [Cin][HW2]
    float *s_tab = s_img + g.Cin * HW2;       // [3][64]
    float *s_red = s_tab + 3 * 64;            // [max(512, 4*64*9)]
    const uint32_t co = t & 63, sub = t >> 6, q0 = pair * 64 + sub * 16;
    float dzv[16], xhv[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const uint32_t q = q0 + j;
        const size_t e = ((size_t)im * g.P + (q < g.P ? q : 0)) * kC + co;
        dzv[j] = q < g.P ? ws.dz[0][e] : 0.0f;
        xhv[j] = q < g.P ? ws.xhat[0][e] : 0.0f;
    }
    for (uint32_t i = t; i < g.Cin * HW2; i += 256) {
        const uint32_t c = i / HW2, pos = i - c * HW2, hr = pos / W2, hc = pos - hr * W2;
        const bool in = hr >= 1 && hr <= g.H && hc >= 1 && hc <= g.W;
        s_img[i] = in ? input_value(img, inp, im, c, (hr - 1) * g.W + (hc - 1), g.Cin, g.P) : 0.0f;
    }
    Partials pv;
    load_partials(ws.bsum[0], g.B * g.npair, pv);
    combine_bwd_sums(pv, prm.gamma[0], ws.minv[0], g, s_red, s_tab);
    const float N = (float)(g.B * g.P);
    float dx[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) dx[j] = q0 + j < g.P ? s_tab[co] * (N * dzv[j] - s_tab[64 + co] - xhv[j] * s_tab[128 + co]) : 0.0f;
    for (uint32_t c = 0; c < g.Cin; ++c) {
        float acc[9];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) acc[tap] = 0.0f;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const uint32_t q = q0 + j;
            if (q < g.P) {
                const uint32_t py = q / g.W, px = q - py * g.W;
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) acc[tap] += dx[j] * s_img[c * HW2 + (py + tap / 3) * W2 + px + tap % 3];
            }
        }
        __syncthreads();
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) s_red[(sub * 64 + co) * 9 + tap] = acc[tap];
        __syncthreads();
        if (sub == 0) {
            float *po = ws.wpart0 + (((size_t)im * g.npair + pair) * kC + co) * 9 * g.Cin + c * 9;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap)
                po[tap] = s_red[co * 9 + tap] + s_red[(64 + co) * 9 + tap] + s_red[(128 + co) * 9 + tap] + s_red[(192 + co) * 9 + tap];
        }
    }
}

// ----------------------------------------------------------------------------- final reductions

// grid (64 co, 7 layers), 576 threads = (tap, ci): dW[co][ci][tap] = sum over the (image, band) partials.
__global__ void __launch_bounds__(576) k_dec_wreduce(DecWs ws, DecGrads gr, DecGeom g) {
    const uint32_t co = blockIdx.x, layer = 1 + blockIdx.y, tap = threadIdx.x >> 6, ci = threadIdx.x & 63;
    const uint32_t nwg = g.B * g.nband;
    const float *p = ws.wpart + (size_t)(layer - 1) * nwg * 9 * kC * kC + ((size_t)tap * kC + co) * kC + ci;
    float acc = 0.0f;
    for (uint32_t w0 = 0; w0 < nwg; w0 += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = w0 + u < nwg ? p[(size_t)(w0 + u) * 9 * kC * kC] : 0.0f;
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    gr.w[layer][((size_t)co * kC + ci) * 9 + tap] = acc;
}

// Sum of `n` partial vectors (stride `stride` floats) for 64 consecutive outputs starting at `e0`; 256 threads = (output, 4 groups).
__device__ inline float reduce64(const float *__restrict__ part, uint32_t n, size_t stride, uint32_t e0, uint32_t e_end, float *red) {
    const uint32_t o = threadIdx.x & 63, grp = threadIdx.x >> 6, e = e0 + o;
    float s = 0.0f;
    if (e < e_end)
        for (uint32_t base = grp; base < n; base += 32) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = base + 4 * u < n ? part[(size_t)(base + 4 * u) * stride + e] : 0.0f;
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
    __syncthreads();
    red[grp * 64 + o] = s;
    __syncthreads();
    return red[o] + red[64 + o] + red[128 + o] + red[192 + o];
}

// grid (16 + ceil(64*9*Cin/64) + 9 + 1), 256 threads.
//   blocks [0,16): layer l = b/2 < 8, which = b&1: dbeta_l (sum dz) / dgamma_l (sum dz*xhat)
//   then dW0 in chunks of 64 outputs, dW8 in 9 chunks, then one block for layer 8's BN and the Linear.
__global__ void __launch_bounds__(256) k_dec_sreduce(const float *__restrict__ grad_dec, DecWs ws, DecGrads gr, DecGeom g) {
    __shared__ float red[256];
    __shared__ float scratch[4];
    const uint32_t b = blockIdx.x, t = threadIdx.x, n = g.B * g.npair, K0 = kC * 9 * g.Cin, nb0 = ceil_div(K0, 64);
    if (b < 16) {
        const uint32_t l = b >> 1, which = b & 1;
        // partials are [i][c][2]: view as 128 interleaved outputs, take every other one
        const uint32_t o = t & 63, grp = t >> 6;
        float s = 0.0f;
        for (uint32_t base = grp; base < n; base += 32) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = base + 4 * u < n ? ws.bsum[l][((size_t)(base + 4 * u) * kC + o) * 2 + which] : 0.0f;
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        red[grp * 64 + o] = s;
        __syncthreads();
        if (grp == 0) (which ? gr.gamma[l] : gr.beta[l])[o] = red[o] + red[64 + o] + red[128 + o] + red[192 + o];
    } else if (b < 16 + nb0) {
        const uint32_t e0 = (b - 16) * 64;
        const float s = reduce64(ws.wpart0, n, K0, e0, K0, red);
        if (t < 64 && e0 + t < K0) gr.w[0][e0 + t] = s;
    } else if (b < 16 + nb0 + 9) {
        const uint32_t e0 = (b - 16 - nb0) * 64;
        const float s = reduce64(ws.wpart8, n, kC * 9, e0, kC * 9, red);
        if (t < 64) gr.w[8][e0 + t] = s;
    } else {
        float s1 = 0.0f, s2 = 0.0f, s3 = 0.0f, s4 = 0.0f;
        for (uint32_t i = t; i < n; i += 256) {
            s1 += ws.bsum[8][(size_t)i * 2];
            s2 += ws.bsum[8][(size_t)i * 2 + 1];
        }
        for (uint32_t i = t; i < g.B; i += 256) {
            s3 += grad_dec[i] * ws.pool[i];
            s4 += grad_dec[i];
        }
        s1 = block_sum256(s1, scratch);
        s2 = block_sum256(s2, scratch);
        s3 = block_sum256(s3, scratch);
        s4 = block_sum256(s4, scratch);
        if (t == 0) {
            gr.beta[8][0] = s1;
            gr.gamma[8][0] = s2;
            gr.lin_w[0] = s3;
            gr.lin_b[0] = s4;
        }
    }
}

// ----------------------------------------------------------------------------- host side

static inline uint32_t round_up(uint32_t v, uint32_t m) { return (v + m - 1) / m * m; }

constexpr size_t kLdsLimit = 160 * 1024;

static size_t conv_lds(const DecGeom &g) { return (size_t)2 * g.rows_max * (g.W + 2) * kPitch + (4 * 64 + 1024 + 2 * 2 * 32 * 33 + 2 * 3 * 16 * 64 + 2 * 32) * 4; }
static size_t l8_lds(const DecGeom &g) { return (size_t)g.rows_max * (g.W + 2) * kC * 4 + (4 * 64 + 1024 + 4 * 16 * 65) * 4; }
static size_t l8b_lds(const DecGeom &g) { return (size_t)g.rows_max * (g.W + 2) * (kC + 1) * 4 + 4 * 64 * 11 * 4; }
static size_t l0_lds(const DecGeom &g) { return ((size_t)g.Cin * g.rows_max * (g.W + 2) + 4 * 64 + (g.Cin == 3 ? 0 : (size_t)kC * 9 * g.Cin)) * 4; }
static size_t l0w_lds(const DecGeom &g) { return ((size_t)g.Cin * (g.H + 2) * (g.W + 2) + 3 * 64 + 4 * 64 * 9) * 4; }   // scratch 2304 >= 1024
static size_t wgrad_lds(const DecGeom &g) { return (size_t)128 * (g.pd + g.pa) + (3 * 64 + 1024) * 4; }

static bool make_geom(uint32_t B, uint32_t Cin, uint32_t H, uint32_t W, float eps, DecGeom &g) {
    g.B = B; g.H = H; g.W = W; g.P = H * W; g.Cin = Cin; g.eps = eps;
    g.ntile = ceil_div(g.P, 32);
    g.npair = ceil_div(g.ntile, 2);
    g.rows_max = 0;
    for (uint32_t p = 0; p < g.npair; ++p) g.rows_max = std::max(g.rows_max, pair_rows(p, g.P, W).nrow);
    g.magic_w2 = 65536 / (W + 2) + 1;
    for (uint32_t pos = 0; pos < g.rows_max * (W + 2); ++pos)
        if (((pos * g.magic_w2) >> 16) != pos / (W + 2)) return false;
    g.RS = round_up(W + 2, 8);
    // wgrad row bands: as few as LDS allows (every band writes a 147 KB partial of dW)
    for (g.nband = 1;; ++g.nband) {
        g.R = ceil_div(H, g.nband);
        g.nks = ceil_div(g.R * g.RS, 16);
        const uint32_t Kpad = g.nks * 16, La = Kpad + 2 * g.RS + 16;
        g.pd = round_up(Kpad * 2, 32) + 16;
        g.pa = round_up(La * 2, 32) + 16;
        if (wgrad_lds(g) <= kLdsLimit - 1024) break;
        if (g.nband >= H) return false;
    }
    g.nband = ceil_div(H, g.R);   // no empty bands
    if (B * g.npair > 8 * kPartMax) return false;        // batch partials are combined from registers
    if (g.rows_max * (W + 2) > 512) return false;        // k_dec_l8_bwd stages dx_8 with two positions per thread
    if ((uint64_t)B * g.P * kC >= (1u << 30)) return false;   // 32-bit element offsets
    return conv_lds(g) <= kLdsLimit && l8_lds(g) <= kLdsLimit && l8b_lds(g) <= kLdsLimit && l0_lds(g) <= 64 * 1024 && l0w_lds(g) <= 64 * 1024;
}

struct Carver {
    char *base;
    size_t off = 0;
    template <typename T>
    T *take(size_t count) {
        T *p = base ? reinterpret_cast<T *>(base + off) : nullptr;
        off += (count * sizeof(T) + 255) / 256 * 256;
        return p;
    }
};

static size_t carve(void *base, const DecGeom &g, DecWs &ws) {
    Carver c{reinterpret_cast<char *>(base)};
    const size_t BP = (size_t)g.B * g.P, np = (size_t)g.B * g.npair;
    for (int l = 0; l < kLayers; ++l) {
        const size_t C = l < 8 ? kC : 1;
        ws.x[l] = c.take<float>(BP * C);
        ws.xhat[l] = c.take<float>(BP * C);
        ws.gprime[l] = c.take<float>(BP * C);
        ws.dz[l] = c.take<float>(BP * C);
        if (l < 8) ws.act[l] = c.take<float>(BP * C);
        ws.stat[l] = c.take<float>(np * C * 2);
        ws.bsum[l] = c.take<float>(np * C * 2);
        ws.minv[l] = c.take<float>(2 * C);
    }
    ws.pool = c.take<float>(g.B);
    ws.wpart = c.take<float>((size_t)7 * g.B * g.nband * 9 * kC * kC);
    ws.wpart0 = c.take<float>((size_t)g.B * g.npair * kC * 9 * g.Cin);
    ws.wpart8 = c.take<float>((size_t)g.B * g.npair * kC * 9);
    ws.packed = c.take<uint4>(kSets * kSetU4);
    return c.off;
}

static int load_params(const float *const *params, DecParams &p) {
    for (int l = 0; l < kLayers; ++l) {
        p.w[l] = params[3 * l];
        p.gamma[l] = params[3 * l + 1];
        p.beta[l] = params[3 * l + 2];
        if (!p.w[l] || !p.gamma[l] || !p.beta[l]) return 1;
    }
    p.lin_w = params[27];
    p.lin_b = params[28];
    return !(p.lin_w && p.lin_b);
}

// Dynamic LDS above 64 KB has to be allowed per kernel; remembered per kernel so the attribute call happens once per size.
template <typename K>
static int allow_lds(K kernel, size_t bytes, size_t &allowed) {
    if (bytes <= allowed) return 0;
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) return 1;
    allowed = bytes;
    return 0;
}
static int allow_all_lds(const DecGeom &g) {
    static size_t a[6] = {0, 0, 0, 0, 0, 0};
    return allow_lds(k_dec_conv<kFwd, 512>, conv_lds(g), a[0]) | allow_lds(k_dec_conv<kDgrad, 512>, conv_lds(g), a[1]) |
           allow_lds(k_dec_conv<kDgradImg, 512>, conv_lds(g), a[2]) | allow_lds(k_dec_l8_fwd, l8_lds(g), a[3]) |
           allow_lds(k_dec_l8_bwd, l8b_lds(g), a[4]) | allow_lds(k_dec_wgrad, wgrad_lds(g), a[5]);
}

// Threads per conv workgroup: 512 = four MFMA waves + four helper waves for the request / statistics / prologue phases.
template <int MODE>
static void launch_conv(dim3 grid, size_t lds, hipStream_t s, int layer, const DecParams &prm, const DecWs &ws, const DecGeom &g, float *grad_img,
                        const float *img, const DecInput &inp) {
    k_dec_conv<MODE, 512><<<grid, 512, lds, s>>>(layer, prm, ws, g, grad_img, img, inp);
}

}  // namespace nsig

using namespace nsig;

#ifdef NSIG_DEC_TIMING
NSIG_EXPORT int dec_timing_stamps(unsigned long long *out48) {
    return hipMemcpyFromSymbol(out48, HIP_SYMBOL(g_dec_stamps), sizeof(unsigned long long) * 48) == hipSuccess ? 0 : 1;
}
#endif

NSIG_EXPORT size_t dec_workspace_bytes(uint32_t B, uint32_t Cin, uint32_t H, uint32_t W) {
    DecGeom g;
    DecWs ws;
    if (B == 0 || Cin == 0 || Cin > kMaxCin || H == 0 || W == 0 || (uint64_t)H * W > kMaxP || (uint64_t)B * H * W < 2 || !make_geom(B, Cin, H, W, 0.0f, g)) return 0;
    return carve(nullptr, g, ws);
}

// grad_img (through the clamp) = clamp mask x the transposed blur of gy: gx[p] = sum over the (q, tap) whose reflected source is p of k[tap] * gy[q].
// One thread per (image, pixel, channel); the 3 x 3 outputs q around p are visited and each one's taps re-reflected.
__global__ void __launch_bounds__(256) k_dec_blur_adjoint(const float *__restrict__ gy, const float *__restrict__ img, const float *__restrict__ dparam, uint32_t B,
                                                          uint32_t H, uint32_t W, uint32_t Cin, float *__restrict__ grad_img) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * H * W * Cin) return;
    const uint32_t c = i % Cin, pix = (i / Cin) % (H * W), im = i / (Cin * H * W);
    const int y = (int)(pix / W), x = (int)(pix % W);
    float a, b;
    blur_taps(dparam[0], a, b);
    float acc = 0.0f;
    for (int qy = y - 2; qy <= y + 2; ++qy) {
        if (qy < 0 || qy >= (int)H) continue;
        float wy = 0.0f;          // total weight with which output row qy reads input row y (reflection can map two taps onto it)
        for (int dy = -1; dy <= 1; ++dy)
            if (reflect1(qy + dy, (int)H) == y) wy += dy == 0 ? b : a;
        if (wy == 0.0f) continue;
        for (int qx = x - 2; qx <= x + 2; ++qx) {
            if (qx < 0 || qx >= (int)W) continue;
            float wx = 0.0f;
            for (int dx = -1; dx <= 1; ++dx)
                if (reflect1(qx + dx, (int)W) == x) wx += dx == 0 ? b : a;
            if (wx != 0.0f) acc += wy * wx * gy[((size_t)im * H * W + (size_t)qy * W + qx) * Cin + c];
        }
    }
    const float v = img[i];
    grad_img[i] = (v >= 0.0f && v <= 1.0f) ? acc : 0.0f;
}

// The step's random draws, counter-based: everything is a function of (seed, *step, element) -- a replayed graph draws fresh values every
// step without a host-side generator.  param[0]: brightness factor U[0.5, 1.5] / blur sigma U[0.01, 0.5]; noise[i] ~ N(0, 0.1) (Box-Muller).
__device__ inline uint64_t mix64(uint64_t z) {      // splitmix64's finaliser
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__global__ void __launch_bounds__(256) k_distort_draw(uint32_t kind, uint64_t seed, const uint32_t *__restrict__ step, uint32_t n, float *__restrict__ param,
                                                      float *__restrict__ noise) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    const uint64_t key = mix64(seed ^ (0x9E3779B97F4A7C15ull * ((uint64_t)(step ? step[0] : 0u) + 1ull)));
    if (i == 0 && param && kind != kDistRotation) {
        const float u = (float)(mix64(key ^ 0xFFFFFFFFFFFFFFFFull) >> 40) * (1.0f / 16777216.0f);      // [0, 1)
        param[0] = kind == kDistBrightness ? 0.5f + u : (kind == kDistBlur ? 0.01f + 0.49f * u : 0.0f);
    }
    if (kind == kDistRotation && i < n) {       // one angle in [-30, 30) degrees per image, stored as (cos, sin): param[2 i], param[2 i + 1]
        const float u = (float)(mix64(key ^ (0xA0761D6478BD642Full * ((uint64_t)i + 1ull))) >> 40) * (1.0f / 16777216.0f);
        const float a = (60.0f * u - 30.0f) * 0.0174532925199432958f;
        param[2 * i] = cosf(a);
        param[2 * i + 1] = sinf(a);
    }
    if (kind == kDistNoise && noise && i < n) {
        const uint64_t h = mix64(key + 0xD1B54A32D192ED03ull * ((uint64_t)i + 1ull));
        const float u1 = ((float)(uint32_t)(h >> 40) + 1.0f) * (1.0f / 16777216.0f);                   // (0, 1]
        const float u2 = (float)(uint32_t)((h >> 8) & 0xFFFFFFu) * (1.0f / 16777216.0f);              // [0, 1)
        noise[i] = 0.316227766016837933f * sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958648f * u2);  // sigma = sqrt(0.1)
    }
}

// the layer alone, [B][H][W][C] -> [B][H][W][C] (decoder shapes the fused chain does not implement; tests)
__global__ void __launch_bounds__(256) k_distort_fwd(const float *__restrict__ img, DecInput in, uint32_t B, uint32_t Cin, float *__restrict__ out) {
    const uint32_t P = in.H * in.W, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * P * Cin) return;
    out[i] = distorted_value(img, in, i / (P * Cin), i % Cin, (i / Cin) % P, Cin, P);
}
__global__ void __launch_bounds__(256) k_distort_bwd(const float *__restrict__ gy, const float *__restrict__ img, DecInput in, uint32_t B, uint32_t Cin,
                                                     float *__restrict__ gx) {
    const uint32_t P = in.H * in.W, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * P * Cin) return;
    const float x = img[i];
    float g = gy[i];
    if (in.distort == kDistBrightness) {
        const float f = in.dparam[0], fx = f * clamp01(x);
        g = (fx >= 0.0f && fx <= 1.0f) ? g * f : 0.0f;
    }
    gx[i] = (x >= 0.0f && x <= 1.0f) ? g : 0.0f;
}

// ---- the two kinds that change the sampling geometry (rotation, scaling): kernels of their own in front of / behind the decoder, which then reads their
// output as an ordinary rendered image (values in [0, 1]: its clamp is the identity and passes every gradient).
//
// rotation (torchvision RandomRotation((-30, 30)) per image: nearest-neighbour resampling about the centre, zeros outside, same size): the source of output
// pixel (x, y), measured from the centre ((W - 1) / 2, (H - 1) / 2), is round-half-even(cos * x - sin * y, sin * x + cos * y).  Every product and sum is
// rounded on its own (no fused multiply-add): the host-side statement of the same map (distortion.rotate_nearest) picks the same pixels bit for bit.
__device__ inline bool rotation_source(float c, float s, int x, int y, int H, int W, int &ix, int &iy) {
    const float cx = 0.5f * (float)(W - 1), cy = 0.5f * (float)(H - 1);
    const float xs = __fsub_rn((float)x, cx), ys = __fsub_rn((float)y, cy);
    const float sx = __fadd_rn(__fsub_rn(__fmul_rn(c, xs), __fmul_rn(s, ys)), cx);
    const float sy = __fadd_rn(__fadd_rn(__fmul_rn(s, xs), __fmul_rn(c, ys)), cy);
    ix = (int)rintf(sx);
    iy = (int)rintf(sy);
    return ix >= 0 && ix < W && iy >= 0 && iy < H;
}
// scaling (F.interpolate(image [3, H, W], scale_factor = sf, mode = 'linear'): 1-d, along W only, W_out = floor(W * sf), align_corners = False with the
// given factor kept for the coordinates): source position max(0, (xd + 0.5) / sf - 0.5), its two neighbours blended linearly -- except where W_out == W,
// which the operator special-cases as a plain copy whatever the factor (ATen upsample_linear1d: "special case: just copy")
__device__ inline void scaling_source(float rscale, int xd, int W, int Wo, int &x0, int &x1, float &l0, float &l1) {
    if (Wo == W) {
        x0 = x1 = xd;
        l0 = 1.0f;
        l1 = 0.0f;
        return;
    }
    float src = rscale * ((float)xd + 0.5f) - 0.5f;
    src = src < 0.0f ? 0.0f : src;
    x0 = (int)src;
    x0 = x0 > W - 1 ? W - 1 : x0;
    x1 = x0 + (x0 < W - 1 ? 1 : 0);
    l1 = src - (float)x0;
    l0 = 1.0f - l1;
}
__device__ inline float scaling_rscale(const float *__restrict__ param) { return (float)(1.0 / (double)param[0]); }

// out [B][H][Wo][C] = the layer on clamp(img [B][H][W][C]); clamped_out (optional) = clamp(img)
__global__ void __launch_bounds__(256) k_distort_geom_fwd(const float *__restrict__ img, uint32_t kind, const float *__restrict__ param, uint32_t B, uint32_t H,
                                                          uint32_t W, uint32_t Wo, uint32_t C, float *__restrict__ out, float *__restrict__ clamped_out) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (clamped_out && i < B * H * W * C) clamped_out[i] = clamp01(img[i]);
    if (i >= B * H * Wo * C) return;
    const uint32_t c = i % C, x = (i / C) % Wo, y = (i / (C * Wo)) % H, im = i / (C * Wo * H);
    const float *base = img + (size_t)im * H * W * C + c;
    if (kind == kDistRotation) {
        int ix, iy;
        out[i] = rotation_source(param[2 * im], param[2 * im + 1], (int)x, (int)y, (int)H, (int)W, ix, iy) ? clamp01(base[((size_t)iy * W + ix) * C]) : 0.0f;
    } else {
        int x0, x1;
        float l0, l1;
        scaling_source(scaling_rscale(param), (int)x, (int)W, (int)Wo, x0, x1, l0, l1);
        out[i] = l0 * clamp01(base[((size_t)y * W + x0) * C]) + l1 * clamp01(base[((size_t)y * W + x1) * C]);
    }
}
// grad_img [B][H][W][C] = the clamp's mask x the layer's adjoint applied to gy [B][H][Wo][C].  Gather form (one thread per source element, fixed summation
// order: deterministic).  rotation: an output pixel q reads source p only if |R^-1 q - p| <= 1/2 per axis, hence |q - R p| < 0.71: q lies in the 3 x 3
// neighbourhood of round(R p) -- each candidate's source is re-derived with the forward's own arithmetic.  scaling: every output column of the row is tried.
__global__ void __launch_bounds__(256) k_distort_geom_bwd(const float *__restrict__ gy, const float *__restrict__ img, uint32_t kind, const float *__restrict__ param,
                                                          uint32_t B, uint32_t H, uint32_t W, uint32_t Wo, uint32_t C, float *__restrict__ gx) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * H * W * C) return;
    const uint32_t c = i % C, x = (i / C) % W, y = (i / (C * W)) % H, im = i / (C * W * H);
    const float v = img[i];
    if (!(v >= 0.0f && v <= 1.0f)) {
        gx[i] = 0.0f;
        return;
    }
    const float *g = gy + (size_t)im * H * Wo * C + c;
    float acc = 0.0f;
    if (kind == kDistRotation) {
        const float cs = param[2 * im], sn = param[2 * im + 1];
        const float cx = 0.5f * (float)(W - 1), cy = 0.5f * (float)(H - 1), u = (float)x - cx, w = (float)y - cy;
        const int qx0 = (int)rintf(cs * u + sn * w + cx), qy0 = (int)rintf(-sn * u + cs * w + cy);
        for (int qy = qy0 - 1; qy <= qy0 + 1; ++qy)
            for (int qx = qx0 - 1; qx <= qx0 + 1; ++qx) {
                if (qx < 0 || qx >= (int)W || qy < 0 || qy >= (int)H) continue;
                int ix, iy;
                if (rotation_source(cs, sn, qx, qy, (int)H, (int)W, ix, iy) && ix == (int)x && iy == (int)y) acc += g[((size_t)qy * Wo + qx) * C];
            }
    } else {
        const float rs = scaling_rscale(param);
        for (int xd = 0; xd < (int)Wo; ++xd) {
            int x0, x1;
            float l0, l1;
            scaling_source(rs, xd, (int)W, (int)Wo, x0, x1, l0, l1);
            const float gq = g[((size_t)y * Wo + xd) * C];
            if (x0 == (int)x) acc += l0 * gq;
            if (x1 == (int)x) acc += l1 * gq;
        }
    }
    gx[i] = acc;
}

static int make_input(uint32_t mode, const float *mean, const float *stdev, uint32_t Cin, DecInput &in) {
    in.mode = mode;
    in.distort = kDistNone;
    in.H = in.W = 0;
    in.dparam = in.dnoise = nullptr;
    for (int c = 0; c < 8; ++c) in.mean[c] = 0.0f, in.istd[c] = 1.0f;
    if (mode == 0) return 0;
    if (mode != 1 || !mean || !stdev || Cin > 8) return 1;
    for (uint32_t c = 0; c < Cin; ++c) {
        if (!(stdev[c] > 0.0f)) return 1;
        in.mean[c] = mean[c];
        in.istd[c] = 1.0f / stdev[c];
    }
    return 0;
}

static int set_distortion(DecInput &in, uint32_t distortion, const float *param, const float *noise, uint32_t H, uint32_t W) {
    if (distortion == kDistNone) return 0;
    if (in.mode != 1 || distortion > kDistBlur) return 1;
    if (distortion == kDistNoise ? noise == nullptr : param == nullptr) return 1;
    if (distortion == kDistBlur && (H < 2 || W < 2)) return 1;      // reflect padding by one needs two pixels
    in.distort = distortion;
    in.H = H;
    in.W = W;
    in.dparam = param;
    in.dnoise = noise;
    return 0;
}

static int dec_forward_impl(const float *img, uint32_t input_mode, const float *mean_host, const float *std_host, const float *const *params,
                            uint32_t B, uint32_t Cin, uint32_t H, uint32_t W, float eps, void *workspace, float *decoded, float *clamped_out,
                            uint32_t distortion, const float *dist_param, const float *dist_noise, const float *bce_message, float bce_temp, float bce_scale,
                            float *bce_seed, nsig_stream_t stream) {
    NSIG_REQUIRE(img && params && workspace && decoded, "dec_forward: null pointer");
    NSIG_REQUIRE(bce_seed == nullptr || bce_message != nullptr, "dec_forward_train: bce_seed needs bce_message");
    DecInput inp;
    NSIG_REQUIRE(make_input(input_mode, mean_host, std_host, Cin, inp) == 0, "dec_forward: input_mode is 0, or 1 with Cin <= 8 means and positive stds");
    NSIG_REQUIRE(set_distortion(inp, distortion, dist_param, dist_noise, H, W) == 0,
                 "dec_forward_train: distortion is 0..3, needs input_mode 1, its device parameter (2, 3) or noise tensor (1), and H, W >= 2 for the blur");
    DecGeom g;
    NSIG_REQUIRE(B >= 1 && Cin >= 1 && Cin <= kMaxCin && H >= 1 && W >= 1 && (uint64_t)H * W <= kMaxP && (uint64_t)B * H * W > 1 &&
                     make_geom(B, Cin, H, W, eps, g),
                 "dec_forward: unsupported image shape %ux%ux%ux%u (see dec_workspace_bytes)", B, Cin, H, W);
    DecParams prm;
    NSIG_REQUIRE(load_params(params, prm) == 0, "dec_forward: params must hold 29 device pointers");
    NSIG_REQUIRE(allow_all_lds(g) == 0, "dec_forward: could not raise the dynamic LDS limit");
    DecWs ws;
    carve(workspace, g, ws);
    hipStream_t s = as_stream(stream);
    const dim3 grid(g.npair, B);
    k_dec_pack<<<kSets * 2 * kKS, 64, 0, s>>>(prm, ws.packed, Cin);
    if (Cin == 3)
        k_dec_l0_fwd<3><<<grid, 256, l0_lds(g), s>>>(img, inp, clamped_out, prm, ws, g);
    else
        k_dec_l0_fwd<0><<<grid, 256, l0_lds(g), s>>>(img, inp, clamped_out, prm, ws, g);
    for (int l = 1; l <= 7; ++l) launch_conv<kFwd>(dim3(g.npair, B, 2), conv_lds(g), s, l, prm, ws, g, nullptr, nullptr, inp);
    k_dec_l8_fwd<<<grid, 256, l8_lds(g), s>>>(prm, ws, g);
    k_dec_head_fwd<<<B, 256, 0, s>>>(prm, ws, g, decoded, bce_message, bce_temp, bce_scale, bce_seed);
    return check_launch("dec_forward");
}

NSIG_EXPORT int dec_forward(const float *img, uint32_t input_mode, const float *mean_host, const float *std_host, const float *const *params,
                            uint32_t B, uint32_t Cin, uint32_t H, uint32_t W, float eps, void *workspace, float *decoded, float *clamped_out,
                            nsig_stream_t stream) {
    return dec_forward_impl(img, input_mode, mean_host, std_host, params, B, Cin, H, W, eps, workspace, decoded, clamped_out, kDistNone, nullptr, nullptr, nullptr,
                            0.0f, 0.0f, nullptr, stream);
}

NSIG_EXPORT int dec_forward_train(const float *img, const float *mean_host, const float *std_host, const float *const *params, uint32_t B, uint32_t Cin,
                                  uint32_t H, uint32_t W, float eps, void *workspace, float *decoded, float *clamped_out, uint32_t distortion,
                                  const float *dist_param, const float *dist_noise, const float *bce_message, float bce_temp, float bce_scale, float *bce_seed,
                                  nsig_stream_t stream) {
    return dec_forward_impl(img, 1, mean_host, std_host, params, B, Cin, H, W, eps, workspace, decoded, clamped_out, distortion, dist_param, dist_noise,
                            bce_message, bce_temp, bce_scale, bce_seed, stream);
}

static int dec_backward_impl(const float *grad_decoded, const float *img, uint32_t input_mode, const float *mean_host, const float *std_host,
                             const float *const *params, uint32_t B, uint32_t Cin, uint32_t H, uint32_t W, void *workspace, float *const *grads,
                             float *grad_img, uint32_t distortion, const float *dist_param, const float *dist_noise, float *grad_scratch,
                             nsig_stream_t stream, nsig_stream_t weights_stream) {
    NSIG_REQUIRE(grad_decoded && img && params && workspace && grads && grad_img, "dec_backward: null pointer");
    DecInput inp;
    NSIG_REQUIRE(make_input(input_mode, mean_host, std_host, Cin, inp) == 0, "dec_backward: input_mode is 0, or 1 with Cin <= 8 means and positive stds");
    NSIG_REQUIRE(set_distortion(inp, distortion, dist_param, dist_noise, H, W) == 0 && (distortion != kDistBlur || grad_scratch != nullptr),
                 "dec_backward_train: distortion is 0..3, needs input_mode 1, its device parameter / noise tensor, and for the blur a scratch image");
    DecGeom g;
    NSIG_REQUIRE(B >= 1 && Cin >= 1 && Cin <= kMaxCin && H >= 1 && W >= 1 && (uint64_t)H * W <= kMaxP && make_geom(B, Cin, H, W, 0.0f, g),
                 "dec_backward: unsupported image shape");
    DecParams prm;
    NSIG_REQUIRE(load_params(params, prm) == 0, "dec_backward: params must hold 29 device pointers");
    DecGrads gr;
    for (int l = 0; l < kLayers; ++l) {
        gr.w[l] = grads[3 * l];
        gr.gamma[l] = grads[3 * l + 1];
        gr.beta[l] = grads[3 * l + 2];
        NSIG_REQUIRE(gr.w[l] && gr.gamma[l] && gr.beta[l], "dec_backward: grads must hold 29 device pointers");
    }
    gr.lin_w = grads[27];
    gr.lin_b = grads[28];
    NSIG_REQUIRE(gr.lin_w && gr.lin_b, "dec_backward: grads must hold 29 device pointers");
    NSIG_REQUIRE(allow_all_lds(g) == 0, "dec_backward: could not raise the dynamic LDS limit");
    DecWs ws;
    carve(workspace, g, ws);
    hipStream_t s = as_stream(stream);
    const dim3 grid(g.npair, B);
    k_dec_head_bwd<<<B, 256, 0, s>>>(grad_decoded, prm, ws, g);
    k_dec_l8_bwd<<<grid, 256, l8b_lds(g), s>>>(prm, ws, g);
    for (int l = 7; l >= 1; --l) launch_conv<kDgrad>(dim3(g.npair, B, 2), conv_lds(g), s, l, prm, ws, g, nullptr, nullptr, inp);
    if (inp.distort == kDistBlur) {     // the epilogue leaves d loss / d (blurred image) in the scratch image; the blur's adjoint and the clamp mask follow
        launch_conv<kDgradImg>(grid, conv_lds(g), s, 0, prm, ws, g, grad_scratch, img, inp);
        k_dec_blur_adjoint<<<ceil_div(B * H * W * Cin, 256u), 256, 0, s>>>(grad_scratch, img, inp.dparam, B, H, W, Cin, grad_img);
    } else {
        launch_conv<kDgradImg>(grid, conv_lds(g), s, 0, prm, ws, g, grad_img, img, inp);
    }
    // The parameter gradients are not needed before the optimiser; what waits on this function is the image gradient (the block
    // render's backward).  On request they are queued on a second stream, ordered after the data-gradient chain by an event.
    hipStream_t sw = as_stream(weights_stream);
    if (sw != s) {
        static hipEvent_t chain_done = nullptr;
        if (chain_done == nullptr && hipEventCreateWithFlags(&chain_done, hipEventDisableTiming) != hipSuccess) {
            set_error("dec_backward: hipEventCreateWithFlags failed");
            return NSIG_ERR_LAUNCH;
        }
        if (hipEventRecord(chain_done, s) != hipSuccess || hipStreamWaitEvent(sw, chain_done, 0) != hipSuccess) {
            set_error("dec_backward: could not order the weight-gradient stream after the data-gradient chain");
            return NSIG_ERR_LAUNCH;
        }
    }
    k_dec_wgrad<<<dim3(g.nband, B, 7), 256, wgrad_lds(g), sw>>>(prm, ws, g);
    k_dec_l0_wgrad<<<grid, 256, l0w_lds(g), sw>>>(img, inp, prm, ws, g);
    k_dec_wreduce<<<dim3(kC, 7), 576, 0, sw>>>(ws, gr, g);
    k_dec_sreduce<<<16 + ceil_div(kC * 9 * Cin, 64) + 9 + 1, 256, 0, sw>>>(grad_decoded, ws, gr, g);
    return check_launch("dec_backward");
}

NSIG_EXPORT int dec_backward(const float *grad_decoded, const float *img, uint32_t input_mode, const float *mean_host, const float *std_host,
                             const float *const *params, uint32_t B, uint32_t Cin, uint32_t H, uint32_t W, void *workspace, float *const *grads,
                             float *grad_img, nsig_stream_t stream, nsig_stream_t weights_stream) {
    return dec_backward_impl(grad_decoded, img, input_mode, mean_host, std_host, params, B, Cin, H, W, workspace, grads, grad_img, kDistNone, nullptr, nullptr,
                             nullptr, stream, weights_stream);
}

NSIG_EXPORT int dec_backward_train(const float *grad_decoded, const float *img, const float *mean_host, const float *std_host, const float *const *params,
                                       uint32_t B, uint32_t Cin, uint32_t H, uint32_t W, void *workspace, float *const *grads, float *grad_img,
                                       uint32_t distortion, const float *dist_param, const float *dist_noise, float *grad_scratch, nsig_stream_t stream,
                                       nsig_stream_t weights_stream) {
    return dec_backward_impl(grad_decoded, img, 1, mean_host, std_host, params, B, Cin, H, W, workspace, grads, grad_img, distortion, dist_param, dist_noise,
                             grad_scratch, stream, weights_stream);
}

NSIG_EXPORT int wm_distort_draw(uint32_t distortion, uint64_t seed, const uint32_t *step_counter, uint32_t n_noise, float *param_out, float *noise_out,
                                nsig_stream_t stream) {
    NSIG_REQUIRE(distortion >= kDistNoise && distortion <= kDistRotation,
                 "wm_distort_draw: distortion is 1 (noise), 2 (brightness), 3 (blurring) or 4 (rotation); the scaling factor decides a tensor shape: the host draws it");
    NSIG_REQUIRE(distortion == kDistNoise ? (noise_out != nullptr && n_noise >= 1) : param_out != nullptr, "wm_distort_draw: missing output buffer");
    NSIG_REQUIRE(distortion != kDistRotation || n_noise >= 1, "wm_distort_draw: rotation draws one angle per image (n_noise = the number of images)");
    const uint32_t n = (distortion == kDistNoise || distortion == kDistRotation) ? n_noise : 1u;
    k_distort_draw<<<ceil_div(n, 256u), 256, 0, as_stream(stream)>>>(distortion, seed, step_counter, n, param_out, noise_out);
    return check_launch("wm_distort_draw");
}

static int standalone_input(DecInput &in, uint32_t distortion, const float *param, const float *noise, uint32_t H, uint32_t W) {
    in.mode = 1;
    for (int c = 0; c < 8; ++c) in.mean[c] = 0.0f, in.istd[c] = 1.0f;
    in.distort = kDistNone;
    in.H = H;
    in.W = W;
    in.dparam = in.dnoise = nullptr;
    return set_distortion(in, distortion, param, noise, H, W);
}

NSIG_EXPORT int wm_distort_fwd(const float *img, uint32_t B, uint32_t H, uint32_t W, uint32_t C, uint32_t distortion, const float *dist_param,
                               const float *dist_noise, float *out, nsig_stream_t stream) {
    NSIG_REQUIRE(img && out && B >= 1 && H >= 1 && W >= 1 && C >= 1 && (uint64_t)B * H * W * C < (1ull << 31), "wm_distort_fwd: bad arguments");
    DecInput in;
    NSIG_REQUIRE(standalone_input(in, distortion, dist_param, dist_noise, H, W) == 0, "wm_distort_fwd: distortion is 0..3 with its device parameter / noise tensor (blur: H, W >= 2)");
    k_distort_fwd<<<ceil_div(B * H * W * C, 256u), 256, 0, as_stream(stream)>>>(img, in, B, C, out);
    return check_launch("wm_distort_fwd");
}

NSIG_EXPORT int wm_distort_bwd(const float *grad_out, const float *img, uint32_t B, uint32_t H, uint32_t W, uint32_t C, uint32_t distortion,
                               const float *dist_param, const float *dist_noise, float *grad_img, nsig_stream_t stream) {
    NSIG_REQUIRE(grad_out && img && grad_img && B >= 1 && H >= 1 && W >= 1 && C >= 1 && (uint64_t)B * H * W * C < (1ull << 31), "wm_distort_bwd: bad arguments");
    DecInput in;
    NSIG_REQUIRE(standalone_input(in, distortion, dist_param, dist_noise, H, W) == 0, "wm_distort_bwd: distortion is 0..3 with its device parameter / noise tensor (blur: H, W >= 2)");
    if (distortion == kDistBlur)
        k_dec_blur_adjoint<<<ceil_div(B * H * W * C, 256u), 256, 0, as_stream(stream)>>>(grad_out, img, dist_param, B, H, W, C, grad_img);
    else
        k_distort_bwd<<<ceil_div(B * H * W * C, 256u), 256, 0, as_stream(stream)>>>(grad_out, img, in, B, C, grad_img);
    return check_launch("wm_distort_bwd");
}

NSIG_EXPORT int wm_distort_geom_fwd(const float *img, uint32_t B, uint32_t H, uint32_t W, uint32_t C, uint32_t distortion, const float *dist_param, uint32_t W_out,
                                    float *out, float *clamped_out, nsig_stream_t stream) {
    NSIG_REQUIRE(img && out && dist_param && B >= 1 && H >= 1 && W >= 1 && C >= 1 && W_out >= 1 && (uint64_t)B * H * (W > W_out ? W : W_out) * C < (1ull << 31),
                 "wm_distort_geom_fwd: bad arguments");
    NSIG_REQUIRE((distortion == kDistRotation && W_out == W) || distortion == kDistScaling,
                 "wm_distort_geom_fwd: distortion is 4 (rotation: W_out = W, dist_param = (cos, sin) per image) or 5 (scaling: dist_param[0] = the factor, W_out = floor(W * factor))");
    k_distort_geom_fwd<<<ceil_div(B * H * (W > W_out ? W : W_out) * C, 256u), 256, 0, as_stream(stream)>>>(img, distortion, dist_param, B, H, W, W_out, C, out, clamped_out);
    return check_launch("wm_distort_geom_fwd");
}

NSIG_EXPORT int wm_distort_geom_bwd(const float *grad_out, const float *img, uint32_t B, uint32_t H, uint32_t W, uint32_t C, uint32_t distortion,
                                    const float *dist_param, uint32_t W_out, float *grad_img, nsig_stream_t stream) {
    NSIG_REQUIRE(grad_out && img && grad_img && dist_param && B >= 1 && H >= 1 && W >= 1 && C >= 1 && W_out >= 1 &&
                     (uint64_t)B * H * (W > W_out ? W : W_out) * C < (1ull << 31),
                 "wm_distort_geom_bwd: bad arguments");
    NSIG_REQUIRE((distortion == kDistRotation && W_out == W) || distortion == kDistScaling, "wm_distort_geom_bwd: distortion is 4 (rotation, W_out = W) or 5 (scaling)");
    k_distort_geom_bwd<<<ceil_div(B * H * W * C, 256u), 256, 0, as_stream(stream)>>>(grad_out, img, distortion, dist_param, B, H, W, W_out, C, grad_img);
    return check_launch("wm_distort_geom_bwd");
}
