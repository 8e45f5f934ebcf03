// Fused BatchNorm (batch statistics) + GELU for the HiDDeN decoder's ConvBNRelu blocks
// (/root/reference/nerf/hidden_models.py:16-35: Conv2d -> BatchNorm2d(eps=1e-3, track_running_stats=False) -> GELU).
//
// The decoder works on D images of ~12x12 pixels: every tensor is about 1 MB and every stock operator is a
// launch-latency-bound kernel (BatchNorm alone is 2 kernels forward and 3 backward, GELU 1 + 1, plus reductions).
// Here one workgroup owns one channel: it reduces the channel's N*P values for the statistics, then applies
// normalisation + affine + GELU (forward), or recomputes them and produces dx, dgamma, dbeta (backward), in one launch.
// Tensors are channels-last (NHWC), the layout MIOpen's convolution kernels use.  A channel's values are 4-byte
// elements 4*C bytes apart, so every load is its own cache line and a read-modify loop would be a chain of exposed
// L2 latencies: the workgroup (1024 threads) instead issues all of its loads back to back into registers
// (<= 8 per thread, N*P <= 8192), reduces twice from registers, and stores; larger batches take the looping kernels.
#include "common.h"

namespace nsig {

__device__ inline float block_sum(float v, float *scratch) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    __syncthreads();  // scratch reuse
    if (lane == 0) scratch[wid] = v;
    __syncthreads();
    float t = 0.0f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += scratch[w];
    return t;
}

__device__ inline float gelu(float z) { return 0.5f * z * (1.0f + erff(z * 0.70710678118654752f)); }
__device__ inline float gelu_grad(float z) {
    return 0.5f * (1.0f + erff(z * 0.70710678118654752f)) + z * 0.39894228040143268f * expf(-0.5f * z * z);
}

__global__ void __launch_bounds__(256) k_bn_gelu_fwd(const float *__restrict__ x, const float *__restrict__ gamma, const float *__restrict__ beta,
                                                     uint32_t NP, uint32_t C, float eps, float *__restrict__ y, float *__restrict__ save) {
    __shared__ float scratch[4];
    const uint32_t c = blockIdx.x;
    float s = 0.0f;
    for (uint32_t i = threadIdx.x; i < NP; i += blockDim.x) s += x[(size_t)i * C + c];
    const float mean = block_sum(s, scratch) / (float)NP;
    float q = 0.0f;
    for (uint32_t i = threadIdx.x; i < NP; i += blockDim.x) {
        const float d = x[(size_t)i * C + c] - mean;
        q += d * d;
    }
    const float var = block_sum(q, scratch) / (float)NP;  // biased, as BatchNorm normalises with
    const float inv = 1.0f / sqrtf(var + eps);
    const float g = gamma[c], b = beta[c];
    for (uint32_t i = threadIdx.x; i < NP; i += blockDim.x) y[(size_t)i * C + c] = gelu((x[(size_t)i * C + c] - mean) * inv * g + b);
    if (threadIdx.x == 0) { save[c] = mean; save[C + c] = inv; }
}

__global__ void __launch_bounds__(256) k_bn_gelu_bwd(const float *__restrict__ dy, const float *__restrict__ x, const float *__restrict__ gamma,
                                                     const float *__restrict__ beta, const float *__restrict__ save, uint32_t NP, uint32_t C,
                                                     float *__restrict__ dx, float *__restrict__ dgamma, float *__restrict__ dbeta) {
    __shared__ float scratch[4];
    const uint32_t c = blockIdx.x;
    const float mean = save[c], inv = save[C + c], g = gamma[c], b = beta[c];
    // dz = dy * gelu'(z), z = gamma * xhat + beta;  dbeta = sum dz, dgamma = sum dz * xhat
    float s0 = 0.0f, s1 = 0.0f;
    for (uint32_t i = threadIdx.x; i < NP; i += blockDim.x) {
        const float xh = (x[(size_t)i * C + c] - mean) * inv;
        const float dz = dy[(size_t)i * C + c] * gelu_grad(xh * g + b);
        s0 += dz;
        s1 += dz * xh;
    }
    const float sum_dz = block_sum(s0, scratch), sum_dz_xh = block_sum(s1, scratch);
    const float k = g * inv / (float)NP;
    for (uint32_t i = threadIdx.x; i < NP; i += blockDim.x) {
        const float xh = (x[(size_t)i * C + c] - mean) * inv;
        const float dz = dy[(size_t)i * C + c] * gelu_grad(xh * g + b);
        dx[(size_t)i * C + c] = k * ((float)NP * dz - sum_dz - xh * sum_dz_xh);
    }
    if (threadIdx.x == 0) { dgamma[c] = sum_dz_xh; dbeta[c] = sum_dz; }
}

constexpr int kRegThreads = 1024, kRegVals = 8;   // register-resident variant: N*P <= 8192

__global__ void __launch_bounds__(kRegThreads) k_bn_gelu_fwd_reg(const float *__restrict__ x, const float *__restrict__ gamma,
                                                                 const float *__restrict__ beta, uint32_t NP, uint32_t C, float eps,
                                                                 float *__restrict__ y, float *__restrict__ save) {
    __shared__ float scratch[kRegThreads / 64];
    const uint32_t c = blockIdx.x;
    float v[kRegVals];
#pragma unroll
    for (int k = 0; k < kRegVals; ++k) {
        const uint32_t i = threadIdx.x + k * kRegThreads;
        v[k] = i < NP ? x[(size_t)i * C + c] : 0.0f;
    }
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < kRegVals; ++k) s += v[k];
    const float mean = block_sum(s, scratch) / (float)NP;
    float q = 0.0f;
#pragma unroll
    for (int k = 0; k < kRegVals; ++k) {
        const float d = v[k] - mean;
        q += (threadIdx.x + k * kRegThreads < NP) ? d * d : 0.0f;
    }
    const float var = block_sum(q, scratch) / (float)NP;
    const float inv = 1.0f / sqrtf(var + eps);
    const float g = gamma[c], b = beta[c];
#pragma unroll
    for (int k = 0; k < kRegVals; ++k) {
        const uint32_t i = threadIdx.x + k * kRegThreads;
        if (i < NP) y[(size_t)i * C + c] = gelu((v[k] - mean) * inv * g + b);
    }
    if (threadIdx.x == 0) { save[c] = mean; save[C + c] = inv; }
}

__global__ void __launch_bounds__(kRegThreads) k_bn_gelu_bwd_reg(const float *__restrict__ dy, const float *__restrict__ x,
                                                                 const float *__restrict__ gamma, const float *__restrict__ beta,
                                                                 const float *__restrict__ save, uint32_t NP, uint32_t C, float *__restrict__ dx,
                                                                 float *__restrict__ dgamma, float *__restrict__ dbeta) {
    __shared__ float scratch[kRegThreads / 64];
    const uint32_t c = blockIdx.x;
    float xh[kRegVals], dz[kRegVals];
#pragma unroll
    for (int k = 0; k < kRegVals; ++k) {
        const uint32_t i = threadIdx.x + k * kRegThreads;
        xh[k] = i < NP ? x[(size_t)i * C + c] : 0.0f;
        dz[k] = i < NP ? dy[(size_t)i * C + c] : 0.0f;
    }
    const float mean = save[c], inv = save[C + c], g = gamma[c], b = beta[c];
    float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
    for (int k = 0; k < kRegVals; ++k) {
        xh[k] = (xh[k] - mean) * inv;
        dz[k] = dz[k] * gelu_grad(xh[k] * g + b);   // 0 for the padding slots (dy = 0)
        s0 += dz[k];
        s1 += dz[k] * xh[k];
    }
    const float sum_dz = block_sum(s0, scratch), sum_dz_xh = block_sum(s1, scratch);
    const float kk = g * inv / (float)NP;
#pragma unroll
    for (int k = 0; k < kRegVals; ++k) {
        const uint32_t i = threadIdx.x + k * kRegThreads;
        if (i < NP) dx[(size_t)i * C + c] = kk * ((float)NP * dz[k] - sum_dz - xh[k] * sum_dz_xh);
    }
    if (threadIdx.x == 0) { dgamma[c] = sum_dz_xh; dbeta[c] = sum_dz; }
}

}  // namespace nsig

using namespace nsig;

NSIG_EXPORT int dec_bn_gelu_fwd(const float *x, const float *gamma, const float *beta, uint32_t N, uint32_t C, uint32_t P, float eps, float *y,
                                float *save, nsig_stream_t stream) {
    NSIG_REQUIRE(x && gamma && beta && y && save, "dec_bn_gelu_fwd: null pointer");
    NSIG_REQUIRE(N * P > 1 && C >= 1, "dec_bn_gelu_fwd: need more than one value per channel");
    if (N * P <= (uint32_t)(kRegThreads * kRegVals))
        k_bn_gelu_fwd_reg<<<C, kRegThreads, 0, as_stream(stream)>>>(x, gamma, beta, N * P, C, eps, y, save);
    else
        k_bn_gelu_fwd<<<C, 256, 0, as_stream(stream)>>>(x, gamma, beta, N * P, C, eps, y, save);
    return check_launch("dec_bn_gelu_fwd");
}

NSIG_EXPORT int dec_bn_gelu_bwd(const float *dy, const float *x, const float *gamma, const float *beta, const float *save, uint32_t N, uint32_t C,
                                uint32_t P, float *dx, float *dgamma, float *dbeta, nsig_stream_t stream) {
    NSIG_REQUIRE(dy && x && gamma && beta && save && dx && dgamma && dbeta, "dec_bn_gelu_bwd: null pointer");
    if (N * P <= (uint32_t)(kRegThreads * kRegVals))
        k_bn_gelu_bwd_reg<<<C, kRegThreads, 0, as_stream(stream)>>>(dy, x, gamma, beta, save, N * P, C, dx, dgamma, dbeta);
    else
        k_bn_gelu_bwd<<<C, 256, 0, as_stream(stream)>>>(dy, x, gamma, beta, save, N * P, C, dx, dgamma, dbeta);
    return check_launch("dec_bn_gelu_bwd");
}
