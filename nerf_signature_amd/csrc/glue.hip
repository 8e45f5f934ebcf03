// The elementwise tails of the render and the losses of the watermark training step, each as one kernel per direction.
//
// On this path a "tensor" is a few thousand rays, so every stock elementwise/reduction operator is a ~3 us launch and the
// reference's op chains (renderer_wtmk.py:316-319: background mix and depth normalisation; utils_wtmk_disen.py:615-640:
// MSE + BCE-with-logits + weighted sum, and their autograd backward) come to ~60 launches per step: 0.2 ms of a 1.8 ms step.
#include "common.h"

namespace nsig {

__device__ inline float block_reduce_1024(float v, float *scratch) {   // scratch: 16 floats
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.0f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += scratch[w];
    return t;
}

// renderer_wtmk.py:316-319 / 369-372.
__global__ void __launch_bounds__(256) k_finish_fwd(const float *__restrict__ image, const float *__restrict__ depth, const float *__restrict__ ws,
                                                    const float *__restrict__ nears, const float *__restrict__ fars, const float *__restrict__ bg,
                                                    uint32_t bg_stride, uint32_t N, float *__restrict__ image_out, float *__restrict__ depth_out) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const float rest = 1.0f - ws[i];
#pragma unroll
    for (int c = 0; c < 3; ++c) image_out[i * 3 + c] = image[i * 3 + c] + rest * bg[i * bg_stride + c];
    depth_out[i] = fmaxf(depth[i] - nears[i], 0.0f) / (fars[i] - nears[i]);   // NaN for rays that miss the box, as the reference
}

__global__ void __launch_bounds__(256) k_finish_bwd(const float *__restrict__ g_image, const float *__restrict__ g_depth, const float *__restrict__ depth,
                                                    const float *__restrict__ nears, const float *__restrict__ fars, const float *__restrict__ bg,
                                                    uint32_t bg_stride, uint32_t N, float *__restrict__ g_ws, float *__restrict__ g_depth_in) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    float s = 0.0f;
    if (g_image)
#pragma unroll
        for (int c = 0; c < 3; ++c) s += g_image[i * 3 + c] * bg[i * bg_stride + c];
    g_ws[i] = -s;
    if (g_depth_in) g_depth_in[i] = (g_depth && depth[i] - nears[i] >= 0.0f) ? g_depth[i] / (fars[i] - nears[i]) : 0.0f;
}

// loss_i = mean((content - gt)^2); loss_w = mean BCE-with-logits(temp * decoded, message); loss = lambda_w*loss_w + lambda_i*loss_i.
// One workgroup; also leaves the two gradient directions d(loss_i)/d(content), d(loss_w)/d(decoded) for the backward kernel.
__global__ void __launch_bounds__(1024) k_wm_loss_fwd(const float *__restrict__ content, const float *__restrict__ gt, uint32_t n_content,
                                                      const float *__restrict__ decoded, const float *__restrict__ message, uint32_t D, float temp,
                                                      float lambda_w, float lambda_i, float *__restrict__ losses, float *__restrict__ d_content,
                                                      float *__restrict__ d_decoded) {
    __shared__ float scratch[16];
    float si = 0.0f, sw = 0.0f;
    const float ki = 2.0f / (float)n_content, kw = temp / (float)D;
    for (uint32_t i = threadIdx.x; i < n_content; i += 1024) {
        const float d = content[i] - gt[i];
        si += d * d;
        d_content[i] = ki * d;
    }
    for (uint32_t i = threadIdx.x; i < D; i += 1024) {
        const float x = temp * decoded[i], y = message[i];
        // binary_cross_entropy_with_logits: (1 - y) * x + log(1 + exp(-|x|)) + max(-x, 0)
        sw += (1.0f - y) * x + fmaxf(-x, 0.0f) + log1pf(expf(-fabsf(x)));
        d_decoded[i] = kw * (1.0f / (1.0f + expf(-x)) - y);
    }
    si = block_reduce_1024(si, scratch);
    sw = block_reduce_1024(sw, scratch);
    if (threadIdx.x == 0) {
        const float li = si / (float)n_content, lw = sw / (float)D;
        losses[0] = li;
        losses[1] = lw;
        losses[2] = lambda_w * lw + lambda_i * li;
    }
}

// grads: [g_lossi, g_lossw, g_loss] (device scalars; a null pointer is 0).
__global__ void __launch_bounds__(256) k_wm_loss_bwd(const float *__restrict__ g_li, const float *__restrict__ g_lw, const float *__restrict__ g_l,
                                                     float lambda_w, float lambda_i, const float *__restrict__ d_content, uint32_t n_content,
                                                     const float *__restrict__ d_decoded, uint32_t D, float *__restrict__ g_content,
                                                     float *__restrict__ g_decoded) {
    const float gl = g_l ? g_l[0] : 0.0f;
    const float ci = (g_li ? g_li[0] : 0.0f) + lambda_i * gl, cw = (g_lw ? g_lw[0] : 0.0f) + lambda_w * gl;
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n_content) g_content[i] = ci * d_content[i];
    if (i < D) g_decoded[i] = cw * d_decoded[i];
}

// Opening kernel of a captured training step: zero-fills the shared gradient buffer (torch's zero_grad) and stages this
// step's message words from a ring in pinned host memory into the device tensor the step's kernels read -- slot = the number
// of replays so far modulo `slots`, counted on the device, so a replayed graph needs no host-to-device copy in front of it.
__global__ void __launch_bounds__(256) k_step_begin(float4 *__restrict__ G4, uint32_t n4, const float *__restrict__ ring, uint32_t slots, uint32_t width,
                                                    uint32_t *__restrict__ counter, float *__restrict__ msg) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n4) G4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (blockIdx.x == 0) {
        const uint32_t slot = *counter % slots;      // every thread reads the count before thread 0 advances it
        if (threadIdx.x < width) msg[threadIdx.x] = ring[(size_t)slot * width + threadIdx.x];
        __syncthreads();
        if (threadIdx.x == 0) *counter = *counter + 1u;
    }
}

}  // namespace nsig

using namespace nsig;

NSIG_EXPORT void *nsig_host_device_pointer(const void *pinned_host) {
    void *dev = nullptr;
    if (pinned_host == nullptr || hipHostGetDevicePointer(&dev, const_cast<void *>(pinned_host), 0) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return dev;
}

NSIG_EXPORT int loop_step_begin(float *G, uint32_t n_floats, const float *ring_dev, uint32_t slots, uint32_t width, uint32_t *counter, float *msg,
                                nsig_stream_t stream) {
    NSIG_REQUIRE(G && ring_dev && counter && msg, "loop_step_begin: null pointer");
    NSIG_REQUIRE((reinterpret_cast<uintptr_t>(G) & 15) == 0 && n_floats % 4 == 0, "loop_step_begin: G must be 16-byte aligned with a multiple of 4 floats");
    NSIG_REQUIRE(slots >= 1 && width >= 1 && width <= 256, "loop_step_begin: slots >= 1 and 1 <= width <= 256");
    const uint32_t n4 = n_floats / 4;
    k_step_begin<<<ceil_div(n4 > 0 ? n4 : 1u, 256u), 256, 0, as_stream(stream)>>>(reinterpret_cast<float4 *>(G), n4, ring_dev, slots,
                                                                                 width, counter, msg);
    return check_launch("loop_step_begin");
}

NSIG_EXPORT int rm_finish_fwd(const float *image, const float *depth, const float *weights_sum, const float *nears, const float *fars,
                              const float *bg, uint32_t bg_stride, uint32_t N, float *image_out, float *depth_out, nsig_stream_t stream) {
    if (N == 0) return NSIG_OK;
    NSIG_REQUIRE(image && depth && weights_sum && nears && fars && bg && image_out && depth_out, "rm_finish_fwd: null pointer");
    NSIG_REQUIRE(bg_stride == 0 || bg_stride == 3, "rm_finish_fwd: bg_stride is 0 (one colour) or 3 (per ray)");
    k_finish_fwd<<<ceil_div(N, 256u), 256, 0, as_stream(stream)>>>(image, depth, weights_sum, nears, fars, bg, bg_stride, N, image_out, depth_out);
    return check_launch("rm_finish_fwd");
}

NSIG_EXPORT int rm_finish_bwd(const float *grad_image, const float *grad_depth, const float *depth, const float *nears, const float *fars,
                              const float *bg, uint32_t bg_stride, uint32_t N, float *grad_weights_sum, float *grad_depth_in, nsig_stream_t stream) {
    if (N == 0) return NSIG_OK;
    NSIG_REQUIRE(depth && nears && fars && bg && grad_weights_sum, "rm_finish_bwd: null pointer");
    NSIG_REQUIRE(bg_stride == 0 || bg_stride == 3, "rm_finish_bwd: bg_stride is 0 (one colour) or 3 (per ray)");
    k_finish_bwd<<<ceil_div(N, 256u), 256, 0, as_stream(stream)>>>(grad_image, grad_depth, depth, nears, fars, bg, bg_stride, N, grad_weights_sum, grad_depth_in);
    return check_launch("rm_finish_bwd");
}

NSIG_EXPORT int wm_loss_fwd(const float *content, const float *gt, uint32_t n_content, const float *decoded, const float *message, uint32_t D,
                            float temp, float lambda_w, float lambda_i, float *losses3, float *d_content, float *d_decoded, nsig_stream_t stream) {
    NSIG_REQUIRE(content && gt && decoded && message && losses3 && d_content && d_decoded, "wm_loss_fwd: null pointer");
    NSIG_REQUIRE(n_content >= 1 && D >= 1, "wm_loss_fwd: empty input");
    k_wm_loss_fwd<<<1, 1024, 0, as_stream(stream)>>>(content, gt, n_content, decoded, message, D, temp, lambda_w, lambda_i, losses3, d_content, d_decoded);
    return check_launch("wm_loss_fwd");
}

NSIG_EXPORT int wm_loss_bwd(const float *g_lossi, const float *g_lossw, const float *g_loss, float lambda_w, float lambda_i, const float *d_content,
                            uint32_t n_content, const float *d_decoded, uint32_t D, float *grad_content, float *grad_decoded, nsig_stream_t stream) {
    NSIG_REQUIRE(d_content && d_decoded && grad_content && grad_decoded, "wm_loss_bwd: null pointer");
    const uint32_t n = n_content > D ? n_content : D;
    k_wm_loss_bwd<<<ceil_div(n, 256u), 256, 0, as_stream(stream)>>>(g_lossi, g_lossw, g_loss, lambda_w, lambda_i, d_content, n_content, d_decoded, D,
                                                                    grad_content, grad_decoded);
    return check_launch("wm_loss_bwd");
}
