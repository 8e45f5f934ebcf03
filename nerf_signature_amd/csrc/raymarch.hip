// Ray-marching kernels for gfx950: aabb intersection, morton / bitfield utilities, the occupancy-grid
// marcher (training: count -> scan -> write; inference: burst march + device-side compaction) and the
// front-to-back compositor (forward / backward).
//
// Behavioural reference: /root/reference/raymarching/src/raymarching.cu (cited per kernel).  The
// structure is not the reference's: the training march is split so that the serial, divergent part
// (walking the occupancy grid) runs once and only records the parameter t of every sample; offsets
// come from a deterministic prefix sum instead of atomics; and the point buffers are then filled by
// one fully parallel, coalesced pass that also writes the zero padding (no N*max_steps memset).
//
// Built with -ffp-contract=off: every fused multiply-add below is explicit, so the integer outputs
// (sample counts) are reproducible bit-for-bit against the CPU oracle (see oracle/raymarch_ref.c).
#include "common.h"

#include <float.h>

namespace nsig {

constexpr float kSqrt3 = 1.7320508075688772f;
constexpr float kInvPi = 0.3183098861837907f;
constexpr uint32_t kScanWriteMaxRays = 12288;   // k_march_scan_write keeps N + 1 offsets in LDS (48 KiB: three workgroups per CU at the limit)

// ----------------------------------------------------------------------------- small utilities

// raymarching.cu:92-145.  One definition for the stand-alone kernel and for the march that computes its own limits: same instructions,
// same results bit for bit.
__device__ inline void near_far_of(const float *__restrict__ o3, const float *__restrict__ d3, const float *__restrict__ aabb, float min_near,
                                   float &near, float &far) {
    float tmin = 0.f, tmax = 0.f;
    bool miss = false;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float o = o3[a];
        const float inv = 1.0f / d3[a];
        float lo = (aabb[a] - o) * inv;
        float hi = (aabb[a + 3] - o) * inv;
        if (lo > hi) { const float s = lo; lo = hi; hi = s; }
        if (a == 0) { tmin = lo; tmax = hi; }
        else if (!miss) {
            if (tmin > hi || lo > tmax) miss = true;
            else { if (lo > tmin) tmin = lo; if (hi < tmax) tmax = hi; }
        }
    }
    if (miss) { near = FLT_MAX; far = FLT_MAX; return; }
    near = tmin < min_near ? min_near : tmin;
    far = tmax;
}

__global__ void k_near_far(const float *__restrict__ rays_o, const float *__restrict__ rays_d,
                           const float *__restrict__ aabb, uint32_t N, float min_near, float *__restrict__ nears,
                           float *__restrict__ fars) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float near, far;
    near_far_of(rays_o + 3 * (size_t)n, rays_d + 3 * (size_t)n, aabb, min_near, near, far);
    nears[n] = near;
    fars[n] = far;
}

__global__ void k_sph_from_ray(const float *__restrict__ rays_o, const float *__restrict__ rays_d, float radius,
                               uint32_t N, float *__restrict__ coords) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float ox = rays_o[3 * n], oy = rays_o[3 * n + 1], oz = rays_o[3 * n + 2];
    const float dx = rays_d[3 * n], dy = rays_d[3 * n + 1], dz = rays_d[3 * n + 2];
    const float A = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
    const float B = fmaf(oz, dz, fmaf(oy, dy, ox * dx));
    const float C = fmaf(oz, oz, fmaf(oy, oy, ox * ox)) - radius * radius;
    const float t = (-B + sqrtf(B * B - A * C)) / A;
    const float x = fmaf(t, dx, ox), y = fmaf(t, dy, oy), z = fmaf(t, dz, oz);
    coords[2 * n] = 2.0f * atan2f(sqrtf(fmaf(z, z, x * x)), y) * kInvPi - 1.0f;
    coords[2 * n + 1] = atan2f(z, x) * kInvPi;
}

__global__ void k_morton(const int32_t *__restrict__ coords, uint32_t N, int32_t *__restrict__ indices) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    indices[n] = (int32_t)morton3((uint32_t)coords[3 * n], (uint32_t)coords[3 * n + 1], (uint32_t)coords[3 * n + 2]);
}

__global__ void k_morton_invert(const int32_t *__restrict__ indices, uint32_t N, int32_t *__restrict__ coords) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const int32_t v = indices[n];
    coords[3 * n] = (int32_t)compact3((uint32_t)v);
    coords[3 * n + 1] = (int32_t)compact3((uint32_t)(v >> 1));
    coords[3 * n + 2] = (int32_t)compact3((uint32_t)(v >> 2));
}

// One thread packs 4 bytes: two float4 loads per byte, 32 B/lane of reads in flight, dword store.
__global__ void k_packbits(const float *__restrict__ grid, uint32_t n_bytes, float thresh, uint8_t *__restrict__ bits) {
    const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;  // word index
    const uint32_t first = w * 4;
    if (first >= n_bytes) return;
    uint32_t word = 0;
    const uint32_t nb = min(4u, n_bytes - first);
    for (uint32_t b = 0; b < nb; ++b) {
        const float4 lo = reinterpret_cast<const float4 *>(grid)[2 * (size_t)(first + b)];
        const float4 hi = reinterpret_cast<const float4 *>(grid)[2 * (size_t)(first + b) + 1];
        uint32_t v = (lo.x > thresh) | ((lo.y > thresh) << 1) | ((lo.z > thresh) << 2) | ((lo.w > thresh) << 3) |
                     ((hi.x > thresh) << 4) | ((hi.y > thresh) << 5) | ((hi.z > thresh) << 6) | ((hi.w > thresh) << 7);
        word |= v << (8 * b);
    }
    if (nb == 4 && (reinterpret_cast<uintptr_t>(bits) & 3) == 0) reinterpret_cast<uint32_t *>(bits)[w] = word;
    else for (uint32_t b = 0; b < nb; ++b) bits[first + b] = (uint8_t)(word >> (8 * b));
}

// get_rays (nerf/utils_wtmk_disen.py:121-141) for given pixel indices: one lane per ray, no [B, H*W] meshgrid.
__global__ void k_get_rays(const float *__restrict__ poses, float fx, float fy, float cx, float cy, uint32_t H, uint32_t W,
                           const int64_t *__restrict__ inds, uint32_t B, uint32_t N, float *__restrict__ rays_o,
                           float *__restrict__ rays_d) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= B * N) return;
    const uint32_t b = gid / N, n = gid % N;
    const int64_t ind = inds ? inds[(size_t)b * N + n] : (int64_t)n;
    const float i = (float)(ind % (int64_t)W) + 0.5f, j = (float)(ind / (int64_t)W) + 0.5f;  // pixel centres (:76-77)
    const float x = (i - cx) / fx, y = (j - cy) / fy, z = 1.0f;                               // (:131-133), zs == 1
    const float nrm = sqrtf(x * x + y * y + z * z);                                          // torch.norm(dim=-1)
    const float dx = x / nrm, dy = y / nrm, dz = z / nrm;
    const float *P = poses + 16 * (size_t)b;                                                  // row-major [4,4]
    float *o = rays_o + 3 * (size_t)gid, *d = rays_d + 3 * (size_t)gid;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        d[r] = dx * P[4 * r] + dy * P[4 * r + 1] + dz * P[4 * r + 2];                          // directions @ R^T (:135)
        o[r] = P[4 * r + 3];                                                                 // camera centre (:137)
    }
}

// One training batch straight from a device-resident store of poses and images (the loader step in front of the
// path, nerf/provider_wtmk.py:585-600 + get_rays with N > 0, nerf/utils_wtmk_disen.py:103-106,121-141): pose p = (step * stride +
// offset) mod P with `step` read from a device counter, N pixel indices drawn uniformly from a counter-based hash of (seed, step, ray)
// (the reference draws torch.randint on its device: same distribution, another generator), their rays, and the pixels' colours of
// image p as ground truth.  One launch, no host value in it: it sits inside a captured training step.
__device__ inline uint32_t mix32(uint32_t v) {   // a finaliser with full avalanche (murmur3)
    v ^= v >> 16; v *= 0x85ebca6bu; v ^= v >> 13; v *= 0xc2b2ae35u; v ^= v >> 16;
    return v;
}

__global__ void k_sample_rays(const float *__restrict__ poses, uint32_t P, const float *__restrict__ images, float fx, float fy, float cx, float cy,
                              uint32_t H, uint32_t W, uint32_t N, const int32_t *__restrict__ step_counter, uint32_t stride, uint32_t offset,
                              uint32_t seed_lo, uint32_t seed_hi, float *__restrict__ rays_o, float *__restrict__ rays_d, float *__restrict__ gt,
                              int64_t *__restrict__ inds_out, int32_t *__restrict__ pose_out) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const uint32_t step = step_counter ? (uint32_t)step_counter[0] : 0u;
    const uint32_t p = (uint32_t)(((uint64_t)step * stride + offset) % P);
    const uint32_t r = mix32(mix32(mix32(seed_lo ^ (step * 0x9e3779b9u)) + seed_hi) ^ (n * 0x85ebca6bu + 0x6b43a9b5u));
    const uint32_t ind = (uint32_t)(((uint64_t)r * ((uint64_t)H * W)) >> 32);                  // uniform in [0, H*W)
    const float i = (float)(ind % W) + 0.5f, j = (float)(ind / W) + 0.5f;                      // as k_get_rays
    const float x = (i - cx) / fx, y = (j - cy) / fy, z = 1.0f;
    const float nrm = sqrtf(x * x + y * y + z * z);
    const float dx = x / nrm, dy = y / nrm, dz = z / nrm;
    const float *Pm = poses + 16 * (size_t)p;
    float *o = rays_o + 3 * (size_t)n, *d = rays_d + 3 * (size_t)n;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        d[c] = dx * Pm[4 * c] + dy * Pm[4 * c + 1] + dz * Pm[4 * c + 2];
        o[c] = Pm[4 * c + 3];
    }
    if (gt != nullptr) {
        const float *px = images + 3 * ((size_t)p * H * W + ind);
        gt[3 * (size_t)n] = px[0]; gt[3 * (size_t)n + 1] = px[1]; gt[3 * (size_t)n + 2] = px[2];
    }
    if (inds_out != nullptr) inds_out[n] = (int64_t)ind;
    if (pose_out != nullptr && n == 0) pose_out[0] = (int32_t)p;
}

// ----------------------------------------------------------------------------- the occupancy walk

struct GridView {
    const uint8_t *bits;
    float bound, dt_gamma, dt_min, dt_max, dt_const, inv_H, H3f, Hf, Cf, top;
    double Hd;
};

__host__ inline GridView make_grid_view(const uint8_t *bits, float bound, float dt_gamma, uint32_t max_steps,
                                        uint32_t C, uint32_t H) {
    GridView g;
    g.bits = bits;
    g.bound = bound;
    g.dt_gamma = dt_gamma;
    g.dt_min = (2.0f * kSqrt3) / (float)max_steps;                      // raymarching.cu:345
    g.dt_max = ((2.0f * kSqrt3) * (float)(1u << (C - 1))) / (float)H;   // raymarching.cu:346
    g.dt_const = fminf(g.dt_max, fmaxf(g.dt_min, 0.0f));                 // the step when dt_gamma == 0: clamp(0, dt_min, dt_max), which is dt_max when max_steps is so small that dt_min > dt_max
    g.inv_H = 1.0f / (float)H;
    g.H3f = (float)(H * H * H);
    g.Hf = (float)H;
    g.Cf = (float)C;
    g.top = (float)(H - 1);
    g.Hd = (double)H;
    return g;
}

struct Ray {
    float ox, oy, oz, dx, dy, dz, rdx, rdy, rdz;
    __device__ Ray(const float *o, const float *d)
        : ox(o[0]), oy(o[1]), oz(o[2]), dx(d[0]), dy(d[1]), dz(d[2]), rdx(1.0f / d[0]), rdy(1.0f / d[1]), rdz(1.0f / d[2]) {}
};

__device__ inline float step_len(const GridView &g, float t) { return clampf(t * g.dt_gamma, g.dt_min, g.dt_max); }

__device__ inline int cascade_of(const GridView &g, float x, float y, float z, float dt) {
    // raymarching.cu:42-54,368: the coarser of "which shell is the point in" and "which shell matches dt".
    const float m = fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z)));
    int e_pos, e_dt;
    frexpf(m, &e_pos);
    frexpf((float)((double)(dt * g.Hf) * 0.5), &e_dt);
    const int lp = (int)fminf(g.Cf - 1.0f, fmaxf(0.0f, (float)e_pos));
    const int ls = (int)fminf(g.Cf - 1.0f, fmaxf(0.0f, (float)e_dt));
    return max(lp, ls);
}

__device__ inline int cell_coord(const GridView &g, float v, float inv_extent) {
    // raymarching.cu:374-376: float bracket, double product, float clamp, truncation.
    return (int)clampf((float)(0.5 * (double)fmaf(v, inv_extent, 1.0f) * g.Hd), 0.0f, g.top);
}

// Tests the cell containing o + t d.  Occupied: returns true (x,y,z,dt valid).  Empty: returns false and
// t_exit = parameter at which the ray leaves the cell (raymarching.cu:390-394).
__device__ inline bool probe(const GridView &g, const Ray &r, float t, float &x, float &y, float &z, float &dt,
                             float &t_exit) {
    x = clampf(fmaf(t, r.dx, r.ox), -g.bound, g.bound);
    y = clampf(fmaf(t, r.dy, r.oy), -g.bound, g.bound);
    z = clampf(fmaf(t, r.dz, r.oz), -g.bound, g.bound);
    dt = step_len(g, t);
    const int level = cascade_of(g, x, y, z, dt);
    const float extent = fminf(ldexpf(1.0f, level), g.bound);
    const float inv_extent = 1.0f / extent;
    const int nx = cell_coord(g, x, inv_extent), ny = cell_coord(g, y, inv_extent), nz = cell_coord(g, z, inv_extent);
    const uint32_t bit = (uint32_t)((float)level * g.H3f + (float)morton3((uint32_t)nx, (uint32_t)ny, (uint32_t)nz));
    if (g.bits[bit >> 3] & (1u << (bit & 7u))) return true;
    const float fx = (fmaf(0.5f, copysignf(1.0f, r.dx), (float)nx + 0.5f) * g.inv_H) * 2.0f - 1.0f;
    const float fy = (fmaf(0.5f, copysignf(1.0f, r.dy), (float)ny + 0.5f) * g.inv_H) * 2.0f - 1.0f;
    const float fz = (fmaf(0.5f, copysignf(1.0f, r.dz), (float)nz + 0.5f) * g.inv_H) * 2.0f - 1.0f;
    const float tx = fmaf(fx, extent, -x) * r.rdx;
    const float ty = fmaf(fy, extent, -y) * r.rdy;
    const float tz = fmaf(fz, extent, -z) * r.rdz;
    t_exit = t + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
    return false;
}

__device__ inline float leave_cell(const GridView &g, float t, float t_exit) {  // raymarching.cu:396-398
    do { t += step_len(g, t); } while (t < t_exit);
    return t;
}

__device__ inline float start_param(const GridView &g, float near, float noise) {  // raymarching.cu:348-351
    return fmaf(step_len(g, near), noise, near);
}

// Occupancy test specialised for the walk kernels.  kOneCascade: C == 1, so the cascade level is always 0 and the
// cell extent is loop-invariant.  H is a power of two (checked on the host), so the reference's double-precision
// product 0.5 * f * H (raymarching.cu:374-376) is an exact scaling and is evaluated in fp32.
template <bool kOneCascade>
struct Walker {
    const GridView &g;
    const Ray &r;
    float extent, inv_extent, half_H;
    int level;
    __device__ Walker(const GridView &g_, const Ray &r_) : g(g_), r(r_), half_H(0.5f * g_.Hf), level(-1) {
        if (kOneCascade) { level = 0; extent = fminf(1.0f, g.bound); inv_extent = 1.0f / extent; }
    }
    __device__ inline int cell(float v) const { return (int)clampf(fmaf(v, inv_extent, 1.0f) * half_H, 0.0f, g.top); }
    // returns occupancy at parameter t; when empty, t_exit is where the ray leaves the cell
    __device__ inline bool probe(float t, float dt, float &t_exit) {
        const float x = clampf(fmaf(t, r.dx, r.ox), -g.bound, g.bound);
        const float y = clampf(fmaf(t, r.dy, r.oy), -g.bound, g.bound);
        const float z = clampf(fmaf(t, r.dz, r.oz), -g.bound, g.bound);
        if (!kOneCascade) {
            const int lv = cascade_of(g, x, y, z, dt);
            if (lv != level) { level = lv; extent = fminf(ldexpf(1.0f, lv), g.bound); inv_extent = 1.0f / extent; }
        }
        const int nx = cell(x), ny = cell(y), nz = cell(z);
        const uint32_t bit = (uint32_t)((float)level * g.H3f + (float)morton3((uint32_t)nx, (uint32_t)ny, (uint32_t)nz));
        if (g.bits[bit >> 3] & (1u << (bit & 7u))) return true;
        const float fx = (fmaf(0.5f, copysignf(1.0f, r.dx), (float)nx + 0.5f) * g.inv_H) * 2.0f - 1.0f;
        const float fy = (fmaf(0.5f, copysignf(1.0f, r.dy), (float)ny + 0.5f) * g.inv_H) * 2.0f - 1.0f;
        const float fz = (fmaf(0.5f, copysignf(1.0f, r.dz), (float)nz + 0.5f) * g.inv_H) * 2.0f - 1.0f;
        const float tx = fmaf(fx, extent, -x) * r.rdx;
        const float ty = fmaf(fy, extent, -y) * r.rdy;
        const float tz = fmaf(fz, extent, -z) * r.rdz;
        t_exit = t + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
        return false;
    }
};

// Training march, pass 1 (raymarching.cu:354-400), in index space.
//
// Both branches of the reference's loop advance t by the same rule -- an occupied step does t += dt with
// dt = step(t) (:385-386), an empty cell is left by repeating t += step(t) (:396-398) -- so every parameter the
// walk can ever visit belongs to ONE sequence t_0 = start, t_{k+1} = t_k + step(t_k) that does not depend on the
// occupancy grid.  The walk is therefore a pointer chase over indices k of that sequence:
//     occupied(t_k) : k is a sample, go to k + 1
//     empty(t_k)    : go to L(k) = the first m > k with t_m >= t_exit(t_k)
// where occupied(.) and t_exit(.) are functions of t_k alone.  One wave owns one ray: its 64 lanes hold 64
// consecutive t_k (each accumulated with exactly the sequential rounding), probe the grid for all of them at once
// (the bitfield loads of a whole chunk are in flight together instead of one dependent load per step), find L(k)
// by binary search over the chunk's t values in LDS, and the chase itself runs on wave-uniform scalars over the
// 64-bit occupancy ballot (ctz over runs of occupied steps, v_readlane for the jumps).  A jump that leaves the
// chunk carries t_exit to the next one.  Counts and sampled parameters are bit-identical to the sequential walk.
// The chunk t_j = t_0 advanced j times by the constant step dt, j = 0..63, with the exact rounding of the sequential chain
// t <- fl(t + dt), in closed form.  Inside a binade [2^e, 2^(e+1)) every t is a multiple of u = ulp; fl(t + dt) adds the
// constant increment inc = (floor(dt/u) + [frac(dt/u) > 1/2]) * u (a tie, frac == 1/2, would alternate: not handled), so
// t_j = t_0 + j*inc, a product and a sum that are both exact in fp32.  At most one binade boundary falls inside a chunk
// (t_0 >= 1/4 and 63*dt below a binade's width are required): the first value past it is one real fp32 addition from
// its predecessor, and the values after it follow the same rule with u doubled.  Returns false (wave-uniformly) when a
// precondition fails; the caller then runs the sequential chain.  t0, dt: wave-uniform.
__device__ inline bool chunk_closed_form(float t0, float dt, int lane, float &tj) {
    const uint32_t b0 = __float_as_uint(t0);
    const uint32_t e = b0 & 0x7f800000u;
    if (!(t0 >= 0.25f) || !(t0 < 1048576.0f) || !(dt > 0.0f) || dt * 64.0f > t0) return false;   // the last test: 63 steps < one binade width
    const float u = __uint_as_float(e - (23u << 23)), inv_u = __uint_as_float((254u << 23) - (e - (23u << 23)));   // 2^(e-23), 2^-(e-23)
    const float top = __uint_as_float(e + (1u << 23));
    const float q = dt * inv_u, n = floorf(q), r = q - n;       // exact: scaling by a power of two
    if (r == 0.5f || n >= 131072.0f) return false;               // tie, or 64*(n+1) would leave the exact-integer range of fp32
    const float N = n + (r > 0.5f ? 1.0f : 0.0f), inc = N * u;
    // first index whose exact value reaches the next binade: jc = ceil((top - t0) / inc), in exact integer arithmetic (units of u)
    const float D = (top - t0) * inv_u;                          // exact integer < 2^23
    float jc = floorf(D / N);
    while (jc * N < D) jc += 1.0f;                               // correct the division's rounding (at most one step each way)
    while (jc >= 1.0f && (jc - 1.0f) * N >= D) jc -= 1.0f;
    const float j = (float)lane;
    if (j < jc) {
        tj = t0 + j * inc;
        return true;
    }
    // past the boundary: ulp 2u
    const float tc = (t0 + (jc - 1.0f) * inc) + dt;             // one rounded addition, exactly as the chain does it (jc >= 1: t0 < top)
    const float q2 = q * 0.5f, n2 = floorf(q2), r2 = q2 - n2;
    if (r2 == 0.5f) return false;                                // uniform: depends on dt and the binade only
    const float inc2 = (n2 + (r2 > 0.5f ? 1.0f : 0.0f)) * (2.0f * u);
    tj = tc + (j - jc) * inc2;
    return true;
}

// aabb != nullptr: the ray's limits are computed here (near_far_of) and stored to nears / fars for the kernels behind the march --
// the stand-alone near/far launch in front of every training march disappears (rm_march_train_count_nf).
template <bool kOneCascade, bool kConstDt>
__global__ void __launch_bounds__(256) k_march_index(const float *__restrict__ rays_o, const float *__restrict__ rays_d, GridView g,
                                                      uint32_t max_steps, uint32_t N, float *__restrict__ nears,
                                                      float *__restrict__ fars, const float *__restrict__ noises,
                                                      int32_t *__restrict__ counts, float *__restrict__ t_rec,
                                                      const float *__restrict__ aabb, float min_near) {
    __shared__ float ts[4][64];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const uint32_t n = blockIdx.x * 4 + wid;  // one wave per ray
    if (n >= N) return;
    const Ray r(rays_o + 3 * (size_t)n, rays_d + 3 * (size_t)n);
    Walker<kOneCascade> w(g, r);
    float near, far;
    if (aabb != nullptr) {      // (wave-uniform: every lane computes the same two numbers)
        near_far_of(rays_o + 3 * (size_t)n, rays_d + 3 * (size_t)n, aabb, min_near, near, far);
        if (lane == 0) { nears[n] = near; fars[n] = far; }
    } else {
        near = nears[n];
        far = fars[n];
    }
    auto step = [&](float t) { return kConstDt ? g.dt_const : step_len(g, t); };  // dt_gamma == 0: clamp(0, dt_min, dt_max), a constant
    float t_base = start_param(g, near, noises ? noises[n] : 0.0f);
    float *rec = t_rec + (size_t)n * max_steps;
    float *tl = ts[wid];
    uint32_t cnt = 0;
    float carry_exit = -FLT_MAX;  // the next visited index is the first one with t >= carry_exit
    while (t_base < far && cnt < max_steps) {
        // lane j: t_j = t_base advanced j times (its own sequential chain, identical rounding to the one-lane walk)
        float tj = t_base;
        bool closed = false;
        if (kConstDt) {
            // every return path of chunk_closed_form before the lane-dependent part is wave-uniform; the tie test after the
            // boundary is uniform too, so `closed` is the same in all lanes
            closed = chunk_closed_form(t_base, g.dt_const, lane, tj);
            closed = __all(closed);
            if (!closed) tj = t_base;
        }
        if (!closed)
            for (int i = 0; i < 63; ++i)
                if (i < lane) tj += step(tj);
        const float dt = step(tj);
        tl[lane] = tj;
        __builtin_amdgcn_wave_barrier();  // the binary search below reads other lanes' entries (same wave: LDS is in order)
        const bool valid = tj < far;
        float t_exit = 0.0f;
        const bool occ = valid && w.probe(tj, dt, t_exit);
        const unsigned long long occ_mask = __ballot(occ), valid_mask = __ballot(valid);
        // L(lane): first index m > lane of this chunk with t_m >= t_exit (64 = beyond the chunk); t is increasing
        int L = 64;
        if (valid && !occ) {
            // first m in (lane, 64) with t_m >= t_exit, else 64.  With a constant step that is lane + ceil((t_exit - t_lane) / dt) up to
            // rounding: start there and step to the exact index with the chunk's actual values (two LDS reads in flight instead of the six
            // dependent rounds of a binary search)
            const float q = kConstDt ? (t_exit - tj) / dt : -1.0f;
            if (kConstDt && q >= 0.0f && q < 1.0e6f) {
                int m = lane + max(1, (int)ceilf(q));
                m = m > 64 ? 64 : m;
                for (;;) {
                    const float a = m > lane + 1 ? tl[m - 1] : -FLT_MAX, b = m < 64 ? tl[m] : FLT_MAX;
                    if (a >= t_exit) { --m; continue; }
                    if (b < t_exit) { ++m; continue; }
                    break;
                }
                L = m;
            } else {
                int lo = lane + 1, hi = 64;
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (tl[mid] >= t_exit) hi = mid; else lo = mid + 1;
                }
                L = lo;
            }
        }
        // ---- chains of empty cells collapsed in parallel.  The chase below used to visit every empty cell on its own (one trip of ~30 dependent
        // scalar instructions per cell, ~14 cells per chunk in empty space: 22 of the 34 us a ray's walk takes, profiles/r02_march_pipelined_experiment.txt).
        // An empty index k goes to L(k), which is again an index of this chunk or 64; following L from k until the first index that is not an
        // empty one (an occupied index, one with t >= far, or 64) is a composition of the SAME map, so six rounds of pointer doubling over
        // the 64 lanes give every empty index its chain's end (`hop & 0xff`) and the last empty index of the chain (`hop >> 8`, whose exit
        // parameter is what the serial walk would carry out of the chunk).  The chase then takes ONE trip per chain of empty cells.
        const bool empty = valid && !occ;
        const unsigned long long link_mask = __ballot(empty);        // indices whose successor is another jump
        int hop = empty ? (L | (lane << 8)) : (lane | (lane << 8));
#pragma unroll
        for (int round = 0; round < 6; ++round) {
            const int j = hop & 0xff;
            const int other = __shfl(hop, j < 64 ? j : lane, 64);
            if (empty && j < 64 && ((link_mask >> j) & 1ull)) hop = other;
        }
        // ---- wave-uniform chase over this chunk
        const unsigned long long reach = __ballot(tj >= carry_exit);
        int k = reach ? __builtin_ctzll(reach) : 64;
        unsigned long long samples = 0ull;
        bool done = false;
        while (k < 64) {
            if (!((valid_mask >> k) & 1ull)) { done = true; break; }  // t_k >= far: the loop condition fails here
            if ((occ_mask >> k) & 1ull) {
                const unsigned long long rest = ~(occ_mask >> k);     // zero bits = consecutive occupied steps from k
                const int run = rest ? __builtin_ctzll(rest) : 64;
                const int len = run < 64 - k ? run : 64 - k;
                samples |= (len == 64 ? ~0ull : ((1ull << len) - 1ull)) << k;
                k += len;
                carry_exit = -FLT_MAX;                                  // if the run reaches the chunk end, index 0 of the next chunk is visited
            } else {
                const int from = __builtin_amdgcn_readfirstlane(k);
                const int chain = __builtin_amdgcn_readlane(hop, from);
                k = chain & 0xff;
                carry_exit = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, t_exit), chain >> 8));
            }
        }
        // ---- sample cap (raymarching.cu:359: num_steps < max_steps) and coalesced store of the sampled parameters
        uint32_t take = (uint32_t)__popcll(samples);
        if (cnt + take >= max_steps) {
            take = max_steps - cnt;
            done = true;
        }
        const uint32_t before = (uint32_t)__popcll(samples & ((1ull << lane) - 1ull));
        if (((samples >> lane) & 1ull) && before < take) rec[cnt + before] = tj;
        cnt += take;
        if (done) break;
        t_base = __shfl(tj + dt, 63, 64);
    }
    if (lane == 0) counts[n] = (int32_t)cnt;
}

// Training march, pass 2: exclusive prefix sum of the counts in ray-id order (replaces the two
// atomicAdd reservations of raymarching.cu:405-406).  One 1024-thread workgroup; each thread owns a
// contiguous chunk, wave-level scan by DPP shuffles, wave totals through LDS.
__global__ void __launch_bounds__(1024) k_march_scan(const int32_t *__restrict__ counts, uint32_t N,
                                                     int32_t *__restrict__ rays, int32_t *__restrict__ counter) {
    __shared__ int32_t wave_tot[16];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const uint32_t chunk = ceil_div(N, 1024u);
    const uint32_t beg = min(N, tid * chunk), end = min(N, beg + chunk);
    int32_t sum = 0;
    for (uint32_t i = beg; i < end; ++i) sum += counts[i];
    int32_t incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int32_t v = __shfl_up(incl, d, 64);
        if ((int)lane >= d) incl += v;
    }
    if (lane == 63) wave_tot[wid] = incl;
    __syncthreads();
    int32_t base = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        const int32_t v = wave_tot[w];
        if (w < (int)wid) base += v;
        total += v;
    }
    int32_t off = base + incl - sum;
    for (uint32_t i = beg; i < end; ++i) {
        const int32_t c = counts[i];
        rays[3 * (size_t)i] = (int32_t)i;
        rays[3 * (size_t)i + 1] = off;
        rays[3 * (size_t)i + 2] = c;
        off += c;
    }
    if (tid == 0) { counter[0] = total; counter[1] = (int32_t)N; }
}

// The same prefix sum for MANY rays (a staged full-image render marches up to 262 144 rays per launch sequence: the single workgroup above
// takes 316 us there, 8 % of a 1008 x 756 image, profiles/r03_fern_kernel_stats.txt): two launches of workgroups that own 4096 rays each --
// (1) every workgroup's total, (2) every workgroup adds up the totals in front of it (at most a few hundred values) and scans its own rays.
// Same ray-id order, same table.
constexpr uint32_t kScanChunk = 4096;      // rays per workgroup: 4 per thread
__global__ void __launch_bounds__(1024) k_march_scan_sums(const int32_t *__restrict__ counts, uint32_t N, int32_t *__restrict__ sums) {
    __shared__ int32_t wave_tot[16];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, base = blockIdx.x * kScanChunk;
    int32_t s = 0;
#pragma unroll
    for (uint32_t u = 0; u < 4; ++u) {
        const uint32_t i = base + tid + 1024u * u;
        s += i < N ? counts[i] : 0;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
    if (lane == 0) wave_tot[wid] = s;
    __syncthreads();
    if (tid == 0) {
        int32_t t = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) t += wave_tot[w];
        sums[blockIdx.x] = t;
    }
}
__global__ void __launch_bounds__(1024) k_march_scan_apply(const int32_t *__restrict__ counts, uint32_t N, const int32_t *__restrict__ sums,
                                                           int32_t *__restrict__ rays, int32_t *__restrict__ counter) {
    __shared__ int32_t wave_tot[16], part[16];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, base = blockIdx.x * kScanChunk;
    // the totals in front of this workgroup (and, for the last one, the grand total)
    int32_t before = 0, all = 0;
    for (uint32_t b = tid; b < gridDim.x; b += 1024u) {
        const int32_t v = sums[b];
        all += v;
        if (b < blockIdx.x) before += v;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { before += __shfl_xor(before, d, 64); all += __shfl_xor(all, d, 64); }
    if (lane == 0) { wave_tot[wid] = before; part[wid] = all; }
    __syncthreads();
    before = 0; all = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) { before += wave_tot[w]; all += part[w]; }
    __syncthreads();
    // own rays: thread t owns the 4 consecutive rays base + 4 t ..
    const uint32_t first = base + 4u * tid;
    int32_t c[4], sum = 0;
#pragma unroll
    for (uint32_t u = 0; u < 4; ++u) {
        c[u] = first + u < N ? counts[first + u] : 0;
        sum += c[u];
    }
    int32_t incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int32_t v = __shfl_up(incl, d, 64);
        if ((int)lane >= d) incl += v;
    }
    if (lane == 63) wave_tot[wid] = incl;
    __syncthreads();
    int32_t off = before + incl - sum;
#pragma unroll
    for (int w = 0; w < 16; ++w)
        if (w < (int)wid) off += wave_tot[w];
#pragma unroll
    for (uint32_t u = 0; u < 4; ++u) {
        const uint32_t i = first + u;
        if (i < N) {
            rays[3 * (size_t)i] = (int32_t)i;
            rays[3 * (size_t)i + 1] = off;
            rays[3 * (size_t)i + 2] = c[u];
        }
        off += c[u];
    }
    if (blockIdx.x == 0 && tid == 0) { counter[0] = all; counter[1] = (int32_t)N; }
}

// Training march, pass 3 (raymarching.cu:422-479 without the second walk): one lane per output row.
__global__ void k_march_write(const float *__restrict__ rays_o, const float *__restrict__ rays_d, GridView g,
                              uint32_t max_steps, uint32_t N, uint32_t M, const float *__restrict__ nears,
                              const float *__restrict__ noises, const float *__restrict__ t_rec,
                              const int32_t *__restrict__ rays, const int32_t *__restrict__ counter,
                              float *__restrict__ xyzs, float *__restrict__ dirs, float *__restrict__ deltas) {
    const uint32_t total = (uint32_t)counter[0];
    for (uint32_t m = blockIdx.x * blockDim.x + threadIdx.x; m < M; m += gridDim.x * blockDim.x) {
        float px = 0, py = 0, pz = 0, qx = 0, qy = 0, qz = 0, d0 = 0, d1 = 0;
        if (m < total) {
            // last ray whose offset <= m (rays with no samples share their successor's offset)
            uint32_t lo = 0, hi = N;
            while (hi - lo > 1) {
                const uint32_t mid = (lo + hi) >> 1;
                if ((uint32_t)rays[3 * (size_t)mid + 1] <= m) lo = mid; else hi = mid;
            }
            const uint32_t off = (uint32_t)rays[3 * (size_t)lo + 1], cnt = (uint32_t)rays[3 * (size_t)lo + 2];
            if (off + cnt <= M) {  // raymarching.cu:416: a ray that does not fit writes nothing
                const uint32_t s = m - off;
                const float *rec = t_rec + (size_t)lo * max_steps;
                const float t = rec[s];
                const float3 rd = *reinterpret_cast<const float3 *>(rays_d + 3 * (size_t)lo), ro = *reinterpret_cast<const float3 *>(rays_o + 3 * (size_t)lo);
                qx = rd.x; qy = rd.y; qz = rd.z;
                px = clampf(fmaf(t, qx, ro.x), -g.bound, g.bound);
                py = clampf(fmaf(t, qy, ro.y), -g.bound, g.bound);
                pz = clampf(fmaf(t, qz, ro.z), -g.bound, g.bound);
                d0 = step_len(g, t);
                float last;
                if (s == 0) last = start_param(g, nears[lo], noises ? noises[lo] : 0.0f);
                else { const float tp = rec[s - 1]; last = tp + step_len(g, tp); }
                d1 = (t + d0) - last;
            }
        }
        // three stores per row instead of eight (12 + 12 + 8 bytes)
        *reinterpret_cast<float3 *>(xyzs + 3 * (size_t)m) = make_float3(px, py, pz);
        *reinterpret_cast<float3 *>(dirs + 3 * (size_t)m) = make_float3(qx, qy, qz);
        *reinterpret_cast<float2 *>(deltas + 2 * (size_t)m) = make_float2(d0, d1);
    }
}

// Passes 2 + 3 in ONE launch for ray counts whose offsets fit in LDS (the training step's 4096 / 4608 rays): every workgroup computes the
// whole exclusive prefix sum of the counts for itself -- N int32 from L2 in batches of 16 loads per thread, a 256-thread scan -- instead of one single-workgroup launch between the walk and the writes (5 us alone, 60-90 us when its loads
// queue behind the optimiser's HBM stream, profiles/r02_f_kernel_stats.csv); a row's ray is then found by a binary search in LDS
// (13 dependent ~50-cycle reads instead of 13 dependent L2 round trips).  Workgroup 0 also stores the (id, offset, count) table
// and the totals.  Same ray-id order, same rows, same padding as k_march_scan + k_march_write.
// 256-thread workgroups ON PURPOSE: beside the block render's encoder (40 320 workgroups of 256 threads that backfill every freed wave
// slot) a 1024-thread workgroup needs half a CU's wave slots free AT ONCE and almost never finds them -- the first version of this kernel
// took 258 us there instead of 8 (profiles/r03_b_timeline_headline.txt), and so did the single 1024-thread scan workgroup before it (67 us).
__global__ void __launch_bounds__(256) k_march_scan_write(const float *__restrict__ rays_o, const float *__restrict__ rays_d, GridView g,
                                                          uint32_t max_steps, uint32_t N, uint32_t M, const float *__restrict__ nears,
                                                          const float *__restrict__ noises, const float *__restrict__ t_rec,
                                                          const int32_t *__restrict__ counts, int32_t *__restrict__ rays,
                                                          int32_t *__restrict__ counter, float *__restrict__ xyzs, float *__restrict__ dirs,
                                                          float *__restrict__ deltas) {
    extern __shared__ int32_t off[];      // [N + 1]: counts, then in place their exclusive prefix sums; off[N] = total
    __shared__ int32_t wave_tot[4];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    {   // all of a thread's loads in flight together (one L2 round trip per batch of 16, not one per element): N <= 48 * 256
        constexpr uint32_t kBatchLoads = 16;
        for (uint32_t base = 0; base < N; base += 256u * kBatchLoads) {
            int32_t v[kBatchLoads];
#pragma unroll
            for (uint32_t u = 0; u < kBatchLoads; ++u) {
                const uint32_t i = base + tid + 256u * u;
                v[u] = i < N ? counts[i] : 0;
            }
#pragma unroll
            for (uint32_t u = 0; u < kBatchLoads; ++u) {
                const uint32_t i = base + tid + 256u * u;
                if (i < N) off[i] = v[u];
            }
        }
    }
    __syncthreads();
    const uint32_t chunk = ceil_div(N, 256u);
    const uint32_t beg = min(N, tid * chunk), end = min(N, beg + chunk);
    int32_t sum = 0;
    for (uint32_t i = beg; i < end; ++i) sum += off[i];
    int32_t incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int32_t v = __shfl_up(incl, d, 64);
        if ((int)lane >= d) incl += v;
    }
    if (lane == 63) wave_tot[wid] = incl;
    __syncthreads();
    int32_t run = incl - sum;
#pragma unroll
    for (int w = 0; w < 4; ++w)
        if (w < (int)wid) run += wave_tot[w];
    const bool first = blockIdx.x == 0;
    for (uint32_t i = beg; i < end; ++i) {
        const int32_t c = off[i];
        off[i] = run;
        if (first) {
            rays[3 * (size_t)i] = (int32_t)i;
            rays[3 * (size_t)i + 1] = run;
            rays[3 * (size_t)i + 2] = c;
        }
        run += c;
    }
    if (tid == 255) {
        off[N] = run;
        if (first) { counter[0] = run; counter[1] = (int32_t)N; }
    }
    __syncthreads();
    const uint32_t total = (uint32_t)off[N];
    for (uint32_t m = blockIdx.x * blockDim.x + threadIdx.x; m < M; m += gridDim.x * blockDim.x) {
        float px = 0, py = 0, pz = 0, qx = 0, qy = 0, qz = 0, d0 = 0, d1 = 0;
        if (m < total) {
            uint32_t lo = 0, hi = N;      // last ray whose offset <= m (rays with no samples share their successor's offset)
            while (hi - lo > 1) {
                const uint32_t mid = (lo + hi) >> 1;
                if ((uint32_t)off[mid] <= m) lo = mid; else hi = mid;
            }
            const uint32_t o0 = (uint32_t)off[lo], cnt = (uint32_t)off[lo + 1] - o0;
            if (o0 + cnt <= M) {  // raymarching.cu:416: a ray that does not fit writes nothing
                const uint32_t s = m - o0;
                const float *rec = t_rec + (size_t)lo * max_steps;
                const float t = rec[s];
                const float3 rd = *reinterpret_cast<const float3 *>(rays_d + 3 * (size_t)lo), ro = *reinterpret_cast<const float3 *>(rays_o + 3 * (size_t)lo);
                qx = rd.x; qy = rd.y; qz = rd.z;
                px = clampf(fmaf(t, qx, ro.x), -g.bound, g.bound);
                py = clampf(fmaf(t, qy, ro.y), -g.bound, g.bound);
                pz = clampf(fmaf(t, qz, ro.z), -g.bound, g.bound);
                d0 = step_len(g, t);
                float last;
                if (s == 0) last = start_param(g, nears[lo], noises ? noises[lo] : 0.0f);
                else { const float tp = rec[s - 1]; last = tp + step_len(g, tp); }
                d1 = (t + d0) - last;
            }
        }
        *reinterpret_cast<float3 *>(xyzs + 3 * (size_t)m) = make_float3(px, py, pz);
        *reinterpret_cast<float3 *>(dirs + 3 * (size_t)m) = make_float3(qx, qy, qz);
        *reinterpret_cast<float2 *>(deltas + 2 * (size_t)m) = make_float2(d0, d1);
    }
}

// ----------------------------------------------------------------------------- compositing (training)
//
// One wave per ray.  The reference walks a ray's samples serially (raymarching.cu:540-567); here the 64 lanes take 64
// consecutive samples (coalesced loads), the transmittance T_j = prod_{i<j} (1 - alpha_i) comes from a wave-level
// prefix product, the running depth parameter and (backward) the running colour from prefix sums, and the early
// exit `T < T_thresh` becomes a prefix mask: sample j is accumulated iff the transmittance after sample j-1 is still
// >= T_thresh (the reference tests after accumulating, :557).  Products and sums associate differently from the
// serial loop, so results agree to fp32 round-off, not bit for bit (they never could: the reference uses __expf).

template <typename Op>
__device__ inline float wave_scan(float v, int lane, Op op) {  // inclusive scan over the 64 lanes
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const float t = __shfl_up(v, d, 64);
        if (lane >= d) v = op(v, t);
    }
    return v;
}

__device__ inline float wave_sum(float v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

struct Chunk {
    float alpha, w, T_after, c0, c1, c2, dt, dreal;
    bool valid, live;
};

// loads 64 samples of a ray and derives alpha, weight and transmittance; T_carry = transmittance entering the chunk
__device__ inline Chunk load_chunk(const float *__restrict__ sigmas, const float *__restrict__ rgbs, const float *__restrict__ deltas, size_t m0,
                                   uint32_t base, uint32_t cnt, int lane, float T_carry, float T_thresh) {
    Chunk c;
    c.valid = base + (uint32_t)lane < cnt;
    const size_t m = m0 + base + lane;
    float sigma = 0.0f;
    c.dt = c.dreal = c.c0 = c.c1 = c.c2 = 0.0f;
    if (c.valid) {
        sigma = sigmas[m];
        const float2 dl = reinterpret_cast<const float2 *>(deltas)[m];
        c.dt = dl.x; c.dreal = dl.y;
        c.c0 = rgbs[3 * m]; c.c1 = rgbs[3 * m + 1]; c.c2 = rgbs[3 * m + 2];
    }
    c.alpha = c.valid ? 1.0f - __expf(-sigma * c.dt) : 0.0f;
    const float P = wave_scan(1.0f - c.alpha, lane, [](float a, float b) { return a * b; });  // inclusive product
    float P_excl = __shfl_up(P, 1, 64);
    if (lane == 0) P_excl = 1.0f;
    const float T_before = T_carry * P_excl;
    c.T_after = T_carry * P;
    c.live = c.valid && T_before >= T_thresh;   // every earlier sample left T >= T_thresh (T_carry itself did)
    c.w = c.live ? c.alpha * T_before : 0.0f;
    return c;
}

// Optional tail of the render (renderer_wtmk.py:316-319) done by the ray's wave instead of a separate launch: background mix
// and depth normalisation in the forward, the weights_sum gradient of the background mix in the backward.
struct FinishArgs {
    const float *nears, *fars, *bg;   // bg == nullptr: no tail
    uint32_t bg_stride;               // 0: one colour [3]; 3: per ray
    float *image_out, *depth_out;     // forward outputs
    uint32_t self_zero;               // backward: the kernel zero-fills every gradient row it does not write (no memsets before it)
};

__global__ void __launch_bounds__(256) k_composite_fwd(const float *__restrict__ sigmas, const float *__restrict__ rgbs,
                                                       const float *__restrict__ deltas, const int32_t *__restrict__ rays, uint32_t M,
                                                       uint32_t N, float T_thresh, float *__restrict__ weights_sum,
                                                       float *__restrict__ depth, float *__restrict__ image, FinishArgs fin = FinishArgs{}) {
    const int lane = threadIdx.x & 63;
    const uint32_t n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const uint32_t id = (uint32_t)rays[3 * (size_t)n], off = (uint32_t)rays[3 * (size_t)n + 1], cnt = (uint32_t)rays[3 * (size_t)n + 2];
    float r = 0, g = 0, b = 0, ws = 0, d = 0, T = 1.0f, tt = 0.0f;
    if (cnt != 0 && off + cnt <= M) {
        for (uint32_t base = 0; base < cnt; base += 64) {
            const Chunk c = load_chunk(sigmas, rgbs, deltas, off, base, cnt, lane, T, T_thresh);
            const float t_incl = tt + wave_scan(c.dreal, lane, [](float a, float b2) { return a + b2; });  // accumulated real deltas (:549)
            r += c.w * c.c0; g += c.w * c.c1; b += c.w * c.c2; ws += c.w; d += c.w * t_incl;
            T = __shfl(c.T_after, 63, 64);
            tt = __shfl(t_incl, 63, 64);
            if (T < T_thresh) break;  // some sample of this chunk ended the ray
        }
        r = wave_sum(r); g = wave_sum(g); b = wave_sum(b); ws = wave_sum(ws); d = wave_sum(d);
    }
    if (lane == 0) {
        weights_sum[id] = ws;
        depth[id] = d;
        image[3 * (size_t)id] = r; image[3 * (size_t)id + 1] = g; image[3 * (size_t)id + 2] = b;
        if (fin.bg != nullptr) {   // the arithmetic of k_finish_fwd, operation by operation
            const float rest = 1.0f - ws;
            const float *bgp = fin.bg + (size_t)id * fin.bg_stride;
            fin.image_out[3 * (size_t)id] = r + rest * bgp[0];
            fin.image_out[3 * (size_t)id + 1] = g + rest * bgp[1];
            fin.image_out[3 * (size_t)id + 2] = b + rest * bgp[2];
            fin.depth_out[id] = fmaxf(d - fin.nears[id], 0.0f) / (fin.fars[id] - fin.nears[id]);
        }
    }
}

// raymarching.cu:602-682 (grad_depth does not propagate, raymarching.py:275).  grad buffers are pre-zeroed by the host
// wrapper; only live samples are written.
__global__ void __launch_bounds__(256) k_composite_bwd(const float *__restrict__ grad_ws, const float *__restrict__ grad_image,
                                                       const float *__restrict__ sigmas, const float *__restrict__ rgbs,
                                                       const float *__restrict__ deltas, const int32_t *__restrict__ rays,
                                                       const float *__restrict__ weights_sum, const float *__restrict__ image, uint32_t M,
                                                       uint32_t N, float T_thresh, float *__restrict__ grad_sigmas,
                                                       float *__restrict__ grad_rgbs, FinishArgs fin = FinishArgs{}) {
    const int lane = threadIdx.x & 63;
    if (fin.self_zero && blockIdx.x >= ceil_div(N, 4u)) {
        // tail blocks: rows past the last ray's samples (padding / unused capacity).  Rays are in ascending, gapless offset order.
        const uint32_t total = (uint32_t)rays[3 * (size_t)(N - 1) + 1] + (uint32_t)rays[3 * (size_t)(N - 1) + 2];
        for (size_t m = (size_t)total + (size_t)(blockIdx.x - ceil_div(N, 4u)) * 256 + threadIdx.x; m < M; m += (size_t)(gridDim.x - ceil_div(N, 4u)) * 256) {
            grad_sigmas[m] = 0.0f;
            grad_rgbs[3 * m] = 0.0f; grad_rgbs[3 * m + 1] = 0.0f; grad_rgbs[3 * m + 2] = 0.0f;
        }
        return;
    }
    const uint32_t n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const uint32_t id = (uint32_t)rays[3 * (size_t)n], off = (uint32_t)rays[3 * (size_t)n + 1], cnt = (uint32_t)rays[3 * (size_t)n + 2];
    auto zero_rows = [&](uint32_t first, uint32_t last) {   // [first, last) of this ray's range, clipped to M
        for (size_t m = (size_t)off + first + lane; m < (size_t)off + last && m < M; m += 64) {
            grad_sigmas[m] = 0.0f;
            grad_rgbs[3 * m] = 0.0f; grad_rgbs[3 * m + 1] = 0.0f; grad_rgbs[3 * m + 2] = 0.0f;
        }
    };
    if (cnt == 0) return;
    if (off + cnt > M) {   // a ray that did not fit (bounded mode): no gradient, but its rows below M belong to nobody else
        if (fin.self_zero) zero_rows(0, cnt);
        return;
    }
    const float g0 = grad_image[3 * (size_t)id], g1 = grad_image[3 * (size_t)id + 1], g2 = grad_image[3 * (size_t)id + 2];
    const float rf = image[3 * (size_t)id], gf = image[3 * (size_t)id + 1], bf = image[3 * (size_t)id + 2];
    float gws = grad_ws != nullptr ? grad_ws[id] : 0.0f;
    if (fin.bg != nullptr) {   // image_out = image + (1 - weights_sum) * bg  =>  d weights_sum -= sum_c grad_image_c * bg_c (k_finish_bwd)
        const float *bgp = fin.bg + (size_t)id * fin.bg_stride;
        float sgb = 0.0f;
        sgb += g0 * bgp[0]; sgb += g1 * bgp[1]; sgb += g2 * bgp[2];
        gws = gws + (-sgb);
    }
    const float tail = gws * (1.0f - weights_sum[id]);
    float T = 1.0f, r = 0.0f, g = 0.0f, b = 0.0f;  // carries: transmittance and accumulated colour entering the chunk
    auto add = [](float a, float c) { return a + c; };
    for (uint32_t base = 0; base < cnt; base += 64) {
        const Chunk c = load_chunk(sigmas, rgbs, deltas, off, base, cnt, lane, T, T_thresh);
        const float r_incl = r + wave_scan(c.w * c.c0, lane, add);
        const float g_incl = g + wave_scan(c.w * c.c1, lane, add);
        const float b_incl = b + wave_scan(c.w * c.c2, lane, add);
        if (c.live) {
            const size_t m = (size_t)off + base + lane;
            grad_rgbs[3 * m] = g0 * c.w; grad_rgbs[3 * m + 1] = g1 * c.w; grad_rgbs[3 * m + 2] = g2 * c.w;
            grad_sigmas[m] = c.dt * (g0 * (c.T_after * c.c0 - (rf - r_incl)) + g1 * (c.T_after * c.c1 - (gf - g_incl)) +
                                     g2 * (c.T_after * c.c2 - (bf - b_incl)) + tail);
        } else if (fin.self_zero && base + lane < cnt) {
            const size_t m = (size_t)off + base + lane;
            grad_sigmas[m] = 0.0f;
            grad_rgbs[3 * m] = 0.0f; grad_rgbs[3 * m + 1] = 0.0f; grad_rgbs[3 * m + 2] = 0.0f;
        }
        T = __shfl(c.T_after, 63, 64);
        r = __shfl(r_incl, 63, 64); g = __shfl(g_incl, 63, 64); b = __shfl(b_incl, 63, 64);
        if (T < T_thresh) {   // the ray ended in this chunk: later samples get no gradient
            if (fin.self_zero) zero_rows(base + 64, cnt);
            break;
        }
    }
}

// ----------------------------------------------------------------------------- inference

// raymarching.cu:701-805: march up to n_step occupied samples from rays_t[id].
__global__ void k_march_burst(uint32_t n_alive, uint32_t n_step, const int32_t *__restrict__ rays_alive,
                              const float *__restrict__ rays_t, const float *__restrict__ rays_o,
                              const float *__restrict__ rays_d, GridView g, const float *__restrict__ fars,
                              float *__restrict__ xyzs, float *__restrict__ dirs, float *__restrict__ deltas,
                              const float *__restrict__ noises) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= n_alive) return;
    const int32_t id = rays_alive[n];
    const Ray r(rays_o + 3 * (size_t)id, rays_d + 3 * (size_t)id);
    const float far = fars[id];
    float t = start_param(g, rays_t[id], noises ? noises[n] : 0.0f);
    float last = t;
    float *px = xyzs + 3 * (size_t)n * n_step, *pd = dirs + 3 * (size_t)n * n_step, *pl = deltas + 2 * (size_t)n * n_step;
    uint32_t step = 0;
    float x, y, z, dt, t_exit;
    while (t < far && step < n_step) {
        if (probe(g, r, t, x, y, z, dt, t_exit)) {
            px[0] = x; px[1] = y; px[2] = z;
            pd[0] = r.dx; pd[1] = r.dy; pd[2] = r.dz;
            t += dt;
            pl[0] = dt; pl[1] = t - last;
            last = t;
            px += 3; pd += 3; pl += 2; ++step;
        } else t = leave_cell(g, t, t_exit);
    }
}

// The eval loop with its control state on the device (rm_eval_*): ctl = {n_alive, n_step, rows = n_alive * n_step, samples marched so far}.  A round's
// launches are sized for the worst case (n_alive * n_step <= N rays) and early-out on the counts; the host reads ctl back only now and then, to stop.
__global__ void k_march_burst_ctl(const uint32_t *__restrict__ ctl, const int32_t *__restrict__ rays_alive, const float *__restrict__ rays_t,
                                  const float *__restrict__ rays_o, const float *__restrict__ rays_d, GridView g, const float *__restrict__ fars,
                                  float *__restrict__ xyzs, float *__restrict__ dirs, float *__restrict__ deltas, const float *__restrict__ noises) {
    const uint32_t n_alive = ctl[0], n_step = ctl[1];
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= n_alive) return;
    const int32_t id = rays_alive[n];
    const Ray r(rays_o + 3 * (size_t)id, rays_d + 3 * (size_t)id);
    const float far = fars[id];
    float t = start_param(g, rays_t[id], noises ? noises[n] : 0.0f);
    float last = t;
    float *px = xyzs + 3 * (size_t)n * n_step, *pd = dirs + 3 * (size_t)n * n_step, *pl = deltas + 2 * (size_t)n * n_step;
    uint32_t step = 0;
    float x, y, z, dt, t_exit;
    while (t < far && step < n_step) {
        if (probe(g, r, t, x, y, z, dt, t_exit)) {
            px[0] = x; px[1] = y; px[2] = z;
            pd[0] = r.dx; pd[1] = r.dy; pd[2] = r.dz;
            t += dt;
            pl[0] = dt; pl[1] = t - last;
            last = t;
            px += 3; pd += 3; pl += 2; ++step;
        } else t = leave_cell(g, t, t_exit);
    }
    for (; step < n_step; ++step) {      // the ray ended inside the burst: zero rows (rm_march zero-fills its whole buffers up front; the compositor stops at delta 0)
        px[0] = px[1] = px[2] = 0.0f;
        pd[0] = pd[1] = pd[2] = 0.0f;
        pl[0] = pl[1] = 0.0f;
        px += 3; pd += 3; pl += 2;
    }
}

// Stable compaction (as k_compact_alive) + the next round's control words: n_alive' = survivors (0 once max_steps samples have been marched:
// renderer_wtmk.py:335), n_step' = clamp(N / n_alive', 1, 8) (:340), rows', samples += n_step.
__global__ void __launch_bounds__(1024) k_compact_alive_ctl(uint32_t *__restrict__ ctl, const int32_t *__restrict__ in, int32_t *__restrict__ out, uint32_t N,
                                                            uint32_t max_steps) {
    __shared__ uint32_t wave_cnt[16];
    __shared__ uint32_t running;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const uint32_t n = ctl[0], n_step = ctl[1], marched = ctl[3];
    if (tid == 0) running = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n; base += 1024) {
        const uint32_t i = base + tid;
        const int32_t v = i < n ? in[i] : -1;
        const bool keep = v >= 0;
        const unsigned long long ballot = __ballot(keep);
        const uint32_t before = __popcll(ballot & ((1ull << lane) - 1ull));
        if (lane == 0) wave_cnt[wid] = __popcll(ballot);
        __syncthreads();
        uint32_t wave_base = running, round_total = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            const uint32_t c = wave_cnt[w];
            if (w < (int)wid) wave_base += c;
            round_total += c;
        }
        if (keep) out[wave_base + before] = v;
        __syncthreads();
        if (tid == 0) running += round_total;
        __syncthreads();
    }
    if (tid == 0 && n > 0) {
        const uint32_t done = marched + n_step;
        const uint32_t alive = done < max_steps ? running : 0u;
        const uint32_t next = alive ? min(max(N / alive, 1u), 8u) : 1u;
        ctl[0] = alive;
        ctl[1] = next;
        ctl[2] = alive * next;
        ctl[3] = done;
    }
}

// raymarching.cu:819-905: in-place accumulation, T = 1 - weight_sum.
__global__ void k_composite_burst(uint32_t n_alive, uint32_t n_step, float T_thresh, int32_t *__restrict__ rays_alive,
                                  float *__restrict__ rays_t, const float *__restrict__ sigmas,
                                  const float *__restrict__ rgbs, const float *__restrict__ deltas,
                                  float *__restrict__ weights_sum, float *__restrict__ depth, float *__restrict__ image,
                                  const uint32_t *__restrict__ ctl = nullptr, float density_scale = 1.0f) {
    if (ctl != nullptr) {       // (rm_eval_composite: the counts live on the device; sigma is scaled here -- renderer_wtmk.py:353 -- instead of by a pass of its own)
        n_alive = ctl[0];
        n_step = ctl[1];
    }
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= n_alive) return;
    const int32_t id = rays_alive[n];
    const float *ps = sigmas + (size_t)n * n_step, *pc = rgbs + 3 * (size_t)n * n_step, *pl = deltas + 2 * (size_t)n * n_step;
    float t = rays_t[id], wsum = weights_sum[id], d = depth[id];
    float r = image[3 * (size_t)id], g = image[3 * (size_t)id + 1], b = image[3 * (size_t)id + 2];
    uint32_t step = 0;
    while (step < n_step) {
        if (pl[0] == 0.0f) break;
        const float alpha = 1.0f - __expf(-(ctl != nullptr ? density_scale * ps[0] : ps[0]) * pl[0]);
        const float T = 1.0f - wsum;
        const float w = alpha * T;
        wsum += w;
        t += pl[1];
        d = fmaf(w, t, d);
        r = fmaf(w, pc[0], r);
        g = fmaf(w, pc[1], g);
        b = fmaf(w, pc[2], b);
        if (T < T_thresh) break;
        ++ps; pc += 3; pl += 2; ++step;
    }
    if (step < n_step) rays_alive[n] = -1;
    else rays_t[id] = t;
    weights_sum[id] = wsum;
    depth[id] = d;
    image[3 * (size_t)id] = r; image[3 * (size_t)id + 1] = g; image[3 * (size_t)id + 2] = b;
}

__global__ void k_eval_begin(uint32_t N, uint32_t *__restrict__ ctl, int32_t *__restrict__ rays_alive) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) rays_alive[i] = (int32_t)i;
    if (i == 0) {
        ctl[0] = N;
        ctl[1] = 1u;
        ctl[2] = N;
        ctl[3] = 0u;
    }
}

// Stable compaction of the non-negative ray ids: ballot + popcount inside a wave, wave bases through LDS,
// a running base across 1024-element rounds.  One workgroup (alive lists are at most a few 10^5 long).
__global__ void __launch_bounds__(1024) k_compact_alive(const int32_t *__restrict__ in, uint32_t n,
                                                        int32_t *__restrict__ out, int32_t *__restrict__ n_out) {
    __shared__ uint32_t wave_cnt[16];
    __shared__ uint32_t running;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (tid == 0) running = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n; base += 1024) {
        const uint32_t i = base + tid;
        const int32_t v = i < n ? in[i] : -1;
        const bool keep = v >= 0;
        const unsigned long long ballot = __ballot(keep);
        const uint32_t before = __popcll(ballot & ((1ull << lane) - 1ull));
        if (lane == 0) wave_cnt[wid] = __popcll(ballot);
        __syncthreads();
        uint32_t wave_base = running, round_total = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            const uint32_t c = wave_cnt[w];
            if (w < (int)wid) wave_base += c;
            round_total += c;
        }
        if (keep) out[wave_base + before] = v;
        __syncthreads();
        if (tid == 0) running += round_total;
        __syncthreads();
    }
    if (tid == 0) *n_out = (int32_t)running;
}

}  // namespace nsig

// ============================================================================= C ABI

using namespace nsig;

static inline uint32_t lanes_for_walk(uint32_t n) {
    // active lanes per wave for the latency-bound grid walk: spread small batches over all CUs
    uint32_t lanes = 64;
    while (lanes > 16 && (uint64_t)ceil_div(n, lanes) < 4ull * kCUs) lanes >>= 1;
    return lanes;
}

NSIG_EXPORT int rm_near_far_from_aabb(const float *rays_o, const float *rays_d, const float *aabb, uint32_t N,
                                      float min_near, float *nears, float *fars, nsig_stream_t stream) {
    if (N == 0) return NSIG_OK;
    NSIG_REQUIRE(rays_o && rays_d && aabb && nears && fars, "rm_near_far_from_aabb: null pointer");
    k_near_far<<<ceil_div(N, 256), 256, 0, as_stream(stream)>>>(rays_o, rays_d, aabb, N, min_near, nears, fars);
    return check_launch("rm_near_far_from_aabb");
}

NSIG_EXPORT int rm_sph_from_ray(const float *rays_o, const float *rays_d, float radius, uint32_t N, float *coords,
                                nsig_stream_t stream) {
    if (N == 0) return NSIG_OK;
    NSIG_REQUIRE(rays_o && rays_d && coords, "rm_sph_from_ray: null pointer");
    k_sph_from_ray<<<ceil_div(N, 256), 256, 0, as_stream(stream)>>>(rays_o, rays_d, radius, N, coords);
    return check_launch("rm_sph_from_ray");
}

NSIG_EXPORT int rm_morton3D(const int32_t *coords, uint32_t N, int32_t *indices, nsig_stream_t stream) {
    if (N == 0) return NSIG_OK;
    NSIG_REQUIRE(coords && indices, "rm_morton3D: null pointer");
    k_morton<<<ceil_div(N, 256), 256, 0, as_stream(stream)>>>(coords, N, indices);
    return check_launch("rm_morton3D");
}

NSIG_EXPORT int rm_morton3D_invert(const int32_t *indices, uint32_t N, int32_t *coords, nsig_stream_t stream) {
    if (N == 0) return NSIG_OK;
    NSIG_REQUIRE(coords && indices, "rm_morton3D_invert: null pointer");
    k_morton_invert<<<ceil_div(N, 256), 256, 0, as_stream(stream)>>>(indices, N, coords);
    return check_launch("rm_morton3D_invert");
}

NSIG_EXPORT int rm_packbits(const float *grid, uint32_t n_bytes, float density_thresh, uint8_t *bitfield,
                            nsig_stream_t stream) {
    NSIG_REQUIRE(grid && bitfield, "rm_packbits: null pointer");
    NSIG_REQUIRE((reinterpret_cast<uintptr_t>(grid) & 15) == 0, "rm_packbits: grid must be 16-byte aligned");
    if (n_bytes == 0) return NSIG_OK;
    k_packbits<<<ceil_div(ceil_div(n_bytes, 4), 256), 256, 0, as_stream(stream)>>>(grid, n_bytes, density_thresh, bitfield);
    return check_launch("rm_packbits");
}

NSIG_EXPORT size_t rm_march_train_scratch_bytes(uint32_t N, uint32_t max_steps) {
    return (size_t)N * max_steps * sizeof(float);
}

static int check_grid_args(const char *who, uint32_t C, uint32_t H, uint32_t max_steps, float bound) {
    NSIG_REQUIRE(C >= 1 && C <= 8, "%s: cascade count %u out of range [1,8]", who, C);
    NSIG_REQUIRE(H >= 8 && H <= 1024 && (H & (H - 1)) == 0, "%s: grid size %u must be a power of two in [8,1024]", who, H);
    NSIG_REQUIRE(max_steps >= 1, "%s: max_steps must be positive", who);
    NSIG_REQUIRE(bound > 0.0f, "%s: bound must be positive", who);
    return NSIG_OK;
}

static int launch_march_index(const char *who, const float *rays_o, const float *rays_d, const uint8_t *grid, float bound, float dt_gamma,
                              uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, float *nears, float *fars, const float *noises,
                              int32_t *counts, float *t_rec, const float *aabb, float min_near, nsig_stream_t stream) {
    if (int e = check_grid_args(who, C, H, max_steps, bound)) return e;
    const GridView gv = make_grid_view(grid, bound, dt_gamma, max_steps, C, H);
    const uint32_t blocks = ceil_div(N, 4u);
    hipStream_t st = as_stream(stream);
    if (C == 1 && dt_gamma == 0.0f) k_march_index<true, true><<<blocks, 256, 0, st>>>(rays_o, rays_d, gv, max_steps, N, nears, fars, noises, counts, t_rec, aabb, min_near);
    else if (C == 1) k_march_index<true, false><<<blocks, 256, 0, st>>>(rays_o, rays_d, gv, max_steps, N, nears, fars, noises, counts, t_rec, aabb, min_near);
    else if (dt_gamma == 0.0f) k_march_index<false, true><<<blocks, 256, 0, st>>>(rays_o, rays_d, gv, max_steps, N, nears, fars, noises, counts, t_rec, aabb, min_near);
    else k_march_index<false, false><<<blocks, 256, 0, st>>>(rays_o, rays_d, gv, max_steps, N, nears, fars, noises, counts, t_rec, aabb, min_near);
    return check_launch(who);
}

NSIG_EXPORT int rm_march_train_count(const float *rays_o, const float *rays_d, const uint8_t *grid, float bound,
                                     float dt_gamma, uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H,
                                     const float *nears, const float *fars, const float *noises, int32_t *counts,
                                     float *t_rec, nsig_stream_t stream) {
    if (N == 0) return NSIG_OK;
    NSIG_REQUIRE(rays_o && rays_d && grid && nears && fars && counts && t_rec, "rm_march_train_count: null pointer");
    return launch_march_index("rm_march_train_count", rays_o, rays_d, grid, bound, dt_gamma, max_steps, N, C, H, const_cast<float *>(nears),
                              const_cast<float *>(fars), noises, counts, t_rec, nullptr, 0.0f, stream);
}

NSIG_EXPORT int rm_march_train_count_nf(const float *rays_o, const float *rays_d, const float *aabb, float min_near, const uint8_t *grid,
                                        float bound, float dt_gamma, uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H,
                                        const float *noises, float *nears, float *fars, int32_t *counts, float *t_rec, nsig_stream_t stream) {
    if (N == 0) return NSIG_OK;
    NSIG_REQUIRE(rays_o && rays_d && aabb && grid && nears && fars && counts && t_rec, "rm_march_train_count_nf: null pointer");
    return launch_march_index("rm_march_train_count_nf", rays_o, rays_d, grid, bound, dt_gamma, max_steps, N, C, H, nears, fars, noises, counts, t_rec,
                              aabb, min_near, stream);
}

NSIG_EXPORT int rm_march_train_scan(const int32_t *counts, uint32_t N, int32_t *rays, int32_t *counter,
                                    nsig_stream_t stream) {
    NSIG_REQUIRE(counts && rays && counter, "rm_march_train_scan: null pointer");
    k_march_scan<<<1, 1024, 0, as_stream(stream)>>>(counts, N, rays, counter);
    return check_launch("rm_march_train_scan");
}

// The same table for many rays in two launches of 4096-ray workgroups; `block_sums` is caller-owned scratch of
// rm_march_train_scan_blocks(N) int32 words (one total per workgroup), need not be initialised.
NSIG_EXPORT int rm_march_train_scan_blocks(uint32_t N) { return (int)ceil_div(N, kScanChunk); }
NSIG_EXPORT int rm_march_train_scan_wide(const int32_t *counts, uint32_t N, int32_t *rays, int32_t *counter, int32_t *block_sums,
                                         nsig_stream_t stream) {
    NSIG_REQUIRE(counts && rays && counter && block_sums, "rm_march_train_scan_wide: null pointer");
    NSIG_REQUIRE(N >= 1, "rm_march_train_scan_wide: N must be positive");
    const uint32_t nb = ceil_div(N, kScanChunk);
    k_march_scan_sums<<<nb, 1024, 0, as_stream(stream)>>>(counts, N, block_sums);
    k_march_scan_apply<<<nb, 1024, 0, as_stream(stream)>>>(counts, N, block_sums, rays, counter);
    return check_launch("rm_march_train_scan_wide");
}

NSIG_EXPORT int rm_march_train_write(const float *rays_o, const float *rays_d, float bound, float dt_gamma,
                                     uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, uint32_t M,
                                     const float *nears, const float *noises, const float *t_rec, const int32_t *rays,
                                     const int32_t *counter, float *xyzs, float *dirs, float *deltas,
                                     nsig_stream_t stream) {
    NSIG_REQUIRE(rays_o && rays_d && nears && t_rec && rays && counter && xyzs && dirs && deltas,
                 "rm_march_train_write: null pointer");
    if (int e = check_grid_args("rm_march_train_write", C, H, max_steps, bound)) return e;
    if (M == 0 || N == 0) return NSIG_OK;
    const uint32_t blocks = min(ceil_div(M, 256), (uint32_t)(kCUs * 8));
    k_march_write<<<blocks, 256, 0, as_stream(stream)>>>(rays_o, rays_d, make_grid_view(nullptr, bound, dt_gamma, max_steps, C, H),
                                                        max_steps, N, M, nears, noises, t_rec, rays, counter, xyzs, dirs, deltas);
    return check_launch("rm_march_train_write");
}

NSIG_EXPORT int rm_march_train_scan_write_max_rays(void) { return (int)kScanWriteMaxRays; }

NSIG_EXPORT int rm_march_train_scan_write(const float *rays_o, const float *rays_d, float bound, float dt_gamma, uint32_t max_steps, uint32_t N,
                                          uint32_t C, uint32_t H, uint32_t M, const float *nears, const float *noises, const float *t_rec,
                                          const int32_t *counts, int32_t *rays, int32_t *counter, float *xyzs, float *dirs, float *deltas,
                                          nsig_stream_t stream) {
    NSIG_REQUIRE(rays_o && rays_d && nears && t_rec && counts && rays && counter, "rm_march_train_scan_write: null pointer");
    NSIG_REQUIRE(M == 0 || (xyzs && dirs && deltas), "rm_march_train_scan_write: null point buffers");
    NSIG_REQUIRE(N >= 1 && N <= kScanWriteMaxRays, "rm_march_train_scan_write: N=%u outside [1, %u] (use rm_march_train_scan + rm_march_train_write)", N, kScanWriteMaxRays);
    if (int e = check_grid_args("rm_march_train_scan_write", C, H, max_steps, bound)) return e;
    const uint32_t blocks = max(1u, min(ceil_div(M, 1024), (uint32_t)(kCUs * 4)));     // (M == 0: workgroup 0 still writes the ray table and the totals)
    k_march_scan_write<<<blocks, 256, (N + 1) * sizeof(int32_t), as_stream(stream)>>>(rays_o, rays_d, make_grid_view(nullptr, bound, dt_gamma, max_steps, C, H),
                                                                                      max_steps, N, M, nears, noises, t_rec, counts, rays, counter, xyzs, dirs, deltas);
    return check_launch("rm_march_train_scan_write");
}

NSIG_EXPORT int rm_composite_train_fwd(const float *sigmas, const float *rgbs, const float *deltas,
                                       const int32_t *rays, uint32_t M, uint32_t N, float T_thresh, float *weights_sum,
                                       float *depth, float *image, nsig_stream_t stream) {
    if (N == 0) return NSIG_OK;
    NSIG_REQUIRE(sigmas && rgbs && deltas && rays && weights_sum && depth && image, "rm_composite_train_fwd: null pointer");
    k_composite_fwd<<<ceil_div(N, 4u), 256, 0, as_stream(stream)>>>(sigmas, rgbs, deltas, rays, M, N, T_thresh, weights_sum, depth, image);
    return check_launch("rm_composite_train_fwd");
}

NSIG_EXPORT int rm_composite_train_finish_fwd(const float *sigmas, const float *rgbs, const float *deltas, const int32_t *rays, uint32_t M,
                                              uint32_t N, float T_thresh, const float *nears, const float *fars, const float *bg,
                                              uint32_t bg_stride, float *weights_sum, float *depth, float *image, float *image_out,
                                              float *depth_out, nsig_stream_t stream) {
    if (N == 0) return NSIG_OK;
    NSIG_REQUIRE(sigmas && rgbs && deltas && rays && weights_sum && depth && image && nears && fars && bg && image_out && depth_out,
                 "rm_composite_train_finish_fwd: null pointer");
    NSIG_REQUIRE(bg_stride == 0 || bg_stride == 3, "rm_composite_train_finish_fwd: bg_stride is 0 (one colour) or 3 (per ray)");
    const FinishArgs fin{nears, fars, bg, bg_stride, image_out, depth_out, 0u};
    k_composite_fwd<<<ceil_div(N, 4u), 256, 0, as_stream(stream)>>>(sigmas, rgbs, deltas, rays, M, N, T_thresh, weights_sum, depth, image, fin);
    return check_launch("rm_composite_train_finish_fwd");
}

NSIG_EXPORT int rm_composite_train_finish_bwd(const float *grad_weights_sum, const float *grad_image_out, const float *sigmas, const float *rgbs,
                                              const float *deltas, const int32_t *rays, const float *weights_sum, const float *image,
                                              const float *bg, uint32_t bg_stride, uint32_t M, uint32_t N, float T_thresh, uint32_t rays_in_order,
                                              float *grad_sigmas, float *grad_rgbs, nsig_stream_t stream) {
    if (M == 0) return NSIG_OK;
    NSIG_REQUIRE(grad_image_out && sigmas && rgbs && deltas && rays && weights_sum && image && bg && grad_sigmas && grad_rgbs,
                 "rm_composite_train_finish_bwd: null pointer");
    NSIG_REQUIRE(bg_stride == 0 || bg_stride == 3, "rm_composite_train_finish_bwd: bg_stride is 0 (one colour) or 3 (per ray)");
    const bool self_zero = rays_in_order != 0 && N != 0;
    if (!self_zero && (hipMemsetAsync(grad_sigmas, 0, (size_t)M * sizeof(float), as_stream(stream)) != hipSuccess ||
                       hipMemsetAsync(grad_rgbs, 0, (size_t)M * 3 * sizeof(float), as_stream(stream)) != hipSuccess)) {
        set_error("rm_composite_train_finish_bwd: hipMemsetAsync failed");
        return NSIG_ERR_LAUNCH;
    }
    if (N == 0) return NSIG_OK;
    const FinishArgs fin{nullptr, nullptr, bg, bg_stride, nullptr, nullptr, self_zero ? 1u : 0u};
    k_composite_bwd<<<ceil_div(N, 4u) + (self_zero ? 64u : 0u), 256, 0, as_stream(stream)>>>(grad_weights_sum, grad_image_out, sigmas, rgbs, deltas, rays, weights_sum, image, M, N,
                                                                    T_thresh, grad_sigmas, grad_rgbs, fin);
    return check_launch("rm_composite_train_finish_bwd");
}

// Stage 1's compositing in ONE launch (stage1.GraphedCleanLoop): k_composite_fwd with the render tail, the gradient half of k_clean_loss
// (nerf/utils.py:503: d loss / d image = grad_k * (image - gt), grad_k = grad_scale * 2 / n_values) and k_composite_bwd with the tail's adjoint, for rays in
// ascending gapless offset order.  A wave owns a ray in all three, so the ray's image never leaves its registers: forward over the ray's chunks, the three seed
// values, backward over the same chunks (read again, from L2).  The same arithmetic, operation by operation, as the three launches: the same bits in every output
// (tests/test_gpu_stage1.py).  The loss VALUE and the loop's books stay with clean_loss, launched behind this kernel: two launches in a row where there were three.
__global__ void __launch_bounds__(256) k_composite_mse(const float *__restrict__ sigmas, const float *__restrict__ rgbs, const float *__restrict__ deltas,
                                                       const int32_t *__restrict__ rays, uint32_t M, uint32_t N, float T_thresh, const float *__restrict__ gt,
                                                       float grad_k, FinishArgs fin, float *__restrict__ weights_sum, float *__restrict__ depth,
                                                       float *__restrict__ image, float *__restrict__ grad_image, float *__restrict__ grad_sigmas,
                                                       float *__restrict__ grad_rgbs) {
    const int lane = threadIdx.x & 63;
    if (blockIdx.x >= ceil_div(N, 4u)) {      // tail blocks: rows past the last ray's samples (k_composite_bwd's self_zero)
        const uint32_t total = (uint32_t)rays[3 * (size_t)(N - 1) + 1] + (uint32_t)rays[3 * (size_t)(N - 1) + 2];
        for (size_t m = (size_t)total + (size_t)(blockIdx.x - ceil_div(N, 4u)) * 256 + threadIdx.x; m < M; m += (size_t)(gridDim.x - ceil_div(N, 4u)) * 256) {
            grad_sigmas[m] = 0.0f;
            grad_rgbs[3 * m] = 0.0f; grad_rgbs[3 * m + 1] = 0.0f; grad_rgbs[3 * m + 2] = 0.0f;
        }
        return;
    }
    const uint32_t n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const uint32_t id = (uint32_t)rays[3 * (size_t)n], off = (uint32_t)rays[3 * (size_t)n + 1], cnt = (uint32_t)rays[3 * (size_t)n + 2];
    const bool fits = cnt != 0 && off + cnt <= M;
    // ---- forward (k_composite_fwd)
    float r = 0, g = 0, b = 0, ws = 0, d = 0, T = 1.0f, tt = 0.0f;
    if (fits) {
        for (uint32_t base = 0; base < cnt; base += 64) {
            const Chunk c = load_chunk(sigmas, rgbs, deltas, off, base, cnt, lane, T, T_thresh);
            const float t_incl = tt + wave_scan(c.dreal, lane, [](float a, float b2) { return a + b2; });
            r += c.w * c.c0; g += c.w * c.c1; b += c.w * c.c2; ws += c.w; d += c.w * t_incl;
            T = __shfl(c.T_after, 63, 64);
            tt = __shfl(t_incl, 63, 64);
            if (T < T_thresh) break;
        }
        r = wave_sum(r); g = wave_sum(g); b = wave_sum(b); ws = wave_sum(ws); d = wave_sum(d);
    }
    const float rest = 1.0f - ws;
    const float *bgp = fin.bg + (size_t)id * fin.bg_stride;
    const float bg0 = bgp[0], bg1 = bgp[1], bg2 = bgp[2];
    const float o0 = r + rest * bg0, o1 = g + rest * bg1, o2 = b + rest * bg2;
    // ---- the seed (k_clean_loss: g_image[i] = k * (image[i] - gt[i]))
    const float g0 = grad_k * (o0 - gt[3 * (size_t)id]), g1 = grad_k * (o1 - gt[3 * (size_t)id + 1]), g2 = grad_k * (o2 - gt[3 * (size_t)id + 2]);
    if (lane == 0) {
        weights_sum[id] = ws;
        depth[id] = d;
        image[3 * (size_t)id] = r; image[3 * (size_t)id + 1] = g; image[3 * (size_t)id + 2] = b;
        fin.image_out[3 * (size_t)id] = o0; fin.image_out[3 * (size_t)id + 1] = o1; fin.image_out[3 * (size_t)id + 2] = o2;
        fin.depth_out[id] = fmaxf(d - fin.nears[id], 0.0f) / (fin.fars[id] - fin.nears[id]);
        grad_image[3 * (size_t)id] = g0; grad_image[3 * (size_t)id + 1] = g1; grad_image[3 * (size_t)id + 2] = g2;
    }
    // ---- backward (k_composite_bwd with the tail's adjoint and self_zero)
    auto zero_rows = [&](uint32_t first, uint32_t last) {
        for (size_t m = (size_t)off + first + lane; m < (size_t)off + last && m < M; m += 64) {
            grad_sigmas[m] = 0.0f;
            grad_rgbs[3 * m] = 0.0f; grad_rgbs[3 * m + 1] = 0.0f; grad_rgbs[3 * m + 2] = 0.0f;
        }
    };
    if (cnt == 0) return;
    if (!fits) {
        zero_rows(0, cnt);
        return;
    }
    float sgb = 0.0f;
    sgb += g0 * bg0; sgb += g1 * bg1; sgb += g2 * bg2;
    const float gws = 0.0f + (-sgb);
    const float tail = gws * (1.0f - ws);
    const float rf = r, gf = g, bf = b;
    T = 1.0f;
    float ra = 0.0f, ga = 0.0f, ba = 0.0f;
    auto add = [](float a, float c) { return a + c; };
    for (uint32_t base = 0; base < cnt; base += 64) {
        const Chunk c = load_chunk(sigmas, rgbs, deltas, off, base, cnt, lane, T, T_thresh);
        const float r_incl = ra + wave_scan(c.w * c.c0, lane, add);
        const float g_incl = ga + wave_scan(c.w * c.c1, lane, add);
        const float b_incl = ba + wave_scan(c.w * c.c2, lane, add);
        if (c.live) {
            const size_t m = (size_t)off + base + lane;
            grad_rgbs[3 * m] = g0 * c.w; grad_rgbs[3 * m + 1] = g1 * c.w; grad_rgbs[3 * m + 2] = g2 * c.w;
            grad_sigmas[m] = c.dt * (g0 * (c.T_after * c.c0 - (rf - r_incl)) + g1 * (c.T_after * c.c1 - (gf - g_incl)) +
                                     g2 * (c.T_after * c.c2 - (bf - b_incl)) + tail);
        } else if (base + lane < cnt) {
            const size_t m = (size_t)off + base + lane;
            grad_sigmas[m] = 0.0f;
            grad_rgbs[3 * m] = 0.0f; grad_rgbs[3 * m + 1] = 0.0f; grad_rgbs[3 * m + 2] = 0.0f;
        }
        T = __shfl(c.T_after, 63, 64);
        ra = __shfl(r_incl, 63, 64); ga = __shfl(g_incl, 63, 64); ba = __shfl(b_incl, 63, 64);
        if (T < T_thresh) {
            zero_rows(base + 64, cnt);
            break;
        }
    }
}

NSIG_EXPORT int rm_composite_train_mse(const float *sigmas, const float *rgbs, const float *deltas, const int32_t *rays, uint32_t M, uint32_t N, float T_thresh,
                                       const float *nears, const float *fars, const float *bg, uint32_t bg_stride, const float *gt, uint32_t n_values,
                                       float grad_scale, float *weights_sum, float *depth, float *image, float *image_out, float *depth_out, float *grad_image,
                                       float *grad_sigmas, float *grad_rgbs, nsig_stream_t stream) {
    if (N == 0) return NSIG_OK;
    NSIG_REQUIRE(sigmas && rgbs && deltas && rays && nears && fars && bg && gt && weights_sum && depth && image && image_out && depth_out && grad_image &&
                 grad_sigmas && grad_rgbs, "rm_composite_train_mse: null pointer");
    NSIG_REQUIRE(bg_stride == 0 || bg_stride == 3, "rm_composite_train_mse: bg_stride is 0 (one colour) or 3 (per ray)");
    NSIG_REQUIRE(n_values >= 1, "rm_composite_train_mse: n_values must be positive");
    const FinishArgs fin{nears, fars, bg, bg_stride, image_out, depth_out, 1u};
    k_composite_mse<<<ceil_div(N, 4u) + 64u, 256, 0, as_stream(stream)>>>(sigmas, rgbs, deltas, rays, M, N, T_thresh, gt, grad_scale * 2.0f / (float)n_values, fin,
                                                                         weights_sum, depth, image, grad_image, grad_sigmas, grad_rgbs);
    return check_launch("rm_composite_train_mse");
}

NSIG_EXPORT int rm_composite_train_bwd(const float *grad_weights_sum, const float *grad_image, const float *sigmas,
                                       const float *rgbs, const float *deltas, const int32_t *rays,
                                       const float *weights_sum, const float *image, uint32_t M, uint32_t N,
                                       float T_thresh, float *grad_sigmas, float *grad_rgbs, nsig_stream_t stream) {
    if (M == 0) return NSIG_OK;
    NSIG_REQUIRE(grad_weights_sum && grad_image && sigmas && rgbs && deltas && rays && weights_sum && image && grad_sigmas && grad_rgbs,
                 "rm_composite_train_bwd: null pointer");
    if (hipMemsetAsync(grad_sigmas, 0, (size_t)M * sizeof(float), as_stream(stream)) != hipSuccess ||
        hipMemsetAsync(grad_rgbs, 0, (size_t)M * 3 * sizeof(float), as_stream(stream)) != hipSuccess) {
        set_error("rm_composite_train_bwd: hipMemsetAsync failed");
        return NSIG_ERR_LAUNCH;
    }
    if (N == 0) return NSIG_OK;
    k_composite_bwd<<<ceil_div(N, 4u), 256, 0, as_stream(stream)>>>(grad_weights_sum, grad_image, sigmas, rgbs, deltas, rays,
                                                                        weights_sum, image, M, N, T_thresh, grad_sigmas, grad_rgbs);
    return check_launch("rm_composite_train_bwd");
}

NSIG_EXPORT int rm_march(uint32_t n_alive, uint32_t n_step, const int32_t *rays_alive, const float *rays_t,
                         const float *rays_o, const float *rays_d, float bound, float dt_gamma, uint32_t max_steps,
                         uint32_t C, uint32_t H, const uint8_t *grid, const float *nears, const float *fars, float *xyzs,
                         float *dirs, float *deltas, const float *noises, uint32_t M_rows, nsig_stream_t stream) {
    if (n_alive == 0) return NSIG_OK;
    (void)nears;  // read but unused by the reference as well (raymarching.cu:737)
    NSIG_REQUIRE(rays_alive && rays_t && rays_o && rays_d && grid && fars && xyzs && dirs && deltas, "rm_march: null pointer");
    NSIG_REQUIRE((uint64_t)n_alive * n_step <= M_rows, "rm_march: M_rows=%u < n_alive*n_step", M_rows);
    if (int e = check_grid_args("rm_march", C, H, max_steps, bound)) return e;
    if (M_rows == 0) return NSIG_OK;
    hipStream_t s = as_stream(stream);
    if (hipMemsetAsync(xyzs, 0, (size_t)M_rows * 12, s) != hipSuccess || hipMemsetAsync(dirs, 0, (size_t)M_rows * 12, s) != hipSuccess ||
        hipMemsetAsync(deltas, 0, (size_t)M_rows * 8, s) != hipSuccess) {
        set_error("rm_march: hipMemsetAsync failed");
        return NSIG_ERR_LAUNCH;
    }
    const uint32_t lanes = lanes_for_walk(n_alive);
    k_march_burst<<<ceil_div(n_alive, lanes), lanes, 0, s>>>(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d,
                                                            make_grid_view(grid, bound, dt_gamma, max_steps, C, H), fars, xyzs, dirs, deltas, noises);
    return check_launch("rm_march");
}

NSIG_EXPORT int rm_composite(uint32_t n_alive, uint32_t n_step, float T_thresh, int32_t *rays_alive, float *rays_t,
                             const float *sigmas, const float *rgbs, const float *deltas, float *weights_sum,
                             float *depth, float *image, nsig_stream_t stream) {
    if (n_alive == 0) return NSIG_OK;
    NSIG_REQUIRE(rays_alive && rays_t && sigmas && rgbs && deltas && weights_sum && depth && image, "rm_composite: null pointer");
    k_composite_burst<<<ceil_div(n_alive, 64), 64, 0, as_stream(stream)>>>(n_alive, n_step, T_thresh, rays_alive, rays_t, sigmas, rgbs,
                                                                           deltas, weights_sum, depth, image);
    return check_launch("rm_composite");
}

NSIG_EXPORT int rm_compact_alive(const int32_t *rays_alive, uint32_t n_alive, int32_t *rays_alive_out, int32_t *n_out,
                                 nsig_stream_t stream) {
    NSIG_REQUIRE(rays_alive && rays_alive_out && n_out, "rm_compact_alive: null pointer");
    NSIG_REQUIRE(rays_alive != rays_alive_out, "rm_compact_alive: in-place compaction is not supported");
    k_compact_alive<<<1, 1024, 0, as_stream(stream)>>>(rays_alive, n_alive, rays_alive_out, n_out);
    return check_launch("rm_compact_alive");
}

NSIG_EXPORT int rm_eval_begin(uint32_t N, uint32_t *ctl, int32_t *rays_alive, nsig_stream_t stream) {
    NSIG_REQUIRE(ctl && rays_alive && N >= 1, "rm_eval_begin: null pointer or no rays");
    k_eval_begin<<<ceil_div(N, 256u), 256, 0, as_stream(stream)>>>(N, ctl, rays_alive);      // all rays alive, n_step = clamp(N / N, 1, 8) = 1
    return check_launch("rm_eval_begin");
}

NSIG_EXPORT int rm_eval_march(const uint32_t *ctl, uint32_t N, const int32_t *rays_alive, const float *rays_t, const float *rays_o, const float *rays_d,
                              float bound, float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H, const uint8_t *grid, const float *fars, float *xyzs,
                              float *dirs, float *deltas, const float *noises, nsig_stream_t stream) {
    NSIG_REQUIRE(ctl && rays_alive && rays_t && rays_o && rays_d && grid && fars && xyzs && dirs && deltas && N >= 1, "rm_eval_march: null pointer");
    if (int e = check_grid_args("rm_eval_march", C, H, max_steps, bound)) return e;
    const uint32_t lanes = lanes_for_walk(N);
    k_march_burst_ctl<<<ceil_div(N, lanes), lanes, 0, as_stream(stream)>>>(ctl, rays_alive, rays_t, rays_o, rays_d, make_grid_view(grid, bound, dt_gamma, max_steps, C, H),
                                                                         fars, xyzs, dirs, deltas, noises);
    return check_launch("rm_eval_march");
}

NSIG_EXPORT int rm_eval_composite(const uint32_t *ctl, uint32_t N, float T_thresh, float density_scale, int32_t *rays_alive, float *rays_t, const float *sigmas,
                                  const float *rgbs, const float *deltas, float *weights_sum, float *depth, float *image, nsig_stream_t stream) {
    NSIG_REQUIRE(ctl && rays_alive && rays_t && sigmas && rgbs && deltas && weights_sum && depth && image && N >= 1, "rm_eval_composite: null pointer");
    k_composite_burst<<<ceil_div(N, 64u), 64, 0, as_stream(stream)>>>(0u, 0u, T_thresh, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image, ctl,
                                                                    density_scale);
    return check_launch("rm_eval_composite");
}

NSIG_EXPORT int rm_eval_compact(uint32_t *ctl, uint32_t N, uint32_t max_steps, const int32_t *rays_alive, int32_t *rays_alive_out, nsig_stream_t stream) {
    NSIG_REQUIRE(ctl && rays_alive && rays_alive_out && rays_alive != rays_alive_out && N >= 1, "rm_eval_compact: null pointer or in-place compaction");
    k_compact_alive_ctl<<<1, 1024, 0, as_stream(stream)>>>(ctl, rays_alive, rays_alive_out, N, max_steps);
    return check_launch("rm_eval_compact");
}

NSIG_EXPORT int rg_sample_rays(const float *poses, uint32_t P, const float *images, float fx, float fy, float cx, float cy, uint32_t H, uint32_t W,
                               uint32_t N, const int32_t *step_counter, uint32_t stride, uint32_t offset, uint64_t seed, float *rays_o, float *rays_d,
                               float *gt, int64_t *inds_out, int32_t *pose_out, nsig_stream_t stream) {
    if (N == 0) return NSIG_OK;
    NSIG_REQUIRE(poses && rays_o && rays_d, "rg_sample_rays: null pointer");
    NSIG_REQUIRE(P > 0 && H > 0 && W > 0 && fx != 0.0f && fy != 0.0f, "rg_sample_rays: empty pose store, bad image size or focal length");
    NSIG_REQUIRE(gt == nullptr || images != nullptr, "rg_sample_rays: ground truth requested without an image store");
    NSIG_REQUIRE((uint64_t)H * W < (1ull << 32), "rg_sample_rays: image too large");
    k_sample_rays<<<ceil_div(N, 256u), 256, 0, as_stream(stream)>>>(poses, P, images, fx, fy, cx, cy, H, W, N, step_counter, stride, offset, (uint32_t)seed,
                                                                    (uint32_t)(seed >> 32), rays_o, rays_d, gt, inds_out, pose_out);
    return check_launch("rg_sample_rays");
}

NSIG_EXPORT int rg_get_rays(const float *poses, float fx, float fy, float cx, float cy, uint32_t H, uint32_t W, const int64_t *inds,
                            uint32_t B, uint32_t N, float *rays_o, float *rays_d, nsig_stream_t stream) {
    NSIG_REQUIRE(poses && rays_o && rays_d, "rg_get_rays: null pointer");
    NSIG_REQUIRE(H > 0 && W > 0 && fx != 0.0f && fy != 0.0f, "rg_get_rays: bad image size or focal length");
    NSIG_REQUIRE(inds != nullptr || N == H * W, "rg_get_rays: without indices N must equal H*W");
    NSIG_REQUIRE((uint64_t)B * N < (1ull << 32), "rg_get_rays: too many rays");
    if (B * N == 0) return NSIG_OK;
    k_get_rays<<<ceil_div(B * N, 256), 256, 0, as_stream(stream)>>>(poses, fx, fy, cx, cy, H, W, inds, B, N, rays_o, rays_d);
    return check_launch("rg_get_rays");
}
