/*
 * oracle/raymarch_ref.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Scalar CPU restatement of the reference's native ray-marching kernels
 * (/root/reference/raymarching/src/raymarching.cu, cited per function below).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this; the product path (nerf_signature_amd/) never does.
 *
 * Pinning status: the reference file is CUDA device code and cannot be built
 * or run in this container (no nvcc, no GPU; see DESIGN.md "Oracle"), and the
 * reference ships no tests or golden vectors for it.  This restatement is
 * therefore pinned by (a) line-by-line correspondence with the cited source
 * and (b) property tests in tests/test_oracle_raymarch.py (morton round trip,
 * packbits vs numpy.packbits, composite vs a cumprod formulation, fp64
 * finite-difference check of the composite backward).
 *
 * Floating-point contract.  The reference is compiled by nvcc with its
 * default -fmad=true, i.e. a*b+c patterns are contracted to fused
 * multiply-adds.  To make integer outputs (per-ray sample counts) comparable
 * bit-for-bit between this file (x86, gcc) and the HIP kernels (gfx950), every
 * place where that contraction changes a rounding is written as an explicit
 * fmaf() here and in the kernels, and both are built with -ffp-contract=off.
 * Division and sqrt are IEEE correctly rounded on both sides.  The reference's
 * __expf (raymarching.cu:542,645,860) is a hardware approximation that cannot
 * be reproduced bit-for-bit on any other device; this file uses expf() and the
 * compositing outputs are compared within a tolerance, not bit-exactly.
 *
 * Ordering contract.  The reference reserves output ranges with two
 * atomicAdd()s (raymarching.cu:405-406), so point offsets and ray slots are in
 * nondeterministic arrival order.  This file uses ray-id order (what a
 * sequential execution of the same kernel produces): ray n occupies slot n and
 * its points start at the exclusive prefix sum of the counts of rays < n.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>
#include <float.h>

#define ORACLE_API __attribute__((visibility("default")))

static const float kSqrt3 = 1.7320508075688772f; /* raymarching.cu:19 */
static const float kInvPi = 0.3183098861837907f; /* raymarching.cu:22 */

static inline float clampf(float v, float lo, float hi) { /* raymarching.cu:34-36 */
    return fminf(hi, fmaxf(lo, v));
}

static inline float sign1(float v) { /* raymarching.cu:30-32 */
    return copysignf(1.0f, v);
}

/* raymarching.cu:42-47: cascade level from the largest |coordinate|. */
static inline int level_from_position(float x, float y, float z, float cascades) {
    float m = fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z)));
    int e;
    frexpf(m, &e);
    return (int)fminf(cascades - 1.0f, fmaxf(0.0f, (float)e));
}

/* raymarching.cu:49-54: cascade level from the step length. */
static inline int level_from_step(float dt, float H, float cascades) {
    float m = (float)((double)(dt * H) * 0.5);
    int e;
    frexpf(m, &e);
    return (int)fminf(cascades - 1.0f, fmaxf(0.0f, (float)e));
}

/* raymarching.cu:56-63 */
static inline uint32_t spread3(uint32_t v) {
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}

/* raymarching.cu:65-71 */
static inline uint32_t morton_encode(uint32_t x, uint32_t y, uint32_t z) {
    return spread3(x) | (spread3(y) << 1) | (spread3(z) << 2);
}

/* raymarching.cu:73-81 */
static inline uint32_t morton_compact(uint32_t v) {
    v &= 0x49249249u;
    v = (v | (v >> 2)) & 0xc30c30c3u;
    v = (v | (v >> 4)) & 0x0f00f00fu;
    v = (v | (v >> 8)) & 0xff0000ffu;
    v = (v | (v >> 16)) & 0x0000ffffu;
    return v;
}

/* ------------------------------------------------------------------ utils */

/* raymarching.cu:92-145 */
ORACLE_API void oracle_near_far_from_aabb(const float *rays_o, const float *rays_d, const float *aabb,
                                          uint32_t N, float min_near, float *nears, float *fars) {
    for (uint32_t n = 0; n < N; ++n) {
        const float *o = rays_o + 3 * (size_t)n, *d = rays_d + 3 * (size_t)n;
        float tmin = 0.f, tmax = 0.f;
        int miss = 0;
        for (int a = 0; a < 3 && !miss; ++a) {
            float inv = 1.0f / d[a];
            float lo = (aabb[a] - o[a]) * inv;
            float hi = (aabb[a + 3] - o[a]) * inv;
            if (lo > hi) { float s = lo; lo = hi; hi = s; }
            if (a == 0) { tmin = lo; tmax = hi; continue; }
            if (tmin > hi || lo > tmax) { miss = 1; break; }
            if (lo > tmin) tmin = lo;
            if (hi < tmax) tmax = hi;
        }
        if (miss) { nears[n] = fars[n] = FLT_MAX; continue; }
        if (tmin < min_near) tmin = min_near;
        nears[n] = tmin;
        fars[n] = tmax;
    }
}

/* raymarching.cu:163-198 */
ORACLE_API void oracle_sph_from_ray(const float *rays_o, const float *rays_d, float radius, uint32_t N,
                                    float *coords) {
    for (uint32_t n = 0; n < N; ++n) {
        const float *o = rays_o + 3 * (size_t)n, *d = rays_d + 3 * (size_t)n;
        float A = fmaf(d[2], d[2], fmaf(d[1], d[1], d[0] * d[0]));
        float B = fmaf(o[2], d[2], fmaf(o[1], d[1], o[0] * d[0]));
        float C = fmaf(o[2], o[2], fmaf(o[1], o[1], o[0] * o[0])) - radius * radius;
        float t = (-B + sqrtf(B * B - A * C)) / A;
        float x = fmaf(t, d[0], o[0]), y = fmaf(t, d[1], o[1]), z = fmaf(t, d[2], o[2]);
        float theta = atan2f(sqrtf(fmaf(z, z, x * x)), y);
        float phi = atan2f(z, x);
        coords[2 * (size_t)n + 0] = 2.0f * theta * kInvPi - 1.0f;
        coords[2 * (size_t)n + 1] = phi * kInvPi;
    }
}

/* raymarching.cu:214-226 */
ORACLE_API void oracle_morton3D(const int32_t *coords, uint32_t N, int32_t *indices) {
    for (uint32_t n = 0; n < N; ++n)
        indices[n] = (int32_t)morton_encode((uint32_t)coords[3 * (size_t)n], (uint32_t)coords[3 * (size_t)n + 1],
                                            (uint32_t)coords[3 * (size_t)n + 2]);
}

/* raymarching.cu:237-254 */
ORACLE_API void oracle_morton3D_invert(const int32_t *indices, uint32_t N, int32_t *coords) {
    for (uint32_t n = 0; n < N; ++n) {
        int32_t v = indices[n];
        coords[3 * (size_t)n + 0] = (int32_t)morton_compact((uint32_t)(v >> 0));
        coords[3 * (size_t)n + 1] = (int32_t)morton_compact((uint32_t)(v >> 1));
        coords[3 * (size_t)n + 2] = (int32_t)morton_compact((uint32_t)(v >> 2));
    }
}

/* raymarching.cu:268-289: bit i of byte n is grid[8n+i] > thresh. */
ORACLE_API void oracle_packbits(const float *grid, uint32_t n_bytes, float thresh, uint8_t *bitfield) {
    for (uint32_t n = 0; n < n_bytes; ++n) {
        uint8_t b = 0;
        for (int i = 0; i < 8; ++i)
            if (grid[8 * (size_t)n + i] > thresh) b |= (uint8_t)(1u << i);
        bitfield[n] = b;
    }
}

/* ---------------------------------------------------------------- marching */

typedef struct {
    float ox, oy, oz, dx, dy, dz, rdx, rdy, rdz;
    float bound, dt_gamma, dt_min, dt_max, rH, H3f, Hf, Cf;
    uint32_t H;
    const uint8_t *grid;
} MarchCtx;

static void march_ctx_init(MarchCtx *c, const float *o, const float *d, const uint8_t *grid, float bound,
                           float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H) {
    c->ox = o[0]; c->oy = o[1]; c->oz = o[2];
    c->dx = d[0]; c->dy = d[1]; c->dz = d[2];
    c->rdx = 1.0f / c->dx; c->rdy = 1.0f / c->dy; c->rdz = 1.0f / c->dz; /* :337 */
    c->bound = bound; c->dt_gamma = dt_gamma;
    c->rH = 1.0f / (float)H;                                            /* :338 */
    c->H3f = (float)(H * H * H);                                        /* :339 */
    c->Hf = (float)H; c->Cf = (float)C; c->H = H;
    c->dt_min = (2.0f * kSqrt3) / (float)max_steps;                     /* :345 */
    c->dt_max = ((2.0f * kSqrt3) * (float)(1u << (C - 1))) / (float)H;  /* :346 */
    c->grid = grid;
}

/* One probe of the occupancy grid at parameter t (raymarching.cu:360-379 and the
 * identical blocks at :428-448, :751-770).  Returns occupancy; outputs the clamped
 * position, the step length, and the exit distance to use if the cell is empty. */
static inline int probe(const MarchCtx *c, float t, float *px, float *py, float *pz, float *pdt, float *t_exit) {
    const float x = clampf(fmaf(t, c->dx, c->ox), -c->bound, c->bound);
    const float y = clampf(fmaf(t, c->dy, c->oy), -c->bound, c->bound);
    const float z = clampf(fmaf(t, c->dz, c->oz), -c->bound, c->bound);
    const float dt = clampf(t * c->dt_gamma, c->dt_min, c->dt_max);

    int lp = level_from_position(x, y, z, c->Cf);
    int ls = level_from_step(dt, c->Hf, c->Cf);
    const int level = lp > ls ? lp : ls;

    const float mip_bound = fminf(scalbnf(1.0f, level), c->bound);
    const float mip_rbound = 1.0f / mip_bound;
    const float top = (float)(c->H - 1);

    /* :374-376 -- the 0.5 literal is a double, the bracket is float. */
    const int nx = (int)clampf((float)(0.5 * (double)fmaf(x, mip_rbound, 1.0f) * (double)c->H), 0.0f, top);
    const int ny = (int)clampf((float)(0.5 * (double)fmaf(y, mip_rbound, 1.0f) * (double)c->H), 0.0f, top);
    const int nz = (int)clampf((float)(0.5 * (double)fmaf(z, mip_rbound, 1.0f) * (double)c->H), 0.0f, top);

    /* :378 -- float arithmetic, then conversion to uint32. */
    const uint32_t index = (uint32_t)((float)level * c->H3f + (float)morton_encode((uint32_t)nx, (uint32_t)ny, (uint32_t)nz));
    const int occ = (c->grid[index >> 3] & (1u << (index & 7u))) != 0;

    *px = x; *py = y; *pz = z; *pdt = dt;
    if (!occ) {
        /* :390-394 */
        const float fx = (fmaf(0.5f, sign1(c->dx), (float)nx + 0.5f) * c->rH) * 2.0f - 1.0f;
        const float fy = (fmaf(0.5f, sign1(c->dy), (float)ny + 0.5f) * c->rH) * 2.0f - 1.0f;
        const float fz = (fmaf(0.5f, sign1(c->dz), (float)nz + 0.5f) * c->rH) * 2.0f - 1.0f;
        const float tx = fmaf(fx, mip_bound, -x) * c->rdx;
        const float ty = fmaf(fy, mip_bound, -y) * c->rdy;
        const float tz = fmaf(fz, mip_bound, -z) * c->rdz;
        *t_exit = t + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
    }
    return occ;
}

static inline float skip_empty(const MarchCtx *c, float t, float t_exit) { /* :396-398 */
    do {
        t += clampf(t * c->dt_gamma, c->dt_min, c->dt_max);
    } while (t < t_exit);
    return t;
}

/*
 * raymarching.cu:312-480 (training march), in ray-id order.
 *   rays[n] = (n, offset_n, count_n); counter[0] = total points, counter[1] = N.
 * xyzs/dirs/deltas hold M rows and must be zero-filled by the caller
 * (raymarching.py:205-207).  A ray whose range would exceed M is skipped after
 * its `rays` record is written (:416).  Passing xyzs == NULL runs the count pass
 * only.
 */
ORACLE_API void oracle_march_rays_train(const float *rays_o, const float *rays_d, const uint8_t *grid, float bound,
                                        float dt_gamma, uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H,
                                        uint32_t M, const float *nears, const float *fars, float *xyzs, float *dirs,
                                        float *deltas, int32_t *rays, int32_t *counter, const float *noises) {
    for (uint32_t n = 0; n < N; ++n) {
        MarchCtx c;
        march_ctx_init(&c, rays_o + 3 * (size_t)n, rays_d + 3 * (size_t)n, grid, bound, dt_gamma, max_steps, C, H);
        const float near = nears[n], far = fars[n], noise = noises[n];

        float t0 = near;
        t0 = fmaf(clampf(t0 * dt_gamma, c.dt_min, c.dt_max), noise, t0); /* :351 */

        /* pass 1 (:354-400) */
        float t = t0;
        uint32_t count = 0;
        float x, y, z, dt, t_exit = 0.f;
        while (t < far && count < max_steps) {
            if (probe(&c, t, &x, &y, &z, &dt, &t_exit)) { ++count; t += dt; }
            else t = skip_empty(&c, t, t_exit);
        }

        /* :405-413 with sequential "atomics" */
        const uint32_t offset = (uint32_t)counter[0];
        const uint32_t slot = (uint32_t)counter[1];
        counter[0] += (int32_t)count;
        counter[1] += 1;
        rays[3 * (size_t)slot + 0] = (int32_t)n;
        rays[3 * (size_t)slot + 1] = (int32_t)offset;
        rays[3 * (size_t)slot + 2] = (int32_t)count;

        if (count == 0 || xyzs == NULL) continue;
        if (offset + count > M) continue; /* :416 */

        /* pass 2 (:422-479) */
        float *px = xyzs + 3 * (size_t)offset, *pd = dirs + 3 * (size_t)offset, *pl = deltas + 2 * (size_t)offset;
        t = t0;
        float last_t = t;
        uint32_t step = 0;
        while (t < far && step < count) {
            if (probe(&c, t, &x, &y, &z, &dt, &t_exit)) {
                px[0] = x; px[1] = y; px[2] = z;
                pd[0] = c.dx; pd[1] = c.dy; pd[2] = c.dz;
                t += dt;
                pl[0] = dt;
                pl[1] = t - last_t;
                last_t = t;
                px += 3; pd += 3; pl += 2; ++step;
            } else t = skip_empty(&c, t, t_exit);
        }
    }
}

/* raymarching.cu:501-577 */
ORACLE_API void oracle_composite_rays_train_forward(const float *sigmas, const float *rgbs, const float *deltas,
                                                    const int32_t *rays, uint32_t M, uint32_t N, float T_thresh,
                                                    float *weights_sum, float *depth, float *image) {
    for (uint32_t n = 0; n < N; ++n) {
        const uint32_t id = (uint32_t)rays[3 * (size_t)n], off = (uint32_t)rays[3 * (size_t)n + 1],
                       cnt = (uint32_t)rays[3 * (size_t)n + 2];
        float r = 0, g = 0, b = 0, ws = 0, tt = 0, d = 0, T = 1.0f;
        if (!(cnt == 0 || off + cnt > M)) {
            for (uint32_t s = 0; s < cnt; ++s) {
                const size_t m = (size_t)off + s;
                const float alpha = 1.0f - expf(-sigmas[m] * deltas[2 * m]);
                const float w = alpha * T;
                r = fmaf(w, rgbs[3 * m + 0], r);
                g = fmaf(w, rgbs[3 * m + 1], g);
                b = fmaf(w, rgbs[3 * m + 2], b);
                tt += deltas[2 * m + 1];
                d = fmaf(w, tt, d);
                ws += w;
                T *= 1.0f - alpha;
                if (T < T_thresh) break; /* :557 -- after accumulating this sample */
            }
        }
        weights_sum[id] = ws; depth[id] = d;
        image[3 * (size_t)id] = r; image[3 * (size_t)id + 1] = g; image[3 * (size_t)id + 2] = b;
    }
}

/* raymarching.cu:602-682; grad_sigmas/grad_rgbs are pre-zeroed by the caller (raymarching.py:283-284). */
ORACLE_API void oracle_composite_rays_train_backward(const float *grad_weights_sum, const float *grad_image,
                                                     const float *sigmas, const float *rgbs, const float *deltas,
                                                     const int32_t *rays, const float *weights_sum, const float *image,
                                                     uint32_t M, uint32_t N, float T_thresh, float *grad_sigmas,
                                                     float *grad_rgbs) {
    for (uint32_t n = 0; n < N; ++n) {
        const uint32_t id = (uint32_t)rays[3 * (size_t)n], off = (uint32_t)rays[3 * (size_t)n + 1],
                       cnt = (uint32_t)rays[3 * (size_t)n + 2];
        if (cnt == 0 || off + cnt > M) continue;
        const float gws = grad_weights_sum[id];
        const float *gi = grad_image + 3 * (size_t)id;
        const float rf = image[3 * (size_t)id], gf = image[3 * (size_t)id + 1], bf = image[3 * (size_t)id + 2];
        const float wsf = weights_sum[id];
        float r = 0, g = 0, b = 0, ws = 0, T = 1.0f;
        for (uint32_t s = 0; s < cnt; ++s) {
            const size_t m = (size_t)off + s;
            const float alpha = 1.0f - expf(-sigmas[m] * deltas[2 * m]);
            const float w = alpha * T;
            r = fmaf(w, rgbs[3 * m + 0], r);
            g = fmaf(w, rgbs[3 * m + 1], g);
            b = fmaf(w, rgbs[3 * m + 2], b);
            ws += w;
            T *= 1.0f - alpha;
            grad_rgbs[3 * m + 0] = gi[0] * w;
            grad_rgbs[3 * m + 1] = gi[1] * w;
            grad_rgbs[3 * m + 2] = gi[2] * w;
            grad_sigmas[m] = deltas[2 * m] * (gi[0] * (T * rgbs[3 * m + 0] - (rf - r)) +
                                              gi[1] * (T * rgbs[3 * m + 1] - (gf - g)) +
                                              gi[2] * (T * rgbs[3 * m + 2] - (bf - b)) + gws * (1.0f - wsf));
            if (T < T_thresh) break;
        }
    }
}

/* --------------------------------------------------------------- inference */

/* raymarching.cu:701-805; outputs hold n_alive*n_step (padded) rows, pre-zeroed (raymarching.py:334-336). */
ORACLE_API void oracle_march_rays(uint32_t n_alive, uint32_t n_step, const int32_t *rays_alive, const float *rays_t,
                                  const float *rays_o, const float *rays_d, float bound, float dt_gamma,
                                  uint32_t max_steps, uint32_t C, uint32_t H, const uint8_t *grid, const float *nears,
                                  const float *fars, float *xyzs, float *dirs, float *deltas, const float *noises) {
    for (uint32_t n = 0; n < n_alive; ++n) {
        const int32_t id = rays_alive[n];
        MarchCtx c;
        march_ctx_init(&c, rays_o + 3 * (size_t)id, rays_d + 3 * (size_t)id, grid, bound, dt_gamma, max_steps, C, H);
        (void)nears; /* read but unused by the reference too (:737) */
        const float far = fars[id];
        float t = rays_t[id];
        t = fmaf(clampf(t * dt_gamma, c.dt_min, c.dt_max), noises[n], t); /* :746 */
        float last_t = t;
        float *px = xyzs + 3 * (size_t)n * n_step, *pd = dirs + 3 * (size_t)n * n_step,
              *pl = deltas + 2 * (size_t)n * n_step;
        uint32_t step = 0;
        float x, y, z, dt, t_exit = 0.f;
        while (t < far && step < n_step) {
            if (probe(&c, t, &x, &y, &z, &dt, &t_exit)) {
                px[0] = x; px[1] = y; px[2] = z;
                pd[0] = c.dx; pd[1] = c.dy; pd[2] = c.dz;
                t += dt;
                pl[0] = dt;
                pl[1] = t - last_t;
                last_t = t;
                px += 3; pd += 3; pl += 2; ++step;
            } else t = skip_empty(&c, t, t_exit);
        }
    }
}

/* raymarching.cu:819-905 (in-place accumulation; T = 1 - weight_sum, :868). */
ORACLE_API void oracle_composite_rays(uint32_t n_alive, uint32_t n_step, float T_thresh, int32_t *rays_alive,
                                      float *rays_t, const float *sigmas, const float *rgbs, const float *deltas,
                                      float *weights_sum, float *depth, float *image) {
    for (uint32_t n = 0; n < n_alive; ++n) {
        const int32_t id = rays_alive[n];
        const float *ps = sigmas + (size_t)n * n_step, *pc = rgbs + 3 * (size_t)n * n_step,
                    *pl = deltas + 2 * (size_t)n * n_step;
        float t = rays_t[id], wsum = weights_sum[id], d = depth[id];
        float r = image[3 * (size_t)id], g = image[3 * (size_t)id + 1], b = image[3 * (size_t)id + 2];
        uint32_t step = 0;
        while (step < n_step) {
            if (pl[0] == 0.0f) break; /* :858 */
            const float alpha = 1.0f - expf(-ps[0] * pl[0]);
            const float T = 1.0f - wsum;
            const float w = alpha * T;
            wsum += w;
            t += pl[1];
            d = fmaf(w, t, d);
            r = fmaf(w, pc[0], r);
            g = fmaf(w, pc[1], g);
            b = fmaf(w, pc[2], b);
            if (T < T_thresh) break; /* :882 */
            ++ps; pc += 3; pl += 2; ++step;
        }
        if (step < n_step) rays_alive[n] = -1; /* :894-898 */
        else rays_t[id] = t;
        weights_sum[id] = wsum; depth[id] = d;
        image[3 * (size_t)id] = r; image[3 * (size_t)id + 1] = g; image[3 * (size_t)id + 2] = b;
    }
}
