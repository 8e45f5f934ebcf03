/*
 * nerfsig.h -- C ABI of libnerfsig.so, the MI355X (gfx950) native layer of the watermarked-NeRF
 * render path.
 *
 * This is the drop-in boundary for the reference's native surface
 *   /root/reference/raymarching/src/raymarching.h:7-17  (prototypes)
 *   /root/reference/raymarching/src/bindings.cpp:5-18   (pybind module `_raymarching`)
 * plus the encoder / MLP evaluations the reference performs through PyTorch op chains
 * (hash_encoding.py, hash_encoding_wtmk_bit.py) and tiny-cuda-nn (nerf/network_wtmk_tcnn.py:52-88).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name ends in `_host`;
 *   - the caller owns every buffer; nothing is allocated or freed inside; no call synchronises;
 *   - all kernels are enqueued on `stream` (a hipStream_t passed as void*; NULL = default stream);
 *   - arrays are dense row-major fp32 / int32 / uint8 exactly as the reference's tensors;
 *   - return value 0 = enqueued; non-zero = rejected or launch failure, message in nsig_last_error();
 *   - calls on distinct streams are thread-safe.
 *
 * Ordering contract of the training march (differs from the reference by being deterministic):
 * ray n owns slot n of `rays`, and its points start at the exclusive prefix sum of the counts of
 * rays < n; the reference hands out both with atomicAdd (raymarching.cu:405-406) in arrival order.
 */
#ifndef NERFSIG_H_
#define NERFSIG_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NSIG_OK 0
#define NSIG_ERR_ARG 1
#define NSIG_ERR_LAUNCH 2

#define NSIG_TABLE_ROWS (1u << 19) /* log2_hashmap_size = 19, hash_encoding.py:49 */
#define NSIG_BASE_LEVELS 16
#define NSIG_MAX_MESSAGE_DIM 64

typedef void *nsig_stream_t;

int nsig_abi_version(void);
const char *nsig_last_error(void);

/* ------------------------------------------------------------------ raymarching: utilities */

/* replaces near_far_from_aabb (raymarching.h:7, raymarching.cu:92-156) */
int rm_near_far_from_aabb(const float *rays_o, const float *rays_d, const float *aabb, uint32_t N, float min_near,
                          float *nears, float *fars, nsig_stream_t stream);
/* replaces sph_from_ray (raymarching.h:8, raymarching.cu:163-209) */
int rm_sph_from_ray(const float *rays_o, const float *rays_d, float radius, uint32_t N, float *coords,
                    nsig_stream_t stream);
/* replaces morton3D / morton3D_invert (raymarching.h:9-10, raymarching.cu:214-260) */
int rm_morton3D(const int32_t *coords, uint32_t N, int32_t *indices, nsig_stream_t stream);
int rm_morton3D_invert(const int32_t *indices, uint32_t N, int32_t *coords, nsig_stream_t stream);
/* replaces packbits (raymarching.h:11, raymarching.cu:268-300); n_bytes = C*H^3/8 */
int rm_packbits(const float *grid, uint32_t n_bytes, float density_thresh, uint8_t *bitfield, nsig_stream_t stream);

/*
 * The density-grid refresh on the device: NeRFRenderer.update_extra_state (renderer_wtmk.py:445-538; the trainer calls it every 16 steps, nerf/utils.py:852-858)
 * without its host reads (`nonzero`, two `.item()`), so that a captured training loop replays it as a graph (nerf_signature_amd/gridrefresh.py).  Per cascade:
 *   rg_refresh_begin   fresh[0, n_cells) = -1 (:454)
 *   rg_refresh_draw    the partial form's 2N cells (:488-500): draws [0,N) uniform over the grid, [N,2N) uniform with repetition over the occupied cells of the cascade
 *                      (`grid_cas > 0`, morton order; found through a prefix sum of the flags); out: keys[2N] = (z*H + y)*H + x grouped by grid row (z, y) --
 *                      the order the density query wants -- and ids[2N], the draw each came from.  scratch: rg_refresh_draw_scratch_bytes(N, H) bytes
 *   rg_refresh_points  probe point and morton cell index of every key (keys / ids NULL: the full form, key = id = i, n = H^3): cell centre (:474,480-482) + jitter (:484)
 *   (the density query is hg_encode_planes + field_fwd)
 *   rg_refresh_scatter fresh[cell] = max(fresh[cell], sigma * density_scale), fresh pre-filled with -1 (:489,:514; repeated cells: the largest candidate, deterministically)
 * and once per refresh
 *   rg_refresh_finish  grid = max(grid * decay, fresh) where both >= 0 (:521-522), *mean_density = mean(clamp(grid, 0)) (:523), bitfield packed at
 *                      min(*mean_density, density_thresh) (:528-529), *iter_dev += 1, and for window > 0 *mean_count = int(sum of the ring's last `window` sample totals /
 *                      window) (:533-536; count_ring [16][2] int32, rows (*step_dev - window + i) % 16).  partials: rg_refresh_partials_bytes(n_cells) bytes of scratch.
 * Random numbers are a pure function of (seed, *iter_dev, cascade, draw): a replay draws fresh values and two runs draw the same.
 */
size_t rg_refresh_draw_scratch_bytes(uint32_t N, uint32_t H);
int rg_refresh_begin(float *fresh, uint32_t n_cells, nsig_stream_t stream);
int rg_refresh_draw(int32_t *keys, int32_t *ids, uint32_t N, uint32_t H, const float *grid_cas, void *scratch, uint64_t seed, const int32_t *iter_dev,
                    uint32_t cas, nsig_stream_t stream);
int rg_refresh_points(const int32_t *keys, const int32_t *ids, uint32_t n, uint32_t H, float extent, float half_cell, uint64_t seed, const int32_t *iter_dev,
                      uint32_t cas, float *xyz, int32_t *cells, nsig_stream_t stream);
int rg_refresh_scatter(const float *sigmas, const int32_t *cells, uint32_t n, float density_scale, float *fresh, nsig_stream_t stream);
size_t rg_refresh_partials_bytes(uint32_t n_cells);
int rg_refresh_finish(float *grid, const float *fresh, uint32_t n_cells, float decay, void *partials, float density_thresh, uint8_t *bitfield,
                      float *mean_density, int32_t *iter_dev, const int32_t *count_ring, uint32_t window, const uint32_t *step_dev, int32_t *mean_count,
                      nsig_stream_t stream);

/* ------------------------------------------------------------------ raymarching: training */

/*
 * march_rays_train (raymarching.h:13, raymarching.cu:312-490) as three enqueues:
 *   count : per-ray sample count + the parameter t of every sample (t_rec[n*max_steps + s])
 *   scan  : rays[n] = (n, exclusive-prefix offset, count); counter[0] = total, counter[1] = N
 *   write : xyzs/dirs/deltas rows [0,total) from t_rec; rows [total, M) are zero-filled
 * A ray whose range would exceed M is recorded in `rays` but writes nothing (raymarching.cu:416).
 * t_rec needs rm_march_train_scratch_bytes(N, max_steps) bytes; it need not be initialised.
 */
size_t rm_march_train_scratch_bytes(uint32_t N, uint32_t max_steps);
int rm_march_train_count(const float *rays_o, const float *rays_d, const uint8_t *grid, float bound, float dt_gamma,
                         uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, const float *nears,
                         const float *fars, const float *noises, int32_t *counts, float *t_rec,
                         nsig_stream_t stream);
int rm_march_train_scan(const int32_t *counts, uint32_t N, int32_t *rays, int32_t *counter, nsig_stream_t stream);
/* the same table for MANY rays (staged full-image renders march up to 262 144 rays per launch sequence) in two launches of
 * 4096-ray workgroups; block_sums: caller-owned scratch of rm_march_train_scan_blocks(N) int32 words, need not be initialised */
int rm_march_train_scan_blocks(uint32_t N);
int rm_march_train_scan_wide(const int32_t *counts, uint32_t N, int32_t *rays, int32_t *counter, int32_t *block_sums,
                             nsig_stream_t stream);
int rm_march_train_write(const float *rays_o, const float *rays_d, float bound, float dt_gamma, uint32_t max_steps,
                         uint32_t N, uint32_t C, uint32_t H, uint32_t M, const float *nears, const float *noises,
                         const float *t_rec, const int32_t *rays, const int32_t *counter, float *xyzs, float *dirs,
                         float *deltas, nsig_stream_t stream);

/*
 * The same march in TWO enqueues (what a captured training step runs):
 *   rm_march_train_count_nf  = near_far_from_aabb (raymarching.h:7) + count: the walk computes each ray's limits itself (same
 *                              arithmetic as rm_near_far_from_aabb, bit for bit) and stores them to nears / fars for the kernels behind it;
 *   rm_march_train_scan_write = scan + write: every workgroup scans the N counts for itself (N + 1 offsets in LDS), so no
 *                              single-workgroup launch sits between the walk and the writes; N <= rm_march_train_scan_write_max_rays().
 * Results (rays, counter, every row, the zero padding) are identical to the three-enqueue form.
 */
int rm_march_train_count_nf(const float *rays_o, const float *rays_d, const float *aabb, float min_near, const uint8_t *grid,
                            float bound, float dt_gamma, uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H,
                            const float *noises, float *nears, float *fars, int32_t *counts, float *t_rec, nsig_stream_t stream);
int rm_march_train_scan_write_max_rays(void);
int rm_march_train_scan_write(const float *rays_o, const float *rays_d, float bound, float dt_gamma, uint32_t max_steps, uint32_t N,
                              uint32_t C, uint32_t H, uint32_t M, const float *nears, const float *noises, const float *t_rec,
                              const int32_t *counts, int32_t *rays, int32_t *counter, float *xyzs, float *dirs, float *deltas,
                              nsig_stream_t stream);

/* replaces composite_rays_train_forward (raymarching.h:14, raymarching.cu:501-588) */
int rm_composite_train_fwd(const float *sigmas, const float *rgbs, const float *deltas, const int32_t *rays,
                           uint32_t M, uint32_t N, float T_thresh, float *weights_sum, float *depth, float *image,
                           nsig_stream_t stream);
/* replaces composite_rays_train_backward (raymarching.h:15, raymarching.cu:602-693).
 * grad_sigmas / grad_rgbs are fully written (rows the reference leaves at their pre-zeroed value are
 * written as zero), so the caller does not have to clear them (cf. raymarching.py:283-284). */
int rm_composite_train_bwd(const float *grad_weights_sum, const float *grad_image, const float *sigmas,
                           const float *rgbs, const float *deltas, const int32_t *rays, const float *weights_sum,
                           const float *image, uint32_t M, uint32_t N, float T_thresh, float *grad_sigmas,
                           float *grad_rgbs, nsig_stream_t stream);

/* ------------------------------------------------------------------ raymarching: inference */

/* replaces march_rays (raymarching.h:17, raymarching.cu:701-815); all M_rows rows of the outputs are
 * written (unused slots as zero), so they need not be pre-zeroed (cf. raymarching.py:334-336). */
int rm_march(uint32_t n_alive, uint32_t n_step, const int32_t *rays_alive, const float *rays_t, const float *rays_o,
             const float *rays_d, float bound, float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H,
             const uint8_t *grid, const float *nears, const float *fars, float *xyzs, float *dirs, float *deltas,
             const float *noises, uint32_t M_rows, nsig_stream_t stream);
/* replaces composite_rays (raymarching.h:18, raymarching.cu:819-914) */
int rm_composite(uint32_t n_alive, uint32_t n_step, float T_thresh, int32_t *rays_alive, float *rays_t,
                 const float *sigmas, const float *rgbs, const float *deltas, float *weights_sum, float *depth,
                 float *image, nsig_stream_t stream);
/* replaces the host-side `rays_alive[rays_alive >= 0]` (nerf/renderer_wtmk.py:363): stable compaction of
 * the non-negative entries; *n_out (device) receives the survivor count. */
int rm_compact_alive(const int32_t *rays_alive, uint32_t n_alive, int32_t *rays_alive_out, int32_t *n_out,
                     nsig_stream_t stream);

/*
 * The same eval loop (renderer_wtmk.py:335-372) with its CONTROL on the device: no per-round read-back of the survivor count.
 *   ctl  device uint32[4] = {n_alive, n_step, rows = n_alive * n_step, samples marched so far}
 * rm_eval_begin fills rays_alive with 0..N-1 and ctl with {N, 1, N, 0}.  A round is rm_eval_march -> the field pass over `rows` rows
 * (hg_encode_planes_rows / field_fwd_rows: launches sized for the capacity N, the row count read from ctl + 2) -> rm_eval_composite ->
 * rm_eval_compact, which compacts the alive list into rays_alive_out and writes the NEXT round's ctl: n_alive = survivors (0 once max_steps
 * samples have been marched), n_step = clamp(N / n_alive, 1, 8).  Every launch is sized for the worst case (n_alive * n_step <= N) and
 * early-outs on the device counts, so the host may enqueue rounds blindly and read ctl[0] back once every few rounds, only to stop.
 * Buffers: xyzs / dirs [N, 3], deltas [N, 2], sigmas [N], rgbs [N, 3] (no 128-row padding: rows beyond `rows` are never read); rm_eval_march writes
 * zero rows for a ray that ends inside its burst.  density_scale multiplies sigma inside the compositor (:353).  Same bursts, alive lists and images as
 * rm_march / rm_composite / rm_compact_alive driven from the host.
 */
int rm_eval_begin(uint32_t N, uint32_t *ctl, int32_t *rays_alive, nsig_stream_t stream);
int rm_eval_march(const uint32_t *ctl, uint32_t N, const int32_t *rays_alive, const float *rays_t, const float *rays_o, const float *rays_d,
                  float bound, float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H, const uint8_t *density_bitfield, const float *fars,
                  float *xyzs, float *dirs, float *deltas, const float *noises, nsig_stream_t stream);
int rm_eval_composite(const uint32_t *ctl, uint32_t N, float T_thresh, float density_scale, int32_t *rays_alive, float *rays_t,
                      const float *sigmas, const float *rgbs, const float *deltas, float *weights_sum, float *depth, float *image,
                      nsig_stream_t stream);
int rm_eval_compact(uint32_t *ctl, uint32_t N, uint32_t max_steps, const int32_t *rays_alive, int32_t *rays_alive_out, nsig_stream_t stream);

/* ------------------------------------------------------------------ ray generation */

/* The arithmetic of get_rays (nerf/utils_wtmk_disen.py:59-143): pinhole rays through pixel centres, normalised,
 * rotated by the camera-to-world pose.  poses [B,4,4]; inds [B,N] int64 pixel indices (row-major, i = ind % W,
 * j = ind / W) or NULL for all H*W pixels in order (then N must be H*W); outputs rays_o, rays_d [B,N,3]. */
int rg_get_rays(const float *poses, float fx, float fy, float cx, float cy, uint32_t H, uint32_t W, const int64_t *inds,
                uint32_t B, uint32_t N, float *rays_o, float *rays_d, nsig_stream_t stream);

/* A whole training batch from a device-resident store (the loader step in front of the path: nerf/provider_wtmk.py:585-600 hands one pose per
 * step to get_rays(..., N), nerf/utils_wtmk_disen.py:103-106 draws N pixel indices with torch.randint on the device).  poses [P,4,4],
 * images [P,H*W,3] or NULL; pose p = (step * stride + offset) mod P where step = *step_counter (a DEVICE int32, NULL = 0; stride/offset =
 * world size / rank shard the pose sequence); N indices uniform in [0,H*W) from a counter-based hash of (seed, step, ray): same distribution as
 * the reference's draw, a different generator.  Writes rays_o, rays_d [N,3], gt [N,3] (images[p][ind]; NULL to skip), optionally the drawn
 * indices (int64 [N]) and the pose number (int32 [1]).  No host value enters the launch: it can sit inside a captured step. */
int rg_sample_rays(const float *poses, uint32_t P, const float *images, float fx, float fy, float cx, float cy, uint32_t H, uint32_t W,
                   uint32_t N, const int32_t *step_counter, uint32_t stride, uint32_t offset, uint64_t seed, float *rays_o, float *rays_d,
                   float *gt, int64_t *inds_out, int32_t *pose_out, nsig_stream_t stream);

/* ------------------------------------------------------------------ hash grids */

/* S[t] = sum_i tables[i][t] over the D selected codebook tables (the tables 2i+bit_i of
 * hash_encoding_wtmk_bit.py:110).  Interpolation is linear and every codebook level shares one
 * resolution and hash (network_wtmk_tcnn.py:43-44), so interpolating S equals summing the D
 * interpolations (hash_encoding_wtmk_bit.py:116).  tables_host: host array of D device pointers. */
int hg_codebook_presum(const float *const *tables_host, uint32_t D, float *S, nsig_stream_t stream);

/* Same sum with the selection made on the device: all 2D tables are passed (table 2i and 2i+1 for bit i) and
 * message [D] (device floats, 0. or 1.) picks one of each pair.  Every launch argument is then independent of the
 * message, which is what lets a whole training step be captured once in a hipGraph and replayed with new messages. */
int hg_codebook_presum_sel(const float *const *all_tables_host, const float *message, uint32_t D, float *S,
                           nsig_stream_t stream);

/* HashEmbedder.forward (hash_encoding.py:96-111) [+ codebook added into channels 30:32,
 * network_wtmk_tcnn.py:106, when S != NULL].  x01 in [0,1]; feat is [M,32]. */
int hg_encode_fwd(const float *x01, uint32_t M, const float *const *base_tables_host, const float *S, float *feat,
                  nsig_stream_t stream);

/* HashEmbedder(msg).forward evaluated literally: D separate gathers, summed (hash_encoding_wtmk_bit.py:99-116).
 * out is [M,2]. */
int hg_codebook_encode_fwd(const float *x01, uint32_t M, const float *const *tables_host, uint32_t D, float *out,
                           nsig_stream_t stream);

/* Backward of the codebook lookup w.r.t. the tables: G[row] += w_corner * dfeat for the 8 corners of every
 * point.  All D selected tables receive this same gradient.  G is [T,2] and is accumulated into. */
int hg_codebook_bwd(const float *x01, uint32_t M, const float *dfeat, float *G, nsig_stream_t stream);

/* Same accumulation as hg_codebook_bwd, organised for the memory system: `rec` is the [M, 8]-dword record field_bwd emits
 * per point: { ix | iy << 16, iz (cell of the 2048^3 codebook grid), wx, wy, wz (interpolation weights, fp32),
 * d feature[30], d feature[31], 0 }.  256 workgroups = 32 slices of G x 8
 * replicas; a workgroup keeps its 16384-row slice (128 KiB) in LDS, scans one eighth of the points, accumulates the
 * corners that hash into its slice with LDS atomics and finally adds the slice to G with contiguous global atomics
 * (1/10 of the scattered atomics' requests).  G is accumulated into. */
int hg_scatter_sliced(const float *rec, uint32_t M, float *G, nsig_stream_t stream);

/* The same accumulation with the redundancy removed: the (point, corner-pair) hits are first grouped by slice with an exact
 * two-pass counting sort (self-contained 16-byte entries in `scratch`: hg_scatter_binned_scratch_bytes(M) bytes, 16-byte
 * aligned; since round 6 it also holds the owners' slabs), then every slice owner streams only its own entries into 64-bit fixed-point accumulators; a slice's four
 * replica owners leave them as slabs and a last launch adds the slabs as integers, converts once and adds ONE float per element to G: the sums are bit-reproducible
 * (the determinism of the reference's embedding_dense_backward), and G still accumulates over calls.  ~3x less time than hg_scatter_sliced on a million points.
 * G (here and for hg_scatter_planned, hg_scatter_levels, hg_levels_scatter: every table) must be 16-byte aligned: an owner that is alone on its
 * slice stores its rows as 16-byte vectors. */
size_t hg_scatter_binned_scratch_bytes(uint32_t M);
int hg_scatter_binned(const float *rec, uint32_t M, float *G, void *scratch, nsig_stream_t stream);

/* The binned scatter with its sort planned ahead: where a (point, corner pair) entry goes in the slice-sorted queue depends on
 * the point's position only, so hg_scatter_plan computes counts, offsets and per-point destinations from xyzs [M,3] as soon as
 * the samples exist (any stream, typically beside the forward pass); field_bwd_planned then writes the gradients straight into
 * the queue and hg_scatter_planned runs the slice owners.  `plan`: hg_scatter_plan_bytes(M) bytes, 16-byte aligned, must stay
 * untouched between the three calls.  Same sums as hg_scatter_binned (fixed-point per slice: order-independent).  G is
 * accumulated into.  Replaces the same reference step as hg_codebook_bwd (autograd of hash_encoding.py:73-94). */
size_t hg_scatter_plan_bytes(uint32_t M);
int hg_scatter_plan(const float *xyzs, uint32_t M, float bound, void *plan, nsig_stream_t stream);
int hg_scatter_planned(const void *plan, uint32_t M, float *G, nsig_stream_t stream);

/* grads[i][t] (+)= G[t] for the D selected tables; grads_host: host array of D device pointers. */
int hg_fanout_grad(const float *G, float *const *grads_host, uint32_t D, int accumulate, nsig_stream_t stream);

/* Test/diagnostic: hashed rows [M,8] (int32) and interpolation weights [M,3] of one level. */
int hg_level_lookup(const float *x01, uint32_t M, float resolution, int32_t *rows, float *weights,
                    nsig_stream_t stream);

/* ------------------------------------------------------------------ optimiser */

/*
 * Adam step (torch.optim.Adam semantics, no weight decay / amsgrad; main_nerf_wtmk.py:110) for the D selected
 * codebook tables, which all carry the same gradient G [T,2]: one pass reads G once and updates param, exp_avg,
 * exp_avg_sq of every table.  step_size_host[i] = lr / (1 - beta1^step_i), inv_bc2_sqrt_host[i] =
 * 1 / sqrt(1 - beta2^step_i) (per table: a table's step count advances only when it is selected).
 * grad_scale multiplies G first (1/world_size, or a loss-scale reciprocal).
 */
int opt_codebook_adam(const float *G, float *const *params_host, float *const *exp_avg_host, float *const *exp_avg_sq_host,
                      uint32_t D, float beta1, float beta2, float eps, const float *step_size_host,
                      const float *inv_bc2_sqrt_host, float grad_scale, nsig_stream_t stream);

/* opt_codebook_adam with the selection, the per-table step counts and the learning rate on the device (graph
 * replay): params/exp_avg/exp_avg_sq/steps are host arrays of 2D device pointers (steps[j]: one fp32 scalar per
 * table, torch's capturable-Adam state format), message [D] picks table 2i+bit_i, *lr is read on the device.
 * scratch: 2*D floats.  Selected tables' step counts are incremented, the others are left untouched. */
int opt_codebook_adam_sel(const float *G, float *const *params_host, float *const *exp_avg_host, float *const *exp_avg_sq_host,
                          float *const *steps_host, const float *message, uint32_t D, const float *lr, float beta1,
                          float beta2, float eps, float grad_scale, float *scratch, nsig_stream_t stream);

/* opt_codebook_adam_sel that also leaves, in S_next [2^19,2], the pre-summed codebook of the NEXT step's message
 * (next_message [D] on the device): sum_i table[2i + next_i] over the tables as updated by this very step, bit-identical to
 * hg_codebook_presum_sel run afterwards -- the next step then starts without its 128 MiB pre-sum pass (the reference
 * re-gathers every selected table per point, hash_encoding_wtmk_bit.py:99-116; see hg_codebook_presum). */
int opt_codebook_adam_sel_next(const float *G, float *const *params_host, float *const *exp_avg_host, float *const *exp_avg_sq_host,
                               float *const *steps_host, const float *message, uint32_t D, const float *lr, float beta1,
                               float beta2, float eps, float grad_scale, float *scratch, const float *next_message,
                               float *S_next, nsig_stream_t stream);

/* ------------------------------------------------------------------ field network */

/* Re-lays the two flat tcnn-style parameter vectors (sigma: 3072, color: 7168 fp32, layout in
 * INTEGRATION.md) into MFMA operand order, as split-bf16 (hi, lo) AND as fp16 fragments.  packed needs mlp_packed_bytes() bytes. */
size_t mlp_packed_bytes(void);
int mlp_pack_weights(const float *sigma_params, const float *color_params, void *packed, nsig_stream_t stream);

/* Arithmetic of the MLP kernels (field_fwd / field_color_fwd / field_bwd / field_bwd_planned), process-wide:
 *   0 = split-bf16: three v_mfma_f32_32x32x16_bf16 per product (hi*hi + hi*lo + lo*hi), fp32 accumulate, ~2^-16 per product;
 *   1 = fp16: one v_mfma_f32_32x32x16_f16 per product, fp32 accumulate (tinycudann's FullyFusedMLP, which the reference calls at
 *       nerf/network_wtmk_tcnn.py:52-88, computes in fp16 with fp16 accumulate); the backward normalises every point's upstream
 *       gradient by a power of two, so loss-scaled gradients neither overflow nor underflow.
 * Default: environment variable NERFSIG_MLP ("bf16x3" | "f16"), else 1.  The stage-1 trace entry points always use mode 0. */
int mlp_get_precision(void);
int mlp_set_precision(int mode);
/* Which launches evaluate the MLPs of the TRAINING render at fp16 precision: bit 0 set = the forward (all 17 planes in, sigma + rgb + masks out)
 * runs software-pipelined over a wave's tiles (next tile's inputs requested and the previous tile's results stored at a tile's head, every layer's
 * weight fragments fetched from LDS in one burst); bit 1 set = the planned backward likewise.  Cleared bits select the plain per-tile loops.
 * Bit 0 also selects the stage-1 forward's launch (field_fwd_trace(_rows): inputs of the next tile requested in front of a tile's ~100 trace stores).
 * Results are bit-identical either way (tests/test_gpu_field.py, tests/test_gpu_stage1.py); default 3.
 * No counterpart in the reference (tinycudann's fully fused MLP is one fixed kernel). */
int mlp_get_pipelined(void);
int mlp_set_pipelined(int mask);

#define FIELD_MASK_WORDS 6 /* uint32 words of ReLU masks per point saved by field_fwd for field_bwd */

/*
 * NeRFNetwork.forward (nerf/network_wtmk_tcnn.py:97-124) for M points, fused:
 *   x01 = (x+bound)/(2 bound); 16-level hash encode (+ codebook via S, may be NULL = clean model);
 *   sigma MLP 32->64->16, sigma = exp(h0); SH degree 4 of d; color MLP 32->64->64->3, sigmoid.
 * Outputs: sigmas [M]; rgbs [M,3] (NULL: skip the color branch = NeRFNetwork.density);
 *          geo_feat [M,15] (optional); masks [M_pad32, 6] uint32 (optional, needed by field_bwd).
 * planes: NULL = one fused kernel (features gathered inside the MLP kernel); otherwise the level-major feature planes
 *   written by hg_encode_planes for the same (xyzs, M, bound, tables, S): the two-kernel route (XCD-partitioned
 *   encoder, then the MLP kernel) is the faster one for large batches.  Both give bit-identical results.
 * planes_layout: the layout `planes` was written in -- NSIG_PLANES_F32 (hg_encode_planes / _rows) or NSIG_PLANES_MIXED (hg_encode_planes_mixed); the owner of
 *   the buffer says so at every call (ignored when planes is NULL).
 */
#define NSIG_PLANES_F32 0
#define NSIG_PLANES_MIXED 1
int field_fwd(const float *xyzs, const float *dirs, uint32_t M, float bound, const float *const *base_tables_host,
              const float *S, const void *packed, float *sigmas, float *rgbs, float *geo_feat, uint32_t *masks,
              const void *planes, int planes_layout, nsig_stream_t stream);

/* The encoder of field_fwd as its own pass: planes[level][point] (float2; 16 base levels + the pre-summed codebook as
 * plane 16), hg_planes_bytes(M) bytes.  Workgroup (tile, slot = blockIdx % 8) encodes only the levels assigned to
 * its slot, so each XCD's L2 serves the gathers of one fine table (see csrc/field.hip). */
size_t hg_planes_bytes(uint32_t M);
int hg_encode_planes(const float *xyzs, uint32_t M, float bound, const float *const *base_tables_host, const float *S,
                     void *planes, nsig_stream_t stream);
/* hg_encode_planes / field_fwd (inference outputs: sigmas + rgbs) over a row count that lives on the DEVICE (*rows_dev <= M_capacity): the
 * launch, the plane stride and every buffer are sized for M_capacity, rows beyond the count are skipped (the eval loop above). */
int hg_encode_planes_rows(const float *xyzs, uint32_t M_capacity, const uint32_t *rows_dev, float bound,
                          const float *const *base_tables_host, const float *S, void *planes, nsig_stream_t stream);
/* The plane set in the MIXED layout, for the fp16 MLP only (mlp_set_precision(1)): levels 0..14 as fp16 pairs -- exactly the words of the MLP's first-layer
 * operand, rounded to nearest even here instead of at the MLP's load: bit-identical results -- followed by level 15 and the codebook level as float2 (the codebook
 * is added to level 15 before the rounding, network_wtmk_tcnn.py:106).  76 instead of 136 bytes per point each way between encoder and MLP; same buffer size
 * (hg_planes_bytes).  Whoever owns the buffer passes planes_layout = NSIG_PLANES_MIXED to the entry points that read or complete the set (field_fwd / field_fwd_rows /
 * hg_encode_codebook_plane); the split-bf16 MLP refuses it, field_fwd_trace (stage 1) takes fp32 sets only.  rows_dev may be NULL. */
int hg_encode_planes_mixed(const float *xyzs, uint32_t M_capacity, const uint32_t *rows_dev, float bound, const float *const *base_tables_host,
                           const float *S, void *planes, nsig_stream_t stream);
int field_fwd_rows(const float *xyzs, const float *dirs, uint32_t M_capacity, const uint32_t *rows_dev, float bound,
                   const float *const *base_tables_host, const float *S, const void *packed, float *sigmas, float *rgbs, const void *planes,
                   int planes_layout, nsig_stream_t stream);

/* Reads the 16 base tables (and S when given) once, each through the XCD whose workgroups will gather from it in hg_encode_planes, so that
 * the launch finds its tables in L2 instead of starting on caches a streaming pass (the optimiser's) has flushed: the bench workload's block
 * launch 282 -> 258 us; the pass itself takes ~20 us alone -- it pays where it can run beside something else (trainer.GraphedWatermarkLoop).
 * Replaces nothing in the reference (a scheduling aid); sink: one writable float, never written. */
int hg_warm_tables(const float *const *base_tables_host, const float *S, float *sink, nsig_stream_t stream);

/* Rays that do not change between training steps (the watermark blocks: nerf/provider_wtmk.py:442-494 computes
 * rays_o_block / rays_d_block once per dataset and hands the same tensors to every train_step, utils_wtmk_disen.py:588-590;
 * the occupancy grid, the base tables and both MLPs are frozen in the watermark stage, network_wtmk_tcnn.py:90-95) keep their
 * samples and the 16 base-level planes (hg_encode_planes with S = NULL, once); per step only the codebook level -- the one
 * thing a step changes -- is gathered again, into plane 16 of the same plane set.  Same interpolation code, bit-identical planes.
 * plan_to_reset: NULL, or a scatter plan (hg_scatter_plan) kept across steps for the same points: its per-launch largest-gradient
 * word is cleared here, ahead of the step's field_bwd_planned. */
int hg_encode_codebook_plane(const float *xyzs, uint32_t M, float bound, const float *S, void *planes, int planes_layout, void *plan_to_reset,
                             nsig_stream_t stream);

/* NeRFNetwork.color (nerf/network_wtmk_tcnn.py:147-176) without the mask: rgb from dirs + geo_feat. */
int field_color_fwd(const float *dirs, const float *geo_feat, uint32_t M, const void *packed, float *rgbs,
                    nsig_stream_t stream);

/*
 * Backward of field_fwd w.r.t. the codebook: (dL/dsigma, dL/drgb) -> MLP input gradients (weights are
 * frozen, nerf/network_wtmk_tcnn.py:90-95) -> d feature[30:32] -> scatter into G [T,2] (accumulated).
 * Outputs (each optional, at least one required): G -- direct scatter with global atomics; dfeat_out [M,2] --
 * d feature[30:32]; rec_out [M,8] dwords (32-byte aligned) -- the scatter record consumed by hg_scatter_sliced (the fast route).
 */
int field_bwd(const float *xyzs, uint32_t M, float bound, const float *grad_sigmas, const float *grad_rgbs,
              const float *sigmas, const float *rgbs, const uint32_t *masks, const void *packed, float *G,
              float *dfeat_out, float *rec_out, nsig_stream_t stream);
/* The same backward with the planned scatter's queue as its only output (see hg_scatter_plan). */
int field_bwd_planned(const float *xyzs, uint32_t M, float bound, const float *grad_sigmas, const float *grad_rgbs,
                      const float *sigmas, const float *rgbs, const uint32_t *masks, const void *packed, void *plan,
                      nsig_stream_t stream);

/*
 * torch.optim.Adam (betas, eps; no weight decay / amsgrad) in its capturable form over n dense fp32 tensors in one pass:
 * steps[i] (device scalar) is incremented, bias corrections use the new value, lr is a device scalar.  Replaces the generic
 * multi-tensor kernel for the decoder's parameters (main_nerf_wtmk.py:110).  grad_scale multiplies every gradient first
 * (1/world after a sum all-reduce).  scratch: 2 * 32 * ceil(n/32) floats.
 */
int opt_adam_dense(uint32_t n, float *const *params_host, const float *const *grads_host, float *const *exp_avg_host,
                   float *const *exp_avg_sq_host, float *const *steps_host, const uint32_t *numel_host, const float *lr, float beta1,
                   float beta2, float eps, float grad_scale, float *scratch, nsig_stream_t stream);
/* opt_adam_dense for torch.optim.Adam's NON-capturable state format (step counts in host tensors): the caller passes each tensor's
 * lr / (1 - beta1^t) and 1 / sqrt(1 - beta2^t) as host arrays; one launch per 32 tensors (the drop-in model's optimiser hook). */
int opt_adam_dense_host(uint32_t n, float *const *params_host, const float *const *grads_host, float *const *exp_avg_host,
                        float *const *exp_avg_sq_host, const uint32_t *numel_host, const float *step_sizes_host, const float *inv_bc2_host,
                        float beta1, float beta2, float eps, float grad_scale, nsig_stream_t stream);

/*
 * rm_composite_train_fwd followed by rm_finish_fwd in the compositing launch (the ray's wave writes the background-mixed image
 * and the normalised depth next to the raw weights_sum / depth / image, which the backward still needs), and
 * rm_composite_train_bwd with rm_finish_bwd's weights_sum term folded in: grad_image_out is the gradient of image_out,
 * grad_weights_sum (optional) any further gradient of weights_sum; the depth gradient is not propagated (raymarching.py:275).
 * rays_in_order != 0: the caller guarantees what rm_march_train_scan produces -- rays in ascending, gapless offset order
 * (offset[n+1] = offset[n] + count[n]) -- and the kernel then zero-fills every gradient row it does not write itself
 * (terminated samples, rays that did not fit, the padding past the last ray) instead of two memsets over all M rows.
 */
int rm_composite_train_finish_fwd(const float *sigmas, const float *rgbs, const float *deltas, const int32_t *rays, uint32_t M, uint32_t N,
                                  float T_thresh, const float *nears, const float *fars, const float *bg, uint32_t bg_stride,
                                  float *weights_sum, float *depth, float *image, float *image_out, float *depth_out, nsig_stream_t stream);
int rm_composite_train_finish_bwd(const float *grad_weights_sum, const float *grad_image_out, const float *sigmas, const float *rgbs,
                                  const float *deltas, const int32_t *rays, const float *weights_sum, const float *image, const float *bg,
                                  uint32_t bg_stride, uint32_t M, uint32_t N, float T_thresh, uint32_t rays_in_order, float *grad_sigmas,
                                  float *grad_rgbs, nsig_stream_t stream);

/* ------------------------------------------------------------------ elementwise tails, one kernel per direction */

/*
 * renderer_wtmk.py:316-319 (and 369-372): image_out = image + (1 - weights_sum) * bg_color,
 * depth_out = clamp(depth - near, min=0) / (far - near).  bg: one colour [3] (bg_stride 0) or per ray [N,3] (bg_stride 3).
 * Backward: grad_weights_sum = -sum_c grad_image*bg, grad_depth_in = grad_depth/(far-near) where depth >= near
 * (grad_image passes through unchanged; grad_image / grad_depth / grad_depth_in may be NULL).
 */
int rm_finish_fwd(const float *image, const float *depth, const float *weights_sum, const float *nears, const float *fars, const float *bg,
                  uint32_t bg_stride, uint32_t N, float *image_out, float *depth_out, nsig_stream_t stream);
int rm_finish_bwd(const float *grad_image, const float *grad_depth, const float *depth, const float *nears, const float *fars, const float *bg,
                  uint32_t bg_stride, uint32_t N, float *grad_weights_sum, float *grad_depth_in, nsig_stream_t stream);

/*
 * Opening launch of a captured training step (the loop body of utils_wtmk_disen.py:1164-1181 replayed as a hipGraph): zero-fills
 * G [n_floats] (optimizer.zero_grad for the shared codebook gradient) and copies `width` floats of slot (*counter % slots) of a
 * ring in PINNED host memory (ring_dev = its device-side address, [slots][width]: this step's message and the next one's, :1165) into msg [width], then
 * advances *counter (device uint32).  The host fills slot k % slots before launching replay k and must not overwrite a slot
 * before the replay that reads it has finished.
 */
int loop_step_begin(float *G, uint32_t n_floats, const float *ring_dev, uint32_t slots, uint32_t width, uint32_t *counter, float *msg,
                    nsig_stream_t stream);
/* The device-side address of pinned (hipHostMalloc / torch pin_memory) host memory, or NULL: ring_dev above is
 * nsig_host_device_pointer(ring_host), resolved once, outside any stream capture. */
void *nsig_host_device_pointer(const void *pinned_host);

/*
 * The losses of Trainer.train_step (utils_wtmk_disen.py:615-640, loss_w = 'bce'):
 *   losses3 = { mean((content - gt)^2),  mean BCEWithLogits(temp * decoded, message),  lambda_w * loss_w + lambda_i * loss_i }
 * over n_content floats and D logits.  d_content / d_decoded receive d(loss_i)/d(content), d(loss_w)/d(decoded); wm_loss_bwd
 * scales them by the incoming gradients of the three outputs (device scalars, NULL = 0).
 */
int wm_loss_fwd(const float *content, const float *gt, uint32_t n_content, const float *decoded, const float *message, uint32_t D, float temp,
                float lambda_w, float lambda_i, float *losses3, float *d_content, float *d_decoded, nsig_stream_t stream);
int wm_loss_bwd(const float *g_lossi, const float *g_lossw, const float *g_loss, float lambda_w, float lambda_i, const float *d_content,
                uint32_t n_content, const float *d_decoded, uint32_t D, float *grad_content, float *grad_decoded, nsig_stream_t stream);

/* ------------------------------------------------------------------ decoder */

/*
 * BatchNorm2d with batch statistics (track_running_stats=False, hidden_models.py:26) followed by GELU (erf form), as one
 * kernel each way for the HiDDeN decoder's ConvBNRelu blocks (hidden_models.py:16-35).  x, y, grads: [N, C, H, W] in
 * channels-last memory (element (n,c,p) at (n*P + p)*C + c, P = H*W).  save: 2*C floats (mean, inverse std).
 *   forward : y = gelu(gamma * (x - mean_c) / sqrt(var_c + eps) + beta), var biased over N*P
 *   backward: dx, dgamma, dbeta from dy (the gradient of y); gelu'(.) is recomputed from x.
 * One workgroup per channel (the tensors are ~1 MB: the cost is launch count, which this replaces 5:1 and 6:1).
 */
int dec_bn_gelu_fwd(const float *x, const float *gamma, const float *beta, uint32_t N, uint32_t C, uint32_t P, float eps, float *y,
                    float *save, nsig_stream_t stream);
int dec_bn_gelu_bwd(const float *dy, const float *x, const float *gamma, const float *beta, const float *save, uint32_t N, uint32_t C,
                    uint32_t P, float *dx, float *dgamma, float *dbeta, nsig_stream_t stream);

/*
 * The whole HiDDeN decoder of hidden_models.py:104-137 -- ConvBNRelu(Cin,64), 7 x ConvBNRelu(64,64), ConvBNRelu(64,1),
 * AdaptiveAvgPool2d(1), Linear(1,1) -- as one chain of fused kernels per direction (one launch per layer each way: MFMA
 * implicit-GEMM convolutions with the BatchNorm statistics in their epilogue and BatchNorm+GELU in the consumer's prologue).
 * Replaces `msg_decoder(img)` and its autograd backward (network_wtmk_tcnn.py:46, utils_wtmk_disen.py:499-505).
 *   img      input_mode 0: [B, Cin, H, W] fp32 (NCHW, already normalised), mean/std ignored (may be NULL);
 *            input_mode 1: the rendered blocks [B, H, W, Cin] as the compositor left them: torch.clamp(0,1), the permute and
 *            normalize_img(mean, std) of utils_wtmk_disen.py:599-601 are applied on load (host arrays of Cin <= 8 floats), their
 *            backward in dec_backward; clamped_out (optional, [B, H, W, Cin]) receives the clamped blocks (the step's pred_rgb).
 *   decoded  [B] (the Linear output; num_bits = redundancy = 1)
 *   params   host array of 29 device pointers: for l = 0..8 {conv weight [Cout,Cin_l,3,3], bn weight, bn bias}, then
 *            linear weight [1,1], linear bias [1].  conv biases are not inputs: BatchNorm's batch-mean subtraction cancels them.
 *   grads    host array of 29 device pointers, same order and shapes (written, not accumulated); grad_img has img's layout
 *   workspace: dec_workspace_bytes(B, Cin, H, W) bytes (0 = shape not supported: H*W <= 1024, Cin <= 32, LDS limits);
 *            dec_backward reads what dec_forward left there, so the pair must share it and nothing may overwrite it between.
 *   weights_stream: pass `stream` again for one in-order launch sequence.  A different stream receives the parameter-gradient
 *            kernels (ordered after the data-gradient chain by an event), so that grad_img's consumers on `stream` need not wait
 *            for them; the caller then joins weights_stream before reading `grads`.
 */
size_t dec_workspace_bytes(uint32_t B, uint32_t Cin, uint32_t H, uint32_t W);
int dec_forward(const float *img, uint32_t input_mode, const float *mean_host, const float *std_host, const float *const *params_host,
                uint32_t B, uint32_t Cin, uint32_t H, uint32_t W, float eps, void *workspace, float *decoded, float *clamped_out,
                nsig_stream_t stream);
int dec_backward(const float *grad_decoded, const float *img, uint32_t input_mode, const float *mean_host, const float *std_host,
                 const float *const *params_host, uint32_t B, uint32_t Cin, uint32_t H, uint32_t W, void *workspace,
                 float *const *grads_host, float *grad_img, nsig_stream_t stream, nsig_stream_t weights_stream);

/*
 * The distortion layer of the training step (Trainer.distortion_layer, nerf/utils_wtmk_disen.py:551-577; applied to the clamped block
 * renders in front of the decoder, :594; `--distortion`, main_nerf_wtmk.py:75).  distortion: 0 none, 1 noise (x + n, the reference
 * draws n ~ N(0, 0.1) per element), 2 brightness (torchvision ColorJitter(brightness=0.5): clamp(f * x, 0, 1), one f in [0.5, 1.5]
 * per call), 3 blurring (torchvision GaussianBlur(3, sigma in [0.01, 0.5]): taps exp(-0.5 (d / sigma)^2) normalised, reflect padding).
 * 4 rotation (torchvision RandomRotation((-30, 30)) per image: nearest-neighbour resampling about the centre, zero fill) and 5 scaling (per image
 * [3, H, W]: F.interpolate(scale_factor = sf in [0.75, 1.25], mode = 'linear') -- 1-d, along W only, W_out = floor(W * sf)) change the sampling geometry:
 * kernels of their own in front of / behind the decoder (wm_distort_geom_fwd / _bwd), whose output the decoder reads as an ordinary rendered image.
 *   dist_param  device float[1]: f (2) or sigma (3), read on the device -- a captured step draws it itself (wm_distort_draw);
 *   dist_noise  device [B, H, W, Cin] (1).
 * dec_forward_train / dec_backward_train = dec_forward / dec_backward with input_mode 1 (the training step's call) and the layer applied on load,
 * between the clamp and the normalisation (no extra launch; the blur's adjoint is one small launch behind the image-gradient epilogue and needs
 * grad_scratch [B, H, W, Cin]); clamped_out still receives the UNdistorted clamped blocks (the step's pred_rgb, :592).
 * dec_forward_train can also leave the watermark loss's gradient behind (bce_seed [B], optional): the step's loss is
 * lambda_w * mean_i BCE-with-logits(temp * decoded_i, message_i) (loss_w 'bce', utils_wtmk_disen.py:441,641-644), whose derivative with respect to
 * decoded_i is element-wise -- bce_seed[i] = bce_scale * (sigmoid(bce_temp * decoded_i) - bce_message[i]), bce_scale = lambda_w * temp / B from the caller
 * -- so dec_backward_train(grad_decoded = bce_seed) starts right behind the forward chain, with no loss kernel on the path between them (the loss VALUES
 * are still wm_loss_fwd's, computed off that path).
 * wm_distort_fwd / _bwd: the layer alone on [B, H, W, C] (img = the unclamped render; out = D(clamp(img)); grad_img through the clamp).
 * wm_distort_draw: counter-based draws, a pure function of (seed, *step_counter, element): param_out[0] ~ U[0.5, 1.5] (2) /
 * U[0.01, 0.5] (3), noise_out[0 .. n_noise) ~ N(0, 0.1) (1); (4): n_noise = the number of images, param_out[2 i], [2 i + 1] = (cos, sin) of image i's angle
 * ~ U[-30, 30) degrees.  step_counter may be NULL (step 0).  The scaling factor decides a tensor shape and is drawn by the host.
 * wm_distort_geom_fwd: out [B, H, W_out, C] = the layer (4: dist_param = (cos, sin) per image, W_out = W; 5: dist_param[0] = sf, W_out = floor(W * sf),
 * computed by the caller) on clamp(img [B, H, W, C]); clamped_out (optional) = clamp(img).  _bwd: grad_img [B, H, W, C] = clamp mask x the adjoint of
 * the resampling applied to grad_out [B, H, W_out, C] (gather form, deterministic).
 */
int dec_forward_train(const float *img, const float *mean_host, const float *std_host, const float *const *params_host, uint32_t B,
                      uint32_t Cin, uint32_t H, uint32_t W, float eps, void *workspace, float *decoded, float *clamped_out,
                      uint32_t distortion, const float *dist_param, const float *dist_noise, const float *bce_message, float bce_temp,
                      float bce_scale, float *bce_seed, nsig_stream_t stream);
int dec_backward_train(const float *grad_decoded, const float *img, const float *mean_host, const float *std_host,
                       const float *const *params_host, uint32_t B, uint32_t Cin, uint32_t H, uint32_t W, void *workspace,
                       float *const *grads_host, float *grad_img, uint32_t distortion, const float *dist_param,
                       const float *dist_noise, float *grad_scratch, nsig_stream_t stream, nsig_stream_t weights_stream);
int wm_distort_draw(uint32_t distortion, uint64_t seed, const uint32_t *step_counter, uint32_t n_noise, float *param_out,
                    float *noise_out, nsig_stream_t stream);
int wm_distort_fwd(const float *img, uint32_t B, uint32_t H, uint32_t W, uint32_t C, uint32_t distortion, const float *dist_param,
                   const float *dist_noise, float *out, nsig_stream_t stream);
int wm_distort_bwd(const float *grad_out, const float *img, uint32_t B, uint32_t H, uint32_t W, uint32_t C, uint32_t distortion,
                   const float *dist_param, const float *dist_noise, float *grad_img, nsig_stream_t stream);
int wm_distort_geom_fwd(const float *img, uint32_t B, uint32_t H, uint32_t W, uint32_t C, uint32_t distortion, const float *dist_param,
                        uint32_t W_out, float *out, float *clamped_out, nsig_stream_t stream);
int wm_distort_geom_bwd(const float *grad_out, const float *img, uint32_t B, uint32_t H, uint32_t W, uint32_t C, uint32_t distortion,
                        const float *dist_param, uint32_t W_out, float *grad_img, nsig_stream_t stream);

/* ------------------------------------------------------------------ stage-1 (clean model) training, SURVEY.md 8(f) N3 */

/*
 * The clean model of stage 1 (nerf/network_hash.py) trains everything: base tables and both MLPs (:154-166).
 * field_fwd_trace = field_fwd on pre-encoded planes (no codebook) that also saves each layer's input, feature-major
 * [width][stride] fp32 with stride = M rounded up to 32: act_hs [64], act_cin [32] (16 SH, 15 geometry features, the
 * padded 1.0), act_h1 [64], act_h2 [64]; M <= 2^26 for the trace entry points.  field_bwd_trace back-propagates (dL/dsigma, dL/drgb) and writes every
 * layer's pre-activation gradient (d_hs [64], d_so [16], d_h1 [64], d_h2 [64], d_out [16]; same layout) and the gradient
 * of all 32 encoder features as level-major planes d_planes [16][stride] float2.  The weight gradients are then plain
 * GEMMs over the point dimension (d_pre x act^T, left to the BLAS library); hg_scatter_level scatters one level's
 * feature gradient into that level's table gradient G_l [T,2] (owner-computes, as hg_scatter_sliced).
 */
int field_fwd_trace(const float *xyzs, const float *dirs, uint32_t M, float bound, const float *const *base_tables_host,
                    const void *packed, const void *planes, float *sigmas, float *rgbs, uint32_t *masks, float *act_hs,
                    float *act_cin, float *act_h1, float *act_h2, nsig_stream_t stream);
int field_bwd_trace(uint32_t M, const float *grad_sigmas, const float *grad_rgbs, const float *sigmas, const float *rgbs,
                    const uint32_t *masks, const void *packed, float *d_hs, float *d_so, float *d_h1, float *d_h2, float *d_out,
                    void *d_planes, nsig_stream_t stream);
int hg_scatter_level(const float *xyzs, float bound, const void *d_plane, uint32_t M, uint32_t level, float *G,
                     nsig_stream_t stream);
/* All 16 levels at once: d_planes is field_bwd_trace's [16][stride] float2 output, G_host 16 device tables [T,2] that are
 * WRITTEN (not accumulated into).  The (point, level) cell indices and weights are computed once into `scratch`
 * (hg_scatter_levels_scratch_bytes(M) bytes, 16-byte aligned: sixteen [M,8] records in hg_scatter_sliced's format plus the
 * binning scratch), then the binned fixed-point scatter of hg_scatter_binned runs over the 16 record sets at once; every table
 * row has a single owner workgroup that stores it (no atomics, no zero-fill, bit-reproducible). */
size_t hg_scatter_levels_scratch_bytes(uint32_t M);
int hg_scatter_levels(const float *xyzs, float bound, const void *d_planes, uint32_t M, uint32_t stride, float *const *G_host,
                      void *scratch, nsig_stream_t stream);


/*
 * The captured stage-1 step (nerf_signature_amd/stage1.py GraphedCleanLoop; loop body of nerf/utils.py:852-869).
 *
 * *_rows: the point count is a DEVICE value (the march's counter): launches are sized for the buffers' capacity M_capacity, rows past
 * min(M_capacity, *rows_dev) are neither read nor written; every buffer keeps the layout of M_capacity points (stride = M_capacity
 * rounded up to 32).  field_wgrad: the five weight-gradient reductions sum_p d_pre[o][p] * act[i][p] on MFMA (split bf16, fp32
 * accumulate; K = points), from field_fwd_trace's layer inputs + the encoder planes and field_bwd_trace's pre-activation gradients;
 * writes all of grad_sigma_params [3072] = [W1s 64x32 | W2s 16x64] and grad_color_params [7168] = [Wc1 64x32 | Wc2 64x64 | Wc3 16x64]
 * (tcnn's layout: INTEGRATION.md section 3); scratch = field_wgrad_scratch_bytes(M) bytes, 16-byte aligned; bit-reproducible (partial sums
 * per workgroup, added in workgroup order).  Replaces the library GEMMs of the note above (5 x 145 us per 125 k points).
 * clean_loss: loss[0] = mean((image - gt)^2) over n_values = N * 3 elements (nerf/utils.py:503, `.mean(-1)` then `.mean()`),
 * grad_image = grad_scale * d loss / d image.  Optional bookkeeping of a captured loop, by the same workgroup (any pointer may be
 * NULL): count_ring [16][2] row (*step_dev % 16) = march_counter[0..1] (the ring NeRFRenderer.step_counter is, renderer_wtmk.py:282-284),
 * loss_ring[*step_dev % loss_ring_len] = loss, noise_next[0..n_noise) = the next step's per-ray march offsets, U[0,1) as a pure function
 * of (seed, *step_dev + 1, ray) (the reference draws them with torch.rand, raymarching.py:213), then *step_dev += 1.
 */
int field_fwd_trace_rows(const float *xyzs, const float *dirs, uint32_t M_capacity, const uint32_t *rows_dev, float bound,
                         const float *const *base_tables_host, const void *packed, const void *planes, float *sigmas, float *rgbs,
                         uint32_t *masks, float *act_hs, float *act_cin, float *act_h1, float *act_h2, nsig_stream_t stream);
int field_bwd_trace_rows(uint32_t M_capacity, const uint32_t *rows_dev, const float *grad_sigmas, const float *grad_rgbs, const float *sigmas,
                         const float *rgbs, const uint32_t *masks, const void *packed, float *d_hs, float *d_so, float *d_h1, float *d_h2,
                         float *d_out, void *d_planes, nsig_stream_t stream);
size_t field_wgrad_scratch_bytes(uint32_t M);
int field_wgrad(uint32_t M, const uint32_t *rows_dev, const void *planes, const float *act_hs, const float *act_cin, const float *act_h1,
                const float *act_h2, const float *d_hs, const float *d_so, const float *d_h1, const float *d_h2, const float *d_out,
                void *scratch, float *grad_sigma_params, float *grad_color_params, nsig_stream_t stream);
/*
 * field_bwd_trace(_rows) + field_wgrad in ONE launch (csrc/stage1_fused.hip): the same backward chain (d_planes: the same bits), but every layer's
 * pre-activation gradient stays on the chip -- transposed through wave-private LDS into MFMA operands over the point dimension -- and the five weight
 * gradients accumulate in registers over a wave's tiles; only the layer INPUTS (field_fwd_trace's act_*, the encoder planes) are read from memory.
 * Removes 896 B per point written and read back.  Same outputs as the pair it replaces: d_planes [16][stride] float2, grad_sigma_params [3072],
 * grad_color_params [7168] (written, tcnn's layout); weight gradients equal to field_wgrad's up to the order of the partial sums, bit-reproducible
 * (fixed tile -> wave assignment, slabs added in workgroup order).  rows_dev may be NULL (= M points); 1 <= M <= 2^26; scratch = field_bwd_wgrad_scratch_bytes(M)
 * bytes; packed, planes, act_*, d_planes and scratch 16-byte aligned.  Split-bf16 arithmetic like field_bwd_trace (mlp_set_precision does not apply).
 */
size_t field_bwd_wgrad_scratch_bytes(uint32_t M);
int field_bwd_wgrad(uint32_t M, const uint32_t *rows_dev, const float *grad_sigmas, const float *grad_rgbs, const float *sigmas, const float *rgbs,
                    const uint32_t *masks, const void *packed, const void *planes, const float *act_hs, const float *act_cin, const float *act_h1,
                    const float *act_h2, void *d_planes, void *scratch, float *grad_sigma_params, float *grad_color_params, nsig_stream_t stream);
/*
 * The same pair with the saved layer inputs kept as fp16: act_hs / act_h1 / act_h2 [64][stride] and act_cin [32][stride] of _Float16 -- 448 instead of 896
 * bytes per point written by the forward and read back by the backward.  The forward's OUTPUTS (sigmas, rgbs, masks) are field_fwd_trace's bit for bit: only
 * what is saved is rounded (to the precision the reference's MLPs keep their activations in, tinycudann FullyFusedMLP, network_hash.py:39-49,65-75); the
 * backward widens the rows when it stages them and multiplies in the same split-bf16 arithmetic (an fp16 value splits into bf16 hi + lo exactly).
 * rows_dev may be NULL (= M_capacity points).
 */
int field_fwd_trace_f16(const float *xyzs, const float *dirs, uint32_t M_capacity, const uint32_t *rows_dev, float bound,
                        const float *const *base_tables_host, const void *packed, const void *planes, float *sigmas, float *rgbs,
                        uint32_t *masks, void *act_hs, void *act_cin, void *act_h1, void *act_h2, nsig_stream_t stream);
int field_bwd_wgrad_f16(uint32_t M, const uint32_t *rows_dev, const float *grad_sigmas, const float *grad_rgbs, const float *sigmas, const float *rgbs,
                        const uint32_t *masks, const void *packed, const void *planes, const void *act_hs, const void *act_cin, const void *act_h1,
                        const void *act_h2, void *d_planes, void *scratch, float *grad_sigma_params, float *grad_color_params, nsig_stream_t stream);
/* Stage 1's compositing in one launch: rm_composite_train_finish_fwd + the gradient half of clean_loss (grad_image = grad_scale * 2 / n_values *
 * (image_out - gt), nerf/utils.py:503) + rm_composite_train_finish_bwd (rays in ascending gapless offset order: every gradient row written or zeroed
 * by the kernel).  One wave per ray in all three, so a ray's image stays in registers; the same bits in every output as the three launches.  The loss value
 * and a captured loop's books remain clean_loss's job (launched behind it). */
int rm_composite_train_mse(const float *sigmas, const float *rgbs, const float *deltas, const int32_t *rays, uint32_t M, uint32_t N, float T_thresh,
                           const float *nears, const float *fars, const float *bg, uint32_t bg_stride, const float *gt, uint32_t n_values,
                           float grad_scale, float *weights_sum, float *depth, float *image, float *image_out, float *depth_out, float *grad_image,
                           float *grad_sigmas, float *grad_rgbs, nsig_stream_t stream);
/*
 * The stage-1 trainer's exponential moving average of the parameters (main_nerf.py:130 ema_decay=0.95; utils.py:389-390,761-762: torch_ema's
 * ExponentialMovingAverage.update() after every optimiser step): for every tensor shadow -= (shadow - param) * (1 - d), d = min(decay, (1 + u) / (10 + u)),
 * u = *num_updates -- the count of updates INCLUDING this one (a captured loop's device step counter, already advanced by the step's loss kernel).  One launch,
 * n <= 32 tensors.
 */
int opt_ema_update(uint32_t n, const float *const *params_host, float *const *shadow_host, const uint32_t *numel_host, const uint32_t *num_updates,
                   double decay, nsig_stream_t stream);
int clean_loss(const float *image, const float *gt, uint32_t n_values, float grad_scale, float *loss, float *grad_image,
               uint32_t *step_dev, const int32_t *march_counter, int32_t *count_ring, float *loss_ring, uint32_t loss_ring_len,
               float *noise_next, uint32_t n_noise, uint64_t seed, nsig_stream_t stream);
/*
 * hg_scatter_levels in two halves.  hg_levels_plan needs the sample positions only -- which slice owner every (point, level, (dy,dz)
 * pair) entry goes to, and where in that owner's queue -- so a step runs it beside its forward pass; hg_levels_scatter, behind the MLP
 * backward, turns d_planes into queue entries at the planned places and runs the 16 x 64 owners, which WRITE every row of the 16
 * tables G_host (no zero-fill, no atomics, fixed-point accumulation: bit-reproducible).  plan = hg_levels_plan_bytes(M) bytes,
 * 16-byte aligned (16 headers, 16 queues of 4 M entries, 16 x M destinations); rows_dev may be NULL (= M points).
 */
size_t hg_levels_plan_bytes(uint32_t M);
int hg_levels_plan(const float *xyzs, uint32_t M, const uint32_t *rows_dev, float bound, void *plan, nsig_stream_t stream);
int hg_levels_scatter(const float *xyzs, uint32_t M, const uint32_t *rows_dev, float bound, const void *d_planes, uint32_t stride,
                      void *plan, float *const *G_host, nsig_stream_t stream);
/*
 * hg_levels_scatter with the optimiser step of the 16 tables inside the owners (one process: no gradient exchange between the two): an owner that has
 * summed its slice's rows applies torch.optim.Adam's update to them -- opt_adam_dense's arithmetic, element by element, state in the same capturable format
 * (device step counts, advanced here; scratch: 64 floats) -- instead of storing them, so the 64 MiB of table gradients are neither written nor read back and
 * the tables' pass leaves the step's serial tail.  Replaces, for the tables, the reference's `optimizer.step()` behind 16 x embedding_dense_backward
 * (nerf/utils.py:469-517 with nerf/network_hash.py:154-166).  The tables' .grad is NOT produced.  params / exp_avg / exp_avg_sq: 16 device tables [T,2],
 * 16-byte aligned; steps: 16 device scalars; lr: device scalar.
 */
int hg_levels_scatter_adam(const float *xyzs, uint32_t M, const uint32_t *rows_dev, float bound, const void *d_planes, uint32_t stride,
                           void *plan, float *const *params_host, float *const *exp_avg_host, float *const *exp_avg_sq_host,
                           float *const *steps_host, const float *lr, float beta1, float beta2, float eps, float grad_scale, float *scratch,
                           nsig_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* NERFSIG_H_ */
