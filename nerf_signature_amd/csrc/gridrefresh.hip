// The density-grid refresh as device-side work (gfx950): NeRFRenderer.update_extra_state (reference: nerf/renderer_wtmk.py:445-538, called
// every 16 training steps, nerf/utils.py:852-858) without a host read, so that a captured training loop replays it as a second hipGraph.
//
// What the reference does per refresh, and where it is here:
//   * the probe points: every cell of every cascade (the first 16 refreshes) or, per cascade, H^3/4 uniformly drawn cells plus as many cells
//     drawn (with repetition) from the occupied ones (`nonzero` + `randint`: two host synchronisations), each jittered inside its cell
//         -> rg_refresh_draw (cell keys, grouped by grid row with a counting sort; the occupied draw is a binary search in a prefix sum of the occupancy flags) and rg_refresh_points
//            (centre + jitter, and the cell's morton index).  Random numbers: splitmix64 of (seed, refresh count, cascade, draw) -- the
//            generator the captured loop already uses for its march offsets (k_clean_loss), so a replayed graph draws fresh values.
//   * density at the probe points                 -> the ordinary encoder + sigma-MLP launches (hg_encode_planes, field_fwd), issued by the caller
//   * tmp_grid[cas, indices] = sigma * scale       -> rg_refresh_scatter.  The reference's index_put with repeated indices keeps whichever
//     write lands last (undefined on a GPU); here the LARGEST candidate wins (integer atomicMax on the bits of non-negative floats): one of the
//     values the reference may produce, and the same one every time.
//   * EMA `max(grid * decay, tmp)` where both are >= 0, mean of clamp(grid, 0), threshold min(mean, density_thresh), packbits, mean_count
//         -> rg_refresh_finish: k_grid_ema (per-workgroup partial sums in double, fixed order) -> k_grid_stats (one workgroup: the mean, the refresh
//            count, the mean sample count of the window from the loop's ring) -> k_packbits_dev (threshold read from device memory).
// Everything is a pure function of (grid, parameters, seed, refresh count): two runs leave the same bits (tests/test_gpu_stage1.py, test_gpu_grid.py).
#include "common.h"

namespace nsig {

__device__ inline uint64_t rg_mix64(uint64_t z) {      // splitmix64's finaliser (as stage1.hip's mix64)
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// the key of one (refresh, cascade, purpose) stream of draws
__device__ inline uint64_t rg_stream(uint64_t seed, int32_t iter, uint32_t cas, uint32_t purpose) {
    return rg_mix64(seed ^ (0x9E3779B97F4A7C15ull * ((uint64_t)(uint32_t)iter * 64ull + (uint64_t)cas * 8ull + (uint64_t)purpose + 1ull)));
}
__device__ inline uint64_t rg_draw(uint64_t stream, uint64_t i) { return rg_mix64(stream + 0xD1B54A32D192ED03ull * (i + 1ull)); }
__device__ inline float rg_u01(uint64_t bits) { return (float)(uint32_t)(bits >> 40) * (1.0f / 16777216.0f); }      // 24 bits, [0, 1): torch.rand's grid

// ---- the inclusive prefix sum of the occupancy flags `grid[cas] > 0` (morton order), in three launches of this file's own (no library scan, no memset node: the
// captured refresh holds nothing but kernel nodes): per-workgroup counts of 4096 cells -> exclusive scan of the counts by one workgroup (k_refresh_bins) -> the
// prefix inside each workgroup's cells, on top of its offset.  The first launch also clears the counting sort's row bins.
constexpr uint32_t kOccThreads = 1024, kOccPerThread = 4, kOccCells = kOccThreads * kOccPerThread;

__device__ inline int32_t occ_block_exclusive(int32_t mine, int32_t *wave_tot) {      // exclusive prefix of `mine` over the 1024 threads of a workgroup; wave_tot[16]
    int32_t incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int32_t up = __shfl_up(incl, d, 64);
        if ((int)(threadIdx.x & 63u) >= d) incl += up;
    }
    if ((threadIdx.x & 63u) == 63u) wave_tot[threadIdx.x >> 6] = incl;
    __syncthreads();
    int32_t before = 0;
    for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) before += wave_tot[w];
    return before + incl - mine;
}

__global__ void __launch_bounds__(kOccThreads) k_occ_count(const float *__restrict__ grid, uint32_t cells, int32_t *__restrict__ block_sums, int32_t *__restrict__ bins,
                                                          uint32_t n_bins) {
    __shared__ int32_t wave_tot[16];
    for (uint32_t i = blockIdx.x * kOccThreads + threadIdx.x; i < n_bins; i += gridDim.x * kOccThreads) bins[i] = 0;
    const uint32_t first = blockIdx.x * kOccCells + threadIdx.x * kOccPerThread;
    int32_t c = 0;
    if (first + kOccPerThread <= cells) {
        const float4 g = *reinterpret_cast<const float4 *>(grid + first);
        c = (g.x > 0.0f) + (g.y > 0.0f) + (g.z > 0.0f) + (g.w > 0.0f);
    } else {
        for (uint32_t u = 0; u < kOccPerThread; ++u) c += (first + u < cells && grid[first + u] > 0.0f);
    }
    const int32_t before = occ_block_exclusive(c, wave_tot);
    if (threadIdx.x == kOccThreads - 1u) block_sums[blockIdx.x] = before + c;
}

__global__ void __launch_bounds__(kOccThreads) k_occ_prefix(const float *__restrict__ grid, uint32_t cells, const int32_t *__restrict__ block_offsets,
                                                           int32_t *__restrict__ occ_prefix) {
    __shared__ int32_t wave_tot[16];
    const uint32_t first = blockIdx.x * kOccCells + threadIdx.x * kOccPerThread;
    int32_t f[kOccPerThread];
#pragma unroll
    for (uint32_t u = 0; u < kOccPerThread; ++u) f[u] = (first + u < cells && grid[first + u] > 0.0f) ? 1 : 0;
    int32_t run = occ_block_exclusive(f[0] + f[1] + f[2] + f[3], wave_tot) + block_offsets[blockIdx.x];
#pragma unroll
    for (uint32_t u = 0; u < kOccPerThread; ++u) {
        run += f[u];
        if (first + u < cells) occ_prefix[first + u] = run;
    }
}

__global__ void __launch_bounds__(256) k_fill_f32(float *__restrict__ p, uint32_t n, float v) {
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) p[i] = v;
}

// drawn[0, N): cells drawn uniformly (renderer_wtmk.py:490 `torch.randint(0, H, (N, 3))`); drawn[N, 2N): cells drawn uniformly, with repetition, from the occupied
// ones (:493-496 `nonzero(grid > 0)[randint(0, count, [N])]`).  occ_prefix[i] = number of occupied cells among morton indices 0..i (inclusive scan of the flags);
// the t-th occupied cell is the first index whose prefix exceeds t.  A grid without an occupied cell (the reference raises there) draws cell 0.
// key = (z * H + y) * H + x.  The probe is then walked row by row of the grid ((z, y) fixed, all x): the encoder's gathers share lines along x (the hash takes x
// un-multiplied).  The rows are the bins of a counting sort -- this kernel counts, k_refresh_bins scans, k_refresh_place places -- three small launches where a
// comparison sort of the million keys (torch.sort: rocprim's merge sort, 186 us) cost as much as two thirds of the density query it was there to speed up.
__device__ inline uint32_t draw_cell_key(uint32_t i, uint32_t N, uint32_t H, const int32_t *__restrict__ occ_prefix, uint64_t seed, int32_t iter, uint32_t cas) {
    const uint32_t cells = H * H * H;
    uint32_t x, y, z;
    if (i < N) {
        const uint64_t b = rg_draw(rg_stream(seed, iter, cas, 0u), i);
        x = (uint32_t)b & (H - 1u); y = (uint32_t)(b >> 20) & (H - 1u); z = (uint32_t)(b >> 40) & (H - 1u);
    } else {
        const uint32_t total = (uint32_t)occ_prefix[cells - 1u];
        uint32_t m = 0;
        if (total > 0) {
            const uint64_t b = rg_draw(rg_stream(seed, iter, cas, 1u), i - N);
            const uint32_t t = (uint32_t)(((b >> 32) * (uint64_t)total) >> 32);      // uniform in [0, total)
            uint32_t lo = 0, hi = cells - 1u;                                        // first index with occ_prefix > t
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if ((uint32_t)occ_prefix[mid] > t) hi = mid; else lo = mid + 1u;
            }
            m = lo;
        }
        x = compact3(m); y = compact3(m >> 1); z = compact3(m >> 2);      // morton3D_invert (raymarching.cu:57-63)
    }
    return (z * H + y) * H + x;
}

__global__ void __launch_bounds__(256) k_refresh_draw(int32_t *__restrict__ drawn, uint32_t N, uint32_t H, const int32_t *__restrict__ occ_prefix, uint64_t seed,
                                                     const int32_t *__restrict__ iter_dev, uint32_t cas, int32_t *__restrict__ bins) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= 2u * N) return;
    const uint32_t key = draw_cell_key(i, N, H, occ_prefix, seed, *iter_dev, cas);
    drawn[i] = (int32_t)key;
    atomicAdd(bins + key / H, 1);
}

// ---- the same counting sort with the row histogram PRIVATE to a workgroup (H^2 <= 16 384 rows = 64 KiB of LDS; the occupied half of the draws lands in the few rows
// of the scene: ~350 global atomics per row address made the two passes above 74 + 72 us at 1 M draws).  A workgroup owns kDrawChunk consecutive draws in both passes:
// it counts them (k_refresh_hist_lds) in LDS and stores its histogram; k_refresh_offsets turns the histograms into every workgroup's first position in every row; the workgroup then places
// its draws with LDS atomics on its own cursors.  No global atomic anywhere.
constexpr uint32_t kDrawThreads = 1024, kDrawChunk = 8 * kDrawThreads, kDrawMaxBins = 16384;

// (the draw itself -- a hash, and for the occupied half a 21-step binary search of dependent loads -- wants every compute unit: a launch of its own, 256-thread workgroups;
// inside the 128 histogram workgroups it ran at their occupancy: 120 us instead of 40)
__global__ void __launch_bounds__(256) k_refresh_draw_keys(int32_t *__restrict__ drawn, uint32_t N, uint32_t H, const int32_t *__restrict__ occ_prefix, uint64_t seed,
                                                          const int32_t *__restrict__ iter_dev, uint32_t cas) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < 2u * N) drawn[i] = (int32_t)draw_cell_key(i, N, H, occ_prefix, seed, *iter_dev, cas);
}

__global__ void __launch_bounds__(kDrawThreads) k_refresh_hist_lds(const int32_t *__restrict__ drawn, uint32_t n, uint32_t H, int32_t *__restrict__ wg_hist) {
    extern __shared__ int32_t rows[];
    const uint32_t n_bins = H * H;
    for (uint32_t b = threadIdx.x; b < n_bins; b += kDrawThreads) rows[b] = 0;
    __syncthreads();
    for (uint32_t u = 0; u < kDrawChunk / kDrawThreads; ++u) {
        const uint32_t i = blockIdx.x * kDrawChunk + u * kDrawThreads + threadIdx.x;
        if (i < n) atomicAdd(rows + (uint32_t)drawn[i] / H, 1);
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < n_bins; b += kDrawThreads) wg_hist[(size_t)blockIdx.x * n_bins + b] = rows[b];
}

// wg_hist[w][r] (draws of workgroup w in row r) -> the position of workgroup w's first draw INSIDE row r; totals[r] = the row's draws (k_refresh_bins scans them next).
// A thread per row; eight workgroups' counts requested at a time (the loads do not depend on the running sum).
__global__ void __launch_bounds__(256) k_refresh_offsets(int32_t *__restrict__ wg_hist, uint32_t n_wg, uint32_t n_bins, int32_t *__restrict__ totals) {
    const uint32_t b = blockIdx.x * 256u + threadIdx.x;
    if (b >= n_bins) return;
    int32_t run = 0;
    for (uint32_t w0 = 0; w0 < n_wg; w0 += 8u) {
        int32_t c[8];
#pragma unroll
        for (uint32_t u = 0; u < 8u; ++u) c[u] = w0 + u < n_wg ? wg_hist[(size_t)(w0 + u) * n_bins + b] : 0;
#pragma unroll
        for (uint32_t u = 0; u < 8u; ++u) {
            if (w0 + u < n_wg) wg_hist[(size_t)(w0 + u) * n_bins + b] = run;
            run += c[u];
        }
    }
    totals[b] = run;
}

__global__ void __launch_bounds__(kDrawThreads) k_refresh_place_lds(const int32_t *__restrict__ drawn, uint32_t n, uint32_t H, const int32_t *__restrict__ wg_hist,
                                                                   const int32_t *__restrict__ row_start, int32_t *__restrict__ keys, int32_t *__restrict__ ids) {
    extern __shared__ int32_t rows[];
    const uint32_t n_bins = H * H;
    for (uint32_t b = threadIdx.x; b < n_bins; b += kDrawThreads) rows[b] = row_start[b] + wg_hist[(size_t)blockIdx.x * n_bins + b];
    __syncthreads();
    for (uint32_t u = 0; u < kDrawChunk / kDrawThreads; ++u) {
        const uint32_t i = blockIdx.x * kDrawChunk + u * kDrawThreads + threadIdx.x;
        if (i < n) {
            const int32_t key = drawn[i];
            const int32_t p = atomicAdd(rows + (uint32_t)key / H, 1);
            keys[p] = key;
            ids[p] = (int32_t)i;
        }
    }
}

// bins[r] (entries of row r) -> the row's first position: an exclusive scan by ONE workgroup (H^2 bins: 16 384 at H = 128)
__global__ void __launch_bounds__(1024) k_refresh_bins(int32_t *__restrict__ bins, uint32_t n_bins) {
    __shared__ int32_t wave_tot[16];
    __shared__ int32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n_bins; base += 1024u) {      // (uniform trip count)
        const uint32_t i = base + threadIdx.x;
        const int32_t v = i < n_bins ? bins[i] : 0;
        int32_t incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int32_t up = __shfl_up(incl, d, 64);
            if ((int)(threadIdx.x & 63u) >= d) incl += up;
        }
        if ((threadIdx.x & 63u) == 63u) wave_tot[threadIdx.x >> 6] = incl;
        __syncthreads();
        int32_t before = carry;
        for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) before += wave_tot[w];
        if (i < n_bins) bins[i] = before + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023u) carry = before + incl;
        __syncthreads();
    }
}

// keys[p] / ids[p]: the draws grouped by grid row.  WHERE inside its row a draw lands depends on the order the atomics arrive in and may differ from run to run;
// nothing downstream depends on it -- a draw's jitter is a function of its id, and the scatter keeps the maximum over a cell's candidates.
__global__ void __launch_bounds__(256) k_refresh_place(const int32_t *__restrict__ drawn, uint32_t n, uint32_t H, int32_t *__restrict__ bins, int32_t *__restrict__ keys,
                                                      int32_t *__restrict__ ids) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const int32_t key = drawn[i];
    const int32_t p = atomicAdd(bins + (uint32_t)key / H, 1);
    keys[p] = key;
    ids[p] = (int32_t)i;
}

// Probe point p: the centre of its cell in cascade `cas`, jittered inside the cell -- renderer_wtmk.py:474,480-484 operation for operation:
//   xyzs = 2 * coords.float() / (H - 1) - 1;  cas_xyzs = xyzs * (bound - half);  cas_xyzs += (rand * 2 - 1) * half
// keys == nullptr: the full refresh, key = p.  The jitter is drawn for the draw (ids[p]; the full refresh: p), not for the position p.
__global__ void __launch_bounds__(256) k_refresh_points(const int32_t *__restrict__ keys, const int32_t *__restrict__ ids, uint32_t n, uint32_t H, float extent,
                                                       float half_cell, uint64_t seed, const int32_t *__restrict__ iter_dev, uint32_t cas, float *__restrict__ xyz,
                                                       int32_t *__restrict__ cells) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint32_t key = keys != nullptr ? (uint32_t)keys[i] : i, id = ids != nullptr ? (uint32_t)ids[i] : i;
    const uint32_t x = key % H, y = (key / H) % H, z = key / (H * H);
    const uint64_t stream = rg_stream(seed, *iter_dev, cas, 2u);
    const float scale = extent - half_cell, hm1 = (float)(H - 1u);
    const uint32_t c[3] = {x, y, z};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float u = rg_u01(rg_draw(stream, 3ull * id + (uint64_t)a));
        const float centre = ((2.0f * (float)c[a]) / hm1 - 1.0f) * scale;
        xyz[3u * i + a] = centre + (u * 2.0f - 1.0f) * half_cell;
    }
    cells[i] = (int32_t)morton3(x, y, z);
}

// fresh[cell] = max(fresh[cell], sigma * density_scale): fresh starts at -1 (negative as an integer too), the candidates are >= 0, and for non-negative floats the
// integer order of the bit patterns is the float order.  A NaN density (diverged parameters) has the largest pattern of all and stays visible.
__global__ void __launch_bounds__(256) k_refresh_scatter(const float *__restrict__ sigmas, const int32_t *__restrict__ cells, uint32_t n, float density_scale,
                                                        float *__restrict__ fresh) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const float v = sigmas[i] * density_scale;
    atomicMax(reinterpret_cast<int *>(fresh) + cells[i], __float_as_int(v >= 0.0f || v != v ? v : 0.0f));
}

constexpr uint32_t kEmaThreads = 256, kEmaPerThread = 16;      // 4096 cells per workgroup: 512 partial sums for one 128^3 cascade
// renderer_wtmk.py:521-522: grid = max(grid * decay, fresh) where both are >= 0; the partial sum of clamp(grid, 0) of this workgroup's cells, in double, lanes
// and waves combined in a fixed order.
__global__ void __launch_bounds__(kEmaThreads) k_grid_ema(float *__restrict__ grid, const float *__restrict__ fresh, uint32_t n, float decay, double *__restrict__ partials) {
    __shared__ double part[kEmaThreads / 64];
    const uint32_t base = blockIdx.x * kEmaThreads * kEmaPerThread;
    double s = 0.0;
#pragma unroll 4
    for (uint32_t u = 0; u < kEmaPerThread; ++u) {
        const uint32_t i = base + u * kEmaThreads + threadIdx.x;
        if (i < n) {
            float g = grid[i];
            const float f = fresh[i];
            if (g >= 0.0f && f >= 0.0f) {
                g = fmaxf(g * decay, f);
                grid[i] = g;
            }
            s += (double)fmaxf(g, 0.0f);
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
    if ((threadIdx.x & 63u) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (uint32_t w = 0; w < kEmaThreads / 64; ++w) t += part[w];
        partials[blockIdx.x] = t;
    }
}

// One workgroup: mean_density = sum(partials) / n (:523), the refresh count advanced (:525), and -- window > 0 -- mean_count = int(sum of the window's sample totals
// / window) (:533-536) from the captured loop's ring: rows (steps - window + i) % 16, i < window, `steps` = the loop's device step count.
__global__ void __launch_bounds__(256) k_grid_stats(const double *__restrict__ partials, uint32_t n_partials, uint32_t n, float *__restrict__ mean_density,
                                                   int32_t *__restrict__ iter_dev, const int32_t *__restrict__ count_ring, uint32_t window,
                                                   const uint32_t *__restrict__ step_dev, int32_t *__restrict__ mean_count) {
    __shared__ double part[256];
    double s = 0.0;
    for (uint32_t i = threadIdx.x; i < n_partials; i += 256u) s += partials[i];      // (a thread's partials in index order)
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (uint32_t w = 0; w < 256u; ++w) t += part[w];
        mean_density[0] = (float)(t / (double)n);
        iter_dev[0] += 1;
        if (window > 0 && count_ring != nullptr && mean_count != nullptr) {
            const uint32_t steps = step_dev != nullptr ? *step_dev : window;
            long long total = 0;
            for (uint32_t i = 0; i < window; ++i) total += (long long)count_ring[2u * ((steps - window + i) % 16u)];
            mean_count[0] = (int32_t)((double)total / (double)window);
        }
    }
}

// k_packbits (raymarch.hip) with the threshold min(mean_density, density_thresh) (:528) read from device memory
__global__ void k_packbits_dev(const float *__restrict__ grid, uint32_t n_bytes, const float *__restrict__ mean_density, float density_thresh, uint8_t *__restrict__ bits) {
    const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;  // word index
    const uint32_t first = w * 4;
    if (first >= n_bytes) return;
    const float thresh = fminf(mean_density[0], density_thresh);
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

}  // namespace nsig

using namespace nsig;

static int check_refresh_grid(const char *who, uint32_t H) {
    NSIG_REQUIRE(H >= 8 && H <= 1024 && (H & (H - 1)) == 0, "%s: grid size %u must be a power of two in [8,1024]", who, H);
    return NSIG_OK;
}

// scratch of rg_refresh_draw, in int32 words: [2N draws | H^2 row bins | H^3 occupancy prefix | ceil(H^3 / 4096) workgroup counts | (H^2 <= 16 384) ceil(2N / 8192) x H^2 histograms]
NSIG_EXPORT size_t rg_refresh_draw_scratch_bytes(uint32_t N, uint32_t H) {
    const size_t cells = (size_t)H * H * H, n_bins = (size_t)H * H;
    const size_t hist = n_bins <= kDrawMaxBins ? (((size_t)2 * N + kDrawChunk - 1) / kDrawChunk) * n_bins : 0;
    return ((size_t)2 * N + n_bins + cells + (cells + kOccCells - 1) / kOccCells + hist) * sizeof(int32_t);
}

NSIG_EXPORT int rg_refresh_begin(float *fresh, uint32_t n_cells, nsig_stream_t stream) {
    NSIG_REQUIRE(fresh && n_cells >= 1, "rg_refresh_begin: null pointer or no cells");
    k_fill_f32<<<min(ceil_div(n_cells, 256u), 2048u), 256, 0, as_stream(stream)>>>(fresh, n_cells, -1.0f);
    return check_launch("rg_refresh_begin");
}

NSIG_EXPORT int rg_refresh_draw(int32_t *keys, int32_t *ids, uint32_t N, uint32_t H, const float *grid_cas, void *scratch, uint64_t seed, const int32_t *iter_dev,
                                uint32_t cas, nsig_stream_t stream) {
    if (int e = check_refresh_grid("rg_refresh_draw", H)) return e;
    NSIG_REQUIRE(keys && ids && grid_cas && scratch && iter_dev, "rg_refresh_draw: null pointer");
    NSIG_REQUIRE(N >= 1 && N < (1u << 29) && cas < 8, "rg_refresh_draw: N must be in [1, 2^29), cascade < 8");
    NSIG_REQUIRE((reinterpret_cast<uintptr_t>(scratch) & 3) == 0 && (reinterpret_cast<uintptr_t>(grid_cas) & 15) == 0, "rg_refresh_draw: scratch must be 4-byte, the grid 16-byte aligned");
    hipStream_t st = as_stream(stream);
    const uint32_t cells = H * H * H, n_bins = H * H, occ_blocks = ceil_div(cells, kOccCells);
    int32_t *drawn = reinterpret_cast<int32_t *>(scratch), *bins = drawn + 2 * (size_t)N, *occ_prefix = bins + n_bins, *block_sums = occ_prefix + cells;
    k_occ_count<<<occ_blocks, kOccThreads, 0, st>>>(grid_cas, cells, block_sums, bins, n_bins);
    k_refresh_bins<<<1, 1024, 0, st>>>(block_sums, occ_blocks);
    k_occ_prefix<<<occ_blocks, kOccThreads, 0, st>>>(grid_cas, cells, block_sums, occ_prefix);
    if (n_bins <= kDrawMaxBins) {      // the row histogram fits a workgroup's LDS: no global atomics (k_occ_count's clearing of `bins` is then unused: the totals overwrite them)
        static bool attr_set = false;
        const size_t lds = (size_t)n_bins * sizeof(int32_t);
        if (!attr_set) {
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_refresh_hist_lds), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kDrawMaxBins * sizeof(int32_t))) != hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void *>(k_refresh_place_lds), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kDrawMaxBins * sizeof(int32_t))) != hipSuccess) {
                set_error("rg_refresh_draw: cannot reserve %zu bytes of LDS", (size_t)kDrawMaxBins * sizeof(int32_t));
                return NSIG_ERR_LAUNCH;
            }
            attr_set = true;
        }
        const uint32_t n_wg = ceil_div(2u * N, kDrawChunk);
        int32_t *wg_hist = block_sums + occ_blocks;
        k_refresh_draw_keys<<<ceil_div(2u * N, 256u), 256, 0, st>>>(drawn, N, H, occ_prefix, seed, iter_dev, cas);
        k_refresh_hist_lds<<<n_wg, kDrawThreads, lds, st>>>(drawn, 2u * N, H, wg_hist);
        k_refresh_offsets<<<ceil_div(n_bins, 256u), 256, 0, st>>>(wg_hist, n_wg, n_bins, bins);
        k_refresh_bins<<<1, 1024, 0, st>>>(bins, n_bins);
        k_refresh_place_lds<<<n_wg, kDrawThreads, lds, st>>>(drawn, 2u * N, H, wg_hist, bins, keys, ids);
        return check_launch("rg_refresh_draw");
    }
    k_refresh_draw<<<ceil_div(2u * N, 256u), 256, 0, st>>>(drawn, N, H, occ_prefix, seed, iter_dev, cas, bins);
    k_refresh_bins<<<1, 1024, 0, st>>>(bins, n_bins);
    k_refresh_place<<<ceil_div(2u * N, 256u), 256, 0, st>>>(drawn, 2u * N, H, bins, keys, ids);
    return check_launch("rg_refresh_draw");
}

NSIG_EXPORT int rg_refresh_points(const int32_t *keys, const int32_t *ids, uint32_t n, uint32_t H, float extent, float half_cell, uint64_t seed, const int32_t *iter_dev,
                                  uint32_t cas, float *xyz, int32_t *cells, nsig_stream_t stream) {
    if (int e = check_refresh_grid("rg_refresh_points", H)) return e;
    NSIG_REQUIRE(iter_dev && xyz && cells, "rg_refresh_points: null pointer");
    NSIG_REQUIRE(n >= 1 && extent > 0.0f && half_cell > 0.0f && cas < 8, "rg_refresh_points: n, extent and half_cell must be positive, cascade < 8");
    NSIG_REQUIRE((keys != nullptr) == (ids != nullptr), "rg_refresh_points: keys and ids come together (rg_refresh_draw) or not at all");
    NSIG_REQUIRE(keys != nullptr || n == H * H * H, "rg_refresh_points: without keys the probe is the whole grid (n = H^3)");
    k_refresh_points<<<ceil_div(n, 256u), 256, 0, as_stream(stream)>>>(keys, ids, n, H, extent, half_cell, seed, iter_dev, cas, xyz, cells);
    return check_launch("rg_refresh_points");
}

NSIG_EXPORT int rg_refresh_scatter(const float *sigmas, const int32_t *cells, uint32_t n, float density_scale, float *fresh, nsig_stream_t stream) {
    NSIG_REQUIRE(sigmas && cells && fresh, "rg_refresh_scatter: null pointer");
    if (n == 0) return NSIG_OK;
    k_refresh_scatter<<<ceil_div(n, 256u), 256, 0, as_stream(stream)>>>(sigmas, cells, n, density_scale, fresh);
    return check_launch("rg_refresh_scatter");
}

NSIG_EXPORT size_t rg_refresh_partials_bytes(uint32_t n_cells) { return (size_t)ceil_div(n_cells, kEmaThreads * kEmaPerThread) * sizeof(double); }

NSIG_EXPORT int rg_refresh_finish(float *grid, const float *fresh, uint32_t n_cells, float decay, void *partials, float density_thresh, uint8_t *bitfield,
                                  float *mean_density, int32_t *iter_dev, const int32_t *count_ring, uint32_t window, const uint32_t *step_dev, int32_t *mean_count,
                                  nsig_stream_t stream) {
    NSIG_REQUIRE(grid && fresh && partials && bitfield && mean_density && iter_dev, "rg_refresh_finish: null pointer");
    NSIG_REQUIRE(n_cells >= 8 && n_cells % 8 == 0 && window <= 16, "rg_refresh_finish: the cell count must be a positive multiple of 8, window <= 16");
    NSIG_REQUIRE(window == 0 || (count_ring && mean_count), "rg_refresh_finish: a window needs the count ring and somewhere to put the mean");
    NSIG_REQUIRE((reinterpret_cast<uintptr_t>(grid) & 15) == 0 && (reinterpret_cast<uintptr_t>(partials) & 7) == 0, "rg_refresh_finish: grid must be 16-byte, partials 8-byte aligned");
    hipStream_t st = as_stream(stream);
    const uint32_t blocks = ceil_div(n_cells, kEmaThreads * kEmaPerThread), n_bytes = n_cells / 8;
    k_grid_ema<<<blocks, kEmaThreads, 0, st>>>(grid, fresh, n_cells, decay, reinterpret_cast<double *>(partials));
    k_grid_stats<<<1, 256, 0, st>>>(reinterpret_cast<const double *>(partials), blocks, n_cells, mean_density, iter_dev, count_ring, window, step_dev, mean_count);
    k_packbits_dev<<<ceil_div(ceil_div(n_bytes, 4), 256), 256, 0, st>>>(grid, n_bytes, mean_density, density_thresh, bitfield);
    return check_launch("rg_refresh_finish");
}
