// Hash-grid kernels for gfx950: codebook pre-sum, stand-alone encoders, codebook backward scatter,
// gradient fan-out.  Behavioural references: /root/reference/hash_encoding.py:96-111 (base encoder),
// /root/reference/hash_encoding_wtmk_bit.py:99-116 (codebook encoder) and their autograd.
//
// Two algebraic facts of the reference drive the design (both verified against the reference itself in
// tests/test_oracle_golden.py):
//   (1) all D codebook levels have the same resolution (2048) and hash, hence identical corner rows and
//       weights, and the encoder output is the SUM of the D interpolations.  Interpolation is linear in
//       the table, so  sum_i lerp(table_i) == lerp(sum_i table_i):  one streaming pass builds
//       S = sum_i table_{2i+bit_i}  (D x 4 MiB read, 4 MiB written) and every point then needs ONE
//       8-corner gather in S instead of D of them;
//   (2) for the same reason every selected table receives the SAME gradient G[T,2]; the backward
//       scatters once into G and a fan-out pass copies G into the D gradient tensors autograd expects.
#include <cstdlib>
#include "hashgrid.h"

namespace nsig {

// S[e] = sum_i tables[i][e]; float4 per lane (1 KiB per wave-instruction), D independent streams.
__global__ void __launch_bounds__(256) k_codebook_presum(CodebookPtrs tabs, uint32_t D, float4 *__restrict__ S) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;  // float4 index, T*2/4 of them
    if (e >= NSIG_TABLE_ROWS / 2) return;
    float4 acc = {0.f, 0.f, 0.f, 0.f};
    uint32_t i = 0;
    for (; i + 4 <= D; i += 4) {  // four loads in flight per lane
        const float4 a = reinterpret_cast<const float4 *>(tabs.p[i])[e];
        const float4 b = reinterpret_cast<const float4 *>(tabs.p[i + 1])[e];
        const float4 c = reinterpret_cast<const float4 *>(tabs.p[i + 2])[e];
        const float4 d = reinterpret_cast<const float4 *>(tabs.p[i + 3])[e];
        acc.x = (((acc.x + a.x) + b.x) + c.x) + d.x;
        acc.y = (((acc.y + a.y) + b.y) + c.y) + d.y;
        acc.z = (((acc.z + a.z) + b.z) + c.z) + d.z;
        acc.w = (((acc.w + a.w) + b.w) + c.w) + d.w;
    }
    for (; i < D; ++i) {
        const float4 a = reinterpret_cast<const float4 *>(tabs.p[i])[e];
        acc.x += a.x; acc.y += a.y; acc.z += a.z; acc.w += a.w;
    }
    S[e] = acc;
}

// Stand-alone base encoder (+ optional codebook via S): one lane per (point, level); a wave covers
// 4 points x 16 levels and stores 4 full 128-byte feature rows.
__global__ void __launch_bounds__(256) k_encode(const float *__restrict__ x01, uint32_t M, TablePtrs base, LevelGeom geom,
                                                const float *__restrict__ S, float *__restrict__ feat) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t m = gid >> 4, l = gid & 15u;
    if (m >= M) return;
    const float x = x01[3 * (size_t)m], y = x01[3 * (size_t)m + 1], z = x01[3 * (size_t)m + 2];
    float2 f = encode_level(base.p[l], x, y, z, geom.cell[l]);
    if (S != nullptr && l == 15u) {  // network_wtmk_tcnn.py:106: codebook feature added into channels 30:32
        const float2 c = encode_level(S, x, y, z, geom.cell[NSIG_BASE_LEVELS]);
        f.x = f.x + c.x;
        f.y = f.y + c.y;
    }
    reinterpret_cast<float2 *>(feat)[(size_t)m * 16 + l] = f;
}

// The codebook encoder evaluated literally (D gathers per corner, summed in bit order).
__global__ void __launch_bounds__(256) k_codebook_encode(const float *__restrict__ x01, uint32_t M, CodebookPtrs tabs, uint32_t D,
                                                         float cell, float *__restrict__ out) {
    const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    Corner8 c;
    corner_rows(x01[3 * (size_t)m], x01[3 * (size_t)m + 1], x01[3 * (size_t)m + 2], cell, c);
    float2 acc = {0.f, 0.f};
    for (uint32_t i = 0; i < D; ++i) {
        float2 e[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) e[k] = reinterpret_cast<const float2 *>(tabs.p[i])[c.row[k]];
        const float2 v = trilerp(e, c.wx, c.wy, c.wz);
        acc.x += v.x;
        acc.y += v.y;
    }
    reinterpret_cast<float2 *>(out)[m] = acc;
}

// G[row_k] += w_k * dfeat for the 8 corners of every point: 16 lanes per point, lane = (corner, feature),
// so the two features of a row are adjacent lanes (one 8-byte segment per pair of lanes).
__global__ void __launch_bounds__(256) k_codebook_bwd(const float *__restrict__ x01, uint32_t M, const float *__restrict__ dfeat,
                                                      float cell, float *__restrict__ G) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t m = gid >> 4, k = (gid >> 1) & 7u, f = gid & 1u;
    if (m >= M) return;
    const float g = dfeat[2 * (size_t)m + f];
    if (g == 0.0f) return;  // padded rows and rays past early termination carry exact zeros
    Corner8 c;
    corner_rows(x01[3 * (size_t)m], x01[3 * (size_t)m + 1], x01[3 * (size_t)m + 2], cell, c);
    atomicAdd(G + 2 * (size_t)c.row[k] + f, corner_weight(c, (int)k, g));
}

// Owner-computes scatter.  Memory-side float atomics retire ~2e10 requests/s chip-wide however they are spread
// (MI355X_MICROARCH.md, Global float atomics), and the point-wise scatter needs >= 4 requests per point; LDS
// atomics and a contiguous flush need 1/10 of that.  Workgroup (slice s, replica r) owns rows [s*16384, (s+1)*16384)
// of G in LDS, scans points [r*M/8, (r+1)*M/8) and keeps only the corners whose row falls in its slice (the hash
// makes that 1 corner in 32, so every workgroup tests every point: 32x redundant work, cheap next to the atomics it
// replaces -- and kept cheap: the record carries the integer cell and the weights, computed once by k_field_bwd).
// blockIdx = s*8 + r, so the 32 workgroups that scan the same points share one XCD (blockIdx % 8) and its L2 serves the
// re-reads.  Phase stamps (tools/scatter_timing.py): 1.4 us zeroing, the scan, 6 us flush; the scan was a chain of exposed L2
// latencies (one record in flight per thread) until four were prefetched.  Tried and rejected: collecting hits in a bit mask or
// in a per-wave LDS ring to run the weight/atomic code with full lanes -- both slower than the plain predicated body.
constexpr int kSliceRows = 16384, kSlices = NSIG_TABLE_ROWS / kSliceRows, kReplicas = 8;

#ifdef NSIG_DEC_TIMING
__device__ unsigned long long g_scatter_stamps[8];
#define SC_STAMP(k)                                                                      \
    do {                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                               \
        if (blockIdx.x == 0 && threadIdx.x == 0) g_scatter_stamps[k] = wall_clock64();   \
        __builtin_amdgcn_sched_barrier(0);                                               \
    } while (0)
#else
#define SC_STAMP(k)
#endif

// `sets` record arrays of M points each (one for the codebook; 16, one per base level, in stage-1 training), each scattered into
// its own table.  replicas == 1: every row has one owner, which stores.  Workgroup order for sets = 16: rounds of 8 sets, set =
// 8*round + blockIdx % 8, so the 32*replicas workgroups that scan one set's records share an XCD (blockIdx % 8) whose L2 holds
// that one record array (4 MiB at 131 k points); interleaving all 16 sets put two arrays on every XCD and doubled the time.
struct ScatterTargets {
    float *g[NSIG_BASE_LEVELS];
};
__global__ void __launch_bounds__(1024) k_scatter_sliced(const float *__restrict__ rec_all, uint32_t M, ScatterTargets tg, uint32_t sets, uint32_t replicas) {
    extern __shared__ float acc[];  // [kSliceRows][2]
    uint32_t set = 0, sr = blockIdx.x;
    if (sets > 1) {
        const uint32_t per_round = 8u * kSlices * replicas, round = blockIdx.x / per_round, b = blockIdx.x - round * per_round;
        set = round * 8u + (b & 7u);
        sr = b >> 3;
        if (set >= sets) return;
    }
    const uint32_t slice = sr / replicas, replica = sr - slice * replicas;
    const float *__restrict__ rec = rec_all + (size_t)set * M * 8;
    float *__restrict__ G = tg.g[set];
    const uint32_t kReplicasRt = replicas;
    SC_STAMP(0);
    for (uint32_t i = threadIdx.x; i < 2u * kSliceRows; i += blockDim.x) acc[i] = 0.0f;
    __syncthreads();
    SC_STAMP(1);
    const uint32_t chunk = ceil_div(M, kReplicasRt);
    const uint32_t beg = min(M, replica * chunk), end = min(M, beg + chunk);
    const uint4 *__restrict__ rec4 = reinterpret_cast<const uint4 *>(rec);
    // With 16 waves per CU and ~150 points per thread the scan is a chain of exposed L2 latencies unless several points'
    // records are in flight per thread: four at a time.
    constexpr int kAhead = 4;
    for (uint32_t m0 = beg + threadIdx.x; m0 < end; m0 += blockDim.x * kAhead) {
        uint4 ra[kAhead], rb[kAhead];
#pragma unroll
        for (int u = 0; u < kAhead; ++u) {
            const uint32_t m = min(m0 + u * blockDim.x, end - 1);   // clamped: the duplicate is skipped below
            ra[u] = rec4[2 * (size_t)m];
            rb[u] = rec4[2 * (size_t)m + 1];
        }
#pragma unroll
        for (int u = 0; u < kAhead; ++u) {
            if (m0 + u * blockDim.x >= end) break;
            const float g0 = __uint_as_float(rb[u].y), g1 = __uint_as_float(rb[u].z);
            if (g0 == 0.0f && g1 == 0.0f) continue;  // padding rows and terminated rays
            Corner8 c;
            c.wx = __uint_as_float(ra[u].z); c.wy = __uint_as_float(ra[u].w); c.wz = __uint_as_float(rb[u].x);
            const uint32_t ix = ra[u].x & 0xffffu, iy = ra[u].x >> 16, iz = ra[u].y;
            const uint32_t hy[2] = {iy * kPrimeY, (iy + 1u) * kPrimeY};
            const uint32_t hz[2] = {iz * kPrimeZ, (iz + 1u) * kPrimeZ};
            // ix <= 2^11 never reaches bits 14..18 of the row, so the slice of a corner depends on (dy, dz) only: four tests,
            // and a hit brings both x-corners.
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t hyz = hy[q >> 1] ^ hz[q & 1];
                if (((hyz >> 14) & (uint32_t)(kSlices - 1)) != slice) continue;
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    const int k = 4 * dx + q;   // corner (dx, dy, dz) = (dx, q>>1, q&1)
                    float *dst = acc + 2u * (((ix + dx) ^ hyz) & (uint32_t)(kSliceRows - 1));
                    atomicAdd(dst, corner_weight(c, k, g0));
                    atomicAdd(dst + 1, corner_weight(c, k, g1));
                }
            }
        }
    }
    SC_STAMP(2);
    __syncthreads();
    SC_STAMP(3);
    float *out = G + 2 * (size_t)slice * kSliceRows;
    if (replicas == 1) {
        for (uint32_t i = threadIdx.x; i < 2u * kSliceRows; i += blockDim.x) out[i] = acc[i];
    } else {
        for (uint32_t i = threadIdx.x; i < 2u * kSliceRows; i += blockDim.x) {
            const float v = acc[i];
            if (v != 0.0f) atomicAdd(out + i, v);
        }
    }
    SC_STAMP(4);
}

#ifdef NSIG_DEC_TIMING
NSIG_EXPORT int scatter_timing_stamps(unsigned long long *out8) {
    return hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_scatter_stamps), sizeof(unsigned long long) * 8) == hipSuccess ? 0 : 1;
}
#endif

// Owner-computes scatter of one base level's feature gradient (stage-1 training): as k_scatter_sliced, with the level's
// own resolution (not a power of two, so the cell index uses the same IEEE divisions as the forward: corner_rows()).
__global__ void __launch_bounds__(1024) k_scatter_level(const float *__restrict__ xyzs, float bound, const float2 *__restrict__ dplane, uint32_t M,
                                                        float cell, float *__restrict__ G) {
    extern __shared__ float acc[];
    const uint32_t slice = blockIdx.x >> 3, replica = blockIdx.x & 7u;
    for (uint32_t i = threadIdx.x; i < 2u * kSliceRows; i += blockDim.x) acc[i] = 0.0f;
    __syncthreads();
    const uint32_t chunk = ceil_div(M, (uint32_t)kReplicas);
    const uint32_t beg = min(M, replica * chunk), end = min(M, beg + chunk);
    const float two_b = 2.0f * bound;
    for (uint32_t m = beg + threadIdx.x; m < end; m += blockDim.x) {
        const float2 g = dplane[m];
        if (g.x == 0.0f && g.y == 0.0f) continue;
        Corner8 c;
        corner_rows((xyzs[3 * (size_t)m] + bound) / two_b, (xyzs[3 * (size_t)m + 1] + bound) / two_b, (xyzs[3 * (size_t)m + 2] + bound) / two_b, cell, c);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if ((c.row[k] >> 14) != slice) continue;
            float *dst = acc + 2u * (c.row[k] & (kSliceRows - 1));
            atomicAdd(dst, corner_weight(c, k, g.x));
            atomicAdd(dst + 1, corner_weight(c, k, g.y));
        }
    }
    __syncthreads();
    float *out = G + 2 * (size_t)slice * kSliceRows;
    for (uint32_t i = threadIdx.x; i < 2u * kSliceRows; i += blockDim.x) {
        const float v = acc[i];
        if (v != 0.0f) atomicAdd(out + i, v);
    }
}

// ----------------------------------------------------------------------------- binned scatter
//
// Two measurements shape this version (tools/micro/lds_atomic_rate.hip, tools/scatter_timing.py):
//  * ds_add_f32 retires 0.33 lane-operations per clock per CU; ds_add_u64 4.4 (ds_add_u32 5.2).  The scatter's 20.6 M LDS float
//    atomics alone are 100 us -- that, not the instruction count, bounded k_scatter_sliced.  So the owners accumulate in 64-bit
//    FIXED POINT: contribution * 2^k rounded to an integer, k chosen from the largest |gradient| of the launch so that 2^11
//    maximal contributions cannot overflow; everything up to 2^-51 of that maximum is represented, the integer sum is exact
//    and independent of the order of the atomics, and the owner converts back once per row.
//  * with 16 bytes per (row, feature pair) an LDS slice holds 8192 rows: 64 slices.  Letting every owner test every point would
//    double the redundant work, so the (point, (dy,dz) pair) hits are first grouped by slice with an exact two-pass counting
//    sort -- count, then write at atomically reserved positions -- and an owner streams only its own entries.  An entry is
//    self-contained (16 bytes: x cell, the pair's 13 row bits, the x weight, the two gradients already multiplied by the y and
//    z weights); gathering 32-byte records at random from the 41 MB array instead made this slower than what it replaced.
// The slice of a pair depends on (iy, iz, dy, dz) only (ix < 2^12 stays below bit 13), and a hit carries both x corners.
static_assert(kBinRows == 8192, "slice layout");   // constants and BinHeader: hashgrid.h

__device__ inline bool record_pairs(const uint4 *__restrict__ rec4, uint32_t m, uint32_t (&slices)[4], float &gabs) {
    const uint4 ra = rec4[2 * (size_t)m], rb = rec4[2 * (size_t)m + 1];
    const float g0 = __uint_as_float(rb.y), g1 = __uint_as_float(rb.z);
    if (g0 == 0.0f && g1 == 0.0f) return false;   // padding rows and terminated rays
    gabs = __uint_as_float(max(__float_as_uint(g0) & 0x7fffffffu, __float_as_uint(g1) & 0x7fffffffu));   // integer max of |bits|: a NaN stays visible
#pragma unroll
    for (uint32_t q = 0; q < 4; ++q) slices[q] = pair_slice(pair_hash(ra.x >> 16, ra.y, q));
    return true;
}

// Pass 1: wg[w][s] = entries of slice s among the chunks of workgroup w (plain stores: no global atomics); gmax = max |gradient|.
// (all four kernels: blockIdx.y = record set -- one for the codebook, 16 base levels in stage-1 training)
__global__ void __launch_bounds__(kBinThreads) k_bin_count(const float *__restrict__ rec_all, uint32_t M, BinHeader *__restrict__ hd_all) {
    const float *__restrict__ rec = rec_all + (size_t)blockIdx.y * M * 8;
    BinHeader *__restrict__ hd = hd_all + blockIdx.y;
    __shared__ uint32_t h[kBinSlices], gm;
    if (threadIdx.x < kBinSlices) h[threadIdx.x] = 0;
    if (threadIdx.x == 0) gm = 0;
    __syncthreads();
    uint32_t gb = 0;
    for (uint32_t m = blockIdx.x * kBinThreads + threadIdx.x; m < M; m += gridDim.x * kBinThreads) {
        uint32_t sl[4];
        float gabs;
        if (record_pairs(reinterpret_cast<const uint4 *>(rec), m, sl, gabs)) {
#pragma unroll
            for (int q = 0; q < 4; ++q) atomicAdd(&h[sl[q]], 1u);
            gb = max(gb, __float_as_uint(gabs));   // non-negative floats (and +inf, NaN) order like their bit patterns
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) gb = max(gb, (uint32_t)__shfl_xor((int)gb, d, 64));
    if ((threadIdx.x & 63) == 0 && gb) atomicMax(&gm, gb);
    __syncthreads();
    if (threadIdx.x < kBinSlices) hd->wg[blockIdx.x][threadIdx.x] = h[threadIdx.x];
    if (threadIdx.x == 0 && gm) atomicMax(&hd->gmax_bits, gm);
}

// One workgroup: slice totals, slice starts, and every (workgroup, slice) write offset -- the queue order is a pure function of
// the input, so the whole scatter is reproducible bit for bit up to the final float adds of the four replicas.
__global__ void __launch_bounds__(1024) k_bin_scan(BinHeader *__restrict__ hd_all, uint32_t n_wg) {
    BinHeader *__restrict__ hd = hd_all + blockIdx.x;
    __shared__ uint32_t seg[16][kBinSlices], start[kBinSlices];
    const uint32_t s = threadIdx.x & (kBinSlices - 1), g = threadIdx.x >> 6;   // 16 segments of 16 workgroups
    uint32_t c[16], sum = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const uint32_t w = g * 16 + i;
        c[i] = w < n_wg ? hd->wg[w][s] : 0u;
        sum += c[i];
    }
    seg[g][s] = sum;
    __syncthreads();
    if (threadIdx.x < kBinSlices) {
        uint32_t tot = 0;
        for (int k = 0; k < 16; ++k) tot += seg[k][threadIdx.x];
        hd->counts[threadIdx.x] = tot;
        start[threadIdx.x] = tot;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (uint32_t k = 0; k < kBinSlices; ++k) {
            const uint32_t t = start[k];
            start[k] = run;
            run += t;
        }
    }
    __syncthreads();
    uint32_t off = start[s];
    for (uint32_t k = 0; k < g; ++k) off += seg[k][s];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const uint32_t w = g * 16 + i;
        if (w < n_wg) hd->wg[w][s] = off;
        off += c[i];
    }
}

// Pass 2: queue[offset of (w, s) + rank] = entry
__global__ void __launch_bounds__(kBinThreads) k_bin_write(const float *__restrict__ rec_all, uint32_t M, const BinHeader *__restrict__ hd_all,
                                                           uint4 *__restrict__ queue_all) {
    const float *__restrict__ rec = rec_all + (size_t)blockIdx.y * M * 8;
    const BinHeader *__restrict__ hd = hd_all + blockIdx.y;
    uint4 *__restrict__ queue = queue_all + (size_t)blockIdx.y * 4 * M;
    __shared__ uint32_t h[kBinSlices], running[kBinSlices];
    if (threadIdx.x < kBinSlices) running[threadIdx.x] = hd->wg[blockIdx.x][threadIdx.x];
    const uint4 *__restrict__ rec4 = reinterpret_cast<const uint4 *>(rec);
    for (uint32_t m0 = blockIdx.x * kBinThreads; m0 < M; m0 += gridDim.x * kBinThreads) {   // uniform trip count: barriers inside
        if (threadIdx.x < kBinSlices) h[threadIdx.x] = 0;
        __syncthreads();
        const uint32_t m = m0 + threadIdx.x;
        uint32_t sl[4], local[4];
        float gabs;
        const bool live = m < M && record_pairs(rec4, m, sl, gabs);
        if (live)
#pragma unroll
            for (int q = 0; q < 4; ++q) local[q] = atomicAdd(&h[sl[q]], 1u);
        __syncthreads();
        if (live) {
            const uint4 ra = rec4[2 * (size_t)m], rb = rec4[2 * (size_t)m + 1];
            const float wy = __uint_as_float(ra.w), wz = __uint_as_float(rb.x);
            const float g0 = __uint_as_float(rb.y), g1 = __uint_as_float(rb.z);
#pragma unroll
            for (uint32_t q = 0; q < 4; ++q)
                queue[running[sl[q]] + local[q]] = pair_entry(ra.x & 0xffffu, pair_hash(ra.x >> 16, ra.y, q), __uint_as_float(ra.z), wy, wz, g0, g1, q);
        }
        __syncthreads();
        if (threadIdx.x < kBinSlices) running[threadIdx.x] += h[threadIdx.x];
    }
}

// ---- planned variant: where an entry goes depends on the point's position only, so the count, the scan and the destinations
// are computed from xyzs while the forward pass is still running (off the critical path, on a side stream), and k_field_bwd
// writes its gradients straight into the queue: no 32-byte record round trip and no binning passes between the MLP backward
// and the owners.  Every point gets its four entries (a zero gradient adds zero).
__device__ inline void point_slices(const float *__restrict__ xyzs, uint32_t m, float bound, uint32_t (&sl)[4]) {
    const float two_b = 2.0f * bound;
    uint32_t iy, iz;
    float w;
    codebook_axis((xyzs[3 * (size_t)m + 1] + bound) / two_b, iy, w);
    codebook_axis((xyzs[3 * (size_t)m + 2] + bound) / two_b, iz, w);
#pragma unroll
    for (uint32_t q = 0; q < 4; ++q) sl[q] = pair_slice(pair_hash(iy, iz, q));
}

__global__ void __launch_bounds__(kBinThreads) k_plan_count(const float *__restrict__ xyzs, uint32_t M, float bound, BinHeader *__restrict__ hd) {
    __shared__ uint32_t h[kBinSlices];
    if (threadIdx.x < kBinSlices) h[threadIdx.x] = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) hd->gmax_bits = 0;   // k_field_bwd raises it (stream-ordered after the plan)
    __syncthreads();
    for (uint32_t m = blockIdx.x * kBinThreads + threadIdx.x; m < M; m += gridDim.x * kBinThreads) {
        uint32_t sl[4];
        point_slices(xyzs, m, bound, sl);
#pragma unroll
        for (int q = 0; q < 4; ++q) atomicAdd(&h[sl[q]], 1u);
    }
    __syncthreads();
    if (threadIdx.x < kBinSlices) hd->wg[blockIdx.x][threadIdx.x] = h[threadIdx.x];
}

// Every workgroup derives its own write offsets from the count table (wg[w][s], 64 KiB at most, L2-resident): slice starts = exclusive
// prefix over the slice totals, plus what the workgroups in front of it found for that slice -- the single-workgroup scan launch that
// used to sit between the two passes (k_bin_scan: 2-16 us in a chain of small launches) is gone.  Workgroup 0 stores the slice totals
// the owners read.  Same queue order as count -> scan -> dest.
__global__ void __launch_bounds__(kBinThreads) k_plan_dest(const float *__restrict__ xyzs, uint32_t M, float bound, BinHeader *__restrict__ hd,
                                                           uint4 *__restrict__ dest) {
    __shared__ uint32_t h[kBinSlices], running[kBinSlices], seg_tot[16][kBinSlices], seg_before[16][kBinSlices];
    {
        static_assert(kBinThreads == 16 * kBinSlices && kBinGrid == 256, "16 segments of 16 workgroups, one thread per (segment, slice)");
        const uint32_t s = threadIdx.x & (kBinSlices - 1), g = threadIdx.x >> 6, n_wg = gridDim.x;
        uint32_t tot = 0, before = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const uint32_t w = g * 16 + i;
            const uint32_t c = w < n_wg ? hd->wg[w][s] : 0u;
            tot += c;
            if (w < blockIdx.x) before += c;
        }
        seg_tot[g][s] = tot;
        seg_before[g][s] = before;
        __syncthreads();
        if (threadIdx.x < kBinSlices) {      // one wave: slice totals, their exclusive prefix, this workgroup's offset inside each slice
            uint32_t t = 0, b = 0;
#pragma unroll
            for (int k = 0; k < 16; ++k) { t += seg_tot[k][threadIdx.x]; b += seg_before[k][threadIdx.x]; }
            uint32_t incl = t;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t v = (uint32_t)__shfl_up((int)incl, d, 64);
                if ((int)threadIdx.x >= d) incl += v;
            }
            running[threadIdx.x] = incl - t + b;
            if (blockIdx.x == 0) hd->counts[threadIdx.x] = t;
        }
    }
    for (uint32_t m0 = blockIdx.x * kBinThreads; m0 < M; m0 += gridDim.x * kBinThreads) {   // uniform trip count: barriers inside
        if (threadIdx.x < kBinSlices) h[threadIdx.x] = 0;
        __syncthreads();
        const uint32_t m = m0 + threadIdx.x;
        uint32_t sl[4], local[4];
        if (m < M) {
            point_slices(xyzs, m, bound, sl);
#pragma unroll
            for (int q = 0; q < 4; ++q) local[q] = atomicAdd(&h[sl[q]], 1u);
        }
        __syncthreads();
        if (m < M) dest[m] = make_uint4(running[sl[0]] + local[0], running[sl[1]] + local[1], running[sl[2]] + local[2], running[sl[3]] + local[3]);
        __syncthreads();
        if (threadIdx.x < kBinSlices) running[threadIdx.x] += h[threadIdx.x];
    }
}

// ---- the planned variant over the 16 base levels (stage-1 training: every base table has its own gradient).  blockIdx.y = level,
// blockIdx.x = a CHUNK of 1024 consecutive sample points (consecutive samples of a few rays).
// Where an entry goes depends on the sample positions and the level's cell only, so the counting and the offsets are computed beside the
// forward pass (hg_levels_plan: k_levels_count -> k_levels_scan); behind the MLP backward one pass turns the feature gradients into queue
// entries (k_level_entries) and the 16 x 64 slice owners run (k_scatter_binned).  Three things differ from the codebook's planned scatter:
//   * the point count is a device value (`rows_dev`, may be null): a captured step sizes its launches for the buffers' capacity M and walks
//     only the rows the march produced;
//   * a chunk's entries are sorted by slice in LDS and leave as whole 128-byte lines (a run of a slice's entries is contiguous in the queue):
//     written straight from the lanes -- 64 lanes, 64 different lines per store -- the pass was bound by the L2's request rate (82 us for
//     8 M entries); the per-point destination array of the first version (32 MB written by the plan, read here) is gone: a chunk needs its
//     64 run offsets only;
//   * MERGED RUNS on the coarse levels.  A ray's consecutive samples share a cell there (level 0: ~37 samples per cell), i.e. the same 8 rows;
//     worse, a coarse level has few distinct (y, z) cell pairs, so the slices -- bits of hash(y, z) -- are loaded very unevenly and ONE owner
//     streamed 10-20 % of a level's entries (87 us at level 0 against 12.5 us at the fine levels, profiles/r05_stage1_steps.txt).  For levels
//     < kMergeLevels the wave adds up each run of lanes in the same cell first (16 corner-feature sums, fp32, fixed lane order) and only the
//     run's first lane emits entries: two per (dy, dz) pair, one for each x side (x weight 0 and 1: the owner's (1 - wx, wx) split then routes
//     the sum to one row).  Level 0 shrinks 18-fold.  The sums differ from the unmerged ones by fp32 rounding of the partial sums (1e-7).
constexpr uint32_t kMergeLevels = 10;        // resolutions 16..294: >= 2 samples per cell along a ray at the bench step length
constexpr uint32_t kLevelQueueStride = 8;    // queue entries reserved per point and level (a merged level whose every point is its own run emits 8)
constexpr uint32_t kLevelStage = 4 * kBinThreads;   // entries a chunk sorts in LDS (64 KiB: two workgroups per compute unit, or one beside the weight-gradient kernel);
                                                    // a merged chunk with more (runs shorter than two samples on average) stores its entries straight from the lanes

struct LevelsPlan {
    BinHeader *hd;          // [16]
    uint32_t *chunk_tab;    // [16][n_chunks][64]: entries of slice s in chunk c, then (k_levels_scan) the queue offset of that run
    uint32_t *chunk_max;    // [16][n_chunks]: max |contribution| among a chunk's entries, as a float bit pattern (0 for chunks past the live rows)
    uint4 *queue;           // [16][8 M]
    uint32_t n_chunks;
};
static size_t levels_plan_bytes(uint32_t M) {
    const size_t n_chunks = ceil_div(M, kBinThreads);
    return (size_t)NSIG_BASE_LEVELS * (sizeof(BinHeader) + n_chunks * (kBinSlices + 4) * sizeof(uint32_t) + (size_t)kLevelQueueStride * M * sizeof(uint4));
}
static LevelsPlan levels_plan_view(void *plan, uint32_t M) {
    LevelsPlan v;
    v.n_chunks = ceil_div(M, kBinThreads);
    v.hd = reinterpret_cast<BinHeader *>(plan);
    v.chunk_tab = reinterpret_cast<uint32_t *>(v.hd + NSIG_BASE_LEVELS);
    v.chunk_max = v.chunk_tab + (size_t)NSIG_BASE_LEVELS * v.n_chunks * kBinSlices;
    v.queue = reinterpret_cast<uint4 *>(v.chunk_max + (size_t)NSIG_BASE_LEVELS * v.n_chunks * 4);     // (16-byte aligned: 64 + 4 words per chunk and level; a quarter of the second block is used)
    return v;
}

struct LevelPoint {
    uint32_t ix, iy, iz;
    float wx, wy, wz;
};
__device__ inline void level_point(const float *__restrict__ xyzs, uint32_t m, float bound, float cell, LevelPoint &p) {
    const float two_b = 2.0f * bound;
    axis_cell((xyzs[3 * (size_t)m] + bound) / two_b, cell, p.ix, p.wx);
    axis_cell((xyzs[3 * (size_t)m + 1] + bound) / two_b, cell, p.iy, p.wy);
    axis_cell((xyzs[3 * (size_t)m + 2] + bound) / two_b, cell, p.iz, p.wz);
}
// does this lane open a run?  (merged levels: first lane of a 16-lane row -- the runs are added up with DPP row shifts, which stay inside a
// row --, or another cell than the lane below; other levels: every live lane)
__device__ inline bool run_leader(bool merged, bool live, const LevelPoint &p) {
    if (!merged) return live;
    const uint32_t key = live ? (p.ix | (p.iy << 10) | (p.iz << 20)) : 0xffffffffu;      // resolutions < 1024 on the merged levels
    const uint32_t below = (uint32_t)__builtin_amdgcn_update_dpp((int)~key, (int)key, 0x111, 0xf, 0xf, false);      // row_shr:1 (lane 0 of a row keeps ~key)
    return live && key != below;
}

__global__ void __launch_bounds__(kBinThreads) k_levels_count(const float *__restrict__ xyzs, uint32_t M, const uint32_t *__restrict__ rows_dev, float bound,
                                                              LevelGeom geom, LevelsPlan pl) {
    const uint32_t level = blockIdx.y, chunk = blockIdx.x;
    const uint32_t n = rows_dev != nullptr ? min(M, *rows_dev) : M;
    if (chunk * kBinThreads >= n) return;
    __shared__ uint32_t h[kBinSlices];
    if (threadIdx.x < kBinSlices) h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t m = chunk * kBinThreads + threadIdx.x;
    const bool live = m < n, merged = level < kMergeLevels;
    LevelPoint p{};
    if (live) level_point(xyzs, m, bound, geom.cell[level], p);
    if (run_leader(merged, live, p)) {
#pragma unroll
        for (uint32_t q = 0; q < 4; ++q) atomicAdd(&h[pair_slice(pair_hash(p.iy, p.iz, q))], merged ? 2u : 1u);
    }
    __syncthreads();
    if (threadIdx.x < kBinSlices) pl.chunk_tab[((size_t)level * pl.n_chunks + chunk) * kBinSlices + threadIdx.x] = h[threadIdx.x];
}

// One workgroup per level: slice totals (what the owners read), slice starts, and -- in place of the counts -- the queue offset of every
// (chunk, slice) run: the queue is slice-major, a slice's runs in chunk order.  16 segments of chunks x 64 slices = 1024 threads.
__global__ void __launch_bounds__(kBinThreads) k_levels_scan(uint32_t M, const uint32_t *__restrict__ rows_dev, LevelsPlan pl) {
    const uint32_t level = blockIdx.x;
    const uint32_t n = rows_dev != nullptr ? min(M, *rows_dev) : M;
    const uint32_t chunks = ceil_div(n, kBinThreads), per = ceil_div(chunks, 16u);
    uint32_t *__restrict__ tab = pl.chunk_tab + (size_t)level * pl.n_chunks * kBinSlices;
    __shared__ uint32_t seg[16][kBinSlices], start[kBinSlices];
    const uint32_t s = threadIdx.x & (kBinSlices - 1), g = threadIdx.x >> 6;
    const uint32_t c0 = min(chunks, g * per), c1 = min(chunks, c0 + per);
    uint32_t sum = 0;
    for (uint32_t c = c0; c < c1; ++c) sum += tab[(size_t)c * kBinSlices + s];
    seg[g][s] = sum;
    __syncthreads();
    if (threadIdx.x < kBinSlices) {      // one wave: totals and their exclusive prefix
        uint32_t t = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += seg[k][threadIdx.x];
        uint32_t incl = t;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t v = (uint32_t)__shfl_up((int)incl, d, 64);
            if ((int)threadIdx.x >= d) incl += v;
        }
        start[threadIdx.x] = incl - t;
        pl.hd[level].counts[threadIdx.x] = t;
    }
    __syncthreads();
    uint32_t off = start[s];
    for (uint32_t k = 0; k < g; ++k) off += seg[k][s];
    for (uint32_t c = c0; c < c1; ++c) {
        const uint32_t t = tab[(size_t)c * kBinSlices + s];
        tab[(size_t)c * kBinSlices + s] = off;
        off += t;
    }
}

// sum over the run of lanes [lane, run_end] of this 16-lane row, delivered to `lane` (meaningful at a run's first lane): four row shifts, no LDS
template <int D>
__device__ inline float row_shl(float v) {      // lane i reads lane i + D of its row (0 past the row's end)
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x100 + D, 0xf, 0xf, true));
}
// the maximum over the wave's 64 lanes (all active), the same value in every lane: four row shifts, then the four rows' first lanes
__device__ inline uint32_t wave_max_u32(uint32_t v) {
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x101, 0xf, 0xf, true));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x102, 0xf, 0xf, true));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x104, 0xf, 0xf, true));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x108, 0xf, 0xf, true));
    return max(max((uint32_t)__builtin_amdgcn_readlane((int)v, 0), (uint32_t)__builtin_amdgcn_readlane((int)v, 16)),
               max((uint32_t)__builtin_amdgcn_readlane((int)v, 32), (uint32_t)__builtin_amdgcn_readlane((int)v, 48)));
}
__device__ inline float run_sum(float v, uint32_t l16, uint32_t run_end) {
    float o = row_shl<1>(v);
    v += l16 + 1u <= run_end ? o : 0.0f;
    o = row_shl<2>(v);
    v += l16 + 2u <= run_end ? o : 0.0f;
    o = row_shl<4>(v);
    v += l16 + 4u <= run_end ? o : 0.0f;
    o = row_shl<8>(v);
    v += l16 + 8u <= run_end ? o : 0.0f;
    return v;
}

#ifdef NSIG_ENT_TIMING      // diagnostic build: where a workgroup of k_level_entries spends its time (10 ns ticks, summed over workgroups, per level; [level][8] = workgroups counted)
__device__ unsigned long long g_ent_phase[16][9];
#define ENT_STAMP(k)                                                                                          \
    do {                                                                                                      \
        if (threadIdx.x == 0) {                                                                               \
            const unsigned long long now_ = wall_clock64();                                                   \
            atomicAdd(&g_ent_phase[level][k], now_ - ent_t_);                                                  \
            if ((k) == 7) atomicAdd(&g_ent_phase[level][8], 1ull);                                             \
            ent_t_ = now_;                                                                                    \
        }                                                                                                     \
    } while (0)
#else
#define ENT_STAMP(k)
#endif

// feature gradients of a chunk's points at one level -> queue entries, sorted by slice in LDS, stored as contiguous runs; raises the level's
// max |contribution| (the owners' fixed-point scale)
__global__ void __launch_bounds__(kBinThreads) k_level_entries(const float *__restrict__ xyzs, uint32_t M, const uint32_t *__restrict__ rows_dev, float bound,
                                                               const float2 *__restrict__ dplanes, uint32_t stride, LevelGeom geom, LevelsPlan pl) {
    extern __shared__ __attribute__((aligned(16))) uint4 staged[];      // [kLevelStage]
    __shared__ uint32_t h[kBinSlices], base[kBinSlices + 1], roff[kBinSlices], wg_max;
    __shared__ uint8_t slice_of[kLevelStage];      // which slice's run a staged entry belongs to
    const uint32_t level = blockIdx.y, chunk = blockIdx.x;
#ifdef NSIG_ENT_TIMING
    unsigned long long ent_t_ = wall_clock64();
#endif
    // every input requested before the row count is known (rows past it are inside the buffers, their values unused): one memory latency at the
    // head of the workgroup instead of three in a chain (count -> positions -> gradients); two 1024-thread workgroups per unit hide little
    const uint32_t m = chunk * kBinThreads + threadIdx.x, lane = threadIdx.x & 63u, mc = min(m, M - 1u);
    const float px = xyzs[3 * (size_t)mc], py = xyzs[3 * (size_t)mc + 1], pz = xyzs[3 * (size_t)mc + 2];
    float2 g = dplanes[(size_t)level * stride + mc];
    const uint32_t my_roff = threadIdx.x < kBinSlices ? pl.chunk_tab[((size_t)level * pl.n_chunks + chunk) * kBinSlices + threadIdx.x] : 0u;
    const uint32_t n = rows_dev != nullptr ? min(M, *rows_dev) : M;
    if (chunk * kBinThreads >= n) {
        if (threadIdx.x == 0) pl.chunk_max[(size_t)level * pl.n_chunks + chunk] = 0u;
        return;
    }
    if (threadIdx.x < kBinSlices) {
        h[threadIdx.x] = 0;
        roff[threadIdx.x] = my_roff;      // where this chunk's run of each slice starts in the queue
    }
    if (threadIdx.x == 0) wg_max = 0;
    __syncthreads();
    ENT_STAMP(0);      // inputs arrived (the row count, the run offsets)
    const bool live = m < n, merged = level < kMergeLevels;
    LevelPoint p{};
    if (live) {
        const float two_b = 2.0f * bound, cell = geom.cell[level];
        axis_cell((px + bound) / two_b, cell, p.ix, p.wx);
        axis_cell((py + bound) / two_b, cell, p.iy, p.wy);
        axis_cell((pz + bound) / two_b, cell, p.iz, p.wz);
    } else {
        g = make_float2(0.0f, 0.0f);
    }
    const bool leader = run_leader(merged, live, p);
    uint32_t sl[4], hyz[4], rank[4];
    if (leader) {
#pragma unroll
        for (uint32_t q = 0; q < 4; ++q) {
            hyz[q] = pair_hash(p.iy, p.iz, q);
            sl[q] = pair_slice(hyz[q]);
            rank[q] = atomicAdd(&h[sl[q]], merged ? 2u : 1u);
        }
    }
    ENT_STAMP(1);      // cells, leaders, ranks
    // merged levels: the 16 corner-feature contributions of a run, summed into its first lane
    float c[4][2][2];      // [(dy, dz) pair][x side][feature]
    uint32_t gb = max(__float_as_uint(g.x) & 0x7fffffffu, __float_as_uint(g.y) & 0x7fffffffu);
    if (merged) {
        const uint32_t l16 = lane & 15u;
        const uint32_t row_leaders = (uint32_t)(__ballot(leader || !live) >> (lane & 48u)) & 0xffffu;
        const uint32_t higher = row_leaders & ~((2u << l16) - 1u);
        const uint32_t run_end = higher ? (uint32_t)__ffs((int)higher) - 2u : 15u;
        gb = 0;
#pragma unroll
        for (uint32_t q = 0; q < 4; ++q) {
            const float fz = (q & 1u) ? p.wz : 1.0f - p.wz, fy = (q >> 1) ? p.wy : 1.0f - p.wy;
            const float a0 = (g.x * fz) * fy, a1 = (g.y * fz) * fy;      // corner_weight()'s order; zero on dead lanes
            c[q][0][0] = run_sum(a0 * (1.0f - p.wx), l16, run_end);
            c[q][0][1] = run_sum(a1 * (1.0f - p.wx), l16, run_end);
            c[q][1][0] = run_sum(a0 * p.wx, l16, run_end);
            c[q][1][1] = run_sum(a1 * p.wx, l16, run_end);
            if (leader)
#pragma unroll
                for (int t = 0; t < 4; ++t) gb = max(gb, __float_as_uint(c[q][t >> 1][t & 1]) & 0x7fffffffu);
        }
    }
    ENT_STAMP(2);      // run sums
    __syncthreads();
    ENT_STAMP(3);      // barrier
    if (threadIdx.x < kBinSlices) {      // one wave: where each slice's run starts in the staging area
        const uint32_t t = h[threadIdx.x];
        uint32_t incl = t;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t v = (uint32_t)__shfl_up((int)incl, d, 64);
            if ((int)threadIdx.x >= d) incl += v;
        }
        base[threadIdx.x] = incl - t;
        if (threadIdx.x == kBinSlices - 1) base[kBinSlices] = incl;
    }
    __syncthreads();
    ENT_STAMP(4);      // prefix + barrier
    const uint32_t total = base[kBinSlices];
    const bool direct = total > kLevelStage;      // (uniform; merged levels only)
    uint4 *__restrict__ queue = pl.queue + (size_t)level * kLevelQueueStride * M;
    if (leader) {
#pragma unroll
        for (uint32_t q = 0; q < 4; ++q) {
            uint4 e0, e1 = make_uint4(0u, 0u, 0u, 0u);
            if (merged) {
                const uint32_t key = p.ix | ((hyz[q] & (kBinRows - 1)) << 16);
                e0 = make_uint4(key, __float_as_uint(0.0f), __float_as_uint(c[q][0][0]), __float_as_uint(c[q][0][1]));
                e1 = make_uint4(key, __float_as_uint(1.0f), __float_as_uint(c[q][1][0]), __float_as_uint(c[q][1][1]));
            } else {
                e0 = pair_entry(p.ix, hyz[q], p.wx, p.wy, p.wz, g.x, g.y, q);
            }
            // (two spelled-out branches: one pointer that may address LDS or global memory becomes a FLAT store -- 50 us of this kernel's 66)
            if (direct) {
                uint4 *__restrict__ dst = queue + roff[sl[q]] + rank[q];
                dst[0] = e0;
                if (merged) dst[1] = e1;
            } else {
                const uint32_t at = base[sl[q]] + rank[q];
                staged[at] = e0;
                slice_of[at] = (uint8_t)sl[q];
                if (merged) {
                    staged[at + 1] = e1;
                    slice_of[at + 1] = (uint8_t)sl[q];
                }
            }
        }
    }
    ENT_STAMP(5);      // staging writes
    __syncthreads();
    ENT_STAMP(6);      // barrier
    // Copy-out: consecutive lanes take consecutive entries of a run = whole lines.  At most kLevelStage / kBinThreads = 4 entries per thread; all of a thread's LDS reads of
    // one kind are issued together (as a loop over j this was, per entry, two dependent LDS round trips in front of its store: the workgroup's last phase, 25-42 % of its
    // time -- and a workgroup's time is what this launch is made of: two workgroups per unit, ~9 us each, 19 rounds of them at 620 k points).  The empty asm statements
    // pin the reads in front of the tests that guard the stores (a read used only behind a test is sunk behind it).
    if (!direct && total != 0u) {      // (uniform)
        typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
        constexpr uint32_t kPer = kLevelStage / kBinThreads;
        uint32_t sj[kPer], ro[kPer], ba[kPer];
        u32x4_t e[kPer];
#pragma unroll
        for (uint32_t u = 0; u < kPer; ++u) {
            const uint32_t j = min(threadIdx.x + u * kBinThreads, total - 1u);
            sj[u] = slice_of[j];
            const uint4 t = staged[j];
            e[u] = u32x4_t{t.x, t.y, t.z, t.w};
        }
#pragma unroll
        for (uint32_t u = 0; u < kPer; ++u) asm volatile("" : "+v"(sj[u]));
#pragma unroll
        for (uint32_t u = 0; u < kPer; ++u) {
            ro[u] = roff[sj[u]];
            ba[u] = base[sj[u]];
        }
#pragma unroll
        for (uint32_t u = 0; u < kPer; ++u) asm volatile("" : "+v"(ro[u]), "+v"(ba[u]), "+v"(e[u]));
#pragma unroll
        for (uint32_t u = 0; u < kPer; ++u) {
            const uint32_t j = threadIdx.x + u * kBinThreads;
            // streamed out: a queue line is written here once and read once by an owner on some other XCD -- no L2 can serve it (the launch itself takes the same time; the
            // step is 1 % shorter for what the caches keep instead: same box, two rounds, 1.201 / 1.198 -> 1.189 / 1.187 ms with the owners' loads streamed as well)
            if (j < total) __builtin_nontemporal_store(e[u], reinterpret_cast<u32x4_t *>(queue + ro[u] + (j - ba[u])));
        }
    }
    // the chunk's max |contribution|: a plain store; the level's owners take the maximum over the chunks themselves.  (One word per level raised with
    // atomics by 31 k waves -- the codebook scatter's scheme, where 3 k waves do it -- was the largest single item of this kernel.)  The wave's maximum by
    // lane exchanges (left to the compiler, `if (gb) atomicMax(...)` becomes a scalar loop over the wave's lanes), one LDS operation per wave.
    gb = wave_max_u32(gb);
    if (lane == 0 && gb) atomicMax(&wg_max, gb);      // (LDS)
    __syncthreads();
    if (threadIdx.x == 0) pl.chunk_max[(size_t)level * pl.n_chunks + chunk] = wg_max;
    ENT_STAMP(7);      // copy-out issued, maximum raised
}

#ifdef NSIG_ENT_TIMING
NSIG_EXPORT int level_entries_phase_ticks(unsigned long long *out144, int reset) {
    if (hipMemcpyFromSymbol(out144, HIP_SYMBOL(g_ent_phase), sizeof(unsigned long long) * 144) != hipSuccess) return 1;
    if (reset) {
        unsigned long long zero[144] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_ent_phase), zero, sizeof(zero)) != hipSuccess) return 1;
    }
    return 0;
}
#endif

// (a single-precision formulation of this conversion -- t = c * 2^(k-27) split into its integer part and a 27-bit fraction, two exact cvt_i32 -- changes nothing,
// also with the owners' loads really in flight: 369-375 us against 365-373 on the 16-level scatter; the f64 instructions are ~5 us of the owners, LABNOTES 17a)
__device__ inline long long to_fixed(float c, int k) { return __double2ll_rn(ldexp((double)c, k)); }

// torch.optim.Adam's update of one element (the dense passes further down and the owners' fused form share it)
__device__ inline void adam_update(float g, float &p, float &m, float &v, float beta1, float beta2, float eps, float step_size, float inv_bc2_sqrt) {
    m = m + (1.0f - beta1) * (g - m);                 // exp_avg.lerp_(grad, 1 - beta1)
    v = v * beta2 + ((1.0f - beta2) * g) * g;         // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
    const float denom = sqrtf(v) * inv_bc2_sqrt + eps;  // (exp_avg_sq.sqrt() / bias_correction2_sqrt).add_(eps)
    p = p - step_size * (m / denom);                  // param.addcdiv_(exp_avg, denom, value=-step_size)
}

// The owners of hg_levels_scatter_adam finish with the optimiser step of their rows: a slice's gradient never leaves the chip (no 64 MiB of G written and read back,
// no table pass on the step's serial tail).  scratch: k_adam_dense_prepare's two scalars per table ([set] = lr / (1 - beta1^t), [kOwnerAdamMax + set] = 1 / sqrt(1 - beta2^t)).
constexpr int kOwnerAdamMax = 32;      // (= kDenseMax, the prepare kernel's scratch layout)
struct OwnerAdam {
    float *p[NSIG_BASE_LEVELS], *m[NSIG_BASE_LEVELS], *v[NSIG_BASE_LEVELS];
    const float *scratch;
    float beta1, beta2, eps, grad_scale;
    uint32_t on;
};

// blockIdx.x = slice * replicas + replica: the slice's entries, split evenly over the replicas.  replicas == 1: the owner is
// alone and stores its rows (no atomics, no zero-fill of the table, bit-reproducible).  replicas > 1 (`slabs`: hashgrid.h, "Replicas ... merge EXACTLY"):
// every replica leaves its fixed-point accumulators as a slab and k_scatter_merge, the next launch, sums a slice's slabs as integers, converts once and adds
// ONE float per element to G -- the slice's sum is that of a single owner bit for bit, in whatever order the replicas (or the entries) ran; G itself
// accumulates over the launches of a step (block render, content render: a + b is commutative, so their order is immaterial too).  This is the
// determinism of the reference's embedding_dense_backward (hash_encoding_wtmk_bit.py:99-116 -> a sort + segmented reduce).  (Until round 6 the replicas
// each converted their partial sum and merged with float atomics: G moved by half an ulp from run to run, and Adam(eps = 1e-15) amplified that.  A merge
// inside this launch -- slabs, a ticket per slice, the last arriver sums -- needs an agent-scope release / acquire around the ticket, i.e. a write-back and an
// invalidate of the XCD's whole L2 per workgroup: 29 -> 150 us, profiles/r06_exact_merge_ab.txt.  A launch boundary gives the same visibility for one
// launch gap.)
// scale_by_count: the fixed-point scale leaves room for as many maximal contributions as the slice has entries (the base levels of stage 1: at
// level 0 a million samples share 4913 rows, far more than the 2^11 per row the codebook level's scale assumes).
// set_max (optional): [sets][n_set_max] partial maxima of |contribution| (float bit patterns) whose maximum replaces the header's gmax_bits.
__global__ void __launch_bounds__(1024) k_scatter_binned(const BinHeader *__restrict__ hd_all, const uint4 *__restrict__ queue_all, uint32_t M,
                                                         ScatterTargets tg, uint32_t replicas, uint32_t scale_by_count = 0,
                                                         const uint32_t *__restrict__ set_max = nullptr, uint32_t n_set_max = 0, OwnerAdam adam = OwnerAdam{},
                                                         unsigned long long *__restrict__ slabs = nullptr) {
    extern __shared__ unsigned long long acc64[];  // [kBinRows][2] fixed point
    const BinHeader *__restrict__ hd = hd_all + blockIdx.y;
    const uint4 *__restrict__ queue = queue_all + (size_t)blockIdx.y * 4 * M;
    float *__restrict__ G = tg.g[blockIdx.y];
    const uint32_t slice = blockIdx.x / replicas, replica = blockIdx.x - slice * replicas;
    // An owner is alone on its compute unit (128 KiB of accumulators), so every latency at its head is exposed, and with few entries the head IS the owner (41 us of
    // the launch at 33 k points).  Order: the slice counts as ONE load per wave (was: up to 63 scalar loads, each waited for), then the first round of queue entries
    // and the chunk maxima REQUESTED, then the accumulators cleared and the maximum reduced while those are on their way.
    static_assert(kBinSlices == 64, "one slice count per lane");
    const uint32_t lane = threadIdx.x & 63u, cnt = hd->counts[lane];
    uint32_t before = lane < slice ? cnt : 0u;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) before += (uint32_t)__shfl_xor((int)before, d, 64);
    const uint32_t start = (uint32_t)__builtin_amdgcn_readfirstlane((int)before);
    const uint32_t n = (uint32_t)__builtin_amdgcn_readfirstlane(__shfl((int)cnt, (int)slice, 64)), chunk = ceil_div(n, replicas);
    const uint32_t beg = min(n, replica * chunk), end = min(n, beg + chunk);
    // The queue walk, software-pipelined: a round's four entries per thread are requested one round ahead, and nothing in the loop is conditional.  (The first
    // form -- four clamped loads, then `if (beyond the end) break;` in front of each entry's work -- compiled to load, s_waitcnt vmcnt(0), convert, four atomics,
    // next load: every load is used only behind its own test, so the compiler sank it there, and a thread had ONE entry in flight whatever the unroll depth; the
    // owners ran at the latency of ~40 dependent streaming loads.)  Requests beyond the end load the slice's last entry again; nothing is added for them.
    constexpr int kAhead = 4;      // (2 / 4 / 8 entries ahead per thread: 375 / 369 / 370 us on the 16-level scatter at 671 k points, same box)
    typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
    const u32x4_t *__restrict__ q = reinterpret_cast<const u32x4_t *>(queue + start);
    const uint32_t round = blockDim.x * kAhead;
    u32x4_t cur[kAhead] = {}, nxt[kAhead];
    if (beg < end) {      // (uniform)
#pragma unroll
        for (int u = 0; u < kAhead; ++u) cur[u] = __builtin_nontemporal_load(q + min(beg + u * blockDim.x + threadIdx.x, end - 1));
    }
    // |contribution| <= gmax < 2^E; 2^11 of them stay below 2^62 with k = 51 - E.  A non-finite gradient anywhere in the launch
    // (an overflowing scaled loss under torch's GradScaler) poisons every row of G with NaN, so that the scaler's inf check sees it
    // exactly as it sees the inf/NaN float sums of the reference's dense gradients and skips the step.
    uint32_t gb = hd->gmax_bits, mx = 0;
    __shared__ uint32_t smax;
    if (set_max != nullptr) {
        for (uint32_t i = threadIdx.x; i < n_set_max; i += blockDim.x) mx = max(mx, set_max[(size_t)blockIdx.y * n_set_max + i]);
        if (threadIdx.x == 0) smax = 0;
    }
    if (beg < end || adam.on)      // (uniform)
        for (uint32_t i = threadIdx.x; i < 2u * kBinRows; i += blockDim.x) acc64[i] = 0ull;
    __syncthreads();
    if (set_max != nullptr) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, d, 64));
        if (lane == 0 && mx) atomicMax(&smax, mx);
        __syncthreads();
        gb = smax;
    }
    const bool poisoned = gb >= 0x7f800000u;
    int E;
    frexpf(__uint_as_float(poisoned ? 0x3f800000u : gb), &E);
    int k = poisoned ? 0 : 51 - E;
    if (scale_by_count && !poisoned && n > 2048u) k = 62 - E - (32 - __builtin_clz(n));      // |sum| <= n * gmax < 2^(E + ceil(log2(n + 1))) stays below 2^62
    float *out = G + 2 * (size_t)slice * kBinRows;
    if (beg >= end && !adam.on) {      // (uniform) nothing for this owner (two of level 0's 64 slices hold no cell pair at all; a replica beyond a short slice's end -- it leaves no slab, and the merge knows): its rows are zeros, no accumulators needed
        if (poisoned || replicas == 1) {
            const float f = poisoned ? __uint_as_float(0x7fc00000u) : 0.0f;
            for (uint32_t i4 = threadIdx.x; i4 < kBinRows / 2u; i4 += blockDim.x) *reinterpret_cast<float4 *>(out + 4u * i4) = make_float4(f, f, f, f);
        }
        return;
    }
    if (beg < end) {      // (uniform; an owner with the optimiser step inside comes here without entries too)
        for (uint32_t base = beg; base < end; base += round) {      // (uniform trip count)
#pragma unroll
            for (int u = 0; u < kAhead; ++u) nxt[u] = __builtin_nontemporal_load(q + min(base + round + u * blockDim.x + threadIdx.x, end - 1));
            __builtin_amdgcn_sched_barrier(0);      // (the requests stay in front of the round's work)
#pragma unroll
            for (int u = 0; u < kAhead; ++u) {
                if (base + u * blockDim.x + threadIdx.x < end) {      // (the requests are behind us: this test guards arithmetic and atomics only)
                    const uint32_t ix = cur[u][0] & 0xffffu, hyz = cur[u][0] >> 16;     // hyz: the pair's 13 row bits
                    const float wx = __uint_as_float(cur[u][1]), a0 = __uint_as_float(cur[u][2]), a1 = __uint_as_float(cur[u][3]);
                    unsigned long long *d0 = acc64 + 2u * ((ix ^ hyz) & (kBinRows - 1)), *d1 = acc64 + 2u * (((ix + 1u) ^ hyz) & (kBinRows - 1));
                    // (an entry of a merged level carries ONE x side -- wx is exactly 0 or 1 -- and the other side's two adds would add zeros)
                    if (wx != 1.0f) {
                        atomicAdd(d0, (unsigned long long)to_fixed(a0 * (1.0f - wx), k));
                        atomicAdd(d0 + 1, (unsigned long long)to_fixed(a1 * (1.0f - wx), k));
                    }
                    if (wx != 0.0f) {
                        atomicAdd(d1, (unsigned long long)to_fixed(a0 * wx, k));
                        atomicAdd(d1 + 1, (unsigned long long)to_fixed(a1 * wx, k));
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < kAhead; ++u) cur[u] = nxt[u];
        }
    }
    __syncthreads();
    if (adam.on) {      // (uniform; replicas == 1) the optimiser step of this slice's rows: parameters and moments in, the gradient from LDS, parameters and moments out
        constexpr uint32_t kTrips = kBinRows / 2u / 1024u;      // float4 (two rows) per thread: the launch has 1024 threads
        typedef float f32x4_t __attribute__((ext_vector_type(4)));
        const size_t first = (size_t)slice * (kBinRows / 2u);
        f32x4_t *pp = reinterpret_cast<f32x4_t *>(adam.p[blockIdx.y]) + first, *pm = reinterpret_cast<f32x4_t *>(adam.m[blockIdx.y]) + first,
                *pv = reinterpret_cast<f32x4_t *>(adam.v[blockIdx.y]) + first;
        const float ss = adam.scratch[blockIdx.y], ib = adam.scratch[kOwnerAdamMax + blockIdx.y];
        f32x4_t p[kTrips], m[kTrips], v[kTrips];
#pragma unroll
        for (uint32_t u = 0; u < kTrips; ++u) {      // all twelve requests first (the moments stream; the parameters were gathered from by this step's encoder)
            const uint32_t i4 = threadIdx.x + u * 1024u;
            p[u] = pp[i4];
            m[u] = __builtin_nontemporal_load(pm + i4);
            v[u] = __builtin_nontemporal_load(pv + i4);
        }
#pragma unroll
        for (uint32_t u = 0; u < kTrips; ++u) {
            const uint32_t i4 = threadIdx.x + u * 1024u;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const long long a = (long long)acc64[4u * i4 + c];
                const float g = (poisoned ? __uint_as_float(0x7fc00000u) : (a != 0 ? (float)ldexp((double)a, -k) : 0.0f)) * adam.grad_scale;      // (what the plain owner stores, times opt_adam_dense's factor)
                float pe = p[u][c], me = m[u][c], ve = v[u][c];
                adam_update(g, pe, me, ve, adam.beta1, adam.beta2, adam.eps, ss, ib);
                p[u][c] = pe; m[u][c] = me; v[u][c] = ve;
            }
            pp[i4] = p[u];      // (through the caches: the next step's encoder gathers from them; the moments stream out -- k_adam_dense_v4's policy)
            __builtin_nontemporal_store(m[u], pm + i4);
            __builtin_nontemporal_store(v[u], pv + i4);
        }
        return;
    }
    if (poisoned || replicas == 1) {
        for (uint32_t i4 = threadIdx.x; i4 < kBinRows / 2u; i4 += blockDim.x) {      // two rows = four sums per thread and trip: 16-byte stores (the slice is 64 KiB-aligned in G)
            float r[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const long long v = (long long)acc64[4u * i4 + c];
                r[c] = poisoned ? __uint_as_float(0x7fc00000u) : (v != 0 ? (float)ldexp((double)v, -k) : 0.0f);
            }
            *reinterpret_cast<float4 *>(out + 4u * i4) = make_float4(r[0], r[1], r[2], r[3]);
        }
    } else {
        // this replica's slab for k_scatter_merge (plain 16-byte stores; the launch boundary makes them visible)
        typedef unsigned long long u64x2_t __attribute__((ext_vector_type(2)));
        constexpr uint32_t kSlab = 2u * kBinRows;
        u64x2_t *mine = reinterpret_cast<u64x2_t *>(slabs + ((size_t)blockIdx.y * kBinSlices * replicas + (size_t)slice * replicas + replica) * kSlab);
        for (uint32_t i2 = threadIdx.x; i2 < kBinRows; i2 += blockDim.x) mine[i2] = *reinterpret_cast<const u64x2_t *>(acc64 + 2u * i2);
    }
}

// The exact merge of a slice's replicas (see k_scatter_binned).  blockIdx.x = slice * kMergeParts + part: a quarter of the slice's words per workgroup, one 8-byte
// fixed-point sum per thread and trip -- consecutive lanes on consecutive words, for the loads and for the atomics behind them (an atomic wave instruction leaves
// the L2 as one request per 64 bytes it touches; a row pair per lane would double them).  The scale is the one the replicas used (the launch's gmax); a poisoned
// launch has written its NaNs already.  One float atomic per non-zero element and launch.
constexpr uint32_t kMergeParts = 4, kMergeThreads = 1024;
__global__ void __launch_bounds__(kMergeThreads) k_scatter_merge(const BinHeader *__restrict__ hd_all, ScatterTargets tg, uint32_t replicas,
                                                                 const unsigned long long *__restrict__ slabs) {
    constexpr uint32_t kSlab = 2u * kBinRows, kPartWords = kSlab / kMergeParts, kTrips = kPartWords / kMergeThreads;
    static_assert(kPartWords % kMergeThreads == 0, "whole trips");
    const BinHeader *__restrict__ hd = hd_all + blockIdx.y;
    const uint32_t slice = blockIdx.x / kMergeParts, part = blockIdx.x - slice * kMergeParts;
    const uint32_t n = hd->counts[slice], gb = hd->gmax_bits;
    if (n == 0 || gb >= 0x7f800000u) return;      // (uniform) an empty slice; a poisoned launch
    int E;
    frexpf(__uint_as_float(gb), &E);
    const int k = 51 - E;
    const uint32_t chunk = ceil_div(n, replicas);
    const unsigned long long *slice_slabs = slabs + ((size_t)blockIdx.y * kBinSlices + slice) * replicas * kSlab + (size_t)part * kPartWords;
    unsigned long long sum[kTrips];
#pragma unroll
    for (uint32_t u = 0; u < kTrips; ++u) sum[u] = 0ull;
    for (uint32_t r = 0; r < replicas; ++r) {
        if (min(n, r * chunk) >= n) break;      // (uniform) this replica and the ones behind it had no entries and left no slab
        unsigned long long got[kTrips];
#pragma unroll
        for (uint32_t u = 0; u < kTrips; ++u) got[u] = __builtin_nontemporal_load(slice_slabs + (size_t)r * kSlab + threadIdx.x + u * kMergeThreads);
#pragma unroll
        for (uint32_t u = 0; u < kTrips; ++u) sum[u] += got[u];
    }
    float *out = tg.g[blockIdx.y] + 2 * (size_t)slice * kBinRows + (size_t)part * kPartWords;
#pragma unroll
    for (uint32_t u = 0; u < kTrips; ++u) {
        const long long v = (long long)sum[u];
        if (v != 0) atomicAdd(out + threadIdx.x + u * kMergeThreads, (float)ldexp((double)v, -k));
    }
}

// Stage-1 training scatters into all 16 base tables.  The cell index and weights of a (point, level) need the level's IEEE
// divisions (corner_rows), which the 32 slice owners of that level would each repeat: instead they are computed once here, into
// the same 32-byte record the codebook scatter consumes, and k_scatter_sliced runs over 16 record sets.
__global__ void __launch_bounds__(256) k_level_records(const float *__restrict__ xyzs, float bound, const float2 *__restrict__ dplanes, uint32_t M,
                                                       uint32_t stride, LevelGeom geom, uint4 *__restrict__ rec) {
    const uint32_t m = blockIdx.x * 256 + threadIdx.x, level = blockIdx.y;
    if (m >= M) return;
    const float2 g = dplanes[(size_t)level * stride + m];
    const float two_b = 2.0f * bound, cell = geom.cell[level];
    uint32_t idx[3];
    float w[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) axis_cell((xyzs[3 * (size_t)m + a] + bound) / two_b, cell, idx[a], w[a]);
    uint4 *r4 = rec + 2 * ((size_t)level * M + m);
    r4[0] = make_uint4(idx[0] | (idx[1] << 16), idx[2], __float_as_uint(w[0]), __float_as_uint(w[1]));
    r4[1] = make_uint4(__float_as_uint(w[2]), __float_as_uint(g.x), __float_as_uint(g.y), 0u);
}

// grads[i][e] (+)= G[e]: float4 per lane, D output streams.
__global__ void __launch_bounds__(256) k_fanout(const float4 *__restrict__ G, GradPtrs grads, uint32_t D, int accumulate) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= NSIG_TABLE_ROWS / 2) return;
    const float4 g = G[e];
    for (uint32_t i = 0; i < D; ++i) {
        float4 *dst = reinterpret_cast<float4 *>(grads.p[i]) + e;
        if (accumulate) {
            float4 v = *dst;
            v.x += g.x; v.y += g.y; v.z += g.z; v.w += g.w;
            *dst = v;
        } else *dst = g;
    }
}

__global__ void __launch_bounds__(256) k_level_lookup(const float *__restrict__ x01, uint32_t M, float cell, int32_t *__restrict__ rows,
                                                      float *__restrict__ weights) {
    const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    Corner8 c;
    corner_rows(x01[3 * (size_t)m], x01[3 * (size_t)m + 1], x01[3 * (size_t)m + 2], cell, c);
#pragma unroll
    for (int k = 0; k < 8; ++k) rows[8 * (size_t)m + k] = (int32_t)c.row[k];
    weights[3 * (size_t)m] = c.wx; weights[3 * (size_t)m + 1] = c.wy; weights[3 * (size_t)m + 2] = c.wz;
}

// Fused Adam over the D selected tables (shared gradient): float4 per lane, D x (param, exp_avg, exp_avg_sq) streams.
struct AdamPtrs {
    float *p[NSIG_MAX_MESSAGE_DIM];
    float *m[NSIG_MAX_MESSAGE_DIM];
    float *v[NSIG_MAX_MESSAGE_DIM];
    float step_size[NSIG_MAX_MESSAGE_DIM];
    float inv_bc2_sqrt[NSIG_MAX_MESSAGE_DIM];
};

__global__ void __launch_bounds__(256) k_codebook_adam(const float4 *__restrict__ G, AdamPtrs a, uint32_t D, float beta1, float beta2, float eps,
                                                       float grad_scale) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= NSIG_TABLE_ROWS / 2) return;
    float4 g = G[e];
    g.x *= grad_scale; g.y *= grad_scale; g.z *= grad_scale; g.w *= grad_scale;
    for (uint32_t i = 0; i < D; ++i) {
        float4 *pp = reinterpret_cast<float4 *>(a.p[i]) + e, *pm = reinterpret_cast<float4 *>(a.m[i]) + e, *pv = reinterpret_cast<float4 *>(a.v[i]) + e;
        float4 p = *pp, m = *pm, v = *pv;
        const float ss = a.step_size[i], ib = a.inv_bc2_sqrt[i];
        adam_update(g.x, p.x, m.x, v.x, beta1, beta2, eps, ss, ib);
        adam_update(g.y, p.y, m.y, v.y, beta1, beta2, eps, ss, ib);
        adam_update(g.z, p.z, m.z, v.z, beta1, beta2, eps, ss, ib);
        adam_update(g.w, p.w, m.w, v.w, beta1, beta2, eps, ss, ib);
        *pp = p; *pm = m; *pv = v;
    }
}

// ----------------------------------------------------------------------------- device-side table selection
// (every launch argument independent of the message: the enclosing step can be captured in a hipGraph)

struct PairPtrs {
    const float *p[2 * NSIG_MAX_MESSAGE_DIM];
};

__global__ void __launch_bounds__(256) k_codebook_presum_sel(PairPtrs tabs, const float *__restrict__ message, uint32_t D, float4 *__restrict__ S) {
    // Resolve the D selected tables once per workgroup (a chain message -> pointer -> data per table would otherwise sit in
    // front of every load), then stream them eight at a time.
    __shared__ const float4 *sel[NSIG_MAX_MESSAGE_DIM];
    if (threadIdx.x < D) sel[threadIdx.x] = reinterpret_cast<const float4 *>(tabs.p[2 * threadIdx.x + (message[threadIdx.x] != 0.0f)]);
    __syncthreads();
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= NSIG_TABLE_ROWS / 2) return;
    float4 acc = {0.f, 0.f, 0.f, 0.f};
    uint32_t i = 0;
    for (; i + 8 <= D; i += 8) {   // the sum keeps the table order (bit-identical to the serial sum)
        float4 t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = sel[i + u][e];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            acc.x += t[u].x; acc.y += t[u].y; acc.z += t[u].z; acc.w += t[u].w;
        }
    }
    for (; i < D; ++i) {
        const float4 a = sel[i][e];
        acc.x += a.x; acc.y += a.y; acc.z += a.z; acc.w += a.w;
    }
    S[e] = acc;
}

struct AdamPairPtrs {
    float *p[2 * NSIG_MAX_MESSAGE_DIM];
    float *m[2 * NSIG_MAX_MESSAGE_DIM];
    float *v[2 * NSIG_MAX_MESSAGE_DIM];
};
struct StepPtrs {
    float *s[2 * NSIG_MAX_MESSAGE_DIM];
};

// one thread per bit: advance the selected table's step count and derive its bias-correction scalars
__global__ void k_adam_prepare(StepPtrs steps, const float *__restrict__ message, uint32_t D, const float *__restrict__ lr, float beta1, float beta2,
                               float *__restrict__ scratch) {
    const uint32_t i = threadIdx.x;
    if (i >= D) return;
    float *sp = steps.s[2 * i + (message[i] != 0.0f)];
    const float step = *sp + 1.0f;
    *sp = step;
    // beta^step as exp(step * log(beta)) in double: same value to ~1e-13 relative, a fraction of pow()'s latency in this tiny kernel
    scratch[i] = (float)((double)*lr / (1.0 - exp((double)step * log((double)beta1))));
    scratch[D + i] = (float)(1.0 / sqrt(1.0 - exp((double)step * log((double)beta2))));
}

// Streaming accesses of the optimiser pass: 836 MiB go through once per step; marked non-temporal so that they do not displace the
// base tables (64 MiB) from the L2 / Infinity Cache right before the next step's gather.
typedef float nsig_f32x4 __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ float4 ld4(const float4 *p) {
    if (!NT) return *p;
    const nsig_f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const nsig_f32x4 *>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
template <bool NT>
__device__ __forceinline__ void st4(float4 *p, const float4 &a) {
    if (!NT) {
        *p = a;
        return;
    }
    const nsig_f32x4 v = {a.x, a.y, a.z, a.w};
    __builtin_nontemporal_store(v, reinterpret_cast<nsig_f32x4 *>(p));
}

// NEXT: the same pass also produces the pre-summed codebook of the NEXT step's message, S_next = sum_i table[2i + next_i]: where the
// next bit equals the current one the freshly updated row is already in registers, otherwise the partner table's row is read
// (about D/2 extra 4 MiB streams: +9 % traffic) -- instead of a separate 128 MiB pre-sum pass at the head of the next step.
// The sum keeps the table order, so S_next is bit-identical to k_codebook_presum_sel's.
template <bool NEXT, bool NT>
__global__ void __launch_bounds__(256) k_codebook_adam_sel(const float4 *__restrict__ G, AdamPairPtrs a, const float *__restrict__ message,
                                                           const float *__restrict__ scratch, uint32_t D, float beta1, float beta2, float eps,
                                                           float grad_scale, const float *__restrict__ next_message, float4 *__restrict__ S_next) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= NSIG_TABLE_ROWS / 2) return;
    float4 g = ld4<NT>(G + e);
    g.x *= grad_scale; g.y *= grad_scale; g.z *= grad_scale; g.w *= grad_scale;
    float4 acc = {0.f, 0.f, 0.f, 0.f};
    const float4 zero = {0.f, 0.f, 0.f, 0.f};
    // two tables per trip: six 16-byte loads in flight per thread before the first dependent store
    uint32_t i = 0;
    for (; i + 2 <= D; i += 2) {
        const uint32_t j0 = 2 * i + (message[i] != 0.0f), j1 = 2 * i + 2 + (message[i + 1] != 0.0f);
        float4 *pp0 = reinterpret_cast<float4 *>(a.p[j0]) + e, *pm0 = reinterpret_cast<float4 *>(a.m[j0]) + e, *pv0 = reinterpret_cast<float4 *>(a.v[j0]) + e;
        float4 *pp1 = reinterpret_cast<float4 *>(a.p[j1]) + e, *pm1 = reinterpret_cast<float4 *>(a.m[j1]) + e, *pv1 = reinterpret_cast<float4 *>(a.v[j1]) + e;
        bool other0 = false, other1 = false;      // wave-uniform
        float4 o0 = zero, o1 = zero;
        if (NEXT) {
            other0 = (next_message[i] != 0.0f) != (message[i] != 0.0f);
            other1 = (next_message[i + 1] != 0.0f) != (message[i + 1] != 0.0f);
            if (other0) o0 = ld4<NT>(reinterpret_cast<const float4 *>(a.p[j0 ^ 1u]) + e);
            if (other1) o1 = ld4<NT>(reinterpret_cast<const float4 *>(a.p[j1 ^ 1u]) + e);
        }
        float4 p0 = ld4<NT>(pp0), m0 = ld4<NT>(pm0), v0 = ld4<NT>(pv0), p1 = ld4<NT>(pp1), m1 = ld4<NT>(pm1), v1 = ld4<NT>(pv1);
        const float ss0 = scratch[i], ib0 = scratch[D + i], ss1 = scratch[i + 1], ib1 = scratch[D + i + 1];
        adam_update(g.x, p0.x, m0.x, v0.x, beta1, beta2, eps, ss0, ib0);
        adam_update(g.y, p0.y, m0.y, v0.y, beta1, beta2, eps, ss0, ib0);
        adam_update(g.z, p0.z, m0.z, v0.z, beta1, beta2, eps, ss0, ib0);
        adam_update(g.w, p0.w, m0.w, v0.w, beta1, beta2, eps, ss0, ib0);
        adam_update(g.x, p1.x, m1.x, v1.x, beta1, beta2, eps, ss1, ib1);
        adam_update(g.y, p1.y, m1.y, v1.y, beta1, beta2, eps, ss1, ib1);
        adam_update(g.z, p1.z, m1.z, v1.z, beta1, beta2, eps, ss1, ib1);
        adam_update(g.w, p1.w, m1.w, v1.w, beta1, beta2, eps, ss1, ib1);
        st4<NT>(pp0, p0); st4<NT>(pm0, m0); st4<NT>(pv0, v0); st4<NT>(pp1, p1); st4<NT>(pm1, m1); st4<NT>(pv1, v1);
        if (NEXT) {
            const float4 c0 = other0 ? o0 : p0, c1 = other1 ? o1 : p1;
            acc.x += c0.x; acc.y += c0.y; acc.z += c0.z; acc.w += c0.w;
            acc.x += c1.x; acc.y += c1.y; acc.z += c1.z; acc.w += c1.w;
        }
    }
    for (; i < D; ++i) {
        const uint32_t j = 2 * i + (message[i] != 0.0f);
        float4 *pp = reinterpret_cast<float4 *>(a.p[j]) + e, *pm = reinterpret_cast<float4 *>(a.m[j]) + e, *pv = reinterpret_cast<float4 *>(a.v[j]) + e;
        bool other = false;
        float4 o = zero;
        if (NEXT) {
            other = (next_message[i] != 0.0f) != (message[i] != 0.0f);
            if (other) o = ld4<NT>(reinterpret_cast<const float4 *>(a.p[j ^ 1u]) + e);
        }
        float4 p = ld4<NT>(pp), m = ld4<NT>(pm), v = ld4<NT>(pv);
        const float ss = scratch[i], ib = scratch[D + i];
        adam_update(g.x, p.x, m.x, v.x, beta1, beta2, eps, ss, ib);
        adam_update(g.y, p.y, m.y, v.y, beta1, beta2, eps, ss, ib);
        adam_update(g.z, p.z, m.z, v.z, beta1, beta2, eps, ss, ib);
        adam_update(g.w, p.w, m.w, v.w, beta1, beta2, eps, ss, ib);
        st4<NT>(pp, p); st4<NT>(pm, m); st4<NT>(pv, v);
        if (NEXT) {
            const float4 c = other ? o : p;
            acc.x += c.x; acc.y += c.y; acc.z += c.z; acc.w += c.w;
        }
    }
    if (NEXT) S_next[e] = acc;
}

}  // namespace nsig

using namespace nsig;

static int aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

NSIG_EXPORT int hg_codebook_presum(const float *const *tables_host, uint32_t D, float *S, nsig_stream_t stream) {
    NSIG_REQUIRE(tables_host && S, "hg_codebook_presum: null pointer");
    NSIG_REQUIRE(D >= 1 && D <= NSIG_MAX_MESSAGE_DIM, "hg_codebook_presum: D=%u out of range [1,%d]", D, NSIG_MAX_MESSAGE_DIM);
    CodebookPtrs tabs{};
    for (uint32_t i = 0; i < D; ++i) {
        NSIG_REQUIRE(tables_host[i] && aligned16(tables_host[i]), "hg_codebook_presum: table %u is null or not 16-byte aligned", i);
        tabs.p[i] = tables_host[i];
    }
    NSIG_REQUIRE(aligned16(S), "hg_codebook_presum: S must be 16-byte aligned");
    k_codebook_presum<<<NSIG_TABLE_ROWS / 2 / 256, 256, 0, as_stream(stream)>>>(tabs, D, reinterpret_cast<float4 *>(S));
    return check_launch("hg_codebook_presum");
}

static int fill_base(const float *const *host, TablePtrs &base, const char *who) {
    NSIG_REQUIRE(host, "%s: null base table list", who);
    for (int l = 0; l < NSIG_BASE_LEVELS; ++l) {
        NSIG_REQUIRE(host[l] != nullptr, "%s: base table %d is null", who, l);
        base.p[l] = host[l];
    }
    return NSIG_OK;
}

NSIG_EXPORT int hg_encode_fwd(const float *x01, uint32_t M, const float *const *base_tables_host, const float *S, float *feat,
                              nsig_stream_t stream) {
    if (M == 0) return NSIG_OK;
    NSIG_REQUIRE(x01 && feat, "hg_encode_fwd: null pointer");
    TablePtrs base{};
    if (int e = fill_base(base_tables_host, base, "hg_encode_fwd")) return e;
    NSIG_REQUIRE(M <= (1u << 27), "hg_encode_fwd: M=%u too large", M);
    k_encode<<<ceil_div(M * 16u, 256), 256, 0, as_stream(stream)>>>(x01, M, base, make_level_geom(), S, feat);
    return check_launch("hg_encode_fwd");
}

NSIG_EXPORT int hg_codebook_encode_fwd(const float *x01, uint32_t M, const float *const *tables_host, uint32_t D, float *out,
                                       nsig_stream_t stream) {
    if (M == 0) return NSIG_OK;
    NSIG_REQUIRE(x01 && tables_host && out, "hg_codebook_encode_fwd: null pointer");
    NSIG_REQUIRE(D >= 1 && D <= NSIG_MAX_MESSAGE_DIM, "hg_codebook_encode_fwd: D=%u out of range", D);
    CodebookPtrs tabs{};
    for (uint32_t i = 0; i < D; ++i) {
        NSIG_REQUIRE(tables_host[i] != nullptr, "hg_codebook_encode_fwd: table %u is null", i);
        tabs.p[i] = tables_host[i];
    }
    k_codebook_encode<<<ceil_div(M, 256), 256, 0, as_stream(stream)>>>(x01, M, tabs, D, 1.0f / kCodebookResolution, out);
    return check_launch("hg_codebook_encode_fwd");
}

NSIG_EXPORT int hg_codebook_bwd(const float *x01, uint32_t M, const float *dfeat, float *G, nsig_stream_t stream) {
    if (M == 0) return NSIG_OK;
    NSIG_REQUIRE(x01 && dfeat && G, "hg_codebook_bwd: null pointer");
    NSIG_REQUIRE(M <= (1u << 27), "hg_codebook_bwd: M=%u too large", M);
    k_codebook_bwd<<<ceil_div(M * 16u, 256), 256, 0, as_stream(stream)>>>(x01, M, dfeat, 1.0f / kCodebookResolution, G);
    return check_launch("hg_codebook_bwd");
}

NSIG_EXPORT int hg_fanout_grad(const float *G, float *const *grads_host, uint32_t D, int accumulate, nsig_stream_t stream) {
    NSIG_REQUIRE(G && grads_host, "hg_fanout_grad: null pointer");
    NSIG_REQUIRE(D >= 1 && D <= NSIG_MAX_MESSAGE_DIM, "hg_fanout_grad: D=%u out of range", D);
    GradPtrs grads{};
    for (uint32_t i = 0; i < D; ++i) {
        NSIG_REQUIRE(grads_host[i] && aligned16(grads_host[i]), "hg_fanout_grad: gradient %u is null or not 16-byte aligned", i);
        grads.p[i] = grads_host[i];
    }
    NSIG_REQUIRE(aligned16(G), "hg_fanout_grad: G must be 16-byte aligned");
    k_fanout<<<NSIG_TABLE_ROWS / 2 / 256, 256, 0, as_stream(stream)>>>(reinterpret_cast<const float4 *>(G), grads, D, accumulate);
    return check_launch("hg_fanout_grad");
}

NSIG_EXPORT int hg_level_lookup(const float *x01, uint32_t M, float resolution, int32_t *rows, float *weights, nsig_stream_t stream) {
    if (M == 0) return NSIG_OK;
    NSIG_REQUIRE(x01 && rows && weights, "hg_level_lookup: null pointer");
    NSIG_REQUIRE(resolution >= 1.0f, "hg_level_lookup: resolution must be >= 1");
    k_level_lookup<<<ceil_div(M, 256), 256, 0, as_stream(stream)>>>(x01, M, 1.0f / resolution, rows, weights);
    return check_launch("hg_level_lookup");
}

NSIG_EXPORT int opt_codebook_adam(const float *G, float *const *params_host, float *const *exp_avg_host, float *const *exp_avg_sq_host,
                                  uint32_t D, float beta1, float beta2, float eps, const float *step_size_host,
                                  const float *inv_bc2_sqrt_host, float grad_scale, nsig_stream_t stream) {
    NSIG_REQUIRE(G && params_host && exp_avg_host && exp_avg_sq_host && step_size_host && inv_bc2_sqrt_host, "opt_codebook_adam: null pointer");
    NSIG_REQUIRE(D >= 1 && D <= NSIG_MAX_MESSAGE_DIM, "opt_codebook_adam: D=%u out of range", D);
    NSIG_REQUIRE(aligned16(G), "opt_codebook_adam: G must be 16-byte aligned");
    AdamPtrs a{};
    for (uint32_t i = 0; i < D; ++i) {
        NSIG_REQUIRE(params_host[i] && exp_avg_host[i] && exp_avg_sq_host[i], "opt_codebook_adam: table %u has a null pointer", i);
        NSIG_REQUIRE(aligned16(params_host[i]) && aligned16(exp_avg_host[i]) && aligned16(exp_avg_sq_host[i]), "opt_codebook_adam: table %u is not 16-byte aligned", i);
        a.p[i] = params_host[i]; a.m[i] = exp_avg_host[i]; a.v[i] = exp_avg_sq_host[i];
        a.step_size[i] = step_size_host[i]; a.inv_bc2_sqrt[i] = inv_bc2_sqrt_host[i];
    }
    k_codebook_adam<<<NSIG_TABLE_ROWS / 2 / 256, 256, 0, as_stream(stream)>>>(reinterpret_cast<const float4 *>(G), a, D, beta1, beta2, eps, grad_scale);
    return check_launch("opt_codebook_adam");
}

NSIG_EXPORT int hg_scatter_sliced(const float *rec, uint32_t M, float *G, nsig_stream_t stream) {
    if (M == 0) return NSIG_OK;
    NSIG_REQUIRE(rec && G, "hg_scatter_sliced: null pointer");
    static bool attr_set = false;
    const size_t lds = (size_t)kSliceRows * 2 * sizeof(float);
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_scatter_sliced), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            set_error("hg_scatter_sliced: cannot reserve %zu bytes of LDS", lds);
            return NSIG_ERR_LAUNCH;
        }
        attr_set = true;
    }
    ScatterTargets tg{};
    tg.g[0] = G;
    k_scatter_sliced<<<kSlices * kReplicas, 1024, lds, as_stream(stream)>>>(rec, M, tg, 1u, (uint32_t)kReplicas);
    return check_launch("hg_scatter_sliced");
}

NSIG_EXPORT int hg_codebook_presum_sel(const float *const *all_tables_host, const float *message, uint32_t D, float *S, nsig_stream_t stream) {
    NSIG_REQUIRE(all_tables_host && message && S, "hg_codebook_presum_sel: null pointer");
    NSIG_REQUIRE(D >= 1 && D <= NSIG_MAX_MESSAGE_DIM, "hg_codebook_presum_sel: D=%u out of range [1,%d]", D, NSIG_MAX_MESSAGE_DIM);
    PairPtrs tabs{};
    for (uint32_t j = 0; j < 2 * D; ++j) {
        NSIG_REQUIRE(all_tables_host[j] && aligned16(all_tables_host[j]), "hg_codebook_presum_sel: table %u is null or not 16-byte aligned", j);
        tabs.p[j] = all_tables_host[j];
    }
    NSIG_REQUIRE(aligned16(S), "hg_codebook_presum_sel: S must be 16-byte aligned");
    k_codebook_presum_sel<<<NSIG_TABLE_ROWS / 2 / 256, 256, 0, as_stream(stream)>>>(tabs, message, D, reinterpret_cast<float4 *>(S));
    return check_launch("hg_codebook_presum_sel");
}

static int codebook_adam_sel(const char *who, const float *G, float *const *params_host, float *const *exp_avg_host, float *const *exp_avg_sq_host,
                             float *const *steps_host, const float *message, uint32_t D, const float *lr, float beta1, float beta2,
                             float eps, float grad_scale, float *scratch, const float *next_message, float *S_next, nsig_stream_t stream) {
    NSIG_REQUIRE(G && params_host && exp_avg_host && exp_avg_sq_host && steps_host && message && lr && scratch, "%s: null pointer", who);
    NSIG_REQUIRE(D >= 1 && D <= NSIG_MAX_MESSAGE_DIM, "%s: D=%u out of range", who, D);
    NSIG_REQUIRE(aligned16(G), "%s: G must be 16-byte aligned", who);
    AdamPairPtrs a{};
    StepPtrs s{};
    for (uint32_t j = 0; j < 2 * D; ++j) {
        NSIG_REQUIRE(params_host[j] && exp_avg_host[j] && exp_avg_sq_host[j] && steps_host[j], "%s: table %u has a null pointer", who, j);
        NSIG_REQUIRE(aligned16(params_host[j]) && aligned16(exp_avg_host[j]) && aligned16(exp_avg_sq_host[j]), "%s: table %u is not 16-byte aligned", who, j);
        a.p[j] = params_host[j]; a.m[j] = exp_avg_host[j]; a.v[j] = exp_avg_sq_host[j]; s.s[j] = steps_host[j];
    }
    hipStream_t st = as_stream(stream);
    k_adam_prepare<<<1, NSIG_MAX_MESSAGE_DIM, 0, st>>>(s, message, D, lr, beta1, beta2, scratch);
    if (int e = check_launch(who)) return e;
    // non-temporal accesses (same-box A/B of the bench step, three pairs: 1.124-1.138 ms against 1.151-1.157 with plain accesses; the kernel alone takes
    // the same 160-164 us either way, the next step's gather gains)
    if (next_message)
        k_codebook_adam_sel<true, true><<<NSIG_TABLE_ROWS / 2 / 256, 256, 0, st>>>(reinterpret_cast<const float4 *>(G), a, message, scratch, D, beta1, beta2, eps,
                                                                                 grad_scale, next_message, reinterpret_cast<float4 *>(S_next));
    else
        k_codebook_adam_sel<false, false><<<NSIG_TABLE_ROWS / 2 / 256, 256, 0, st>>>(reinterpret_cast<const float4 *>(G), a, message, scratch, D, beta1, beta2, eps,
                                                                               grad_scale, nullptr, nullptr);
    return check_launch(who);
}

NSIG_EXPORT int opt_codebook_adam_sel(const float *G, float *const *params_host, float *const *exp_avg_host, float *const *exp_avg_sq_host,
                                      float *const *steps_host, const float *message, uint32_t D, const float *lr, float beta1, float beta2,
                                      float eps, float grad_scale, float *scratch, nsig_stream_t stream) {
    return codebook_adam_sel("opt_codebook_adam_sel", G, params_host, exp_avg_host, exp_avg_sq_host, steps_host, message, D, lr, beta1, beta2, eps,
                             grad_scale, scratch, nullptr, nullptr, stream);
}

NSIG_EXPORT int opt_codebook_adam_sel_next(const float *G, float *const *params_host, float *const *exp_avg_host, float *const *exp_avg_sq_host,
                                           float *const *steps_host, const float *message, uint32_t D, const float *lr, float beta1, float beta2,
                                           float eps, float grad_scale, float *scratch, const float *next_message, float *S_next,
                                           nsig_stream_t stream) {
    NSIG_REQUIRE(next_message && S_next && aligned16(S_next), "opt_codebook_adam_sel_next: next_message / S_next null or S_next not 16-byte aligned");
    return codebook_adam_sel("opt_codebook_adam_sel_next", G, params_host, exp_avg_host, exp_avg_sq_host, steps_host, message, D, lr, beta1, beta2, eps,
                             grad_scale, scratch, next_message, S_next, stream);
}

// ----------------------------------------------------------------------------- dense multi-tensor Adam (the decoder's parameters)

// torch's fused multi-tensor Adam walks 64K-element chunks, one workgroup each: the decoder's 29 tensors (262k parameters) become
// ~30 workgroups that each stream 64K elements serially (28 us).  Here a chunk is 1024 elements, so the same update is ~260
// workgroups of one pass (~3 us).  Same arithmetic as adam_update above (torch.optim.Adam, no weight decay / amsgrad).
constexpr int kDenseMax = 32;
constexpr uint32_t kDenseChunk = 1024;
struct DenseAdam {
    float *p[kDenseMax], *m[kDenseMax], *v[kDenseMax], *step[kDenseMax];
    const float *g[kDenseMax];
    uint32_t numel[kDenseMax], chunk0[kDenseMax + 1];   // chunk0: first chunk of tensor i
    uint8_t slot[kDenseMax];                            // where the tensor's two step scalars sit in the scratch (k_adam_dense_prepare's index)
};

__global__ void k_adam_dense_prepare(DenseAdam a, uint32_t n, const float *__restrict__ lr, float beta1, float beta2, float *__restrict__ scratch) {
    const uint32_t i = threadIdx.x;
    if (i >= n) return;
    const float step = *a.step[i] + 1.0f;
    *a.step[i] = step;
    scratch[i] = (float)((double)*lr / (1.0 - exp((double)step * log((double)beta1))));
    scratch[kDenseMax + i] = (float)(1.0 / sqrt(1.0 - exp((double)step * log((double)beta2))));
}

__global__ void __launch_bounds__(256) k_adam_dense(DenseAdam a, uint32_t n, const float *__restrict__ scratch, float beta1, float beta2, float eps,
                                                    float grad_scale) {
    uint32_t i = 0;
    while (i + 1 < n && blockIdx.x >= a.chunk0[i + 1]) ++i;   // uniform: which tensor this chunk belongs to
    const uint32_t base = (blockIdx.x - a.chunk0[i]) * kDenseChunk;
    const float ss = scratch[a.slot[i]], ib = scratch[kDenseMax + a.slot[i]];
    float *__restrict__ pp = a.p[i], *__restrict__ pm = a.m[i], *__restrict__ pv = a.v[i];
    const float *__restrict__ pg = a.g[i];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const uint32_t e = base + u * 256 + threadIdx.x;
        if (e < a.numel[i]) {
            float p = pp[e], m = pm[e], v = pv[e];
            adam_update(pg[e] * grad_scale, p, m, v, beta1, beta2, eps, ss, ib);
            pp[e] = p; pm[e] = m; pv[e] = v;
        }
    }
}

// The same update with the per-tensor step size lr / (1 - beta1^t) and 1 / sqrt(1 - beta2^t) computed by the HOST (torch.optim.Adam's non-capturable state
// keeps its step counts in host tensors: opt_adam_dense_host, the drop-in model's optimiser hook).
struct DenseScalars {
    float ss[kDenseMax], ib[kDenseMax];
};
__global__ void __launch_bounds__(256) k_adam_dense_host(DenseAdam a, uint32_t n, DenseScalars sc, float beta1, float beta2, float eps, float grad_scale) {
    uint32_t i = 0;
    while (i + 1 < n && blockIdx.x >= a.chunk0[i + 1]) ++i;   // uniform: which tensor this chunk belongs to
    const uint32_t base = (blockIdx.x - a.chunk0[i]) * kDenseChunk;
    const float ss = sc.ss[i], ib = sc.ib[i];
    float *__restrict__ pp = a.p[i], *__restrict__ pm = a.m[i], *__restrict__ pv = a.v[i];
    const float *__restrict__ pg = a.g[i];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const uint32_t e = base + u * 256 + threadIdx.x;
        if (e < a.numel[i]) {
            float p = pp[e], m = pm[e], v = pv[e];
            adam_update(pg[e] * grad_scale, p, m, v, beta1, beta2, eps, ss, ib);
            pp[e] = p; pm[e] = m; pv[e] = v;
        }
    }
}

NSIG_EXPORT int opt_adam_dense_host(uint32_t n, float *const *params_host, const float *const *grads_host, float *const *exp_avg_host,
                                    float *const *exp_avg_sq_host, const uint32_t *numel_host, const float *step_sizes_host, const float *inv_bc2_host,
                                    float beta1, float beta2, float eps, float grad_scale, nsig_stream_t stream) {
    NSIG_REQUIRE(params_host && grads_host && exp_avg_host && exp_avg_sq_host && numel_host && step_sizes_host && inv_bc2_host, "opt_adam_dense_host: null pointer");
    hipStream_t st = as_stream(stream);
    for (uint32_t first = 0; first < n; first += kDenseMax) {
        const uint32_t cnt = n - first < (uint32_t)kDenseMax ? n - first : (uint32_t)kDenseMax;
        DenseAdam a{};
        DenseScalars sc{};
        uint32_t chunks = 0;
        for (uint32_t i = 0; i < cnt; ++i) {
            const uint32_t j = first + i;
            NSIG_REQUIRE(params_host[j] && grads_host[j] && exp_avg_host[j] && exp_avg_sq_host[j] && numel_host[j] > 0,
                         "opt_adam_dense_host: tensor %u has a null pointer or no elements", j);
            a.p[i] = params_host[j]; a.g[i] = grads_host[j]; a.m[i] = exp_avg_host[j]; a.v[i] = exp_avg_sq_host[j];
            a.numel[i] = numel_host[j];
            a.chunk0[i] = chunks;
            chunks += ceil_div(numel_host[j], kDenseChunk);
            sc.ss[i] = step_sizes_host[j];
            sc.ib[i] = inv_bc2_host[j];
        }
        a.chunk0[cnt] = chunks;
        k_adam_dense_host<<<chunks, 256, 0, st>>>(a, cnt, sc, beta1, beta2, eps, grad_scale);
        if (int e = check_launch("opt_adam_dense_host")) return e;
    }
    return NSIG_OK;
}

// Large tensors (stage 1: sixteen 4 MiB base tables with their own gradients): 4096-element chunks, float4 per lane, sixteen 16-byte loads in
// flight per lane before the first dependent store, non-temporal loads and moment stores (a 448 MiB stream that nothing re-reads before it is evicted anyway).
constexpr uint32_t kDenseChunk4 = 4096, kDenseBigNumel = 1u << 16, kDenseRideAlong = 1024;
__global__ void __launch_bounds__(256) k_adam_dense_v4(DenseAdam a, uint32_t n, const float *__restrict__ scratch, float beta1, float beta2, float eps,
                                                       float grad_scale) {
    uint32_t i = 0;
    while (i + 1 < n && blockIdx.x >= a.chunk0[i + 1]) ++i;   // uniform: which tensor this chunk belongs to
    const uint32_t base = (blockIdx.x - a.chunk0[i]) * (kDenseChunk4 / 4), n4 = a.numel[i] / 4;
    const float ss = scratch[a.slot[i]], ib = scratch[kDenseMax + a.slot[i]];
    float4 *__restrict__ pp = reinterpret_cast<float4 *>(a.p[i]), *__restrict__ pm = reinterpret_cast<float4 *>(a.m[i]), *__restrict__ pv = reinterpret_cast<float4 *>(a.v[i]);
    const float4 *__restrict__ pg = reinterpret_cast<const float4 *>(a.g[i]);
    float4 p[4], m[4], v[4], g[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const uint32_t e = min(base + u * 256u + threadIdx.x, n4 - 1u);      // clamped: the duplicate is not stored
        p[u] = ld4<true>(pp + e); m[u] = ld4<true>(pm + e); v[u] = ld4<true>(pv + e); g[u] = ld4<true>(pg + e);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const uint32_t e = base + u * 256u + threadIdx.x;
        if (e >= n4) break;
        adam_update(g[u].x * grad_scale, p[u].x, m[u].x, v[u].x, beta1, beta2, eps, ss, ib);
        adam_update(g[u].y * grad_scale, p[u].y, m[u].y, v[u].y, beta1, beta2, eps, ss, ib);
        adam_update(g[u].z * grad_scale, p[u].z, m[u].z, v[u].z, beta1, beta2, eps, ss, ib);
        adam_update(g[u].w * grad_scale, p[u].w, m[u].w, v[u].w, beta1, beta2, eps, ss, ib);
        // the moments stream out; the PARAMETERS go through the caches: the next step's encoder gathers from these 64 MiB (same box, two rounds: encoder 139-141 -> 129-131 us,
        // this pass 95 -> 91, step -0.6 %, on the sparse grid -1.8 %)
        st4<false>(pp + e, p[u]); st4<true>(pm + e, m[u]); st4<true>(pv + e, v[u]);
    }
}

NSIG_EXPORT int opt_adam_dense(uint32_t n, float *const *params_host, const float *const *grads_host, float *const *exp_avg_host,
                               float *const *exp_avg_sq_host, float *const *steps_host, const uint32_t *numel_host, const float *lr, float beta1,
                               float beta2, float eps, float grad_scale, float *scratch, nsig_stream_t stream) {
    NSIG_REQUIRE(params_host && grads_host && exp_avg_host && exp_avg_sq_host && steps_host && numel_host && lr && scratch, "opt_adam_dense: null pointer");
    hipStream_t st = as_stream(stream);
    for (uint32_t first = 0; first < n; first += kDenseMax) {   // 32 tensors per group of launches
        const uint32_t cnt = n - first < (uint32_t)kDenseMax ? n - first : (uint32_t)kDenseMax;
        DenseAdam all{}, small{}, big{};
        uint32_t n_small = 0, n_big = 0, chunks_small = 0, chunks_big = 0;
        auto vectorisable = [&](uint32_t j) {
            return numel_host[j] % 4 == 0 && aligned16(params_host[j]) && aligned16(grads_host[j]) && aligned16(exp_avg_host[j]) && aligned16(exp_avg_sq_host[j]);
        };
        // a group with a large tensor launches the wide kernel anyway: its medium-sized tensors (stage 1: the two MLPs' 3072 + 7168 parameters beside sixteen 4 MiB
        // tables) ride along as a few more workgroups instead of a launch of their own on the step's serial tail (the same update, element by element)
        bool any_big = false;
        for (uint32_t i = 0; i < cnt; ++i) any_big = any_big || (numel_host[first + i] >= kDenseBigNumel && vectorisable(first + i));
        for (uint32_t i = 0; i < cnt; ++i) {
            const uint32_t j = first + i;
            NSIG_REQUIRE(params_host[j] && grads_host[j] && exp_avg_host[j] && exp_avg_sq_host[j] && steps_host[j] && numel_host[j] > 0,
                         "opt_adam_dense: tensor %u has a null pointer or no elements", j);
            all.step[i] = steps_host[j];
            const bool wide = numel_host[j] >= (any_big ? kDenseRideAlong : kDenseBigNumel) && vectorisable(j);
            DenseAdam &d = wide ? big : small;
            uint32_t &k = wide ? n_big : n_small, &chunks = wide ? chunks_big : chunks_small;
            d.p[k] = params_host[j]; d.g[k] = grads_host[j]; d.m[k] = exp_avg_host[j]; d.v[k] = exp_avg_sq_host[j];
            d.numel[k] = numel_host[j];
            d.slot[k] = (uint8_t)i;
            d.chunk0[k] = chunks;
            chunks += ceil_div(numel_host[j], wide ? kDenseChunk4 : kDenseChunk);
            ++k;
        }
        small.chunk0[n_small] = chunks_small;
        big.chunk0[n_big] = chunks_big;
        float *sc = scratch + (size_t)first * 2;
        k_adam_dense_prepare<<<1, kDenseMax, 0, st>>>(all, cnt, lr, beta1, beta2, sc);
        if (int e = check_launch("opt_adam_dense (prepare)")) return e;
        if (n_big) {
            k_adam_dense_v4<<<chunks_big, 256, 0, st>>>(big, n_big, sc, beta1, beta2, eps, grad_scale);
            if (int e = check_launch("opt_adam_dense (wide)")) return e;
        }
        if (n_small) {
            k_adam_dense<<<chunks_small, 256, 0, st>>>(small, n_small, sc, beta1, beta2, eps, grad_scale);
            if (int e = check_launch("opt_adam_dense")) return e;
        }
    }
    return NSIG_OK;
}

static size_t binned_scratch_bytes(uint32_t M, uint32_t sets) { return (size_t)sets * (sizeof(BinHeader) + (size_t)4 * M * sizeof(uint4)); }

static int reserve_owner_lds(const char *who) {
    static bool attr_set = false;
    const size_t lds = (size_t)kBinRows * 2 * sizeof(unsigned long long);
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_scatter_binned), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            set_error("%s: cannot reserve %zu bytes of LDS", who, lds);
            return NSIG_ERR_LAUNCH;
        }
        attr_set = true;
    }
    return NSIG_OK;
}

// sets record arrays [sets][M][8] -> sets tables; scratch = sets headers, then sets queues
// the owners store their rows as 16-byte vectors
static int owner_targets_aligned(const ScatterTargets &tg, uint32_t sets, const char *who) {
    for (uint32_t i = 0; i < sets; ++i) NSIG_REQUIRE(aligned16(tg.g[i]), "%s: gradient table %u must be 16-byte aligned", who, i);
    return NSIG_OK;
}

static int launch_binned(const float *rec, uint32_t M, uint32_t sets, const ScatterTargets &tg, uint32_t replicas, void *scratch, hipStream_t st,
                         const char *who) {
    static_assert(sizeof(BinHeader) % 16 == 0 && kBinGrid == 16 * 16, "the queues follow the headers 16-byte aligned; k_bin_scan walks 16 x 16 workgroups");
    BinHeader *hd = reinterpret_cast<BinHeader *>(scratch);
    uint4 *queue = reinterpret_cast<uint4 *>(hd + sets);
    for (uint32_t i = 0; i < sets; ++i)
        if (hipMemsetAsync(&hd[i].gmax_bits, 0, sizeof(uint32_t), st) != hipSuccess) {
            set_error("%s: hipMemsetAsync failed", who);
            return NSIG_ERR_LAUNCH;
        }
    const size_t lds = (size_t)kBinRows * 2 * sizeof(unsigned long long);
    if (int e = reserve_owner_lds(who)) return e;
    if (int e = owner_targets_aligned(tg, sets, who)) return e;
    const uint32_t blocks = ceil_div(M, kBinThreads) < kBinGrid ? ceil_div(M, kBinThreads) : kBinGrid;
    k_bin_count<<<dim3(blocks, sets), kBinThreads, 0, st>>>(rec, M, hd);
    k_bin_scan<<<sets, 1024, 0, st>>>(hd, blocks);
    k_bin_write<<<dim3(blocks, sets), kBinThreads, 0, st>>>(rec, M, hd, queue);
    unsigned long long *slabs = replicas > 1 ? reinterpret_cast<unsigned long long *>(queue + (size_t)sets * 4 * M) : nullptr;      // (the merge scratch follows the queues)
    k_scatter_binned<<<dim3(kBinSlices * replicas, sets), 1024, lds, st>>>(hd, queue, M, tg, replicas, 0u, nullptr, 0u, OwnerAdam{}, slabs);
    if (replicas > 1) k_scatter_merge<<<dim3(kBinSlices * kMergeParts, sets), kMergeThreads, 0, st>>>(hd, tg, replicas, slabs);
    return check_launch(who);
}

NSIG_EXPORT size_t hg_scatter_binned_scratch_bytes(uint32_t M) { return binned_scratch_bytes(M, 1) + scatter_merge_bytes(kBinReplicas); }

NSIG_EXPORT int hg_scatter_binned(const float *rec, uint32_t M, float *G, void *scratch, nsig_stream_t stream) {
    if (M == 0) return NSIG_OK;
    NSIG_REQUIRE(rec && G && scratch, "hg_scatter_binned: null pointer");
    NSIG_REQUIRE((reinterpret_cast<uintptr_t>(rec) & 15) == 0 && (reinterpret_cast<uintptr_t>(scratch) & 15) == 0 && M < (1u << 28),
                 "hg_scatter_binned: rec and scratch must be 16-byte aligned and M < 2^28");
    ScatterTargets tg{};
    tg.g[0] = G;
    return launch_binned(rec, M, 1, tg, kBinReplicas, scratch, as_stream(stream), "hg_scatter_binned");
}

NSIG_EXPORT size_t hg_scatter_plan_bytes(uint32_t M) { return scatter_plan_bytes(M); }

NSIG_EXPORT int hg_scatter_plan(const float *xyzs, uint32_t M, float bound, void *plan, nsig_stream_t stream) {
    if (M == 0) return NSIG_OK;
    NSIG_REQUIRE(xyzs && plan, "hg_scatter_plan: null pointer");
    NSIG_REQUIRE((reinterpret_cast<uintptr_t>(plan) & 15) == 0 && M < (1u << 28) && bound > 0.0f, "hg_scatter_plan: plan must be 16-byte aligned, M < 2^28, bound > 0");
    const ScatterPlan pl = scatter_plan_view(plan, M);
    hipStream_t st = as_stream(stream);
    const uint32_t blocks = ceil_div(M, kBinThreads) < kBinGrid ? ceil_div(M, kBinThreads) : kBinGrid;
    k_plan_count<<<blocks, kBinThreads, 0, st>>>(xyzs, M, bound, pl.hd);
    k_plan_dest<<<blocks, kBinThreads, 0, st>>>(xyzs, M, bound, pl.hd, pl.dest);
    return check_launch("hg_scatter_plan");
}

NSIG_EXPORT int hg_scatter_planned(const void *plan, uint32_t M, float *G, nsig_stream_t stream) {
    if (M == 0) return NSIG_OK;
    NSIG_REQUIRE(plan && G, "hg_scatter_planned: null pointer");
    NSIG_REQUIRE((reinterpret_cast<uintptr_t>(plan) & 15) == 0 && M < (1u << 28), "hg_scatter_planned: plan must be 16-byte aligned and M < 2^28");
    if (int e = reserve_owner_lds("hg_scatter_planned")) return e;
    const ScatterPlan pl = scatter_plan_view(const_cast<void *>(plan), M);
    ScatterTargets tg{};
    tg.g[0] = G;
    if (int e = owner_targets_aligned(tg, 1, "hg_scatter_planned")) return e;
    // replicas: 1 -> 63 us, 2 -> 40, 4 -> 32, 8 -> 43 on 5.2 M entries (more owners stream less each, but merge with more float atomics)
    // (with the owners' pipelined queue walk, round 5: 2 -> 31.0, 3 -> 28.5, 4 -> 28.9, 6 -> 37.4, 8 -> 43.2 us; the step the same for 2..4)
    k_scatter_binned<<<dim3(kBinSlices * kBinReplicas, 1), 1024, (size_t)kBinRows * 2 * sizeof(unsigned long long), as_stream(stream)>>>(pl.hd, pl.queue, M, tg,
                                                                                                                                           kBinReplicas, 0u, nullptr, 0u,
                                                                                                                                           OwnerAdam{}, pl.slabs);
    k_scatter_merge<<<dim3(kBinSlices * kMergeParts, 1), kMergeThreads, 0, as_stream(stream)>>>(pl.hd, tg, kBinReplicas, pl.slabs);
    return check_launch("hg_scatter_planned");
}

NSIG_EXPORT size_t hg_scatter_levels_scratch_bytes(uint32_t M) {
    return (size_t)NSIG_BASE_LEVELS * M * 32 + binned_scratch_bytes(M, NSIG_BASE_LEVELS);   // the 16 record arrays, then the binning scratch
}

NSIG_EXPORT int hg_scatter_levels(const float *xyzs, float bound, const void *d_planes, uint32_t M, uint32_t stride, float *const *G_host,
                                  void *scratch, nsig_stream_t stream) {
    NSIG_REQUIRE(xyzs && d_planes && G_host && scratch, "hg_scatter_levels: null pointer");
    NSIG_REQUIRE(bound > 0.0f && stride >= M && M < (1u << 28), "hg_scatter_levels: bad bound, stride < M or M >= 2^28");
    NSIG_REQUIRE((reinterpret_cast<uintptr_t>(scratch) & 15) == 0, "hg_scatter_levels: scratch must be 16-byte aligned");
    ScatterTargets tg{};
    for (int l = 0; l < NSIG_BASE_LEVELS; ++l) {
        NSIG_REQUIRE(G_host[l], "hg_scatter_levels: table %d has a null pointer", l);
        tg.g[l] = G_host[l];
    }
    hipStream_t st = as_stream(stream);
    if (M == 0) {
        for (int l = 0; l < NSIG_BASE_LEVELS; ++l)
            if (hipMemsetAsync(tg.g[l], 0, (size_t)NSIG_TABLE_ROWS * 2 * sizeof(float), st) != hipSuccess) {
                set_error("hg_scatter_levels: hipMemsetAsync failed");
                return NSIG_ERR_LAUNCH;
            }
        return NSIG_OK;
    }
    uint4 *rec = reinterpret_cast<uint4 *>(scratch);
    k_level_records<<<dim3(ceil_div(M, 256u), NSIG_BASE_LEVELS), 256, 0, st>>>(xyzs, bound, reinterpret_cast<const float2 *>(d_planes), M, stride,
                                                                              make_level_geom(), rec);
    if (int e = check_launch("hg_scatter_levels (records)")) return e;
    // every (level, slice) has a single owner, which stores its rows: the tables are written, not accumulated into
    return launch_binned(reinterpret_cast<const float *>(rec), M, NSIG_BASE_LEVELS, tg, 1u,
                         reinterpret_cast<char *>(scratch) + (size_t)NSIG_BASE_LEVELS * M * 32, st, "hg_scatter_levels");
}

NSIG_EXPORT size_t hg_levels_plan_bytes(uint32_t M) { return levels_plan_bytes(M); }

NSIG_EXPORT int hg_levels_plan(const float *xyzs, uint32_t M, const uint32_t *rows_dev, float bound, void *plan, nsig_stream_t stream) {
    if (M == 0) return NSIG_OK;
    NSIG_REQUIRE(xyzs && plan, "hg_levels_plan: null pointer");
    NSIG_REQUIRE((reinterpret_cast<uintptr_t>(plan) & 15) == 0 && M < (1u << 27) && bound > 0.0f, "hg_levels_plan: plan must be 16-byte aligned, M < 2^27, bound > 0");
    const LevelsPlan pl = levels_plan_view(plan, M);
    hipStream_t st = as_stream(stream);
    k_levels_count<<<dim3(pl.n_chunks, NSIG_BASE_LEVELS), kBinThreads, 0, st>>>(xyzs, M, rows_dev, bound, make_level_geom(), pl);
    k_levels_scan<<<NSIG_BASE_LEVELS, kBinThreads, 0, st>>>(M, rows_dev, pl);
    return check_launch("hg_levels_plan");
}

// entries + owners of the planned 16-level scatter; adam.on: the owners end with the optimiser step of their rows instead of storing them (tg unused)
// `prepare` (optional): launched between the entries pass and the owners -- behind everything that can fail without a launch (argument checks, the LDS reservations) and
// behind the first launch: hg_levels_scatter_adam's step-count increment then never happens without the owners' update that goes with it.
template <typename Prepare>
static int levels_scatter_launch(const char *who, const float *xyzs, uint32_t M, const uint32_t *rows_dev, float bound, const void *d_planes, uint32_t stride, void *plan,
                                 const ScatterTargets &tg, const OwnerAdam &adam, hipStream_t st, Prepare prepare) {
    NSIG_REQUIRE(xyzs && d_planes && plan, "%s: null pointer", who);
    NSIG_REQUIRE((reinterpret_cast<uintptr_t>(plan) & 15) == 0 && (reinterpret_cast<uintptr_t>(d_planes) & 7) == 0 && M < (1u << 27) && bound > 0.0f && stride >= M,
                 "%s: plan must be 16-byte and d_planes 8-byte aligned, M < 2^27, bound > 0, stride >= M", who);
    if (!adam.on)
        if (int e = owner_targets_aligned(tg, NSIG_BASE_LEVELS, who)) return e;
    if (int e = reserve_owner_lds(who)) return e;
    const size_t staging = (size_t)kLevelStage * sizeof(uint4);
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_level_entries), hipFuncAttributeMaxDynamicSharedMemorySize, (int)staging) != hipSuccess) {
            set_error("%s: cannot reserve %zu bytes of LDS", who, staging);
            return NSIG_ERR_LAUNCH;
        }
        attr_set = true;
    }
    const LevelsPlan pl = levels_plan_view(plan, M);
    k_level_entries<<<dim3(pl.n_chunks, NSIG_BASE_LEVELS), kBinThreads, staging, st>>>(xyzs, M, rows_dev, bound, reinterpret_cast<const float2 *>(d_planes), stride,
                                                                                     make_level_geom(), pl);
    if (int e = check_launch(who)) return e;
    if (int e = prepare()) return e;
    // every (level, slice) has a single owner, which stores its rows: the tables are written, not accumulated into.  (The owners address set l's
    // queue at l * 4 * M' entries: M' = 2 M is this plan's stride of 8 M.)
    const uint32_t M_stride = M * (kLevelQueueStride / 4);
#ifdef NSIG_LEVELS_SPLIT      // diagnostic build (tools/build_variant.sh): one owner launch per level, so that a kernel trace shows each level's time (the plain owners only)
    for (int l = 0; l < NSIG_BASE_LEVELS; ++l) {
        ScatterTargets one{};
        one.g[0] = tg.g[l];
        k_scatter_binned<<<dim3(kBinSlices, 1), 1024, (size_t)kBinRows * 2 * sizeof(unsigned long long), st>>>(pl.hd + l, pl.queue + (size_t)l * kLevelQueueStride * M,
                                                                                                               M_stride, one, 1u, 1u, pl.chunk_max + (size_t)l * pl.n_chunks, pl.n_chunks);
    }
#else
    k_scatter_binned<<<dim3(kBinSlices, NSIG_BASE_LEVELS), 1024, (size_t)kBinRows * 2 * sizeof(unsigned long long), st>>>(pl.hd, pl.queue, M_stride, tg, 1u, 1u,
                                                                                                                          pl.chunk_max, pl.n_chunks, adam);
#endif
    return check_launch(who);
}

NSIG_EXPORT int hg_levels_scatter(const float *xyzs, uint32_t M, const uint32_t *rows_dev, float bound, const void *d_planes, uint32_t stride, void *plan,
                                  float *const *G_host, nsig_stream_t stream) {
    NSIG_REQUIRE(G_host, "hg_levels_scatter: null pointer");
    ScatterTargets tg{};
    for (int l = 0; l < NSIG_BASE_LEVELS; ++l) {
        NSIG_REQUIRE(G_host[l], "hg_levels_scatter: table %d has a null pointer", l);
        tg.g[l] = G_host[l];
    }
    hipStream_t st = as_stream(stream);
    if (M == 0) {
        for (int l = 0; l < NSIG_BASE_LEVELS; ++l)
            if (hipMemsetAsync(tg.g[l], 0, (size_t)NSIG_TABLE_ROWS * 2 * sizeof(float), st) != hipSuccess) {
                set_error("hg_levels_scatter: hipMemsetAsync failed");
                return NSIG_ERR_LAUNCH;
            }
        return NSIG_OK;
    }
    return levels_scatter_launch("hg_levels_scatter", xyzs, M, rows_dev, bound, d_planes, stride, plan, tg, OwnerAdam{}, st, []() { return (int)NSIG_OK; });
}

NSIG_EXPORT int hg_levels_scatter_adam(const float *xyzs, uint32_t M, const uint32_t *rows_dev, float bound, const void *d_planes, uint32_t stride, void *plan,
                                       float *const *params_host, float *const *exp_avg_host, float *const *exp_avg_sq_host, float *const *steps_host, const float *lr,
                                       float beta1, float beta2, float eps, float grad_scale, float *scratch, nsig_stream_t stream) {
    NSIG_REQUIRE(params_host && exp_avg_host && exp_avg_sq_host && steps_host && lr && scratch, "hg_levels_scatter_adam: null pointer");
    NSIG_REQUIRE(M >= 1, "hg_levels_scatter_adam: M must be positive (a step without points still runs the owners: the device row count may be zero)");
    NSIG_REQUIRE(xyzs && d_planes && plan, "hg_levels_scatter_adam: null pointer");
    NSIG_REQUIRE((reinterpret_cast<uintptr_t>(plan) & 15) == 0 && (reinterpret_cast<uintptr_t>(d_planes) & 7) == 0 && M < (1u << 27) && bound > 0.0f && stride >= M,
                 "hg_levels_scatter_adam: plan must be 16-byte and d_planes 8-byte aligned, M < 2^27, bound > 0, stride >= M");      // (before the step counts are touched)
    static_assert(kOwnerAdamMax == kDenseMax, "k_adam_dense_prepare's scratch layout");
    OwnerAdam adam{};
    DenseAdam all{};
    for (int l = 0; l < NSIG_BASE_LEVELS; ++l) {
        NSIG_REQUIRE(params_host[l] && exp_avg_host[l] && exp_avg_sq_host[l] && steps_host[l], "hg_levels_scatter_adam: table %d has a null pointer", l);
        NSIG_REQUIRE(aligned16(params_host[l]) && aligned16(exp_avg_host[l]) && aligned16(exp_avg_sq_host[l]), "hg_levels_scatter_adam: table %d must be 16-byte aligned", l);
        adam.p[l] = params_host[l]; adam.m[l] = exp_avg_host[l]; adam.v[l] = exp_avg_sq_host[l];
        all.step[l] = steps_host[l];
    }
    adam.scratch = scratch;
    adam.beta1 = beta1; adam.beta2 = beta2; adam.eps = eps; adam.grad_scale = grad_scale;
    adam.on = 1u;
    hipStream_t st = as_stream(stream);
    return levels_scatter_launch("hg_levels_scatter_adam", xyzs, M, rows_dev, bound, d_planes, stride, plan, ScatterTargets{}, adam, st, [&]() {
        k_adam_dense_prepare<<<1, kDenseMax, 0, st>>>(all, NSIG_BASE_LEVELS, lr, beta1, beta2, scratch);      // step counts + 1, the two scalars per table
        return check_launch("hg_levels_scatter_adam (prepare)");
    });
}

NSIG_EXPORT int hg_scatter_level(const float *xyzs, float bound, const void *d_plane, uint32_t M, uint32_t level, float *G, nsig_stream_t stream) {
    if (M == 0) return NSIG_OK;
    NSIG_REQUIRE(xyzs && d_plane && G, "hg_scatter_level: null pointer");
    NSIG_REQUIRE(level < NSIG_BASE_LEVELS && bound > 0.0f, "hg_scatter_level: level %u out of range or bad bound", level);
    static bool attr_set = false;
    const size_t lds = (size_t)kSliceRows * 2 * sizeof(float);
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_scatter_level), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            set_error("hg_scatter_level: cannot reserve %zu bytes of LDS", lds);
            return NSIG_ERR_LAUNCH;
        }
        attr_set = true;
    }
    k_scatter_level<<<kSlices * kReplicas, 1024, lds, as_stream(stream)>>>(xyzs, bound, reinterpret_cast<const float2 *>(d_plane), M,
                                                                         1.0f / kBaseResolution[level], G);
    return check_launch("hg_scatter_level");
}
