// Device-side hash-grid lookup shared by hashgrid.hip and field.hip.
//
// Behavioural reference: /root/reference/hash_encoding.py:11-46,73-94 (get_voxel_vertices, hash,
// trilinear_interp) with bounding box (0,1), T = 2^19 rows of 2 features, every level hashed.
// Arithmetic is written operation by operation as the reference's tensor ops execute it (separate
// rounded mul / add, IEEE division), so rows are bit-identical and features agree to the last bit
// with the CPU oracle.
#pragma once

#include "common.h"

namespace nsig {

constexpr uint32_t kPrimeY = 2654435761u, kPrimeZ = 805459861u;  // hash_encoding.py:16

// Per-level resolution of the base encoder: floor(16 * b^l) evaluated in fp32 with
// b = exp((ln 2048 - ln 16)/15) (hash_encoding.py:60,100).  The values are tabulated rather than
// recomputed because they depend on the last bit of the reference's exp/log/pow (note the finest
// level is 2047, not 2048); tests/test_oracle_golden.py pins them against the reference itself.
constexpr float kBaseResolution[NSIG_BASE_LEVELS] = {16.f,  22.f,  30.f,  42.f,  58.f,  80.f,   111.f,  153.f,
                                                     212.f, 294.f, 406.f, 561.f, 776.f, 1072.f, 1482.f, 2047.f};
constexpr float kCodebookResolution = 2048.f;  // network_wtmk_tcnn.py:43-44: b == 1

struct LevelGeom {
    float cell[NSIG_BASE_LEVELS + 1];  // fp32(1/res) per base level; [16] = codebook level
};

inline LevelGeom make_level_geom() {
    LevelGeom g;
    for (int l = 0; l < NSIG_BASE_LEVELS; ++l) g.cell[l] = 1.0f / kBaseResolution[l];  // hash_encoding.py:37
    g.cell[NSIG_BASE_LEVELS] = 1.0f / kCodebookResolution;
    return g;
}

struct TablePtrs {
    const float *p[NSIG_BASE_LEVELS];
};
struct CodebookPtrs {
    const float *p[NSIG_MAX_MESSAGE_DIM];
};
struct GradPtrs {
    float *p[NSIG_MAX_MESSAGE_DIM];
};

struct Corner8 {
    uint32_t row[8];  // corner c = (c>>2 &1, c>>1 &1, c &1) = (dx,dy,dz), hash_encoding.py:8
    float wx, wy, wz;
};

__device__ inline void axis_cell(float v, float cell, uint32_t &idx, float &w) {
    const float vc = fminf(fmaxf(v, 0.0f), 1.0f);     // hash_encoding.py:33-35 (indexing only)
    const int i = (int)floorf(vc / cell);             // :39
    const float lo = (float)i * cell;                 // :40
    const float hi = lo + cell;                       // :41
    w = (v - lo) / (hi - lo);                         // :80 (unclamped v)
    idx = (uint32_t)i;
}

__device__ inline void corner_rows(float x, float y, float z, float cell, Corner8 &c) {
    uint32_t ix, iy, iz;
    axis_cell(x, cell, ix, c.wx);
    axis_cell(y, cell, iy, c.wy);
    axis_cell(z, cell, iz, c.wz);
    const uint32_t hx[2] = {ix, ix + 1u};
    const uint32_t hy[2] = {iy * kPrimeY, (iy + 1u) * kPrimeY};
    const uint32_t hz[2] = {iz * kPrimeZ, (iz + 1u) * kPrimeZ};
#pragma unroll
    for (int k = 0; k < 8; ++k) c.row[k] = (hx[(k >> 2) & 1] ^ hy[(k >> 1) & 1] ^ hz[k & 1]) & kRowMask;  // :11-22
}

// hash_encoding.py:73-94: x first (pairs (0,4),(1,5),(2,6),(3,7)), then y, then z.
__device__ inline float2 trilerp(const float2 (&e)[8], float wx, float wy, float wz) {
    const float ux = 1.0f - wx, uy = 1.0f - wy, uz = 1.0f - wz;
    float2 c00, c01, c10, c11, c0, c1, c;
    c00.x = e[0].x * ux + e[4].x * wx; c00.y = e[0].y * ux + e[4].y * wx;
    c01.x = e[1].x * ux + e[5].x * wx; c01.y = e[1].y * ux + e[5].y * wx;
    c10.x = e[2].x * ux + e[6].x * wx; c10.y = e[2].y * ux + e[6].y * wx;
    c11.x = e[3].x * ux + e[7].x * wx; c11.y = e[3].y * ux + e[7].y * wx;
    c0.x = c00.x * uy + c10.x * wy; c0.y = c00.y * uy + c10.y * wy;
    c1.x = c01.x * uy + c11.x * wy; c1.y = c01.y * uy + c11.y * wy;
    c.x = c0.x * uz + c1.x * wz; c.y = c0.y * uz + c1.y * wz;
    return c;
}

// One level: 8 independent 8-byte gathers in flight, then the interpolation.
__device__ inline float2 encode_level(const float *__restrict__ table, float x, float y, float z, float cell) {
    Corner8 c;
    corner_rows(x, y, z, cell, c);
    // Issue order 0,4,1,5,2,6,3,7: corners k and k+4 differ only in x, i.e. in the low bits of the row index, so
    // they share a 128-byte line in 15 of 16 cells; back to back, the second load hits the line the first one brought in.
    float2 e[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int k = (q >> 1) + 4 * (q & 1);
        e[k] = reinterpret_cast<const float2 *>(table)[c.row[k]];
    }
    return trilerp(e, c.wx, c.wy, c.wz);
}

// Weight with which corner k enters the interpolation, multiplied in the order autograd applies it
// (z factor first, then y, then x).
__device__ inline float corner_weight(const Corner8 &c, int k, float g) {
    const float fz = (k & 1) ? c.wz : 1.0f - c.wz;
    const float fy = ((k >> 1) & 1) ? c.wy : 1.0f - c.wy;
    const float fx = ((k >> 2) & 1) ? c.wx : 1.0f - c.wx;
    return ((g * fz) * fy) * fx;
}

// ---- slice-binned codebook scatter: the layout shared by the binning passes (hashgrid.hip) and the producer of the queue
// entries (k_field_bwd in field.hip, or k_bin_write from 32-byte records).  See the comment above BinHeader's users.
#ifndef NSIG_BIN_REPLICAS      // (diagnostic builds sweep it: tools/build_variant.sh rep2 -DNSIG_BIN_REPLICAS=2)
#define NSIG_BIN_REPLICAS 4
#endif
constexpr uint32_t kBinThreads = 1024, kBinSlices = 64, kBinRows = NSIG_TABLE_ROWS / kBinSlices, kBinReplicas = NSIG_BIN_REPLICAS;
constexpr uint32_t kBinGrid = 256;   // workgroups of the binning passes (each walks chunks w, w + kBinGrid, ...)

struct BinHeader {                 // scratch header
    uint32_t counts[kBinSlices];   // entries per slice (written by k_bin_scan)
    uint32_t gmax_bits;            // max |gradient| of the launch as a float bit pattern
    uint32_t pad[3];
    uint32_t wg[kBinGrid][kBinSlices];   // count pass: entries of slice s found by workgroup w; k_bin_scan: its write offset
};

// integer cell and interpolation weight of one axis at the codebook level (resolution 2^11: exact scalings, identical to
// corner_rows() with cell = 2^-11)
__device__ inline void codebook_axis(float v01, uint32_t &i, float &w) {
    const float res = kCodebookResolution, cell = 1.0f / kCodebookResolution;
    const int c = (int)floorf(fminf(fmaxf(v01, 0.0f), 1.0f) * res);
    i = (uint32_t)c;
    w = (v01 - (float)c * cell) * res;
}

// hash bits shared by the two x corners of the (dy, dz) = (q >> 1, q & 1) pair of cell (., iy, iz); its slice is bits 13..18
__device__ inline uint32_t pair_hash(uint32_t iy, uint32_t iz, uint32_t q) { return ((iy + (q >> 1)) * kPrimeY) ^ ((iz + (q & 1u)) * kPrimeZ); }
__device__ inline uint32_t pair_slice(uint32_t hyz) { return (hyz >> 13) & (kBinSlices - 1); }

// queue entry of one pair: x cell | the pair's 13 row bits, the x weight, the two gradients times the z and y weights
// (built in corner_weight()'s order ((g*fz)*fy)*fx; the owner applies fx)
__device__ inline uint4 pair_entry(uint32_t ix, uint32_t hyz, float wx, float wy, float wz, float g0, float g1, uint32_t q) {
    const float fz = (q & 1u) ? wz : 1.0f - wz, fy = (q >> 1) ? wy : 1.0f - wy;
    return make_uint4(ix | ((hyz & (kBinRows - 1)) << 16), __float_as_uint(wx), __float_as_uint((g0 * fz) * fy), __float_as_uint((g1 * fz) * fy));
}

// A planned scatter's scratch: header, queue [4M] entries, destinations [M] (queue index of each of the point's four pairs)
// Replicas of a slice owner (kBinReplicas > 1) merge EXACTLY: each leaves its 64-bit fixed-point accumulators as a slab, and a second launch (k_scatter_merge) adds
// a slice's slabs as integers (any order gives the same sum), converts once and adds ONE float per element to G.  The sums are then those of a single owner, bit
// for bit, whatever order the replicas ran in.  Per record set: [slice][replica] slabs of 2 * kBinRows words.
inline size_t scatter_merge_bytes(uint32_t replicas) { return (size_t)kBinSlices * replicas * (2 * kBinRows) * sizeof(unsigned long long); }

struct ScatterPlan {
    BinHeader *hd;
    uint4 *queue;
    uint4 *dest;
    unsigned long long *slabs;      // [kBinSlices][kBinReplicas][2 * kBinRows]
};
inline size_t scatter_plan_bytes(uint32_t M) { return sizeof(BinHeader) + (size_t)5 * M * sizeof(uint4) + scatter_merge_bytes(kBinReplicas); }
inline ScatterPlan scatter_plan_view(void *scratch, uint32_t M) {
    ScatterPlan pl;
    pl.hd = reinterpret_cast<BinHeader *>(scratch);
    pl.queue = reinterpret_cast<uint4 *>(pl.hd + 1);
    pl.dest = pl.queue + (size_t)4 * M;
    pl.slabs = reinterpret_cast<unsigned long long *>(pl.dest + M);
    return pl;
}

}  // namespace nsig
