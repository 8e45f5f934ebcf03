// Shared host/device helpers of libnerfsig (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/nerfsig.h"

#define NSIG_EXPORT extern "C" __attribute__((visibility("default")))

namespace nsig {

constexpr int kWave = 64;         // CDNA wavefront
constexpr int kCUs = 256;         // MI355X
constexpr uint32_t kRowMask = NSIG_TABLE_ROWS - 1;

void set_error(const char *fmt, ...);
int check_launch(const char *what);

inline hipStream_t as_stream(nsig_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

#define NSIG_REQUIRE(cond, ...)           \
    do {                                  \
        if (!(cond)) {                    \
            nsig::set_error(__VA_ARGS__); \
            return NSIG_ERR_ARG;          \
        }                                 \
    } while (0)

__host__ __device__ inline uint32_t ceil_div(uint32_t a, uint32_t b) { return (a + b - 1) / b; }

__device__ inline float clampf(float v, float lo, float hi) { return fminf(hi, fmaxf(lo, v)); }

// 10-bit-per-axis bit interleave (x -> bit 0, y -> bit 1, z -> bit 2 of every triple).
__host__ __device__ inline uint32_t spread3(uint32_t v) {
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
__host__ __device__ inline uint32_t morton3(uint32_t x, uint32_t y, uint32_t z) {
    return spread3(x) | (spread3(y) << 1) | (spread3(z) << 2);
}
__host__ __device__ inline uint32_t compact3(uint32_t v) {
    v &= 0x49249249u;
    v = (v | (v >> 2)) & 0xc30c30c3u;
    v = (v | (v >> 4)) & 0x0f00f00fu;
    v = (v | (v >> 8)) & 0xff0000ffu;
    v = (v | (v >> 16)) & 0x0000ffffu;
    return v;
}

}  // namespace nsig
