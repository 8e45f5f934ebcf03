// Microbenchmark: what does a wave-wide gather from an L2-resident 4 MiB table cost as a function of its address pattern?
//   hipcc --offload-arch=gfx950 -O3 tools/micro/gather_rate.hip -o tools/_build/gather_rate   (build here, run on the GPU box)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

constexpr uint32_t kRows = 1u << 19;   // rows of 8 bytes: 4 MiB

__device__ inline uint32_t mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// mode 0: all lanes of a wave read the same row; 1: every lane a random row; 2: lanes 2i, 2i+1 read the two rows of one aligned
// pair; 3: random 16-byte loads (aligned pairs); 4: 8 lanes share a row; 5: consecutive rows (coalesced)
template <int MODE>
__global__ void __launch_bounds__(256) k_gather(const float2 *__restrict__ table, uint32_t iters, float *__restrict__ out) {
    const uint32_t gid = blockIdx.x * 256 + threadIdx.x, wave = gid >> 6, lane = gid & 63;
    float acc = 0.0f;
    for (uint32_t it = 0; it < iters; it += 8) {
        float2 v[8];
        float4 w[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const uint32_t h = mix((wave * 131071u + it + u) * 2654435761u);
            uint32_t row;
            if (MODE == 0) row = h;
            else if (MODE == 1) row = mix(h + lane * 0x9e3779b9u);
            else if (MODE == 2) row = (mix(h + (lane >> 1) * 0x9e3779b9u) & ~1u) | (lane & 1u);
            else if (MODE == 3) row = mix(h + lane * 0x9e3779b9u) & ~1u;
            else if (MODE == 4) row = mix(h + (lane >> 3) * 0x9e3779b9u);
            else row = h + lane;
            row &= kRows - 1;
            if (MODE == 3) w[u] = reinterpret_cast<const float4 *>(table)[row >> 1];
            else v[u] = table[row];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += MODE == 3 ? w[u].x + w[u].z : v[u].x;
    }
    if (acc == 123.456f) out[gid] = acc;
}

template <int MODE>
static void run(const char *name, const float2 *table, float *out, uint32_t iters) {
    const int blocks = 256 * 8;   // 8 workgroups of 256 threads per CU
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k_gather<MODE><<<blocks, 256>>>(table, iters, out);
    hipEventRecord(e0);
    k_gather<MODE><<<blocks, 256>>>(table, iters, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double wave_loads = (double)blocks * 4 * iters, per_cu = wave_loads / 256.0;
    printf("%-34s %8.1f us  %6.1f clk per wave-load per CU  (%.2f lane-addresses/clk/CU)\n", name, ms * 1e3, ms * 1e-3 * 2.4e9 / per_cu,
           per_cu * 64 / (ms * 1e-3 * 2.4e9));
}

int main() {
    float2 *table; float *out;
    hipMalloc(&table, kRows * sizeof(float2));
    hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    hipMemset(table, 0, kRows * sizeof(float2));
    const uint32_t iters = 512;
    run<0>("same row for the whole wave", table, out, iters);
    run<4>("8 lanes share a row", table, out, iters);
    run<2>("lane pairs in one aligned 16 B", table, out, iters);
    run<1>("every lane a random row (8 B)", table, out, iters);
    run<3>("every lane a random pair (16 B)", table, out, iters);
    run<5>("consecutive rows (coalesced)", table, out, iters);
    return 0;
}
