// Microbenchmark: LDS atomic throughput per CU for random addresses in a 128 KiB slice (what the owner-computes scatter does).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/lds_atomic_rate.hip -o tools/_build/lds_atomic_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__device__ inline uint32_t mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// MODE 0: atomicAdd float; 1: atomicAdd uint32; 2: plain read-modify-write (racy, rate only); 3: float atomics, lane pairs adjacent;
// 4: atomicAdd uint64 (16384 slots)
template <int MODE>
__global__ void __launch_bounds__(1024) k(uint32_t iters, float *out) {
    extern __shared__ float acc[];   // 32768 floats
    for (uint32_t i = threadIdx.x; i < 32768; i += 1024) acc[i] = 0.0f;
    __syncthreads();
    uint32_t h = mix(blockIdx.x * 1024 + threadIdx.x);
    for (uint32_t it = 0; it < iters; ++it) {
        h = mix(h + it);
        uint32_t a = h & 32767u;
        if (MODE == 3) a = (mix(h >> 1) & 32766u) | (threadIdx.x & 1u);
        if (MODE == 0 || MODE == 3) atomicAdd(acc + a, 1.0f);
        else if (MODE == 1) atomicAdd(reinterpret_cast<uint32_t *>(acc) + a, 1u);
        else if (MODE == 4) atomicAdd(reinterpret_cast<unsigned long long *>(acc) + (a & 16383u), (unsigned long long)h);
        else acc[a] += 1.0f;
    }
    __syncthreads();
    if (acc[threadIdx.x] == 123.0f) out[0] = 1.0f;
}

template <int MODE>
static void run(const char *name, float *out) {
    const uint32_t iters = 2000;
    hipFuncSetAttribute(reinterpret_cast<const void *>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<256, 1024, 131072>>>(iters, out);
    hipEventRecord(e0);
    k<MODE><<<256, 1024, 131072>>>(iters, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double per_cu = 1024.0 * iters;   // lane-operations per CU (one workgroup per CU)
    printf("%-44s %8.1f us   %.2f clk per lane-op per CU   (%.2f lane-ops/clk/CU)\n", name, ms * 1e3, ms * 1e-3 * 2.4e9 / per_cu, per_cu / (ms * 1e-3 * 2.4e9));
}

int main() {
    float *out;
    hipMalloc(&out, 4);
    run<0>("ds_add_f32, random address", out);
    run<3>("ds_add_f32, lane pairs on adjacent dwords", out);
    run<1>("ds_add_u32, random address", out);
    run<4>("ds_add_u64, random address", out);
    run<2>("plain read-modify-write (no atomicity)", out);
    return 0;
}
