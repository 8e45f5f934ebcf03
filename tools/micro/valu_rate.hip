// Microbenchmark: issue rate of the VALU operations the MLP kernels lean on (one wave per SIMD, 8 independent chains each).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/valu_rate.hip -o tools/_build/valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int OP>
__global__ void __launch_bounds__(256) k(uint32_t iters, float *out, float seed) {
    float a[8];
    uint32_t u[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = seed + threadIdx.x * 0.001f + i; u[i] = threadIdx.x * 2654435761u + i; }
    for (uint32_t it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) a[i] = a[i] + 1.0001f;                                                    // v_add_f32
            else if (OP == 1) { const f32x2 v = {a[i], a[i] * 0.5f}; u[i] ^= __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2)); a[i] += 1.0f; }   // cvt_pk + mul + xor + add
            else if (OP == 2) u[i] = __builtin_amdgcn_alignbit(u[i], u[i] + it, 31);               // v_alignbit + add
            else if (OP == 3) a[i] = __expf(a[i]) * 0.5f;                                          // v_exp_f32 (+mul, mul)
            else if (OP == 4) a[i] = __builtin_amdgcn_rcpf(a[i]) + 1.0f;                           // v_rcp_f32 + add
            else if (OP == 5) a[i] = a[i] / (a[i] + 3.0f);                                         // IEEE division + add
            else if (OP == 6) { int m; asm("v_bfe_i32 %0, %1, 3, 1" : "=v"(m) : "v"(u[i])); u[i] = (u[i] + 1u) & (uint32_t)m | 1u; }   // bfe + add + and + or
        }
    }
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i] + (float)u[i];
    if (s == 123.456f) out[0] = s;
}

template <int OP>
static void run(const char *name, float *out, int ops_per_step) {
    const uint32_t iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<256, 256>>>(iters, out, 1.0f);
    hipEventRecord(e0);
    k<OP><<<256, 256>>>(iters, out, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double steps = (double)iters * 8;   // per wave; one wave per SIMD
    printf("%-34s %8.1f us   %.2f clk per step per wave (%d VALU ops per step)\n", name, ms * 1e3, ms * 1e-3 * 2.4e9 / steps, ops_per_step);
}

int main() {
    float *out;
    hipMalloc(&out, 4);
    run<0>("v_add_f32", out, 1);
    run<1>("cvt_pk_bf16_f32 + mul + xor + add", out, 4);
    run<2>("v_alignbit + add", out, 2);
    run<3>("v_exp_f32 + 2 mul", out, 3);
    run<4>("v_rcp_f32 + add", out, 2);
    run<5>("IEEE division + add", out, 11);
    run<6>("v_bfe_i32 + add + and + or", out, 4);
    return 0;
}
