// Micro-probe: achievable v_mfma_f32_16x16x32_bf16 / 32x32x16 issue rate on this GPU for a given residency (no memory traffic).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void k16(float* out, int iters) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(threadIdx.x * 0.002f - i); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void k32(float* out, int iters) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(threadIdx.x * 0.002f - i); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <typename F>
void timeit(const char* name, F launch, double flops) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0); for (int r = 0; r < 5; ++r) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("%-34s %8.3f ms  %7.0f TF/s\n", name, ms, flops / (ms * 1e-3) / 1e12);
}
int main() {
    float* out; hipMalloc(&out, 256 * 8 * 256 * 4 * 4);
    const int iters = 20000;
    for (int bpc = 1; bpc <= 4; bpc *= 2) {
        int blocks = 256 * bpc; char nm[64];
        snprintf(nm, 64, "16x16x32 acc16 %d waves/SIMD", bpc);
        timeit(nm, [&] { hipLaunchKernelGGL(k16<16>, dim3(blocks), dim3(256), 0, 0, out, iters); }, 2.0 * 16 * 16 * 32 * 16 * iters * 4.0 * blocks);
        snprintf(nm, 64, "16x16x32 acc4  %d waves/SIMD", bpc);
        timeit(nm, [&] { hipLaunchKernelGGL(k16<4>, dim3(blocks), dim3(256), 0, 0, out, iters); }, 2.0 * 16 * 16 * 32 * 4 * iters * 4.0 * blocks);
        snprintf(nm, 64, "32x32x16 acc4  %d waves/SIMD", bpc);
        timeit(nm, [&] { hipLaunchKernelGGL(k32<4>, dim3(blocks), dim3(256), 0, 0, out, iters); }, 2.0 * 32 * 32 * 16 * 4 * iters * 4.0 * blocks);
    }
    return 0;
}
