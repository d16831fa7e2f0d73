// Packed-fp32 probe (round 5): in the decoder's regime -- ~224 workgroups of 8 waves, two waves per SIMD, nothing else on the chip --
// does a v_pk_fma_f32 / v_pk_mul_f32 (two fp32 operations per lane) issue at the rate of a v_fma_f32, or at half of it?
// (tools/issue_probe.hip measured ~1.9 plain slots per v_pk_fma on a FULL chip under the power-managed clock.)
//   build: hipcc --offload-arch=gfx950 -O3 -o tools/_bin/pk_probe tools/pk_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(512) void probe(float* out, int iters, float seed) {
    f32x2 a[8];
    float s[16];
    for (int i = 0; i < 8; ++i) a[i] = f32x2{seed + i, seed - i};
    for (int i = 0; i < 16; ++i) s[i] = seed * i;
    f32x2 x = f32x2{1.0001f + seed, 0.9999f - seed}, y = f32x2{seed, -seed};
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (MODE == 0) {        // 16 plain FMAs
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s[i]) : "v"(x[0]), "v"(y[0]));
            } else if (MODE == 1) { // 8 packed FMAs = 16 fp32 FMAs
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(x), "v"(y));
            } else if (MODE == 2) { // 16 packed FMAs = 32 fp32 FMAs
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i & 7]) : "v"(x), "v"(y));
            } else {                // 8 packed multiplies + 8 packed adds
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(x));
                    asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(y));
                }
            }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float acc = 0;
    for (int i = 0; i < 8; ++i) acc += a[i][0] + a[i][1];
    for (int i = 0; i < 16; ++i) acc += s[i];
    out[blockIdx.x * 512 + threadIdx.x] = acc;
    if (blockIdx.x == 0 && threadIdx.x == 0) out[1 << 20] = (float)(t1 - t0) / (float)iters / 4.f;
}

template <int MODE>
void run(const char* what, int blocks, int threads, float* out) {
    hipLaunchKernelGGL((probe<MODE>), dim3(blocks), dim3(threads), 0, 0, out, 10, 0.001f);
    hipDeviceSynchronize();
    hipLaunchKernelGGL((probe<MODE>), dim3(blocks), dim3(threads), 0, 0, out, 2000, 0.001f);
    hipDeviceSynchronize();
    float c; hipMemcpy(&c, out + (1 << 20), 4, hipMemcpyDeviceToHost);
    printf("%-34s %3d workgroups x %d waves: %6.1f clocks (100 MHz x ?) per group of 16 instructions -> see ratio\n", what, blocks, threads / 64, c);
}

int main() {
    float* out; hipMalloc(&out, ((1 << 20) + 4) * 4);
    for (int threads : {256, 512}) {
        for (int blocks : {8, 224}) {
            run<0>("16 v_fma_f32", blocks, threads, out);
            run<1>("8 v_pk_fma_f32 (same flops)", blocks, threads, out);
            run<2>("16 v_pk_fma_f32 (twice the flops)", blocks, threads, out);
            run<3>("8 v_pk_mul_f32 + 8 v_pk_add_f32", blocks, threads, out);
        }
    }
    return 0;
}
