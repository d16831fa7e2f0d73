// Micro-probe: issue cost of the VALU ops the attention softmax is made of (cycles per wave-instruction on one SIMD).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int OP>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seedv) {
    float a[8]; f32x2 p[8];
    for (int i = 0; i < 8; ++i) { a[i] = seedv + threadIdx.x * 1e-3f + i; p[i] = f32x2{a[i], a[i] + 1.f}; }
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) a[i] = __builtin_amdgcn_exp2f(a[i]) ;
            if (OP == 1) a[i] = __builtin_fmaf(a[i], 1.0001f, 0.5f);
            if (OP == 2) p[i] = p[i] * 1.0001f + 0.5f;
            if (OP == 3) a[i] = __builtin_amdgcn_rcpf(a[i]);
            if (OP == 4) a[i] = fmaxf(fmaxf(a[i], a[(i + 1) & 7]), a[(i + 2) & 7]);
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0; for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) out[1 << 20] = (float)(t1 - t0) / (iters * 8.f);
}
int main() {
    float* out; (void)hipMalloc(&out, ((1 << 20) + 4) * 4);
    const char* names[] = {"v_exp_f32", "v_fma_f32", "v_pk_fma_f32", "v_rcp_f32", "v_max3_f32"};
    for (int waves = 1; waves <= 2; ++waves)
        for (int op = 0; op < 5; ++op) {
            dim3 g(256 * waves);
            if (op == 0) hipLaunchKernelGGL(k<0>, g, dim3(256), 0, 0, out, 20000, 0.001f);
            if (op == 1) hipLaunchKernelGGL(k<1>, g, dim3(256), 0, 0, out, 20000, 0.001f);
            if (op == 2) hipLaunchKernelGGL(k<2>, g, dim3(256), 0, 0, out, 20000, 0.001f);
            if (op == 3) hipLaunchKernelGGL(k<3>, g, dim3(256), 0, 0, out, 20000, 0.001f);
            if (op == 4) hipLaunchKernelGGL(k<4>, g, dim3(256), 0, 0, out, 20000, 0.001f);
            (void)hipDeviceSynchronize();
            float c; (void)hipMemcpy(&c, out + (1 << 20), 4, hipMemcpyDeviceToHost);
            printf("%d wave(s)/SIMD  %-14s %.2f cycles per wave-instruction (one wave's view)\n", waves, names[op], c);
        }
    return 0;
}
