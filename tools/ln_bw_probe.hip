// What bandwidth does the LayerNorm backward's access MIX allow?  One wave per row of D = 512: reads x fp32 (2 KB), dy bf16 (1 KB),
// d_residual fp32 (2 KB); writes dx fp32 (2 KB) + its bf16 copy (1 KB).  Variants: with / without a 64-lane reduction between loads and
// stores (the dependency LayerNorm has), rows per wave in flight (1 or 2), blocks.
//   build: hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o tools/_bin/ln_bw_probe tools/ln_bw_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float wsum(float v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <int RED, int R2>
__global__ __launch_bounds__(256) void probe(const float* __restrict__ x, const unsigned short* __restrict__ dy, const float* __restrict__ dres,
                                             float* __restrict__ dx, unsigned short* __restrict__ dx16, int T, int rows_per_block) {
    // PROBE_LDS (compile-time): a dummy LDS allocation that caps the resident blocks per CU (160 KiB / PROBE_LDS), i.e. the occupancy
#ifdef PROBE_LDS
    __shared__ float occupancy_cap[PROBE_LDS / 4];
    if (T < 0) occupancy_cap[threadIdx.x] = 1.f;
#endif
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int r0 = blockIdx.x * rows_per_block, r1 = min(T, r0 + rows_per_block);
    for (int row = r0 + w * R2; row < r1; row += 4 * R2) {
        f32x4 xv[R2][2], dr[R2][2]; u32x2 dv[R2][2];
#pragma unroll
        for (int q = 0; q < R2; ++q)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const long o = (long)(row + q) * 512 + (lane + 64 * i) * 4;
                xv[q][i] = *reinterpret_cast<const f32x4*>(x + o);
                dr[q][i] = *reinterpret_cast<const f32x4*>(dres + o);
                dv[q][i] = *reinterpret_cast<const u32x2*>(dy + o);
            }
#pragma unroll
        for (int q = 0; q < R2; ++q) {
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i) s += xv[q][i][0] * __uint_as_float(dv[q][i][0] << 16) + xv[q][i][1] + xv[q][i][2] + xv[q][i][3];
            if (RED) { s = wsum(s); s = wsum(s * 0.5f + 1.f); }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const long o = (long)(row + q) * 512 + (lane + 64 * i) * 4;
                const f32x4 r = (xv[q][i] - s) * 0.5f + dr[q][i];
                *reinterpret_cast<f32x4*>(dx + o) = r;
                u32x2 p; p[0] = (__float_as_uint(r[0]) >> 16) | (__float_as_uint(r[1]) & 0xffff0000u); p[1] = (__float_as_uint(r[2]) >> 16) | (__float_as_uint(r[3]) & 0xffff0000u);
                *reinterpret_cast<u32x2*>(dx16 + o) = p;
            }
        }
    }
}

template <int RED, int R2>
void run(const float* x, const unsigned short* dy, const float* dres, float* dx, unsigned short* dx16, int T, int blocks) {
    int rpb = (T + blocks - 1) / blocks; rpb = (rpb + 7) / 8 * 8;
    const int grid = (T + rpb - 1) / rpb;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<RED, R2>), dim3(grid), dim3(256), 0, 0, x, dy, dres, dx, dx16, T, rpb);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((probe<RED, R2>), dim3(grid), dim3(256), 0, 0, x, dy, dres, dx, dx16, T, rpb);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    const double bytes = (double)T * 512 * (4 + 2 + 4 + 4 + 2);
    printf("reduction %d, rows in flight per wave %d, %5d blocks (%3d rows each): %7.1f us  %5.2f TB/s\n", RED, R2, grid, rpb, ms * 1e3, bytes / (ms * 1e-3) / 1e12);
}

int main() {
    const int T = 131072; const long n = (long)T * 512;
    float *x, *dres, *dx; unsigned short *dy, *dx16;
    hipMalloc(&x, n * 4); hipMalloc(&dres, n * 4); hipMalloc(&dx, n * 4); hipMalloc(&dy, n * 2); hipMalloc(&dx16, n * 2);
    // PROBE_FILL=1: pseudo-random operand bytes instead of zeros (the memory system moves zeros at lower power: up to 15 % faster)
    if (getenv("PROBE_FILL")) {
        unsigned* h = (unsigned*)malloc(n * 4); unsigned s = 12345u;
        for (long i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = (s & 0x007fffffu) | 0x3f000000u; }
        hipMemcpy(x, h, n * 4, hipMemcpyHostToDevice); hipMemcpy(dres, h, n * 4, hipMemcpyHostToDevice); hipMemcpy(dy, h, n * 2, hipMemcpyHostToDevice); free(h);
    } else { hipMemset(x, 0, n * 4); hipMemset(dres, 0, n * 4); hipMemset(dy, 0, n * 2); }
    for (int blocks : {1024, 2048, 4096, 8192, 16384}) {
        run<0, 1>(x, dy, dres, dx, dx16, T, blocks);
        run<1, 1>(x, dy, dres, dx, dx16, T, blocks);
        run<1, 2>(x, dy, dres, dx, dx16, T, blocks);
    }
    return 0;
}
