// Cluster probe (round 5): would the attention kernels gain from v_mfma_f32_32x32x16_bf16 instead of v_mfma_f32_16x16x32_bf16?
// tools/issue_probe.hip interleaves ONE MFMA with N VALU fillers; the real kernels issue their MFMAs in CLUSTERS (16 per score block,
// 16 per P.V block) between VALU bursts (the softmax), and the 2-3 waves of a SIMD overlap one wave's cluster with another's burst.
// This probe runs that shape: per "key tile"   [MF MFMAs] [V1 VALU] [MF MFMAs] [V2 VALU]   with MF = 16 (16x16x32) or 8 (32x32x16) --
// the same flops -- and the forward kernel's measured VALU mix (per tile and wave: 339 VALU, 32 of them v_exp_f32; the cluster in
// front of the softmax is the score product, the one behind it P.V).  Every instruction is an asm volatile statement (program order =
// issue order); accumulators rotate (8 x f32x4 or 4 x f32x16, chains of 2 as in the kernels), VALU operands over 16 registers.
//   build: hipcc --offload-arch=gfx950 -O3 -o tools/_bin/cluster_probe tools/cluster_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// VALU mix of the forward's linear tile: fma, max3, exp, add, add_co + cndmask (dropout), cvt_pk, integer hash ops.  The index is a
// TEMPLATE argument: with a run-time index a 250-iteration burst is not unrolled, r[i & 15] becomes a scratch access and every
// "instruction" a memory round trip (the first version of this probe measured 200 cycles per VALU that way).
template <int i>
__device__ __forceinline__ void valu(float (&r)[16], float c1, float c2) {
    float& x = r[i & 15];
    float& y = r[(i + 5) & 15];
    switch (i % 11) {
    case 0: asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c1), "v"(c2)); break;
    case 1: asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(c2)); break;
    case 2: asm volatile("v_exp_f32 %0, %0" : "+v"(x)); break;
    case 3: asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(y)); break;
    case 4: asm volatile("v_add_co_u32 %0, vcc, %0, %0" : "+v"(x) : : "vcc"); break;
    case 5: asm volatile("v_cndmask_b32 %0, 0, %0, vcc" : "+v"(x) : : "vcc"); break;
    case 6: asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(x) : "v"(y)); break;
    case 7: asm volatile("v_lshrrev_b32 %0, 11, %0" : "+v"(x)); break;
    case 8: asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x) : "v"(y)); break;
    case 9: asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(c1)); break;
    default: asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c2), "v"(c1)); break;
    }
}

template <int I, int N>
struct Burst {
    static __device__ __forceinline__ void run(float (&r)[16], float c1, float c2) {
        if constexpr (I < N) { valu<I>(r, c1, c2); Burst<I + 1, N>::run(r, c1, c2); }
    }
};

template <int SHAPE, int V1, int V2>
__global__ __launch_bounds__(1024) void probe(float* out, int iters, float seedv) {
    f32x16 acc32[4];
    f32x4 acc16[8];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc32[i][r] = 0.f;
    for (int i = 0; i < 8; ++i) acc16[i] = f32x4{0, 0, 0, 0};
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(threadIdx.x * 0.002f - i); }
    float r[16];
    for (int i = 0; i < 16; ++i) r[i] = seedv + threadIdx.x * 1e-3f + i;
    const float c1 = 0.999f + seedv, c2 = 0.001f + seedv;
    // de-synchronise the waves of a SIMD the way independent workgroups are: wave w starts w thirds of a burst late
    for (int d = 0; d < (int)(threadIdx.x >> 8); ++d) Burst<0, (V1 + V2) / 3>::run(r, c1, c2);
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if (SHAPE == 0) {
            } else if (SHAPE == 32) {
#pragma unroll
                for (int m = 0; m < 8; ++m) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc32[m & 3]) : "v"(a), "v"(b));
            } else {
#pragma unroll
                for (int m = 0; m < 16; ++m) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc16[m & 7]) : "v"(a), "v"(b));
            }
            if (half == 0) Burst<0, V1>::run(r, c1, c2);
            else Burst<3, V2 + 3>::run(r, c1, c2);
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int q = 0; q < 16; ++q) s += acc32[i][q];
    for (int i = 0; i < 8; ++i) s += acc16[i][0] + acc16[i][3];
    for (int i = 0; i < 16; ++i) s += r[i];
    out[blockIdx.x * 1024 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) out[1 << 22] = (float)(t1 - t0) / iters;
}

float* g_out;
int g_blocks = 256;
template <int SHAPE, int V1, int V2>
void run(const char* what) {
    const int iters = 2000;
    for (int waves = 1; waves <= 4; ++waves) {
        dim3 g(g_blocks), blk(256 * waves);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL((probe<SHAPE, V1, V2>), g, blk, 0, 0, g_out, 100, 0.001f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL((probe<SHAPE, V1, V2>), g, blk, 0, 0, g_out, iters, 0.001f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        float c; hipMemcpy(&c, g_out + (1 << 22), 4, hipMemcpyDeviceToHost);
        const double ns_tile_simd = ms * 1e6 / ((double)waves * iters);      // SIMD time per (tile of one wave)
        printf("%-22s MFMA %dx%d  VALU %3d + %3d per tile  waves/SIMD %d : %7.1f cyc/tile (wave view) | SIMD: %7.1f ns per tile\n", what, SHAPE, SHAPE,
               V1, V2, waves, c, ns_tile_simd);
    }
}

int main(int argc, char** argv) {
    if (argc > 1) g_blocks = atoi(argv[1]);
    hipMalloc(&g_out, ((1 << 22) + 4) * 4);
    printf("# %d workgroups; one tile = 2 clusters (16 + 16 MFMA 16x16x32 or 8 + 8 MFMA 32x32x16: equal flops) and two VALU bursts\n", g_blocks);
    run<16, 250, 90>("fwd  (339 VALU)"); run<32, 250, 90>("fwd  (339 VALU)");
    run<16, 200, 60>("fwd diet (260 VALU)"); run<32, 200, 60>("fwd diet (260 VALU)");
    run<16, 120, 60>("bwd-like (180 VALU)"); run<32, 120, 60>("bwd-like (180 VALU)");
    run<16, 0, 0>("MFMA only"); run<32, 0, 0>("MFMA only");
    run<0, 250, 90>("VALU only (339)"); run<0, 120, 60>("VALU only (180)");
    return 0;
}
