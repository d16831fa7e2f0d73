// Issue-model probe for the attention redesign: how many VALU instructions does a SIMD issue per cycle next to a stream of MFMAs, as a
// function of the MFMA shape, the number of VALU fillers between two MFMAs, the filler kind, and the number of waves per SIMD?
// Every instruction is an `asm volatile` statement, so program order is exactly what is written here (the compiler neither reorders
// nor pads them); MFMAs rotate over 4 independent accumulators, fillers over 8 independent registers: no data hazards inside the loop.
//   build: hipcc --offload-arch=gfx950 -O3 -o tools/_bin/issue_probe tools/issue_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// KIND: 0 = v_fma_f32, 1 = v_exp_f32, 2 = v_pk_fma_f32, 3 = v_and_b32, 4 = v_cvt_pk_bf16_f32, 5 = mix (fma, exp, and, fma, cvt, max3 ...)
template <int KIND>
__device__ __forceinline__ void filler(int i, float (&r)[8], float c1, float c2, float (&pk)[8][2]) {
    float& x = r[i & 7];
    int kind = KIND;
    if (KIND == 5) { const int m[8] = {0, 1, 3, 0, 4, 0, 3, 6}; kind = m[i & 7]; }
    if (kind == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c1), "v"(c2));
    if (kind == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
    if (kind == 2) {
        typedef float f2 __attribute__((ext_vector_type(2)));
        f2 v = f2{pk[i & 7][0], pk[i & 7][1]}, a = f2{c1, c1}, b = f2{c2, c2};
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(a), "v"(b));
        pk[i & 7][0] = v[0]; pk[i & 7][1] = v[1];
    }
    if (kind == 3) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x) : "v"(c1));
    if (kind == 4) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(x) : "v"(c2));
    if (kind == 6) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c1), "v"(c2));
}

template <int SHAPE, int NFILL, int KIND>
__global__ __launch_bounds__(1024) void probe(float* out, int iters, float seedv) {
    f32x16 acc32[4];
    f32x4 acc16[4];
    for (int i = 0; i < 4; ++i) { for (int r = 0; r < 16; ++r) acc32[i][r] = 0.f; acc16[i] = f32x4{0, 0, 0, 0}; }
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(threadIdx.x * 0.002f - i); }
    float r[8], pk[8][2];
    for (int i = 0; i < 8; ++i) { r[i] = seedv + threadIdx.x * 1e-3f + i; pk[i][0] = r[i]; pk[i][1] = r[i] + 1.f; }
    const float c1 = 0.999f + seedv, c2 = 0.001f + seedv;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            if (SHAPE == 32) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc32[m]) : "v"(a), "v"(b));
            if (SHAPE == 16) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc16[m]) : "v"(a), "v"(b));
#pragma unroll
            for (int f = 0; f < NFILL; ++f) filler<KIND>(m * NFILL + f, r, c1, c2, pk);
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 4; ++i) { for (int q = 0; q < 16; ++q) s += acc32[i][q]; s += acc16[i][0] + acc16[i][3]; }
    for (int i = 0; i < 8; ++i) s += r[i] + pk[i][0] + pk[i][1];
    out[blockIdx.x * 1024 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) out[1 << 22] = (float)(t1 - t0) / (iters * 4.f);
}

float* g_out;
int g_blocks = 256;   // workgroups (= CUs used): 256 = whole chip (power-managed clock), 8 = one CU per XCD (clock at its maximum)
template <int SHAPE, int NFILL, int KIND>
void run(const char* kname) {
    const int iters = 4000;
    for (int waves = 1; waves <= 4; waves *= 2) {
        dim3 g(g_blocks), blk(256 * waves);   // ONE workgroup per CU, `waves` waves on every SIMD of it (a workgroup never spans CUs)
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL((probe<SHAPE, NFILL, KIND>), g, blk, 0, 0, g_out, 100, 0.001f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL((probe<SHAPE, NFILL, KIND>), g, blk, 0, 0, g_out, iters, 0.001f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        float c; hipMemcpy(&c, g_out + (1 << 22), 4, hipMemcpyDeviceToHost);
        // per SIMD: waves * iters * 4 MFMAs in ms
        const double mf_per_simd = (double)waves * iters * 4;
        const double ns_per_mfma_simd = ms * 1e6 / mf_per_simd;
        const double flops = (SHAPE == 32 ? 2.0 * 32 * 32 * 16 : 2.0 * 16 * 16 * 32) * mf_per_simd * 4 * g_blocks;
        printf("%s MFMA %dx%d  fillers/MFMA %2d  waves/SIMD %d : %6.1f cyc/MFMA (wave view) | SIMD: %6.2f ns per MFMA+fillers = %5.2f ns per instr | %6.0f TF/s\n",
               kname, SHAPE, SHAPE, NFILL, waves, c, ns_per_mfma_simd, ns_per_mfma_simd / (1 + NFILL), flops / (ms * 1e-3) / 1e12);
    }
}

int main(int argc, char** argv) {
    if (argc > 1) g_blocks = atoi(argv[1]);
    hipMalloc(&g_out, ((1 << 22) + 4) * 4);
    printf("# %d workgroups\n", g_blocks);
    printf("# issue model probe: ns figures are per SIMD (all waves of the SIMD together); 1 cycle = 0.417 ns at 2.4 GHz\n");
    run<32, 0, 0>("fma "); run<32, 4, 0>("fma "); run<32, 8, 0>("fma "); run<32, 12, 0>("fma "); run<32, 16, 0>("fma "); run<32, 24, 0>("fma ");
    run<16, 0, 0>("fma "); run<16, 2, 0>("fma "); run<16, 4, 0>("fma "); run<16, 6, 0>("fma "); run<16, 8, 0>("fma "); run<16, 12, 0>("fma ");
    run<32, 16, 1>("exp "); run<32, 16, 2>("pkfma"); run<32, 16, 3>("and "); run<32, 16, 4>("cvtpk");
    run<32, 8, 5>("mix "); run<32, 16, 5>("mix "); run<32, 24, 5>("mix ");
    run<16, 4, 5>("mix "); run<16, 8, 5>("mix "); run<16, 12, 5>("mix ");
    return 0;
}
