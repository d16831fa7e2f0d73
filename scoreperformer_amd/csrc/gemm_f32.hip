// Exact-fp32 GEMM for the small / skinny contractions of the path (VALU, LDS-tiled 64x64x16):
//   the VAE heads `MMDVAE.linear` with growing input width 512->544->564->572 and N <= 32
//   (models/scoreperformer/mmd_transformer.py:53-56,144-156), the dense-continuous embedding MLP
//   (modules/transformer/embeddings.py:202-213) and their backward contractions.
// Arbitrary element strides:  A(m,k) = a[m*sam + k*sak],  B(k,n) = b[k*sbk + n*sbn],  C[m*ldc + n].
// C = alpha * A.B + bias[n] (+ C if accumulate).  No alignment requirements.
// Weight-gradient shapes (a handful of output tiles, contraction over every segment of the batch) are split along K
// over blockIdx.z and reduced with fp32 atomics, otherwise 9 workgroups would walk 60k rows serially.
#include "common.h"

namespace {

constexpr int TK = 16;

// Output tile TM x TN: 64x64 (4x4 outputs per thread), or 64x16 / 16x64 (1x4 per thread) for the skinny products -- the VAE heads have
// 4 .. 32 outputs (forward: N small; weight gradient: M small), and on the 64x64 tile 16 / 8 / 3 / 2 times as many multiply-adds were
// issued as their results needed (the kernel is VALU-bound there).
template <int TM, int TN>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ a, long sam, long sak, const float* __restrict__ b,
                                                       long sbk, long sbn, float* __restrict__ c, long ldc,
                                                       const float* __restrict__ bias, int M, int N, int K, float alpha,
                                                       int accumulate, const uint8_t* __restrict__ rowmask, int k_per_split) {
    __shared__ float As[TK][TM + 4];
    __shared__ float Bs[TK][TN + 4];
    constexpr int TXN = TN / 4, RPT = TM * TN / 1024;   // threads along n, rows per thread
    const int tid = threadIdx.x, tx = tid & (TXN - 1), ty = tid / TXN;
    const int m0 = blockIdx.y * TM, n0 = blockIdx.x * TN;
    const bool split = gridDim.z > 1;
    const int k_begin = blockIdx.z * k_per_split, k_end = min(K, k_begin + k_per_split);
    float acc[RPT][4] = {};
    for (int k0 = k_begin; k0 < k_end; k0 += TK) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + 256 * i;        // 1024 elements per tile
            if (e < TM * TK) {   // walk the contiguous dimension of each operand with consecutive threads
                const int m = (sam == 1) ? (e % TM) : (e >> 4), k = (sam == 1) ? (e / TM) : (e & 15);
                const int gm = m0 + m, gk = k0 + k;
                As[k][m] = (gm < M && gk < k_end) ? a[gm * sam + gk * sak] : 0.f;
            }
            if (e < TN * TK) {
                const int n = (sbn == 1) ? (e % TN) : (e >> 4), k = (sbn == 1) ? (e / TN) : (e & 15);
                const int gn = n0 + n, gk = k0 + k;
                Bs[k][n] = (gn < N && gk < k_end) ? b[gk * sbk + gn * sbn] : 0.f;
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < TK; ++k) {
            float av[RPT], bv[4];
#pragma unroll
            for (int i = 0; i < RPT; ++i) av[i] = As[k][ty * RPT + i];
#pragma unroll
            for (int j = 0; j < 4; ++j) bv[j] = Bs[k][tx * 4 + j];
#pragma unroll
            for (int i = 0; i < RPT; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(av[i], bv[j], acc[i][j]);
        }
        __syncthreads();
    }
    const bool lead = blockIdx.z == 0;
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const int m = m0 + ty * RPT + i;
        if (m >= M) continue;
        const float rs = rowmask ? (rowmask[m] ? 1.f : 0.f) : 1.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + tx * 4 + j;
            if (n >= N) continue;
            float v = acc[i][j] * alpha + ((bias && lead) ? bias[n] : 0.f);
            v *= rs;
            if (split) { atomicAdd(c + m * ldc + n, v); continue; }
            if (accumulate) v += c[m * ldc + n];
            c[m * ldc + n] = v;
        }
    }
}

}  // namespace

extern "C" int spn_gemm_f32(const float* a, long sam, long sak, const float* b, long sbk, long sbn, float* c, long ldc,
                            const float* bias, const uint8_t* rowmask, int M, int N, int K, float alpha, int accumulate,
                            hipStream_t stream) {
    SPN_REQUIRE(a && b && c && M > 0 && N > 0 && K > 0, "spn_gemm_f32: bad arguments");
    const int TN = N <= 32 ? 16 : 64;
    const int TM = (M <= 32 && TN == 64) ? 16 : 64;
    const int tiles = cdiv(N, TN) * cdiv(M, TM);
    int splits = 1, k_per = K;
    if (tiles < 128 && K >= 2048) {
        splits = cdiv(512, tiles);
        if (splits > K / 256) splits = K / 256;
        if (splits > 1) {
            k_per = cdiv(cdiv(K, splits), TK) * TK;
            splits = cdiv(K, k_per);
            if (!accumulate) {
                if (ldc == N) hipMemsetAsync(c, 0, (size_t)M * N * 4, stream);
                else hipMemset2DAsync(c, (size_t)ldc * 4, 0, (size_t)N * 4, M, stream);
            }
        } else { splits = 1; k_per = K; }
    }
    dim3 grid(cdiv(N, TN), cdiv(M, TM), splits);
#define GF32(TM_, TN_) hipLaunchKernelGGL((gemm_f32_kernel<TM_, TN_>), grid, dim3(256), 0, stream, a, sam, sak, b, sbk, sbn, c, ldc, bias, M, N, K, alpha, \
                                         accumulate, rowmask, k_per)
    if (TN == 16) GF32(64, 16); else if (TM == 16) GF32(16, 64); else GF32(64, 64);
#undef GF32
    SPN_LAUNCH_CHECK();
    return SPN_OK;
}
