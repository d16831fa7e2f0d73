// Exact-fp32 GEMM for the small / skinny contractions of the path (VALU, LDS-tiled 64x64x16):
//   the VAE heads `MMDVAE.linear` with growing input width 512->544->564->572 and N <= 32
//   (models/scoreperformer/mmd_transformer.py:53-56,144-156), the dense-continuous embedding MLP
//   (modules/transformer/embeddings.py:202-213) and their backward contractions.
// Arbitrary element strides:  A(m,k) = a[m*sam + k*sak],  B(k,n) = b[k*sbk + n*sbn],  C[m*ldc + n].
// C = alpha * A.B + bias[n] (+ C if accumulate).  No alignment requirements.
// Weight-gradient shapes (a handful of output tiles, contraction over every segment of the batch) are split along K
// over blockIdx.z and reduced with fp32 atomics, otherwise 9 workgroups would walk 60k rows serially.
#include "common.h"
#include "tuning.h"

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


// ---- fp32 MFMA kernel (round 5) for the products with BOTH operands contiguous along K: A [M, K] rows, B as nn.Linear keeps its weight,
// [N, K] -- the batched re-priming of a render window (decode.py RenderSession.prefill: 280 - 512 rows through every projection of the
// decoder, 884 ms of gemm_f32_kernel per 12 renders in tools/bench_render.py against 960 ms for all decode steps) and the wide VAE /
// embedding-MLP layers of the train step.  v_mfma_f32_32x32x2_f32 multiplies and accumulates in fp32 (no reduced-precision inputs):
// 256 flop per clock and CU, where the VALU tile above issues 4x4 FMAs per thread and 2 LDS reads per 16 of them.
// Tile 128 x 128 x 16, four waves with 64 x 64 each (2 x 2 MFMA blocks, 64 accumulator registers), operands k-major in LDS (a lane of the
// MFMA owns one row / column and ONE k: consecutive lanes read consecutive words), two LDS stages, one barrier per K tile.
// Needs K, lda, ldb multiples of 4 and 16-byte aligned bases (float4 loads along K); everything else goes to gemm_f32_kernel.
typedef float f32x16m __attribute__((ext_vector_type(16)));

// TT x TT output tile (128: four waves of 64 x 64 = 2 x 2 MFMA blocks; 64: four waves of one 32 x 32 block), K tile TK_ in two LDS stages
// (dynamic LDS: 2 stages x 2 operands x TK_ x (TT + 1) floats), the next K tile in registers while the current one is multiplied.
// The 64-tile / K-tile-64 form is for the few-hundred-row products of a render window's re-priming: those are LATENCY-bound (80 - 512
// workgroups, each alone on its CU, walking K through dependent trips to memory), so it keeps 8 float4 loads per thread in flight per
// trip and gives every wave of the CU one MFMA block; the 128-tile form is the throughput kernel (94 TF/s at 2048 x 4096 x 512).
// SP > 1 (64-tiles only): SP groups of four waves share one output tile and walk every SP-th K tile each -- the serial chain of trips to
// memory is what these few-workgroup products wait for, and a second / fourth group halves / quarters it; the partial tiles are added
// in group order through LDS at the end (a fixed order: the result does not depend on timing).
template <int TT, int TK_, int SP>
__global__ __launch_bounds__(256 * SP) void gemm_f32_mfma_kernel(const float* __restrict__ a, long lda, const float* __restrict__ b, long ldb,
                                                            float* __restrict__ c, long ldc, const float* __restrict__ bias, int M, int N,
                                                            int K, float alpha, int accumulate, const uint8_t* __restrict__ rowmask) {
    constexpr int LD = TT + 1, NB = TT / 64;            // LDS row stride (k-major; ODD: the transposing stores of a wave -- 16 k quads x 4 rows --
                                                        // then spread over all banks, 2-way; + 4 was 8-way), MFMA blocks per wave and dimension
    constexpr int RW = TT * TK_ / 1024;                 // float4 loads per thread and operand per K tile
    constexpr int QK = TK_ / 4;                         // float4 per row of a K tile
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int grp = threadIdx.x >> 8;                   // K group of this wave (SP > 1)
    float* As = smem + grp * (4 * TK_ * LD);            // [2][TK_][LD] per group
    float* Bs = As + 2 * TK_ * LD;
    const int tid = threadIdx.x & 255, lane = tid & 63, w = tid >> 6, wm = w >> 1, wn = w & 1;
    const int m0 = blockIdx.y * TT, n0 = blockIdx.x * TT;
    // global -> registers: load i of thread t is float4 number e = t + 256 i of the tile: row e / QK, k quad e % QK.  TWO register sets:
    // the K tile after next is requested while the current one is multiplied (one tile ahead left ~0.7 us of every trip exposed)
    f32x4 ra0[RW], rb0[RW], ra1[RW], rb1[RW];
    auto gload = [&](f32x4 (&ra)[RW], f32x4 (&rb)[RW], int k0) {
#pragma unroll
        for (int i = 0; i < RW; ++i) {
            const int e = tid + 256 * i, r = e / QK, gk = k0 + (e % QK) * 4;
            const int gm = m0 + r, gn = n0 + r;
            ra[i] = (gm < M && gk < K) ? *reinterpret_cast<const f32x4*>(a + (long)gm * lda + gk) : f32x4{0.f, 0.f, 0.f, 0.f};
            rb[i] = (gn < N && gk < K) ? *reinterpret_cast<const f32x4*>(b + (long)gn * ldb + gk) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto lstore = [&](const f32x4 (&ra)[RW], const f32x4 (&rb)[RW], int st) {
#pragma unroll
        for (int i = 0; i < RW; ++i) {
            const int e = tid + 256 * i, r = e / QK, kq = (e % QK) * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                As[(st * TK_ + kq + j) * LD + r] = ra[i][j];
                Bs[(st * TK_ + kq + j) * LD + r] = rb[i][j];
            }
        }
    };
    f32x16m acc[NB][NB];
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const int T = ((K + TK_ - 1) / TK_ + SP - 1) / SP;   // K tiles per group (tile number u of group g is K tile u SP + g; past K: zeros)
    const int l32 = lane & 31, lk = lane >> 5;
    auto multiply = [&](int st) {
#pragma unroll
        for (int kk = 0; kk < TK_; kk += 2) {
            float av[NB], bv[NB];
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                av[i] = As[(st * TK_ + kk + lk) * LD + wm * (TT / 2) + i * 32 + l32];
                bv[i] = Bs[(st * TK_ + kk + lk) * LD + wn * (TT / 2) + i * 32 + l32];
            }
#pragma unroll
            for (int i = 0; i < NB; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
    };
    auto ktile = [&](int u) { return (u * SP + grp) * TK_; };
    gload(ra0, rb0, ktile(0));
    if (T > 1) gload(ra1, rb1, ktile(1));
    lstore(ra0, rb0, 0);
    __syncthreads();
    for (int t = 0; t < T; t += 2) {
        // stage 0 holds tile t, set 1 tile t + 1, set 0 is free
        if (t + 2 < T) gload(ra0, rb0, ktile(t + 2));
        multiply(0);
        if (t + 1 < T) lstore(ra1, rb1, 1);
        __syncthreads();
        if (t + 1 >= T) break;
        // stage 1 holds tile t + 1, set 0 tile t + 2, set 1 is free
        if (t + 3 < T) gload(ra1, rb1, ktile(t + 3));
        multiply(1);
        if (t + 2 < T) lstore(ra0, rb0, 0);
        __syncthreads();
    }
    if constexpr (SP > 1) {   // partial tiles of groups 1 .. SP - 1 through LDS (every operand stage is dead by now), added in group order
        static_assert(SP == 1 || NB == 1, "K groups: 64-tiles only");
        float* part = smem;   // [SP - 1][256 threads][16]
        if (grp > 0) {
#pragma unroll
            for (int e = 0; e < 16; ++e) part[((grp - 1) * 256 + tid) * 16 + e] = acc[0][0][e];
        }
        __syncthreads();
        if (grp > 0) return;
#pragma unroll
        for (int g = 1; g < SP; ++g)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[0][0][e] += part[((g - 1) * 256 + tid) * 16 + e];
    }
    // accumulator layout of the 32x32 MFMA: lane = column (l32), element e = row 8 (e / 4) + 4 (lane / 32) + e % 4
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int n = n0 + wn * (TT / 2) + j * 32 + l32;
            const float bn = (bias && n < N) ? bias[n] : 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm * (TT / 2) + i * 32 + 8 * (e >> 2) + 4 * lk + (e & 3);
                if (m < M && n < N) {
                    float v = acc[i][j][e] * alpha + bn;
                    if (rowmask) v *= rowmask[m] ? 1.f : 0.f;
                    if (accumulate) v += c[(long)m * ldc + n];
                    c[(long)m * ldc + n] = v;
                }
            }
        }
}

template <int TT, int TK_, int SP>
static void launch_mfma(const float* a, long lda, const float* b, long ldb, float* c, long ldc, const float* bias, int M, int N, int K, float alpha,
                        int accumulate, const uint8_t* rowmask, hipStream_t stream) {
    static std::atomic<unsigned> optin{0};
    constexpr int bytes = SP * 2 * 2 * TK_ * (TT + 1) * 4;
    static_assert(bytes <= 160 * 1024 && (SP - 1) * 256 * 16 * 4 <= bytes, "LDS budget");
    spn_lds_optin(optin, reinterpret_cast<const void*>(&gemm_f32_mfma_kernel<TT, TK_, SP>), bytes);
    hipLaunchKernelGGL((gemm_f32_mfma_kernel<TT, TK_, SP>), dim3(cdiv(N, TT), cdiv(M, TT)), dim3(256 * SP), bytes, stream, a, lda, b, ldb, c, ldc, bias,
                       M, N, K, alpha, accumulate, rowmask);
}

}  // namespace

extern "C" int spn_gemm_f32(const float* a, long sam, long sak, const float* b, long sbk, long sbn, float* c, long ldc,
                            const float* bias, const uint8_t* rowmask, int M, int N, int K, float alpha, int accumulate,
                            hipStream_t stream) {
    SPN_REQUIRE(a && b && c && M > 0 && N > 0 && K > 0, "spn_gemm_f32: bad arguments");
    // both operands contiguous along K, float4-loadable: the fp32 MFMA kernels -- 128-tiles when they fill the chip twice over, else 64-tiles
    // (round 6: also the tall products with a handful of output columns -- the VAE head projections, M = all latents of a level, N = 4..32:
    // they stream A once, and the 64-tile kernel streams it at 2-3x the rate of the VALU tiles whatever fraction of its columns is padding)
    if (sak == 1 && sbk == 1 && (K & 3) == 0 && (sam & 3) == 0 && (sbn & 3) == 0 && M >= 48 && (N >= 64 || (N >= 4 && M >= 4096)) &&
        ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 15) == 0 && spn_tune_i(SPN_TUNE_GEMM_F32_MFMA) != 0) {
        const long wg64 = (long)cdiv(M, 64) * cdiv(N, 64);
#define MF(TT_, TK__, SP_) launch_mfma<TT_, TK__, SP_>(a, sam, b, sbn, c, ldc, bias, M, N, K, alpha, accumulate, rowmask, stream)
        if (N >= 64 && (long)cdiv(M, 128) * cdiv(N, 128) >= 512) MF(128, 16, 1);   // (a skinny product's second 64 columns would be padding)
        else if (wg64 >= 384 || K < 256) MF(64, 64, 1);      // every CU has its workgroup(s) already, or there is hardly a K loop
        else if (K >= 2048) MF(64, 32, 4);                   // few workgroups, long K: four K groups per tile
        else MF(64, 64, 2);
#undef MF
        SPN_LAUNCH_CHECK();
        return SPN_OK;
    }
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
