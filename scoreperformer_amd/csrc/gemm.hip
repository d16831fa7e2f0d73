// bf16 MFMA GEMM for gfx950:  C[M,N] = epilogue(alpha * A.B)  with fp32 accumulation.
//
// Replaces every nn.Linear / F.linear call site on the ScorePerformer hot path (SURVEY.md Appendix A,
// K3/K5/K7/K8/K11: attention.py:135-142,210-218; feedforward.py:13-21,51-64; embeddings.py:104,139,211,255,
// 345-349; transformer.py:131,185; layers.py:37,46) and their autograd backward GEMMs.
//
// Operand storage (all bf16, leading dimensions in elements, multiples of 8, 16-byte aligned bases):
//   A: !TA -> A[m][k] = a[m*lda + k]   (K contiguous; activations in forward, dY in dX = dY.W)
//       TA -> A[m][k] = a[k*lda + m]   (M contiguous; dY^T in dW = dY^T.X)
//   B: !TB -> B[k][n] = b[n*ldb + k]   (K contiguous; an nn.Linear weight [N,K])
//       TB -> B[k][n] = b[k*ldb + n]   (N contiguous; W in dX = dY.W, X in dW = dY^T.X)
// K-contiguous tiles are staged [rows][64] with a 16-byte XOR swizzle and read with ds_read_b128;
// M/N-contiguous tiles are staged [64][rows] and read with ds_read_b64_tr_b16 (hardware transpose), so no
// operand ever needs a transposed copy in HBM.
//
// Tile 128x128x64, 256 threads = 4 waves (2x2), each wave 64x64 = 4x4 v_mfma_f32_16x16x32_bf16 accumulators.
// Staging: interior tiles go HBM/L2 -> LDS directly with global_load_lds_dwordx4 (1 KiB per wave-instruction, no VGPR
// round trip and no ds_write, whose ~80 B/clk/CU is the first thing a register-staged 128^2 GEMM saturates).  The DMA writes
// LDS linearly (wave base + lane*16), so the XOR swizzle is applied on the per-lane SOURCE address.  Tiles that touch an M/N/K
// edge take the register path with zero fill.  Two LDS stages: tile t+1 streams in while tile t is multiplied.
// The MFMA is issued with operands swapped (D^T = B^T.A^T), which leaves each lane with 4 CONSECUTIVE output columns of one
// row: the epilogue stores 8/16 bytes per lane instead of scattered 2/4-byte elements.
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int BM = 128, BN = 128;   // BK = 64 (long K) or 32 (short K: smaller LDS footprint -> more resident blocks)

struct GemmArgs {
    const bf16_t* A;
    const bf16_t* B;
    void* C;
    const float* bias;        // [N] or null
    const float* residual;    // [M, ldr] fp32 or null: C = residual + rowscale * (alpha*acc + bias)
    const uint8_t* rowmask;   // [M] or null (query-row mask of attention.py:216-218)
    int M, N, K;
    int lda, ldb, ldc, ldr;
    float alpha;
    int accumulate;           // C += (fp32 C only)
    int batch;                // blockIdx.z (batched) ...
    long sA, sB, sC;          // batch strides in elements
    int splitk;               // ... or, when > 1, blockIdx.z = K slice: partial products are atomically added into fp32 C
    int kt_per_split;         // K tiles (of 64) per slice
};

// ---- LDS addressing -------------------------------------------------------------------------------------
// K-contiguous tile: [128 rows][64 k] bf16, row = 128 B = 8 chunks of 16 B, chunk index XOR (row & 7).
template <int BK>
__device__ __forceinline__ int kc_off(int row, int chunk) { return row * (BK * 2) + ((chunk ^ (row & (BK / 8 - 1))) << 4); }
// R-contiguous tile: [64 k][128 r] bf16, row = 256 B = 16 chunks; swizzle spreads the 4 k-rows of one
// transpose-read (and the neighbouring lane group's 4 rows) over distinct 32-byte windows.
__device__ __forceinline__ int rc_swz(int krow) { return (((krow & 3) | (((krow >> 3) & 1) << 2)) << 1); }
__device__ __forceinline__ int rc_off(int krow, int chunk) { return krow * 256 + ((chunk ^ rc_swz(krow)) << 4); }

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// interior tile: 4 DMA instructions per wave; LDS chunk L = (wave*4+i)*64 + lane, source = inverse-swizzled global chunk
template <bool T, int BK>
__device__ __forceinline__ void dma_tile(const bf16_t* __restrict__ p, int ld, int row0, int k0, char* lds_tile, int wave, int lane) {
    constexpr int PER_WAVE = BK / 16;   // 1 KiB pieces per wave: tile = 128*BK*2 bytes = BK/4 KiB over 4 waves
#pragma unroll
    for (int i = 0; i < PER_WAVE; ++i) {
        const int L = (wave * PER_WAVE + i) * 64 + lane;
        const bf16_t* src;
        if (!T) {
            constexpr int CH = BK / 8;
            const int row = L / CH, kc = (L % CH) ^ (row & (CH - 1));
            src = p + (long)(row0 + row) * ld + k0 + kc * 8;
        } else {
            const int krow = L >> 4, rc = (L & 15) ^ rc_swz(krow);
            src = p + (long)(k0 + krow) * ld + row0 + rc * 8;
        }
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(lds_tile + (wave * PER_WAVE + i) * 1024), 16, 0, 0);
    }
}

// edge tile: registers with zero fill
template <bool T, int BK>
struct Stage {
    static constexpr int NR = BK / 16;   // 16-byte chunks per thread
    uint4 r[NR];
    __device__ __forceinline__ void load(const bf16_t* __restrict__ p, int ld, int R, int K, int row0, int k0, int tid) {
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            uint4 v = make_uint4(0, 0, 0, 0);
            const int c = tid + 256 * i;
            if (!T) {
                constexpr int CH = BK / 8;
                const int row = c / CH, kc = c % CH;
                const int gr = row0 + row, gk = k0 + kc * 8;
                if (gr < R && gk < K) v = *reinterpret_cast<const uint4*>(p + (long)gr * ld + gk);
            } else {
                const int krow = c >> 4, rc = c & 15;
                const int gk = k0 + krow, gr = row0 + rc * 8;
                if (gk < K && gr < R) v = *reinterpret_cast<const uint4*>(p + (long)gk * ld + gr);
            }
            r[i] = v;
        }
    }
    __device__ __forceinline__ void store(char* lds, int tid) const {
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int c = tid + 256 * i;
            int off;
            if (!T) {
                constexpr int CH = BK / 8;
                off = kc_off<BK>(c / CH, c % CH);
            } else {
                off = rc_off(c >> 4, c & 15);
            }
            *reinterpret_cast<uint4*>(lds + off) = r[i];
        }
    }
};

// fragment for 16 rows starting at r_base, k-step ks (32 k each): lane l holds row r_base+(l&15), k = 32ks+(l>>4)*8+0..7
template <bool T, int BK>
__device__ __forceinline__ bf16x8 read_frag(const char* lds, int r_base, int ks, int lane) {
    if (!T) {
        const int row = r_base + (lane & 15);
        const int chunk = ks * 4 + (lane >> 4);
        return *reinterpret_cast<const bf16x8*>(lds + kc_off<BK>(row, chunk));
    } else {
        typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
        const int g = lane >> 4, p = lane & 15;
        const int col_byte = (r_base + 4 * (p & 3)) * 2;
        const int chunk = col_byte >> 4, within = col_byte & 15;
        const int k0 = ks * 32 + g * 8 + (p >> 2);
        const int o0 = rc_off(k0, chunk) + within;
        const int o1 = rc_off(k0 + 4, chunk) + within;
        bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(lds + o0));
        bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(lds + o1));
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    }
}

template <bool TA, bool TB, typename OutT, int BK, int STAGES>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs g) {
    constexpr int TILE_BYTES = 128 * BK * 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // stage s: A tile at smem + s*2*TILE_BYTES, B tile right behind it
#define LDS_A(s_) (smem + (s_) * 2 * TILE_BYTES)
#define LDS_B(s_) (smem + (s_) * 2 * TILE_BYTES + TILE_BYTES)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    // XCD-aware tile order (speed only): workgroup id b runs on XCD b % 8, each with a private L2.  Re-number so that every
    // XCD walks a CONTIGUOUS range of (m-tile, n-tile) pairs, n fastest: the n-tiles that share an A panel hit the same L2
    // instead of fetching the panel once per XCD.  Bijective for any grid size.
    int m0, n0;
    {
        const int nwg = gridDim.x * gridDim.y, bid = blockIdx.y * gridDim.x + blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
        m0 = (wg / gridDim.x) * BM;
        n0 = (wg % gridDim.x) * BN;
    }
    const bool split = g.splitk > 1;
    const bf16_t* A = g.A + (split ? 0 : (long)blockIdx.z * g.sA);
    const bf16_t* B = g.B + (split ? 0 : (long)blockIdx.z * g.sB);

    f32x4 acc[4][4];   // acc[i][j]: rows (of C) m = 16i + (lane&15), cols n = 16j + (lane>>4)*4 + r   (swapped-operand layout)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nt_all = (g.K + BK - 1) / BK;
    const int t_begin = split ? blockIdx.z * g.kt_per_split : 0;
    const int nt = split ? min(nt_all - t_begin, g.kt_per_split) : nt_all;
    if (nt <= 0) return;
    const int kbase = t_begin * BK;
    const bool mn_interior = (m0 + BM <= g.M) && (n0 + BN <= g.N);

    Stage<TA, BK> sa;
    Stage<TB, BK> sb;
    auto stage_in = [&](int t, int buf) {   // bring K tile t (relative to kbase) into LDS stage buf
        const int k0 = kbase + t * BK;
        if (mn_interior && k0 + BK <= g.K) {
            dma_tile<TA, BK>(A, g.lda, m0, k0, LDS_A(buf), wave, lane);
            dma_tile<TB, BK>(B, g.ldb, n0, k0, LDS_B(buf), wave, lane);
        } else {
            sa.load(A, g.lda, g.M, g.K, m0, k0, tid);
            sb.load(B, g.ldb, g.N, g.K, n0, k0, tid);
            sa.store(LDS_A(buf), tid);
            sb.store(LDS_B(buf), tid);
        }
    };
    // STAGES-deep software pipeline: tiles t+1 .. t+STAGES-1 are in flight (LDS DMA) while tile t is multiplied.  Each wave waits
    // only for ITS OWN pieces of tile t with a counted vmcnt (DMA instructions retire in order), then one raw s_barrier makes all
    // waves' pieces visible and at the same time proves that everybody is done reading the stage that is refilled next.
    // (__syncthreads() would drain vmcnt to 0 and serialise the pipeline on memory latency.)
    constexpr int PER_TILE = 2 * (BK / 16);     // DMA instructions per wave per K tile (A + B)
#pragma unroll
    for (int s_ = 0; s_ < STAGES - 1; ++s_)
        if (s_ < nt) stage_in(s_, s_);
    for (int t = 0; t < nt; ++t) {
        const int cur = t % STAGES;
        const int ahead = min(STAGES - 2, nt - 1 - t);   // younger tiles that may stay in flight
        if (ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(2 * PER_TILE) : "memory");
        else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(PER_TILE) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (t + STAGES - 1 < nt) stage_in(t + STAGES - 1, (t + STAGES - 1) % STAGES);   // refills the stage read in iteration t-1
#pragma unroll
        for (int ks = 0; ks < BK / 32; ++ks) {
            bf16x8 af[4], bfr[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = read_frag<TA, BK>(LDS_A(cur), wm * 64 + 16 * i, ks, lane);
#pragma unroll
            for (int j = 0; j < 4; ++j) bfr[j] = read_frag<TB, BK>(LDS_B(cur), wn * 64 + 16 * j, ks, lane);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
        }
    }

    // ---- epilogue: lane owns row m = ..+(lane&15) and 4 consecutive columns n = ..+(lane>>4)*4 + 0..3 ----------
    OutT* C = reinterpret_cast<OutT*>(g.C) + (split ? 0 : (long)blockIdx.z * g.sC);
    const bool lead = !split || blockIdx.z == 0;   // bias / residual are added by the first K slice only
    const bool vec_ok = (g.ldc % 4 == 0) && ((reinterpret_cast<uintptr_t>(C) & 15) == 0) && (!g.residual || g.ldr % 4 == 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + wm * 64 + 16 * i + (lane & 15);
        if (m >= g.M) continue;
        const float rs = g.rowmask ? (g.rowmask[m] ? 1.f : 0.f) : 1.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wn * 64 + 16 * j + (lane >> 4) * 4;
            if (n >= g.N) continue;
            f32x4 v = acc[i][j] * g.alpha;
            const bool full4 = n + 4 <= g.N;
            if (g.bias && lead) {
#pragma unroll
                for (int r = 0; r < 4; ++r) if (n + r < g.N) v[r] += g.bias[n + r];
            }
            v *= rs;
            OutT* dst = C + (long)m * g.ldc + n;
            if (full4 && vec_ok) {
                if (g.residual && lead) v += *reinterpret_cast<const f32x4*>(g.residual + (long)m * g.ldr + n);
                if constexpr (sizeof(OutT) == 4) {
                    if (split) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) atomicAdd(dst + r, v[r]);
                    } else {
                        if (g.accumulate) v += *reinterpret_cast<const f32x4*>(dst);
                        *reinterpret_cast<f32x4*>(dst) = v;
                    }
                } else {
                    uint2 pk; pk.x = pack_bf2(v[0], v[1]); pk.y = pack_bf2(v[2], v[3]);
                    *reinterpret_cast<uint2*>(dst) = pk;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (n + r >= g.N) continue;
                    float x = v[r];
                    if (g.residual && lead) x += g.residual[(long)m * g.ldr + n + r];
                    if constexpr (sizeof(OutT) == 4) {
                        if (split) { atomicAdd(dst + r, x); continue; }
                        if (g.accumulate) x += dst[r];
                        dst[r] = x;
                    } else {
                        dst[r] = f2bf(x);
                    }
                }
            }
        }
    }
}

template <bool TA, bool TB, typename OutT, int BK, int STAGES>
int launch_bk(GemmArgs g, hipStream_t stream) {
    // split-K for the weight-gradient shapes (tiny M x N, contraction over all tokens): fill the chip with K slices
    const int tiles = cdiv(g.N, BN) * cdiv(g.M, BM), nt = cdiv(g.K, BK);
    g.splitk = 1; g.kt_per_split = nt;
    if (sizeof(OutT) == 4 && g.batch == 1 && tiles < 384 && nt >= 32) {
        int want = cdiv(768, tiles);
        if (want > nt / 8) want = nt / 8;
        if (want > 1) {
            g.kt_per_split = cdiv(nt, want);
            g.splitk = cdiv(nt, g.kt_per_split);
            if (!g.accumulate) {   // slices add atomically: C must start from zero (stream-ordered memset)
                if (g.ldc == g.N) hipMemsetAsync(g.C, 0, (size_t)g.M * g.N * 4, stream);
                else hipMemset2DAsync(g.C, (size_t)g.ldc * 4, 0, (size_t)g.N * 4, g.M, stream);
            }
        }
    }
    dim3 grid(cdiv(g.N, BN), cdiv(g.M, BM), g.splitk > 1 ? g.splitk : g.batch);
    constexpr int LDS_BYTES = STAGES * 2 * 128 * BK * 2;
    if (LDS_BYTES > 64 * 1024) {
        static bool attr_done = false;   // > 64 KiB of dynamic LDS needs the opt-in attribute (once per instantiation)
        if (!attr_done) {
            hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<TA, TB, OutT, BK, STAGES>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
            attr_done = true;
        }
    }
    hipLaunchKernelGGL((gemm_kernel<TA, TB, OutT, BK, STAGES>), grid, dim3(256), LDS_BYTES, stream, g);
    SPN_LAUNCH_CHECK();
    return SPN_OK;
}

template <bool TA, bool TB, typename OutT>
int launch(const GemmArgs& g, hipStream_t stream) {
    // short contractions (K <= 1024: every projection with d_model = 512 on the input side) are latency-bound per block:
    // BK = 32 halves the LDS footprint (32 KiB) so that 4 blocks stay resident per CU and hide each other's pipeline fill
    static const int variant = getenv("SPN_GEMM_VARIANT") ? atoi(getenv("SPN_GEMM_VARIANT")) : 0;   // tuning aid
    if (variant == 1) return launch_bk<TA, TB, OutT, 64, 2>(g, stream);
    if (variant == 2) return launch_bk<TA, TB, OutT, 64, 3>(g, stream);
    if (variant == 3) return launch_bk<TA, TB, OutT, 32, 4>(g, stream);
    if (variant == 4) return launch_bk<TA, TB, OutT, 32, 2>(g, stream);
    if (variant == 5) return launch_bk<TA, TB, OutT, 64, 4>(g, stream);
    if (variant == 6) return launch_bk<TA, TB, OutT, 32, 3>(g, stream);
    // measured on MI355X (tools/bench_gemm.py): residency beats in-block pipelining -- two 16 KiB-per-operand stages with
    // BK = 32 (32 KiB LDS, 4 blocks/CU) win everywhere except the long-K all-K-contiguous case
    if (!TA && !TB && g.K >= 2048) return launch_bk<TA, TB, OutT, 64, 2>(g, stream);
    return launch_bk<TA, TB, OutT, 32, 2>(g, stream);
}

}  // namespace

// C-ABI ---------------------------------------------------------------------------------------------------
// flags: bit0 = A is M-contiguous (transposed storage), bit1 = B is N-contiguous, bit2 = C is fp32 (else bf16),
//        bit3 = accumulate into C (fp32 C only).
extern "C" int spn_gemm_bf16(const void* A, const void* B, void* C, const float* bias, const float* residual,
                             const uint8_t* rowmask, int M, int N, int K, int lda, int ldb, int ldc, int ldr,
                             float alpha, int flags, int batch, long strideA, long strideB, long strideC,
                             hipStream_t stream) {
    SPN_REQUIRE(A && B && C, "spn_gemm_bf16: null operand");
    SPN_REQUIRE(M > 0 && N > 0 && K > 0 && batch > 0, "spn_gemm_bf16: empty problem");
    SPN_REQUIRE((lda % 8) == 0 && (ldb % 8) == 0, "spn_gemm_bf16: lda/ldb must be multiples of 8 elements");
    SPN_REQUIRE((((uintptr_t)A) & 15) == 0 && (((uintptr_t)B) & 15) == 0, "spn_gemm_bf16: A/B must be 16-byte aligned");
    const bool ta = flags & 1, tb = flags & 2, f32 = flags & 4, accum = flags & 8;
    SPN_REQUIRE(!(accum && !f32), "spn_gemm_bf16: accumulate requires fp32 C");
    // contiguous extents are read in 8-element chunks
    SPN_REQUIRE(ta ? (M % 8 == 0) : (K % 8 == 0), "spn_gemm_bf16: contiguous extent of A must be a multiple of 8");
    SPN_REQUIRE(tb ? (N % 8 == 0) : (K % 8 == 0), "spn_gemm_bf16: contiguous extent of B must be a multiple of 8");
    GemmArgs g;
    g.A = (const bf16_t*)A; g.B = (const bf16_t*)B; g.C = C; g.bias = bias; g.residual = residual; g.rowmask = rowmask;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.ldr = ldr; g.alpha = alpha;
    g.accumulate = accum ? 1 : 0; g.batch = batch; g.sA = strideA; g.sB = strideB; g.sC = strideC;
    if (f32) {
        if (!ta && !tb) return launch<false, false, float>(g, stream);
        if (!ta && tb) return launch<false, true, float>(g, stream);
        if (ta && !tb) return launch<true, false, float>(g, stream);
        return launch<true, true, float>(g, stream);
    } else {
        if (!ta && !tb) return launch<false, false, bf16_t>(g, stream);
        if (!ta && tb) return launch<false, true, bf16_t>(g, stream);
        if (ta && !tb) return launch<true, false, bf16_t>(g, stream);
        return launch<true, true, bf16_t>(g, stream);
    }
}
