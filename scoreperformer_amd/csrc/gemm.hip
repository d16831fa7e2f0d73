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
// Two kernels (dispatch in launch<>):
//  * gemm_pp_kernel: 256x256x64 tiles, 8 waves in two groups one barrier apart (one multiplies with v_mfma_f32_32x32x16_bf16 while
//    the other reads fragments and issues LDS DMA), 8-slot half-tile ring in 128 KiB LDS.  Takes every shape whose K is a
//    multiple of 64 (>= 256) and whose M, N are multiples of 8 -- M / N edge tiles included.  See the comment above the kernel.
//  * gemm_kernel: 128x128x{32,64} tiles, 4 waves (2x2) x (4x4) v_mfma_f32_16x16x32_bf16, two LDS stages; everything else
//    (ragged K, tiny problems).  Tiles that touch an M/N/K edge take a register path with zero fill.
// Common to both: interior tiles go HBM/L2 -> LDS directly with global_load_lds_dwordx4 (1 KiB per wave-instruction, no VGPR
// round trip and no ds_write); the DMA writes LDS linearly (wave base + lane*16), so the XOR swizzle is applied on the per-lane
// SOURCE address.  The MFMA is issued with operands swapped (D^T = B^T.A^T), which leaves each lane with 4 CONSECUTIVE output
// columns of one row.  The epilogue (alpha, bias, row mask, fp32 residual, accumulate) stages the wave's block through its own LDS
// slab and stores whole 128-byte row segments: partial-line writes make the L2 fetch every line of C first.  Weight-gradient
// shapes (tiny MxN, K = all tokens) split K over blocks that write fp32 partials to a workspace; one small kernel reduces them.
// Tile order: XCD-contiguous (workgroup id % 8 = XCD), column groups of 8 n-tiles, m-tiles down each group.
#include "common.h"
#include "tuning.h"
#include <type_traits>

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
    int splitk;               // ... or, when > 1, blockIdx.z = K slice writing its partial product to C + z*sC (a workspace)
    int kt_per_split;         // K tiles per slice
    int ngroup;               // n-tiles per column group of the tile order
    int pp_addr_ok;           // both operands span < 2^31 bytes (the ping-pong kernel addresses them with 32-bit offsets)
    int slice_xcd;            // ping-pong kernel, split-K: remap blocks so that an XCD runs whole K slices (see the kernel)
    int stagger;              // persistent launch: start delay per CU slot (shader cycles)
    int tx, ty;               // ping-pong kernel: tile grid (n-tiles, m-tiles); a launch with fewer blocks walks it persistently
    // gated-linear-unit epilogue (spn_gemm_glu): N = I gated outputs, B = [2I, K] (value rows | gate rows), C = u [M, 2I]
    void* ws;                 // caller-owned split-K workspace (or null) and its size
    size_t ws_bytes;
    bf16_t* G;                // [M, ldg]: dropout(value * act(gate))
    int ldg;
    uint32_t thr16, seed;     // dropout threshold (0 = none) and seed, as in spn_act_fwd
    float keep_scale;
#ifdef SPN_GEMM_TIMING
    long long* dbg;
#endif
};

// field-by-field copy out of the kernel-argument segment (constant address space: s_load; see the epilogue of gemm_pp_kernel)
typedef const __attribute__((address_space(4))) GemmArgs* GemmKernargPtr;
__device__ __forceinline__ GemmArgs kernarg_copy(GemmKernargPtr k) {
    GemmArgs r;
    r.A = k->A; r.B = k->B; r.C = k->C; r.bias = k->bias; r.residual = k->residual; r.rowmask = k->rowmask;
    r.M = k->M; r.N = k->N; r.K = k->K; r.lda = k->lda; r.ldb = k->ldb; r.ldc = k->ldc; r.ldr = k->ldr;
    r.alpha = k->alpha; r.accumulate = k->accumulate; r.batch = k->batch; r.sA = k->sA; r.sB = k->sB; r.sC = k->sC;
    r.splitk = k->splitk; r.kt_per_split = k->kt_per_split; r.ngroup = k->ngroup; r.pp_addr_ok = k->pp_addr_ok;
    r.slice_xcd = k->slice_xcd; r.stagger = k->stagger; r.tx = k->tx; r.ty = k->ty;
    r.ws = k->ws; r.ws_bytes = k->ws_bytes; r.G = k->G; r.ldg = k->ldg; r.thr16 = k->thr16; r.seed = k->seed; r.keep_scale = k->keep_scale;
#ifdef SPN_GEMM_TIMING
    r.dbg = k->dbg;
#endif
    return r;
}

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
        // within that order: column groups of `ngroup` n-tiles, m-tiles down each group, n fastest inside.  The tiles an XCD
        // runs together then share few B panels AND few A panels (wide N: all of B no longer cycles through the 4 MiB L2 per
        // m-row)
        const int G = g.ngroup, mt = gridDim.y, per = G * mt;
        const int c = wg / per, within = wg - c * per;
        const int gw = min(G, (int)gridDim.x - c * G);   // width of this (possibly last, narrower) group
        m0 = (within / gw) * BM;
        n0 = (c * G + within % gw) * BN;
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
    OutT* C = reinterpret_cast<OutT*>(g.C) + (long)blockIdx.z * g.sC;
    const bool lead = !split || blockIdx.z == 0;   // bias / residual are added by the first K slice only
    const bool vec_ok = (g.ldc % 4 == 0) && ((reinterpret_cast<uintptr_t>(C) & 15) == 0) && (!g.residual || g.ldr % 4 == 0);
    if (mn_interior && vec_ok && (!g.bias || (reinterpret_cast<uintptr_t>(g.bias) & 15) == 0)) {
        // Whole tile inside C.  The accumulator layout gives every store instruction 32-byte (bf16) / 64-byte (fp32) pieces of
        // 16 different rows; partial-line writes make the L2 fetch each 128-byte line of C from HBM first (measured: FETCH_SIZE
        // ~ size of C on the 131072x4096x512 product).  So the wave's 64x64 block goes through its own LDS slab (the operand
        // stages are dead by now) and leaves as whole 128-byte row segments, 16 bytes per lane.
        __syncthreads();   // every wave is done reading the operand stages
        constexpr int ES = sizeof(OutT), ROWB = 64 * ES, CPR = ROWB / 16;   // bytes per row, 16-byte chunks per row
        constexpr int RPP = 8192 / ROWB, NPASS = 64 / RPP;                   // rows per 8 KiB pass (64 bf16 / 32 fp32)
        char* stg = smem + wave * 8192;
        f32x4 bv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            bv[j] = (g.bias && lead) ? *reinterpret_cast<const f32x4*>(g.bias + n0 + wn * 64 + 16 * j + (lane >> 4) * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) {
#pragma unroll
            for (int ii = 0; ii < RPP / 16; ++ii) {
                const int i = pass * (RPP / 16) + ii;
                const int row = 16 * ii + (lane & 15);
                const int m = m0 + wm * 64 + 16 * i + (lane & 15);
                const float rs = g.rowmask ? (g.rowmask[m] ? 1.f : 0.f) : 1.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int n = n0 + wn * 64 + 16 * j + (lane >> 4) * 4;
                    f32x4 v = (acc[i][j] * g.alpha + bv[j]) * rs;
                    if (g.residual && lead) v += *reinterpret_cast<const f32x4*>(g.residual + (long)m * g.ldr + n);
                    const int colb = (16 * j + (lane >> 4) * 4) * ES;
                    char* dst = stg + row * ROWB + ((((colb >> 4) ^ row) & (CPR - 1)) << 4) + (colb & 15);
                    if constexpr (ES == 4) {
                        *reinterpret_cast<f32x4*>(dst) = v;
                    } else {
                        uint2 pk; pk.x = pack_bf2(v[0], v[1]); pk.y = pack_bf2(v[2], v[3]);
                        *reinterpret_cast<uint2*>(dst) = pk;
                    }
                }
            }
            // the slab is private to the wave and DS operations of one wave complete in order: no barrier
#pragma unroll
            for (int it = 0; it < RPP * CPR / 64; ++it) {
                const int row = it * (64 / CPR) + lane / CPR, chunk = lane % CPR;
                const uint4 val = *reinterpret_cast<const uint4*>(stg + row * ROWB + (((chunk ^ row) & (CPR - 1)) << 4));
                OutT* dst = C + (long)(m0 + wm * 64 + pass * RPP + row) * g.ldc + n0 + wn * 64 + chunk * (16 / ES);
                if constexpr (ES == 4) {
                    f32x4 v = __builtin_bit_cast(f32x4, val);
                    if (g.accumulate) v += *reinterpret_cast<const f32x4*>(dst);
                    *reinterpret_cast<f32x4*>(dst) = v;
                } else {
                    *reinterpret_cast<uint4*>(dst) = val;
                }
            }
        }
        return;
    }
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
                    if (g.accumulate) v += *reinterpret_cast<const f32x4*>(dst);
                    *reinterpret_cast<f32x4*>(dst) = v;
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

#ifdef SPN_GEMM_TIMING
static long long* g_dbg = nullptr;   // tuning aid (tools/gemm_timing.py): per-segment s_memtime sums of block 0, waves 0 and 4
#define TSTAMP(k_) do { const long long t__ = __builtin_amdgcn_s_memtime(); seg[k_] += t__ - tprev; tprev = t__; } while (0)
#else
#define TSTAMP(k_) do {} while (0)
#endif

// ---- 256x256 ping-pong kernel ------------------------------------------------------------------------------
// The two-barrier loop above tops out near 0.9 PFLOP/s: every K step each wave stages, waits, reads and multiplies in lockstep,
// so the matrix pipe idles through the staging part.  This kernel splits the 8 waves of a 256x256 tile into two groups
// (waves 0-3 / 4-7: one wave of each group per SIMD) that run ONE BARRIER APART: while one group issues its MFMAs the other one
// reads fragments and issues LDS DMA, and they swap at every barrier.
//   * K tile 64 (128-byte rows: every DMA lane group fetches whole cache lines).  A wave owns 128x64 of C = 2x2 quadrants of 64x32;
//     phase 4t+p multiplies one quadrant over K tile t:  p0 (A0,B0)  p1 (A0,B1)  p2 (A1,B1)  p3 (A1,B0), Ah = the wave's rows
//     64h..64h+63, Bh = its columns 32h..32h+31.  Reads per phase: A0+B0 / B1 / A1 / none (B0 stays in registers).
//   * the operands are staged as 16 KiB half-tiles: AX = the A0 rows of both row groups, AY = the A1 rows, BX / BY alike for the
//     four column groups.  Stream S = AX(0) BX(0) BY(0) AY(0) AX(1) ...: exactly the order of first use.  Each kind has 2 slots
//     (K tile parity): 8 x 16 KiB = 128 KiB.
//   * the read part of phase p issues S[p+6] (2 DMA instructions per wave) and then waits with vmcnt(8): S[<= p+2] has landed
//     (this wave's pieces; 4 half-tiles stay in flight), the barrier publishes it, and it is read in phase p+1 or later
//     (AX(t) = S[4t], BX(t) = S[4t+1] first read in phase 4t, BY(t) = S[4t+2] in 4t+1, AY(t) = S[4t+3] in 4t+2)
//   * S[p+6] overwrites a slot last read in phase <= p-2: both groups have retired those reads (lgkmcnt(0) right after the barrier
//     that starts their MFMA part) at least one barrier before either group issues the refill
// M / N edges: clamped loads + guarded stores.  K must be whole 64-tiles and >= 256; other shapes take the kernel above.
constexpr int PP_BM = 256, PP_BN = 256, PP_BK = 64, PP_LEAD = 6;
// DMA instructions a wave leaves in flight at the steady-state wait (2 per half-tile).  8 = the design point; tools/build_variant.py
// builds 6 / 4 for the in-flight sensitivity probe of round 5 (docs/LOG.md section 7)
#ifndef PP_STEADY_VM
#define PP_STEADY_VM 8
#endif
constexpr int PP_HALF = 128 * PP_BK * 2;   // bytes of one half-tile (128 rows x 64 k)

// LDS addressing of the ping-pong kernel (32-row fragments of v_mfma_f32_32x32x16_bf16).
// K-contiguous half-tile [128 rows][64 k]: row = 128 B = 8 chunks of 16 B; two rows share a 256-byte bank window and the XOR key is the
// window index, so the 16 rows of one ds_read_b128 lane group (rows {0-3,12-15,20-27} / {4-11,16-19,28-31}) hit 16 distinct slots.
__device__ __forceinline__ int pp_kc_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }
// R-contiguous half-tile [64 k][128 r]: row = 256 B = 16 chunks.  One transpose-read half (32 lanes) covers 4 k-rows x 2 adjacent
// 32-byte column windows; XOR-ing the window index with 2*(krow & 3) puts the 8 pieces on 8 distinct windows.
__device__ __forceinline__ int pp_rc_swz(int krow) { return (krow & 3) << 2; }
__device__ __forceinline__ int pp_rc_off(int krow, int chunk) { return krow * 256 + ((chunk ^ pp_rc_swz(krow)) << 4); }

// fragment of v_mfma_f32_32x32x16_bf16 for 32 rows starting at r_base, k-step ks (16 k each):
// lane l holds row r_base + (l & 31), k = 16 ks + (l >> 5) * 8 + 0..7
template <bool T>
__device__ __forceinline__ bf16x8 pp_read_frag(const char* lds, int r_base, int ks, int lane) {
    if (!T) {
        return *reinterpret_cast<const bf16x8*>(lds + pp_kc_off(r_base + (lane & 31), ks * 2 + (lane >> 5)));
    } else {
        // ds_read_b64_tr_b16: within a 16-lane group lane p addresses k-row p/4, columns 4(p%4)..+3 and receives column p
        typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
        const int g = lane >> 4, p = lane & 15;
        const int col_byte = (r_base + 16 * (g & 1) + 4 * (p & 3)) * 2;
        const int chunk = col_byte >> 4, within = col_byte & 15;
        const int k0 = ks * 16 + (g >> 1) * 8 + (p >> 2);
        bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(lds + pp_rc_off(k0, chunk) + within));
        bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(lds + pp_rc_off(k0 + 4, chunk) + within));
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    }
}

// half-tile row x (0..127) -> row of the 256-row block tile.  A: x = 64*group + r -> 128*group + 64*h + r  (2 row groups);
// B: x = 32*group + r -> 64*group + 32*h + r  (4 column groups).  Multiples of 8 stay contiguous.
template <int SUB> __device__ __forceinline__ int pp_row(int x, int h) { return (x / SUB) * (2 * SUB) + SUB * h + (x % SUB); }

// one half-tile = 16 pieces of 1 KiB, 2 per wave; LDS image linear, swizzle applied on the source address
// R = rows (A: M, B: N) of the operand: tiles that hang over the edge re-read the last row (last 8-row group when the rows
// are the contiguous dimension) -- valid memory, and the epilogue never stores what was computed from it.
// One half-tile = 16 pieces of 1 KiB, 2 per wave; the LDS image is linear, the swizzle is applied on the source address.
// The per-lane part of the address is loop-invariant (one 32-bit VGPR per piece, computed once), the K-tile part is a scalar offset,
// the base sits in an SGPR buffer resource: buffer_load_dwordx4 ... offen lds with NO vector address arithmetic inside the K loop
// (+10% over flat global_load_lds with 64-bit per-lane addresses on the long-K shapes).
template <bool T, int SUB>
__device__ __forceinline__ uint32_t pp_voffset(int ld, int row0, int R, int h, int wave, int lane, int i) {
    const int L = (wave * 2 + i) * 64 + lane;
    if (!T) {
        const int x = L >> 3, kc = (L & 7) ^ ((x >> 1) & 7);
        return (uint32_t)(((long)min(row0 + pp_row<SUB>(x, h), R - 1) * ld + kc * 8) * 2);
    } else {
        const int krow = L >> 4, rc = (L & 15) ^ pp_rc_swz(krow);
        return (uint32_t)(((long)krow * ld + min(row0 + pp_row<SUB>(rc * 8, h), R - 8)) * 2);
    }
}

// GLU mode: half-tile row x of BX is weight row row0 + x (value rows), of BY row I + row0 + x (gate rows): wave column group wc then
// holds value AND gate of the same 32 outputs in acc[.][0] / acc[.][1] -- the pairing costs an address, not a weight permutation
__device__ __forceinline__ uint32_t pp_voffset_lin(int ld, int row0, int wave, int lane, int i) {
    const int L = (wave * 2 + i) * 64 + lane;
    const int x = L >> 3, kc = (L & 7) ^ ((x >> 1) & 7);
    return (uint32_t)(((long)(row0 + x) * ld + kc * 8) * 2);
}

// wave-private LDS slab exchange (lane A writes, lane B of the same wave reads): order the accesses for the compiler; no instruction
#define PP_SLAB_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); \
                            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)

// ---- tile epilogue shared by the 256x256 ping-pong kernel and the 256x128 duo kernel -----------------------------------------
// A wave holds 128 rows x 64 columns of C as acc[i][j] (32-row block i, 32-column block j; operand-swapped MFMA: lane l owns row
// l & 31 and, per register quad q, the 4 consecutive columns 8q + 4(l >> 5) ..).  Rows start at cm0 + 128 wr; plain GEMM: columns
// cn0 + 64 wc + 32 j; gated (GLU): j = 0 / 1 are the VALUE / GATE columns of the outputs cn0 + 32 wc ...  The block goes through the
// wave's own 4 KiB LDS slab `stg` and leaves as whole 128-byte row segments (partial-line writes make the L2 fetch C first).
template <typename OutT, int GLU>
__device__ __forceinline__ void pp_store_tile(const GemmArgs& g, f32x16 (&acc)[4][2], const f32x4 (&bv)[8], OutT* C, bool lead,
                                              int cm0, int cn0, int wr, int wc, int lane, char* stg, char* ring = nullptr) {
    constexpr int ES = sizeof(OutT);
    constexpr int JP = ES == 2 ? 2 : 1;                 // 32-column halves of the wave's 128x64 block staged per pass
    constexpr int ROWB = 32 * JP * ES, CPR = ROWB / 16; // 128-byte rows, 8 chunks: a pass = 32 rows = 4 KiB
    if constexpr (GLU >= 3) {
        // GATED BACKWARD (spn_gemm_glu_bwd): the accumulators are dg = dy W2 (input gradient of the FFN's output projection) for the
        // outputs cn0 + 64 wc .. of 128 rows; instead of storing dg for a separate pass the epilogue applies the activation backward:
        //   d = dropout_mask(bf16(dg)),  du[:, c] = d * act(gate),  du[:, I + c] = d * value * act'(gate)
        // with value / gate read from u [M, 2I] (g.G) and du written to g.C -- rounded exactly where GEMM + spn_act_bwd round, so the
        // result is bit-identical to the two-kernel path.  The u blocks come in and the du blocks leave through the wave's LDS slab
        // as whole 128-byte row segments; the column sums of du (the bias gradient of the input projection) are taken from the
        // staged du blocks (bf16-rounded, as the consumer GEMMs see them) and left in row cm0/128 + wr of the partial buffer g.ws.
        //
        // Round 4: this epilogue is VALU-ISSUE bound (two waves per SIMD x ~7.8 k instructions per tile and wave were 2/3 of the
        // launch), so it is written for instruction count:
        //   * no per-element row guard: u and du are addressed through RAW BUFFER resources that end at row M -- rows past M read
        //     zeros (value = gate = 0  =>  both du halves are exactly 0, so the column sums need no mask either) and their stores
        //     are dropped by the bounds check; every tile issues exactly the same instructions (the hand-counted vmcnt holds on
        //     edge tiles as well) and the exec mask is never touched (the guarded version had 310 exec-mask regions per tile);
        //   * ONE sigmoid per element (v_exp + v_rcp are quarter-rate) for SiLU and its derivative;
        //   * the counter hash of the dropout mask starts from a per-block base (row counter + first column pair), one v_add per pair;
        //   * column sums by transposed LDS reads (ds_read_b64_tr_b16: 4 rows of one column per lane) + v_dot2c_f32_bf16 against
        //     (1, 1): 24 instructions per staged block instead of 80.
        constexpr int ACT = GLU - 3;
        const int I = g.N;
        const int colw = cn0 + wc * 64;                       // first of this wave's 64 gated outputs
        const int rl = lane & 31, hl = lane >> 5;
        // The u blocks (32 rows x 64 value columns, 32 x 64 gate columns per 32-row block i) arrive by LDS DMA into this wave's 16 KiB of the
        // (now dead) operand ring: no registers, two 32-row blocks (4 x 4 KiB) in flight while one is processed.  LDS image of a
        // block: [32 rows][8 chunks of 16 B], chunk position = source chunk ^ ((row >> 1) & 7) (swizzle on the SOURCE address).
        // (the launcher guarantees M * ld * 2 < 2^32 for both)
        const u32x4 rsU = spn_buffer_rsrc(g.G, (uint32_t)((long)g.M * g.ldg * 2));
        const __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc(g.C, 0, (int)(uint32_t)((long)g.M * g.ldc * 2), 0x00020000);
        const uint32_t ring_w = spn_lds_addr(ring);
        // per-lane byte offsets: the ROW part lives in the vector offset (the bounds check covers vector + immediate offset only, the scalar
        // offset -- used for the value / gate column half -- is not checked)
        uint32_t vo[4];                                       // u rows it * 8 + (lane >> 3) of the NEXT block to request
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int r = it * 8 + (lane >> 3), c = (lane & 7) ^ ((r >> 1) & 7);
            vo[it] = (uint32_t)(((long)(cm0 + wr * 128 + r) * g.ldg + colw + c * 8) * 2);
        }
        const uint32_t vo_step = (uint32_t)(32 * g.ldg * 2);
        auto issue_u = [&](int i) __attribute__((always_inline)) {   // blocks are requested in order i = 0, 1, 2, 3 -> ring slots 2 (i & 1), 2 (i & 1) + 1
#pragma unroll
            for (int part = 0; part < 2; ++part) {
                const uint32_t dst = ring_w + (uint32_t)((2 * (i & 1) + part) * 4096);
                spn_dma16x2(rsU, dst, vo[0], vo[1], (uint32_t)(part * I) * 2u);
                spn_dma16x2(rsU, dst + 2048u, vo[2], vo[3], (uint32_t)(part * I) * 2u);
            }
#pragma unroll
            for (int it = 0; it < 4; ++it) vo[it] += vo_step;
        };
        issue_u(0); issue_u(1);
#ifdef SPN_GEMM_TIMING
        long long est[13];
        est[0] = __builtin_amdgcn_s_memrealtime();
#define ESTAMP(k_) est[k_] = __builtin_amdgcn_s_memrealtime()
#else
#define ESTAMP(k_) do {} while (0)
#endif
        uint32_t so = (uint32_t)(((long)(cm0 + wr * 128 + (lane >> 3)) * g.ldc + colw + (lane & 7) * 8) * 2);   // du row (lane >> 3) of block 0
        const uint32_t so_step8 = (uint32_t)(8 * g.ldc * 2);
        float csum[2] = {0.f, 0.f};                           // [value / gate] of column colw + lane, all rows of the wave's slab
        const bf16x2_t ones = __builtin_bit_cast(bf16x2_t, 0x3f803f80u);
        const uint32_t thr16 = g.thr16;
        const float ks = g.keep_scale;
        // hash base of this lane's first column pair; the pairs of (j, q, e) follow at constant distances (ffn_drop_bits: x0 = rowc + pair * C)
        const uint32_t hcol = __umul24((uint32_t)(colw >> 1) + 2u * (uint32_t)hl, 0xEBCA77u);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            // VM counter, oldest first (U = the 8 DMA instructions of a block, S = its 8 stores), at the top of block i:
            //   i = 0: U0 U1 | i = 1: U1 U2 S0 | i = 2: U2 S0 U3 S1 | i = 3: U3 S1 S2      (loads and stores retire in order)
            if (i == 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (i == 3) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else if (i == 1) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
            ESTAMP(1 + 3 * i);
            uint2 uv[2][4], ug[2][4];                          // this lane's value / gate pieces: row rl, columns 32 j + 8 q + 4 hl ..
            const char* blk = ring + (2 * (i & 1)) * 4096;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int s_ = 8 * j + 2 * q + hl;
                    const int o = rl * 128 + ((((s_ >> 1) ^ ((rl >> 1) & 7)) & 7) << 4) + (s_ & 1) * 8;
                    uv[j][q] = *reinterpret_cast<const uint2*>(blk + o);
                    ug[j][q] = *reinterpret_cast<const uint2*>(blk + 4096 + o);
                }
            if (i + 2 < 4) {   // the slots of block i are free once the reads above have returned
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                issue_u(i + 2);
            }
            // d = dg rounded to bf16 first (what the unfused GEMM stores and spn_act_bwd reads), then the dropout mask of spn_act_fwd / spn_act_bwd
            float d[2][4][4];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const uint32_t p0 = pack_bf2(acc[i][j][4 * q], acc[i][j][4 * q + 1]), p1 = pack_bf2(acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
                    d[j][q][0] = __uint_as_float(p0 << 16); d[j][q][1] = __uint_as_float(p0 & 0xffff0000u);
                    d[j][q][2] = __uint_as_float(p1 << 16); d[j][q][3] = __uint_as_float(p1 & 0xffff0000u);
                }
            if (thr16) {
                const uint32_t hb = ffn_drop_rowc(cm0 + wr * 128 + 32 * i + rl, g.seed) + hcol;
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            uint32_t x = hb + (uint32_t)(16 * j + 4 * q + e) * 0xEBCA77u;     // = rowc + umul24(pair, C): the pairs stay below 2^24
                            x ^= x >> 11; x = __umul24(x, 0xD35A2Du) + (x >> 8);
                            x ^= x >> 13; x = __umul24(x, 0x9E3B35u) + (x >> 9);
                            x ^= x >> 15;
                            const float s0 = d[j][q][2 * e] * ks, s1 = d[j][q][2 * e + 1] * ks;
                            d[j][q][2 * e] = (x & 0xffffu) >= thr16 ? s0 : 0.f;
                            d[j][q][2 * e + 1] = (x >> 16) >= thr16 ? s1 : 0.f;
                        }
            }
            uint2 pa[2][4], pg[2][4];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float a[4] = {__uint_as_float(uv[j][q].x << 16), __uint_as_float(uv[j][q].x & 0xffff0000u),
                                        __uint_as_float(uv[j][q].y << 16), __uint_as_float(uv[j][q].y & 0xffff0000u)};
                    const float t[4] = {__uint_as_float(ug[j][q].x << 16), __uint_as_float(ug[j][q].x & 0xffff0000u),
                                        __uint_as_float(ug[j][q].y << 16), __uint_as_float(ug[j][q].y & 0xffff0000u)};
                    float da[4], dt[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if constexpr (ACT == 0) {   // silu_f / silu_grad of common.h on ONE sigmoid (same expressions, same rounding)
                            const float sg = __builtin_amdgcn_rcpf(1.f + __expf(-t[e]));
                            da[e] = d[j][q][e] * (t[e] * sg);
                            dt[e] = d[j][q][e] * a[e] * (sg * (1.f + t[e] * (1.f - sg)));
                        } else {
                            da[e] = d[j][q][e] * gelu_f(t[e]);
                            dt[e] = d[j][q][e] * a[e] * gelu_grad(t[e]);
                        }
                    }
                    pa[j][q].x = pack_bf2(da[0], da[1]); pa[j][q].y = pack_bf2(da[2], da[3]);
                    pg[j][q].x = pack_bf2(dt[0], dt[1]); pg[j][q].y = pack_bf2(dt[2], dt[3]);
                }
#ifdef SPN_GEMM_TIMING
            asm volatile("" :: "v"(pa[1][3].y), "v"(pg[1][3].y));
            ESTAMP(2 + 3 * i);
#endif
#pragma unroll
            for (int part = 0; part < 2; ++part) {
                PP_SLAB_SYNC();
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        *reinterpret_cast<uint2*>(stg + rl * 128 + ((((8 * j + 2 * q + hl) ^ rl) & 15) << 3)) = part == 0 ? pa[j][q] : pg[j][q];
                PP_SLAB_SYNC();
                if (g.ws) {   // column sums of the staged block: lane l owns column l; a transposed read hands it rows 4 t .. 4 t + 3 of that column
                    typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
                    const int gq = lane >> 4, p = lane & 15;
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        const int rr = 4 * t + (p >> 2), sl = 4 * gq + (p & 3);
                        const bf16x4 w = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(stg + rr * 128 + (((sl ^ rr) & 15) << 3)));
                        csum[part] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(w, w, 0, 1), ones, csum[part], false);
                        csum[part] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(w, w, 2, 3), ones, csum[part], false);
                    }
                }
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int r = it * 8 + (lane >> 3), chunk = lane & 7;
                    uint4 val = *reinterpret_cast<const uint4*>(stg + r * 128 + (((chunk ^ (r >> 1)) & 7) << 4));
                    if (r & 1) val = uint4{val.z, val.w, val.x, val.y};
                    // every tile issues exactly 8 store instructions per block and wave (the vmcnt counts above rely on it): rows past M are
                    // dropped by the buffer's bounds check, not by the exec mask
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, val), rsD, (int)(so + (uint32_t)it * so_step8), part * I * 2, 0);
                }
            }
            so += 4u * so_step8;
            ESTAMP(3 + 3 * i);
        }
#ifdef SPN_GEMM_TIMING
        if (g.dbg && wr == 0 && wc == 0 && lane == 0) {   // wave 0 of the workgroup: epilogue stamps relative to its start (tools/duo_timeline.py)
            long long* o = g.dbg + 32 + 5 * (long)g.tx * g.ty + 13 * (long)blockIdx.x;
#pragma unroll
            for (int k = 0; k < 13; ++k) o[k] = est[k];
        }
#endif
        // the partial buffer has ceil(M / 128) rows: a wave slab that starts at or beyond M (last row tile, 0 < M % 256 <= 128) owns none
        if (g.ws && cm0 + wr * 128 < g.M) {
            float* P = reinterpret_cast<float*>(g.ws) + (long)(cm0 / 128 + wr) * (2 * I);
            P[colw + lane] = csum[0];
            P[I + colw + lane] = csum[1];
        }
    } else if constexpr (GLU != 0) {
        // u = x W^T + b leaves in its natural [value | gate] layout (the backward re-reads it), rounded to bf16 FIRST; the gated
        // output is computed from the rounded values, so it equals spn_act_fwd on the stored u bit for bit.
        // Round 4 (as the gated backward above): u and g leave through RAW BUFFER stores whose resources end at row M -- no exec-mask
        // guard, every tile issues exactly 24 store instructions per wave (the persistent walk credits them on edge tiles too); the
        // dropout hash starts from a per-block base; the gated output is staged with the XOR key on row >> 2 (64-byte rows: the 16 rows
        // of a quarter-wave then hit 16 distinct bank groups; keyed on the row itself the write was a 2-way conflict, the
        // SQ_LDS_BANK_CONFLICT cycles that only this kernel had in profiles/r03_gemm_pmc.txt).
        const __amdgpu_buffer_rsrc_t rsU = __builtin_amdgcn_make_buffer_rsrc(g.C, 0, (int)(uint32_t)((long)g.M * g.ldc * 2), 0x00020000);
        const __amdgpu_buffer_rsrc_t rsG = __builtin_amdgcn_make_buffer_rsrc(g.G, 0, (int)(uint32_t)((long)g.M * g.ldg * 2), 0x00020000);
        const int row = lane & 31, hl = lane >> 5;
        // u: row (lane >> 3) of a block, chunk lane & 7 = (value | gate half, 8 columns); g: row (lane >> 2), chunk lane & 3
        uint32_t so_u = (uint32_t)(((long)(cm0 + wr * 128 + (lane >> 3)) * g.ldc + ((lane & 7) >> 2) * g.N + cn0 + wc * 32 + (lane & 3) * 8) * 2);
        uint32_t so_g = (uint32_t)(((long)(cm0 + wr * 128 + (lane >> 2)) * g.ldg + cn0 + wc * 32 + (lane & 3) * 8) * 2);
        const uint32_t su8 = (uint32_t)(8 * g.ldc * 2), sg16 = (uint32_t)(16 * g.ldg * 2);
        const uint32_t thr16 = g.thr16;
        const float ks = g.keep_scale;
        const uint32_t hcol = __umul24((uint32_t)((cn0 + wc * 32) >> 1) + 2u * (uint32_t)hl, 0xEBCA77u);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float o[4][4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 a = f32x4{acc[i][0][4 * q], acc[i][0][4 * q + 1], acc[i][0][4 * q + 2], acc[i][0][4 * q + 3]} + bv[q];
                const f32x4 t = f32x4{acc[i][1][4 * q], acc[i][1][4 * q + 1], acc[i][1][4 * q + 2], acc[i][1][4 * q + 3]} + bv[4 + q];
                uint2 pa, pt;
                pa.x = pack_bf2(a[0], a[1]); pa.y = pack_bf2(a[2], a[3]);
                pt.x = pack_bf2(t[0], t[1]); pt.y = pack_bf2(t[2], t[3]);
                const int slot = 2 * q + hl;   // 8-byte slot of the 64-byte value half; the gate half is slots 8..15
                *reinterpret_cast<uint2*>(stg + row * 128 + (((slot ^ row) & 15) << 3)) = pa;
                *reinterpret_cast<uint2*>(stg + row * 128 + ((((8 + slot) ^ row) & 15) << 3)) = pt;
                const float ar[4] = {__uint_as_float(pa.x << 16), __uint_as_float(pa.x & 0xffff0000u), __uint_as_float(pa.y << 16), __uint_as_float(pa.y & 0xffff0000u)};
                const float tr[4] = {__uint_as_float(pt.x << 16), __uint_as_float(pt.x & 0xffff0000u), __uint_as_float(pt.y << 16), __uint_as_float(pt.y & 0xffff0000u)};
#pragma unroll
                for (int e = 0; e < 4; ++e) o[q][e] = ar[e] * (GLU == 1 ? silu_f(tr[e]) : gelu_f(tr[e]));
            }
            if (thr16) {   // the mask of spn_act_fwd / spn_act_bwd (common.h: ffn_drop_bits), hash base hoisted
                const uint32_t hb = ffn_drop_rowc(cm0 + wr * 128 + 32 * i + row, g.seed) + hcol;
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        uint32_t x = hb + (uint32_t)(4 * q + e) * 0xEBCA77u;
                        x ^= x >> 11; x = __umul24(x, 0xD35A2Du) + (x >> 8);
                        x ^= x >> 13; x = __umul24(x, 0x9E3B35u) + (x >> 9);
                        x ^= x >> 15;
                        const float s0 = o[q][2 * e] * ks, s1 = o[q][2 * e + 1] * ks;
                        o[q][2 * e] = (x & 0xffffu) >= thr16 ? s0 : 0.f;
                        o[q][2 * e + 1] = (x >> 16) >= thr16 ? s1 : 0.f;
                    }
            }
            PP_SLAB_SYNC();
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int r = it * 8 + (lane >> 3), chunk = lane & 7;
                uint4 val = *reinterpret_cast<const uint4*>(stg + r * 128 + (((chunk ^ (r >> 1)) & 7) << 4));
                if (r & 1) val = uint4{val.z, val.w, val.x, val.y};
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, val), rsU, (int)(so_u + (uint32_t)it * su8), 0, 0);
            }
            so_u += 4u * su8;
            PP_SLAB_SYNC();
            // the gated output through the same slab: 64-byte rows, 8 slots of 8 bytes, slot index XOR (row >> 2)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                uint2 go;
                go.x = pack_bf2(o[q][0], o[q][1]); go.y = pack_bf2(o[q][2], o[q][3]);
                *reinterpret_cast<uint2*>(stg + row * 64 + ((((2 * q + hl) ^ (row >> 2)) & 7) << 3)) = go;
            }
            PP_SLAB_SYNC();
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int r = it * 16 + (lane >> 2), chunk = lane & 3;
                uint4 val = *reinterpret_cast<const uint4*>(stg + r * 64 + (((chunk ^ (r >> 3)) & 3) << 4));
                if ((r >> 2) & 1) val = uint4{val.z, val.w, val.x, val.y};
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, val), rsG, (int)(so_g + (uint32_t)it * sg16), 0, 0);
            }
            so_g += 2u * sg16;
            PP_SLAB_SYNC();
        }
    } else {
    if constexpr (ES == 4) {
        // fp32 output with a residual (y = residual + ...: the attention and feed-forward output projections), interior tile: the
        // residual block of every pass (32 rows x 32 columns = 128-byte row segments) arrives by LDS DMA into this wave's 16 KiB of the
        // dead operand ring -- whole cache lines, no registers, four passes in flight -- instead of 32-byte pieces per lane in the
        // accumulator layout.  VM counter, oldest first (D = 4 DMA instructions, S = the 4 stores of a pass):
        //   D0 D1 D2 D3 | pass 0: S0 D4 | pass 1: S1 D5 | pass 2: S2 D6 | pass 3: S3 D7 | S4 | S5 | S6 | S7
        if (ring && g.residual && lead && !g.accumulate && cm0 + PP_BM <= g.M && cn0 + PP_BN <= g.N && (long)g.M * g.ldr * 4 < (1L << 31)) {
            const int row = lane & 31, hl = lane >> 5;
            float rsv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) rsv[i] = g.rowmask ? (g.rowmask[cm0 + wr * 128 + 32 * i + row] ? 1.f : 0.f) : 1.f;
            const u32x4 rsR = spn_buffer_rsrc(g.residual, 0x7fffffffu);
            const uint32_t ring_w = spn_lds_addr(ring);
            uint32_t vo[4];
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int r = it * 8 + (lane >> 3), c = (lane & 7) ^ (r & 7);
                vo[it] = (uint32_t)((r * g.ldr + c * 4) * 4);
            }
            auto issue_r = [&](int k) __attribute__((always_inline)) {   // pass k = (row block k >> 1, column half k & 1) -> ring slot k & 3
                const uint32_t soff = (uint32_t)(((long)(cm0 + wr * 128 + 32 * (k >> 1)) * g.ldr + cn0 + wc * 64 + 32 * (k & 1)) * 4);
                const uint32_t dst = ring_w + (uint32_t)((k & 3) * 4096);
                spn_dma16x2(rsR, dst, vo[0], vo[1], soff);
                spn_dma16x2(rsR, dst + 2048u, vo[2], vo[3], soff);
            };
            issue_r(0); issue_r(1); issue_r(2); issue_r(3);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int i = k >> 1, jh = k & 1;
                if (k == 0 || k == 7) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
                else if (k == 1 || k == 6) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                else if (k == 2 || k == 5) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
                f32x4 res[4];
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    res[q] = *reinterpret_cast<const f32x4*>(ring + (k & 3) * 4096 + row * 128 + ((((2 * q + hl) ^ row) & 7) << 4));
                if (k + 4 < 8) {   // the slot is free once the reads above have returned
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    issue_r(k + 4);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 v = (f32x4{acc[i][jh][4 * q], acc[i][jh][4 * q + 1], acc[i][jh][4 * q + 2], acc[i][jh][4 * q + 3]} * g.alpha + bv[jh * 4 + q]) * rsv[i] + res[q];
                    const int colb = (8 * q + hl * 4) * 4;
                    *reinterpret_cast<f32x4*>(stg + row * 128 + ((((colb >> 4) ^ row) & 7) << 4)) = v;
                }
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int r = it * 8 + (lane >> 3), chunk = lane & 7;
                    const uint4 val = *reinterpret_cast<const uint4*>(stg + r * 128 + (((chunk ^ r) & 7) << 4));
                    *reinterpret_cast<uint4*>(C + (long)(cm0 + wr * 128 + 32 * i + r) * g.ldc + cn0 + wc * 64 + jh * 32 + chunk * 4) = val;
                }
            }
            return;
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = lane & 31;
        const int m = cm0 + wr * 128 + 32 * i + row;
        const float rs = g.rowmask ? (g.rowmask[min(m, g.M - 1)] ? 1.f : 0.f) : 1.f;
#pragma unroll
        for (int jh = 0; jh < 2 / JP; ++jh) {
#pragma unroll
            for (int jj = 0; jj < JP; ++jj) {
                const int j = jh * JP + jj;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int n = cn0 + wc * 64 + 32 * j + 8 * q + (lane >> 5) * 4;
                    f32x4 v = (f32x4{acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]} * g.alpha + bv[j * 4 + q]) * rs;
                    if (g.residual && lead && m < g.M && n < g.N) v += *reinterpret_cast<const f32x4*>(g.residual + (long)m * g.ldr + n);
                    const int colb = (32 * jj + 8 * q + (lane >> 5) * 4) * ES;
                    if constexpr (ES == 4) {
                        *reinterpret_cast<f32x4*>(stg + row * ROWB + ((((colb >> 4) ^ row) & (CPR - 1)) << 4)) = v;
                    } else {
                        // 8-byte pieces: the 32 rows of a half-wave share one column, so the XOR key works on 8-byte slots with 4 row
                        // bits (rows r and r + 16 collide, which 256 bytes per half-wave cannot avoid); keyed on 16-byte chunks
                        // with 3 row bits the write was a 4-way bank conflict (SQ_LDS_BANK_CONFLICT: 5 k cycles per tile)
                        uint2 pk; pk.x = pack_bf2(v[0], v[1]); pk.y = pack_bf2(v[2], v[3]);
                        *reinterpret_cast<uint2*>(stg + row * ROWB + ((((colb >> 3) ^ row) & 15) << 3)) = pk;
                    }
                }
            }
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int r = it * 8 + (lane >> 3), chunk = lane & 7;
                uint4 val;
                if constexpr (ES == 4) {
                    val = *reinterpret_cast<const uint4*>(stg + r * ROWB + (((chunk ^ r) & (CPR - 1)) << 4));
                } else {   // slots 2c, 2c+1 of row r live in chunk c ^ (r >> 1), swapped when r is odd
                    val = *reinterpret_cast<const uint4*>(stg + r * ROWB + (((chunk ^ (r >> 1)) & (CPR - 1)) << 4));
                    if (r & 1) val = uint4{val.z, val.w, val.x, val.y};
                }
                const int mo = cm0 + wr * 128 + 32 * i + r, no = cn0 + wc * 64 + jh * JP * 32 + chunk * (16 / ES);
                if (mo >= g.M || no >= g.N) continue;   // edge tiles: N is a multiple of 8, so a 16-byte chunk is in or out as a whole
                OutT* dst = C + (long)mo * g.ldc + no;
                if constexpr (ES == 4) {
                    f32x4 v = __builtin_bit_cast(f32x4, val);
                    if (g.accumulate) v += *reinterpret_cast<const f32x4*>(dst);
                    *reinterpret_cast<f32x4*>(dst) = v;
                } else {
                    *reinterpret_cast<uint4*>(dst) = val;
                }
            }
        }
    }
    }   // !GLU
}

// GLU: 0 = plain GEMM; 1 = SiLU, 2 = GELU gated epilogue (TA = TB = false, bf16 out)
// PERSIST: the launch has fewer blocks than tiles and every block walks the tile ids block, block + stride, ...: no workgroup launch
// between tiles, the next tile's first half-tiles are requested before the epilogue and the stores of the epilogue drain under the next
// main loop (its first waits credit them: the VM counter retires in order).
//
// The LDS DMA is issued through inline asm (spn_dma16x2, common.h): through the builtin hipcc fenced every ds_read_b64_tr_b16 (and the
// LDS staging of the epilogue) behind `s_waitcnt vmcnt(0)`, which drained the four half-tiles in flight once per phase in every
// kernel with a transposed operand (dX and dW GEMMs), and with the descriptors pushed out of the SGPR file (fp32-output
// instantiations) wrapped each DMA instruction in a readfirstlane waterfall loop.  The K loop is peeled into a steady-state body (every
// phase issues, every wait is vmcnt(8): no scalar branches between the barriers), the last-but-one and the last K tile.
template <bool TA, bool TB, typename OutT, int GLU = 0, bool PERSIST = false>
__global__ __launch_bounds__(512) void gemm_pp_kernel(GemmArgs g) {
    constexpr bool GLUF = (GLU == 1 || GLU == 2);   // gated FORWARD epilogue: B half-tiles are value / gate rows, 128 outputs per tile
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // slot of half-tile kind c (0 AX, 1 BX, 2 BY, 3 AY) of K tile t
#define PP_SLOT(c_, t_) (smem + ((c_) * 2 + ((t_) & 1)) * PP_HALF)
    const int tid = threadIdx.x;
    int lane = tid & 63;   // PERSIST launders it at every tile boundary (see the tile loop)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
#ifdef SPN_GEMM_TIMING
    const long long t_start = __builtin_amdgcn_s_memtime();
    long long t_loop_end = t_start;
#endif
    // the tile grid is g.tx x g.ty; a launch of exactly that many blocks runs one tile per block, a PERSIST one (256 blocks, one per CU)
    // walks the tile ids block, block + 256, ...
    const int nwg = g.tx * g.ty, stride = gridDim.x * gridDim.y;
    int first = blockIdx.y * gridDim.x + blockIdx.x, zid = blockIdx.z;
    if (g.slice_xcd) {   // split-K with a multiple of 8 slices: XCD k (linear id % 8) runs whole K slices, so the A / B panels of a slice
                         // are fetched into ONE L2 (each B panel is shared by every m-tile of the slice)
        const int L = first + nwg * (int)blockIdx.z;
        zid = (L & 7) + 8 * (L / (8 * nwg));
        first = (L >> 3) % nwg;
    }
    // tile id -> tile: XCD-contiguous (id % 8 = XCD), then column groups of `ngroup` n-tiles, m-tiles down each group (see
    // gemm_kernel): the 32 tiles an XCD runs together share few B panels and few A panels
    auto tile_of = [&](int id, int& tm, int& tn) {
        const int q = nwg >> 3, r = nwg & 7, xcd = id & 7;
        const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
        const int G = g.ngroup, mt = g.ty, per = G * mt;
        const int c = wg / per, within = wg - c * per;
        const int gw = min(G, g.tx - c * G);
        tm = (within / gw) * PP_BM;
        tn = (c * G + within % gw) * (GLUF ? PP_BN / 2 : PP_BN);
    };
    const bool split = g.splitk > 1;
    const bf16_t* A = g.A + (split ? 0 : (long)zid * g.sA);
    const bf16_t* B = g.B + (split ? 0 : (long)zid * g.sB);
    const int nt_all = g.K / PP_BK;
    const int t_begin = split ? zid * g.kt_per_split : 0;
    const int nt = split ? min(nt_all - t_begin, g.kt_per_split) : nt_all;
    if (nt <= 0 || first >= nwg) return;
    const int kbase = t_begin * PP_BK;
    int m0, n0;
    tile_of(first, m0, n0);
    int extra = 0;              // PERSIST: the previous tile's epilogue left its stores between S[5] and S[6] of this tile in the VM counter
    const u32x4 rsA = spn_buffer_rsrc(A, 0x7fffffffu), rsB = spn_buffer_rsrc(B, 0x7fffffffu);
    const uint32_t ring = spn_lds_addr(smem) + (uint32_t)wave * 2048u;   // this wave's 2 KiB of every half-tile slot
    uint32_t vo[4][2];   // [kind][piece]
    auto set_tile = [&](int tm, int tn) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            vo[0][i] = pp_voffset<TA, 64>(g.lda, tm, g.M, 0, wave, lane, i);
            vo[1][i] = GLUF ? pp_voffset_lin(g.ldb, tn, wave, lane, i) : pp_voffset<TB, 32>(g.ldb, tn, g.N, 0, wave, lane, i);
            vo[2][i] = GLUF ? pp_voffset_lin(g.ldb, g.N + tn, wave, lane, i) : pp_voffset<TB, 32>(g.ldb, tn, g.N, 1, wave, lane, i);
            vo[3][i] = pp_voffset<TA, 64>(g.lda, tm, g.M, 1, wave, lane, i);
        }
    };
    set_tile(m0, n0);
    if (PERSIST && g.stagger > 0) {   // tuning aid: CU slot s of an XCD starts s * stagger cycles late (tile boundaries spread in time)
        const long long t_go = __builtin_amdgcn_s_memtime() + (long long)((first >> 3) & 31) * g.stagger;
        while (__builtin_amdgcn_s_memtime() < t_go) __builtin_amdgcn_s_sleep(8);
    }
    const uint32_t kstepA = TA ? (uint32_t)PP_BK * g.lda * 2u : (uint32_t)PP_BK * 2u;   // bytes per K tile along the operand
    const uint32_t kstepB = TB ? (uint32_t)PP_BK * g.ldb * 2u : (uint32_t)PP_BK * 2u;
    const uint32_t kbaseA = TA ? (uint32_t)kbase * g.lda * 2u : (uint32_t)kbase * 2u;
    const uint32_t kbaseB = TB ? (uint32_t)kbase * g.ldb * 2u : (uint32_t)kbase * 2u;
    auto issue = [&](int c, int t) __attribute__((always_inline)) {   // c is a compile-time constant at every call site
        const bool isA = (c == 0 || c == 3);
        const uint32_t soff = isA ? kbaseA + (uint32_t)t * kstepA : kbaseB + (uint32_t)t * kstepB;
        spn_dma16x2(isA ? rsA : rsB, ring + (uint32_t)((c * 2 + (t & 1)) * PP_HALF), vo[c][0], vo[c][1], soff);
    };
    // S[p+6] has been issued in phase p; everything up to S[p+2] must have landed: 4 half-tiles = 8 DMA instructions stay in flight
    // (fewer at the end of the stream).  PERSIST, first K tile: the stores of the previous tile's epilogue are younger than S[<= 5].
    constexpr int NS = sizeof(OutT) == 2 ? 16 : 32;   // store instructions per wave of one interior-tile epilogue
    constexpr int NSX = GLUF ? 24 : NS;               // (gated: 16 stores of u + 8 of g)
#define PP_VMWAIT(n_) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(n_) : "memory")
// wave priority in the MFMA part / in the read part of a phase (tuning aid of round 5: PP_PRIO_MFMA=0 PP_PRIO_READ=1 swaps them)
#ifndef PP_PRIO_MFMA
#define PP_PRIO_MFMA 1
#endif
#ifndef PP_PRIO_READ
#define PP_PRIO_READ 0
#endif
#define PP_SYNC_MFMA_BEGIN()                                   \
    TSTAMP(2);                                                  \
    __builtin_amdgcn_sched_barrier(0);                          \
    __builtin_amdgcn_s_barrier();                               \
    TSTAMP(3);                                                  \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          \
    TSTAMP(4);                                                  \
    __builtin_amdgcn_sched_barrier(0);                          \
    __builtin_amdgcn_s_setprio(PP_PRIO_MFMA)
#define PP_SYNC_MFMA_END()                                     \
    __builtin_amdgcn_s_setprio(PP_PRIO_READ);                   \
    __builtin_amdgcn_sched_barrier(0);                          \
    TSTAMP(5);                                                  \
    __builtin_amdgcn_s_barrier();                               \
    TSTAMP(6)

    // prologue: S[0..5] = K tile 0 and AX, BX of K tile 1 (nt >= 4)
    auto prologue = [&]() __attribute__((always_inline)) {
        issue(0, 0); issue(1, 0); issue(2, 0); issue(3, 0);
        issue(0, 1); issue(1, 1);
    };
    prologue();
    for (int vid = first; vid < nwg; vid += stride) {
    // PERSIST: an opaque copy of the lane id per tile and per part (K loop / epilogue).  Without it every lane-derived LDS address of the
    // K loop stays in a register through the epilogue and every epilogue constant through the K loop: 30-60 spilled VGPRs whose
    // scratch reloads (a memory round trip each, on the VM counter) cost more than the workgroup launches the persistent walk saves.
    if (PERSIST) asm volatile("" : "+v"(lane));
    f32x16 acc[4][2];   // acc[2h + i][j]: rows 64h + 32i.., columns 32j..
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (PERSIST && extra) PP_VMWAIT(8 + NSX); else PP_VMWAIT(8);   // AX(0), BX(0) have landed
    __builtin_amdgcn_s_barrier();
    if (wave >= 4) __builtin_amdgcn_s_barrier();   // second group runs one barrier behind the first
    bf16x8 apre[4];   // first 32 rows of the current tile's A0 fragments, read one phase ahead
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) apre[ks] = pp_read_frag<TA>(PP_SLOT(0, 0), wr * 64, ks, lane);

#ifdef SPN_GEMM_TIMING
    long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long tprev = __builtin_amdgcn_s_memtime();
#endif
    // one K tile = four phases.  MODE 0: steady state (t <= nt - 3), 1: t = nt - 2, 2: t = nt - 1, 3: t = 0 of a PERSIST launch
    // (steady, but the waits may have to credit the previous epilogue's stores)
    auto ktile = [&](auto mode_c, const int t) __attribute__((always_inline)) {
        constexpr int MODE = decltype(mode_c)::value;
        constexpr bool ISSUE_01 = MODE != 2, ISSUE_23 = (MODE == 0 || MODE == 3);
        bf16x8 af[2][4], b0[4], b1[4];
        // ---- p0: reads B0 and the second 32 rows of A0 (the first 32 were read in the previous tile's read-free p3); issues BY(t+1) ----
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) b0[ks] = pp_read_frag<TB>(PP_SLOT(1, t), wc * 32, ks, lane);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) af[1][ks] = pp_read_frag<TA>(PP_SLOT(0, t), wr * 64 + 32, ks, lane);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) af[0][ks] = apre[ks];
        TSTAMP(0);
        if (ISSUE_01) issue(2, t + 1);
        TSTAMP(1);
        if (MODE == 3) { if (extra) PP_VMWAIT(PP_STEADY_VM + NSX); else PP_VMWAIT(PP_STEADY_VM); }
        else if (MODE == 2) PP_VMWAIT(2);
        else PP_VMWAIT(PP_STEADY_VM);
        PP_SYNC_MFMA_BEGIN();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b0[ks], af[i][ks], acc[i][0], 0, 0, 0);
        }
        PP_SYNC_MFMA_END();
        // ---- p1: reads B1; issues AY(t+1) = S[4t+7] ----
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) b1[ks] = pp_read_frag<TB>(PP_SLOT(2, t), wc * 32, ks, lane);
        TSTAMP(0);
        if (ISSUE_01) issue(3, t + 1);
        TSTAMP(1);
        if (MODE == 3) { if (extra) PP_VMWAIT(PP_STEADY_VM + NSX); else PP_VMWAIT(PP_STEADY_VM); }
        else if (MODE == 2) PP_VMWAIT(0);
        else PP_VMWAIT(PP_STEADY_VM);
        PP_SYNC_MFMA_BEGIN();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b1[ks], af[i][ks], acc[i][1], 0, 0, 0);
        }
        PP_SYNC_MFMA_END();
        // ---- p2: reads A1; issues AX(t+2) = S[4t+8] ----
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) af[i][ks] = pp_read_frag<TA>(PP_SLOT(3, t), wr * 64 + 32 * i, ks, lane);
        TSTAMP(0);
        if (ISSUE_23) issue(0, t + 2);
        TSTAMP(1);
        if (MODE == 3) { if (extra) PP_VMWAIT(PP_STEADY_VM + NSX); else PP_VMWAIT(PP_STEADY_VM); }
        else if (MODE == 1) PP_VMWAIT(6);
        else if (MODE == 0) PP_VMWAIT(PP_STEADY_VM);
        PP_SYNC_MFMA_BEGIN();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[2 + i][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b1[ks], af[i][ks], acc[2 + i][1], 0, 0, 0);
        }
        PP_SYNC_MFMA_END();
        // ---- p3: B0 is still in registers; reads the first 32 rows of the NEXT tile's A0 (AX(t+1) = S[4t+4] was published by the
        //      barrier of phase 4t+2): the reads are 8 / 4 / 8 / 4 per phase instead of 12 / 4 / 8 / 0; issues BX(t+2) = S[4t+9] ----
        if (MODE != 2) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) apre[ks] = pp_read_frag<TA>(PP_SLOT(0, t + 1), wr * 64, ks, lane);
        }
        TSTAMP(0);
        if (ISSUE_23) issue(1, t + 2);
        TSTAMP(1);
        if (MODE == 3) { if (extra) PP_VMWAIT(PP_STEADY_VM + NSX); else PP_VMWAIT(PP_STEADY_VM); }
        else if (MODE == 1) PP_VMWAIT(4);
        else if (MODE == 0) PP_VMWAIT(PP_STEADY_VM);
        PP_SYNC_MFMA_BEGIN();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[2 + i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b0[ks], af[i][ks], acc[2 + i][0], 0, 0, 0);
        }
        PP_SYNC_MFMA_END();
    };
    {
        int t = 0;
        if (PERSIST) { ktile(std::integral_constant<int, 3>{}, 0); t = 1; }
        for (; t < nt - 2; ++t) ktile(std::integral_constant<int, 0>{}, t);
        ktile(std::integral_constant<int, 1>{}, nt - 2);
        ktile(std::integral_constant<int, 2>{}, nt - 1);
    }
    if (wave < 4) __builtin_amdgcn_s_barrier();   // balances the second group's extra barrier
#ifdef SPN_GEMM_TIMING
    t_loop_end = __builtin_amdgcn_s_memtime();
    if (g.dbg && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && (wave == 0 || wave == 4) && lane == 0)
        for (int k = 0; k < 8; ++k) g.dbg[(wave >> 2) * 8 + k] = seg[k];
#endif

    // ---- epilogue: through the wave's own 4 KiB slab of the 32 KiB the ring leaves free, so that C leaves as whole 128-byte row
    //      segments (see gemm_kernel) while (PERSIST) the ring already receives the NEXT tile's first half-tiles ----
    if (PERSIST) asm volatile("" : "+v"(lane));
    // The epilogue reads its arguments (output pointers, strides, bias / residual / mask, dropout words) from the kernel-argument
    // segment AGAIN, through a pointer the optimiser cannot see through: scalar loads that hit the constant cache.  Taken from `g`
    // they are live across the K loop, which has no SGPRs to spare (buffer descriptors, ring addresses, counters): hipcc parked them
    // in scratch memory and fetched them back with vector loads + s_waitcnt vmcnt(0) in the middle of the epilogue.
    GemmKernargPtr gk = (GemmKernargPtr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(gk));
    const GemmArgs ge = kernarg_copy(gk);   // scalar loads of the fields the epilogue uses; the rest is dead code
    OutT* C = reinterpret_cast<OutT*>(ge.C) + (long)zid * ge.sC;
    const bool lead = !split || zid == 0;
    char* stg = smem + 8 * PP_HALF + wave * 4096;
    f32x4 bv[8];
#pragma unroll
    for (int jq = 0; jq < 8; ++jq) {
        // GLU: bv[0..3] = bias of the value columns, bv[4..7] = bias of the gate columns (N = I is a multiple of 128: no edge)
        const int bn = GLUF ? (jq >> 2) * ge.N + n0 + wc * 32 + 8 * (jq & 3) + (lane >> 5) * 4
                            : min(n0 + wc * 64 + 32 * (jq >> 2) + 8 * (jq & 3) + (lane >> 5) * 4, ge.N - 4);
        bv[jq] = (ge.bias && lead) ? *reinterpret_cast<const f32x4*>(ge.bias + bn) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();   // every wave is done with the operand ring
    const int cm0 = m0, cn0 = n0;
    if (PERSIST) {
        const bool has_next = vid + stride < nwg;
        if (has_next) {    // the next tile's prologue DMA goes out now: its latency hides behind this epilogue
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the bias loads above; nothing else is outstanding)
            tile_of(vid + stride, m0, n0);
            set_tile(m0, n0);
            prologue();
        }
        // interior tiles issue exactly NSX store instructions per wave; edge tiles fewer: no credit for them (a stronger wait).  The gated
        // epilogue stores through bounds-checked buffer resources: every tile issues all 24, rows past M are dropped by the hardware
        extra = GLUF ? (has_next ? 1 : 0)
                    : ((has_next && cm0 + PP_BM <= ge.M && cn0 + PP_BN <= ge.N && !ge.residual && !ge.accumulate && !ge.rowmask) ? 1 : 0);
    }
    pp_store_tile<OutT, GLU>(ge, acc, bv, C, lead, cm0, cn0, wr, wc, lane, stg, PERSIST ? nullptr : smem + wave * 16384);
    if (!PERSIST) break;
    }
#undef PP_SYNC_MFMA_BEGIN
#undef PP_SYNC_MFMA_END
#undef PP_VMWAIT
#undef PP_SLOT
#ifdef SPN_GEMM_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // stores acknowledged
    const long long t_end = __builtin_amdgcn_s_memtime();
    if (g.dbg && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && wave == 0 && lane == 0) {
        g.dbg[16] = t_loop_end - t_start;   // prologue + main loop
        g.dbg[17] = t_end - t_loop_end;     // epilogue until the stores are acknowledged
    }
#endif
}

// (the one-wave-per-SIMD kernels of round 5 -- 128x128 wave tiles, LDS-DMA or register-staged operands, 0.85-0.99x of the ping-pong
// kernel on every shape of the step -- live in tools/variants/ and are compiled into variant builds only)
#ifdef SPN_GEMM_OW_VARIANT
#include "../../tools/variants/gemm_ow_kernels.inc"
#endif

// ---- 256x128 "duo" kernel: TWO independent 4-wave workgroups per CU -----------------------------------------------------
// The projections with d_model = 512 on the contraction side (K = 512: 8 K-tiles of 64) spend a third of a 256x256 tile's life in the
// prologue (first DMA latency) and the epilogue (128-256 KiB of C per CU through a 64 B/clk store path) with the matrix pipe idle, and
// one workgroup per CU has nothing to overlap them with.  Here a workgroup is 4 waves (one per SIMD) with a 256x128 tile, <= 80 KiB of
// LDS and <= 256 registers, so TWO of them are resident per CU and run out of phase: while one drains its accumulators and refills
// its pipeline, the other one owns the matrix pipe; inside the K loop the two waves of a SIMD interleave LDS reads and MFMAs the way
// the ping-pong kernel's two groups do, without a barrier between them.
//   * K tile 32, three stages of (A 256x32 = 16 KiB, B 128x32 = 8 KiB): stage t+2 is requested right after the barrier that opens
//     iteration t (it overwrites stage t-1, which every wave has finished reading), so two stages = 48 KiB per workgroup are in flight
//     while stage t is multiplied; ONE barrier per K tile (512 MFMA cycles per wave)
//   * a wave owns 128x64 of C (acc[4][2] 32x32 blocks, v_mfma_f32_32x32x16_bf16, operands swapped as in the ping-pong kernel) and shares
//     its epilogue (pp_store_tile): after the loop the ring is dead and each wave stages through its own 4 KiB of it
//   * K-contiguous stage image [rows][32 k]: 64-byte rows, 16-byte chunk index XOR (row >> 2) & 3 (the 16 rows of a ds_read_b128 lane
//     group then cover all 16 slots of the 256-byte bank row); N-contiguous B image [32 k][128 n] as the ping-pong kernel's
// A must be K-contiguous (TA = false: forward projections and input gradients); M / N edges by clamped loads + guarded stores.
constexpr int DU_BM = 256, DU_BN = 128, DU_BK = 32, DU_STAGES = 3;
constexpr int DU_A_BYTES = DU_BM * DU_BK * 2, DU_B_BYTES = DU_BN * DU_BK * 2, DU_STAGE_BYTES = DU_A_BYTES + DU_B_BYTES;
constexpr int DU_LDS_BYTES = DU_STAGES * DU_STAGE_BYTES;   // 72 KiB

__device__ __forceinline__ int du_kc_off(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4); }

template <bool TB, typename OutT, int GLU = 0>
__global__ __launch_bounds__(256, 2) void gemm_duo_kernel(GemmArgs g) {
    constexpr bool GLUF = (GLU == 1 || GLU == 2);   // gated forward (value / gate rows of W paired in one tile); GLU >= 3: gated backward epilogue
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
#ifdef SPN_GEMM_TIMING
    const long long tl_start = __builtin_amdgcn_s_memrealtime();   // 100 MHz, the same clock on every CU (tools/duo_timeline.py)
#endif
    // tile order as in the ping-pong kernel: XCD-contiguous, column groups of `ngroup` n-tiles, m-tiles down each group
    int m0, n0;
    {
        const int nwg = g.tx * g.ty, id = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = id & 7;
        const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
        const int G = g.ngroup, per = G * g.ty;
        const int c = wg / per, within = wg - c * per;
        const int gw = min(G, g.tx - c * G);
        m0 = (within / gw) * DU_BM;
        n0 = (c * G + within % gw) * (GLUF ? DU_BN / 2 : DU_BN);
    }
    const int nt = g.K / DU_BK;
    const u32x4 rsA = spn_buffer_rsrc(g.A, 0x7fffffffu), rsB = spn_buffer_rsrc(g.B, 0x7fffffffu);
    const uint32_t lds0 = spn_lds_addr(smem);
    // per-lane source offsets of this wave's DMA pieces (loop invariant): LDS chunk L of a stage image <- inverse-swizzled source chunk
    uint32_t voA[4], voB[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int L = (wave * 4 + i) * 64 + lane, row = L >> 2, kc = (L & 3) ^ ((row >> 2) & 3);
        voA[i] = (uint32_t)(((long)min(m0 + row, g.M - 1) * g.lda + kc * 8) * 2);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int L = (wave * 2 + i) * 64 + lane;
        if (!TB) {
            const int row = L >> 2, kc = (L & 3) ^ ((row >> 2) & 3);
            // gated: image rows 0..63 are the VALUE rows n0.. of W, rows 64..127 the GATE rows I + n0..
            const int src = GLUF ? (row < 64 ? n0 + row : g.N + n0 + row - 64) : min(n0 + row, g.N - 1);
            voB[i] = (uint32_t)(((long)src * g.ldb + kc * 8) * 2);
        } else {
            const int krow = L >> 4, rc = (L & 15) ^ pp_rc_swz(krow);
            voB[i] = (uint32_t)(((long)krow * g.ldb + min(n0 + rc * 8, g.N - 8)) * 2);
        }
    }
    auto issue = [&](int t) {   // asm DMA (common.h): the compiler neither counts it nor fences the transposed reads behind it
        const uint32_t dst = lds0 + (uint32_t)((t % DU_STAGES) * DU_STAGE_BYTES);
        const uint32_t k0 = (uint32_t)t * DU_BK;
        const uint32_t soA = k0 * 2u, soB = TB ? k0 * (uint32_t)g.ldb * 2u : k0 * 2u;
        spn_dma16x2(rsA, dst + (uint32_t)(wave * 4) * 1024u, voA[0], voA[1], soA);
        spn_dma16x2(rsA, dst + (uint32_t)(wave * 4 + 2) * 1024u, voA[2], voA[3], soA);
        spn_dma16x2(rsB, dst + DU_A_BYTES + (uint32_t)(wave * 2) * 1024u, voB[0], voB[1], soB);
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    issue(0);
    if (nt > 1) issue(1);
    for (int t = 0; t < nt; ++t) {
        // this wave's pieces of stage t have landed (stage t+1, 6 instructions, may stay in flight) ...
        if (t + 1 < nt) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // ... the barrier publishes everybody's pieces, and proves that every wave is done reading stage t-1 (its fragment reads were
        // waited for before its MFMAs of iteration t-1)
        __builtin_amdgcn_s_barrier();
        if (t + 2 < nt) issue(t + 2);
        const char* sa = smem + (t % DU_STAGES) * DU_STAGE_BYTES;
        const char* sb = sa + DU_A_BYTES;
        bf16x8 af[4][2], bf[2][2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int rb = GLUF ? 64 * j + 32 * wc : 64 * wc + 32 * j;
                if (!TB) bf[j][ks] = *reinterpret_cast<const bf16x8*>(sb + du_kc_off(rb + (lane & 31), ks * 2 + (lane >> 5)));
                else bf[j][ks] = pp_read_frag<true>(sb, rb, ks, lane);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
                af[i][ks] = *reinterpret_cast<const bf16x8*>(sa + du_kc_off(wr * 128 + 32 * i + (lane & 31), ks * 2 + (lane >> 5)));
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[j][ks], af[i][ks], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    }

    OutT* C = reinterpret_cast<OutT*>(g.C);
    f32x4 bv[8];
#pragma unroll
    for (int jq = 0; jq < 8; ++jq) {
        const int bn = GLUF ? (jq >> 2) * g.N + n0 + wc * 32 + 8 * (jq & 3) + (lane >> 5) * 4
                            : min(n0 + wc * 64 + 32 * (jq >> 2) + 8 * (jq & 3) + (lane >> 5) * 4, g.N - 4);
        bv[jq] = g.bias ? *reinterpret_cast<const f32x4*>(g.bias + bn) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();   // every wave is done with the operand ring: its first 16 KiB become the four staging slabs
#ifdef SPN_GEMM_TIMING
    const long long tl_loop = __builtin_amdgcn_s_memrealtime();
#endif
    // gated backward: the four staging slabs, then 16 KiB per wave for the u blocks that arrive by LDS DMA (80 KiB in all)
    pp_store_tile<OutT, GLU>(g, acc, bv, C, true, m0, n0, wr, wc, lane, smem + wave * 4096, GLU >= 3 ? smem + 16384 + wave * 16384 : nullptr);
#ifdef SPN_GEMM_TIMING
    if (g.dbg) {   // per workgroup: start, end of the main loop, end of the epilogue (wave 0, stores acknowledged), HW_ID, XCC_ID
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const long long tl_end = __builtin_amdgcn_s_memrealtime();
        if (tid == 0) {
            long long* o = g.dbg + 32 + 5 * (long)blockIdx.x;
            o[0] = tl_start; o[1] = tl_loop; o[2] = tl_end;
            o[3] = __builtin_amdgcn_s_getreg((31 << 11) | 4); o[4] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        }
    }
#endif
}

// ---- 256x128 "duo8" kernel: the gated backward with EIGHT waves per workgroup, two workgroups per CU -------------------------------
// tools/duo_timeline.py (per-workgroup stamps of the duo kernel above, profiles/r04_duo_timeline.txt) showed where the gated backward
// spends its time: the two workgroups of a CU do run in anti-phase (75 % of a workgroup's epilogue lies under the other's main loop), but
// the epilogue is ONE wave per SIMD issuing ~4 200 dependent VALU instructions (4 ns each: a SIMD issues for a single wave at ~0.6 of
// its two-wave rate, tools/issue_probe.hip) and takes as long as the main loop.  Here a workgroup is 8 waves with 64x64 wave tiles
// (acc[2][2]: 64 registers), <= 128 VGPRs, so FOUR waves share a SIMD: two of one workgroup in their epilogue (half the rows each,
// issuing alternately) while the other workgroup's two run the main loop.  Same 256x128 tile, K tile 32, three LDS-DMA stages (72 KiB).
//   * stage pieces: A 16 x 1 KiB (2 per wave), B 8 x 1 KiB (1 per wave): 3 DMA instructions per wave and stage
//   * epilogue (GLU >= 3 only): a wave's two 32-row blocks go one after the other through its OWN 8 KiB of the dead ring -- the u block
//     (value 4 KiB | gate 4 KiB) arrives by LDS DMA, every lane reads its 8-byte pieces, and writes du back IN PLACE (same bytes, same
//     swizzle: no second slab, no cross-lane hazard); the rows then leave as whole 128-byte segments through bounds-checked buffer
//     stores, the next block's DMA is requested as soon as the read-back has returned.  Arithmetic exactly as pp_store_tile<.., 3>.
template <int ACT>
__global__ __launch_bounds__(512, 2) void gemm_duo8_glu_bwd_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;           // 4 x 2 waves: rows 64 wr .., columns 64 wc ..
#ifdef SPN_GEMM_TIMING
    const long long tl_start = __builtin_amdgcn_s_memrealtime();
#endif
    int m0, n0;
    {
        const int nwg = g.tx * g.ty, id = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = id & 7;
        const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
        const int G = g.ngroup, per = G * g.ty;
        const int c = wg / per, within = wg - c * per;
        const int gw = min(G, g.tx - c * G);
        m0 = (within / gw) * DU_BM;
        n0 = (c * G + within % gw) * DU_BN;
    }
    const int nt = g.K / DU_BK;
    const u32x4 rsA = spn_buffer_rsrc(g.A, 0x7fffffffu), rsB = spn_buffer_rsrc(g.B, 0x7fffffffu);
    const uint32_t lds0 = spn_lds_addr(smem);
    uint32_t voA[2], voB;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int L = (wave * 2 + i) * 64 + lane, row = L >> 2, kc = (L & 3) ^ ((row >> 2) & 3);
        voA[i] = (uint32_t)(((long)min(m0 + row, g.M - 1) * g.lda + kc * 8) * 2);
    }
    {   // B stored [K][N] (N-contiguous): image [32 k][128 n] as the ping-pong kernel's
        const int L = wave * 64 + lane;
        const int krow = L >> 4, rc = (L & 15) ^ pp_rc_swz(krow);
        voB = (uint32_t)(((long)krow * g.ldb + min(n0 + rc * 8, g.N - 8)) * 2);
    }
    auto issue = [&](int t) {
        const uint32_t dst = lds0 + (uint32_t)((t % DU_STAGES) * DU_STAGE_BYTES);
        const uint32_t k0 = (uint32_t)t * DU_BK;
        spn_dma16x2(rsA, dst + (uint32_t)(wave * 2) * 1024u, voA[0], voA[1], k0 * 2u);
        spn_dma16(rsB, dst + DU_A_BYTES + (uint32_t)wave * 1024u, voB, k0 * (uint32_t)g.ldb * 2u);
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    issue(0);
    if (nt > 1) issue(1);
    for (int t = 0; t < nt; ++t) {
        if (t + 1 < nt) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");     // this wave's 3 pieces of stage t (stage t+1 may be in flight)
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (t + 2 < nt) issue(t + 2);
        const char* sa = smem + (t % DU_STAGES) * DU_STAGE_BYTES;
        const char* sb = sa + DU_A_BYTES;
        bf16x8 af[2][2], bf[2][2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[j][ks] = pp_read_frag<true>(sb, 64 * wc + 32 * j, ks, lane);
#pragma unroll
            for (int i = 0; i < 2; ++i)
                af[i][ks] = *reinterpret_cast<const bf16x8*>(sa + du_kc_off(wr * 64 + 32 * i + (lane & 31), ks * 2 + (lane >> 5)));
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[j][ks], af[i][ks], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    }
    __syncthreads();   // every wave is done with the operand ring: 8 KiB of it per wave become the u / du block
#ifdef SPN_GEMM_TIMING
    const long long tl_loop = __builtin_amdgcn_s_memrealtime();
#endif

    // ---- epilogue: activation backward on the wave's 64 rows x 64 gated outputs (see pp_store_tile<.., 3> for the arithmetic) ----
    const int I = g.N;
    const int colw = n0 + wc * 64;
    const int rl = lane & 31, hl = lane >> 5;
    char* blk = smem + wave * 8192;                       // [value 4 KiB | gate 4 KiB], rows of 128 B, 16-byte chunk c at c ^ ((row >> 1) & 7)
    const uint32_t blk_w = spn_lds_addr(blk);
    const u32x4 rsU = spn_buffer_rsrc(g.G, (uint32_t)((long)g.M * g.ldg * 2));
    const __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc(g.C, 0, (int)(uint32_t)((long)g.M * g.ldc * 2), 0x00020000);
    uint32_t vo[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int r = it * 8 + (lane >> 3), c = (lane & 7) ^ ((r >> 1) & 7);
        vo[it] = (uint32_t)(((long)(m0 + wr * 64 + r) * g.ldg + colw + c * 8) * 2);
    }
    const uint32_t vo_step = (uint32_t)(32 * g.ldg * 2);
    auto issue_u = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int part = 0; part < 2; ++part) {
            spn_dma16x2(rsU, blk_w + (uint32_t)(part * 4096), vo[0], vo[1], (uint32_t)(part * I) * 2u);
            spn_dma16x2(rsU, blk_w + (uint32_t)(part * 4096) + 2048u, vo[2], vo[3], (uint32_t)(part * I) * 2u);
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) vo[it] += vo_step;
    };
    issue_u();
    uint32_t so = (uint32_t)(((long)(m0 + wr * 64 + (lane >> 3)) * g.ldc + colw + (lane & 7) * 8) * 2);
    const uint32_t so_step8 = (uint32_t)(8 * g.ldc * 2);
    float csum[2] = {0.f, 0.f};
    const bf16x2_t ones = __builtin_bit_cast(bf16x2_t, 0x3f803f80u);
    const uint32_t thr16 = g.thr16;
    const float ks_ = g.keep_scale;
    const uint32_t hcol = __umul24((uint32_t)(colw >> 1) + 2u * (uint32_t)hl, 0xEBCA77u);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        // VM counter at the top of block i: i = 0: U0 | i = 1: U1 S0 (U1 was requested before block 0's stores were issued)
        if (i == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        const uint32_t hb = ffn_drop_rowc(m0 + wr * 64 + 32 * i + rl, g.seed) + hcol;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int s_ = 8 * j + 2 * q + hl;
                const int o = rl * 128 + ((((s_ >> 1) ^ ((rl >> 1) & 7)) & 7) << 4) + (s_ & 1) * 8;
                const uint2 uv = *reinterpret_cast<const uint2*>(blk + o);
                const uint2 ug = *reinterpret_cast<const uint2*>(blk + 4096 + o);
                const uint32_t p0 = pack_bf2(acc[i][j][4 * q], acc[i][j][4 * q + 1]), p1 = pack_bf2(acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
                float d[4] = {__uint_as_float(p0 << 16), __uint_as_float(p0 & 0xffff0000u), __uint_as_float(p1 << 16), __uint_as_float(p1 & 0xffff0000u)};
                if (thr16) {
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        uint32_t x = hb + (uint32_t)(16 * j + 4 * q + e) * 0xEBCA77u;
                        x ^= x >> 11; x = __umul24(x, 0xD35A2Du) + (x >> 8);
                        x ^= x >> 13; x = __umul24(x, 0x9E3B35u) + (x >> 9);
                        x ^= x >> 15;
                        const float s0 = d[2 * e] * ks_, s1 = d[2 * e + 1] * ks_;
                        d[2 * e] = (x & 0xffffu) >= thr16 ? s0 : 0.f;
                        d[2 * e + 1] = (x >> 16) >= thr16 ? s1 : 0.f;
                    }
                }
                const float a[4] = {__uint_as_float(uv.x << 16), __uint_as_float(uv.x & 0xffff0000u), __uint_as_float(uv.y << 16), __uint_as_float(uv.y & 0xffff0000u)};
                const float t[4] = {__uint_as_float(ug.x << 16), __uint_as_float(ug.x & 0xffff0000u), __uint_as_float(ug.y << 16), __uint_as_float(ug.y & 0xffff0000u)};
                float da[4], dt[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if constexpr (ACT == 0) {
                        const float sg = __builtin_amdgcn_rcpf(1.f + __expf(-t[e]));
                        da[e] = d[e] * (t[e] * sg);
                        dt[e] = d[e] * a[e] * (sg * (1.f + t[e] * (1.f - sg)));
                    } else {
                        da[e] = d[e] * gelu_f(t[e]);
                        dt[e] = d[e] * a[e] * gelu_grad(t[e]);
                    }
                }
                uint2 pa, pg;
                pa.x = pack_bf2(da[0], da[1]); pa.y = pack_bf2(da[2], da[3]);
                pg.x = pack_bf2(dt[0], dt[1]); pg.y = pack_bf2(dt[2], dt[3]);
                // in place: this lane's own 8 bytes of the value / gate image (nobody else reads or writes them before the wave-level sync)
                *reinterpret_cast<uint2*>(blk + o) = pa;
                *reinterpret_cast<uint2*>(blk + 4096 + o) = pg;
            }
        PP_SLAB_SYNC();
        if (g.ws) {   // column sums: lane l owns column l of the 64; a transposed read hands it rows 4 t .. 4 t + 3 of that column
            typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
            const int gq = lane >> 4, p = lane & 15;
#pragma unroll
            for (int part = 0; part < 2; ++part)
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const int rr = 4 * t + (p >> 2), sl = 4 * gq + (p & 3);
                    const bf16x4 w = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                        (lds_bf16x4*)(blk + part * 4096 + rr * 128 + ((((sl >> 1) ^ ((rr >> 1) & 7)) & 7) << 4) + (sl & 1) * 8));
                    csum[part] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(w, w, 0, 1), ones, csum[part], false);
                    csum[part] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(w, w, 2, 3), ones, csum[part], false);
                }
        }
        uint4 val[2][4];
#pragma unroll
        for (int part = 0; part < 2; ++part)
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int r = it * 8 + (lane >> 3), chunk = lane & 7;
                val[part][it] = *reinterpret_cast<const uint4*>(blk + part * 4096 + r * 128 + (((chunk ^ (r >> 1)) & 7) << 4));
            }
        if (i == 0) {   // the block has been read back: its 8 KiB take the next block's u (requested BEFORE the stores below are issued)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            PP_SLAB_SYNC();
            issue_u();
        }
#pragma unroll
        for (int part = 0; part < 2; ++part)
#pragma unroll
            for (int it = 0; it < 4; ++it)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, val[part][it]), rsD, (int)(so + (uint32_t)it * so_step8), part * I * 2, 0);
        so += 4u * so_step8;
    }
    // the partial buffer has one row per 128 rows of M: the two 64-row waves of a 128-row slab (wr = 2k, 2k + 1, same wc: waves w, w + 2)
    // meet through the upper wave's (now dead) LDS block; rows past M contributed exact zeros
    if (g.ws) {   // (kernel argument: uniform over the workgroup)
        float* red = reinterpret_cast<float*>(blk);
        if (wr & 1) { red[lane] = csum[0]; red[64 + lane] = csum[1]; }
        __syncthreads();
        if (!(wr & 1) && m0 + wr * 64 < g.M) {
            const float* other = reinterpret_cast<const float*>(smem + (wave + 2) * 8192);
            float* P = reinterpret_cast<float*>(g.ws) + (long)((m0 + wr * 64) / 128) * (2 * I);
            P[colw + lane] = csum[0] + other[lane];
            P[I + colw + lane] = csum[1] + other[64 + lane];
        }
    }
#ifdef SPN_GEMM_TIMING
    if (g.dbg) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const long long tl_end = __builtin_amdgcn_s_memrealtime();
        if (tid == 0) {
            long long* o = g.dbg + 32 + 5 * (long)blockIdx.x;
            o[0] = tl_start; o[1] = tl_loop; o[2] = tl_end;
            o[3] = __builtin_amdgcn_s_getreg((31 << 11) | 4); o[4] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        }
    }
#endif
}

// ---- split-K plumbing -------------------------------------------------------------------------------------
// K slices write their partial products to a workspace [splits][M][N] with plain stores and one small kernel sums them into C
// (deterministic; fp32 atomics into C cost up to half of the weight-gradient GEMMs: ~16K atomics per block through L2).
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, int splits, long slice, float* __restrict__ C,
                                                            int ldc, int N, long total4, int accumulate) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long)gridDim.x * 256) {
        const long e = i * 4;
        // eight slices' loads in flight at a time, added in slice order (the sum is the one a slice-by-slice loop forms; that loop waited for
        // every load before it requested the next: up to 64 serial trips per element)
        float* dst = C + (e / N) * ldc + (e % N);
        f32x4 old = f32x4{0.f, 0.f, 0.f, 0.f};
        if (accumulate) old = *reinterpret_cast<const f32x4*>(dst);
        f32x4 v = *reinterpret_cast<const f32x4*>(ws + e);
        int z = 1;
        for (; z + 8 <= splits; z += 8) {
            f32x4 t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = *reinterpret_cast<const f32x4*>(ws + (long)(z + u) * slice + e);
#pragma unroll
            for (int u = 0; u < 8; ++u) v += t[u];
        }
        for (; z < splits; ++z) v += *reinterpret_cast<const f32x4*>(ws + (long)z * slice + e);
        if (accumulate) v += old;
        *reinterpret_cast<f32x4*>(dst) = v;
    }
}

// decide the K split of a tiny-MxN / long-K product, redirect the kernel's output to the workspace; returns the final C description
// The workspace [splits][M][N] fp32 is the CALLER's (spn_gemm_workspace_bytes says how much this shape wants): nothing is allocated,
// freed or synchronised here, so a call is capturable and re-entrant.  A null / too small workspace simply runs the product unsplit.
struct SplitPlan { float* C; int ldc; int accumulate; float* ws; };
static void split_shape(const GemmArgs& g, int tiles, int nt, int max_tiles, int want_blocks, int min_kt, int& splitk, int& kt_per_split) {
    splitk = 1; kt_per_split = nt;
    if (g.batch != 1 || tiles >= max_tiles || nt < 4 * min_kt || g.N % 4 != 0) return;
    int want = want_blocks < 0 ? (-want_blocks) / tiles : cdiv(want_blocks, tiles);   // negative target: round down
    if (want > nt / min_kt) want = nt / min_kt;
    if (want <= 1) return;
    kt_per_split = cdiv(nt, want);
    splitk = cdiv(nt, kt_per_split);
    // the ping-pong kernel's peeled K loop wants >= 4 K tiles in every slice, the last one included
    while (splitk > 1 && nt - (splitk - 1) * kt_per_split < 4) { ++kt_per_split; splitk = cdiv(nt, kt_per_split); }
}
static void plan_split(GemmArgs& g, int tiles, int nt, int max_tiles, int want_blocks, int min_kt, SplitPlan& plan) {
    g.splitk = 1; g.kt_per_split = nt; plan.ws = nullptr;
    if (g.ldc % 4 != 0 || (reinterpret_cast<uintptr_t>(g.C) & 15) != 0) return;
    int splitk, kt;
    split_shape(g, tiles, nt, max_tiles, want_blocks, min_kt, splitk, kt);
    if (splitk <= 1) return;
    const size_t need = (size_t)splitk * g.M * g.N * 4;
    if (!g.ws || g.ws_bytes < need || (reinterpret_cast<uintptr_t>(g.ws) & 15) != 0) return;   // unsplit: slower, still correct
    g.splitk = splitk; g.kt_per_split = kt;
    plan = SplitPlan{reinterpret_cast<float*>(g.C), g.ldc, g.accumulate, reinterpret_cast<float*>(g.ws)};
    g.C = g.ws; g.ldc = g.N; g.sC = (long)g.M * g.N; g.accumulate = 0;
}
static void finish_split(const GemmArgs& g, const SplitPlan& plan, hipStream_t stream) {
    if (!plan.ws) return;
    const long total4 = (long)g.M * g.N / 4;
    int blocks = (int)((total4 + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, stream, plan.ws, g.splitk, (long)g.M * g.N, plan.C, plan.ldc, g.N,
                       total4, plan.accumulate);
}

// split-K policy of the ping-pong kernel: one block per CU, aim for ONE full round of 256 blocks (floor, so that no second, nearly
// empty round appears); products with >= 192 tiles or fewer than 64 K tiles are not split
constexpr int PP_SPLIT_MAX_TILES = 192, PP_SPLIT_MIN_KT = 16;
static int pp_split_want() { const int w = spn_tune_i(SPN_TUNE_GEMM_SPLIT_BLOCKS); return w > 0 ? w : -256; }

// shapes the ping-pong kernel takes: K in whole 64-tiles, rows in multiples of 8, vector-aligned epilogue operands
static bool pp_eligible(const GemmArgs& g) {
    // edge tiles in M and N are fine (clamped loads, guarded stores); K must be whole 64-tiles
    // operands are addressed with 32-bit byte offsets from their base
    return g.M >= 128 && g.N >= 128 && g.M % 8 == 0 && g.N % 8 == 0 && g.K % PP_BK == 0 && g.K >= 4 * PP_BK && g.ldc % 8 == 0 &&
           g.pp_addr_ok &&
           (reinterpret_cast<uintptr_t>(g.C) & 15) == 0 && (!g.residual || (g.ldr % 4 == 0 && (reinterpret_cast<uintptr_t>(g.residual) & 15) == 0)) &&
           (!g.bias || (reinterpret_cast<uintptr_t>(g.bias) & 15) == 0) && (g.sC % 4 == 0);
}

template <bool TA, bool TB, typename OutT>
int launch_pp(GemmArgs g, hipStream_t stream) {
    const int tiles = cdiv(g.N, PP_BN) * cdiv(g.M, PP_BM), nt = g.K / PP_BK;
    SplitPlan plan;
    if (sizeof(OutT) == 4) {
        plan_split(g, tiles, nt, PP_SPLIT_MAX_TILES, pp_split_want(), PP_SPLIT_MIN_KT, plan);
    } else {
        g.splitk = 1; g.kt_per_split = nt; plan.ws = nullptr;
    }
    constexpr int LDS_BYTES = 8 * PP_HALF + 8 * 4096;   // operand ring + epilogue staging
    static std::atomic<unsigned> optin{0};
    spn_lds_optin(optin, reinterpret_cast<const void*>(&gemm_pp_kernel<TA, TB, OutT>), LDS_BYTES);
    dim3 grid(cdiv(g.N, PP_BN), cdiv(g.M, PP_BM), g.splitk > 1 ? g.splitk : g.batch);
    g.ngroup = spn_tune_i(SPN_TUNE_GEMM_NGROUP) > 0 ? spn_tune_i(SPN_TUNE_GEMM_NGROUP) : 8;
    if (g.ngroup > (int)grid.x) g.ngroup = grid.x;
    g.tx = grid.x; g.ty = grid.y;
    // (weight gradients at C3: 4096x512x131072 609 -> 590 us, 512x2048x131072 335 -> 323 us)
    g.slice_xcd = (spn_tune_i(SPN_TUNE_GEMM_SLICE_XCD) && g.splitk > 1 && g.splitk % 8 == 0) ? 1 : 0;
    // one block per CU walking the tile list (no workgroup launch between tiles, the next tile's first loads in flight during the
    // epilogue).  A block keeps its share of tiles whatever happens to its CU, so a kernel that holds CUs while this one runs (the
    // all-reduce of a data-parallel backward) would delay that share by a block lifetime: forward products walk by default, input
    // gradients only when the host says nothing else runs (gemm_persist_bwd; scoreperformer_amd/parallel.py sets it for one process)
    const int persist_env = spn_tune_i(SPN_TUNE_GEMM_PERSIST);   // 0 off, else min rounds
    g.stagger = spn_tune_i(SPN_TUNE_GEMM_STAGGER) * 1024 / 32;
    // measured (tools/bench_gemm.py, one box): bf16 outputs with a short contraction gain 4-8 % (131072x4096x512 652 -> 618 us,
    // 131072x640x512 124 -> 115 us, 131072x2048x512 NT 357 -> 343 us); K >= 2048 and the fp32 + residual epilogues (which stage the
    // residual through the ring the next tile's prologue would fill) lose 3-5 %: those keep one block per tile
    const bool persist_ok = sizeof(OutT) == 2 && nt <= 16 && !g.residual && ((!TA && !TB) || spn_tune_i(SPN_TUNE_GEMM_PERSIST_BWD) > 0);
    if (persist_env > 0 && persist_ok && grid.z == 1 && (long)grid.x * grid.y >= 256l * persist_env) {
        static std::atomic<unsigned> optin_p{0};
        spn_lds_optin(optin_p, reinterpret_cast<const void*>(&gemm_pp_kernel<TA, TB, OutT, 0, true>), LDS_BYTES);
        hipLaunchKernelGGL((gemm_pp_kernel<TA, TB, OutT, 0, true>), dim3(256, 1, 1), dim3(512), LDS_BYTES, stream, g);
    } else
    hipLaunchKernelGGL((gemm_pp_kernel<TA, TB, OutT>), grid, dim3(512), LDS_BYTES, stream, g);
    SPN_LAUNCH_CHECK();
    finish_split(g, plan, stream);
    SPN_LAUNCH_CHECK();
    return SPN_OK;
}

#ifdef SPN_GEMM_OW_VARIANT
#include "../../tools/variants/gemm_ow_launch.inc"
#endif

template <bool TA, bool TB, typename OutT, int BK, int STAGES>
int launch_bk(GemmArgs g, hipStream_t stream) {
    // split-K for the weight-gradient shapes (tiny M x N, contraction over all tokens): fill the chip with K slices
    const int tiles = cdiv(g.N, BN) * cdiv(g.M, BM), nt = cdiv(g.K, BK);
    SplitPlan plan;
    if (sizeof(OutT) == 4) {
        plan_split(g, tiles, nt, 384, 768, 8, plan);
    } else {
        g.splitk = 1; g.kt_per_split = nt; plan.ws = nullptr;
    }
    dim3 grid(cdiv(g.N, BN), cdiv(g.M, BM), g.splitk > 1 ? g.splitk : g.batch);
    constexpr int LDS_BYTES = STAGES * 2 * 128 * BK * 2;
    if (LDS_BYTES > 64 * 1024) {   // > 64 KiB of dynamic LDS needs the opt-in attribute (once per instantiation and device)
        static std::atomic<unsigned> optin{0};
        spn_lds_optin(optin, reinterpret_cast<const void*>(&gemm_kernel<TA, TB, OutT, BK, STAGES>), LDS_BYTES);
    }
    g.ngroup = spn_tune_i(SPN_TUNE_GEMM_NGROUP) > 0 ? spn_tune_i(SPN_TUNE_GEMM_NGROUP) : 8;
    if (g.ngroup > (int)grid.x) g.ngroup = grid.x;
    hipLaunchKernelGGL((gemm_kernel<TA, TB, OutT, BK, STAGES>), grid, dim3(256), LDS_BYTES, stream, g);
    SPN_LAUNCH_CHECK();
    finish_split(g, plan, stream);
    SPN_LAUNCH_CHECK();
    return SPN_OK;
}

// shapes the duo kernel takes: A K-contiguous, K in whole 32-tiles, enough tiles to give every CU its two workgroups
static bool duo_eligible(const GemmArgs& g, bool ta) {
    return !ta && g.batch == 1 && g.M >= 128 && g.N >= 64 && g.M % 8 == 0 && g.N % 8 == 0 && g.K % DU_BK == 0 && g.K >= 2 * DU_BK &&
           g.ldc % 8 == 0 && g.pp_addr_ok && (reinterpret_cast<uintptr_t>(g.C) & 15) == 0 &&
           (!g.residual || (g.ldr % 4 == 0 && (reinterpret_cast<uintptr_t>(g.residual) & 15) == 0)) &&
           (!g.bias || (reinterpret_cast<uintptr_t>(g.bias) & 15) == 0) &&
           (long)cdiv(g.N, DU_BN) * cdiv(g.M, DU_BM) >= 512;
}

// measured on MI355X (tools/bench_gemm.py, profiles/r02_gemm_shapes.txt): where the duo kernel beats the ping-pong kernel
// (r02_gemm_ab.txt).  Two workgroups per CU overlap one tile's epilogue with the other's main loop, but a 256x128 tile moves 1.5x the
// operand bytes per flop through the CU's 64 B/clk vector-memory path and issues 1.5x the LDS-DMA instructions per wave, so on every
// wide projection the 256x256 ping-pong kernel stays ahead (K = 512, N = 4096: 855 vs 764 TF/s); duo wins only where a 256-wide
// tile would be mostly padding (N <= 128: the 64-column condition gradients, 51 vs 73 us) and where the ping-pong kernel cannot go at
// all (K < 256: the AdaLN condition projections, K = 64, 131008 x 1024 x 64: 67 vs 78-83 us on the 128x128 kernel).
static bool duo_preferred(const GemmArgs& g, bool tb, bool f32) {
    (void)tb; (void)f32;
    return g.N <= 128 || g.K < 4 * PP_BK;
}

template <bool TB, typename OutT, int GLU>
int launch_duo(GemmArgs g, hipStream_t stream) {
    static std::atomic<unsigned> optin{0};
    spn_lds_optin(optin, reinterpret_cast<const void*>(&gemm_duo_kernel<TB, OutT, GLU>), DU_LDS_BYTES);
    g.tx = GLU ? g.N / (DU_BN / 2) : cdiv(g.N, DU_BN);
    g.ty = cdiv(g.M, DU_BM);
    g.splitk = 1; g.kt_per_split = g.K / DU_BK; g.slice_xcd = 0; g.stagger = 0;
    const int ng = spn_tune_i(SPN_TUNE_GEMM_DUO_NGROUP);
    g.ngroup = ng > 0 ? ng : 8;
    if (g.ngroup > g.tx) g.ngroup = g.tx;
    hipLaunchKernelGGL((gemm_duo_kernel<TB, OutT, GLU>), dim3(g.tx * g.ty), dim3(256), DU_LDS_BYTES, stream, g);
    SPN_LAUNCH_CHECK();
    return SPN_OK;
}

template <bool TA, bool TB, typename OutT>
int launch(const GemmArgs& g, hipStream_t stream) {
    // short contractions (K <= 1024: every projection with d_model = 512 on the input side) are latency-bound per block:
    // BK = 32 halves the LDS footprint (32 KiB) so that 4 blocks stay resident per CU and hide each other's pipeline fill
    const int variant = spn_tune_i(SPN_TUNE_GEMM_VARIANT);   // tuning aid
    if constexpr (!TA) {
        // two 4-wave workgroups per CU: measured per shape (tools/bench_gemm.py), knob 2 forces it wherever it is eligible
        const int duo = spn_tune_i(SPN_TUNE_GEMM_DUO);
        if (variant == 0 && duo && duo_eligible(g, false) && (duo >= 2 || duo_preferred(g, TB, sizeof(OutT) == 4)))
            return launch_duo<TB, OutT, 0>(g, stream);
    }
    if (variant == 1) return launch_bk<TA, TB, OutT, 64, 2>(g, stream);
    if (variant == 2) return launch_bk<TA, TB, OutT, 64, 3>(g, stream);
    if (variant == 3) return launch_bk<TA, TB, OutT, 32, 4>(g, stream);
    if (variant == 4) return launch_bk<TA, TB, OutT, 32, 2>(g, stream);
    if (variant == 5) return launch_bk<TA, TB, OutT, 64, 4>(g, stream);
    if (variant == 6) return launch_bk<TA, TB, OutT, 32, 3>(g, stream);
    if (variant == 9 && pp_eligible(g)) return launch_pp<TA, TB, OutT>(g, stream);
#ifdef SPN_GEMM_OW_VARIANT
    {
        constexpr int ow = SPN_GEMM_OW_VARIANT;
        // 1 / 2: LDS-DMA variant where preferred / everywhere; 3 / 4: register-staged variant where preferred / everywhere
        if (variant == 0 && pp_eligible(g) && (ow == 2 || ow == 4 || ow_preferred(g))) return launch_ow<TA, TB, OutT>(g, stream);
    }
#endif
    // measured (tools/bench_gemm.py): with the LDS-staged epilogue the 256x256 ping-pong kernel wins on every shape it can take
    if (variant == 0 && pp_eligible(g)) return launch_pp<TA, TB, OutT>(g, stream);
    // measured on MI355X (tools/bench_gemm.py): residency beats in-block pipelining -- two 16 KiB-per-operand stages with
    // BK = 32 (32 KiB LDS, 4 blocks/CU) win everywhere except the long-K all-K-contiguous case
    if (!TA && !TB && g.K >= 2048) return launch_bk<TA, TB, OutT, 64, 2>(g, stream);
    return launch_bk<TA, TB, OutT, 32, 2>(g, stream);
}

template <int GLU>
int launch_pp_glu(GemmArgs g, hipStream_t stream) {
    constexpr int LDS_BYTES = 8 * PP_HALF + 8 * 4096;
    static std::atomic<unsigned> optin{0};
    spn_lds_optin(optin, reinterpret_cast<const void*>(&gemm_pp_kernel<false, false, bf16_t, GLU>), LDS_BYTES);
    dim3 grid(g.N / (PP_BN / 2), cdiv(g.M, PP_BM), 1);
    g.ngroup = spn_tune_i(SPN_TUNE_GEMM_NGROUP) > 0 ? spn_tune_i(SPN_TUNE_GEMM_NGROUP) : 8;
    if (g.ngroup > (int)grid.x) g.ngroup = grid.x;
    g.tx = grid.x; g.ty = grid.y; g.slice_xcd = 0;
    // one block per CU walking the tile list: the next tile's first DMA is in flight while this tile's (long, VALU-bound) epilogue runs
    // (764 -> 722 us at 131072 x 2048 x 512, 820 -> 763 us with dropout).  A forward kernel: nothing else holds CUs while it runs (see
    // launch_pp for the backward side of that argument).
    const int persist_env = spn_tune_i(SPN_TUNE_GLU_PERSIST);   // 0 off, else min rounds
    g.stagger = spn_tune_i(SPN_TUNE_GEMM_STAGGER) * 1024 / 32;
    if (persist_env > 0 && (long)grid.x * grid.y >= 256l * persist_env) {
        static std::atomic<unsigned> optin_p{0};
        spn_lds_optin(optin_p, reinterpret_cast<const void*>(&gemm_pp_kernel<false, false, bf16_t, GLU, true>), LDS_BYTES);
        hipLaunchKernelGGL((gemm_pp_kernel<false, false, bf16_t, GLU, true>), dim3(256, 1, 1), dim3(512), LDS_BYTES, stream, g);
    } else
    hipLaunchKernelGGL((gemm_pp_kernel<false, false, bf16_t, GLU>), grid, dim3(512), LDS_BYTES, stream, g);
    SPN_LAUNCH_CHECK();
    return SPN_OK;
}

}  // namespace

// C-ABI ---------------------------------------------------------------------------------------------------
// flags: bit0 = A is M-contiguous (transposed storage), bit1 = B is N-contiguous, bit2 = C is fp32 (else bf16),
//        bit3 = accumulate into C (fp32 C only).
//
// Weight-gradient shapes (fp32 C, tiny M x N, K = all tokens) are split over K through a CALLER-OWNED fp32 workspace:
// spn_gemm_workspace_bytes(M, N, K, flags, batch) is the size this shape can use (0: never split).  With workspace == null or fewer
// bytes the product runs unsplit (correct, slower).  The call never allocates, frees or synchronises.
extern "C" size_t spn_gemm_workspace_bytes(int M, int N, int K, int flags, int batch) {
    if (!(flags & 4) || M <= 0 || N <= 0 || K <= 0) return 0;   // only fp32 outputs are split
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.M = M; g.N = N; g.K = K; g.batch = batch;
    size_t need = 0;
    int splitk, kt;
    if (M >= 128 && N >= 128 && M % 8 == 0 && N % 8 == 0 && K % PP_BK == 0 && K >= 4 * PP_BK) {
        split_shape(g, cdiv(N, PP_BN) * cdiv(M, PP_BM), K / PP_BK, PP_SPLIT_MAX_TILES, pp_split_want(), PP_SPLIT_MIN_KT, splitk, kt);
        if (splitk > 1) need = (size_t)splitk * M * N * 4;
    }
    for (int bk = 32; bk <= 64; bk *= 2) {   // the 128x128 kernels (whichever K tile the dispatch picks)
        split_shape(g, cdiv(N, BN) * cdiv(M, BM), cdiv(K, bk), 384, 768, 8, splitk, kt);
        if (splitk > 1 && (size_t)splitk * M * N * 4 > need) need = (size_t)splitk * M * N * 4;
    }
    return need;
}

extern "C" int spn_gemm_bf16(const void* A, const void* B, void* C, const float* bias, const float* residual,
                             const uint8_t* rowmask, int M, int N, int K, int lda, int ldb, int ldc, int ldr,
                             float alpha, int flags, int batch, long strideA, long strideB, long strideC,
                             void* workspace, size_t workspace_bytes, hipStream_t stream) {
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
    g.G = nullptr; g.ldg = 0; g.thr16 = 0; g.seed = 0; g.keep_scale = 1.f; g.stagger = 0;
    g.ws = workspace; g.ws_bytes = workspace ? workspace_bytes : 0;
    {
        const long a_span = (long)(ta ? K : M) * lda * 2, b_span = (long)(tb ? K : N) * ldb * 2;
        g.pp_addr_ok = (a_span < (1L << 31) && b_span < (1L << 31)) ? 1 : 0;
    }
#ifdef SPN_GEMM_TIMING
    g.dbg = g_dbg;
#endif
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

// Input projection of a gated feed-forward with its activation fused into the GEMM epilogue (feedforward.py:17-20,57-60):
//   u[M, 2I] = x[M, K] . W[2I, K]^T + bias[2I]   (bf16, kept for the backward: value columns 0..I-1, gate columns I..2I-1)
//   g[M, I]  = dropout(u[:, :I] * act(u[:, I:]))  with the mask of spn_act_fwd(seed), so spn_act_bwd is its backward unchanged
// act: 0 = SiLU, 1 = GELU(erf).  Returns SPN_ERR_ARG for shapes the ping-pong kernel does not take (spn_gemm_glu_ok says which):
// the caller then runs spn_gemm_bf16 + spn_act_fwd.
extern "C" int spn_gemm_glu_ok(int M, int I, int K) {
    return (M >= 128 && M % 8 == 0 && I >= 128 && I % (PP_BN / 2) == 0 && K % PP_BK == 0 && K >= 4 * PP_BK) ? 1 : 0;
}

extern "C" int spn_gemm_glu(const void* x, const void* W, void* u, void* gout, const float* bias, int M, int I, int K, int lda, int ldb,
                            int ldu, int ldg, int act, float p_drop, unsigned seed, hipStream_t stream) {
    SPN_REQUIRE(x && W && u && gout, "spn_gemm_glu: null operand");
    SPN_REQUIRE(spn_gemm_glu_ok(M, I, K), "spn_gemm_glu: shape not supported (M >= 128, I a multiple of 128, K a multiple of 64 and >= 256)");
    SPN_REQUIRE(lda % 8 == 0 && ldb % 8 == 0 && ldu % 8 == 0 && ldg % 8 == 0, "spn_gemm_glu: leading dimensions must be multiples of 8");
    SPN_REQUIRE(((((uintptr_t)x) | ((uintptr_t)W) | ((uintptr_t)u) | ((uintptr_t)gout)) & 15) == 0 && (!bias || (((uintptr_t)bias) & 15) == 0),
                "spn_gemm_glu: operands must be 16-byte aligned");
    SPN_REQUIRE((long)M * lda * 2 < (1L << 31) && 2l * I * ldb * 2 < (1L << 31), "spn_gemm_glu: operand spans 2 GiB or more");
    // the PADDED extent: per-lane row offsets are formed for every row of the last 256-row tile, rows past M included -- they must not
    // wrap back into the buffer (the hardware bounds check drops offsets >= the record count, not wrapped ones)
    SPN_REQUIRE(cdiv(M, 256) * 256l * ldu * 2 < (1L << 32) && cdiv(M, 256) * 256l * ldg * 2 < (1L << 32),
                "spn_gemm_glu: u or g (rows rounded up to 256) spans 4 GiB or more (32-bit buffer offsets)");
    SPN_REQUIRE(act == 0 || act == 1, "spn_gemm_glu: act is 0 (SiLU) or 1 (GELU)");
    GemmArgs g;
    g.A = (const bf16_t*)x; g.B = (const bf16_t*)W; g.C = u; g.bias = bias; g.residual = nullptr; g.rowmask = nullptr;
    g.M = M; g.N = I; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldu; g.ldr = 0; g.alpha = 1.f;
    g.accumulate = 0; g.batch = 1; g.sA = g.sB = g.sC = 0; g.splitk = 1; g.kt_per_split = K / PP_BK; g.pp_addr_ok = 1;
    g.G = (bf16_t*)gout; g.ldg = ldg; g.seed = seed; g.ws = nullptr; g.ws_bytes = 0; g.stagger = 0;
    const float t = p_drop * 65536.f;   // thr16_of of elementwise.hip
    g.thr16 = t <= 0.f ? 0u : (t >= 65535.f ? 65535u : (uint32_t)(t + 0.5f));
    g.keep_scale = 1.f / (1.f - (float)g.thr16 / 65536.f);
#ifdef SPN_GEMM_TIMING
    g.dbg = g_dbg;
#endif
    if (spn_tune_i(SPN_TUNE_GEMM_DUO) && I % (DU_BN / 2) == 0 && (long)(I / (DU_BN / 2)) * cdiv(M, DU_BM) >= 512 && K % DU_BK == 0 &&
        spn_tune_i(SPN_TUNE_GEMM_DUO) >= 2)
        return act == 0 ? launch_duo<false, bf16_t, 1>(g, stream) : launch_duo<false, bf16_t, 2>(g, stream);
    return act == 0 ? launch_pp_glu<1>(g, stream) : launch_pp_glu<2>(g, stream);
}

// Input gradient of a gated feed-forward's output projection with the activation backward in its epilogue (feedforward.py:17-20,57-64):
//   dg = dy[M, K] . W2[K, I]              (never stored)
//   d  = dropout_mask(bf16(dg)),  du[:, :I] = d * act(u[:, I:]),  du[:, I:] = d * u[:, :I] * act'(u[:, I:])      (bf16 [M, 2I])
// = spn_gemm_bf16 (N-contiguous B) followed by spn_act_bwd, bit for bit, without the 2 x M x I x 2 bytes of dg between them.
// colsum_partial (fp32 [ceil(M / 128), 2I] or null): row r receives the column sums of du rows 128 r .. (bias gradient of the input
// projection: sum the rows).  W2 is the nn.Linear weight [K, I] of the output projection (row stride ldw).
extern "C" int spn_gemm_glu_bwd_ok(int M, int I, int K) {
    return (M >= 128 && M % 8 == 0 && I >= 256 && I % PP_BN == 0 && K % PP_BK == 0 && K >= 4 * PP_BK) ? 1 : 0;
}

extern "C" int spn_gemm_glu_bwd(const void* dy, const void* W2, const void* u, void* du, float* colsum_partial, int M, int I, int K,
                                int lddy, int ldw, int ldu, int lddu, int act, float p_drop, unsigned seed, hipStream_t stream) {
    SPN_REQUIRE(dy && W2 && u && du, "spn_gemm_glu_bwd: null operand");
    SPN_REQUIRE(spn_gemm_glu_bwd_ok(M, I, K), "spn_gemm_glu_bwd: shape not supported (M >= 128, I a multiple of 256, K a multiple of 64 and >= 256)");
    SPN_REQUIRE(lddy % 8 == 0 && ldw % 8 == 0 && ldu % 8 == 0 && lddu % 8 == 0, "spn_gemm_glu_bwd: leading dimensions must be multiples of 8");
    SPN_REQUIRE(((((uintptr_t)dy) | ((uintptr_t)W2) | ((uintptr_t)u) | ((uintptr_t)du)) & 15) == 0 && (!colsum_partial || (((uintptr_t)colsum_partial) & 7) == 0),
                "spn_gemm_glu_bwd: operands must be 16-byte aligned");
    SPN_REQUIRE((long)M * lddy * 2 < (1L << 31) && (long)K * ldw * 2 < (1L << 31), "spn_gemm_glu_bwd: operand spans 2 GiB or more");
    SPN_REQUIRE(cdiv(M, 256) * 256l * ldu * 2 < (1L << 32) && cdiv(M, 256) * 256l * lddu * 2 < (1L << 32),
                "spn_gemm_glu_bwd: u or du (rows rounded up to 256) spans 4 GiB or more (32-bit buffer offsets)");
    SPN_REQUIRE(act == 0 || act == 1, "spn_gemm_glu_bwd: act is 0 (SiLU) or 1 (GELU)");
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.A = (const bf16_t*)dy; g.B = (const bf16_t*)W2; g.C = du;
    g.M = M; g.N = I; g.K = K; g.lda = lddy; g.ldb = ldw; g.ldc = lddu; g.alpha = 1.f;
    g.batch = 1; g.splitk = 1; g.kt_per_split = K / PP_BK; g.pp_addr_ok = 1;
    g.G = (bf16_t*)u; g.ldg = ldu; g.seed = seed; g.ws = colsum_partial;
    const float t = p_drop * 65536.f;   // thr16_of of elementwise.hip
    g.thr16 = t <= 0.f ? 0u : (t >= 65535.f ? 65535u : (uint32_t)(t + 0.5f));
    g.keep_scale = 1.f / (1.f - (float)g.thr16 / 65536.f);
#ifdef SPN_GEMM_TIMING
    g.dbg = g_dbg;
#endif
    // Two 4-wave workgroups per CU (256x128 tiles): this epilogue moves 4 bytes of u / du per output through HBM and takes twice as
    // long as the K = 512 main loop, so what matters is that one workgroup's main loop runs under the other's epilogue
    if (spn_tune_i(SPN_TUNE_GLU_BWD_DUO) >= 2 && K % DU_BK == 0) {   // eight waves per workgroup (gemm_duo8_glu_bwd_kernel)
        g.tx = I / DU_BN; g.ty = cdiv(M, DU_BM);
        const int ng = spn_tune_i(SPN_TUNE_GEMM_DUO_NGROUP);
        g.ngroup = ng > 0 ? ng : 8;
        if (g.ngroup > g.tx) g.ngroup = g.tx;
        if (act == 0) {
            static std::atomic<unsigned> optin{0};
            spn_lds_optin(optin, reinterpret_cast<const void*>(&gemm_duo8_glu_bwd_kernel<0>), DU_LDS_BYTES);
            hipLaunchKernelGGL((gemm_duo8_glu_bwd_kernel<0>), dim3(g.tx * g.ty), dim3(512), DU_LDS_BYTES, stream, g);
        } else {
            static std::atomic<unsigned> optin{0};
            spn_lds_optin(optin, reinterpret_cast<const void*>(&gemm_duo8_glu_bwd_kernel<1>), DU_LDS_BYTES);
            hipLaunchKernelGGL((gemm_duo8_glu_bwd_kernel<1>), dim3(g.tx * g.ty), dim3(512), DU_LDS_BYTES, stream, g);
        }
        SPN_LAUNCH_CHECK();
        return SPN_OK;
    }
    if (spn_tune_i(SPN_TUNE_GLU_BWD_DUO) && K % DU_BK == 0) {
        constexpr int DUO_LDS = 16384 + 4 * 16384;
        g.tx = I / DU_BN; g.ty = cdiv(M, DU_BM);
        const int ng = spn_tune_i(SPN_TUNE_GEMM_DUO_NGROUP);
        g.ngroup = ng > 0 ? ng : 8;
        if (g.ngroup > g.tx) g.ngroup = g.tx;
        if (act == 0) {
            static std::atomic<unsigned> optin{0};
            spn_lds_optin(optin, reinterpret_cast<const void*>(&gemm_duo_kernel<true, bf16_t, 3>), DUO_LDS);
            hipLaunchKernelGGL((gemm_duo_kernel<true, bf16_t, 3>), dim3(g.tx * g.ty), dim3(256), DUO_LDS, stream, g);
        } else {
            static std::atomic<unsigned> optin{0};
            spn_lds_optin(optin, reinterpret_cast<const void*>(&gemm_duo_kernel<true, bf16_t, 4>), DUO_LDS);
            hipLaunchKernelGGL((gemm_duo_kernel<true, bf16_t, 4>), dim3(g.tx * g.ty), dim3(256), DUO_LDS, stream, g);
        }
        SPN_LAUNCH_CHECK();
        return SPN_OK;
    }
    constexpr int LDS_BYTES = 8 * PP_HALF + 8 * 4096;
    dim3 grid(I / PP_BN, cdiv(M, PP_BM), 1);
    g.ngroup = spn_tune_i(SPN_TUNE_GEMM_NGROUP) > 0 ? spn_tune_i(SPN_TUNE_GEMM_NGROUP) : 8;
    if (g.ngroup > (int)grid.x) g.ngroup = grid.x;
    g.tx = grid.x; g.ty = grid.y;
    if (act == 0) {
        static std::atomic<unsigned> optin{0};
        spn_lds_optin(optin, reinterpret_cast<const void*>(&gemm_pp_kernel<false, true, bf16_t, 3>), LDS_BYTES);
        hipLaunchKernelGGL((gemm_pp_kernel<false, true, bf16_t, 3>), grid, dim3(512), LDS_BYTES, stream, g);
    } else {
        static std::atomic<unsigned> optin{0};
        spn_lds_optin(optin, reinterpret_cast<const void*>(&gemm_pp_kernel<false, true, bf16_t, 4>), LDS_BYTES);
        hipLaunchKernelGGL((gemm_pp_kernel<false, true, bf16_t, 4>), grid, dim3(512), LDS_BYTES, stream, g);
    }
    SPN_LAUNCH_CHECK();
    return SPN_OK;
}

#ifdef SPN_GEMM_TIMING
extern "C" void spn_gemm_set_debug(void* p) { g_dbg = (long long*)p; }
#endif
