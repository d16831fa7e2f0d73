// AdaptiveLayerNorm with the condition projection INSIDE the LayerNorm kernels (gfx950; D = 512 features, C = 64 condition channels).
//
// Replaces `AdaptiveLayerNorm.forward` (modules/layers.py:31-47):
//     gamma_t | beta_t = Linear(C -> 2D)(cond_t),      y_t = gamma_t * LayerNorm(x_t) + beta_t
// The unfused path writes the per-token (gamma | beta) rows [T, 2D] with a K = 64 GEMM, reads them in the LayerNorm forward and again in
// its backward, writes the per-token (d gamma | d beta) rows and reads those twice more (input- and weight-gradient GEMMs with K or
// N = 64): ~1.6 GB per layer at C3 (13 adaptive norms per step), all of it HBM-bound.  The projection is tiny (64 -> 1024 per token):
// here every wave computes it on the matrix cores for its own 16 tokens from the LDS-resident weight (128 KiB, bf16) and nothing
// per-token but x, cond, y and the bf16 gamma rows the backward needs crosses the memory bus.  (A backward in the same style --
// gamma recomputed by MFMA, d cond by MFMA against the transposed weight, dy * xhat rows out -- was built and measured in round 3: the
// state it must hold between its two passes over a row (xhat, dy, d residual) does not fit 16-token MFMA tiles without spills, and at
// 8 tokens per tile it ran 538 us against 272 us for the wave-per-row kernel it was to replace: not kept, docs/LOG.md section 7.)
//
// Orientation: gamma^T[f, t] = sum_k W[f, k] cond[t, k]  (v_mfma_f32_16x16x32_bf16: A = 16 weight rows x 32 k from LDS, B = 32 k x 16
// tokens straight from global memory, 16 bytes per lane).  C/D layout: lane (c = lane & 15, g = lane >> 4) holds token c, features
// 16 blk + 4 g + r (r = 0..3): exactly the layout in which the wave keeps its 16 rows of x (one float4 per 16-feature block and lane),
// so LayerNorm statistics are lane-local sums plus one reduction over the 4 lane groups, and the bias (1 | 0 at initialisation) is the
// C input of the MFMA.  Results leave through a wave-private LDS slab as whole 128-byte lines.
#include "common.h"
#include "tuning.h"

namespace {

// wave-private LDS slab exchange: order cross-lane writes before reads for the compiler (no instruction is emitted)
#define SLAB_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); \
                         __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)

constexpr int AD = 512, AC = 64, NBLK = AD / 16;
constexpr int W_BYTES = 2 * AD * AC * 2;          // 131072
constexpr int STG_ROW = 144;                       // staging row stride: 128 data bytes + 16 (keeps 16-byte alignment, spreads banks)
constexpr int STG_BYTES = 16 * STG_ROW;            // per wave
constexpr int FWD_LDS = W_BYTES + 2 * AD * 4 + 8 * STG_BYTES;

// weight row (128 bytes = 8 chunks of 16 B) in LDS, conflict-light for ds_read_b128 of (row = lane & 15, chunk = const + (lane >> 4))
__device__ __forceinline__ int wa_off(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }

__device__ __forceinline__ float group4_sum(float v) {   // across the 4 lane groups (same lane & 15)
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
}

__global__ __launch_bounds__(512, 2) void adaln_fwd_kernel(const float* __restrict__ x, long ldx, const bf16_t* __restrict__ cond, long ldc,
                                                           const bf16_t* __restrict__ W, const float* __restrict__ bias,
                                                           bf16_t* __restrict__ y, long ldy, bf16_t* __restrict__ gamma_out, long ldg,
                                                           float* __restrict__ mean, float* __restrict__ rstd, int T, float eps) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* w_s = smem;
    float* b_s = reinterpret_cast<float*>(smem + W_BYTES);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, c = lane & 15, g = lane >> 4;
    char* stg = smem + W_BYTES + 2 * AD * 4 + w * STG_BYTES;
    for (int i = tid; i < 2 * AD * 8; i += 512) {
        const int row = i >> 3, ch = i & 7;
        *reinterpret_cast<uint4*>(w_s + wa_off(row, ch)) = *reinterpret_cast<const uint4*>(W + (long)row * AC + ch * 8);
    }
    for (int i = tid; i < 2 * AD; i += 512) b_s[i] = bias[i];
    __syncthreads();

    const int ngroups = (T + 15) / 16;
    for (int grp = blockIdx.x * 8 + w; grp < ngroups; grp += gridDim.x * 8) {
        const int t0 = grp * 16;
        const int trow = min(t0 + c, T - 1);
        f32x4 xq[NBLK];
        const float* xr = x + (long)trow * ldx + 4 * g;
#pragma unroll
        for (int blk = 0; blk < NBLK; ++blk) xq[blk] = *reinterpret_cast<const f32x4*>(xr + 16 * blk);
        bf16x8 bfr[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) bfr[ks] = *reinterpret_cast<const bf16x8*>(cond + (long)trow * ldc + 32 * ks + 8 * g);
        // statistics: fp32 two-pass (mean, then centred variance), like ATen's CPU kernel
        float sum = 0.f;
#pragma unroll
        for (int blk = 0; blk < NBLK; ++blk) sum += (xq[blk][0] + xq[blk][1]) + (xq[blk][2] + xq[blk][3]);
        const float mu = group4_sum(sum) * (1.f / AD);
        float sq = 0.f;
#pragma unroll
        for (int blk = 0; blk < NBLK; ++blk) {
            xq[blk] -= mu;
            sq += (xq[blk][0] * xq[blk][0] + xq[blk][1] * xq[blk][1]) + (xq[blk][2] * xq[blk][2] + xq[blk][3] * xq[blk][3]);
        }
        const float rs = rsqrtf(group4_sum(sq) * (1.f / AD) + eps);
        if (g == 0 && t0 + c < T) { mean[t0 + c] = mu; rstd[t0 + c] = rs; }
#pragma unroll
        for (int ch = 0; ch < NBLK / 4; ++ch) {   // 64 features at a time: 4 gamma blocks + 4 beta blocks, 16 MFMAs
            uint2 gpk[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int blk = 4 * ch + j;
                f32x4 ga = *reinterpret_cast<const f32x4*>(b_s + 16 * blk + 4 * g);
                f32x4 be = *reinterpret_cast<const f32x4*>(b_s + AD + 16 * blk + 4 * g);
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const bf16x8 ag = *reinterpret_cast<const bf16x8*>(w_s + wa_off(16 * blk + c, 4 * ks + g));
                    const bf16x8 ab = *reinterpret_cast<const bf16x8*>(w_s + wa_off(AD + 16 * blk + c, 4 * ks + g));
                    ga = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ag, bfr[ks], ga, 0, 0, 0);
                    be = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bfr[ks], be, 0, 0, 0);
                }
                const f32x4 o = (xq[blk] * rs) * ga + be;
                uint2 pk;
                pk.x = pack_bf2(o[0], o[1]); pk.y = pack_bf2(o[2], o[3]);
                *reinterpret_cast<uint2*>(stg + c * STG_ROW + (16 * j + 4 * g) * 2) = pk;
                gpk[j].x = pack_bf2(ga[0], ga[1]); gpk[j].y = pack_bf2(ga[2], ga[3]);
            }
            // the slab holds [16 tokens][64 features] bf16: 4 lanes per token take 32 bytes each = one whole 128-byte line per token.
            // Lane (c, g) wrote row c, lane (row, pc) reads row `row`: a cross-lane exchange inside ONE wave -- the wave-level fence + wave
            // barrier order the slab writes before the reads (and the reads before the next rewrite) in the compiler's eyes; the hardware
            // executes a wave's LDS instructions in order, so they cost no instruction.
            const int row = lane >> 2, pc = lane & 3;
            SLAB_SYNC();
            const uint4 v0 = *reinterpret_cast<const uint4*>(stg + row * STG_ROW + pc * 32);
            const uint4 v1 = *reinterpret_cast<const uint4*>(stg + row * STG_ROW + pc * 32 + 16);
            if (t0 + row < T) {
                bf16_t* yp = y + (long)(t0 + row) * ldy + 64 * ch + pc * 16;
                *reinterpret_cast<uint4*>(yp) = v0;
                *reinterpret_cast<uint4*>(yp + 8) = v1;
            }
            SLAB_SYNC();
            if (gamma_out) {   // the gamma rows for the backward (bf16 [T, D]): dx needs gamma_t, nothing needs beta_t
#pragma unroll
                for (int j = 0; j < 4; ++j) *reinterpret_cast<uint2*>(stg + c * STG_ROW + (16 * j + 4 * g) * 2) = gpk[j];
                SLAB_SYNC();
                const uint4 g0 = *reinterpret_cast<const uint4*>(stg + row * STG_ROW + pc * 32);
                const uint4 g1 = *reinterpret_cast<const uint4*>(stg + row * STG_ROW + pc * 32 + 16);
                if (t0 + row < T) {
                    bf16_t* gp = gamma_out + (long)(t0 + row) * ldg + 64 * ch + pc * 16;
                    *reinterpret_cast<uint4*>(gp) = g0;
                    *reinterpret_cast<uint4*>(gp + 8) = g1;
                }
                SLAB_SYNC();
            }
        }
    }
}

std::atomic<unsigned> g_fwd_optin{0};

}  // namespace

// 1 when the fused kernels take the shape (D = 512 features, 64 condition channels)
extern "C" int spn_adaln_ok(int D, int C) { return D == AD && C == AC; }

// y [T, D] bf16 = (W_gamma cond_t + b_gamma) * LayerNorm(x_t) + (W_beta cond_t + b_beta);  x fp32 [T, D] (ldx), cond bf16 [T, C] (ldc),
// W bf16 [2D, C] contiguous (gamma rows, then beta rows), bias fp32 [2D]; mean / rstd [T] and (optionally) the gamma rows
// gamma_out bf16 [T, D] are written for the backward (spn_layernorm_bwd_gb16 with ldgb = ldg reads exactly those).
extern "C" int spn_adaln_fwd(const float* x, long ldx, const void* cond, long ldc, const void* W, const float* bias, void* y, long ldy,
                             void* gamma_out, long ldg, float* mean, float* rstd, int T, int D, int C, float eps, hipStream_t s) {
    SPN_REQUIRE(spn_adaln_ok(D, C), "spn_adaln_fwd: D = 512, C = 64 only (use the unfused path otherwise)");
    SPN_REQUIRE(x && cond && W && bias && y && mean && rstd && T > 0, "spn_adaln_fwd: null argument");
    SPN_REQUIRE((ldx % 4) == 0 && (ldc % 8) == 0 && (ldy % 8) == 0 && (ldg % 8) == 0 &&
                (((uintptr_t)x | (uintptr_t)cond | (uintptr_t)W | (uintptr_t)y | (uintptr_t)gamma_out) & 15) == 0,
                "spn_adaln_fwd: rows must be 16-byte aligned");
    spn_lds_optin(g_fwd_optin, (const void*)adaln_fwd_kernel, FWD_LDS);
    const int ngroups = (T + 15) / 16;
    const int blocks = ngroups >= 8 * 256 ? 256 : (ngroups + 7) / 8;
    hipLaunchKernelGGL(adaln_fwd_kernel, dim3(blocks), dim3(512), FWD_LDS, s, x, ldx, (const bf16_t*)cond, ldc, (const bf16_t*)W, bias,
                       (bf16_t*)y, ldy, (bf16_t*)gamma_out, ldg, mean, rstd, T, eps);
    SPN_LAUNCH_CHECK();
    return SPN_OK;
}

