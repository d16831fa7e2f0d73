// LayerNorm / AdaptiveLayerNorm forward + backward for gfx950 (HBM-bound; one 64-lane wave per row, float4 I/O).
//
// Replaces nn.LayerNorm / F.layer_norm (eps 1e-5) at modules/transformer/transformer.py:106,123-125,192-193,217,
// models/scoreperformer/transformer.py:121,171, models/scoreperformer/embeddings.py:101,139,341,346 and
// `AdaptiveLayerNorm.forward` (modules/layers.py:41-47):  y = gamma_t * LN(x) + beta_t with per-token
// (gamma_t, beta_t) = Linear(cond).chunk(2) supplied as a [T, 2D] fp32 tensor.
// Statistics are fp32 two-pass (mean, then centred variance) like ATen's CPU kernel.
#include "common.h"
#include "tuning.h"

namespace {

template <typename T> struct IO;
template <> struct IO<float> {
    static __device__ __forceinline__ f32x4 load4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
    static __device__ __forceinline__ void store4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
};
template <> struct IO<bf16_t> {
    static __device__ __forceinline__ f32x4 load4(const bf16_t* p) {
        uint2 u = *reinterpret_cast<const uint2*>(p);
        return f32x4{bf2f(u.x & 0xffff), bf2f(u.x >> 16), bf2f(u.y & 0xffff), bf2f(u.y >> 16)};
    }
    static __device__ __forceinline__ void store4(bf16_t* p, f32x4 v) {
        uint2 u; u.x = pack_bf2(v[0], v[1]); u.y = pack_bf2(v[2], v[3]);
        *reinterpret_cast<uint2*>(p) = u;
    }
};

// ---------------------------------------------------------------------------------------------------------
template <typename TIn, typename TOut, int NV>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const TIn* __restrict__ x, long ldx, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, const void* __restrict__ gb_, long ldgb, int gb16,
                                                     TOut* __restrict__ y, long ldy, float* __restrict__ mean,
                                                     float* __restrict__ rstd, int T, int D, float eps) {
    const float* gb = reinterpret_cast<const float*>(gb_);            // (gamma | beta) rows: fp32, or bf16 when gb16
    const bf16_t* gbh = reinterpret_cast<const bf16_t*>(gb_);
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= T) return;
    const TIn* xr = x + (long)row * ldx;
    f32x4 v[NV];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int col = (lane + 64 * i) * 4;
        v[i] = col < D ? IO<TIn>::load4(xr + col) : f32x4{0.f, 0.f, 0.f, 0.f};
        sum += v[i][0] + v[i][1] + v[i][2] + v[i][3];
    }
    const float mu = wave_sum(sum) / (float)D;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int col = (lane + 64 * i) * 4;
        if (col < D) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mu; sq += d * d; }
        }
    }
    const float rs = rsqrtf(wave_sum(sq) / (float)D + eps);
    if (lane == 0 && mean) { mean[row] = mu; rstd[row] = rs; }
    TOut* yr = y + (long)row * ldy;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int col = (lane + 64 * i) * 4;
        if (col >= D) continue;
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mu) * rs;
        if (gb) {
            const f32x4 ga = gb16 ? IO<bf16_t>::load4(gbh + (long)row * ldgb + col) : *reinterpret_cast<const f32x4*>(gb + (long)row * ldgb + col);
            const f32x4 be = gb16 ? IO<bf16_t>::load4(gbh + (long)row * ldgb + D + col) : *reinterpret_cast<const f32x4*>(gb + (long)row * ldgb + D + col);
            o = o * ga + be;
        } else if (gamma) {
            const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + col);
            const f32x4 be = *reinterpret_cast<const f32x4*>(beta + col);
            o = o * ga + be;
        }
        IO<TOut>::store4(yr + col, o);
    }
}

// backward.  dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma
// dgamma/dbeta (affine): per-block register partials -> LDS -> one atomicAdd per column per block.
// dgb (adaptive): [T, 2D] bf16 rows (dy * xhat | dy), consumed as a GEMM operand by the AdaLN linear's backward.
template <typename TIn, typename TDx, int NV>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const TIn* __restrict__ x, long ldx, const bf16_t* __restrict__ dy, long lddy,
                                                     const float* __restrict__ gamma, const void* __restrict__ gb_, long ldgb, int gb16,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     const float* __restrict__ dres, long lddres, TDx* __restrict__ dx, long lddx,
                                                     bf16_t* __restrict__ dx16, long lddx16,
                                                     float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                     bf16_t* __restrict__ dgb, long lddgb, int T, int D, int rows_per_block) {
    __shared__ float red[4][64 * NV * 4 + 4];
    const float* gb = reinterpret_cast<const float*>(gb_);
    const bf16_t* gbh = reinterpret_cast<const bf16_t*>(gb_);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    f32x4 pg[NV], pb[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) { pg[i] = f32x4{0.f, 0.f, 0.f, 0.f}; pb[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    const int row_begin = blockIdx.x * rows_per_block;
    const int row_end = min(T, row_begin + rows_per_block);
    // the affine weight is the same for every row: held in registers (re-read per row it was one more L2 round trip in front of each
    // row's arithmetic, 2 KB per row and wave)
    f32x4 gam_r[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int col = (lane + 64 * i) * 4;
        gam_r[i] = (!gb && gamma && col < D) ? *reinterpret_cast<const f32x4*>(gamma + col) : f32x4{1.f, 1.f, 1.f, 1.f};
    }
    for (int row = row_begin + w; row < row_end; row += 4) {
        const float mu = mean[row], rs = rstd[row];
        f32x4 xh[NV], gq[NV], dr[NV];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {   // the residual gradient is requested with the other operands, not behind the two reductions
            const int col = (lane + 64 * i) * 4;
            dr[i] = (dres && col < D) ? *reinterpret_cast<const f32x4*>(dres + (long)row * lddres + col) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int col = (lane + 64 * i) * 4;
            xh[i] = f32x4{0.f, 0.f, 0.f, 0.f}; gq[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (col >= D) continue;
            const f32x4 xv = IO<TIn>::load4(x + (long)row * ldx + col);
            const f32x4 d = IO<bf16_t>::load4(dy + (long)row * lddy + col);
            f32x4 ga = gam_r[i];
            if (gb) ga = gb16 ? IO<bf16_t>::load4(gbh + (long)row * ldgb + col) : *reinterpret_cast<const f32x4*>(gb + (long)row * ldgb + col);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                xh[i][e] = (xv[e] - mu) * rs;
                gq[i][e] = d[e] * ga[e];
                s1 += gq[i][e];
                s2 += gq[i][e] * xh[i][e];
            }
            if (dgb) {
                f32x4 t;
#pragma unroll
                for (int e = 0; e < 4; ++e) t[e] = d[e] * xh[i][e];
                IO<bf16_t>::store4(dgb + (long)row * lddgb + col, t);
                IO<bf16_t>::store4(dgb + (long)row * lddgb + D + col, d);
            }
            if (dgamma) {   // affine: the gamma / beta gradients; adaptive (with dgb): the column sums of the dgb rows = the condition Linear's bias gradient
#pragma unroll
                for (int e = 0; e < 4; ++e) { pg[i][e] += d[e] * xh[i][e]; pb[i][e] += d[e]; }
            }
        }
        s1 = wave_sum(s1) / (float)D;
        s2 = wave_sum(s2) / (float)D;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int col = (lane + 64 * i) * 4;
            if (col >= D) continue;
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = rs * (gq[i][e] - s1 - xh[i][e] * s2);
            o += dr[i];
            IO<TDx>::store4(dx + (long)row * lddx + col, o);
            if (dx16) IO<bf16_t>::store4(dx16 + (long)row * lddx16 + col, o);   // bf16 copy = the next backward GEMM's operand
        }
    }
    if (dgamma) {
        for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
            for (int i = 0; i < NV; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) red[w][(lane + 64 * i) * 4 + e] = pass == 0 ? pg[i][e] : pb[i][e];
            __syncthreads();
            for (int col = threadIdx.x; col < D; col += 256) {
                const float s = red[0][col] + red[1][col] + red[2][col] + red[3][col];
                atomicAdd((pass == 0 ? dgamma : dbeta) + col, s);
            }
            __syncthreads();
        }
    }
}

// The two launches of the residual stream's norms (39 per step at C3), without a single condition inside the row loop: x, the residual
// gradient and dx are fp32, dy bf16, the bf16 copy of dx is wanted, the row is exactly NV * 256 wide.  ADA = false: affine weight, the
// gamma / beta gradients accumulate; ADA = true: per-token bf16 gamma rows, the (dy * xhat | dy) rows are written.  Same arithmetic and
// association order as ln_bwd_kernel (results are bit-identical); what differs is the instruction stream: in the general kernel every
// load sits behind a run-time null / width test in its own basic block, and hipcc's wait-count pass then waits for the FIRST 256-column
// chunk's loads (and, at the loop header, for the previous row's stores) before the second chunk's loads are issued -- half the bytes
// in flight per wave and one more exposed latency per row (4.9 TB/s where the access mix streams 5.85, profiles/r04_ln_bwd_variants.txt).
// FORK = the residual gradient is added and the bf16 copy of an fp32 dx is written (the pre-norm fork of the layer stack); without it
// (norms outside the stack: embedding norms, the LM head's 1536-wide norm, bf16 in and out) neither pointer is touched.
template <typename TIn, typename TDx, int NV, bool ADA, bool FORK>
__global__ __launch_bounds__(256) void ln_bwd_fast_kernel(const TIn* __restrict__ x, long ldx, const bf16_t* __restrict__ dy, long lddy,
                                                          const float* __restrict__ gamma, const bf16_t* __restrict__ gbh, long ldgb,
                                                          const float* __restrict__ mean, const float* __restrict__ rstd,
                                                          const float* __restrict__ dres, long lddres, TDx* __restrict__ dx, long lddx,
                                                          bf16_t* __restrict__ dx16, long lddx16, float* __restrict__ dgamma,
                                                          float* __restrict__ dbeta, bf16_t* __restrict__ dgb, long lddgb, int T,
                                                          int rows_per_block) {
    constexpr int D = NV * 256;
    __shared__ float red[4][64 * NV * 4 + 4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    f32x4 pg[NV], pb[NV], gam_r[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        pg[i] = f32x4{0.f, 0.f, 0.f, 0.f}; pb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        gam_r[i] = ADA ? f32x4{1.f, 1.f, 1.f, 1.f} : *reinterpret_cast<const f32x4*>(gamma + (lane + 64 * i) * 4);
    }
    const int row_begin = blockIdx.x * rows_per_block;
    const int row_end = min(T, row_begin + rows_per_block);
    for (int row = row_begin + w; row < row_end; row += 4) {
        // every operand of the row is requested before anything is waited for
        f32x4 xv[NV], dr[NV], ga[NV];
        uint2 du[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int col = (lane + 64 * i) * 4;
            xv[i] = IO<TIn>::load4(x + (long)row * ldx + col);
            du[i] = *reinterpret_cast<const uint2*>(dy + (long)row * lddy + col);
            dr[i] = FORK ? *reinterpret_cast<const f32x4*>(dres + (long)row * lddres + col) : f32x4{0.f, 0.f, 0.f, 0.f};
            ga[i] = ADA ? IO<bf16_t>::load4(gbh + (long)row * ldgb + col) : gam_r[i];
        }
        const float mu = mean[row], rs = rstd[row];
        f32x4 xh[NV], gq[NV];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int col = (lane + 64 * i) * 4;
            const f32x4 d = f32x4{bf2f(du[i].x & 0xffff), bf2f(du[i].x >> 16), bf2f(du[i].y & 0xffff), bf2f(du[i].y >> 16)};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                xh[i][e] = (xv[i][e] - mu) * rs;
                gq[i][e] = d[e] * ga[i][e];
                s1 += gq[i][e];
                s2 += gq[i][e] * xh[i][e];
            }
            if (ADA) {
                f32x4 t;
#pragma unroll
                for (int e = 0; e < 4; ++e) t[e] = d[e] * xh[i][e];
                IO<bf16_t>::store4(dgb + (long)row * lddgb + col, t);
                IO<bf16_t>::store4(dgb + (long)row * lddgb + D + col, d);
            }
            if (!ADA || dgamma) {   // (adaptive + dgamma: the column sums of the rows just written = the bias gradient of the condition Linear)
#pragma unroll
                for (int e = 0; e < 4; ++e) { pg[i][e] += d[e] * xh[i][e]; pb[i][e] += d[e]; }
            }
        }
        s1 = wave_sum(s1) / (float)D;
        s2 = wave_sum(s2) / (float)D;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int col = (lane + 64 * i) * 4;
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = rs * (gq[i][e] - s1 - xh[i][e] * s2);
            if (FORK) o += dr[i];
            IO<TDx>::store4(dx + (long)row * lddx + col, o);
            if (FORK) IO<bf16_t>::store4(dx16 + (long)row * lddx16 + col, o);
        }
    }
    if (!ADA || dgamma) {
        for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
            for (int i = 0; i < NV; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) red[w][(lane + 64 * i) * 4 + e] = pass == 0 ? pg[i][e] : pb[i][e];
            __syncthreads();
            for (int col = threadIdx.x; col < D; col += 256) {
                const float sum = red[0][col] + red[1][col] + red[2][col] + red[3][col];
                atomicAdd((pass == 0 ? dgamma : dbeta) + col, sum);
            }
            __syncthreads();
        }
    }
}

template <typename TIn, typename TOut>
int launch_fwd(int nv, dim3 grid, hipStream_t s, const void* x, long ldx, const float* gamma, const float* beta, const void* gb,
               long ldgb, int gb16, void* y, long ldy, float* mean, float* rstd, int T, int D, float eps) {
#define CASE(NV_)                                                                                                        \
    case NV_:                                                                                                            \
        hipLaunchKernelGGL((ln_fwd_kernel<TIn, TOut, NV_>), grid, dim3(256), 0, s, (const TIn*)x, ldx, gamma, beta, gb,  \
                           ldgb, gb16, (TOut*)y, ldy, mean, rstd, T, D, eps);                                            \
        break;
    switch (nv) { CASE(1) CASE(2) CASE(4) CASE(6) CASE(8) default: return SPN_ERR_ARG; }
#undef CASE
    return SPN_OK;
}

template <typename TIn, typename TDx>
int launch_bwd(int nv, dim3 grid, hipStream_t s, const void* x, long ldx, const void* dy, long lddy, const float* gamma,
               const void* gb, long ldgb, int gb16, const float* mean, const float* rstd, const float* dres, long lddres, void* dx,
               long lddx, bf16_t* dx16, long lddx16, float* dgamma, float* dbeta, bf16_t* dgb, long lddgb, int T, int D, int rpb) {
#define CASE(NV_)                                                                                                        \
    case NV_:                                                                                                            \
        hipLaunchKernelGGL((ln_bwd_kernel<TIn, TDx, NV_>), grid, dim3(256), 0, s, (const TIn*)x, ldx, (const bf16_t*)dy, \
                           lddy, gamma, gb, ldgb, gb16, mean, rstd, dres, lddres, (TDx*)dx, lddx, dx16, lddx16, dgamma, dbeta, dgb, lddgb, T, D, \
                           rpb);                                                                                         \
        break;
    switch (nv) { CASE(1) CASE(2) CASE(4) CASE(6) CASE(8) default: return SPN_ERR_ARG; }
#undef CASE
    return SPN_OK;
}

static inline int round_nv(int nv) { return nv <= 2 ? nv : nv <= 4 ? 4 : nv <= 6 ? 6 : 8; }

}  // namespace

namespace {
int ln_fwd_impl(const void* x, int x_dtype, long ldx, const float* gamma, const float* beta, const void* gb, long ldgb, int gb16, void* y,
                int y_dtype, long ldy, float* mean, float* rstd, int T, int D, float eps, hipStream_t stream) {
    SPN_REQUIRE(x && y && T > 0 && D > 0, "spn_layernorm_fwd: bad arguments");
    SPN_REQUIRE(D % 4 == 0 && D <= 2048 && ldx % 4 == 0 && ldy % 4 == 0 && ldgb % 4 == 0,
                "spn_layernorm_fwd: D must be a multiple of 4 and <= 2048; leading dims multiples of 4");
    const int nv = round_nv((D + 255) / 256);
    dim3 grid(cdiv(T, 4));
    int rc;
    if (x_dtype == 0 && y_dtype == 1) rc = launch_fwd<float, bf16_t>(nv, grid, stream, x, ldx, gamma, beta, gb, ldgb, gb16, y, ldy, mean, rstd, T, D, eps);
    else if (x_dtype == 0 && y_dtype == 0) rc = launch_fwd<float, float>(nv, grid, stream, x, ldx, gamma, beta, gb, ldgb, gb16, y, ldy, mean, rstd, T, D, eps);
    else if (x_dtype == 1 && y_dtype == 1) rc = launch_fwd<bf16_t, bf16_t>(nv, grid, stream, x, ldx, gamma, beta, gb, ldgb, gb16, y, ldy, mean, rstd, T, D, eps);
    else rc = launch_fwd<bf16_t, float>(nv, grid, stream, x, ldx, gamma, beta, gb, ldgb, gb16, y, ldy, mean, rstd, T, D, eps);
    if (rc) { spn_set_error("spn_layernorm_fwd: unsupported width"); return rc; }
    SPN_LAUNCH_CHECK();
    return SPN_OK;
}

int ln_bwd_impl(const void* x, int x_dtype, long ldx, const void* dy, long lddy, const float* gamma, const void* gb, long ldgb, int gb16,
                const float* mean, const float* rstd, const float* dres, long lddres, void* dx, int dx_dtype, long lddx, void* dx16,
                long lddx16, float* dgamma, float* dbeta, void* dgb, long lddgb, int T, int D, hipStream_t stream) {
    SPN_REQUIRE(x && dy && mean && rstd && dx && T > 0 && D > 0, "spn_layernorm_bwd: bad arguments");
    SPN_REQUIRE(D % 4 == 0 && D <= 2048 && ldx % 4 == 0 && lddy % 4 == 0 && lddx % 4 == 0 && ldgb % 4 == 0 && lddgb % 4 == 0 &&
                lddres % 4 == 0 && lddx16 % 4 == 0, "spn_layernorm_bwd: D must be a multiple of 4 and <= 2048; leading dims multiples of 4");
    const int nv = round_nv((D + 255) / 256);
    const int bwd_blocks = spn_tune_i(SPN_TUNE_LN_BWD_BLOCKS) > 0 ? spn_tune_i(SPN_TUNE_LN_BWD_BLOCKS) : 2048;   // tuning aid
    int rpb = cdiv(T, bwd_blocks);
    rpb = ((rpb + 3) / 4) * 4;
    dim3 grid(cdiv(T, rpb));
    // the residual stream's two launches (fp32 x / dx / residual gradient, bf16 copy wanted, 256- or 512-wide rows): the branch-free kernel
#define LNF(TI_, TD_, NV_, ADA_, FORK_) hipLaunchKernelGGL((ln_bwd_fast_kernel<TI_, TD_, NV_, ADA_, FORK_>), grid, dim3(256), 0, stream, (const TI_*)x, ldx, \
        (const bf16_t*)dy, lddy, gamma, (const bf16_t*)gb, ldgb, mean, rstd, dres, lddres, (TD_*)dx, lddx, (bf16_t*)dx16, lddx16, dgamma, dbeta, (bf16_t*)dgb, lddgb, T, rpb)
    if (x_dtype == 0 && dx_dtype == 0 && dres && dx16 && (D == 256 || D == 512) &&
        ((gb && gb16 && dgb && !gamma && (!dgamma) == (!dbeta)) || (!gb && gamma && dgamma && dbeta && !dgb))) {
        const bool ada = gb != nullptr;
        if (D == 512) { if (ada) LNF(float, float, 2, true, true); else LNF(float, float, 2, false, true); }
        else { if (ada) LNF(float, float, 1, true, true); else LNF(float, float, 1, false, true); }
        SPN_LAUNCH_CHECK();
        return SPN_OK;
    }
    // affine norms outside the layer stack, bf16 in and out, rows of exactly 512 or 1536 (the embedding norms, the LM head's norm)
    if (x_dtype == 1 && dx_dtype == 1 && !dres && !dx16 && !gb && !dgb && gamma && dgamma && dbeta && (D == 512 || D == 1536)) {
        if (D == 512) LNF(bf16_t, bf16_t, 2, false, false); else LNF(bf16_t, bf16_t, 6, false, false);
        SPN_LAUNCH_CHECK();
        return SPN_OK;
    }
#undef LNF
    int rc;
#define LNB(TI_, TD_) launch_bwd<TI_, TD_>(nv, grid, stream, x, ldx, dy, lddy, gamma, gb, ldgb, gb16, mean, rstd, dres, lddres, dx, lddx, (bf16_t*)dx16, lddx16, dgamma, dbeta, (bf16_t*)dgb, lddgb, T, D, rpb)
    if (x_dtype == 0 && dx_dtype == 0) rc = LNB(float, float);
    else if (x_dtype == 0 && dx_dtype == 1) rc = LNB(float, bf16_t);
    else if (x_dtype == 1 && dx_dtype == 0) rc = LNB(bf16_t, float);
    else rc = LNB(bf16_t, bf16_t);
#undef LNB
    if (rc) { spn_set_error("spn_layernorm_bwd: unsupported width"); return rc; }
    SPN_LAUNCH_CHECK();
    return SPN_OK;
}
}  // namespace

// dtype codes: 0 = fp32, 1 = bf16.  gamma/beta: [D] fp32 or null;  gb: [T, 2D] fp32 (ldgb) or null.
extern "C" int spn_layernorm_fwd(const void* x, int x_dtype, long ldx, const float* gamma, const float* beta, const float* gb,
                                 long ldgb, void* y, int y_dtype, long ldy, float* mean, float* rstd, int T, int D, float eps,
                                 hipStream_t stream) {
    return ln_fwd_impl(x, x_dtype, ldx, gamma, beta, gb, ldgb, 0, y, y_dtype, ldy, mean, rstd, T, D, eps, stream);
}
// the adaptive form with the per-token (gamma | beta) rows in bf16 (half the bytes of the largest tensor of an adaptive norm)
extern "C" int spn_layernorm_fwd_gb16(const void* x, int x_dtype, long ldx, const void* gb16, long ldgb, void* y, int y_dtype, long ldy,
                                      float* mean, float* rstd, int T, int D, float eps, hipStream_t stream) {
    SPN_REQUIRE(gb16, "spn_layernorm_fwd_gb16: gb required");
    return ln_fwd_impl(x, x_dtype, ldx, nullptr, nullptr, gb16, ldgb, 1, y, y_dtype, ldy, mean, rstd, T, D, eps, stream);
}

// dx[T,D] (fp32 or bf16) = (dres or 0) + LN backward; dy is bf16.  Affine: dgamma/dbeta [D] fp32 are ACCUMULATED
// (atomics; zero them first).  Adaptive: dgb [T,2D] bf16 rows are written.  dx16 (optional): bf16 copy of dx, row stride lddx16.
extern "C" int spn_layernorm_bwd(const void* x, int x_dtype, long ldx, const void* dy, long lddy, const float* gamma,
                                 const float* gb, long ldgb, const float* mean, const float* rstd, const float* dres,
                                 long lddres, void* dx, int dx_dtype, long lddx, void* dx16, long lddx16, float* dgamma,
                                 float* dbeta, void* dgb, long lddgb, int T, int D, hipStream_t stream) {
    return ln_bwd_impl(x, x_dtype, ldx, dy, lddy, gamma, gb, ldgb, 0, mean, rstd, dres, lddres, dx, dx_dtype, lddx, dx16, lddx16, dgamma,
                       dbeta, dgb, lddgb, T, D, stream);
}
// as spn_layernorm_bwd_gb16, and the column sums of the dgb rows it writes are ACCUMULATED into dgb_colsum [2D] (fp32; the bias gradient of the
// Linear that produced the (gamma | beta) rows) by the same pass -- from the unrounded fp32 products, no second read of dgb
extern "C" int spn_layernorm_bwd_gb16_colsum(const void* x, int x_dtype, long ldx, const void* dy, long lddy, const void* gb16, long ldgb,
                                             const float* mean, const float* rstd, const float* dres, long lddres, void* dx, int dx_dtype,
                                             long lddx, void* dx16, long lddx16, void* dgb, long lddgb, float* dgb_colsum, int T, int D,
                                             hipStream_t stream) {
    SPN_REQUIRE(gb16 && dgb && dgb_colsum, "spn_layernorm_bwd_gb16_colsum: gb, dgb and the column-sum target are required");
    return ln_bwd_impl(x, x_dtype, ldx, dy, lddy, nullptr, gb16, ldgb, 1, mean, rstd, dres, lddres, dx, dx_dtype, lddx, dx16, lddx16,
                       dgb_colsum, dgb_colsum + D, dgb, lddgb, T, D, stream);
}
extern "C" int spn_layernorm_bwd_gb16(const void* x, int x_dtype, long ldx, const void* dy, long lddy, const void* gb16, long ldgb,
                                      const float* mean, const float* rstd, const float* dres, long lddres, void* dx, int dx_dtype,
                                      long lddx, void* dx16, long lddx16, void* dgb, long lddgb, int T, int D, hipStream_t stream) {
    SPN_REQUIRE(gb16, "spn_layernorm_bwd_gb16: gb required");
    return ln_bwd_impl(x, x_dtype, ldx, dy, lddy, nullptr, gb16, ldgb, 1, mean, rstd, dres, lddres, dx, dx_dtype, lddx, dx16, lddx16,
                       nullptr, nullptr, dgb, lddgb, T, D, stream);
}
