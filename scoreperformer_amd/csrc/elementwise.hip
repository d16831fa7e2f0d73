// Element-wise / small reduction kernels (HBM-bound): GLU / activation, casts with row masks, bias gradients,
// Mish, row predicates.  bf16 tensors are processed 8 elements (16 B) per lane.
//
// Replaces: `GLU.forward` x * act(gate) and the non-GLU activation (modules/transformer/feedforward.py:13-21,51-56),
// the `out * mask[..., None]` row masks (modules/transformer/attention.py:216-218,
// models/scoreperformer/mmd_transformer.py:213-214,278), bias gradients of nn.Linear, nn.Mish of the
// dense-continuous embedding MLP (modules/transformer/embeddings.py:202-213).
#include "common.h"

namespace {

// silu_grad / gelu_grad: common.h (shared with the gated-backward GEMM epilogue, which must agree bit for bit)
template <int ACT> __device__ __forceinline__ float act_f(float x) { return ACT == 0 ? silu_f(x) : gelu_f(x); }
template <int ACT> __device__ __forceinline__ float act_g(float x) { return ACT == 0 ? silu_grad(x) : gelu_grad(x); }

__device__ __forceinline__ void unpack8(const uint4& u, float* f) {
    const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) { f[2 * e] = bf2f(w[e] & 0xffff); f[2 * e + 1] = bf2f(w[e] >> 16); }
}
__device__ __forceinline__ uint4 pack8f(const float* f) {
    uint4 u;
    u.x = pack_bf2(f[0], f[1]); u.y = pack_bf2(f[2], f[3]); u.z = pack_bf2(f[4], f[5]); u.w = pack_bf2(f[6], f[7]);
    return u;
}

// out[t, i] = u[t, i] * act(u[t, I + i])     (GLU) ;  GLU=false: out[t,i] = act(u[t,i])
// dropout of the activation output (nn.Dropout between activation and out-projection, feedforward.py:57-60): the 8 elements of
// chunk `chunk` (columns 8 chunk ..) of row t; mask definition in common.h (ffn_drop_bits); kept values are scaled by 1/(1-p)
__device__ __forceinline__ void drop8(float* o, long t, int chunk, uint32_t seed, uint32_t thr16, float keep_scale) {
    const uint32_t rowc = ffn_drop_rowc(t, seed);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint32_t h = ffn_drop_bits(rowc, (uint32_t)chunk * 4u + q);
        o[2 * q] = (h & 0xffffu) >= thr16 ? o[2 * q] * keep_scale : 0.f;
        o[2 * q + 1] = (h >> 16) >= thr16 ? o[2 * q + 1] * keep_scale : 0.f;
    }
}

template <int ACT, bool GLU>
__global__ void act_fwd_kernel(const bf16_t* __restrict__ u, long ldu, bf16_t* __restrict__ out, long ldo, long T, int I,
                               uint32_t thr16, float keep_scale, uint32_t seed) {
    const int chunks = I / 8;
    const long total = T * chunks;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long t = idx / chunks;
        const int c = (idx % chunks) * 8;
        float a[8], g[8], o[8];
        if (GLU) {
            unpack8(*reinterpret_cast<const uint4*>(u + t * ldu + c), a);
            unpack8(*reinterpret_cast<const uint4*>(u + t * ldu + I + c), g);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = a[e] * act_f<ACT>(g[e]);
        } else {
            unpack8(*reinterpret_cast<const uint4*>(u + t * ldu + c), g);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = act_f<ACT>(g[e]);
        }
        if (thr16) drop8(o, t, c >> 3, seed, thr16, keep_scale);
        *reinterpret_cast<uint4*>(out + t * ldo + c) = pack8f(o);
    }
}

template <int ACT, bool GLU>
__global__ void act_bwd_kernel(const bf16_t* __restrict__ u, long ldu, const bf16_t* __restrict__ dout, long lddo,
                               bf16_t* __restrict__ du, long lddu, long T, int I, uint32_t thr16, float keep_scale, uint32_t seed,
                               float* __restrict__ colsum) {
    const int chunks = I / 8;
    const long total = T * chunks;
    // colsum != null (the launcher guarantees grid stride % chunks == 0, so a thread always sees the same 8 columns): every block
    // leaves the column sums of the du rows it produced in row blockIdx.x of `colsum` [gridDim.x, W]; the launcher adds the rows
    // up = the bias gradient of the Linear in front, without re-reading du (1 GB for the C3 FFN)
    float sa[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, sg[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long t = idx / chunks;
        const int c = (idx % chunks) * 8;
        float a[8], g[8], d[8], da[8], dg[8];
        unpack8(*reinterpret_cast<const uint4*>(dout + t * lddo + c), d);
        if (thr16) drop8(d, t, c >> 3, seed, thr16, keep_scale);
        if (GLU) {
            unpack8(*reinterpret_cast<const uint4*>(u + t * ldu + c), a);
            unpack8(*reinterpret_cast<const uint4*>(u + t * ldu + I + c), g);
#pragma unroll
            for (int e = 0; e < 8; ++e) { da[e] = d[e] * act_f<ACT>(g[e]); dg[e] = d[e] * a[e] * act_g<ACT>(g[e]); }
            const uint4 pa = pack8f(da), pg = pack8f(dg);
            *reinterpret_cast<uint4*>(du + t * lddu + c) = pa;
            *reinterpret_cast<uint4*>(du + t * lddu + I + c) = pg;
            if (colsum) {   // sum what the consumer GEMMs will see: the bf16-rounded values
                float ra[8], rg[8];
                unpack8(pa, ra); unpack8(pg, rg);
#pragma unroll
                for (int e = 0; e < 8; ++e) { sa[e] += ra[e]; sg[e] += rg[e]; }
            }
        } else {
            unpack8(*reinterpret_cast<const uint4*>(u + t * ldu + c), g);
#pragma unroll
            for (int e = 0; e < 8; ++e) dg[e] = d[e] * act_g<ACT>(g[e]);
            const uint4 pg = pack8f(dg);
            *reinterpret_cast<uint4*>(du + t * lddu + c) = pg;
            if (colsum) {
                float rg[8];
                unpack8(pg, rg);
#pragma unroll
                for (int e = 0; e < 8; ++e) sg[e] += rg[e];
            }
        }
    }
    if (colsum) {
        const int W = GLU ? 2 * I : I;
        const int c = (int)((((long)blockIdx.x * blockDim.x + threadIdx.x) % chunks) * 8);
        float* row = colsum + (long)blockIdx.x * W;   // blocks with fewer than `chunks` live threads leave zeros (buffer pre-zeroed)
        if ((long)blockIdx.x * blockDim.x + threadIdx.x < total) {
            // several threads of one block share a column chunk when chunks < 256: their partials meet in the block's row
            if (256 % chunks == 0 && chunks < 256) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    if (GLU) { atomicAdd(row + c + e, sa[e]); atomicAdd(row + I + c + e, sg[e]); }
                    else atomicAdd(row + c + e, sg[e]);
                }
            } else {
                if (GLU) {
                    *reinterpret_cast<f32x4*>(row + c) = f32x4{sa[0], sa[1], sa[2], sa[3]};
                    *reinterpret_cast<f32x4*>(row + c + 4) = f32x4{sa[4], sa[5], sa[6], sa[7]};
                    *reinterpret_cast<f32x4*>(row + I + c) = f32x4{sg[0], sg[1], sg[2], sg[3]};
                    *reinterpret_cast<f32x4*>(row + I + c + 4) = f32x4{sg[4], sg[5], sg[6], sg[7]};
                } else {
                    *reinterpret_cast<f32x4*>(row + c) = f32x4{sg[0], sg[1], sg[2], sg[3]};
                    *reinterpret_cast<f32x4*>(row + c + 4) = f32x4{sg[4], sg[5], sg[6], sg[7]};
                }
            }
        }
    }
}

// y[b,t,:] = cast(x[b,t,:] * rowmask[b*t_len + t])  for [B, t_len, D] views with (batch, row) strides; D multiple of 4
template <typename TS, typename TD>
__global__ void cast_kernel(const TS* __restrict__ x, long x_bs, long x_ts, TD* __restrict__ y, long y_bs, long y_ts,
                            const uint8_t* __restrict__ rowmask, long B, long t_len, int D) {
    const int chunks = D / 4;
    const long total = B * t_len * chunks;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long r = idx / chunks;
        const int c = (idx % chunks) * 4;
        const long bb = r / t_len, tt = r % t_len;
        const TS* xp = x + bb * x_bs + tt * x_ts + c;
        TD* yp = y + bb * y_bs + tt * y_ts + c;
        float v[4];
        if constexpr (sizeof(TS) == 4) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(xp);
            v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3];
        } else {
            const uint2 a = *reinterpret_cast<const uint2*>(xp);
            v[0] = bf2f(a.x & 0xffff); v[1] = bf2f(a.x >> 16); v[2] = bf2f(a.y & 0xffff); v[3] = bf2f(a.y >> 16);
        }
        if (rowmask && !rowmask[r]) { v[0] = v[1] = v[2] = v[3] = 0.f; }
        if constexpr (sizeof(TD) == 4) {
            *reinterpret_cast<f32x4*>(yp) = f32x4{v[0], v[1], v[2], v[3]};
        } else {
            uint2 o; o.x = pack_bf2(v[0], v[1]); o.y = pack_bf2(v[2], v[3]);
            *reinterpret_cast<uint2*>(yp) = o;
        }
    }
}

// out[n] += sum_t x[t, n]   (bias gradient). block = 256 threads: 64 columns x 4 row groups; grid.x = column tiles,
// grid.y = row slabs; one atomicAdd per column per block.
template <typename TS>
__global__ void colsum_kernel(const TS* __restrict__ x, long ldx, float* __restrict__ out, long T, int N, int rows_per_block) {
    __shared__ float red[4][64];
    const int col = blockIdx.x * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6;
    const long r0 = (long)blockIdx.y * rows_per_block;
    const long r1 = min(T, r0 + rows_per_block);
    float acc = 0.f;
    if (col < N)
        for (long t = r0 + rg; t < r1; t += 4) {
            if constexpr (sizeof(TS) == 4) acc += x[t * ldx + col];
            else acc += bf2f(x[t * ldx + col]);
        }
    red[rg][threadIdx.x & 63] = acc;
    __syncthreads();
    if (rg == 0 && col < N) atomicAdd(out + col, red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// bf16, N % 8 == 0, 16-byte aligned rows: each lane streams 8 columns (16 B) per row; block = 32 column groups x 8 row groups
__global__ __launch_bounds__(256) void colsum_bf16x8_kernel(const bf16_t* __restrict__ x, long ldx, float* __restrict__ out, long T, int N,
                                                            int rows_per_block) {
    __shared__ float red[8][32][9];
    const int cg = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int col = (blockIdx.x * 32 + cg) * 8;
    const long r0 = (long)blockIdx.y * rows_per_block;
    const long r1 = min(T, r0 + rows_per_block);
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (col < N) {
        long t = r0 + rg;
        for (; t + 56 < r1; t += 64) {   // 8 row loads in flight per lane: a block holds few waves, one load each left HBM idle
            uint4 u[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) u[q] = *reinterpret_cast<const uint4*>(x + (t + 8 * q) * ldx + col);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                float f[8];
                unpack8(u[q], f);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += f[e];
            }
        }
        for (; t < r1; t += 8) {
            float f[8];
            unpack8(*reinterpret_cast<const uint4*>(x + t * ldx + col), f);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += f[e];
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[rg][cg][e] = acc[e];
    __syncthreads();
    const int c2 = threadIdx.x;          // 256 threads = 32 groups x 8 columns
    const int g2 = c2 >> 3, e2 = c2 & 7;
    const int ocol = (blockIdx.x * 32 + g2) * 8 + e2;
    if (ocol < N) {
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) s += red[r][g2][e2];
        atomicAdd(out + ocol, s);
    }
}

__global__ void mish_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float v = x[i];
        const float sp = v > 20.f ? v : log1pf(__expf(v));
        y[i] = v * tanhf(sp);
    }
}
__global__ void mish_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float v = x[i];
        const float sp = v > 20.f ? v : log1pf(__expf(v));
        const float th = tanhf(sp);
        const float sg = 1.f / (1.f + __expf(-v));
        dx[i] = dy[i] * (th + v * (1.f - th * th) * sg);
    }
}

// mask[r] = all(x[r, :] != 0)     (mmd_transformer.py:342); one wave per row
__global__ void rows_all_nonzero_kernel(const float* __restrict__ x, long ldx, uint8_t* __restrict__ mask, long R, int D) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= R) return;
    const int lane = threadIdx.x & 63;
    bool ok = true;
    for (int c = lane; c < D; c += 64) ok = ok && (x[row * ldx + c] != 0.f);
    const unsigned long long b = __ballot(ok);
    if (lane == 0) mask[row] = (b == ~0ull) ? 1 : 0;
}

// y[r, :] = x[r, :] * (m[r] ? 1 : 0)  fp32 rows (latents * latents_mask etc.)
__global__ void mask_rows_kernel(const float* __restrict__ x, long ldx, const uint8_t* __restrict__ m, float* __restrict__ y, long ldy,
                                 long R, int D, int invert) {
    const long total = R * D;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long r = idx / D;
        const int c = idx % D;
        const bool keep = (m[r] != 0) != (invert != 0);
        y[r * ldy + c] = keep ? x[r * ldx + c] : 0.f;
    }
}

inline int grid_for(long total, int block = 256) { long g = (total + block - 1) / block; return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g)); }

// x[r, :] = 0 for the rows with m[r] == 0, in place (bf16 [R, D], D % 8 == 0): only the masked rows are touched, so a batch
// without padding costs one pass over the mask bytes
__global__ void zero_masked_rows_kernel(bf16_t* __restrict__ x, long ldx, const uint8_t* __restrict__ m, long R, int D) {
    // a wave looks at 64 rows' mask bytes with ONE coalesced load and walks only the masked ones (a row's byte fetched by the whole wave
    // is a scalar load: one dependent trip to memory per row and wave, whether or not anything is masked)
    const int lane = threadIdx.x & 63;
    for (long r0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 64; r0 < R; r0 += (long)gridDim.x * 4 * 64) {
        unsigned long long dead = __ballot(r0 + lane < R && m[r0 + lane] == 0);
        while (dead) {
            const int i = __builtin_ctzll(dead);
            dead &= dead - 1;
            for (int c = lane * 8; c < D; c += 512) *reinterpret_cast<uint4*>(x + (r0 + i) * ldx + c) = uint4{0u, 0u, 0u, 0u};
        }
    }
}

}  // namespace

// act: 0 = SiLU, 1 = GELU(erf).  glu != 0: u is [T, 2I] (value | gate), out [T, I];  glu == 0: u [T, I].
static inline uint32_t thr16_of(float p) { const float t = p * 65536.f; return t <= 0.f ? 0u : (t >= 65535.f ? 65535u : (uint32_t)(t + 0.5f)); }

// p_drop > 0: dropout of the output with keep probability 1 - round(p*65536)/65536 and mask bits from (seed, element index)
extern "C" int spn_act_fwd(const void* u, long ldu, void* out, long ldo, long T, int I, int act, int glu, float p_drop,
                           unsigned seed, hipStream_t s) {
    SPN_REQUIRE(u && out && T > 0 && I > 0 && I % 8 == 0 && ldu % 8 == 0 && ldo % 8 == 0, "spn_act_fwd: bad arguments (I, ld multiples of 8)");
    const int g = grid_for(T * (I / 8));
    const bf16_t* up = (const bf16_t*)u; bf16_t* op = (bf16_t*)out;
    const uint32_t thr = thr16_of(p_drop);
    const float ks = 1.f / (1.f - (float)thr / 65536.f);
    if (act == 0 && glu) hipLaunchKernelGGL((act_fwd_kernel<0, true>), dim3(g), dim3(256), 0, s, up, ldu, op, ldo, T, I, thr, ks, seed);
    else if (act == 0) hipLaunchKernelGGL((act_fwd_kernel<0, false>), dim3(g), dim3(256), 0, s, up, ldu, op, ldo, T, I, thr, ks, seed);
    else if (glu) hipLaunchKernelGGL((act_fwd_kernel<1, true>), dim3(g), dim3(256), 0, s, up, ldu, op, ldo, T, I, thr, ks, seed);
    else hipLaunchKernelGGL((act_fwd_kernel<1, false>), dim3(g), dim3(256), 0, s, up, ldu, op, ldo, T, I, thr, ks, seed);
    SPN_LAUNCH_CHECK();
    return SPN_OK;
}

// colsum (optional, fp32 [W], W = 2I for GLU / I): column sums of du are ACCUMULATED into it (bias gradient of the producing
// Linear); needs colsum_ws = scratch of 4096 * W floats (per-block partial rows, reduced here)
extern "C" int spn_act_bwd(const void* u, long ldu, const void* dout, long lddo, void* du, long lddu, long T, int I, int act,
                           int glu, float p_drop, unsigned seed, float* colsum, float* colsum_ws, hipStream_t s) {
    SPN_REQUIRE(u && dout && du && T > 0 && I > 0 && I % 8 == 0 && ldu % 8 == 0 && lddo % 8 == 0 && lddu % 8 == 0,
                "spn_act_bwd: bad arguments (I, ld multiples of 8)");
    int g = grid_for(T * (I / 8));
    const int W = glu ? 2 * I : I;
    float* partial = nullptr;
    if (colsum) {   // a grid stride that keeps thread <-> column chunk fixed
        const int chunks = I / 8;
        SPN_REQUIRE(colsum_ws, "spn_act_bwd: fused column sums need the scratch buffer");
        SPN_REQUIRE(chunks % 256 == 0 || 256 % chunks == 0, "spn_act_bwd: fused column sums need I/8 to divide or be a multiple of 256");
        if (g > 1024) g = 1024;   // 4 blocks per CU still stream at full rate; 4x fewer partial rows to reduce
        if (chunks > 256) { const int per = chunks / 256; g = (g / per) * per; if (g < per) g = per; }
        partial = colsum_ws;
        hipMemsetAsync(partial, 0, (size_t)g * W * 4, s);
    }
    const bf16_t* up = (const bf16_t*)u; const bf16_t* dp = (const bf16_t*)dout; bf16_t* op = (bf16_t*)du;
    const uint32_t thr = thr16_of(p_drop);
    const float ks = 1.f / (1.f - (float)thr / 65536.f);
    if (act == 0 && glu) hipLaunchKernelGGL((act_bwd_kernel<0, true>), dim3(g), dim3(256), 0, s, up, ldu, dp, lddo, op, lddu, T, I, thr, ks, seed, partial);
    else if (act == 0) hipLaunchKernelGGL((act_bwd_kernel<0, false>), dim3(g), dim3(256), 0, s, up, ldu, dp, lddo, op, lddu, T, I, thr, ks, seed, partial);
    else if (glu) hipLaunchKernelGGL((act_bwd_kernel<1, true>), dim3(g), dim3(256), 0, s, up, ldu, dp, lddo, op, lddu, T, I, thr, ks, seed, partial);
    else hipLaunchKernelGGL((act_bwd_kernel<1, false>), dim3(g), dim3(256), 0, s, up, ldu, dp, lddo, op, lddu, T, I, thr, ks, seed, partial);
    if (partial) {   // colsum += sum over the g partial rows
        int slabs = (g + 63) / 64;   // W/64 x g/64 blocks: enough of them to stream the partial rows at full rate
        const int rpb = (g + slabs - 1) / slabs;
        hipLaunchKernelGGL((colsum_kernel<float>), dim3(cdiv(W, 64), cdiv(g, rpb)), dim3(256), 0, s, partial, (long)W, colsum, (long)g, W, rpb);
    }
    SPN_LAUNCH_CHECK();
    return SPN_OK;
}

// dtype: 0 = fp32, 1 = bf16.  y[b,t,:] = cast(x[b,t,:]) * (rowmask ? rowmask[b*t_len+t] != 0 : 1); element strides (batch, row).
extern "C" int spn_cast(const void* x, int x_dtype, long x_bs, long x_ts, void* y, int y_dtype, long y_bs, long y_ts,
                        const uint8_t* rowmask, long B, long t_len, int D, hipStream_t s) {
    SPN_REQUIRE(x && y && B > 0 && t_len > 0 && D > 0 && D % 4 == 0 && x_bs % 4 == 0 && x_ts % 4 == 0 && y_bs % 4 == 0 && y_ts % 4 == 0,
                "spn_cast: bad arguments (D and strides multiples of 4)");
    const int g = grid_for(B * t_len * (D / 4));
    if (x_dtype == 0 && y_dtype == 1) hipLaunchKernelGGL((cast_kernel<float, bf16_t>), dim3(g), dim3(256), 0, s, (const float*)x, x_bs, x_ts, (bf16_t*)y, y_bs, y_ts, rowmask, B, t_len, D);
    else if (x_dtype == 1 && y_dtype == 0) hipLaunchKernelGGL((cast_kernel<bf16_t, float>), dim3(g), dim3(256), 0, s, (const bf16_t*)x, x_bs, x_ts, (float*)y, y_bs, y_ts, rowmask, B, t_len, D);
    else if (x_dtype == 0 && y_dtype == 0) hipLaunchKernelGGL((cast_kernel<float, float>), dim3(g), dim3(256), 0, s, (const float*)x, x_bs, x_ts, (float*)y, y_bs, y_ts, rowmask, B, t_len, D);
    else hipLaunchKernelGGL((cast_kernel<bf16_t, bf16_t>), dim3(g), dim3(256), 0, s, (const bf16_t*)x, x_bs, x_ts, (bf16_t*)y, y_bs, y_ts, rowmask, B, t_len, D);
    SPN_LAUNCH_CHECK();
    return SPN_OK;
}

// Stand-alone dropout of a [T, D] tensor (nn.Dropout where no producing kernel can carry it: `emb_dropout` of
// models/scoreperformer/transformer.py:122,184 and the Dropout behind a post-activation LayerNorm, feedforward.py:58): y = keep ? x / (1 - p)
// : 0 with the counter-based mask of the FFN kernels (common.h: ffn_drop_bits on (row, column pair), 16-bit threshold).  The mask is a
// function of (seed, row, column) only, so the backward is the SAME call on dy with the same seed.
namespace {
template <typename T>
__global__ void dropout_kernel(const T* __restrict__ x, long ldx, T* __restrict__ y, long ldy, long rows, int D, uint32_t thr16, float keep_scale,
                               uint32_t seed) {
    const int chunks = D / 8;
    const long total = rows * chunks;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long t = idx / chunks;
        const int c = (int)(idx % chunks) * 8;
        float f[8];
        if constexpr (sizeof(T) == 2) {
            unpack8(*reinterpret_cast<const uint4*>(x + t * ldx + c), f);
        } else {
            const f32x4 a = *reinterpret_cast<const f32x4*>(x + t * ldx + c), b = *reinterpret_cast<const f32x4*>(x + t * ldx + c + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { f[e] = a[e]; f[4 + e] = b[e]; }
        }
        drop8(f, t, c >> 3, seed, thr16, keep_scale);
        if constexpr (sizeof(T) == 2) {
            *reinterpret_cast<uint4*>(y + t * ldy + c) = pack8f(f);
        } else {
            *reinterpret_cast<f32x4*>(y + t * ldy + c) = f32x4{f[0], f[1], f[2], f[3]};
            *reinterpret_cast<f32x4*>(y + t * ldy + c + 4) = f32x4{f[4], f[5], f[6], f[7]};
        }
    }
}
}  // namespace

// dtype: 0 = fp32, 1 = bf16 (x and y alike); D and the row strides multiples of 8; y may alias x
extern "C" int spn_dropout(const void* x, long ldx, void* y, long ldy, int dtype, long rows, int D, float p_drop, unsigned seed, hipStream_t s) {
    SPN_REQUIRE(x && y && rows > 0 && D > 0 && D % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0 && (dtype == 0 || dtype == 1) && p_drop >= 0.f && p_drop < 1.f,
                "spn_dropout: bad arguments (D and row strides multiples of 8, 0 <= p < 1)");
    const uint32_t thr = thr16_of(p_drop);
    const float ks = 1.f / (1.f - (float)thr / 65536.f);
    const int g = grid_for(rows * (D / 8));
    if (dtype == 1) hipLaunchKernelGGL((dropout_kernel<bf16_t>), dim3(g), dim3(256), 0, s, (const bf16_t*)x, ldx, (bf16_t*)y, ldy, rows, D, thr, ks, seed);
    else hipLaunchKernelGGL((dropout_kernel<float>), dim3(g), dim3(256), 0, s, (const float*)x, ldx, (float*)y, ldy, rows, D, thr, ks, seed);
    SPN_LAUNCH_CHECK();
    return SPN_OK;
}

// out[N] (fp32) += column sums of x [T, N]
extern "C" int spn_colsum(const void* x, int x_dtype, long ldx, float* out, long T, int N, hipStream_t s) {
    SPN_REQUIRE(x && out && T > 0 && N > 0, "spn_colsum: bad arguments");
    int slabs = (int)((T + 1023) / 1024);
    // short, wide inputs (the [M/128, 2I] partial rows of spn_gemm_glu_bwd): enough row slabs for >= 1024 blocks, 16+ rows each
    const int want = (1024 + cdiv(N, 64) - 1) / cdiv(N, 64);
    if (slabs < want) slabs = (int)(want < (T + 15) / 16 ? want : (T + 15) / 16);
    if (slabs < 1) slabs = 1;
    if (slabs > 512) slabs = 512;
    const int rpb = (int)((T + slabs - 1) / slabs);
    dim3 grid(cdiv(N, 64), cdiv(T, rpb));
    if (x_dtype == 1 && N % 8 == 0 && ldx % 8 == 0 && (((uintptr_t)x) & 15) == 0) {
        int rpb8 = ((rpb + 7) / 8) * 8;
        hipLaunchKernelGGL(colsum_bf16x8_kernel, dim3(cdiv(N, 256), cdiv(T, rpb8)), dim3(256), 0, s, (const bf16_t*)x, ldx, out, T, N, rpb8);
        SPN_LAUNCH_CHECK();
        return SPN_OK;
    }
    if (x_dtype == 0) hipLaunchKernelGGL((colsum_kernel<float>), grid, dim3(256), 0, s, (const float*)x, ldx, out, T, N, rpb);
    else hipLaunchKernelGGL((colsum_kernel<bf16_t>), grid, dim3(256), 0, s, (const bf16_t*)x, ldx, out, T, N, rpb);
    SPN_LAUNCH_CHECK();
    return SPN_OK;
}

extern "C" int spn_mish_fwd(const float* x, float* y, long n, hipStream_t s) {
    SPN_REQUIRE(x && y && n > 0, "spn_mish_fwd: bad arguments");
    hipLaunchKernelGGL(mish_fwd_kernel, dim3(grid_for(n)), dim3(256), 0, s, x, y, n);
    SPN_LAUNCH_CHECK();
    return SPN_OK;
}
extern "C" int spn_mish_bwd(const float* x, const float* dy, float* dx, long n, hipStream_t s) {
    SPN_REQUIRE(x && dy && dx && n > 0, "spn_mish_bwd: bad arguments");
    hipLaunchKernelGGL(mish_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, s, x, dy, dx, n);
    SPN_LAUNCH_CHECK();
    return SPN_OK;
}

extern "C" int spn_zero_masked_rows(void* x, long ldx, const uint8_t* m, long R, int D, hipStream_t s) {
    SPN_REQUIRE(x && m && R > 0 && D > 0 && D % 8 == 0 && ldx % 8 == 0 && (((uintptr_t)x) & 15) == 0, "spn_zero_masked_rows: bf16 rows of 8-element pieces");
    long g = (R + 255) / 256;      // a wave takes 64 rows per trip, four waves per block
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(zero_masked_rows_kernel, dim3((unsigned)g), dim3(256), 0, s, (bf16_t*)x, ldx, m, R, D);
    SPN_LAUNCH_CHECK();
    return SPN_OK;
}

extern "C" int spn_rows_all_nonzero(const float* x, long ldx, uint8_t* mask, long R, int D, hipStream_t s) {
    SPN_REQUIRE(x && mask && R > 0 && D > 0, "spn_rows_all_nonzero: bad arguments");
    hipLaunchKernelGGL(rows_all_nonzero_kernel, dim3(cdiv(R, 4)), dim3(256), 0, s, x, ldx, mask, R, D);
    SPN_LAUNCH_CHECK();
    return SPN_OK;
}

extern "C" int spn_mask_rows(const float* x, long ldx, const uint8_t* m, float* y, long ldy, long R, int D, int invert, hipStream_t s) {
    SPN_REQUIRE(x && m && y && R > 0 && D > 0, "spn_mask_rows: bad arguments");
    hipLaunchKernelGGL(mask_rows_kernel, dim3(grid_for(R * D)), dim3(256), 0, s, x, ldx, m, y, ldy, R, D, invert);
    SPN_LAUNCH_CHECK();
    return SPN_OK;
}
