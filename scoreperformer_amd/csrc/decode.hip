// Single-note decode kernels (greedy render, b = 1): everything a cached decoder step needs, fp32 end to end.
//
// Replaces the per-note body of `ScorePerformerMixedLMWrapper.unmask_tokens` (models/scoreperformer/wrappers.py:325-407):
// token embedding of the new position only (models/scoreperformer/embeddings.py:227-229,261-262), the transformer
// layers on `x[:, -1:]` with K/V and hidden caches (modules/transformer/transformer.py:159-181,219-221,
// attention.py:155-156), the LM head on the last position for the masked dims, the PAD/MASK bans and the top-1 choice
// (wrappers.py:368-377, modules/sampling.py:28-59).
//
// Design for hipGraph replay: every kernel reads the current position from a DEVICE scalar (`pos`), so one captured step is
// replayed for every note; caches are static [L, .] buffers written in place (the reference grows them with torch.cat, an
// O(L) copy per note); tokens never leave the device.  fp32 weights/activations: at b = 1 the step is bound by launch
// latency and ~100 MB of weight reads, not by math, and fp32 keeps the arg-max identical to the fp32 reference except at
// exact near-ties.
#include "common.h"

namespace {

// y[n] = act( sum_k W[n*ldw + k] * x[k] + bias[n] ) + residual[n];  one wave per output row, 4 rows per block
// x is read from xbase + (pos ? (*pos + x_off) * x_ld : 0);  y is written to ybase + (pos && y_ld ? (*pos + y_off) * y_ld : 0)
__global__ __launch_bounds__(256) void gemv_nk_kernel(const float* __restrict__ W, long ldw, const float* __restrict__ xbase,
                                                      long x_ld, int x_off, const float* __restrict__ bias,
                                                      const float* __restrict__ residual, float* __restrict__ ybase, long y_ld,
                                                      int y_off, const int* __restrict__ pos, int N, int K) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const int p = pos ? *pos : 0;
    const float* x = xbase + (long)(p + x_off) * x_ld;
    float* y = ybase + (long)(p + y_off) * y_ld;
    const float* w = W + (long)n * ldw;
    float acc = 0.f;
    if ((K & 3) == 0 && (ldw & 3) == 0) {
        for (int k = lane * 4; k < K; k += 256) {
            const f32x4 wv = *reinterpret_cast<const f32x4*>(w + k);
            const f32x4 xv = *reinterpret_cast<const f32x4*>(x + k);
            acc += wv[0] * xv[0] + wv[1] * xv[1] + wv[2] * xv[2] + wv[3] * xv[3];
        }
    } else {
        for (int k = lane; k < K; k += 64) acc = fmaf(w[k], x[k], acc);
    }
    acc = wave_sum(acc);
    if (lane == 0) {
        if (bias) acc += bias[n];
        if (residual) acc += residual[n];
        y[n] = acc;
    }
}

// y[n] = sum_k x[k] * W[k*ldw + n] + bias[n]   (x @ W with W stored [K, N]; thread per column, coalesced over n)
__global__ __launch_bounds__(256) void gemv_kn_kernel(const float* __restrict__ W, long ldw, const float* __restrict__ x,
                                                      const float* __restrict__ bias, float* __restrict__ y, int N, int K) {
    __shared__ float xs[2048];
    for (int k = threadIdx.x; k < K; k += 256) xs[k] = x[k];
    __syncthreads();
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    float acc = bias ? bias[n] : 0.f;
    for (int k = 0; k < K; ++k) acc = fmaf(xs[k], W[(long)k * ldw + n], acc);
    y[n] = acc;
}

// gather the K per-key table rows of the token tuple at row (*pos + row_off) of `tokens` [L, tok_ld], concatenate and LayerNorm
struct DecEmbedDesc {
    const float* table[16];
    int width[16], col0[16];
    int nkeys, D;
};
__global__ __launch_bounds__(256) void dec_embed_kernel(DecEmbedDesc d, const long* __restrict__ tokens, long tok_ld, int row_off,
                                                        const int* __restrict__ pos, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float* __restrict__ y, float eps) {
    __shared__ float buf[2048];
    __shared__ float red[8];
    const long* tok = tokens + (long)(*pos + row_off) * tok_ld;
    float s = 0.f;
    for (int c = threadIdx.x; c < d.D; c += 256) {
        int kk = 0;
        for (int q = 1; q < d.nkeys; ++q) if (c >= d.col0[q]) kk = q;
        const float v = d.table[kk][tok[kk] * d.width[kk] + (c - d.col0[kk])];
        buf[c] = v;
        s += v;
    }
    if (!gamma) {
        __syncthreads();
        for (int c = threadIdx.x; c < d.D; c += 256) y[c] = buf[c];
        return;
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    const float mu = (red[0] + red[1] + red[2] + red[3]) / (float)d.D;
    float q2 = 0.f;
    for (int c = threadIdx.x; c < d.D; c += 256) { const float t = buf[c] - mu; q2 += t * t; }
    q2 = wave_sum(q2);
    if ((threadIdx.x & 63) == 0) red[4 + (threadIdx.x >> 6)] = q2;
    __syncthreads();
    const float rs = rsqrtf((red[4] + red[5] + red[6] + red[7]) / (float)d.D + eps);
    for (int c = threadIdx.x; c < d.D; c += 256) y[c] = (buf[c] - mu) * rs * gamma[c] + beta[c];
}

// dst[(*pos + dst_off) * dst_ld + c] = src[(*pos + src_off) * src_ld + c]   (ld = 0: fixed row)
__global__ void dec_copy_row_kernel(const float* __restrict__ src, long src_ld, int src_off, float* __restrict__ dst, long dst_ld,
                                    int dst_off, const int* __restrict__ pos, int D) {
    const int p = *pos;
    const float* s = src + (long)(p + src_off) * src_ld;
    float* t = dst + (long)(p + dst_off) * dst_ld;
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < D; c += gridDim.x * blockDim.x) t[c] = s[c];
}

// out[i] = a[i] * act(gate[i]),  u = (a | gate) of width 2I
__global__ void dec_glu_kernel(const float* __restrict__ u, float* __restrict__ out, int I, int act, int glu) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < I; i += gridDim.x * blockDim.x) {
        const float g = glu ? u[I + i] : u[i];
        const float a = act == 0 ? g / (1.f + __expf(-g)) : 0.5f * g * (1.f + erff(g * 0.70710678118654752f));
        out[i] = glu ? u[i] * a : a;
    }
}

// single-query attention for one new position t = *pos.  qkv = (q[h*64] | k[kvh*64] | v[kvh*64]) of that position.
// Appends k, v to the caches [L, kvh*64] at row t, then o[h] = softmax(q_h . K^T * scale - slope_h * (t - j)) V over j <= t.
// grid = h blocks of 256 threads: wave w handles keys j = w, w+4, ...; lane = one of 64 dims for the dot products.
__global__ __launch_bounds__(256) void dec_attn_kernel(const float* __restrict__ qkv, float* __restrict__ kcache, float* __restrict__ vcache,
                                                       const float* __restrict__ slopes, const int* __restrict__ pos, float* __restrict__ o,
                                                       int h, int kvh, float scale) {
    __shared__ float m_s[4], l_s[4], o_s[4][64];
    const int hi = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int kh = kvh == 1 ? 0 : hi;
    const int t = *pos;
    const long cw = (long)kvh * 64;
    const float* knew = qkv + h * 64 + kh * 64;
    const float* vnew = qkv + h * 64 + kvh * 64 + kh * 64;
    // append (every block that shares the kv head writes the same values: benign)
    if (w == 0) { kcache[t * cw + kh * 64 + lane] = knew[lane]; vcache[t * cw + kh * 64 + lane] = vnew[lane]; }
    const float qd = qkv[hi * 64 + lane] * scale;
    const float slope = slopes ? slopes[hi] : 0.f;
    float m = -INFINITY, l = 0.f, acc = 0.f;   // acc: this lane's output dim, over this wave's keys
    for (int j = w; j <= t; j += 4) {
        const float kd = (j == t) ? knew[lane] : kcache[j * cw + kh * 64 + lane];
        const float vd = (j == t) ? vnew[lane] : vcache[j * cw + kh * 64 + lane];
        const float s = wave_sum(qd * kd) - slope * (float)(t - j);
        const float m_new = fmaxf(m, s);
        const float alpha = __expf(m - m_new), p = __expf(s - m_new);
        l = l * alpha + p;
        acc = acc * alpha + p * vd;
        m = m_new;
    }
    if (lane == 0) { m_s[w] = m; l_s[w] = l; }
    o_s[w][lane] = acc;
    __syncthreads();
    if (w == 0) {
        const float mm = fmaxf(fmaxf(m_s[0], m_s[1]), fmaxf(m_s[2], m_s[3]));
        float num = 0.f, den = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float f = (m_s[q] == -INFINITY) ? 0.f : __expf(m_s[q] - mm);
            num += o_s[q][lane] * f;
            den += l_s[q] * f;
        }
        o[hi * 64 + lane] = num / den;
    }
}

// arg-max over logits[0..V) with banned ids -> if tokens[(*pos + 1), dim] == mask_id: write it there.  One block.
__global__ __launch_bounds__(256) void dec_argmax_write_kernel(const float* __restrict__ logits, int V, unsigned ban_mask,
                                                               long* __restrict__ tokens, long tok_ld, int dim, int mask_id,
                                                               const int* __restrict__ pos) {
    __shared__ float bv[4];
    __shared__ int bi[4];
    float best = -INFINITY;
    int idx = 0x7fffffff;
    for (int c = threadIdx.x; c < V; c += 256) {
        float v = logits[c];
        if (c < 32 && ((ban_mask >> c) & 1u)) v = -INFINITY;
        if (v > best || (v == best && c < idx)) { best = v; idx = c; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(idx, o, 64);
        if (ov > best || (ov == best && oi < idx)) { best = ov; idx = oi; }
    }
    if ((threadIdx.x & 63) == 0) { bv[threadIdx.x >> 6] = best; bi[threadIdx.x >> 6] = idx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int q = 1; q < 4; ++q) if (bv[q] > best || (bv[q] == best && bi[q] < idx)) { best = bv[q]; idx = bi[q]; }
        long* cell = tokens + (long)(*pos + 1) * tok_ld + dim;
        if (*cell == mask_id) *cell = idx;
    }
}

__global__ void dec_add_pos_kernel(int* pos, int delta) { if (threadIdx.x == 0 && blockIdx.x == 0) *pos += delta; }

}  // namespace

extern "C" int spn_dec_gemv(const float* W, long ldw, const float* x, long x_ld, int x_off, const float* bias, const float* residual,
                            float* y, long y_ld, int y_off, const int* pos, int N, int K, int kn_layout, hipStream_t s) {
    SPN_REQUIRE(W && x && y && N > 0 && K > 0, "spn_dec_gemv: bad arguments");
    if (kn_layout) {
        SPN_REQUIRE(K <= 2048 && !residual && !pos, "spn_dec_gemv: kn layout supports K <= 2048, no residual/pos addressing");
        hipLaunchKernelGGL(gemv_kn_kernel, dim3(cdiv(N, 256)), dim3(256), 0, s, W, ldw, x, bias, y, N, K);
    } else {
        hipLaunchKernelGGL(gemv_nk_kernel, dim3(cdiv(N, 4)), dim3(256), 0, s, W, ldw, x, x_ld, x_off, bias, residual, y, y_ld, y_off, pos, N, K);
    }
    SPN_LAUNCH_CHECK();
    return SPN_OK;
}

extern "C" int spn_dec_embed(int nkeys, const float* const* tables, const int* E, const long* tokens, long tok_ld, int row_off,
                             const int* pos, const float* gamma, const float* beta, float* y, float eps, hipStream_t s) {
    SPN_REQUIRE(nkeys > 0 && nkeys <= 16 && tokens && pos && y, "spn_dec_embed: bad arguments");
    DecEmbedDesc d;
    memset(&d, 0, sizeof(d));
    int col = 0;
    for (int i = 0; i < nkeys; ++i) { d.table[i] = tables[i]; d.width[i] = E[i]; d.col0[i] = col; col += E[i]; }
    d.nkeys = nkeys; d.D = col;
    SPN_REQUIRE(col <= 2048, "spn_dec_embed: total width <= 2048");
    hipLaunchKernelGGL(dec_embed_kernel, dim3(1), dim3(256), 0, s, d, tokens, tok_ld, row_off, pos, gamma, beta, y, eps);
    SPN_LAUNCH_CHECK();
    return SPN_OK;
}

extern "C" int spn_dec_copy_row(const float* src, long src_ld, int src_off, float* dst, long dst_ld, int dst_off, const int* pos, int D,
                                hipStream_t s) {
    SPN_REQUIRE(src && dst && pos && D > 0, "spn_dec_copy_row: bad arguments");
    hipLaunchKernelGGL(dec_copy_row_kernel, dim3(cdiv(D, 256)), dim3(256), 0, s, src, src_ld, src_off, dst, dst_ld, dst_off, pos, D);
    SPN_LAUNCH_CHECK();
    return SPN_OK;
}

extern "C" int spn_dec_glu(const float* u, float* out, int I, int act, int glu, hipStream_t s) {
    SPN_REQUIRE(u && out && I > 0, "spn_dec_glu: bad arguments");
    hipLaunchKernelGGL(dec_glu_kernel, dim3(cdiv(I, 256)), dim3(256), 0, s, u, out, I, act, glu);
    SPN_LAUNCH_CHECK();
    return SPN_OK;
}

extern "C" int spn_dec_attn(const float* qkv, float* kcache, float* vcache, const float* slopes, const int* pos, float* o, int h, int kvh,
                            float scale, hipStream_t s) {
    SPN_REQUIRE(qkv && kcache && vcache && pos && o && h > 0 && (kvh == 1 || kvh == h), "spn_dec_attn: bad arguments");
    hipLaunchKernelGGL(dec_attn_kernel, dim3(h), dim3(256), 0, s, qkv, kcache, vcache, slopes, pos, o, h, kvh, scale);
    SPN_LAUNCH_CHECK();
    return SPN_OK;
}

extern "C" int spn_dec_argmax_write(const float* logits, int V, unsigned ban_mask, long* tokens, long tok_ld, int dim, int mask_id,
                                    const int* pos, hipStream_t s) {
    SPN_REQUIRE(logits && tokens && pos && V > 0, "spn_dec_argmax_write: bad arguments");
    hipLaunchKernelGGL(dec_argmax_write_kernel, dim3(1), dim3(256), 0, s, logits, V, ban_mask, tokens, tok_ld, dim, mask_id, pos);
    SPN_LAUNCH_CHECK();
    return SPN_OK;
}

extern "C" int spn_dec_add_pos(int* pos, int delta, hipStream_t s) {
    SPN_REQUIRE(pos, "spn_dec_add_pos: bad arguments");
    hipLaunchKernelGGL(dec_add_pos_kernel, dim3(1), dim3(64), 0, s, pos, delta);
    SPN_LAUNCH_CHECK();
    return SPN_OK;
}
