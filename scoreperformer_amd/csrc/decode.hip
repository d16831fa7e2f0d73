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
#include "decode_attn.h"
#include "decode_sample.h"

// No floating-point contraction in this file: whether `a * b + c` becomes one fused operation or two is otherwise the optimiser's choice per
// call site (packed multiplies + adds in one loop, fused multiply-adds in its twin), and the persistent layer launch
// (decode_layer.hip, same pragma) must reproduce these kernels' residual stream BIT FOR BIT.  The step is latency-bound: no cost.
#pragma clang fp contract(off)

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
        f32x2 a2 = f32x2{0.f, 0.f};
        for (int k = lane * 4; k < K; k += 256) {
            const f32x4 wv = *reinterpret_cast<const f32x4*>(w + k);
            const f32x4 xv = *reinterpret_cast<const f32x4*>(x + k);
            dec_dot4(a2, wv, xv);
        }
        acc = dec_fold(a2);
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
    float mu, rs;
    if ((d.D & 3) == 0) {   // per wave, common.h dec_ln_stats: as dec_embed_proj_kernel
        __syncthreads();
        const int lane = threadIdx.x & 63;
        f32x4 xv[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int k = lane * 4 + c * 256;
            xv[c] = k < d.D ? *reinterpret_cast<const f32x4*>(buf + k) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        dec_ln_stats<8>(xv, d.D, eps, lane, mu, rs);
    } else {
        s = wave_sum(s);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
        __syncthreads();
        mu = (red[0] + red[1] + red[2] + red[3]) / (float)d.D;
        float q2 = 0.f;
        for (int c = threadIdx.x; c < d.D; c += 256) { const float t = buf[c] - mu; q2 += t * t; }
        q2 = wave_sum(q2);
        if ((threadIdx.x & 63) == 0) red[4 + (threadIdx.x >> 6)] = q2;
        __syncthreads();
        rs = rsqrtf((red[4] + red[5] + red[6] + red[7]) / (float)d.D + eps);
    }
    for (int c = threadIdx.x; c < d.D; c += 256) y[c] = (buf[c] - mu) * rs * gamma[c] + beta[c];
}

// Token-tuple embedding AND its projection for BOTH sequences of a multi-sequence decoder in one launch (blockIdx.y = sequence:
// 0 = the tokens at row *pos, 1 = the masked copy at row *pos + 1):  y[seq * N + n] = W[n, :] . LN(concat_k table_k[token_k]) + bias[n].
// Every block rebuilds the (<= 2048-wide) embedding in LDS -- the same gather, statistics and normalisation as dec_embed_kernel -- and
// then runs the row products of gemv_nk_kernel, so the values equal dec_embed + gemv bit for bit (4 launches of a decode step become 1).
// A second, independent GEMV that rides in the same launch (blocks behind the embedding blocks): y[n] = W[n, :] . x[(p + x_off) * x_ld ..] +
// bias[n] -- the stacked AdaLN condition projections of a step, which depend on the position only.  W = null: no rider.
struct DecRider { const float* W; long ldw; int N, K; const float* x; long x_ld; int x_off; const float* bias; float* y; };
// latch: the FIRST launch of a step reads the position from `pos` and (block 0) republishes it in `pos_latch`, which every later
// launch of the step reads; the LAST launch (dec_head_kernel) writes position + 1 back to `pos`.  Nobody reads a scalar in the launch
// that writes it, so the one-thread "advance" launch between two notes is gone.  pos_latch = null: `pos` is read-only here.
__global__ __launch_bounds__(256) void dec_embed_proj_kernel(DecEmbedDesc d, const long* __restrict__ tok_a, const long* __restrict__ tok_b,
                                                             long tok_ld, const int* __restrict__ pos, int* __restrict__ pos_latch,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float eps, const float* __restrict__ W, long ldw,
                                                             const float* __restrict__ bias, float* __restrict__ y, int N, int nbe, DecRider r) {
    __shared__ __attribute__((aligned(16))) float buf[2048];
    __shared__ float red[8];
    if ((int)blockIdx.x >= 2 * nbe) {   // rider GEMV: one wave per output row, the arithmetic of gemv_nk_kernel
        const int lane = threadIdx.x & 63;
        const int n = ((int)blockIdx.x - 2 * nbe) * 4 + (threadIdx.x >> 6);
        if (n >= r.N) return;
        const float* w = r.W + (long)n * r.ldw;
        const float bias_r = r.bias ? r.bias[n] : 0.f;
        const bool vec = (r.K & 3) == 0 && (r.ldw & 3) == 0;
        f32x4 w0 = f32x4{0.f, 0.f, 0.f, 0.f};
        if (vec && lane * 4 < r.K) w0 = *reinterpret_cast<const f32x4*>(w + lane * 4);   // first chunk before the position is known
        const float* x = r.x + (long)(*pos + r.x_off) * r.x_ld;
        float acc = 0.f;
        if (vec) {
            f32x2 a2 = f32x2{0.f, 0.f};
            for (int k = lane * 4; k < r.K; k += 256) {
                const f32x4 wv = k == lane * 4 ? w0 : *reinterpret_cast<const f32x4*>(w + k);
                const f32x4 xv = *reinterpret_cast<const f32x4*>(x + k);
                dec_dot4(a2, wv, xv);
            }
            acc = dec_fold(a2);
        } else {
            for (int k = lane; k < r.K; k += 64) acc = fmaf(w[k], x[k], acc);
        }
        acc = wave_sum(acc);
        if (lane == 0) r.y[n] = r.bias ? acc + bias_r : acc;
        return;
    }
    const int seq = (int)blockIdx.x / nbe, bx = (int)blockIdx.x % nbe;
    if (pos_latch && blockIdx.x == 0 && threadIdx.x == 0) *pos_latch = *pos;
    // this wave's weight row and bias do not depend on the tokens: requested first, they arrive under the position -> token -> table row
    // -> LayerNorm chain below (three dependent trips to memory) instead of behind it
    const int lane_ = threadIdx.x & 63, n_ = bx * 4 + (threadIdx.x >> 6);
    const bool vec_ = (d.D & 3) == 0 && (ldw & 3) == 0;
    const float* wrow = W + (long)min(n_, N - 1) * ldw;
    f32x4 wreg[8];
    if (vec_) {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int k = lane_ * 4 + c * 256;
            wreg[c] = k < d.D ? *reinterpret_cast<const f32x4*>(wrow + k) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    const float bias_v = bias ? bias[min(n_, N - 1)] : 0.f;
    const long* tok = (seq ? tok_b : tok_a) + (long)(*pos + seq) * tok_ld;
    float s = 0.f;
    for (int c = threadIdx.x; c < d.D; c += 256) {
        int kk = 0;
        for (int q = 1; q < d.nkeys; ++q) if (c >= d.col0[q]) kk = q;
        const float v = d.table[kk][tok[kk] * d.width[kk] + (c - d.col0[kk])];
        buf[c] = v;
        s += v;
    }
    if (gamma) {
        float mu, rs;
        if ((d.D & 3) == 0) {
            // statistics per wave over the dot-product layout (common.h dec_ln_stats: what the embed phase of decode_layer.hip computes)
            __syncthreads();
            f32x4 xv[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int k = lane_ * 4 + c * 256;
                xv[c] = k < d.D ? *reinterpret_cast<const f32x4*>(buf + k) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
            dec_ln_stats<8>(xv, d.D, eps, lane_, mu, rs);
            __syncthreads();   // every wave has read buf before the loop below overwrites it
        } else {
            s = wave_sum(s);
            if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
            __syncthreads();
            mu = (red[0] + red[1] + red[2] + red[3]) / (float)d.D;
            float q2 = 0.f;
            for (int c = threadIdx.x; c < d.D; c += 256) { const float t = buf[c] - mu; q2 += t * t; }
            q2 = wave_sum(q2);
            if ((threadIdx.x & 63) == 0) red[4 + (threadIdx.x >> 6)] = q2;
            __syncthreads();
            rs = rsqrtf((red[4] + red[5] + red[6] + red[7]) / (float)d.D + eps);
        }
        for (int c = threadIdx.x; c < d.D; c += 256) buf[c] = (buf[c] - mu) * rs * gamma[c] + beta[c];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int n = bx * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const float* w = W + (long)n * ldw;
    const int K = d.D;
    float acc = 0.f;
    if (vec_) {   // chunks ascending, common.h dec_dot4
        f32x2 a2 = f32x2{0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int k = lane * 4 + c * 256;
            if (k < K) {
                const f32x4 wv = wreg[c];
                const f32x4 xv = *reinterpret_cast<const f32x4*>(buf + k);
                dec_dot4(a2, wv, xv);
            }
        }
        acc = dec_fold(a2);
    } else {
        for (int k = lane; k < K; k += 64) acc = fmaf(w[k], buf[k], acc);
    }
    acc = wave_sum(acc);
    if (lane == 0) {
        if (bias) acc += bias_v;
        y[seq * N + n] = acc;
    }
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

// ---- batched (prefill) variants: rows t0 .. t0+n-1 of a window whose tokens are all known, one launch per operator instead of one
//      per position.  Same arithmetic as the single-position kernels above; K/V rows are already in the caches.
// grid = (h, n): block (hi, r) is query t = t0 + r over keys j <= t, in the batched form of the decode step (decode_attn.h: 16 lanes per
// key, DEC_NU keys per lane group scored together, 256 keys per workgroup and batch, the two-level group merge) -- round 5: the key-by-key
// online softmax it replaces (one 64-lane reduction and a rescale per key) took 81 us per launch for a 500-row window, 163 ms per 12
// renders of tools/bench_render.py.
__global__ __launch_bounds__(512) void dec_attn_rows_kernel(const float* __restrict__ q, long q_ld, const float* __restrict__ kcache,
                                                            const float* __restrict__ vcache, const float* __restrict__ slopes, int t0,
                                                            float* __restrict__ o, long o_ld, int h, int kvh, float scale) {
    __shared__ float sm[DEC_G], sl[DEC_G];
    __shared__ __attribute__((aligned(16))) float so[DEC_G][64];
    __shared__ float sm2[8], sl2[8];
    __shared__ float so2[8][64];
    const int hi = blockIdx.x, r = blockIdx.y, tid = threadIdx.x, lane = tid & 63, w = tid >> 6, grp = lane >> 4, l16 = lane & 15;
    const int kh = kvh == 1 ? 0 : hi;
    const int t = t0 + r;
    const long cw = (long)kvh * 64;
    const float slope = slopes ? slopes[hi] : 0.f;
    const f32x4 q4 = *reinterpret_cast<const f32x4*>(q + (long)r * q_ld + hi * 64 + l16 * 4) * scale;
    // row t is in the cache like every other: the "new key" operands of the batch are that row itself
    const f32x4 kt4 = *reinterpret_cast<const f32x4*>(kcache + (long)t * cw + kh * 64 + l16 * 4);
    const f32x4 vt4 = *reinterpret_cast<const f32x4*>(vcache + (long)t * cw + kh * 64 + l16 * 4);
    float m = -INFINITY, l = 0.f;
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    const int j1 = t + 1;
    for (int jb0 = w * 4 + grp; jb0 < j1; jb0 += 256) {
        f32x4 k4[DEC_NU], v4[DEC_NU];
#pragma unroll
        for (int u = 0; u < DEC_NU; ++u) {
            const int j = min(jb0 + DEC_G * u, t);
            k4[u] = *reinterpret_cast<const f32x4*>(kcache + (long)j * cw + kh * 64 + l16 * 4);
            v4[u] = *reinterpret_cast<const f32x4*>(vcache + (long)j * cw + kh * 64 + l16 * 4);
        }
        dec_attn_batch<DEC_NU>(k4, v4, kt4, vt4, q4, slope, t, jb0, j1, m, l, acc);
    }
    const int gi = w * 4 + grp;
    if (l16 == 0) { sm[gi] = m; sl[gi] = l; }
    *reinterpret_cast<f32x4*>(&so[gi][l16 * 4]) = acc;
    dec_attn_merge_wave(sm, sl, so, sm2, sl2, so2, w, lane);
    __syncthreads();
    if (tid < 64) {
        float mm, num, den;
        dec_attn_merge_block(sm2, sl2, so2, tid, lane, mm, num, den);
        o[(long)r * o_ld + hi * 64 + tid] = num / den;
    }
}

// out[r, i] = u[r, i] * act(u[r, I + i])  (glu) or act(u[r, i]);  grid.y = rows
__global__ void dec_glu_rows_kernel(const float* __restrict__ u, long u_ld, float* __restrict__ out, long out_ld, int I, int act, int glu) {
    const float* ur = u + (long)blockIdx.y * u_ld;
    float* orow = out + (long)blockIdx.y * out_ld;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < I; i += gridDim.x * blockDim.x) {
        const float g = glu ? ur[I + i] : ur[i];
        const float a = act == 0 ? g / (1.f + __expf(-g)) : 0.5f * g * (1.f + erff(g * 0.70710678118654752f));
        orow[i] = glu ? ur[i] * a : a;
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

// ---- fused step kernels (fewer, fatter launches: a decode step is launch-latency bound) -----------------------------------
// y[n] = [GLU]( W[n,:] . LN?(x) + bias[n] ) + residual[n], optionally mirrored into a second (cache) row.
//   norm: 0 none, 1 affine LayerNorm(gamma, beta), 2 adaptive (gamma | beta = gb[0:K] | gb[K:2K])   -- done in the prologue by
//         every block on its own LDS copy of x (K <= 2048 floats: cheaper than a separate launch + round trip)
//   glu : W has 2N rows (values | gates), y[n] = v_n * act(g_n)
//   xn_out: block 0 also stores the normalised x (the reference caches the final-norm output)
struct DecGemvArgs {
    const float* W; long ldw; int N, K;
    const float* x; long x_ld; int x_off;
    int norm; const float* gamma; const float* beta; float eps;
    const float* bias; const float* residual;
    float* y; long y_ld; int y_off;
    float* y2; long y2_ld; int y2_off;
    float* xn_out; long xn_ld; int xn_off;
    int glu, act;
    const int* pos;
    // x = the attention output merged from the split-key partials of dec_attn2 / dec_xattn (launched with merge = 0): [h][S] records of
    // (running max, normaliser, 64 weighted value sums).  Every block redoes the 8 KiB merge in its prologue; the kernel boundary in
    // front of this launch orders it behind the partials, so the attention kernel needs neither fences nor a last-block tail.
    const float* att_part; int att_h, att_S;
    // x = ( LN?(x[0 : cat_d]) | ctx[(p + 1) * ctx_ld ..][0 : ctx_w] | style[(p + 1) * style_ld ..][0 : style_w] ), K = the sum: the decoder's
    // concatenated input of one position (transformer.py:160-176), built in the prologue instead of by dec_cat_kernel.  cat_d = 0: off.
    int cat_d; const float* cat_ctx; long cat_ctx_ld; int cat_ctx_w; const float* cat_style; long cat_style_ld; int cat_style_w;
};
__device__ __forceinline__ float dec_act(float g, int act) {
    return act == 0 ? g / (1.f + __expf(-g)) : 0.5f * g * (1.f + erff(g * 0.70710678118654752f));
}
__global__ __launch_bounds__(256) void dec_fused_gemv_kernel(DecGemvArgs a) {
    __shared__ __attribute__((aligned(16))) float xs[2048];
    __shared__ float red[8];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    // This wave's weight row (and the gate row of a gated projection) does not depend on the input vector: its loads go out FIRST, so
    // that their HBM latency runs under the input load + LayerNorm prologue below instead of behind it (a note is a chain of ~30 such
    // launches and each one is latency-, not bandwidth-bound).  K <= 2048: at most 8 chunks of 4 floats per lane.
    const int n = blockIdx.x * 4 + w;
    const bool vec = (a.K & 3) == 0 && (a.ldw & 3) == 0;
    const bool gated = a.glu > 0;   // glu = -1: plain activation, no gate rows
    const float* wv_ = a.W + (long)min(n, a.N - 1) * a.ldw;
    const float* wg_ = a.W + (long)(min(n, a.N - 1) + a.N) * a.ldw;
    f32x4 wreg[8], greg[8];
    if (vec) {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int k = lane * 4 + c * 256;
            wreg[c] = k < a.K ? *reinterpret_cast<const f32x4*>(wv_ + k) : f32x4{0.f, 0.f, 0.f, 0.f};
            greg[c] = (gated && k < a.K) ? *reinterpret_cast<const f32x4*>(wg_ + k) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    // ... and so do the bias / residual of this wave's output and the position scalar.  A note is a chain of ~37 launches, each of
    // them a chain of DEPENDENT global loads (a miss costs 1-2 us here: the previous launch's output is fresh in memory, not in this
    // XCD's L2): everything that can be requested up front is, and the input vector is only addressed through the position when it
    // really is a row of a [positions, K] buffer (x_ld != 0).
    const int nn = min(n, a.N - 1);
    const float bias_v = a.bias ? a.bias[nn] : 0.f, bias_g = (a.bias && gated) ? a.bias[nn + a.N] : 0.f;
    const float res_v = a.residual ? a.residual[nn] : 0.f;
    const int p = a.pos ? *a.pos : 0;
    float s = 0.f;
    if (a.att_part) {
        // K = h * 64.  Same expression order as the in-kernel merge of dec_attn2_kernel (bit-identical o), but the loads of a 16-record
        // chunk are issued together: a dependent load per record made this prologue 15 us long
        for (int k = threadIdx.x; k < a.K; k += 256) {
            const float* ph = a.att_part + (long)(k >> 6) * a.att_S * 66;
            const int dcol = 2 + (k & 63);
            float mm = -INFINITY;
            float num = 0.f, den = 0.f;
            if (a.att_S <= 16) {   // the usual split: all three fields of all records requested at once (one memory latency, not two)
                float mv[16], lv[16], nv[16];
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const bool in = q < a.att_S;
                    mv[q] = in ? ph[q * 66] : -INFINITY;
                    lv[q] = in ? ph[q * 66 + 1] : 0.f;
                    nv[q] = in ? ph[q * 66 + dcol] : 0.f;
                }
#pragma unroll
                for (int q = 0; q < 16; ++q) mm = fmaxf(mm, mv[q]);
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    if (q < a.att_S) {
                        const float f = (mv[q] == -INFINITY) ? 0.f : __expf(mv[q] - mm);
                        num += nv[q] * f; den += lv[q] * f;
                    }
                }
                xs[k] = num / den;
                continue;
            }
            for (int q0 = 0; q0 < a.att_S; q0 += 16) {
                float mv[16];
#pragma unroll
                for (int q = 0; q < 16; ++q) mv[q] = (q0 + q < a.att_S) ? ph[(q0 + q) * 66] : -INFINITY;
#pragma unroll
                for (int q = 0; q < 16; ++q) mm = fmaxf(mm, mv[q]);
            }
            for (int q0 = 0; q0 < a.att_S; q0 += 16) {
                float mv[16], lv[16], nv[16];
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const bool in = q0 + q < a.att_S;
                    mv[q] = in ? ph[(q0 + q) * 66] : -INFINITY;
                    lv[q] = in ? ph[(q0 + q) * 66 + 1] : 0.f;
                    nv[q] = in ? ph[(q0 + q) * 66 + dcol] : 0.f;
                }
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    if (q0 + q < a.att_S) {
                        const float f = (mv[q] == -INFINITY) ? 0.f : __expf(mv[q] - mm);
                        num += nv[q] * f; den += lv[q] * f;
                    }
                }
            }
            xs[k] = num / den;
        }
    } else if (a.cat_d > 0) {
        // same arithmetic as dec_cat_kernel: LayerNorm over the first cat_d entries only, the context / style rows appended as they are
        for (int k = threadIdx.x; k < a.cat_d; k += 256) { const float v = a.x[k]; xs[k] = v; s += v; }
        if (a.cat_ctx) for (int k = threadIdx.x; k < a.cat_ctx_w; k += 256) xs[a.cat_d + k] = a.cat_ctx[(long)(p + 1) * a.cat_ctx_ld + k];
        if (a.cat_style)
            for (int k = threadIdx.x; k < a.cat_style_w; k += 256)
                xs[a.cat_d + (a.cat_ctx ? a.cat_ctx_w : 0) + k] = a.cat_style[(long)(p + 1) * a.cat_style_ld + k];
        if (a.gamma) {
            float mu, rs;
            if ((a.cat_d & 3) == 0) {   // per wave, common.h dec_ln_stats (the front phase of decode_layer.hip)
                __syncthreads();
                f32x4 xv[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const int k = lane * 4 + c * 256;
                    xv[c] = k < a.cat_d ? *reinterpret_cast<const f32x4*>(xs + k) : f32x4{0.f, 0.f, 0.f, 0.f};
                }
                dec_ln_stats<8>(xv, a.cat_d, a.eps, lane, mu, rs);
                __syncthreads();
            } else {
                s = wave_sum(s);
                if (lane == 0) red[w] = s;
                __syncthreads();
                mu = (red[0] + red[1] + red[2] + red[3]) / (float)a.cat_d;
                float q2 = 0.f;
                for (int k = threadIdx.x; k < a.cat_d; k += 256) { const float t = xs[k] - mu; q2 += t * t; }
                q2 = wave_sum(q2);
                if (lane == 0) red[4 + w] = q2;
                __syncthreads();
                rs = rsqrtf((red[4] + red[5] + red[6] + red[7]) / (float)a.cat_d + a.eps);
            }
            for (int k = threadIdx.x; k < a.cat_d; k += 256) xs[k] = (xs[k] - mu) * rs * a.gamma[k] + a.beta[k];
        }
    } else {
        const float* x = a.x_ld ? a.x + (long)(p + a.x_off) * a.x_ld : a.x;
        for (int k = threadIdx.x; k < a.K; k += 256) { const float v = x[k]; xs[k] = v; s += v; }
    }
    if (a.norm) {
        float mu, rs;
        if ((a.K & 3) == 0) {
            // statistics per wave over the dot-product layout (common.h dec_ln_stats: the arithmetic of decode_layer.hip's wave_norm)
            __syncthreads();
            f32x4 xv[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int k = lane * 4 + c * 256;
                xv[c] = k < a.K ? *reinterpret_cast<const f32x4*>(xs + k) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
            dec_ln_stats<8>(xv, a.K, a.eps, lane, mu, rs);
            __syncthreads();   // every wave has read xs before the loop below overwrites it
        } else {
            s = wave_sum(s);
            if (lane == 0) red[w] = s;
            __syncthreads();
            mu = (red[0] + red[1] + red[2] + red[3]) / (float)a.K;
            float q2 = 0.f;
            for (int k = threadIdx.x; k < a.K; k += 256) { const float t = xs[k] - mu; q2 += t * t; }
            q2 = wave_sum(q2);
            if (lane == 0) red[4 + w] = q2;
            __syncthreads();
            rs = rsqrtf((red[4] + red[5] + red[6] + red[7]) / (float)a.K + a.eps);
        }
        const float* gam = a.gamma;
        const float* bet = a.norm == 2 ? a.gamma + a.K : a.beta;
        for (int k = threadIdx.x; k < a.K; k += 256) {
            float v = (xs[k] - mu) * rs;
            if (gam) v = v * gam[k] + bet[k];
            xs[k] = v;
            if (a.xn_out && blockIdx.x == 0) a.xn_out[(long)(p + a.xn_off) * a.xn_ld + k] = v;
        }
    }
    __syncthreads();
    if (n >= a.N) return;
    float acc = 0.f, accg = 0.f;
    if (vec) {   // chunks ascending, common.h dec_dot4 (the arithmetic of decode_layer.hip's dot_rows)
        f32x2 a2 = f32x2{0.f, 0.f}, g2 = f32x2{0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int k = lane * 4 + c * 256;
            if (k < a.K) {
                const f32x4 xv = *reinterpret_cast<const f32x4*>(xs + k);
                dec_dot4(a2, wreg[c], xv);
                if (gated) dec_dot4(g2, greg[c], xv);
            }
        }
        acc = dec_fold(a2); accg = dec_fold(g2);
    } else {
        for (int k = lane; k < a.K; k += 64) { acc = fmaf(wv_[k], xs[k], acc); if (gated) accg = fmaf(wg_[k], xs[k], accg); }
    }
    acc = wave_sum(acc);
    if (gated) accg = wave_sum(accg);
    if (lane == 0) {
        if (a.bias) { acc += bias_v; if (gated) accg += bias_g; }
        if (gated) acc = acc * dec_act(accg, a.act);
        else if (a.act >= 0 && a.glu < 0) acc = dec_act(acc, a.act);   // plain activation (glu = -1)
        if (a.residual) acc += res_v;
        a.y[(long)(p + a.y_off) * a.y_ld + n] = acc;
        if (a.y2) a.y2[(long)(p + a.y2_off) * a.y2_ld + n] = acc;
    }
}

// out = ( LN?(x) | ctx[pos + 1] | style[pos + 1] )   (transformer.py:160-176: the decoder's concatenated input of one position)
__global__ __launch_bounds__(256) void dec_cat_kernel(const float* __restrict__ x, int d, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, float eps, const float* __restrict__ ctx, long ctx_ld,
                                                      int ctx_w, const float* __restrict__ style, long style_ld, int style_w,
                                                      const int* __restrict__ pos, float* __restrict__ out) {
    __shared__ float red[8];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, p = *pos;
    float s = 0.f;
    for (int k = threadIdx.x; k < d; k += 256) s += x[k];
    float mu = 0.f, rs = 1.f;
    if (gamma && (d & 3) == 0 && d <= 2048 && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {   // per wave, common.h dec_ln_stats: as the fused prologue
        f32x4 xv[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int k = lane * 4 + c * 256;
            xv[c] = k < d ? *reinterpret_cast<const f32x4*>(x + k) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        dec_ln_stats<8>(xv, d, eps, lane, mu, rs);
    } else if (gamma) {
        s = wave_sum(s);
        if (lane == 0) red[w] = s;
        __syncthreads();
        mu = (red[0] + red[1] + red[2] + red[3]) / (float)d;
        float q2 = 0.f;
        for (int k = threadIdx.x; k < d; k += 256) { const float t = x[k] - mu; q2 += t * t; }
        q2 = wave_sum(q2);
        if (lane == 0) red[4 + w] = q2;
        __syncthreads();
        rs = rsqrtf((red[4] + red[5] + red[6] + red[7]) / (float)d + eps);
    }
    for (int k = threadIdx.x; k < d; k += 256) out[k] = gamma ? (x[k] - mu) * rs * gamma[k] + beta[k] : x[k];
    if (ctx) for (int k = threadIdx.x; k < ctx_w; k += 256) out[d + k] = ctx[(long)(p + 1) * ctx_ld + k];
    if (style) for (int k = threadIdx.x; k < style_w; k += 256) out[d + (ctx ? ctx_w : 0) + k] = style[(long)(p + 1) * style_ld + k];
}

// Single-query attention, keys split over blocks (grid = h x S), 16 lanes per key (4 dims each: one 16-byte load), online softmax per
// lane group, partials merged by the last block of each head.  ALiBi reach: with B = scale*|q|*max|k| every weight further than
// D = (104 + 2B)/slope from the query is below exp(-104) of the largest one -- exactly 0 or one denormal ulp in the fp32 reference
// as well -- so those keys are not read.  kmax2[0] = running max |k|^2 of this layer's cache (MQA: one kv head; else per kv head).
__global__ __launch_bounds__(512) void dec_attn2_kernel(const float* __restrict__ qkv, float* __restrict__ kcache, float* __restrict__ vcache,
                                                        const float* __restrict__ slopes, const int* __restrict__ pos, float* __restrict__ o,
                                                        float* __restrict__ part, int* __restrict__ counter, float* __restrict__ kmax2,
                                                        int h, int kvh, float scale, int merge) {
    __shared__ float sm[DEC_G], sl[DEC_G];
    __shared__ __attribute__((aligned(16))) float so[DEC_G][64];
    __shared__ float sm2[8], sl2[8];
    __shared__ float so2[8][64];
    __shared__ int is_last;
    const int hi = blockIdx.x, sp = blockIdx.y, S = gridDim.y;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, grp = lane >> 4, l16 = lane & 15;
    const int kh = kvh == 1 ? 0 : hi;
    const int t = *pos;
    const long cw = (long)kvh * 64;
    const float* knew = qkv + h * 64 + kh * 64;
    const float* vnew = qkv + h * 64 + kvh * 64 + kh * 64;
    // |k_new|^2, |q|^2 (every block; 64-lane reductions)
    float kn2 = knew[lane] * knew[lane], qn2 = qkv[hi * 64 + lane] * qkv[hi * 64 + lane];
    kn2 = wave_sum(kn2); qn2 = wave_sum(qn2);
    if (sp == 0 && w == 0) {   // append (blocks sharing a kv head write the same values: benign)
        kcache[t * cw + kh * 64 + lane] = knew[lane];
        vcache[t * cw + kh * 64 + lane] = vnew[lane];
        if (lane == 0) atomicMax(reinterpret_cast<unsigned int*>(kmax2 + kh), __float_as_uint(kn2));
    }
    const float slope = slopes ? slopes[hi] : 0.f;
    int j_lo = 0;
    if (slope > 0.f) {
        const float km = fmaxf(kmax2[kh], kn2);
        const float reach = (104.f + 2.f * scale * sqrtf(qn2 * km)) / slope;
        // rounded DOWN to a multiple of 256 (a few more keys than necessary, never fewer): the first key then moves once in 256 notes, which
        // is what lets the persistent layer launch (decode_layer.hip, same expression) request the rows of a split before q exists
        if (reach < (float)t) j_lo = (t - (int)reach - 1) & ~255;
    }
    const int total = t + 1 - j_lo;
    const int chunk = (total + S - 1) / S;
    const int j0 = j_lo + sp * chunk, j1 = min(t + 1, j0 + chunk);
    const f32x4 q4 = *reinterpret_cast<const f32x4*>(qkv + hi * 64 + l16 * 4) * scale;
    float m = -INFINITY, l = 0.f;
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    // the keys of a lane group, DEC_NU at a time (256 keys per block and batch: the whole split at L <= 4096): all row loads of a batch are in flight
    // together (a cache row read costs a trip to the Infinity Cache; one key per trip made this loop the longest part of a note)
    const f32x4 knew4 = *reinterpret_cast<const f32x4*>(knew + l16 * 4), vnew4 = *reinterpret_cast<const f32x4*>(vnew + l16 * 4);
    for (int jb0 = j0 + w * 4 + grp; jb0 < j1; jb0 += 256) {
        f32x4 k4[DEC_NU], v4[DEC_NU];
#pragma unroll
        for (int u = 0; u < DEC_NU; ++u) {
            const int j = max(min(min(jb0 + DEC_G * u, j1 - 1), t - 1), 0);   // row t is knew / vnew
            k4[u] = *reinterpret_cast<const f32x4*>(kcache + (long)j * cw + kh * 64 + l16 * 4);
            v4[u] = *reinterpret_cast<const f32x4*>(vcache + (long)j * cw + kh * 64 + l16 * 4);
        }
        dec_attn_batch<DEC_NU>(k4, v4, knew4, vnew4, q4, slope, t, jb0, j1, m, l, acc);
    }
    const int gi = w * 4 + grp;
    if (l16 == 0) { sm[gi] = m; sl[gi] = l; }
    *reinterpret_cast<f32x4*>(&so[gi][l16 * 4]) = acc;
    dec_attn_merge_wave(sm, sl, so, sm2, sl2, so2, w, lane);
    __syncthreads();
    // block result -> partial (m, l, o[64]) of (head, split)
    float* mine = part + ((long)hi * S + sp) * 66;
    if (tid < 64) {
        float mm, num, den;
        dec_attn_merge_block(sm2, sl2, so2, tid, lane, mm, num, den);
        mine[2 + tid] = num;
        if (tid == 0) { mine[0] = mm; mine[1] = den; }
    }
    if (!merge) return;   // the consumer GEMV merges the partials (DecGemvArgs::att_part)
    __threadfence();
    __syncthreads();
    if (tid == 0) is_last = (atomicAdd(counter + hi, 1) == S - 1);
    __syncthreads();
    if (!is_last) return;
    __threadfence();
    if (tid < 64) {
        const float* ph = part + (long)hi * S * 66;
        float mm = -INFINITY;
        for (int q = 0; q < S; ++q) mm = fmaxf(mm, ph[q * 66]);
        float num = 0.f, den = 0.f;
        for (int q = 0; q < S; ++q) {
            const float mq = ph[q * 66];
            const float f = (mq == -INFINITY) ? 0.f : __expf(mq - mm);
            num += ph[q * 66 + 2 + tid] * f; den += ph[q * 66 + 1] * f;
        }
        o[hi * 64 + tid] = num / den;
        if (tid == 0) counter[hi] = 0;   // ready for the next step
    }
}

// Cross-attention of ONE query over a static context (decoder layer type 'c', modules/transformer/transformer.py:201 with the cache
// protocol of :159-181): the reference calls the block with the last position only and NO cache, so keys / values are the whole
// context every step (here: projected once per render, kctx / vctx [nk, kvh*64]) and the ALiBi distance is measured from the END of the
// context -- get_bias(i = 1, j = nk, k = nk - 1), attention.py:193-197 -- whatever the decoded position is.  Masked context keys get
// the reference's fill value (attend.py:102-108).  Same split-key scheme as dec_attn2_kernel; nothing is appended.
__global__ __launch_bounds__(256) void dec_xattn_kernel(const float* __restrict__ q, const float* __restrict__ kctx, const float* __restrict__ vctx,
                                                        const float* __restrict__ slopes, const uint8_t* __restrict__ kmask, int nk,
                                                        const int* __restrict__ nk_dev,
                                                        float* __restrict__ o, float* __restrict__ part, int* __restrict__ counter,
                                                        int h, int kvh, float scale, int merge) {
    if (nk_dev) nk = *nk_dev;   // render sessions: the context grows from call to call under ONE captured graph (buffers sized for the window)
    __shared__ float sm[16], sl[16];
    __shared__ __attribute__((aligned(16))) float so[16][64];
    __shared__ int is_last;
    const int hi = blockIdx.x, sp = blockIdx.y, S = gridDim.y;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, grp = lane >> 4, l16 = lane & 15;
    const int kh = kvh == 1 ? 0 : hi;
    const long cw = (long)kvh * 64;
    const float slope = slopes ? slopes[hi] : 0.f;
    const int chunk = (nk + S - 1) / S;
    const int j0 = sp * chunk, j1 = min(nk, j0 + chunk);
    const f32x4 q4 = *reinterpret_cast<const f32x4*>(q + hi * 64 + l16 * 4) * scale;
    float m = -INFINITY, l = 0.f;
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int jb = j0 + w * 4 + grp; jb < j1; jb += 64) {   // four keys per lane group and trip to memory (see dec_attn2_kernel)
        f32x4 k4[4], v4[4];
        uint8_t km[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = min(jb + 16 * u, j1 - 1);
            k4[u] = *reinterpret_cast<const f32x4*>(kctx + j * cw + kh * 64 + l16 * 4);
            v4[u] = *reinterpret_cast<const f32x4*>(vctx + j * cw + kh * 64 + l16 * 4);
            km[u] = kmask ? kmask[j] : (uint8_t)1;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = jb + 16 * u;
            if (j >= j1) break;
            float sc = q4[0] * k4[u][0] + q4[1] * k4[u][1] + q4[2] * k4[u][2] + q4[3] * k4[u][3];
            sc = row16_sum(sc);   // the 16 lanes of a key: one DPP row (four ds_bpermute round trips per key before)
            sc -= slope * (float)(nk - 1 - j);
            if (km[u] == 0) sc = -1.7014118e38f;
            const float m_new = fmaxf(m, sc);
            const float alpha = __expf(m - m_new), pj = __expf(sc - m_new);
            l = l * alpha + pj;
            acc = acc * alpha + v4[u] * pj;
            m = m_new;
        }
    }
    const int gi = w * 4 + grp;
    if (l16 == 0) { sm[gi] = m; sl[gi] = l; }
    *reinterpret_cast<f32x4*>(&so[gi][l16 * 4]) = acc;
    __syncthreads();
    float* mine = part + ((long)hi * S + sp) * 66;
    if (tid < 64) {
        float mm = -INFINITY;
#pragma unroll
        for (int x = 0; x < 16; ++x) mm = fmaxf(mm, sm[x]);
        float num = 0.f, den = 0.f;
#pragma unroll
        for (int x = 0; x < 16; ++x) {
            const float f = (sm[x] == -INFINITY) ? 0.f : __expf(sm[x] - mm);
            num += so[x][tid] * f; den += sl[x] * f;
        }
        mine[2 + tid] = num;
        if (tid == 0) { mine[0] = mm; mine[1] = den; }
    }
    if (!merge) return;   // the consumer GEMV merges the partials (DecGemvArgs::att_part)
    __threadfence();
    __syncthreads();
    if (tid == 0) is_last = (atomicAdd(counter + hi, 1) == S - 1);
    __syncthreads();
    if (!is_last) return;
    __threadfence();
    if (tid < 64) {
        const float* ph = part + (long)hi * S * 66;
        float mm = -INFINITY;
        for (int x = 0; x < S; ++x) mm = fmaxf(mm, ph[x * 66]);
        float num = 0.f, den = 0.f;
        for (int x = 0; x < S; ++x) {
            const float mq = ph[x * 66];
            const float f = (mq == -INFINITY) ? 0.f : __expf(mq - mm);
            num += ph[x * 66 + 2 + tid] * f; den += ph[x * 66 + 1] * f;
        }
        o[hi * 64 + tid] = num / den;
        if (tid == 0) counter[hi] = 0;   // ready for the next step
    }
}

// LM head of one position for all candidate dims in one launch (grid = dims): LayerNorm(e) slice . table_dim^T -> arg-max with
// banned ids -> written where the next position holds MASK.  e: [D] head embedding (models/scoreperformer/embeddings.py:345-353).
struct DecHeadDesc {
    const float* table[16];   // [V, width] fp32
    int V[16], width[16], col0[16], dim[16];
    int n, D;
};
// SAMPLE: instead of the arg-max, top-k filtering + temperature + one multinomial draw (modules/sampling.py:28-59 top_k +
// filter_logits_and_sample), on the device: every slab also stores its logits, the last slab block ranks them (keep the topk[q]
// largest, ties to the lower id), normalises exp((l - max) / T) over the kept ones and inverts the CDF at a counter-based uniform
// u = hash(*seed, *pos, q).  The draw cannot reproduce torch.multinomial's stream; its distribution is the reference's.
struct DecSampleArgs { float* logits; int ldl; const int* topk; float inv_temperature; const unsigned* seed; };
__device__ __forceinline__ unsigned dec_mix(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
template <bool SAMPLE>
__global__ __launch_bounds__(256) void dec_head_kernel(DecHeadDesc d, const float* __restrict__ e, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float eps, unsigned ban_mask,
                                                       long* __restrict__ tokens, long tok_ld, int mask_id, const int* __restrict__ pos,
                                                       float* __restrict__ part, int* __restrict__ counter, DecSampleArgs sa,
                                                       int* __restrict__ pos_next) {
    __shared__ __attribute__((aligned(16))) float xs[2048];
    __shared__ float red[8];
    __shared__ float bv[4];
    __shared__ int bi[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, q = blockIdx.x;
    // last launch of a step: the NEXT step's position (read by its first launch, spn_dec_step_begin; nobody reads pos_next here)
    if (pos_next && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *pos_next = *pos + 1;
    const int c0 = d.col0[q], W = d.width[q], V = d.V[q];
    // rows of this dim are split over gridDim.y slabs (4 rows per slab step); the last slab to finish picks the winner.  A wave takes
    // its rows four at a time and requests the NEXT four before it multiplies the current ones; the first four are requested before
    // the LayerNorm prologue (table rows do not depend on the input): one row per trip to memory made this loop ~8 dependent latencies
    const int nsl = gridDim.y, sl = blockIdx.y, vstep = 4 * nsl;
    const bool batched = W <= 256;
    const float* tab = d.table[q];
    float rw[4][4], rn[4][4];
    auto load_rows = [&](float (&dst)[4][4], int vb) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float* row = tab + (long)min(vb + u * vstep, V - 1) * W;
#pragma unroll
            for (int c = 0; c < 4; ++c) dst[u][c] = (lane + 64 * c < W) ? row[lane + 64 * c] : 0.f;
        }
    };
    int vb = sl * 4 + w;
    if (batched && vb < V) load_rows(rw, vb);
    float s = 0.f;
    for (int k = threadIdx.x; k < d.D; k += 256) { const float v = e[k]; xs[k] = v; s += v; }
    float mu, rs;
    if ((d.D & 3) == 0) {   // statistics per wave over the dot-product layout (common.h dec_ln_stats: the head phase of decode_layer.hip)
        __syncthreads();
        f32x4 xv[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int k = lane * 4 + c * 256;
            xv[c] = k < d.D ? *reinterpret_cast<const f32x4*>(xs + k) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        dec_ln_stats<8>(xv, d.D, eps, lane, mu, rs);
        __syncthreads();   // every wave has read xs before the slice below is overwritten
    } else {
        s = wave_sum(s);
        if (lane == 0) red[w] = s;
        __syncthreads();
        mu = (red[0] + red[1] + red[2] + red[3]) / (float)d.D;
        float q2 = 0.f;
        for (int k = threadIdx.x; k < d.D; k += 256) { const float t = xs[k] - mu; q2 += t * t; }
        q2 = wave_sum(q2);
        if (lane == 0) red[4 + w] = q2;
        __syncthreads();
        rs = rsqrtf((red[4] + red[5] + red[6] + red[7]) / (float)d.D + eps);
    }
    for (int k = threadIdx.x; k < W; k += 256) xs[c0 + k] = (xs[c0 + k] - mu) * rs * gamma[c0 + k] + beta[c0 + k];
    __syncthreads();
    float best = -INFINITY;
    int idx = 0x7fffffff;
    auto take = [&](int v, float acc) {
        acc = wave_sum(acc);
        if (v < 32 && ((ban_mask >> v) & 1u)) acc = -INFINITY;
        if (SAMPLE && lane == 0) sa.logits[(long)q * sa.ldl + v] = acc;
        if (acc > best || (acc == best && v < idx)) { best = acc; idx = v; }
    };
    if (batched) {
        for (; vb < V; vb += 4 * vstep) {
            const bool more = vb + 4 * vstep < V;
            if (more) load_rows(rn, vb + 4 * vstep);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int v = vb + u * vstep;
                if (v >= V) break;
                float acc = 0.f;   // k = lane, lane + 64, ...: the order of the one-row-at-a-time loop
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (lane + 64 * c < W) acc = fmaf(rw[u][c], xs[c0 + lane + 64 * c], acc);
                take(v, acc);
            }
            if (more) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int c = 0; c < 4; ++c) rw[u][c] = rn[u][c];
            }
        }
    } else {
        for (int v = vb; v < V; v += vstep) {
            const float* row = tab + (long)v * W;
            float acc = 0.f;
            for (int k = lane; k < W; k += 64) acc = fmaf(row[k], xs[c0 + k], acc);
            take(v, acc);
        }
    }
    if (lane == 0) { bv[w] = best; bi[w] = idx; }
    __syncthreads();
    if (SAMPLE) {
        __shared__ int last_flag;
        if (threadIdx.x == 0) {
            __threadfence();
            last_flag = atomicAdd(counter + q, 1) == nsl - 1;
        }
        __syncthreads();
        if (!last_flag) return;
        __threadfence();
        // all V logits of this key are in global memory now: rank, filter, normalise, draw (V <= 2048: reuse xs)
        for (int v = threadIdx.x; v < V; v += 256) xs[v] = sa.logits[(long)q * sa.ldl + v];
        __syncthreads();
        const int keep_n = sa.topk[q];
        float mx = -INFINITY;
        for (int v = threadIdx.x; v < V; v += 256) mx = fmaxf(mx, xs[v]);
        mx = wave_max(mx);
        if (lane == 0) red[w] = mx;
        __syncthreads();
        mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        __syncthreads();
        float mine_sum = 0.f;
        // weights in place of the logits (two passes: ranks read every logit, so the weights go to the upper half of xs... V <= 1024)
        for (int v = threadIdx.x; v < V; v += 256) {
            const float lv = xs[v];
            int rank = 0;
            for (int u = 0; u < V; ++u) { const float lu = xs[u]; rank += (lu > lv) || (lu == lv && u < v); }
            const float wgt = (rank < keep_n && lv > -INFINITY) ? __expf((lv - mx) * sa.inv_temperature) : 0.f;
            xs[1024 + v] = wgt;
            mine_sum += wgt;
        }
        (void)mine_sum;
        __syncthreads();
        if (w == 0) {   // the draw: decode_sample.h (one definition for this launch and the head phase of the persistent one)
            const unsigned r = dec_mix(dec_mix(*sa.seed ^ 0x9e3779b9u * (unsigned)(*pos + 1)) + 0x85ebca6bu * (unsigned)(q + 1));
            const int pick = dec_sample_pick(xs + 1024, V, (float)(r >> 8) * (1.f / 16777216.f), lane);
            if (lane == 0) {
                long* cell = tokens + (long)(*pos + 1) * tok_ld + d.dim[q];
                if (*cell == mask_id) *cell = pick;
                counter[q] = 0;
            }
        }
        return;
    }
    if (threadIdx.x == 0) {
        for (int r = 1; r < 4; ++r) if (bv[r] > best || (bv[r] == best && bi[r] < idx)) { best = bv[r]; idx = bi[r]; }
        float* mine = part + ((long)q * nsl + sl) * 2;
        mine[0] = best; mine[1] = __int_as_float(idx);
        __threadfence();
        if (atomicAdd(counter + q, 1) == nsl - 1) {
            __threadfence();
            for (int r = 0; r < nsl; ++r) {
                const float bvv = part[((long)q * nsl + r) * 2];
                const int bii = __float_as_int(part[((long)q * nsl + r) * 2 + 1]);
                if (r == 0 || bvv > best || (bvv == best && bii < idx)) { best = bvv; idx = bii; }
            }
            long* cell = tokens + (long)(*pos + 1) * tok_ld + d.dim[q];
            if (*cell == mask_id) *cell = idx;
            counter[q] = 0;
        }
    }
}

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

// both sequences' tuple embeddings + projection in one launch: y [2, N] (tokens_a at row *pos, tokens_b at row *pos + 1)
extern "C" int spn_dec_embed_proj(int nkeys, const float* const* tables, const int* E, const long* tokens_a, const long* tokens_b, long tok_ld,
                                  const int* pos, const float* gamma, const float* beta, float eps, const float* W, long ldw, const float* bias,
                                  float* y, int N, hipStream_t s) {
    SPN_REQUIRE(nkeys > 0 && nkeys <= 16 && tokens_a && tokens_b && pos && W && y && N > 0, "spn_dec_embed_proj: bad arguments");
    DecEmbedDesc d;
    memset(&d, 0, sizeof(d));
    int col = 0;
    for (int i = 0; i < nkeys; ++i) { d.table[i] = tables[i]; d.width[i] = E[i]; d.col0[i] = col; col += E[i]; }
    d.nkeys = nkeys; d.D = col;
    SPN_REQUIRE(col <= 2048, "spn_dec_embed_proj: total width <= 2048");
    DecRider r;
    memset(&r, 0, sizeof(r));
    const int nbe = cdiv(N, 4);
    hipLaunchKernelGGL(dec_embed_proj_kernel, dim3(2 * nbe), dim3(256), 0, s, d, tokens_a, tokens_b, tok_ld, pos, (int*)nullptr, gamma, beta, eps,
                       W, ldw, bias, y, N, nbe, r);
    SPN_LAUNCH_CHECK();
    return SPN_OK;
}

// First launch of a fused step: spn_dec_embed_proj + the position latch (*pos_latch = *pos, see dec_embed_proj_kernel) + an optional
// rider GEMV ry[n] = rW[n, :] . rx[(*pos + rx_off) * rx_ld ..] + rbias[n] (rW = null: none) in the same launch.
extern "C" int spn_dec_step_begin(int nkeys, const float* const* tables, const int* E, const long* tokens_a, const long* tokens_b, long tok_ld,
                                  const int* pos, int* pos_latch, const float* gamma, const float* beta, float eps, const float* W, long ldw,
                                  const float* bias, float* y, int N, const float* rW, long r_ldw, int rN, int rK, const float* rx, long rx_ld,
                                  int rx_off, const float* rbias, float* ry, hipStream_t s) {
    SPN_REQUIRE(nkeys > 0 && nkeys <= 16 && tokens_a && tokens_b && pos && pos_latch && W && y && N > 0, "spn_dec_step_begin: bad arguments");
    SPN_REQUIRE(!rW || (rN > 0 && rK > 0 && rx && ry), "spn_dec_step_begin: bad rider arguments");
    DecEmbedDesc d;
    memset(&d, 0, sizeof(d));
    int col = 0;
    for (int i = 0; i < nkeys; ++i) { d.table[i] = tables[i]; d.width[i] = E[i]; d.col0[i] = col; col += E[i]; }
    d.nkeys = nkeys; d.D = col;
    SPN_REQUIRE(col <= 2048, "spn_dec_step_begin: total width <= 2048");
    DecRider r;
    memset(&r, 0, sizeof(r));
    if (rW) { r.W = rW; r.ldw = r_ldw; r.N = rN; r.K = rK; r.x = rx; r.x_ld = rx_ld; r.x_off = rx_off; r.bias = rbias; r.y = ry; }
    const int nbe = cdiv(N, 4);
    hipLaunchKernelGGL(dec_embed_proj_kernel, dim3(2 * nbe + (rW ? cdiv(rN, 4) : 0)), dim3(256), 0, s, d, tokens_a, tokens_b, tok_ld, pos,
                       pos_latch, gamma, beta, eps, W, ldw, bias, y, N, nbe, r);
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

// ---- fused entry points -------------------------------------------------------------------------------------------------
extern "C" int spn_dec_fused_gemv(const float* W, long ldw, int N, int K, const float* x, long x_ld, int x_off, int norm,
                                  const float* gamma, const float* beta, float eps, const float* bias, const float* residual, float* y,
                                  long y_ld, int y_off, float* y2, long y2_ld, int y2_off, float* xn_out, long xn_ld, int xn_off, int glu,
                                  int act, const int* pos, hipStream_t s) {
    SPN_REQUIRE(W && x && y && N > 0 && K > 0 && K <= 2048, "spn_dec_fused_gemv: bad arguments (K <= 2048)");
    SPN_REQUIRE(norm >= 0 && norm <= 2 && (norm != 2 || gamma), "spn_dec_fused_gemv: bad norm mode");
    DecGemvArgs a{W, ldw, N, K, x, x_ld, x_off, norm, gamma, beta, eps, bias, residual, y, y_ld, y_off, y2, y2_ld, y2_off,
                  xn_out, xn_ld, xn_off, glu, act, pos, nullptr, 0, 0, 0, nullptr, 0, 0, nullptr, 0, 0};
    hipLaunchKernelGGL(dec_fused_gemv_kernel, dim3(cdiv(N, 4)), dim3(256), 0, s, a);
    SPN_LAUNCH_CHECK();
    return SPN_OK;
}

// y = W . ( LN?(x[0:d]) | ctx[*pos + 1] | style[*pos + 1] ) + bias, mirrored into row *pos of y2: spn_dec_cat + spn_dec_fused_gemv in one
// launch (the decoder's input projection over the concatenated embeddings, models/scoreperformer/transformer.py:160-181)
extern "C" int spn_dec_cat_gemv(const float* W, long ldw, int N, const float* x, int d, const float* gamma, const float* beta, float eps,
                                const float* ctx, long ctx_ld, int ctx_w, const float* style, long style_ld, int style_w, const float* bias,
                                float* y, float* y2, long y2_ld, const int* pos, hipStream_t s) {
    const int K = d + (ctx ? ctx_w : 0) + (style ? style_w : 0);
    SPN_REQUIRE(W && x && y && pos && N > 0 && d > 0 && K <= 2048 && (!gamma || beta), "spn_dec_cat_gemv: bad arguments (K <= 2048)");
    DecGemvArgs a{W, ldw, N, K, x, 0, 0, 0, gamma, beta, eps, bias, nullptr, y, 0, 0, y2, y2_ld, 0,
                  nullptr, 0, 0, 0, -1, pos, nullptr, 0, 0, d, ctx, ctx_ld, ctx_w, style, style_ld, style_w};
    hipLaunchKernelGGL(dec_fused_gemv_kernel, dim3(cdiv(N, 4)), dim3(256), 0, s, a);
    SPN_LAUNCH_CHECK();
    return SPN_OK;
}

// Attention output projection whose input is merged on the fly from the split-key partials `part` [h][splits][66] that spn_dec_attn2 /
// spn_dec_xattn leave when called with o = null:  y = residual + W . merge(part)   (W: [N, h*64]; attention.py:210-218 for one query)
extern "C" int spn_dec_attn_out(const float* W, long ldw, int N, const float* part, int h, int splits, const float* residual, float* y,
                                hipStream_t s) {
    SPN_REQUIRE(W && part && y && N > 0 && h > 0 && h * 64 <= 2048 && splits > 0 && splits <= 64, "spn_dec_attn_out: bad arguments");
    DecGemvArgs a{W, ldw, N, h * 64, nullptr, 0, 0, 0, nullptr, nullptr, 0.f, nullptr, residual, y, 0, 0, nullptr, 0, 0,
                  nullptr, 0, 0, 0, -1, nullptr, part, h, splits, 0, nullptr, 0, 0, nullptr, 0, 0};
    hipLaunchKernelGGL(dec_fused_gemv_kernel, dim3(cdiv(N, 4)), dim3(256), 0, s, a);
    SPN_LAUNCH_CHECK();
    return SPN_OK;
}

extern "C" int spn_dec_cat(const float* x, int d, const float* gamma, const float* beta, float eps, const float* ctx, long ctx_ld, int ctx_w,
                           const float* style, long style_ld, int style_w, const int* pos, float* out, hipStream_t s) {
    SPN_REQUIRE(x && out && pos && d > 0, "spn_dec_cat: bad arguments");
    hipLaunchKernelGGL(dec_cat_kernel, dim3(1), dim3(256), 0, s, x, d, gamma, beta, eps, ctx, ctx_ld, ctx_w, style, style_ld, style_w, pos, out);
    SPN_LAUNCH_CHECK();
    return SPN_OK;
}

// part: h * splits * 66 floats scratch; counter: h ints, zero before the first step (the kernel resets it); kmax2: kvh floats, zero at start
// o == null: partials only (no fences, no last-block merge); the consumer spn_dec_fused_gemv merges them (att_part argument)
extern "C" int spn_dec_attn2(const float* qkv, float* kcache, float* vcache, const float* slopes, const int* pos, float* o, float* part,
                             int* counter, float* kmax2, int h, int kvh, float scale, int splits, hipStream_t s) {
    SPN_REQUIRE(qkv && kcache && vcache && pos && part && counter && kmax2 && h > 0 && (kvh == 1 || kvh == h) && splits > 0 && splits <= 64,
                "spn_dec_attn2: bad arguments");
    hipLaunchKernelGGL(dec_attn2_kernel, dim3(h, splits), dim3(512), 0, s, qkv, kcache, vcache, slopes, pos, o, part, counter, kmax2, h, kvh, scale,
                       o ? 1 : 0);
    SPN_LAUNCH_CHECK();
    return SPN_OK;
}

// q: [h*64] fp32; kctx / vctx: [nk, kvh*64] fp32 (projected context, constant over the render); kmask: [nk] uint8 or null;
// part: h * splits * 66 floats scratch; counter: h ints, zero before the first step (the kernel resets it)
extern "C" int spn_dec_xattn(const float* q, const float* kctx, const float* vctx, const float* slopes, const uint8_t* kmask, int nk, float* o,
                             float* part, int* counter, int h, int kvh, float scale, int splits, hipStream_t s) {
    SPN_REQUIRE(q && kctx && vctx && part && counter && nk > 0 && h > 0 && (kvh == 1 || kvh == h) && splits > 0 && splits <= 64,
                "spn_dec_xattn: bad arguments");
    hipLaunchKernelGGL(dec_xattn_kernel, dim3(h, splits), dim3(256), 0, s, q, kctx, vctx, slopes, kmask, nk, (const int*)nullptr, o, part, counter, h,
                       kvh, scale, o ? 1 : 0);
    SPN_LAUNCH_CHECK();
    return SPN_OK;
}

// the same with the number of context rows read from DEVICE memory at run time (*nk_dev, 1 <= *nk_dev <= rows of kctx / vctx / kmask)
extern "C" int spn_dec_xattn_dyn(const float* q, const float* kctx, const float* vctx, const float* slopes, const uint8_t* kmask,
                                 const int* nk_dev, float* o, float* part, int* counter, int h, int kvh, float scale, int splits, hipStream_t s) {
    SPN_REQUIRE(q && kctx && vctx && part && counter && nk_dev && h > 0 && (kvh == 1 || kvh == h) && splits > 0 && splits <= 64,
                "spn_dec_xattn_dyn: bad arguments");
    hipLaunchKernelGGL(dec_xattn_kernel, dim3(h, splits), dim3(256), 0, s, q, kctx, vctx, slopes, kmask, 1, nk_dev, o, part, counter, h, kvh, scale,
                       o ? 1 : 0);
    SPN_LAUNCH_CHECK();
    return SPN_OK;
}

namespace { __global__ void dec_lookup_kernel(const int* __restrict__ tab, const int* __restrict__ pos, int* __restrict__ out) { out[0] = tab[pos[0]]; } }
// out[0] = tab[*pos]: a device-side row index that is a table function of the position (the reference-compatible head row of
// cross-attending decoders, decode.py)
extern "C" int spn_dec_lookup(const int* tab, const int* pos, int* out, hipStream_t s) {
    SPN_REQUIRE(tab && pos && out, "spn_dec_lookup: bad arguments");
    hipLaunchKernelGGL(dec_lookup_kernel, dim3(1), dim3(1), 0, s, tab, pos, out);
    SPN_LAUNCH_CHECK();
    return SPN_OK;
}

extern "C" int spn_dec_head(int n, const float* const* tables, const int* V, const int* width, const int* col0, const int* dim, int D,
                            const float* e, const float* gamma, const float* beta, float eps, unsigned ban_mask, long* tokens, long tok_ld,
                            int mask_id, const int* pos, float* part, int* counter, int slabs, int* pos_next, hipStream_t s) {
    SPN_REQUIRE(n > 0 && n <= 16 && tables && e && gamma && beta && tokens && pos && D > 0 && D <= 2048, "spn_dec_head: bad arguments");
    SPN_REQUIRE(part && counter && slabs > 0 && slabs <= 64, "spn_dec_head: scratch (n*slabs*2 floats, n zeroed ints) required");
    DecHeadDesc d;
    memset(&d, 0, sizeof(d));
    for (int i = 0; i < n; ++i) { d.table[i] = tables[i]; d.V[i] = V[i]; d.width[i] = width[i]; d.col0[i] = col0[i]; d.dim[i] = dim[i]; }
    d.n = n; d.D = D;
    hipLaunchKernelGGL(dec_head_kernel<false>, dim3(n, slabs), dim3(256), 0, s, d, e, gamma, beta, eps, ban_mask, tokens, tok_ld, mask_id, pos, part,
                       counter, DecSampleArgs{}, pos_next);
    SPN_LAUNCH_CHECK();
    return SPN_OK;
}

extern "C" int spn_dec_head_sample(int n, const float* const* tables, const int* V, const int* width, const int* col0, const int* dim, int D,
                                   const float* e, const float* gamma, const float* beta, float eps, unsigned ban_mask, long* tokens,
                                   long tok_ld, int mask_id, const int* pos, float* part, int* counter, int slabs, float* logits, int ldl,
                                   const int* topk, float temperature, const unsigned* seed, int* pos_next, hipStream_t s) {
    SPN_REQUIRE(n > 0 && n <= 16 && tables && e && gamma && beta && tokens && pos && D > 0 && D <= 2048, "spn_dec_head_sample: bad arguments");
    SPN_REQUIRE(part && counter && slabs > 0 && slabs <= 64 && logits && topk && seed && temperature > 0.f, "spn_dec_head_sample: bad arguments");
    DecHeadDesc d;
    memset(&d, 0, sizeof(d));
    for (int i = 0; i < n; ++i) {
        SPN_REQUIRE(V[i] <= 1024 && V[i] <= ldl, "spn_dec_head_sample: vocabularies up to 1024 ids");
        d.table[i] = tables[i]; d.V[i] = V[i]; d.width[i] = width[i]; d.col0[i] = col0[i]; d.dim[i] = dim[i];
    }
    d.n = n; d.D = D;
    hipLaunchKernelGGL(dec_head_kernel<true>, dim3(n, slabs), dim3(256), 0, s, d, e, gamma, beta, eps, ban_mask, tokens, tok_ld, mask_id, pos, part,
                       counter, DecSampleArgs{logits, ldl, topk, 1.f / temperature, seed}, pos_next);
    SPN_LAUNCH_CHECK();
    return SPN_OK;
}

extern "C" int spn_dec_attn_rows(const float* q, long q_ld, const float* kcache, const float* vcache, const float* slopes, int t0, int n,
                                 float* o, long o_ld, int h, int kvh, float scale, hipStream_t s) {
    SPN_REQUIRE(q && kcache && vcache && o && n > 0 && t0 >= 0 && h > 0 && (kvh == 1 || kvh == h), "spn_dec_attn_rows: bad arguments");
    SPN_REQUIRE((q_ld & 3) == 0 && (reinterpret_cast<uintptr_t>(q) & 15) == 0, "spn_dec_attn_rows: query rows must be 16-byte aligned (float4 loads)");
    hipLaunchKernelGGL(dec_attn_rows_kernel, dim3(h, n), dim3(512), 0, s, q, q_ld, kcache, vcache, slopes, t0, o, o_ld, h, kvh, scale);
    SPN_LAUNCH_CHECK();
    return SPN_OK;
}

extern "C" int spn_dec_glu_rows(const float* u, long u_ld, float* out, long out_ld, int n, int I, int act, int glu, hipStream_t s) {
    SPN_REQUIRE(u && out && n > 0 && I > 0, "spn_dec_glu_rows: bad arguments");
    hipLaunchKernelGGL(dec_glu_rows_kernel, dim3((I + 255) / 256, n), dim3(256), 0, s, u, u_ld, out, out_ld, I, act, glu);
    SPN_LAUNCH_CHECK();
    return SPN_OK;
}
