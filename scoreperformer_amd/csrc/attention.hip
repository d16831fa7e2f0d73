// Fused (flash-style) attention for gfx950, head dim 64, bf16 in/out, fp32 softmax statistics.
//
// Replaces `Attend.efficient_attn` (modules/transformer/attend.py:58-126) together with the mask / ALiBi assembly
// of `Attention.forward` (modules/transformer/attention.py:162-197) and `ALiBiPositionalBias`
// (modules/transformer/embeddings.py:294-315):
//     out = softmax(q k^T * scale + slope_h * (-|j - (i + nk - nq)|) + mask) v
// with mask = key-padding AND causal (triu(nk - nq + 1)), masked scores set to -FLT_MAX/2 exactly as the
// reference's additive mask does (attend.py:102-108).  Multi-query attention (one shared K/V head,
// attention.py:67-73) is expressed through a zero head stride.  The b*h*n^2 score tensor is never materialised.
//
// MFMA orientation (v_mfma_f32_16x16x32_bf16; C/D layout col = lane&15, row = (lane>>4)*4 + r):
//   forward / dQ kernels compute S^T = K Q^T, so each lane owns ONE query column and 4 keys per 16x16 block:
//   the softmax row reductions are lane-local plus two cross-lane-group shuffles, and the C-layout registers of
//   P^T are directly the B operand of O^T = V^T P^T (the contraction index only has to be enumerated the same
//   way in A and B).  V^T / K^T / dO^T / Q^T A-operands come from row-major LDS tiles through the hardware
//   transpose read ds_read_b64_tr_b16.
//   the dK/dV kernel uses the S = Q K^T orientation (lane owns one key column), which makes P and dS the
//   B operands of dV^T = dO^T P and dK^T = Q^T dS; it loops over all heads that share the K/V head (MQA), so
//   dK/dV are reduced over heads in registers without atomics.
//
// VALU diet (the softmax, not the MFMAs, bounds a 64-wide-head kernel): scores live in the log2 domain
// (t = s*scale*log2e + bias*log2e, one v_exp_f32 per element); a 64-key tile that lies entirely on one side of a
// wave's 32 query rows has a LINEAR ALiBi term, whose per-row part is folded into the running max, leaving
// add + fma + max + sub + exp + add per score; masks are only evaluated on tiles that contain a masked key, the causal
// diagonal or the sequence end; the O accumulator is rescaled only when a row maximum grows by more than 2^8.
#include "common.h"

namespace {

constexpr float NEG_FILL = -1.7014118e38f;  // -finfo(float32).max // 2   (attend.py:102)
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;
constexpr float RESCALE_THR = 8.f;          // log2 units

struct AttnArgs {
    const bf16_t* q; const bf16_t* k; const bf16_t* v;
    bf16_t* o; float* lse;
    const bf16_t* d_o; const float* delta;   // backward only
    bf16_t* dq; bf16_t* dk; bf16_t* dv; float* dslope;
    const uint8_t* kmask;   // [b, nk] or null
    const float* slopes;    // [h] or null
    int b, h, kvh, nq, nk, causal;
    long q_bs, q_ns, q_hs;
    long k_bs, k_ns, k_hs;
    long v_bs, v_ns, v_hs;
    long o_bs, o_ns, o_hs;     // o and d_o share strides
    long dq_bs, dq_ns, dq_hs;
    long dk_bs, dk_ns, dk_hs;
    long dv_bs, dv_ns, dv_hs;
    float scale;
};

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// [64 rows][64 cols] bf16 tile (128-byte rows, 8 chunks of 16 B) ------------------------------------------
// "a" layout: chunk ^ (row & 7)           -> conflict-light ds_read_b128 of (row = lane&15, chunk = lane>>4)
// "t" layout: chunk ^ (((row>>1)&3) << 1) -> the 8 rows touched by two lane groups of a transpose read fall
//                                            into 8 distinct 32-byte windows
__device__ __forceinline__ int a_off(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }
__device__ __forceinline__ int t_off(int row, int chunk) { return row * 128 + ((chunk ^ (((row >> 1) & 3) << 1)) << 4); }

// A-operand fragment, rows = tile rows r_base + (lane&15), k = 32*ks + (lane>>4)*8 + e  (row-major "a" tile)
__device__ __forceinline__ bf16x8 frag_rows(const char* tile, int r_base, int ks, int lane) {
    return *reinterpret_cast<const bf16x8*>(tile + a_off(r_base + (lane & 15), ks * 4 + (lane >> 4)));
}

// A-operand fragment of the TRANSPOSED tile: rows = tile columns c_base + (lane&15); contraction index e of lane
// group g enumerates tile rows  32*u + 16*(e>>2) + 4*g + (e&3)   ("t" tile) -- the same enumeration that the
// C-layout registers of two stacked 16x16 blocks give when used as a B operand.
__device__ __forceinline__ bf16x8 frag_cols_t(const char* tile, int c_base, int u, int lane) {
    const int g = lane >> 4, p = lane & 15;
    const int col_byte = (c_base + 4 * (p & 3)) * 2;
    const int chunk = col_byte >> 4, within = col_byte & 15;
    const int r0 = 32 * u + 4 * g + (p >> 2);
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(tile + t_off(r0, chunk) + within));
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(tile + t_off(r0 + 16, chunk) + within));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

__device__ __forceinline__ bf16x8 pack8(const f32x4& a, const f32x4& b) {
    uint4 u;
    u.x = pack_bf2(a[0], a[1]); u.y = pack_bf2(a[2], a[3]);
    u.z = pack_bf2(b[0], b[1]); u.w = pack_bf2(b[2], b[3]);
    return __builtin_bit_cast(bf16x8, u);
}

// global -> registers for a [64][64] tile: 512 chunks of 16 B, 2 per thread (256 threads)
struct TileRegs {
    uint4 r[2];
    __device__ __forceinline__ void load(const bf16_t* base, long row_stride, int row0, int nrows, int tid) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = tid + 256 * i, row = c >> 3, ch = c & 7;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (row0 + row < nrows) v = *reinterpret_cast<const uint4*>(base + (long)(row0 + row) * row_stride + ch * 8);
            r[i] = v;
        }
    }
    template <bool T>
    __device__ __forceinline__ void store(char* tile, int tid) const {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = tid + 256 * i, row = c >> 3, ch = c & 7;
            *reinterpret_cast<uint4*>(tile + (T ? t_off(row, ch) : a_off(row, ch))) = r[i];
        }
    }
};

__device__ __forceinline__ bf16x8 load_row_frag(const bf16_t* base, long row_stride, int row, int nrows, int ks, int lane) {
    uint4 v = make_uint4(0, 0, 0, 0);
    if (row < nrows) v = *reinterpret_cast<const uint4*>(base + (long)row * row_stride + ks * 32 + (lane >> 4) * 8);
    return __builtin_bit_cast(bf16x8, v);
}

__device__ __forceinline__ float group_max(float v) {  // across the 4 lane groups (same lane&15)
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float group_sum(float v) {
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
}

// Causal work per query tile grows linearly with its index, and consecutive blockIdx.x land on consecutive XCDs
// (block id % 8): map x -> tile so that XCD k owns tiles {k, k+8, ...} of the first half and their mirror images of the
// second half, i.e. every XCD gets the same total number of key tiles (a speed-only remap; any mapping is correct).
__device__ __forceinline__ int qtile_of(int x, int n, int causal) {
    if (!causal || n < 16 || (n & 15)) return x;
    const int half = n >> 1;
    return x < half ? x : (n - 1 - (x - half));
}

// Tile classes for a wave's block of query rows [i_lo, i_hi] (in key coordinates, i + nk - nq) against keys [j0, j0+63]:
enum { T_GEN = 0, T_LEFT = 1, T_RIGHT = 2, T_SKIP = 3 };
__device__ __forceinline__ int classify(int j0, int i_lo, int i_hi, bool full, bool causal) {
    if (causal && j0 > i_hi) return T_SKIP;            // every key is in the future of every row of this wave
    if (!full) return T_GEN;
    if (j0 + 63 <= i_lo) return T_LEFT;                // all distances j - i <= 0: causal-clean, |d| = i - j
    if (j0 >= i_hi && !causal) return T_RIGHT;         // all distances >= 0: |d| = j - i
    return T_GEN;
}

// log2-domain scores of one (kb, qb) 16x16 block column for this lane.  Returns the value relative to the per-row offset u:
//   LEFT : t = s*c1 + slope2*(j - i)  = [s*c1 + slope2*j] + u,  u = -slope2*i
//   RIGHT: t = s*c1 - slope2*(j - i)  = [s*c1 - slope2*j] + u,  u = +slope2*i
//   GEN  : t = s*c1 - slope2*|j - i| (u = 0), masked entries -> NEG_FILL
template <int MODE>
__device__ __forceinline__ float score(float s, float c1, float slope2, float sj, float jf, float i_f, bool ok) {
    if (MODE == T_LEFT) return fmaf(s, c1, sj);
    if (MODE == T_RIGHT) return fmaf(s, c1, -sj);
    const float t = fmaf(-slope2, fabsf(jf - i_f), s * c1);
    return ok ? t : NEG_FILL;
}

// ==========================================================================================================
// forward: grid (ceil(nq/128), h, b), 256 threads; wave w owns query rows q0 + 32w .. +31
// ==========================================================================================================
template <int MODE>
__device__ __forceinline__ void fwd_softmax(f32x4 (&s)[4][2], f32x4 (&o)[4][2], float (&m_run)[2], float (&l_run)[2],
                                            const float (&i_f)[2], float c1, float slope2, float j0f, int g, const uint8_t* m_tile,
                                            bool causal) {
    uint32_t mbits[4] = {0, 0, 0, 0};
    if (MODE == T_GEN) {
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) mbits[kb] = *reinterpret_cast<const uint32_t*>(m_tile + 16 * kb + 4 * g);
    }
    const float sj0 = slope2 * (j0f + (float)(4 * g));
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const float u = MODE == T_LEFT ? -slope2 * i_f[qb] : (MODE == T_RIGHT ? slope2 * i_f[qb] : 0.f);
        float tmax = NEG_FILL;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float jc = (float)(16 * kb + r);
                const float jf = j0f + (float)(4 * g) + jc;
                const bool ok = MODE != T_GEN || ((((mbits[kb] >> (8 * r)) & 0xff) != 0) && (!causal || jf <= i_f[qb]));
                const float t = score<MODE>(s[kb][qb][r], c1, slope2, sj0 + slope2 * jc, jf, i_f[qb], ok);
                s[kb][qb][r] = t;
                tmax = fmaxf(tmax, t);
            }
        tmax = group_max(tmax) + u;
        if (__any(tmax > m_run[qb] + RESCALE_THR)) {   // wave-uniform: rescale only when some row's max really grew
            const float m_new = fmaxf(m_run[qb], tmax);
            const float alpha = fast_exp2(m_run[qb] - m_new);
            m_run[qb] = m_new;
            l_run[qb] *= alpha;
#pragma unroll
            for (int db = 0; db < 4; ++db) o[db][qb] *= alpha;
        }
        const float mm = m_run[qb] - u;
        float psum = 0.f;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = fast_exp2(s[kb][qb][r] - mm);
                s[kb][qb][r] = p;
                psum += p;
            }
        l_run[qb] += psum;
    }
}

__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(AttnArgs a) {
    __shared__ __attribute__((aligned(16))) char smem[2 * 8192 + 64 + 16];
    char* k_tile = smem;            // "a" layout
    char* v_tile = smem + 8192;     // "t" layout
    uint8_t* m_tile = reinterpret_cast<uint8_t*>(smem + 16384);
    int* full_flag = reinterpret_cast<int*>(smem + 16384 + 64);

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g = lane >> 4, c = lane & 15;
    const int bi = blockIdx.z, hi = blockIdx.y, q0 = qtile_of(blockIdx.x, gridDim.x, a.causal) * 128;
    const int kh = (a.kvh == 1) ? 0 : hi;
    const int off = a.nk - a.nq;
    const bf16_t* qp = a.q + bi * a.q_bs + hi * a.q_hs;
    const bf16_t* kp = a.k + bi * a.k_bs + kh * a.k_hs;
    const bf16_t* vp = a.v + bi * a.v_bs + kh * a.v_hs;
    const uint8_t* mp = a.kmask ? a.kmask + (long)bi * a.nk : nullptr;
    const float slope2 = a.slopes ? a.slopes[hi] * LOG2E : 0.f;
    const float c1 = a.scale * LOG2E;

    bf16x8 qf[2][2];
    float i_f[2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        i_f[qb] = (float)(q0 + 32 * w + 16 * qb + c + off);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) qf[qb][ks] = load_row_frag(qp, a.q_ns, q0 + 32 * w + 16 * qb + c, a.nq, ks, lane);
    }
    const int i_lo = q0 + 32 * w + off, i_hi = i_lo + 31;

    f32x4 o[4][2];
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) o[db][qb] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run[2] = {NEG_FILL, NEG_FILL}, l_run[2] = {0.f, 0.f};

    int nt = (a.nk + 63) / 64;
    if (a.causal) {
        const int last = q0 + 127 + off;  // largest key index any row of this block may see
        nt = last < 0 ? 0 : min(nt, last / 64 + 1);
    }

    TileRegs kr, vr;
    uint8_t mreg = 1;
    if (nt > 0) {
        kr.load(kp, a.k_ns, 0, a.nk, tid);
        vr.load(vp, a.v_ns, 0, a.nk, tid);
        if (tid < 64) mreg = (tid < a.nk) ? (mp ? mp[tid] : 1) : 0;
    }
    for (int t = 0; t < nt; ++t) {
        const int j0 = t * 64;
        __syncthreads();
        kr.store<false>(k_tile, tid);
        vr.store<true>(v_tile, tid);
        if (tid < 64) {
            m_tile[tid] = mreg;
            const unsigned long long all = __ballot(mreg != 0);
            if (tid == 0) *full_flag = (all == ~0ull) ? 1 : 0;
        }
        __syncthreads();
        if (t + 1 < nt) {
            kr.load(kp, a.k_ns, j0 + 64, a.nk, tid);
            vr.load(vp, a.v_ns, j0 + 64, a.nk, tid);
            if (tid < 64) { const int j = j0 + 64 + tid; mreg = (j < a.nk) ? (mp ? mp[j] : 1) : 0; }
        }
        const int cls = classify(j0, i_lo, i_hi, *full_flag != 0, a.causal != 0);
        if (cls == T_SKIP) continue;

        // S^T = K Q^T
        f32x4 s[4][2];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            bf16x8 kf0 = frag_rows(k_tile, 16 * kb, 0, lane), kf1 = frag_rows(k_tile, 16 * kb, 1, lane);
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf0, qf[qb][0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf1, qf[qb][1], acc, 0, 0, 0);
                s[kb][qb] = acc;
            }
        }
        const float j0f = (float)j0;
        if (cls == T_LEFT) fwd_softmax<T_LEFT>(s, o, m_run, l_run, i_f, c1, slope2, j0f, g, m_tile, a.causal);
        else if (cls == T_RIGHT) fwd_softmax<T_RIGHT>(s, o, m_run, l_run, i_f, c1, slope2, j0f, g, m_tile, a.causal);
        else fwd_softmax<T_GEN>(s, o, m_run, l_run, i_f, c1, slope2, j0f, g, m_tile, a.causal);

        // O^T += V^T P^T
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            bf16x8 pf[2];
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) pf[qb] = pack8(s[2 * u][qb], s[2 * u + 1][qb]);
#pragma unroll
            for (int db = 0; db < 4; ++db) {
                bf16x8 vf = frag_cols_t(v_tile, 16 * db, u, lane);
#pragma unroll
                for (int qb = 0; qb < 2; ++qb)
                    o[db][qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[qb], o[db][qb], 0, 0, 0);
            }
        }
    }

#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const int i = q0 + 32 * w + 16 * qb + c;
        const float l = group_sum(l_run[qb]);
        const float inv = l > 0.f ? 1.f / l : 0.f;
        if (i < a.nq) {
            bf16_t* op = a.o + bi * a.o_bs + (long)i * a.o_ns + hi * a.o_hs;
#pragma unroll
            for (int db = 0; db < 4; ++db) {
                uint2 pk;
                pk.x = pack_bf2(o[db][qb][0] * inv, o[db][qb][1] * inv);
                pk.y = pack_bf2(o[db][qb][2] * inv, o[db][qb][3] * inv);
                *reinterpret_cast<uint2*>(op + 16 * db + 4 * g) = pk;
            }
            if (g == 0) a.lse[((long)bi * a.h + hi) * a.nq + i] = (m_run[qb] + log2f(l)) * LN2;
        }
    }
}

// ==========================================================================================================
// delta[b,h,i] = sum_d o * dO      grid: ceil(b*nq*h / 256) threads, one (b,i,h) per thread
// ==========================================================================================================
__global__ void attn_delta_kernel(AttnArgs a, float* delta) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)a.b * a.nq * a.h;
    if (idx >= total) return;
    const int hi = idx % a.h;
    const long bi_i = idx / a.h;
    const int i = bi_i % a.nq, bi = bi_i / a.nq;
    const bf16_t* op = a.o + bi * a.o_bs + (long)i * a.o_ns + hi * a.o_hs;
    const bf16_t* dp = a.d_o + bi * a.o_bs + (long)i * a.o_ns + hi * a.o_hs;
    float acc = 0.f;
#pragma unroll
    for (int ch = 0; ch < 8; ++ch) {
        uint4 x = *reinterpret_cast<const uint4*>(op + ch * 8), y = *reinterpret_cast<const uint4*>(dp + ch * 8);
        const uint32_t xs[4] = {x.x, x.y, x.z, x.w}, ys[4] = {y.x, y.y, y.z, y.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            acc += bf2f(xs[e] & 0xffff) * bf2f(ys[e] & 0xffff);
            acc += bf2f(xs[e] >> 16) * bf2f(ys[e] >> 16);
        }
    }
    delta[((long)bi * a.h + hi) * a.nq + i] = acc;
}

// ==========================================================================================================
// dQ (+ d slope): same decomposition as the forward
// ==========================================================================================================
// d slope_h = sum_ij dS_ij * (-|j - i - off|).  delta is computed from the bf16-rounded O, so each row's dS carries a
// common error -P_ij * eps_i; since sum_j dS_ij must be 0, the measured row sum r_i = -eps_i gives the exact
// correction  + r_i * sum_j P_ij |d_ij|  (otherwise the error is amplified by the mean attended distance).
template <int MODE, bool SLOPE_GRAD>
__device__ __forceinline__ void dq_scores(int u, f32x4 (&s)[2][2], const f32x4 (&dp)[2][2], const float (&l2)[2], const float (&dl)[2],
                                          const float (&i_f)[2], float c1, float slope2, float j0f, int g, const uint8_t* m_tile,
                                          bool causal, float (&acc_d)[2], float (&acc_r)[2], float (&acc_p)[2]) {
    uint32_t mbits[2] = {0, 0};
    if (MODE == T_GEN) {
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) mbits[k2] = *reinterpret_cast<const uint32_t*>(m_tile + 16 * (2 * u + k2) + 4 * g);
    }
    const float sj0 = slope2 * (j0f + (float)(4 * g));
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const float uo = MODE == T_LEFT ? -slope2 * i_f[qb] : (MODE == T_RIGHT ? slope2 * i_f[qb] : 0.f);
        const float mm = l2[qb] - uo;
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float jc = (float)(16 * (2 * u + k2) + r);
                const float jf = j0f + (float)(4 * g) + jc;
                const bool ok = MODE != T_GEN || ((((mbits[k2] >> (8 * r)) & 0xff) != 0) && (!causal || jf <= i_f[qb]));
                const float t = score<MODE>(s[k2][qb][r], c1, slope2, sj0 + slope2 * jc, jf, i_f[qb], ok);
                const float p = fast_exp2(t - mm);
                const float ds = p * (dp[k2][qb][r] - dl[qb]);
                if (SLOPE_GRAD) {
                    const float ad = fabsf(jf - i_f[qb]);
                    acc_d[qb] = fmaf(ds, ad, acc_d[qb]); acc_r[qb] += ds; acc_p[qb] = fmaf(p, ad, acc_p[qb]);
                }
                s[k2][qb][r] = ds;
            }
    }
}

__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(AttnArgs a) {
    __shared__ __attribute__((aligned(16))) char smem[3 * 8192 + 64 + 16];
    char* k_tile = smem;              // "a": A operand of S^T
    char* kt_tile = smem + 8192;      // "t": K^T A operand of dQ^T
    char* v_tile = smem + 16384;      // "a": A operand of dP^T = V dO^T
    uint8_t* m_tile = reinterpret_cast<uint8_t*>(smem + 24576);
    int* full_flag = reinterpret_cast<int*>(smem + 24576 + 64);

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g = lane >> 4, c = lane & 15;
    const int bi = blockIdx.z, hi = blockIdx.y, q0 = qtile_of(blockIdx.x, gridDim.x, a.causal) * 128;
    const int kh = (a.kvh == 1) ? 0 : hi;
    const int off = a.nk - a.nq;
    const bf16_t* qp = a.q + bi * a.q_bs + hi * a.q_hs;
    const bf16_t* dop = a.d_o + bi * a.o_bs + hi * a.o_hs;
    const bf16_t* kp = a.k + bi * a.k_bs + kh * a.k_hs;
    const bf16_t* vp = a.v + bi * a.v_bs + kh * a.v_hs;
    const uint8_t* mp = a.kmask ? a.kmask + (long)bi * a.nk : nullptr;
    const float slope2 = a.slopes ? a.slopes[hi] * LOG2E : 0.f;
    const float c1 = a.scale * LOG2E;
    const bool slope_grad = a.dslope != nullptr;

    bf16x8 qf[2][2], dof[2][2];
    float l2[2], dl[2], i_f[2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const int i = q0 + 32 * w + 16 * qb + c;
        i_f[qb] = (float)(i + off);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            qf[qb][ks] = load_row_frag(qp, a.q_ns, i, a.nq, ks, lane);
            dof[qb][ks] = load_row_frag(dop, a.o_ns, i, a.nq, ks, lane);
        }
        const long si = ((long)bi * a.h + hi) * a.nq + i;
        // rows outside the problem, and rows whose keys were ALL masked (lse ~ -1e38: degenerate uniform attention whose
        // output the caller zeroes), get p = 0
        const float lse_i = i < a.nq ? a.lse[si] : NEG_FILL;
        l2[qb] = lse_i > -1e37f ? lse_i * LOG2E : 1e30f;
        dl[qb] = i < a.nq ? a.delta[si] : 0.f;
    }
    const int i_lo = q0 + 32 * w + off, i_hi = i_lo + 31;
    f32x4 dq[4][2];
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) dq[db][qb] = f32x4{0.f, 0.f, 0.f, 0.f};
    float acc_d[2] = {0.f, 0.f}, acc_r[2] = {0.f, 0.f}, acc_p[2] = {0.f, 0.f};

    int nt = (a.nk + 63) / 64;
    if (a.causal) {
        const int last = q0 + 127 + off;
        nt = last < 0 ? 0 : min(nt, last / 64 + 1);
    }
    TileRegs kr, vr;
    uint8_t mreg = 1;
    if (nt > 0) {
        kr.load(kp, a.k_ns, 0, a.nk, tid);
        vr.load(vp, a.v_ns, 0, a.nk, tid);
        if (tid < 64) mreg = (tid < a.nk) ? (mp ? mp[tid] : 1) : 0;
    }
    for (int t = 0; t < nt; ++t) {
        const int j0 = t * 64;
        __syncthreads();
        kr.store<false>(k_tile, tid);
        kr.store<true>(kt_tile, tid);
        vr.store<false>(v_tile, tid);
        if (tid < 64) {
            m_tile[tid] = mreg;
            const unsigned long long all = __ballot(mreg != 0);
            if (tid == 0) *full_flag = (all == ~0ull) ? 1 : 0;
        }
        __syncthreads();
        if (t + 1 < nt) {
            kr.load(kp, a.k_ns, j0 + 64, a.nk, tid);
            vr.load(vp, a.v_ns, j0 + 64, a.nk, tid);
            if (tid < 64) { const int j = j0 + 64 + tid; mreg = (j < a.nk) ? (mp ? mp[j] : 1) : 0; }
        }
        // rows beyond nq hold zero fragments and are never written; the last q block may straddle nq: use the general path
        const bool rows_ok = q0 + 32 * w + 31 < a.nq;
        int cls = classify(j0, i_lo, i_hi, *full_flag != 0, a.causal != 0);
        if (cls == T_SKIP) continue;
        if (!rows_ok && slope_grad) cls = T_GEN;

        const float j0f = (float)j0;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            f32x4 s[2][2], dp[2][2];
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
                const int kb = 2 * u + k2;
                bf16x8 kf0 = frag_rows(k_tile, 16 * kb, 0, lane), kf1 = frag_rows(k_tile, 16 * kb, 1, lane);
                bf16x8 vf0 = frag_rows(v_tile, 16 * kb, 0, lane), vf1 = frag_rows(v_tile, 16 * kb, 1, lane);
#pragma unroll
                for (int qb = 0; qb < 2; ++qb) {
                    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf0, qf[qb][0], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf1, qf[qb][1], acc, 0, 0, 0);
                    s[k2][qb] = acc;
                    f32x4 acc2 = f32x4{0.f, 0.f, 0.f, 0.f};
                    acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf0, dof[qb][0], acc2, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf1, dof[qb][1], acc2, 0, 0, 0);
                    dp[k2][qb] = acc2;
                }
            }
            if (slope_grad) {
                if (cls == T_LEFT) dq_scores<T_LEFT, true>(u, s, dp, l2, dl, i_f, c1, slope2, j0f, g, m_tile, a.causal, acc_d, acc_r, acc_p);
                else if (cls == T_RIGHT) dq_scores<T_RIGHT, true>(u, s, dp, l2, dl, i_f, c1, slope2, j0f, g, m_tile, a.causal, acc_d, acc_r, acc_p);
                else dq_scores<T_GEN, true>(u, s, dp, l2, dl, i_f, c1, slope2, j0f, g, m_tile, a.causal, acc_d, acc_r, acc_p);
            } else {
                if (cls == T_LEFT) dq_scores<T_LEFT, false>(u, s, dp, l2, dl, i_f, c1, slope2, j0f, g, m_tile, a.causal, acc_d, acc_r, acc_p);
                else if (cls == T_RIGHT) dq_scores<T_RIGHT, false>(u, s, dp, l2, dl, i_f, c1, slope2, j0f, g, m_tile, a.causal, acc_d, acc_r, acc_p);
                else dq_scores<T_GEN, false>(u, s, dp, l2, dl, i_f, c1, slope2, j0f, g, m_tile, a.causal, acc_d, acc_r, acc_p);
            }
            bf16x8 dsf[2];
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) dsf[qb] = pack8(s[0][qb], s[1][qb]);
#pragma unroll
            for (int db = 0; db < 4; ++db) {
                bf16x8 ktf = frag_cols_t(kt_tile, 16 * db, u, lane);
#pragma unroll
                for (int qb = 0; qb < 2; ++qb)
                    dq[db][qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ktf, dsf[qb], dq[db][qb], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const int i = q0 + 32 * w + 16 * qb + c;
        if (i < a.nq) {
            bf16_t* p = a.dq + bi * a.dq_bs + (long)i * a.dq_ns + hi * a.dq_hs;
#pragma unroll
            for (int db = 0; db < 4; ++db) {
                uint2 pk;
                pk.x = pack_bf2(dq[db][qb][0] * a.scale, dq[db][qb][1] * a.scale);
                pk.y = pack_bf2(dq[db][qb][2] * a.scale, dq[db][qb][3] * a.scale);
                *reinterpret_cast<uint2*>(p + 16 * db + 4 * g) = pk;
            }
        }
    }
    if (slope_grad) {
        float dslope = 0.f;
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            const int i = q0 + 32 * w + 16 * qb + c;
            const float d = group_sum(acc_d[qb]), r = group_sum(acc_r[qb]), pa = group_sum(acc_p[qb]);
            if (g == 0 && i < a.nq) dslope += -d + r * pa;
        }
        dslope = wave_sum(dslope);
        if (lane == 0) atomicAdd(a.dslope + hi, dslope);
    }
}

// ==========================================================================================================
// dK, dV: grid (ceil(nk/64), kvh, b); wave w owns keys j0 + 16w .. +15; loops over the heads sharing this K/V head
// ==========================================================================================================
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_kernel(AttnArgs a) {
    __shared__ __attribute__((aligned(16))) char smem[4 * 8192 + 512];
    char* q_tile = smem;               // "a": A operand of S
    char* qt_tile = smem + 8192;       // "t": Q^T A operand of dK^T
    char* do_tile = smem + 16384;      // "a": A operand of dP
    char* dot_tile = smem + 24576;     // "t": dO^T A operand of dV^T
    float* nl2_s = reinterpret_cast<float*>(smem + 32768);   // -lse * log2e per row (+inf-safe: 1e30 for rows >= nq)
    float* dl_s = nl2_s + 64;

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g = lane >> 4, c = lane & 15;
    const int bi = blockIdx.z, kh = blockIdx.y, j0 = blockIdx.x * 64;
    const int off = a.nk - a.nq;
    const int heads_per_kv = a.h / a.kvh;
    const bf16_t* kp = a.k + bi * a.k_bs + kh * a.k_hs;
    const bf16_t* vp = a.v + bi * a.v_bs + kh * a.v_hs;
    const int j = j0 + 16 * w + c;  // this lane's key column
    const bool key_ok = (j < a.nk) && (a.kmask ? a.kmask[(long)bi * a.nk + j] != 0 : true);
    const float jf = (float)j;
    const float c1 = a.scale * LOG2E;
    // wave-uniform: are all 16 keys of this wave valid?
    const bool keys_full = __all(key_ok);
    const int jw_lo = j0 + 16 * w, jw_hi = jw_lo + 15;

    bf16x8 kf[2], vf[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        kf[ks] = load_row_frag(kp, a.k_ns, j, a.nk, ks, lane);
        vf[ks] = load_row_frag(vp, a.v_ns, j, a.nk, ks, lane);
    }
    f32x4 dk[4], dv[4];
#pragma unroll
    for (int db = 0; db < 4; ++db) { dk[db] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[db] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    const int nqt = (a.nq + 63) / 64;
    int t_first = 0;
    if (a.causal) {  // query i sees key j iff j <= i + off  ->  first useful row i = j0 - off
        const int i_min = j0 - off;
        t_first = i_min <= 0 ? 0 : i_min / 64;
    }
    const int n_iter = (nqt - t_first) > 0 ? (nqt - t_first) * heads_per_kv : 0;

    TileRegs qr, dor;
    float lreg = 0.f, dreg = 0.f;
    auto issue = [&](int it) {
        const int hh = kh * heads_per_kv + it / (nqt - t_first);
        const int i0 = (t_first + it % (nqt - t_first)) * 64;
        qr.load(a.q + bi * a.q_bs + hh * a.q_hs, a.q_ns, i0, a.nq, tid);
        dor.load(a.d_o + bi * a.o_bs + hh * a.o_hs, a.o_ns, i0, a.nq, tid);
        if (tid < 64) {
            const int i = i0 + tid;
            const long si = ((long)bi * a.h + hh) * a.nq + i;
            const float lse_i = i < a.nq ? a.lse[si] : NEG_FILL;
            lreg = lse_i > -1e37f ? -lse_i * LOG2E : NEG_FILL;   // rows beyond nq / fully masked rows: p = exp2(t + NEG) = 0
            dreg = i < a.nq ? a.delta[si] : 0.f;
        }
    };
    if (n_iter > 0) issue(0);
    for (int it = 0; it < n_iter; ++it) {
        const int hh = kh * heads_per_kv + it / (nqt - t_first);
        const int i0 = (t_first + it % (nqt - t_first)) * 64;
        const float slope2 = a.slopes ? a.slopes[hh] * LOG2E : 0.f;
        __syncthreads();
        qr.store<false>(q_tile, tid);
        qr.store<true>(qt_tile, tid);
        dor.store<false>(do_tile, tid);
        dor.store<true>(dot_tile, tid);
        if (tid < 64) { nl2_s[tid] = lreg; dl_s[tid] = dreg; }
        __syncthreads();
        if (it + 1 < n_iter) issue(it + 1);

        // tile class of this wave's 16 keys against the 64 rows (key coordinates i + off)
        const int r_lo = i0 + off, r_hi = r_lo + 63;
        int cls = T_GEN;
        if (a.causal && jw_lo > r_hi) cls = T_SKIP;
        else if (keys_full && jw_hi <= r_lo) cls = T_LEFT;              // j - i <= 0 everywhere
        else if (keys_full && !a.causal && jw_lo >= r_hi) cls = T_RIGHT;
        if (cls == T_SKIP) continue;

        f32x4 p[4], ds[4];
#pragma unroll
        for (int qb = 0; qb < 4; ++qb) {
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f}, acc2 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows(q_tile, 16 * qb, ks, lane), kf[ks], acc, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows(do_tile, 16 * qb, ks, lane), vf[ks], acc2, 0, 0, 0);
            }
            const f32x4 n4 = *reinterpret_cast<const f32x4*>(nl2_s + 16 * qb + 4 * g);
            const f32x4 d4 = *reinterpret_cast<const f32x4*>(dl_s + 16 * qb + 4 * g);
            const float ib = (float)(i0 + off + 16 * qb + 4 * g);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float i_f = ib + (float)r;
                float e;   // log2-domain score minus lse
                if (cls == T_LEFT) e = fmaf(acc[r], c1, fmaf(slope2, jf - i_f, n4[r]));
                else if (cls == T_RIGHT) e = fmaf(acc[r], c1, fmaf(-slope2, jf - i_f, n4[r]));
                else {
                    const bool ok = key_ok && (!a.causal || jf <= i_f);
                    const float t = fmaf(-slope2, fabsf(jf - i_f), acc[r] * c1);
                    e = (ok ? t : NEG_FILL) + n4[r];
                }
                const float pv = fast_exp2(e);
                p[qb][r] = pv;
                ds[qb][r] = pv * (acc2[r] - d4[r]);
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            bf16x8 pf = pack8(p[2 * u], p[2 * u + 1]);
            bf16x8 dsf = pack8(ds[2 * u], ds[2 * u + 1]);
#pragma unroll
            for (int db = 0; db < 4; ++db) {
                dv[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_cols_t(dot_tile, 16 * db, u, lane), pf, dv[db], 0, 0, 0);
                dk[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_cols_t(qt_tile, 16 * db, u, lane), dsf, dk[db], 0, 0, 0);
            }
        }
    }
    if (j < a.nk) {
        bf16_t* pk_ = a.dk + bi * a.dk_bs + (long)j * a.dk_ns + kh * a.dk_hs;
        bf16_t* pv_ = a.dv + bi * a.dv_bs + (long)j * a.dv_ns + kh * a.dv_hs;
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            uint2 x, y;
            x.x = pack_bf2(dk[db][0] * a.scale, dk[db][1] * a.scale); x.y = pack_bf2(dk[db][2] * a.scale, dk[db][3] * a.scale);
            y.x = pack_bf2(dv[db][0], dv[db][1]); y.y = pack_bf2(dv[db][2], dv[db][3]);
            *reinterpret_cast<uint2*>(pk_ + 16 * db + 4 * g) = x;
            *reinterpret_cast<uint2*>(pv_ + 16 * db + 4 * g) = y;
        }
    }
}

int check_common(const AttnArgs& a) {
    SPN_REQUIRE(a.q && a.k && a.v, "spn_attn: null q/k/v");
    SPN_REQUIRE(a.b > 0 && a.h > 0 && a.nq > 0 && a.nk > 0, "spn_attn: empty problem");
    SPN_REQUIRE(a.kvh == 1 || a.kvh == a.h, "spn_attn: kv heads must be 1 (MQA) or h");
    SPN_REQUIRE((a.q_ns % 8) == 0 && (a.k_ns % 8) == 0 && (a.v_ns % 8) == 0 && (a.q_hs % 8) == 0 &&
                (a.k_hs % 8) == 0 && (a.v_hs % 8) == 0 && (a.q_bs % 8) == 0 && (a.k_bs % 8) == 0 && (a.v_bs % 8) == 0,
                "spn_attn: q/k/v strides must be multiples of 8 elements");
    SPN_REQUIRE((((uintptr_t)a.q | (uintptr_t)a.k | (uintptr_t)a.v) & 15) == 0, "spn_attn: q/k/v must be 16-byte aligned");
    return SPN_OK;
}

}  // namespace

// strides: 12 longs = {q_bs,q_ns,q_hs, k_bs,k_ns,k_hs, v_bs,v_ns,v_hs, o_bs,o_ns,o_hs} in elements; head dim 64.
extern "C" int spn_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse, const uint8_t* kmask,
                            const float* slopes, int b, int h, int kvh, int nq, int nk, int causal, float scale,
                            const long* strides, hipStream_t stream) {
    AttnArgs a;
    memset(&a, 0, sizeof(a));
    a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.o = (bf16_t*)o; a.lse = lse;
    a.kmask = kmask; a.slopes = slopes; a.b = b; a.h = h; a.kvh = kvh; a.nq = nq; a.nk = nk; a.causal = causal;
    a.scale = scale;
    a.q_bs = strides[0]; a.q_ns = strides[1]; a.q_hs = strides[2];
    a.k_bs = strides[3]; a.k_ns = strides[4]; a.k_hs = strides[5];
    a.v_bs = strides[6]; a.v_ns = strides[7]; a.v_hs = strides[8];
    a.o_bs = strides[9]; a.o_ns = strides[10]; a.o_hs = strides[11];
    int rc = check_common(a);
    if (rc) return rc;
    SPN_REQUIRE(o && lse, "spn_attn_fwd: null output");
    SPN_REQUIRE((a.o_ns % 4) == 0 && (a.o_hs % 4) == 0 && (a.o_bs % 4) == 0, "spn_attn_fwd: o strides must be multiples of 4");
    dim3 grid(cdiv(nq, 128), h, b);
    hipLaunchKernelGGL(attn_fwd_kernel, grid, dim3(256), 0, stream, a);
    SPN_LAUNCH_CHECK();
    return SPN_OK;
}

// strides: 21 longs = forward's 12 followed by {dq_bs,dq_ns,dq_hs, dk_bs,dk_ns,dk_hs, dv_bs,dv_ns,dv_hs}.
// delta: workspace of b*h*nq floats.  dslope: [h] fp32, accumulated with atomics (zero it first), or null.
extern "C" int spn_attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* d_o,
                            const float* lse, float* delta, void* dq, void* dk, void* dv, float* dslope,
                            const uint8_t* kmask, const float* slopes, int b, int h, int kvh, int nq, int nk,
                            int causal, float scale, const long* strides, hipStream_t stream) {
    AttnArgs a;
    memset(&a, 0, sizeof(a));
    a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.o = (bf16_t*)o; a.d_o = (const bf16_t*)d_o;
    a.lse = const_cast<float*>(lse); a.delta = delta; a.dq = (bf16_t*)dq; a.dk = (bf16_t*)dk; a.dv = (bf16_t*)dv;
    a.dslope = dslope; a.kmask = kmask; a.slopes = slopes; a.b = b; a.h = h; a.kvh = kvh; a.nq = nq; a.nk = nk;
    a.causal = causal; a.scale = scale;
    a.q_bs = strides[0]; a.q_ns = strides[1]; a.q_hs = strides[2];
    a.k_bs = strides[3]; a.k_ns = strides[4]; a.k_hs = strides[5];
    a.v_bs = strides[6]; a.v_ns = strides[7]; a.v_hs = strides[8];
    a.o_bs = strides[9]; a.o_ns = strides[10]; a.o_hs = strides[11];
    a.dq_bs = strides[12]; a.dq_ns = strides[13]; a.dq_hs = strides[14];
    a.dk_bs = strides[15]; a.dk_ns = strides[16]; a.dk_hs = strides[17];
    a.dv_bs = strides[18]; a.dv_ns = strides[19]; a.dv_hs = strides[20];
    int rc = check_common(a);
    if (rc) return rc;
    SPN_REQUIRE(o && d_o && lse && delta && dq && dk && dv, "spn_attn_bwd: null tensor");
    SPN_REQUIRE((a.o_ns % 8) == 0 && (a.o_hs % 8) == 0 && (a.o_bs % 8) == 0, "spn_attn_bwd: o/dO strides must be multiples of 8");
    SPN_REQUIRE((((uintptr_t)o | (uintptr_t)d_o) & 15) == 0, "spn_attn_bwd: o/dO must be 16-byte aligned");
    const long total = (long)b * nq * h;
    hipLaunchKernelGGL(attn_delta_kernel, dim3(cdiv(total, 256)), dim3(256), 0, stream, a, delta);
    hipLaunchKernelGGL(attn_bwd_dkv_kernel, dim3(cdiv(nk, 64), kvh, b), dim3(256), 0, stream, a);
    hipLaunchKernelGGL(attn_bwd_dq_kernel, dim3(cdiv(nq, 128), h, b), dim3(256), 0, stream, a);
    SPN_LAUNCH_CHECK();
    return SPN_OK;
}
