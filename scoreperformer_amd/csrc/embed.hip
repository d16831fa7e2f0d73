// Tuple-token embedding kernels for gfx950.
//
// K1  spn_table_build_{fwd,bwd}: all per-key tables of one TupleTokenEmbeddings in ONE launch.
//     `DiscreteDenseContinuousEmbedding.weight` (modules/transformer/embeddings.py:118-152,202-213):
//       rows in `discrete_ids`  <- index_weight rows
//       every other row         <- Linear(E,E)(Mish(Linear(1,E)(token_value)))      (dense)  or  token_value * w (plain)
//     (with `discrete=True` index_weight is added on every row).  The reference rebuilds these ~60 tiny ops per
//     use and 5 uses per step; here one launch per step feeds the 3 encoders/decoder gathers and the tied LM head.
// K2  spn_embed_fwd / spn_embed_bwd: gather the K per-key rows of each note tuple, concatenate to [T, sum E] and
//     LayerNorm (models/scoreperformer/embeddings.py:121-143), output bf16 for the projection GEMM.  Backward
//     recomputes the gather, applies the LayerNorm backward and scatter-adds into the table gradients through an
//     LDS-privatised copy of the (tiny, heavily contended) table: ds_add_f32 per element, one global atomic per
//     table element per block.  Rows with token == padding_idx receive no gradient (F.embedding semantics).
#include "common.h"
#include "tuning.h"

namespace {

constexpr int MAXK = 16;

struct TableKey {
    const float* tv;   // [V] token values
    const float* w0;   // [E]   Linear(1,E).weight  (dense)  or value_layer.weight (plain)
    const float* b0;   // [E]   (dense only)
    const float* w1;   // [E,E] (dense only)
    const float* b1;   // [E]   (dense only)
    const float* iw;   // [V,E] index_weight or null
    float* out;        // [V,E]
    float* h1;         // [V,E] workspace: Mish output (dense only)
    int V, E, row0, dense, discrete;
};
struct TableDesc {
    TableKey k[MAXK];
    int nkeys;
    unsigned ids_mask;  // bit i set <=> i in discrete_ids
};
struct TableGrad {
    const float* dout[MAXK];  // [V,E]
    float* dw0[MAXK]; float* db0[MAXK]; float* db1[MAXK]; float* diw[MAXK];
    float* dval[MAXK];        // [V,E] workspace (rows in discrete_ids zeroed) for the dW1 GEMM
};

__device__ __forceinline__ float mish_f(float v) {
    const float sp = v > 20.f ? v : log1pf(__expf(v));
    return v * tanhf(sp);
}
__device__ __forceinline__ float mish_g(float v) {
    const float sp = v > 20.f ? v : log1pf(__expf(v));
    const float th = tanhf(sp);
    const float sg = 1.f / (1.f + __expf(-v));
    return th + v * (1.f - th * th) * sg;
}

__device__ __forceinline__ int find_key(const TableDesc& d, int row) {
    int kk = 0;
    for (int i = 1; i < d.nkeys; ++i) if (row >= d.k[i].row0) kk = i;
    return kk;
}

__global__ __launch_bounds__(128) void table_fwd_kernel(TableDesc d) {
    __shared__ float hs[512];
    const int kk = find_key(d, blockIdx.x);
    const TableKey& K = d.k[kk];
    const int v = blockIdx.x - K.row0, E = K.E;
    const float tv = K.tv[v];
    const bool is_id = v < 32 && ((d.ids_mask >> v) & 1u);
    if (K.dense) {
        for (int e = threadIdx.x; e < E; e += blockDim.x) {
            const float h = mish_f(tv * K.w0[e] + K.b0[e]);
            hs[e] = h;
            K.h1[(long)v * E + e] = h;
        }
        __syncthreads();
    }
    for (int e = threadIdx.x; e < E; e += blockDim.x) {
        float val;
        if (K.dense) {
            val = K.b1[e];
            const float* wr = K.w1 + (long)e * E;
            for (int j = 0; j < E; ++j) val = fmaf(wr[j], hs[j], val);
        } else {
            val = tv * K.w0[e];
        }
        if (is_id) val = 0.f;
        float tw = 0.f;
        if (K.iw && (K.discrete || is_id)) tw = K.iw[(long)v * E + e];
        K.out[(long)v * E + e] = tw + val;
    }
}

__global__ __launch_bounds__(128) void table_bwd_kernel(TableDesc d, TableGrad g) {
    __shared__ float dv[512];
    const int kk = find_key(d, blockIdx.x);
    const TableKey& K = d.k[kk];
    const int v = blockIdx.x - K.row0, E = K.E;
    const float tv = K.tv[v];
    const bool is_id = v < 32 && ((d.ids_mask >> v) & 1u);
    for (int e = threadIdx.x; e < E; e += blockDim.x) {
        const float go = g.dout[kk][(long)v * E + e];
        const float val_g = is_id ? 0.f : go;
        dv[e] = val_g;
        if (g.diw[kk]) g.diw[kk][(long)v * E + e] = (K.discrete || is_id) ? go : 0.f;
        if (K.dense) {
            g.dval[kk][(long)v * E + e] = val_g;
            if (val_g != 0.f) atomicAdd(g.db1[kk] + e, val_g);
        } else if (val_g != 0.f) {
            atomicAdd(g.dw0[kk] + e, val_g * tv);
        }
    }
    if (!K.dense || is_id) return;
    __syncthreads();
    for (int j = threadIdx.x; j < E; j += blockDim.x) {
        float dh = 0.f;
        for (int e = 0; e < E; ++e) dh = fmaf(K.w1[(long)e * E + j], dv[e], dh);
        const float dpre = dh * mish_g(tv * K.w0[j] + K.b0[j]);
        atomicAdd(g.dw0[kk] + j, dpre * tv);
        atomicAdd(g.db0[kk] + j, dpre);
    }
}

// ---------------------------------------------------------------------------------------------------------
struct GatherDesc {
    const float* table[MAXK];
    float* dtable[MAXK];
    int width[MAXK], col0[MAXK], rows[MAXK];
    int nkeys, D;
};

// One wave per ROW PAIR: the gathers of both rows (2 x NV table-row pieces per lane, from L2) are issued before either row is
// reduced -- the kernel is bound by the latency of those dependent gathers, not by bandwidth.
template <int NV>
__global__ __launch_bounds__(256) void embed_fwd_kernel(GatherDesc d, const long* __restrict__ tokens, long tok_bs, long tok_ts, int t_len,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        bf16_t* __restrict__ y, long ldy, float* __restrict__ mean,
                                                        float* __restrict__ rstd, int T, float eps) {
    constexpr int RPW = 2;
    const int lane = threadIdx.x & 63;
    const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW;
    if (row0 >= T) return;
    f32x4 v[RPW][NV];
    int tok_l[RPW];
    // a row's token tuple: one load by the first nkeys lanes, broadcast (instead of a token load in front of every table load)
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int row = min(row0 + r, T - 1);
        tok_l[r] = lane < d.nkeys ? (int)tokens[(long)(row / t_len) * tok_bs + (long)(row % t_len) * tok_ts + lane] : 0;
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int col = (lane + 64 * i) * 4;
        int kk = 0;
        for (int q = 1; q < d.nkeys; ++q) if (col >= d.col0[q]) kk = q;
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const long tok = __shfl(tok_l[r], kk, 64);
            v[r][i] = col < d.D ? *reinterpret_cast<const f32x4*>(d.table[kk] + tok * d.width[kk] + (col - d.col0[kk])) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int row = row0 + r;
        if (row >= T) break;
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) sum += v[r][i][0] + v[r][i][1] + v[r][i][2] + v[r][i][3];
        float mu = 0.f, rs = 1.f;
        if (gamma) {
            mu = wave_sum(sum) / (float)d.D;
            float sq = 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int col = (lane + 64 * i) * 4;
                if (col < d.D) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const float t = v[r][i][e] - mu; sq += t * t; }
                }
            }
            rs = rsqrtf(wave_sum(sq) / (float)d.D + eps);
            if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int col = (lane + 64 * i) * 4;
            if (col >= d.D) continue;
            f32x4 o = v[r][i];
            if (gamma) {
                const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + col);
                const f32x4 be = *reinterpret_cast<const f32x4*>(beta + col);
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (v[r][i][e] - mu) * rs * ga[e] + be[e];
            }
            uint2 pk; pk.x = pack_bf2(o[0], o[1]); pk.y = pack_bf2(o[2], o[3]);
            *reinterpret_cast<uint2*>(y + (long)row * ldy + col) = pk;
        }
    }
}

// The same with EIGHT columns per lane (all key widths multiples of 8): the output leaves as 16-byte pieces.  With 8-byte stores the
// kernel is bound by store ISSUE, not by bandwidth (402 MB of bf16 rows at C3 through 8-byte-per-lane instructions: 172 us).
template <int NW>
__global__ __launch_bounds__(256) void embed_fwd_wide_kernel(GatherDesc d, const long* __restrict__ tokens, long tok_bs, long tok_ts, int t_len,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             bf16_t* __restrict__ y, long ldy, float* __restrict__ mean,
                                                             float* __restrict__ rstd, int T, float eps) {
    constexpr int RPW = 2;
    const int lane = threadIdx.x & 63;
    const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW;
    if (row0 >= T) return;
    f32x4 v[RPW][NW][2];
    int tok_l[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int row = min(row0 + r, T - 1);
        tok_l[r] = lane < d.nkeys ? (int)tokens[(long)(row / t_len) * tok_bs + (long)(row % t_len) * tok_ts + lane] : 0;
    }
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        const int col = (lane + 64 * i) * 8;
        int kk = 0;
        for (int q = 1; q < d.nkeys; ++q) if (col >= d.col0[q]) kk = q;
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const long tok = __shfl(tok_l[r], kk, 64);
            const float* src = d.table[kk] + tok * d.width[kk] + (col - d.col0[kk]);
            v[r][i][0] = col < d.D ? *reinterpret_cast<const f32x4*>(src) : f32x4{0.f, 0.f, 0.f, 0.f};
            v[r][i][1] = col < d.D ? *reinterpret_cast<const f32x4*>(src + 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int row = row0 + r;
        if (row >= T) break;
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < NW; ++i)
#pragma unroll
            for (int h = 0; h < 2; ++h) sum += v[r][i][h][0] + v[r][i][h][1] + v[r][i][h][2] + v[r][i][h][3];
        float mu = 0.f, rs = 1.f;
        if (gamma) {
            mu = wave_sum(sum) / (float)d.D;
            float sq = 0.f;
#pragma unroll
            for (int i = 0; i < NW; ++i) {
                if ((lane + 64 * i) * 8 < d.D) {
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int e = 0; e < 4; ++e) { const float t = v[r][i][h][e] - mu; sq += t * t; }
                }
            }
            rs = rsqrtf(wave_sum(sq) / (float)d.D + eps);
            if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
        }
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const int col = (lane + 64 * i) * 8;
            if (col >= d.D) continue;
            uint4 pk;
            uint32_t* pw = reinterpret_cast<uint32_t*>(&pk);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                f32x4 o = v[r][i][h];
                if (gamma) {
                    const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + col + 4 * h);
                    const f32x4 be = *reinterpret_cast<const f32x4*>(beta + col + 4 * h);
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (v[r][i][h][e] - mu) * rs * ga[e] + be[e];
                }
                pw[2 * h] = pack_bf2(o[0], o[1]); pw[2 * h + 1] = pack_bf2(o[2], o[3]);
            }
            *reinterpret_cast<uint4*>(y + (long)row * ldy + col) = pk;
        }
    }
}

// backward pass 1: per-row LN-backward sums s1 = mean(g), s2 = mean(g*xhat) with g = dy*gamma, and dgamma/dbeta.
template <int NV>
__global__ __launch_bounds__(256) void embed_bwd_stats_kernel(GatherDesc d, const long* __restrict__ tokens, long tok_bs, long tok_ts, int t_len,
                                                              const bf16_t* __restrict__ dy, long lddy, const float* __restrict__ gamma,
                                                              const float* __restrict__ mean, const float* __restrict__ rstd,
                                                              float* __restrict__ s1o, float* __restrict__ s2o,
                                                              float* __restrict__ dgamma, float* __restrict__ dbeta, int T,
                                                              int rows_per_block) {
    __shared__ float red[4][64 * NV * 4 + 4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    f32x4 pg[NV], pb[NV];
    int key_of[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        pg[i] = f32x4{0.f, 0.f, 0.f, 0.f}; pb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int col = (lane + 64 * i) * 4;
        int kk = 0;
        for (int q = 1; q < d.nkeys; ++q) if (col >= d.col0[q]) kk = q;
        key_of[i] = kk;
    }
    const int row_begin = blockIdx.x * rows_per_block, row_end = min(T, row_begin + rows_per_block);
    // rows that are exactly NV * 256 wide (the step's 12 keys x 128): no column test inside the row loop, so that every load of a row --
    // NV table-row pieces, NV dy pieces -- is in flight before the first wait (behind a per-chunk test each chunk's loads sat in their own
    // basic block and hipcc waited for one chunk before requesting the next: six serial trips per row, 1.7 TB/s); gamma stays in registers
    const bool full = d.D == NV * 256;
    f32x4 gam_r[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) gam_r[i] = full ? *reinterpret_cast<const f32x4*>(gamma + (lane + 64 * i) * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    for (int row = row_begin + w; row < row_end; row += 4) {
        const float mu = mean[row], rs = rstd[row];
        float s1 = 0.f, s2 = 0.f;
        const int tok_l = lane < d.nkeys ? (int)tokens[(long)(row / t_len) * tok_bs + (long)(row % t_len) * tok_ts + lane] : 0;
        if (full) {
            f32x4 xva[NV]; uint2 ua[NV];
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int col = (lane + 64 * i) * 4;
                const int kk = key_of[i];
                const long tok = __shfl(tok_l, kk, 64);
                xva[i] = *reinterpret_cast<const f32x4*>(d.table[kk] + tok * d.width[kk] + (col - d.col0[kk]));
                ua[i] = *reinterpret_cast<const uint2*>(dy + (long)row * lddy + col);
            }
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const float dyv[4] = {bf2f(ua[i].x & 0xffff), bf2f(ua[i].x >> 16), bf2f(ua[i].y & 0xffff), bf2f(ua[i].y >> 16)};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float xh = (xva[i][e] - mu) * rs, gq = dyv[e] * gam_r[i][e];
                    s1 += gq; s2 += gq * xh;
                    pg[i][e] += dyv[e] * xh; pb[i][e] += dyv[e];
                }
            }
            s1 = wave_sum(s1) / (float)d.D;
            s2 = wave_sum(s2) / (float)d.D;
            if (lane == 0) { s1o[row] = s1; s2o[row] = s2; }
            continue;
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int col = (lane + 64 * i) * 4;
            const int kk = key_of[i];
            const long tok = __shfl(tok_l, kk, 64);   // one token-tuple load per row, broadcast
            if (col >= d.D) continue;
            const f32x4 xv = *reinterpret_cast<const f32x4*>(d.table[kk] + tok * d.width[kk] + (col - d.col0[kk]));
            const uint2 u = *reinterpret_cast<const uint2*>(dy + (long)row * lddy + col);
            const float dyv[4] = {bf2f(u.x & 0xffff), bf2f(u.x >> 16), bf2f(u.y & 0xffff), bf2f(u.y >> 16)};
            const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + col);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float xh = (xv[e] - mu) * rs, gq = dyv[e] * ga[e];
                s1 += gq; s2 += gq * xh;
                pg[i][e] += dyv[e] * xh; pb[i][e] += dyv[e];
            }
        }
        s1 = wave_sum(s1) / (float)d.D;
        s2 = wave_sum(s2) / (float)d.D;
        if (lane == 0) { s1o[row] = s1; s2o[row] = s2; }
    }
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) red[w][(lane + 64 * i) * 4 + e] = pass == 0 ? pg[i][e] : pb[i][e];
        __syncthreads();
        for (int col = threadIdx.x; col < d.D; col += 256)
            atomicAdd((pass == 0 ? dgamma : dbeta) + col, red[0][col] + red[1][col] + red[2][col] + red[3][col]);
        __syncthreads();
    }
}

// backward pass 2: grid (chunks, keys).  Each block owns one key's table copy in LDS (if it fits) and a slab of rows.
// A wave takes 64 rows at a time: lane r fetches row r's token and LayerNorm statistics (one latency for 64 rows instead of a
// dependent token -> table-row chain per row), then the rows are walked four at a time with lane-group broadcasts.
// Few fat blocks (one per CU): the final flush is V*E global atomics PER BLOCK, which dominated with ~1000 thin blocks.
template <bool USE_LDS>
__global__ __launch_bounds__(1024) void embed_bwd_scatter_kernel(GatherDesc d, const long* __restrict__ tokens, long tok_bs, long tok_ts, int t_len,
                                                                const bf16_t* __restrict__ dy, long lddy, const float* __restrict__ gamma,
                                                                const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                const float* __restrict__ s1, const float* __restrict__ s2, int T,
                                                                int rows_per_block, int padding_idx) {
    // USE_LDS is a template parameter on purpose: with a run-time choice between the LDS copy and the global table the pointer is
    // generic and every add becomes a flat_atomic_add_f32 through the LDS aperture.  Even as ds_add_f32 the LDS float atomics are
    // what this kernel spends its time on (~120 cycles per wave-instruction: 0.8 of 1.07 ms; bank-conflict-free column orders do
    // not change that).  Tried and slower: one owner wave per token id (mod 16) with plain read-modify-writes -- small vocabularies
    // (16 NotesInOnset ids) serialise a wave on one table row.
    constexpr bool use_lds = USE_LDS;
    extern __shared__ __attribute__((aligned(16))) float acc[];
    const int kk = blockIdx.y, E = d.width[kk], V = d.rows[kk], c0 = d.col0[kk];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (use_lds) {
        for (int i = threadIdx.x; i < V * E; i += 1024) acc[i] = 0.f;
        __syncthreads();
    }
    const float* tab = d.table[kk];
    float* dtab = d.dtable[kk];
    const int row_begin = blockIdx.x * rows_per_block, row_end = min(T, row_begin + rows_per_block);
    // 32 lanes x 4 columns per row, 2 rows per wave step, 4 steps in flight: 8-byte gradient loads instead of a row-per-wave walk
    const int q = lane >> 5, c4 = (lane & 31) * 4;
    for (int base = row_begin + w * 64; base < row_end; base += 16 * 64) {
        const int myrow = base + lane;
        int tok_l = padding_idx;
        float mu_l = 0.f, rs_l = 1.f, a1_l = 0.f, a2_l = 0.f;
        if (myrow < row_end) {
            tok_l = (int)tokens[(long)(myrow / t_len) * tok_bs + (long)(myrow % t_len) * tok_ts + kk];
            if (gamma) { mu_l = mean[myrow]; rs_l = rstd[myrow]; a1_l = s1[myrow]; a2_l = s2[myrow]; }
        }
        const int cnt = min(64, row_end - base);
#pragma unroll 4
        for (int r2 = 0; r2 < cnt; r2 += 2) {
            const int r = r2 + q;
            const int tok = __shfl(tok_l, r, 64);
            const float mu = __shfl(mu_l, r, 64), rs = __shfl(rs_l, r, 64), a1 = __shfl(a1_l, r, 64), a2 = __shfl(a2_l, r, 64);
            const bool live = r < cnt && tok != padding_idx;
            const bf16_t* dyr = dy + (long)(base + (r < cnt ? r : 0)) * lddy + c0;
            const float* tr = tab + (long)(live ? tok : 0) * E;
            for (int c = c4; c < E; c += 128) {
                const uint2 u = *reinterpret_cast<const uint2*>(dyr + c);
                f32x4 g = f32x4{bf2f(u.x & 0xffff), bf2f(u.x >> 16), bf2f(u.y & 0xffff), bf2f(u.y >> 16)};
                if (gamma) {
                    const f32x4 x = *reinterpret_cast<const f32x4*>(tr + c);
                    const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c0 + c);
                    g = rs * (g * ga - a1 - (x - mu) * (rs * a2));
                }
                if (live) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (USE_LDS) atomicAdd(&acc[tok * E + c + e], g[e]);
                        else atomicAdd(dtab + (long)tok * E + c + e, g[e]);
                    }
                }
            }
        }
    }
    if (use_lds) {
        __syncthreads();
        for (int i = threadIdx.x; i < V * E; i += 1024) {
            const float v = acc[i];
            if (v != 0.f) atomicAdd(dtab + i, v);
        }
    }
}

// backward pass 2 on the matrix cores: dTable[V, 128] = OneHot^T[V, T] . dx[T, 128] -- a GEMM whose A operand (one-hot rows of the
// token ids, exact in bf16) is built in registers and whose B operand is the LayerNorm-backward'ed gradient tile, rounded to bf16
// and staged in LDS.  grid (row chunks, keys), 8 waves: wave w owns the vocabulary tiles w and w + 8 (32 ids each, V <= 512) for all
// 128 columns (2 x 4 accumulators of v_mfma_f32_32x32x16_bf16).  A stage = 64 tokens: every thread produces 16 columns of one row
// (16-byte pieces, written in the k-major swizzled layout the transposing fragment read of gemm.hip expects), one barrier per stage
// with two buffers.  Replaces 201 M LDS float atomics per call (1.1 ms) by 0.1 GFLOP-scale MFMA work.
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
__device__ __forceinline__ int em_rc_off(int krow, int chunk) { return krow * 256 + ((chunk ^ ((krow & 3) << 2)) << 4); }
__device__ __forceinline__ bf16x8 em_read_frag_t(const char* lds, int r_base, int ks, int lane) {
    typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
    const int g = lane >> 4, p = lane & 15;
    const int col_byte = (r_base + 16 * (g & 1) + 4 * (p & 3)) * 2;
    const int chunk = col_byte >> 4, within = col_byte & 15;
    const int k0 = ks * 16 + (g >> 1) * 8 + (p >> 2);
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(lds + em_rc_off(k0, chunk) + within));
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(lds + em_rc_off(k0 + 4, chunk) + within));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

__global__ __launch_bounds__(512) void embed_bwd_scatter_mfma_kernel(GatherDesc d, const long* __restrict__ tokens, long tok_bs, long tok_ts,
                                                                     int t_len, const bf16_t* __restrict__ dy, long lddy,
                                                                     const float* __restrict__ gamma, const float* __restrict__ mean,
                                                                     const float* __restrict__ rstd, const float* __restrict__ s1,
                                                                     const float* __restrict__ s2, int T, int rows_per_block, int padding_idx) {
    __shared__ __attribute__((aligned(16))) char stage[2][64 * 256];
    __shared__ __attribute__((aligned(16))) int tok_s[2][64];
    const int kk = blockIdx.y, V = d.rows[kk], c0 = d.col0[kk];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const float* tab = d.table[kk];
    float* dtab = d.dtable[kk];
    const int row_begin = blockIdx.x * rows_per_block, row_end = min(T, row_begin + rows_per_block);
    if (row_begin >= row_end) return;
    const int mtiles = (V + 31) >> 5;
    f32x16 acc[2][4];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][nt][r] = 0.f;
    const int pr = tid >> 3, pc = tid & 7;          // producer role: row of the stage; its two 8-column chunks are pc and pc + 8
    // (8 consecutive lanes cover 128 contiguous bytes of a row: whole-line gradient loads, conflict-free 16-byte LDS writes)
    // the producer is software-pipelined: the raw loads of stage s+1 (gradient pieces, gathered table rows, row statistics) are issued
    // before the MFMAs of stage s and consumed after them; the token of stage s+2 (the gather address) is fetched one stage earlier
    f32x4 gam[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        gam[h][0] = gamma ? *reinterpret_cast<const f32x4*>(gamma + c0 + (pc + 8 * h) * 8) : f32x4{1.f, 1.f, 1.f, 1.f};
        gam[h][1] = gamma ? *reinterpret_cast<const f32x4*>(gamma + c0 + (pc + 8 * h) * 8 + 4) : f32x4{1.f, 1.f, 1.f, 1.f};
    }
    auto load_tok = [&](int base) {
        const int gr = base + pr;
        if (gr >= row_end) return -1;
        const int tok = (int)tokens[(long)(gr / t_len) * tok_bs + (long)(gr % t_len) * tok_ts + kk];
        return (tok == padding_idx || tok < 0 || tok >= V) ? -1 : tok;
    };
    struct Raw { uint4 dyv[2]; f32x4 x[2][2]; float mu, rs, a1, a2; int tok; };
    auto load_raw = [&](int base, int tok) {
        Raw r;
        r.tok = tok; r.mu = 0.f; r.rs = 1.f; r.a1 = 0.f; r.a2 = 0.f;
        const int gr = min(base + pr, row_end - 1);
        const int tk = tok < 0 ? 0 : tok;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int c = (pc + 8 * h) * 8;
            r.dyv[h] = *reinterpret_cast<const uint4*>(dy + (long)gr * lddy + c0 + c);
            if (gamma) {
                r.x[h][0] = *reinterpret_cast<const f32x4*>(tab + (long)tk * 128 + c);
                r.x[h][1] = *reinterpret_cast<const f32x4*>(tab + (long)tk * 128 + c + 4);
            }
        }
        if (gamma) { r.mu = mean[gr]; r.rs = rstd[gr]; r.a1 = s1[gr]; r.a2 = s2[gr]; }
        return r;
    };
    Raw raw = load_raw(row_begin, load_tok(row_begin));
    int tok_next = load_tok(row_begin + 64);
    int buf = 0;
    for (int base = row_begin; base < row_end; base += 64, buf ^= 1) {
        {   // ---- produce: dx of 64 rows x 128 columns of this key, bf16, k-major ----
            if ((tid & 7) == 0) tok_s[buf][pr] = raw.tok;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                uint4 out = uint4{0u, 0u, 0u, 0u};
                if (raw.tok >= 0) {
                    float g[8];
                    unpack8(raw.dyv[h], g);
                    if (gamma) {
                        const float ra2 = raw.rs * raw.a2;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            g[e] = raw.rs * (g[e] * gam[h][0][e] - raw.a1 - (raw.x[h][0][e] - raw.mu) * ra2);
                            g[4 + e] = raw.rs * (g[4 + e] * gam[h][1][e] - raw.a1 - (raw.x[h][1][e] - raw.mu) * ra2);
                        }
                    }
                    out = pack8f(g);
                }
                *reinterpret_cast<uint4*>(stage[buf] + em_rc_off(pr, pc + 8 * h)) = out;
            }
        }
        if (base + 64 < row_end) {   // next stage's loads go out now and land behind the MFMAs below
            raw = load_raw(base + 64, tok_next);
            tok_next = load_tok(base + 128);
        }
        __syncthreads();
        // ---- consume: one-hot fragments of this wave's vocabulary tiles x the four 32-column tiles ----
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int4 t0 = *reinterpret_cast<const int4*>(&tok_s[buf][ks * 16 + (lane >> 5) * 8]);
            const int4 t1 = *reinterpret_cast<const int4*>(&tok_s[buf][ks * 16 + (lane >> 5) * 8 + 4]);
            bf16x8 bf[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) bf[nt] = em_read_frag_t(stage[buf], nt * 32, ks, lane);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                const int mt = w + 8 * mi;
                if (mt >= mtiles) break;
                const int vid = mt * 32 + (lane & 31);
                uint4 oh;
                oh.x = (t0.x == vid ? 0x3F80u : 0u) | (t0.y == vid ? 0x3F800000u : 0u);
                oh.y = (t0.z == vid ? 0x3F80u : 0u) | (t0.w == vid ? 0x3F800000u : 0u);
                oh.z = (t1.x == vid ? 0x3F80u : 0u) | (t1.y == vid ? 0x3F800000u : 0u);
                oh.w = (t1.z == vid ? 0x3F80u : 0u) | (t1.w == vid ? 0x3F800000u : 0u);
                const bf16x8 af = __builtin_bit_cast(bf16x8, oh);
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) acc[mi][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[nt], af, acc[mi][nt], 0, 0, 0);
            }
        }
    }
    // ---- flush: lane = one vocabulary row of the tile, registers 4q + r = columns 8q + 4 (lane >> 5) + r of the 32-column tile ----
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const int mt = w + 8 * mi;
        if (mt >= mtiles) break;
        const int vid = mt * 32 + (lane & 31);
        if (vid >= V) continue;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = acc[mi][nt][4 * q + r];
                    if (v != 0.f) atomicAdd(dtab + (long)vid * 128 + nt * 32 + 8 * q + 4 * (lane >> 5) + r, v);
                }
    }
}

inline int round_nv(int nv) { return nv <= 2 ? nv : nv <= 4 ? 4 : nv <= 6 ? 6 : 8; }

}  // namespace

// Per-key pointer tables are passed as arrays of nkeys entries (host memory; copied into the kernel argument block).
extern "C" int spn_table_build_fwd(int nkeys, const float* const* tv, const float* const* w0, const float* const* b0,
                                   const float* const* w1, const float* const* b1, const float* const* iw, float* const* out,
                                   float* const* h1, const int* V, const int* E, int dense, int discrete, unsigned ids_mask,
                                   hipStream_t stream) {
    SPN_REQUIRE(nkeys > 0 && nkeys <= MAXK, "spn_table_build_fwd: 1..16 keys");
    TableDesc d;
    memset(&d, 0, sizeof(d));
    int rows = 0;
    for (int i = 0; i < nkeys; ++i) {
        SPN_REQUIRE(E[i] <= 512, "spn_table_build_fwd: embedding dim must be <= 512");
        d.k[i] = TableKey{tv[i], w0[i], dense ? b0[i] : nullptr, dense ? w1[i] : nullptr, dense ? b1[i] : nullptr, iw ? iw[i] : nullptr,
                          out[i], dense ? h1[i] : nullptr, V[i], E[i], rows, dense, discrete};
        rows += V[i];
    }
    d.nkeys = nkeys; d.ids_mask = ids_mask;
    hipLaunchKernelGGL(table_fwd_kernel, dim3(rows), dim3(128), 0, stream, d);
    SPN_LAUNCH_CHECK();
    return SPN_OK;
}

// dout: per-key [V,E] gradients of the tables.  dw0/db0/db1 are accumulated with atomics (zero them first);
// diw (may be null per key) and dval (workspace for the dW1 = dval^T.h1 GEMM, dense only) are written.
extern "C" int spn_table_build_bwd(int nkeys, const float* const* tv, const float* const* w0, const float* const* b0,
                                   const float* const* w1, const float* const* dout, float* const* dw0, float* const* db0,
                                   float* const* db1, float* const* diw, float* const* dval, const int* V, const int* E, int dense,
                                   int discrete, unsigned ids_mask, hipStream_t stream) {
    SPN_REQUIRE(nkeys > 0 && nkeys <= MAXK, "spn_table_build_bwd: 1..16 keys");
    TableDesc d;
    TableGrad g;
    memset(&d, 0, sizeof(d));
    memset(&g, 0, sizeof(g));
    int rows = 0;
    for (int i = 0; i < nkeys; ++i) {
        d.k[i] = TableKey{tv[i], w0[i], dense ? b0[i] : nullptr, dense ? w1[i] : nullptr, nullptr, nullptr, nullptr, nullptr, V[i], E[i],
                          rows, dense, discrete};
        g.dout[i] = dout[i]; g.dw0[i] = dw0[i]; g.db0[i] = dense ? db0[i] : nullptr; g.db1[i] = dense ? db1[i] : nullptr;
        g.diw[i] = diw ? diw[i] : nullptr; g.dval[i] = dense ? dval[i] : nullptr;
        rows += V[i];
    }
    d.nkeys = nkeys; d.ids_mask = ids_mask;
    hipLaunchKernelGGL(table_bwd_kernel, dim3(rows), dim3(128), 0, stream, d, g);
    SPN_LAUNCH_CHECK();
    return SPN_OK;
}

static int fill_gather(GatherDesc& d, int nkeys, const float* const* tables, float* const* dtables, const int* V, const int* E) {
    memset(&d, 0, sizeof(d));
    int col = 0;
    for (int i = 0; i < nkeys; ++i) {
        if (E[i] % 4) return -1;
        d.table[i] = tables[i]; d.dtable[i] = dtables ? dtables[i] : nullptr;
        d.width[i] = E[i]; d.col0[i] = col; d.rows[i] = V[i];
        col += E[i];
    }
    d.nkeys = nkeys; d.D = col;
    return col;
}

// tokens: int64 [B, t_len, >= nkeys] view with element strides (tok_bs, tok_ts); T = B * t_len rows.  y: bf16 [T, sum E].  gamma == null: plain concatenation.
extern "C" int spn_embed_fwd(int nkeys, const float* const* tables, const int* V, const int* E, const long* tokens, long tok_bs, long tok_ts, int t_len,
                             const float* gamma, const float* beta, void* y, long ldy, float* mean, float* rstd, int T, float eps,
                             hipStream_t stream) {
    SPN_REQUIRE(nkeys > 0 && nkeys <= MAXK && tokens && y && T > 0, "spn_embed_fwd: bad arguments");
    GatherDesc d;
    const int D = fill_gather(d, nkeys, tables, nullptr, V, E);
    SPN_REQUIRE(D > 0 && D <= 2048 && ldy % 4 == 0, "spn_embed_fwd: widths must be multiples of 4, total <= 2048");
    SPN_REQUIRE(!gamma || (mean && rstd && beta), "spn_embed_fwd: mean/rstd/beta required with gamma");
    const int nv = round_nv((D + 255) / 256);
    dim3 grid(cdiv(T, 8));   // 4 waves x 2 rows
    bool wide = D % 8 == 0 && ldy % 8 == 0 && D <= 2048 && (reinterpret_cast<uintptr_t>(y) & 15) == 0;
    for (int i = 0; i < nkeys; ++i) wide = wide && E[i] % 8 == 0;
    if (wide) {   // 16-byte output pieces: 8 columns per lane, (D + 511) / 512 chunks
        const int nw = (D + 511) / 512;
#define CASEW(NW_) case NW_: hipLaunchKernelGGL((embed_fwd_wide_kernel<NW_>), grid, dim3(256), 0, stream, d, tokens, tok_bs, tok_ts, t_len, gamma, beta, (bf16_t*)y, ldy, mean, rstd, T, eps); break;
        switch (nw) { CASEW(1) CASEW(2) CASEW(3) CASEW(4) default: return SPN_ERR_ARG; }
#undef CASEW
        SPN_LAUNCH_CHECK();
        return SPN_OK;
    }
#define CASE(NV_) case NV_: hipLaunchKernelGGL((embed_fwd_kernel<NV_>), grid, dim3(256), 0, stream, d, tokens, tok_bs, tok_ts, t_len, gamma, beta, (bf16_t*)y, ldy, mean, rstd, T, eps); break;
    switch (nv) { CASE(1) CASE(2) CASE(4) CASE(6) CASE(8) default: return SPN_ERR_ARG; }
#undef CASE
    SPN_LAUNCH_CHECK();
    return SPN_OK;
}

// dtables (fp32 [V,E] per key) and dgamma/dbeta are ACCUMULATED.  ws: 2*T floats of workspace.
extern "C" int spn_embed_bwd(int nkeys, const float* const* tables, float* const* dtables, const int* V, const int* E,
                             const long* tokens, long tok_bs, long tok_ts, int t_len, const void* dy, long lddy, const float* gamma, const float* mean,
                             const float* rstd, float* dgamma, float* dbeta, float* ws, int T, int padding_idx, hipStream_t stream) {
    SPN_REQUIRE(nkeys > 0 && nkeys <= MAXK && tokens && dy && dtables && T > 0, "spn_embed_bwd: bad arguments");
    GatherDesc d;
    const int D = fill_gather(d, nkeys, tables, dtables, V, E);
    SPN_REQUIRE(D > 0 && D <= 2048 && lddy % 4 == 0, "spn_embed_bwd: widths must be multiples of 4, total <= 2048");
    float* s1 = ws; float* s2 = ws ? ws + T : nullptr;
    if (gamma) {
        SPN_REQUIRE(mean && rstd && dgamma && dbeta && ws, "spn_embed_bwd: LayerNorm buffers required with gamma");
        const int nv = round_nv((D + 255) / 256);
        const int stats_blocks = spn_tune_i(SPN_TUNE_EMBED_STATS_BLOCKS) > 0 ? spn_tune_i(SPN_TUNE_EMBED_STATS_BLOCKS) : 2048;   // tuning aid (2048: -5 % over 1024; 256 or 16384: slower)
        int rpb = cdiv(T, stats_blocks); rpb = ((rpb + 3) / 4) * 4;
        dim3 grid(cdiv(T, rpb));
#define CASE(NV_) case NV_: hipLaunchKernelGGL((embed_bwd_stats_kernel<NV_>), grid, dim3(256), 0, stream, d, tokens, tok_bs, tok_ts, t_len, (const bf16_t*)dy, lddy, gamma, mean, rstd, s1, s2, dgamma, dbeta, T, rpb); break;
        switch (nv) { CASE(1) CASE(2) CASE(4) CASE(6) CASE(8) default: return SPN_ERR_ARG; }
#undef CASE
    }
    bool mfma_ok = lddy % 8 == 0 && (((uintptr_t)dy) & 15) == 0;
    for (int i = 0; i < nkeys; ++i) mfma_ok = mfma_ok && E[i] == 128 && V[i] <= 512;
    const int mfma_env = spn_tune_i(SPN_TUNE_EMBED_SCATTER_MFMA);   // 0: LDS-atomic kernel
    if (mfma_ok && mfma_env) {
        const int mfma_blocks = spn_tune_i(SPN_TUNE_EMBED_SCATTER_BLOCKS) > 0 ? spn_tune_i(SPN_TUNE_EMBED_SCATTER_BLOCKS) : 256;   // tuning aid
        int chunks = mfma_blocks / nkeys;
        if (chunks > cdiv(T, 256)) chunks = cdiv(T, 256);
        if (chunks < 1) chunks = 1;
        int rpb = cdiv(T, chunks); rpb = ((rpb + 63) / 64) * 64;
        hipLaunchKernelGGL(embed_bwd_scatter_mfma_kernel, dim3(cdiv(T, rpb), nkeys), dim3(512), 0, stream, d, tokens, tok_bs, tok_ts, t_len,
                           (const bf16_t*)dy, lddy, gamma, mean, rstd, s1, s2, T, rpb, padding_idx);
        SPN_LAUNCH_CHECK();
        return SPN_OK;
    }
    int maxve = 0;
    for (int i = 0; i < nkeys; ++i) maxve = V[i] * E[i] > maxve ? V[i] * E[i] : maxve;
    const int lds_bytes = maxve * 4;
    const int use_lds = lds_bytes <= 150 * 1024;
    static std::atomic<unsigned> optin{0};
    if (use_lds) spn_lds_optin(optin, reinterpret_cast<const void*>(embed_bwd_scatter_kernel<true>), 160 * 1024);
    // one block per CU when the table copy fills the LDS, a few more when it is small
    int chunks = (lds_bytes > 72 * 1024 ? 256 : 512) / nkeys;
    if (chunks > cdiv(T, 512)) chunks = cdiv(T, 512);
    if (chunks < 1) chunks = 1;
    int rpb = cdiv(T, chunks); rpb = ((rpb + 63) / 64) * 64;   // 16 waves per block: twice the loads in flight of 8
    dim3 grid(cdiv(T, rpb), nkeys);
    if (use_lds) hipLaunchKernelGGL(embed_bwd_scatter_kernel<true>, grid, dim3(1024), lds_bytes, stream, d, tokens, tok_bs, tok_ts, t_len,
                                    (const bf16_t*)dy, lddy, gamma, mean, rstd, s1, s2, T, rpb, padding_idx);
    else hipLaunchKernelGGL(embed_bwd_scatter_kernel<false>, grid, dim3(1024), 0, stream, d, tokens, tok_bs, tok_ts, t_len,
                            (const bf16_t*)dy, lddy, gamma, mean, rstd, s1, s2, T, rpb, padding_idx);
    SPN_LAUNCH_CHECK();
    return SPN_OK;
}
