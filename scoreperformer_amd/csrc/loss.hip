// Loss-side kernels: cross-entropy over per-key logits, segment (bar/beat/onset) aggregation, Gaussian-kernel MMD.
//
// K11/CE  `F.cross_entropy(logits.transpose(1,2), labels[..., i], ignore_index=-100)` per key
//         (models/scoreperformer/wrappers.py:49-59): fp32 log-sum-exp per row, mean over non-ignored rows.
// K9      segment means `(out^T @ one_hot) / counts` (models/scoreperformer/mmd_transformer.py:325-340) done as a
//         run-length scan per (sequence, column) instead of the reference's dense (b,t,S) one-hot matmul, and the
//         scatter-back `embeddings[(b, segments)]` (mmd_transformer.py:362).
// K10     `MMDLoss.compute_mmd` (mmd_transformer.py:521-534): k(x,y) = exp(-||x-y||^2 / D^2); weighted by a 0/1 validity
//         weight per latent so that the boolean-mask gather `latents[mask]` (mmd:513) needs no host sync.
#include "common.h"

namespace {

// ---------------------------------------------------------------------------------------------------------
// cross entropy: one wave per row
// ---------------------------------------------------------------------------------------------------------
template <typename TL>
__device__ __forceinline__ float ld_logit(const TL* p) {
    if constexpr (sizeof(TL) == 4) return *p; else return bf2f(*p);
}

// per row: lse[row]; loss_sum += (lse - logit[label]) and count += 1 for non-ignored rows; argmax[row] (optional)
// EVAL (scoreperformer/models/scoreperformer/evaluator.py:48-106): the evaluator's per-key sums come out of the
// same pass -- metrics[0] += #(argmax == label), metrics[1] += distance in token-value space: |tv[argmax] - tv[label]| (evaluator.py:41-42)
// or, weighted, sum_c softmax_c * |tv[label] - tv[c]| (evaluator.py:44-45) -- over rows with a valid label; the valid count is sums[1].
template <typename TL, bool EVAL>
__global__ __launch_bounds__(256) void ce_fwd_kernel(const TL* __restrict__ logits, long ld, const long* __restrict__ labels,
                                                     long lab_bs, long lab_ts, int t_len, int ignore_index, float* __restrict__ lse,
                                                     float* __restrict__ sums /* [2]: loss sum, count */, int* __restrict__ argmax,
                                                     const float* __restrict__ tv, int weighted, float* __restrict__ metrics,
                                                     long T, int V) {
    __shared__ float blk[4][4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float loss = 0.f, cnt = 0.f, hit = 0.f, dist = 0.f;
    // a wave walks rows with a grid stride: one pair of (same-address) atomics per BLOCK at the end, not per 4 rows
    if (V <= 256) {
        // Vocabularies of the tuple keys (<= 260 incl. specials; the predicted ones <= 165): the row lives in four registers per lane, read
        // ONCE (the two-pass loop read it twice behind a max reduction), and the NEXT row and its label are requested before this row's
        // reductions -- the chain load -> max -> exp -> sum -> label -> logit[label] was five dependent trips per row and wave.
        // Same per-lane order of the maximum search and of the exponential sum as the general loop below: identical results.
        const long stride = (long)gridDim.x * 4;
        auto fetch = [&](long row, float (&v)[4], long& lab) __attribute__((always_inline)) {
            const long r = row < T ? row : T - 1;
            const TL* lr = logits + r * ld;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = lane + 64 * i;
                const float x = ld_logit(lr + (c < V ? c : V - 1));      // (clamped address, masked VALUE: no test around the load)
                v[i] = c < V ? x : -INFINITY;
            }
            lab = labels[(r / t_len) * lab_bs + (r % t_len) * lab_ts];
        };
        float vn[4]; long labn;
        long row = (long)blockIdx.x * 4 + w;
        if (row < T) fetch(row, vn, labn);
        for (; row < T; row += stride) {
            float v[4] = {vn[0], vn[1], vn[2], vn[3]};
            const long lab = labn;
            fetch(row + stride, vn, labn);
            // maximum by the DPP ladder, arg-max = the smallest index that holds it (torch.argmax's tie rule) by a second ladder on the
            // negated index (indices < 256 are exact in fp32) -- the shuffle-based pair reduction was twelve LDS-crossbar round trips per row
            const float m = wave_max(fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])));
            float cand = -1.0e9f;
#pragma unroll
            for (int i = 3; i >= 0; --i) if (v[i] == m) cand = -(float)(lane + 64 * i);
            int am = (int)(-wave_max(cand));
            am = am < V ? am : 0;                                   // (a row of NaNs matches nothing: index 0, as the general loop gives)
            float sx = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) if (lane + 64 * i < V) sx += __expf(v[i] - m);
            sx = wave_sum(sx);
            const float l = m + __logf(sx);
            if (EVAL && lab != ignore_index && tv && weighted) {
                const float target = tv[lab];
                float dd = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) if (lane + 64 * i < V) dd += __expf(v[i] - l) * fabsf(target - tv[lane + 64 * i]);
                dist += dd;
            }
            // logit[label]: held by lane label % 64 in slot label / 64 (the label is the same in every lane)
            float at_lab = 0.f;
            if (lab != ignore_index) {
                const int slot = (int)(lab >> 6);
                const float pick = slot == 0 ? v[0] : slot == 1 ? v[1] : slot == 2 ? v[2] : v[3];
                at_lab = __shfl(pick, (int)(lab & 63), 64);
            }
            if (lane == 0) {
                lse[row] = l;
                if (argmax) argmax[row] = am;
                if (lab != ignore_index) {
                    loss += l - at_lab; cnt += 1.f;
                    if (EVAL) {
                        hit += am == lab ? 1.f : 0.f;
                        if (tv && !weighted) dist += fabsf(tv[am] - tv[lab]);
                    }
                }
            }
        }
    } else
    for (long row = (long)blockIdx.x * 4 + w; row < T; row += (long)gridDim.x * 4) {
        const TL* lr = logits + row * ld;
        float m = -INFINITY;
        int am = 0;
        for (int c = lane; c < V; c += 64) {
            const float v = ld_logit(lr + c);
            if (v > m) { m = v; am = c; }
        }
        // wave argmax (first index on ties, like torch.argmax)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float om = __shfl_xor(m, o, 64);
            const int oa = __shfl_xor(am, o, 64);
            if (om > m || (om == m && oa < am)) { m = om; am = oa; }
        }
        float s = 0.f;
        for (int c = lane; c < V; c += 64) s += __expf(ld_logit(lr + c) - m);
        s = wave_sum(s);
        const float l = m + __logf(s);
        const long lab = labels[(row / t_len) * lab_bs + (row % t_len) * lab_ts];   // same address in every lane: one broadcast load
        if (EVAL && lab != ignore_index && tv && weighted) {
            const float target = tv[lab];
            float d = 0.f;
            for (int c = lane; c < V; c += 64) d += __expf(ld_logit(lr + c) - l) * fabsf(target - tv[c]);
            dist += d;                                                               // per-lane partial, reduced once per block
        }
        if (lane == 0) {
            lse[row] = l;
            if (argmax) argmax[row] = am;
            if (lab != ignore_index) {
                loss += l - ld_logit(lr + lab); cnt += 1.f;
                if (EVAL) {
                    hit += am == lab ? 1.f : 0.f;
                    if (tv && !weighted) dist += fabsf(tv[am] - tv[lab]);
                }
            }
        }
    }
    if (EVAL) dist = wave_sum(dist);
    if (lane == 0) { blk[0][w] = loss; blk[1][w] = cnt; blk[2][w] = hit; blk[3][w] = dist; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float ls = blk[0][0] + blk[0][1] + blk[0][2] + blk[0][3], cs = blk[1][0] + blk[1][1] + blk[1][2] + blk[1][3];
        if (cs > 0.f) {
            atomicAdd(sums, ls); atomicAdd(sums + 1, cs);
            if (EVAL) {
                atomicAdd(metrics, blk[2][0] + blk[2][1] + blk[2][2] + blk[2][3]);
                if (tv) atomicAdd(metrics + 1, blk[3][0] + blk[3][1] + blk[3][2] + blk[3][3]);
            }
        }
    }
}

// dlogits[row, c] = coef * (softmax - onehot) for non-ignored rows, 0 otherwise (also for pad columns V..Vpad-1);
// coef = *coef_ptr (device scalar = upstream grad / (count * n_active_keys))
template <typename TL>
__global__ __launch_bounds__(256) void ce_bwd_kernel(const TL* __restrict__ logits, long ld, const long* __restrict__ labels,
                                                     long lab_bs, long lab_ts, int t_len, int ignore_index, const float* __restrict__ lse,
                                                     const float* __restrict__ coef_ptr, bf16_t* __restrict__ dlogits, long ldd,
                                                     long T, int V, int Vpad) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= T) return;
    const long lab = labels[(row / t_len) * lab_bs + (row % t_len) * lab_ts];
    const bool live = lab != ignore_index;
    const float coef = live ? *coef_ptr : 0.f;
    const float l = lse[row];
    const TL* lr = logits + row * ld;
    for (int c = lane; c < Vpad; c += 64) {
        float g = 0.f;
        if (live && c < V) g = coef * (__expf(ld_logit(lr + c) - l) - (c == lab ? 1.f : 0.f));
        dlogits[row * ldd + c] = f2bf(g);
    }
}

// ---------------------------------------------------------------------------------------------------------
// segments
// ---------------------------------------------------------------------------------------------------------
// counts[b, seg[b, i]] += 1.  Ids come in runs (the collator's contract: sorted per sample), and the whole-sequence mode has ONE id per
// sample: a thread walks 32 consecutive positions and flushes one atomic per run instead of one per position (128 k same-address
// atomics on 2 counters per sample cost 0.5 ms).
__global__ void seg_count_kernel(const long* __restrict__ seg, float* __restrict__ counts, long BT, int t, int S) {
    constexpr int CH = 32;
    const long nchunk = (BT + CH - 1) / CH;
    for (long ch = (long)blockIdx.x * blockDim.x + threadIdx.x; ch < nchunk; ch += (long)gridDim.x * blockDim.x) {
        const long i0 = ch * CH, i1 = i0 + CH < BT ? i0 + CH : BT;
        long cur = -1, cb = 0;
        float n = 0.f;
        for (long i = i0; i < i1; ++i) {
            const long b = i / t, sgm = seg[i];
            if (sgm != cur || b != cb) {
                if (n > 0.f) atomicAdd(counts + cb * S + cur, n);
                cur = sgm; cb = b; n = 0.f;
            }
            n += 1.f;
        }
        if (n > 0.f) atomicAdd(counts + cb * S + cur, n);
    }
}

// y[b, t, c] = src[b, seg[b,t], c] * scale * (rowmask ? rowmask[b,t] : 1);  scale = 1/max(counts[b,seg],1) if counts
// VEC = 4: d, y_ld multiples of 4 and 16-byte aligned bases -> one float4 per thread; 32-bit index arithmetic (b*t*d < 2^31).
template <int VEC>
__global__ void seg_gather_kernel(const float* __restrict__ src, const long* __restrict__ seg, const float* __restrict__ counts,
                                  const uint8_t* __restrict__ rowmask, float* __restrict__ y, long y_ld, unsigned BT, int t, int S, int d,
                                  int accumulate, long src_ld) {
    const unsigned dv = (unsigned)d / VEC, total = BT * dv;
    for (unsigned idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const unsigned r = idx / dv, c = (idx - r * dv) * VEC;
        const unsigned b = r / (unsigned)t;
        const long sgm = seg[r];
        const bool live = !(rowmask && !rowmask[r]);
        const float den = counts ? fmaxf(counts[(long)b * S + sgm], 1.f) : 1.f;
        const float* sp = src + ((long)b * S + sgm) * src_ld + c;
        float* yp = y + (long)r * y_ld + c;
        if (VEC == 4) {
            f32x4 v = *reinterpret_cast<const f32x4*>(sp);
            if (counts) v = v / den;
            if (!live) v = f32x4{0.f, 0.f, 0.f, 0.f};
            if (accumulate) v += *reinterpret_cast<const f32x4*>(yp);
            *reinterpret_cast<f32x4*>(yp) = v;
        } else {
            float v = *sp;
            if (counts) v /= den;
            if (!live) v = 0.f;
            *yp = accumulate ? *yp + v : v;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// MMD.  sums[0] = sum_ij k(z_i,z_j), sums[1] = sum_ij w_i w_j k(y_i,y_j), sums[2] = sum_ij w_j k(z_i,y_j), sums[3] = sum w
// grid (ceil(Na/64), ceil(Nb/64), 3 pair types); block 256: thread (i = tid&63 row of a-tile) loops over 16 b columns
// ---------------------------------------------------------------------------------------------------------
constexpr int MMD_MAXD = 64;

__global__ __launch_bounds__(256) void mmd_fwd_kernel(const float* __restrict__ z, int Z, const float* __restrict__ y,
                                                      const float* __restrict__ w, int N, int D, float* __restrict__ sums) {
    __shared__ float at[64][MMD_MAXD + 1], bt[64][MMD_MAXD + 1], wa[64], wb[64], red[4];
    const int type = blockIdx.z;  // 0: zz, 1: yy, 2: zy
    const float* A = type == 1 ? y : z;  const int NA = type == 1 ? N : Z;
    const float* B = type == 0 ? z : y;  const int NB = type == 0 ? Z : N;
    const int a0 = blockIdx.x * 64, b0 = blockIdx.y * 64;
    if (a0 >= NA || b0 >= NB) return;
    for (int e = threadIdx.x; e < 64 * D; e += 256) {
        const int r = e / D, c = e % D;
        at[r][c] = (a0 + r < NA) ? A[(long)(a0 + r) * D + c] : 0.f;
        bt[r][c] = (b0 + r < NB) ? B[(long)(b0 + r) * D + c] : 0.f;
    }
    if (threadIdx.x < 64) {
        const int r = threadIdx.x;
        wa[r] = (a0 + r < NA) ? (type == 1 ? w[a0 + r] : 1.f) : 0.f;
        wb[r] = (b0 + r < NB) ? (type == 0 ? 1.f : w[b0 + r]) : 0.f;
    }
    __syncthreads();
    const int i = threadIdx.x & 63, jg = threadIdx.x >> 6;
    const float inv = 1.f / ((float)D * (float)D);
    float acc = 0.f;
    if (wa[i] != 0.f)
        for (int jj = 0; jj < 16; ++jj) {
            const int j = jg * 16 + jj;
            if (wb[j] == 0.f) continue;
            float d2 = 0.f;
            for (int c = 0; c < D; ++c) { const float t = at[i][c] - bt[j][c]; d2 = fmaf(t, t, d2); }
            acc += wa[i] * wb[j] * __expf(-d2 * inv);
        }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[jg] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(sums + type, red[0] + red[1] + red[2] + red[3]);
        if (type == 1 && blockIdx.y == 0) { float s = 0.f; for (int r = 0; r < 64; ++r) s += wa[r]; atomicAdd(sums + 3, s); }
    }
}

// dy_j = coef[0] * sum_i w_i w_j k(y_i,y_j) * (-2/D^2) (y_j - y_i) * 2      (yy term, symmetric -> factor 2)
//      + coef[1] * sum_i w_j k(z_i,y_j) * (-2/D^2) (y_j - z_i)              (zy term)
// coef (device): coef[0] = g / n^2, coef[1] = -2 g / (Z n).   grid (ceil(N/64)); block 256 = 64 rows j x 4 slices of i.
// DP = D padded to a power of two (zero columns), so that the per-lane gradient lives in registers.
template <int DP>
__global__ __launch_bounds__(256) void mmd_bwd_kernel(const float* __restrict__ z, int Z, const float* __restrict__ y,
                                                      const float* __restrict__ w, int N, int D, const float* __restrict__ coef,
                                                      float* __restrict__ dy) {
    __shared__ float yt[64][DP + 1], ot[64][DP + 1], wo[64], res[64][DP + 1];
    const int j0 = blockIdx.x * 64;
    const int j = threadIdx.x & 63, sl = threadIdx.x >> 6;
    for (int e = threadIdx.x; e < 64 * DP; e += 256) {
        const int r = e / DP, c = e % DP;
        yt[r][c] = (j0 + r < N && c < D) ? y[(long)(j0 + r) * D + c] : 0.f;
        res[r][c] = 0.f;
    }
    const float wj = (j0 + j < N) ? w[j0 + j] : 0.f;
    const float inv = 1.f / ((float)D * (float)D);
    float g[DP];
#pragma unroll
    for (int c = 0; c < DP; ++c) g[c] = 0.f;
    const float c_yy = coef[0] * 2.f, c_zy = coef[1];
    for (int pass = 0; pass < 2; ++pass) {  // pass 0: others = y (weights w), pass 1: others = z (weights 1)
        const float* O = pass == 0 ? y : z;
        const int NO = pass == 0 ? N : Z;
        const float cc = pass == 0 ? c_yy : c_zy;
        for (int o0 = blockIdx.y * 64; o0 < NO; o0 += 64 * gridDim.y) {   // the "others" are split over gridDim.y blocks per j tile
            __syncthreads();
            for (int e = threadIdx.x; e < 64 * DP; e += 256) {
                const int r = e / DP, c = e % DP;
                ot[r][c] = (o0 + r < NO && c < D) ? O[(long)(o0 + r) * D + c] : 0.f;
            }
            if (threadIdx.x < 64) wo[threadIdx.x] = (o0 + threadIdx.x < NO) ? (pass == 0 ? w[o0 + threadIdx.x] : 1.f) : 0.f;
            __syncthreads();
            if (wj != 0.f)
                for (int ii = 0; ii < 16; ++ii) {
                    const int i = sl * 16 + ii;
                    if (wo[i] == 0.f) continue;
                    float d2 = 0.f;
#pragma unroll
                    for (int c = 0; c < DP; ++c) { const float t = yt[j][c] - ot[i][c]; d2 = fmaf(t, t, d2); }
                    const float kk = cc * wj * wo[i] * __expf(-d2 * inv) * (-2.f * inv);
#pragma unroll
                    for (int c = 0; c < DP; ++c) g[c] = fmaf(kk, yt[j][c] - ot[i][c], g[c]);
                }
        }
    }
    for (int s4 = 0; s4 < 4; ++s4) {
        __syncthreads();
        if (sl == s4) {
#pragma unroll
            for (int c = 0; c < DP; ++c) res[j][c] += g[c];
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 64 * D; e += 256) {
        const int r = e / D, c = e % D;
        if (j0 + r < N) {
            if (gridDim.y == 1) dy[(long)(j0 + r) * D + c] = res[r][c];
            else if (res[r][c] != 0.f) atomicAdd(dy + (long)(j0 + r) * D + c, res[r][c]);   // dy zeroed by the launcher
        }
    }
}

inline int grid_for(long total, int block = 256) { long g = (total + block - 1) / block; return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g)); }

}  // namespace

// logits: [T, V] (fp32 dtype 0 / bf16 dtype 1, row stride ld); labels int64 [B, t_len] view, element strides (lab_bs, lab_ts), T = B*t_len.
// sums[2] (loss sum, valid count) are ACCUMULATED (zero first).  argmax may be null.
namespace {
template <bool EVAL>
void launch_ce_fwd(const void* logits, int dtype, long ld, const long* labels, long lab_bs, long lab_ts, int t_len, int ignore_index, float* lse,
                   float* sums, int* argmax, const float* tv, int weighted, float* metrics, long T, int V, hipStream_t s) {
    dim3 grid(cdiv(T, 4) < 2048 ? cdiv(T, 4) : 2048);
    if (dtype == 0)
        hipLaunchKernelGGL((ce_fwd_kernel<float, EVAL>), grid, dim3(256), 0, s, (const float*)logits, ld, labels, lab_bs, lab_ts, t_len, ignore_index, lse,
                           sums, argmax, tv, weighted, metrics, T, V);
    else
        hipLaunchKernelGGL((ce_fwd_kernel<bf16_t, EVAL>), grid, dim3(256), 0, s, (const bf16_t*)logits, ld, labels, lab_bs, lab_ts, t_len, ignore_index,
                           lse, sums, argmax, tv, weighted, metrics, T, V);
}
}  // namespace

extern "C" int spn_ce_fwd(const void* logits, int dtype, long ld, const long* labels, long lab_bs, long lab_ts, int t_len, int ignore_index, float* lse,
                          float* sums, int* argmax, long T, int V, hipStream_t s) {
    SPN_REQUIRE(logits && labels && lse && sums && T > 0 && V > 0, "spn_ce_fwd: bad arguments");
    launch_ce_fwd<false>(logits, dtype, ld, labels, lab_bs, lab_ts, t_len, ignore_index, lse, sums, argmax, nullptr, 0, nullptr, T, V, s);
    SPN_LAUNCH_CHECK();
    return SPN_OK;
}

extern "C" int spn_ce_fwd_eval(const void* logits, int dtype, long ld, const long* labels, long lab_bs, long lab_ts, int t_len, int ignore_index,
                               float* lse, float* sums, int* argmax, const float* token_values, int weighted, float* metrics, long T, int V,
                               hipStream_t s) {
    SPN_REQUIRE(logits && labels && lse && sums && metrics && T > 0 && V > 0, "spn_ce_fwd_eval: bad arguments");
    launch_ce_fwd<true>(logits, dtype, ld, labels, lab_bs, lab_ts, t_len, ignore_index, lse, sums, argmax, token_values, weighted, metrics, T, V, s);
    SPN_LAUNCH_CHECK();
    return SPN_OK;
}

extern "C" int spn_ce_bwd(const void* logits, int dtype, long ld, const long* labels, long lab_bs, long lab_ts, int t_len, int ignore_index,
                          const float* lse, const float* coef, void* dlogits, long ldd, long T, int V, int Vpad, hipStream_t s) {
    SPN_REQUIRE(logits && labels && lse && coef && dlogits && T > 0 && V > 0 && Vpad >= V, "spn_ce_bwd: bad arguments");
    dim3 grid(cdiv(T, 4));
    if (dtype == 0) hipLaunchKernelGGL((ce_bwd_kernel<float>), grid, dim3(256), 0, s, (const float*)logits, ld, labels, lab_bs, lab_ts, t_len, ignore_index, lse, coef, (bf16_t*)dlogits, ldd, T, V, Vpad);
    else hipLaunchKernelGGL((ce_bwd_kernel<bf16_t>), grid, dim3(256), 0, s, (const bf16_t*)logits, ld, labels, lab_bs, lab_ts, t_len, ignore_index, lse, coef, (bf16_t*)dlogits, ldd, T, V, Vpad);
    SPN_LAUNCH_CHECK();
    return SPN_OK;
}

// counts[b, S] fp32 (zeroed by caller) += number of rows per segment id
extern "C" int spn_segment_count(const long* seg, float* counts, int b, int t, int S, hipStream_t s) {
    SPN_REQUIRE(seg && counts && b > 0 && t > 0 && S > 0, "spn_segment_count: bad arguments");
    hipLaunchKernelGGL(seg_count_kernel, dim3(grid_for(((long)b * t + 31) / 32)), dim3(256), 0, s, seg, counts, (long)b * t, t, S);
    SPN_LAUNCH_CHECK();
    return SPN_OK;
}

// out[b,S,d] fp32 (row stride out_ld >= d: a column slice of a wider buffer; zeroed by caller) += segment sums (counts == null) or means
// (counts given) of x[b,t,d]
extern "C" int spn_segment_sum_multi(const void* x, int dtype, long x_bs, long x_ts, const uint8_t* rowmask, int nl, const long* const* seg,
                                     const float* const* counts, float* const* out, const long* out_ld, const int* S, int b, int t, int d,
                                     hipStream_t s);
extern "C" int spn_segment_sum(const void* x, int dtype, long x_bs, long x_ts, const long* seg, const float* counts,
                               const uint8_t* rowmask, float* out, long out_ld, int b, int t, int S, int d, hipStream_t s) {
    SPN_REQUIRE(x && seg && out && b > 0 && t > 0 && S > 0 && d > 0 && out_ld >= d, "spn_segment_sum: bad arguments");
    // one level of the all-levels kernel (ids, counts and the row mask staged in LDS; whole runs stored, cut runs added atomically)
    const long* segs[1] = {seg};
    const float* cnts[1] = {counts};
    float* outs[1] = {out};
    const long lds[1] = {out_ld};
    const int Ss[1] = {S};
    return spn_segment_sum_multi(x, dtype, x_bs, x_ts, rowmask, 1, segs, counts ? cnts : nullptr, outs, lds, Ss, b, t, d, s);
}

// y[b*t, d] (row stride y_ld) (+)= src[b, seg, 0:d] (row stride src_ld >= d) (/ counts) (* rowmask)
extern "C" int spn_segment_gather(const float* src, long src_ld, const long* seg, const float* counts, const uint8_t* rowmask, float* y,
                                  long y_ld, int b, int t, int S, int d, int accumulate, hipStream_t s) {
    SPN_REQUIRE(src && seg && y && b > 0 && t > 0 && S > 0 && d > 0 && src_ld >= d, "spn_segment_gather: bad arguments");
    SPN_REQUIRE((long)b * t * d < (1l << 31), "spn_segment_gather: b*t*d must be below 2^31");
    const bool vec = d % 4 == 0 && y_ld % 4 == 0 && src_ld % 4 == 0 && (((uintptr_t)src | (uintptr_t)y) & 15) == 0;
    if (vec) hipLaunchKernelGGL(seg_gather_kernel<4>, dim3(grid_for((long)b * t * (d / 4))), dim3(256), 0, s, src, seg, counts, rowmask, y, y_ld,
                                (unsigned)(b * t), t, S, d, accumulate, src_ld);
    else hipLaunchKernelGGL(seg_gather_kernel<1>, dim3(grid_for((long)b * t * d)), dim3(256), 0, s, src, seg, counts, rowmask, y, y_ld,
                            (unsigned)(b * t), t, S, d, accumulate, src_ld);
    SPN_LAUNCH_CHECK();
    return SPN_OK;
}

// ---- all latent levels of the hierarchical heads in ONE pass over the hidden states (mmd_transformer.py:325-340 per level) ----------
// Every level aggregates the SAME hidden states under its own segmentation (sequence mean, bars, beats, onsets); level by level that is
// four passes over [b, t, d] forward (4 x 268 MB at C3) and four read-modify-write passes over the gradient backward.  Here a thread owns
// one column of one sample's row chunk, keeps one running sum per level and flushes a level when ITS segment id changes:
//   out_l[b, s, 0:d] += mean over the rows of segment s of x[b, t, 0:d] * rowmask[b, t]          (out_l has row stride ld_l >= d)
// and the backward writes the hidden-state gradient once:  y[b, t, :] = rowmask * sum_l src_l[b, seg_l[b, t], 0:d] / max(count_l, 1).
struct SegMulti {
    const long* seg[8];
    const float* counts[8];
    float* out[8];           // gather: the sources
    long ld[8];
    int S[8];
    int nl;
};

// One COLUMN per lane, like seg_sum_kernel: a wave's flush is one atomicAdd instruction over 64 consecutive floats = two whole cache lines
// (four columns per lane with 16-byte loads were measured in round 4 for the one-level kernel: 247 us against 141 -- each of its four
// atomic instructions touches a quarter of eight lines, and the kernel is bound by its per-run flushes, not by loads in flight).
template <typename TX, int NL>
__global__ __launch_bounds__(256) void seg_sum_multi_kernel(SegMulti a, const TX* __restrict__ x, long x_bs, long x_ts,
                                                            const uint8_t* __restrict__ rowmask, int t, int d, int t_chunk) {
    // the chunk's ids (all levels), their counts and the row mask are staged in LDS once, by coalesced loads: fetched row by row inside
    // the loop they are block-uniform SCALAR loads, a dependent trip to memory per eight rows that no number of resident waves hides
    // (round 6: a pass that flushes nothing took 90 us for 134 MB; 34 us with the staging)
    constexpr int MAXC = 256;
    __shared__ int ids[NL][MAXC];
    __shared__ float cnt[NL][MAXC];
    __shared__ uint8_t live[MAXC];
    const int b = blockIdx.y;
    const int i_begin = blockIdx.z * t_chunk, i_end = min(t, i_begin + t_chunk);
    if (i_begin >= i_end) return;
    const int rows = i_end - i_begin;
    for (int e = threadIdx.x; e < rows; e += blockDim.x) {
        const long r = (long)b * t + i_begin + e;
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            const long sg = a.seg[l][r];
            ids[l][e] = (int)sg;
            cnt[l][e] = a.counts[l] ? a.counts[l][(long)b * a.S[l] + sg] : -1.f;      // -1: plain sums (no count to compare a run with)
        }
        live[e] = rowmask ? rowmask[r] : 1;
    }
    __syncthreads();
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= d) return;
    const TX* xb = x + (long)b * x_bs + (long)i_begin * x_ts + c;
    float acc[NL], len[NL], cn[NL];
    int cur[NL];
#pragma unroll
    for (int l = 0; l < NL; ++l) { acc[l] = 0.f; len[l] = 0.f; cur[l] = ids[l][0]; cn[l] = cnt[l][0]; }
    // a run that holds EVERY row of its id in this sample (its length equals the id's count) is the only writer of its output row: a
    // plain store; only runs cut by a chunk border (or ids that recur later: unsorted input) go through the L2's atomic unit
    auto flush = [&](int l) {
        if (acc[l] != 0.f) {
            float* o = a.out[l] + ((long)b * a.S[l] + cur[l]) * a.ld[l] + c;
            const float v = cn[l] < 0.f ? acc[l] : acc[l] * (1.f / fmaxf(cn[l], 1.f));
            if (len[l] == cn[l]) *o = v; else atomicAdd(o, v);
        }
        acc[l] = 0.f; len[l] = 0.f;
    };
    for (int i = 0; i < rows; i += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {          // eight rows requested before the first is folded
            const int r = min(i + u, rows - 1);
            if constexpr (sizeof(TX) == 4) v[u] = xb[(long)r * x_ts];
            else v[u] = bf2f(xb[(long)r * x_ts]);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (i + u >= rows) break;
            const float xv = live[i + u] ? v[u] : 0.f;
#pragma unroll
            for (int l = 0; l < NL; ++l) {
                const int sg = ids[l][i + u];
                if (sg != cur[l]) { flush(l); cur[l] = sg; cn[l] = cnt[l][i + u]; }
                acc[l] += xv; len[l] += 1.f;
            }
        }
    }
#pragma unroll
    for (int l = 0; l < NL; ++l) flush(l);
}

// y[b, t, c .. c + 3] = live * sum_l src_l[b, id_l, c ..] * scale_l with the ids and scales of the block's rows staged in LDS (a thread
// looked its row's id up itself before: id -> source row, two dependent trips per element); four rows in flight per thread
template <int NL>
__global__ __launch_bounds__(256) void seg_gather_multi_kernel(SegMulti a, const uint8_t* __restrict__ rowmask, float* __restrict__ y, long y_ld,
                                                               int t, int d, int t_chunk) {
    constexpr int MAXC = 64;
    __shared__ int ids[NL][MAXC];
    __shared__ float scl[NL][MAXC];
    __shared__ uint8_t live[MAXC];
    const int b = blockIdx.y;
    const int i_begin = blockIdx.x * t_chunk, i_end = min(t, i_begin + t_chunk);
    if (i_begin >= i_end) return;
    const int rows = i_end - i_begin;
    for (int e = threadIdx.x; e < rows; e += blockDim.x) {
        const long r = (long)b * t + i_begin + e;
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            const long sg = a.seg[l][r];
            ids[l][e] = (int)sg;
            scl[l][e] = a.counts[l] ? 1.f / fmaxf(a.counts[l][(long)b * a.S[l] + sg], 1.f) : 1.f;
        }
        live[e] = rowmask ? rowmask[r] : 1;
    }
    __syncthreads();
    const int dv = d / 4;
    for (int idx = threadIdx.x; idx < rows * dv; idx += 4 * blockDim.x) {
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int id2 = idx + u * blockDim.x;
            v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (id2 >= rows * dv) continue;
            const int e = id2 / dv, c = (id2 - e * dv) * 4;
            if (!live[e]) continue;
#pragma unroll
            for (int l = 0; l < NL; ++l)
                v[u] += *reinterpret_cast<const f32x4*>(a.out[l] + ((long)b * a.S[l] + ids[l][e]) * a.ld[l] + c) * scl[l][e];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int id2 = idx + u * blockDim.x;
            if (id2 >= rows * dv) continue;
            const int e = id2 / dv, c = (id2 - e * dv) * 4;
            *reinterpret_cast<f32x4*>(y + ((long)b * t + i_begin + e) * y_ld + c) = v[u];
        }
    }
}

static int fill_seg_multi(SegMulti& a, int nl, const long* const* seg, const float* const* counts, float* const* ptr, const long* ld,
                          const int* S, int d) {
    SPN_REQUIRE(nl > 0 && nl <= 8 && seg && ptr && ld && S, "spn_segment_*_multi: 1..8 levels");
    memset(&a, 0, sizeof(a));
    a.nl = nl;
    for (int l = 0; l < nl; ++l) {
        SPN_REQUIRE(seg[l] && ptr[l] && S[l] > 0 && ld[l] >= d, "spn_segment_*_multi: every level needs ids and a [b, S, >= d] buffer");
        a.seg[l] = seg[l]; a.counts[l] = counts ? counts[l] : nullptr; a.out[l] = ptr[l]; a.ld[l] = ld[l]; a.S[l] = S[l];
    }
    return SPN_OK;
}

// out_l[b, S_l, 0:d] (row stride ld_l; zeroed by the caller) += segment MEANS of x[b, t, 0:d] * rowmask under seg_l, for all levels in one
// pass.  x fp32 / bf16 (dtype 0 / 1) with strides (x_bs, x_ts); d a multiple of 4; host arrays of nl <= 8 entries.
extern "C" int spn_segment_sum_multi(const void* x, int dtype, long x_bs, long x_ts, const uint8_t* rowmask, int nl, const long* const* seg,
                                     const float* const* counts, float* const* out, const long* out_ld, const int* S, int b, int t, int d,
                                     hipStream_t s) {
    SPN_REQUIRE(x && b > 0 && t > 0 && d > 0, "spn_segment_sum_multi: bad arguments");
    SegMulti a;
    if (int rc = fill_seg_multi(a, nl, seg, counts, out, out_ld, S, d)) return rc;
    const int t_chunk = t > 128 ? 128 : t;   // <= 256 rows (the LDS staging of the kernel)
    const int TH = d >= 256 ? 256 : 64;
    dim3 grid(cdiv(d, TH), b, cdiv(t, t_chunk));
#define SPN_SEG_MULTI(NL_)                                                                                                                 \
    case NL_:                                                                                                                              \
        if (dtype == 0) hipLaunchKernelGGL((seg_sum_multi_kernel<float, NL_>), grid, dim3(TH), 0, s, a, (const float*)x, x_bs, x_ts, rowmask, t, d, t_chunk); \
        else hipLaunchKernelGGL((seg_sum_multi_kernel<bf16_t, NL_>), grid, dim3(TH), 0, s, a, (const bf16_t*)x, x_bs, x_ts, rowmask, t, d, t_chunk);         \
        break;
    switch (nl) {
        SPN_SEG_MULTI(1) SPN_SEG_MULTI(2) SPN_SEG_MULTI(3) SPN_SEG_MULTI(4) SPN_SEG_MULTI(5) SPN_SEG_MULTI(6) SPN_SEG_MULTI(7) SPN_SEG_MULTI(8)
    }
#undef SPN_SEG_MULTI
    SPN_LAUNCH_CHECK();
    return SPN_OK;
}

// y[b * t, 0:d] (row stride y_ld) = rowmask * sum_l src_l[b, seg_l, 0:d] / max(counts_l, 1): the backward of the above, written once
extern "C" int spn_segment_gather_multi(int nl, const float* const* src, const long* src_ld, const long* const* seg,
                                        const float* const* counts, const int* S, const uint8_t* rowmask, float* y, long y_ld, int b, int t,
                                        int d, hipStream_t s) {
    SPN_REQUIRE(y && b > 0 && t > 0 && d > 0 && (d % 4) == 0 && (y_ld % 4) == 0 && ((uintptr_t)y & 15) == 0 && (long)b * t * d < (1l << 31),
                "spn_segment_gather_multi: d, y_ld multiples of 4, y 16-byte aligned, b*t*d below 2^31");
    SegMulti a;
    if (int rc = fill_seg_multi(a, nl, seg, counts, const_cast<float* const*>(src), src_ld, S, d)) return rc;
    for (int l = 0; l < nl; ++l)
        SPN_REQUIRE((src_ld[l] % 4) == 0 && ((uintptr_t)src[l] & 15) == 0, "spn_segment_gather_multi: sources 16-byte aligned, row strides multiples of 4");
    const int t_chunk = 64;
    dim3 grid(cdiv(t, t_chunk), b);
#define SPN_SEG_GATHER(NL_) case NL_: hipLaunchKernelGGL((seg_gather_multi_kernel<NL_>), grid, dim3(256), 0, s, a, rowmask, y, y_ld, t, d, t_chunk); break;
    switch (nl) {
        SPN_SEG_GATHER(1) SPN_SEG_GATHER(2) SPN_SEG_GATHER(3) SPN_SEG_GATHER(4) SPN_SEG_GATHER(5) SPN_SEG_GATHER(6) SPN_SEG_GATHER(7) SPN_SEG_GATHER(8)
    }
#undef SPN_SEG_GATHER
    SPN_LAUNCH_CHECK();
    return SPN_OK;
}

// The scalar tail of compute_mmd (mmd_transformer.py:529-534) on the device, one launch: with n = max(sum w, 1)
//   g == null:  out[0] = kzz / Z^2 + kyy / n^2 - 2 kzy / (Z n)                         (the loss value)
//   g != null:  out[0] = g / n^2,  out[1] = -2 g / (Z n)                                (the coefficients spn_mmd_bwd takes)
// (as separate tensor ops these were ~10 + ~8 launches of one-element kernels per latent level)
__global__ void mmd_scalars_kernel(const float* __restrict__ sums, float Z, const float* __restrict__ g, float* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const float n = fmaxf(sums[3], 1.f);
    if (g) { out[0] = g[0] / (n * n); out[1] = -2.f * g[0] / (Z * n); }
    else out[0] = sums[0] / (Z * Z) + sums[1] / (n * n) - 2.f * sums[2] / (Z * n);
}

extern "C" int spn_mmd_scalars(const float* sums, int Z, const float* g, float* out, hipStream_t s) {
    SPN_REQUIRE(sums && out && Z > 0, "spn_mmd_scalars: bad arguments");
    hipLaunchKernelGGL(mmd_scalars_kernel, dim3(1), dim3(64), 0, s, sums, (float)Z, g, out);
    SPN_LAUNCH_CHECK();
    return SPN_OK;
}

// sums[4] fp32 (zeroed by caller): kzz, kyy (weighted), kzy (weighted), sum of weights
extern "C" int spn_mmd_fwd(const float* z, int Z, const float* y, const float* w, int N, int D, float* sums, hipStream_t s) {
    SPN_REQUIRE(z && y && w && sums && Z > 0 && N > 0 && D > 0 && D <= MMD_MAXD, "spn_mmd_fwd: bad arguments (D <= 64)");
    const int mx = Z > N ? Z : N;
    hipLaunchKernelGGL(mmd_fwd_kernel, dim3(cdiv(mx, 64), cdiv(mx, 64), 3), dim3(256), 0, s, z, Z, y, w, N, D, sums);
    SPN_LAUNCH_CHECK();
    return SPN_OK;
}

extern "C" int spn_mmd_bwd(const float* z, int Z, const float* y, const float* w, int N, int D, const float* coef, float* dy,
                           hipStream_t s) {
    SPN_REQUIRE(z && y && w && coef && dy && Z > 0 && N > 0 && D > 0 && D <= MMD_MAXD, "spn_mmd_bwd: bad arguments (D <= 64)");
    // 64 j rows per block leaves most CUs idle at N = 4096 (64 blocks): split the loop over the other rows 8 ways
    const int split = N >= 1024 ? 8 : 1;
    if (split > 1) hipMemsetAsync(dy, 0, (size_t)N * D * sizeof(float), s);
    dim3 grid(cdiv(N, 64), split);
    if (D <= 4) hipLaunchKernelGGL((mmd_bwd_kernel<4>), grid, dim3(256), 0, s, z, Z, y, w, N, D, coef, dy);
    else if (D <= 8) hipLaunchKernelGGL((mmd_bwd_kernel<8>), grid, dim3(256), 0, s, z, Z, y, w, N, D, coef, dy);
    else if (D <= 16) hipLaunchKernelGGL((mmd_bwd_kernel<16>), grid, dim3(256), 0, s, z, Z, y, w, N, D, coef, dy);
    else if (D <= 32) hipLaunchKernelGGL((mmd_bwd_kernel<32>), grid, dim3(256), 0, s, z, Z, y, w, N, D, coef, dy);
    else hipLaunchKernelGGL((mmd_bwd_kernel<64>), grid, dim3(256), 0, s, z, Z, y, w, N, D, coef, dy);
    SPN_LAUNCH_CHECK();
    return SPN_OK;
}
