// Device-side batch builder: the reference's MixedLMScorePerformanceCollator as ONE kernel.
//
// Replaces scoreperformer/data/collators/score_performance.py:35-115,186-234 (+ the base classes in performance.py:18-92,
// 100-115,213-247): pad/stack the ragged score / performance token arrays of a batch, build the key-padding masks and lengths,
// zero-pad the bar / beat / onset segment ids to the score length, copy the deadpan flags, and derive the MixedLM pair
//   masked_perf = MASK  where the token id is not ignored AND its dim is not ignored, else the token
//   labels      = token where (not ignored id [AND not ignored dim]), else label_pad
// The reference does this per sample in Python on the host and ships 9 derived int64 tensors over PCIe every step; here only the
// raw int32 tokens + row offsets cross, everything else is produced in HBM.  Integer work, HBM-bound:
// one thread per output token, 8-byte coalesced stores; algorithmic bytes per batch =
//   4*(sum_s*Ks + sum_p*Kp + 3*sum_s) read + 8*(b*Ls*(Ks+3) + 3*b*Lp*Kp) + b*(Ls+Lp) written.
#include "common.h"

namespace {

struct CollateArgs {
    const int32_t* score_flat; const int32_t* perf_flat; const int32_t* seg_flat;   // [sum_s,Ks] [sum_p,Kp] [3,sum_s]
    const int32_t* score_off; const int32_t* perf_off;                               // [b+1] row offsets
    const uint8_t* deadpan;                                                          // [b] or null
    int b, Ks, Kp, Ls, Lp; long sum_s;
    int pad_id, mask_id, label_pad_id, n_ignore, label_pad_ignored_dims;
    int ignore_ids[16];
    unsigned ignore_dims;                                                            // bit k: dim k is never masked
    long long* score; uint8_t* score_mask; long long* score_len;
    long long* perf; uint8_t* perf_mask; long long* perf_len;
    long long* masked_perf; long long* labels;
    long long* bar; long long* beat; long long* onset;
    uint8_t* deadpan_mask;
};

typedef long long i64x2 __attribute__((ext_vector_type(2)));

// Idx = unsigned when the batch has < 2^32 output tokens (always, in practice): 32-bit divisions instead of 64-bit ones.
// W = token dims per thread: 2 when both token widths are even (OctupleM: 10 / 12) -> 8-byte loads, 16-byte stores per lane.
template <typename Idx, int W>
__global__ __launch_bounds__(256) void collate_mixlm_kernel(CollateArgs a) {
    const int KpW = a.Kp / W, KsW = a.Ks / W;
    const Idx n_perf = (Idx)a.b * a.Lp * KpW, n_score = (Idx)a.b * a.Ls * KsW, n_seg = a.seg_flat ? (Idx)a.b * a.Ls : 0;
    const Idx total = n_perf + n_score + n_seg;
    for (Idx idx = (Idx)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (Idx)gridDim.x * 256) {
        if (idx < n_perf) {                                   // performance tokens (i, pos, k .. k+W-1)
            const Idx row = idx / KpW;
            const int k = (int)(idx - row * KpW) * W;
            const int i = (int)(row / a.Lp), pos = (int)(row - (Idx)i * a.Lp);
            const int o0 = a.perf_off[i], n = a.perf_off[i + 1] - o0;
            int tok[W];
            if (pos < n) {
                const int32_t* src = a.perf_flat + (long)(o0 + pos) * a.Kp + k;
                if (W == 2) { const int2 t = *reinterpret_cast<const int2*>(src); tok[0] = t.x; tok[W - 1] = t.y; }
                else tok[0] = *src;
            } else {
#pragma unroll
                for (int w = 0; w < W; ++w) tok[w] = a.pad_id;
            }
            long long m[W], l[W], t64[W];
#pragma unroll
            for (int w = 0; w < W; ++w) {
                bool ignored = tok[w] == a.pad_id;
                for (int q = 0; q < a.n_ignore; ++q) ignored = ignored || tok[w] == a.ignore_ids[q];
                const bool dim_ignored = (a.ignore_dims >> (k + w)) & 1u;
                t64[w] = tok[w];
                m[w] = (!ignored && !dim_ignored) ? a.mask_id : tok[w];
                l[w] = (!ignored && !(a.label_pad_ignored_dims && dim_ignored)) ? tok[w] : a.label_pad_id;
            }
            const long e = (long)row * a.Kp + k;
            if (W == 2) {
                *reinterpret_cast<i64x2*>(a.perf + e) = i64x2{t64[0], t64[W - 1]};
                *reinterpret_cast<i64x2*>(a.masked_perf + e) = i64x2{m[0], m[W - 1]};
                *reinterpret_cast<i64x2*>(a.labels + e) = i64x2{l[0], l[W - 1]};
            } else {
                a.perf[e] = t64[0]; a.masked_perf[e] = m[0]; a.labels[e] = l[0];
            }
            if (k == 0) {
                a.perf_mask[row] = pos < n;
                if (pos == 0) { a.perf_len[i] = n; a.deadpan_mask[i] = a.deadpan ? (a.deadpan[i] != 0) : 0; }
            }
        } else if (idx < n_perf + n_score) {                  // score tokens
            const Idx j = idx - n_perf;
            const Idx row = j / KsW;
            const int k = (int)(j - row * KsW) * W;
            const int i = (int)(row / a.Ls), pos = (int)(row - (Idx)i * a.Ls);
            const int o0 = a.score_off[i], n = a.score_off[i + 1] - o0;
            const long e = (long)row * a.Ks + k;
            if (W == 2) {
                int2 t = {a.pad_id, a.pad_id};
                if (pos < n) t = *reinterpret_cast<const int2*>(a.score_flat + (long)(o0 + pos) * a.Ks + k);
                *reinterpret_cast<i64x2*>(a.score + e) = i64x2{t.x, t.y};
            } else {
                a.score[e] = pos < n ? a.score_flat[(long)(o0 + pos) * a.Ks + k] : a.pad_id;
            }
            if (k == 0) {
                a.score_mask[row] = pos < n;
                if (pos == 0) a.score_len[i] = n;
            }
        } else {                                              // segment ids of score position (i, pos)
            const Idx row = idx - n_perf - n_score;
            const int i = (int)(row / a.Ls), pos = (int)(row - (Idx)i * a.Ls);
            const int o0 = a.score_off[i], n = a.score_off[i + 1] - o0;
            const bool in = pos < n;
            a.bar[row] = in ? a.seg_flat[o0 + pos] : 0;
            a.beat[row] = in ? a.seg_flat[a.sum_s + o0 + pos] : 0;
            a.onset[row] = in ? a.seg_flat[2 * a.sum_s + o0 + pos] : 0;
        }
    }
}

// out[i, pos, :] = pos < n_i ? flat[off_i + pos, :] : pad;  mask, lengths alike: one more ragged token array of the batch (the
// "noisy performance" of score_performance.py:48-51,66-69,94-95), same element-per-thread scheme
__global__ __launch_bounds__(256) void collate_pad_kernel(const int32_t* __restrict__ flat, const int32_t* __restrict__ off, int b, int K, int L,
                                                          int pad_id, long long* __restrict__ out, uint8_t* __restrict__ mask,
                                                          long long* __restrict__ len) {
    const long total = (long)b * L * K;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int k = (int)(idx % K);
        const long row = idx / K;
        const int pos = (int)(row % L), i = (int)(row / L);
        const int o0 = off[i], n = off[i + 1] - o0;
        out[idx] = pos < n ? flat[(long)(o0 + pos) * K + k] : pad_id;
        if (k == 0) {
            mask[row] = pos < n;
            if (pos == 0) len[i] = n;
        }
    }
}

}  // namespace

extern "C" int spn_collate_pad_tokens(const int32_t* flat, const int32_t* off, int b, int K, int L, int pad_id, long long* out, uint8_t* mask,
                                      long long* len, hipStream_t stream) {
    SPN_REQUIRE(flat && off && out && mask && len && b > 0 && K > 0 && L > 0, "spn_collate_pad_tokens: bad arguments");
    const long total = (long)b * L * K;
    long g = (total + 255) / 256;
    if (g > 16384) g = 16384;
    hipLaunchKernelGGL(collate_pad_kernel, dim3((unsigned)g), dim3(256), 0, stream, flat, off, b, K, L, pad_id, out, mask, len);
    SPN_LAUNCH_CHECK();
    return SPN_OK;
}

extern "C" int spn_collate_mixlm(const int32_t* score_flat, const int32_t* perf_flat, const int32_t* seg_flat, const int32_t* score_off,
                                 const int32_t* perf_off, const uint8_t* deadpan, int b, int Ks, int Kp, int Ls, int Lp, long sum_s,
                                 int pad_id, int mask_id, int label_pad_id, const int* ignore_ids, int n_ignore, unsigned ignore_dims,
                                 int label_pad_ignored_dims, long long* score, uint8_t* score_mask, long long* score_len, long long* perf,
                                 uint8_t* perf_mask, long long* perf_len, long long* masked_perf, long long* labels, long long* bar,
                                 long long* beat, long long* onset, uint8_t* deadpan_mask, hipStream_t stream) {
    SPN_REQUIRE(score_flat && perf_flat && score_off && perf_off && b > 0 && Ks > 0 && Kp > 0 && Ls > 0 && Lp > 0, "spn_collate_mixlm: bad inputs");
    SPN_REQUIRE(Kp <= 32 && n_ignore >= 0 && n_ignore <= 16 && (n_ignore == 0 || ignore_ids), "spn_collate_mixlm: <= 32 token dims, <= 16 ignored ids");
    SPN_REQUIRE(score && score_mask && score_len && perf && perf_mask && perf_len && masked_perf && labels && deadpan_mask,
                "spn_collate_mixlm: null output");
    SPN_REQUIRE(!seg_flat || (bar && beat && onset), "spn_collate_mixlm: segment outputs required with segment inputs");
    CollateArgs a;
    memset(&a, 0, sizeof(a));
    a.score_flat = score_flat; a.perf_flat = perf_flat; a.seg_flat = seg_flat; a.score_off = score_off; a.perf_off = perf_off; a.deadpan = deadpan;
    a.b = b; a.Ks = Ks; a.Kp = Kp; a.Ls = Ls; a.Lp = Lp; a.sum_s = sum_s;
    a.pad_id = pad_id; a.mask_id = mask_id; a.label_pad_id = label_pad_id; a.n_ignore = n_ignore; a.label_pad_ignored_dims = label_pad_ignored_dims;
    for (int i = 0; i < n_ignore; ++i) a.ignore_ids[i] = ignore_ids[i];
    a.ignore_dims = ignore_dims;
    a.score = score; a.score_mask = score_mask; a.score_len = score_len; a.perf = perf; a.perf_mask = perf_mask; a.perf_len = perf_len;
    a.masked_perf = masked_perf; a.labels = labels; a.bar = bar; a.beat = beat; a.onset = onset; a.deadpan_mask = deadpan_mask;
    const bool pair = Ks % 2 == 0 && Kp % 2 == 0 && ((uintptr_t)score_flat | (uintptr_t)perf_flat) % 8 == 0 &&
                      ((uintptr_t)score | (uintptr_t)perf | (uintptr_t)masked_perf | (uintptr_t)labels) % 16 == 0;
    const int W = pair ? 2 : 1;
    const long total = (long)b * Lp * (Kp / W) + (long)b * Ls * (Ks / W) + (seg_flat ? (long)b * Ls : 0);
    long g = (total + 255) / 256;
    if (g > 16384) g = 16384;
    const bool small = total < (1l << 32) - (1l << 23);
    if (small && pair) hipLaunchKernelGGL((collate_mixlm_kernel<unsigned, 2>), dim3((unsigned)g), dim3(256), 0, stream, a);
    else if (small) hipLaunchKernelGGL((collate_mixlm_kernel<unsigned, 1>), dim3((unsigned)g), dim3(256), 0, stream, a);
    else if (pair) hipLaunchKernelGGL((collate_mixlm_kernel<long, 2>), dim3((unsigned)g), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((collate_mixlm_kernel<long, 1>), dim3((unsigned)g), dim3(256), 0, stream, a);
    SPN_LAUNCH_CHECK();
    return SPN_OK;
}
