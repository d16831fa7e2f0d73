// Single-query attention over ONE lane group's share of a key split, shared by dec_attn2_kernel (decode.hip) and the persistent layer
// launch (decode_layer.hip) so that both round identically (both files compile without floating-point contraction).
//
// A lane group (16 lanes, 4 dims each) owns the keys j = jb0 + 16 u, u = 0..15, below j1.  The batch is scored FIRST (16 independent dot
// products + DPP row sums), then one maximum, 16 independent exponentials and the sums in u order, and the batch joins the running
// (max, normaliser, weighted values) with one rescale.  The key-by-key online update it replaces (softmax state rescaled after every
// key: a chain of ~25 dependent instructions per key) took 3 us for 16 keys on the critical path of a note; this form ~0.6 us.
// attention.py:162-197 / attend.py:58-126 semantics: scores q.k * scale - slope * (t - j), softmax over the prefix.
#pragma once
#include "common.h"

#pragma clang fp contract(off)   // file scope, from here to the end of the including file: see decode.hip

// k4 / v4 are modified: the row of key t (the note being decoded: not in the cache yet) is patched in from knew4 / vnew4 first.
__device__ __forceinline__ void dec_attn_batch16(f32x4 (&k4)[16], f32x4 (&v4)[16], const f32x4 knew4, const f32x4 vnew4,
                                                 const f32x4 q4, float slope, int t, int jb0, int j1, float& m, float& l, f32x4& acc) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const bool is_new = jb0 + 16 * u == t;
        k4[u] = is_new ? knew4 : k4[u];
        v4[u] = is_new ? vnew4 : v4[u];
    }
    float sc[16];
    float mb = -INFINITY;
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const int j = jb0 + 16 * u;
        float s = q4[0] * k4[u][0] + q4[1] * k4[u][1] + q4[2] * k4[u][2] + q4[3] * k4[u][3];
        s = row16_sum(s);   // the 16 lanes of a key: one DPP row
        s -= slope * (float)(t - j);
        sc[u] = j < j1 ? s : -INFINITY;
        mb = fmaxf(mb, sc[u]);
    }
    float lb = 0.f;
    f32x4 ab = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const float p = jb0 + 16 * u < j1 ? __expf(sc[u] - mb) : 0.f;
        lb += p;
        ab += v4[u] * p;
    }
    const float m_new = fmaxf(m, mb);
    const float a_old = __expf(m - m_new), a_b = __expf(mb - m_new);   // m = -inf at the first batch: a_old = 0, a_b = 1
    l = l * a_old + lb * a_b;
    acc = acc * a_old + ab * a_b;
    m = m_new;
}
