// Single-query attention over ONE lane group's share of a key split, shared by dec_attn2_kernel (decode.hip) and the persistent layer
// launch (decode_layer.hip) so that both round identically (both files compile without floating-point contraction).
//
// A lane group (16 lanes, 4 dims each) owns the keys j = jb0 + G u, u = 0..NU-1, below j1 (G = 256 / NU lane groups per workgroup: a
// workgroup covers 256 consecutive keys per batch).  The batch is scored FIRST (NU independent dot products + DPP row sums), then one
// maximum, NU independent exponentials and the sums in u order, and the batch joins the running (max, normaliser, weighted values) with
// one rescale.  The key-by-key online update it replaces (softmax state rescaled after every key: a chain of ~25 dependent instructions
// per key) took 3 us for 16 keys on the critical path of a note; this form ~0.6 us.
// Round 5: NU = 8 on EIGHT waves (32 lane groups) instead of 16 on four: a wave issues one instruction per ~5 clocks whatever its
// neighbour on the SIMD does (tools/pk_probe.hip), the batch is ~45 instructions per key, and the other four waves of the persistent
// launch's workgroups had nothing to do during this phase.  Both decode paths use the same NU (the grouping is part of the arithmetic).
// attention.py:162-197 / attend.py:58-126 semantics: scores q.k * scale - slope * (t - j), softmax over the prefix.
#pragma once
#include "common.h"

#pragma clang fp contract(off)   // file scope, from here to the end of the including file: see decode.hip

// k4 / v4 are modified: the row of key t (the note being decoded: not in the cache yet) is patched in from knew4 / vnew4 first.
constexpr int DEC_NU = 8;             // keys per lane group and batch
constexpr int DEC_G = 256 / DEC_NU;   // lane groups per workgroup = key stride inside a lane group
template <int NU>
__device__ __forceinline__ void dec_attn_batch(f32x4 (&k4)[NU], f32x4 (&v4)[NU], const f32x4 knew4, const f32x4 vnew4,
                                               const f32x4 q4, float slope, int t, int jb0, int j1, float& m, float& l, f32x4& acc) {
    constexpr int G = 256 / NU;
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const bool is_new = jb0 + G * u == t;
        k4[u] = is_new ? knew4 : k4[u];
        v4[u] = is_new ? vnew4 : v4[u];
    }
    float sc[NU];
    float mb = -INFINITY;
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int j = jb0 + G * u;
        f32x2 s2 = f32x2{0.f, 0.f};
        dec_dot4(s2, q4, k4[u]);                // common.h: packed FMA, two partial sums
        float s = row16_sum(dec_fold(s2));      // the 16 lanes of a key: one DPP row
        s -= slope * (float)(t - j);
        sc[u] = j < j1 ? s : -INFINITY;
        mb = fmaxf(mb, sc[u]);
    }
    float lb = 0.f;
    f32x2 ab_lo = f32x2{0.f, 0.f}, ab_hi = f32x2{0.f, 0.f};
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const float p = jb0 + G * u < j1 ? __expf(sc[u] - mb) : 0.f;
        lb += p;
        ab_lo = __builtin_elementwise_fma(f32x2{v4[u][0], v4[u][1]}, f32x2{p, p}, ab_lo);   // two v_pk_fma_f32 per key
        ab_hi = __builtin_elementwise_fma(f32x2{v4[u][2], v4[u][3]}, f32x2{p, p}, ab_hi);
    }
    const f32x4 ab = f32x4{ab_lo[0], ab_lo[1], ab_hi[0], ab_hi[1]};
    const float m_new = fmaxf(m, mb);
    const float a_old = __expf(m - m_new), a_b = __expf(mb - m_new);   // m = -inf at the first batch: a_old = 0, a_b = 1
    l = l * a_old + lb * a_b;
    acc = acc * a_old + ab * a_b;
    m = m_new;
}

// ---- the workgroup's result from its DEC_G lane groups, in two levels (shared by both decode paths: the grouping is part of the arithmetic) ----
// Level 1: wave w folds ITS four lane groups (records 4 w .. 4 w + 3 of sm / sl / so, which its own lanes just wrote: LDS serves a wave's
// accesses in order, no barrier) into record w of sm2 / sl2 / so2, lane = value column.  Level 2 (behind ONE barrier): a whole wave folds
// the eight wave records.  One serial pass over all 32 groups was 32 dependent (read, scale, add) steps on one wave -- longer than the
// scoring of the keys it follows; 4 + 8 steps are not.  A group / wave without keys has m = -inf and weight 0.
__device__ __forceinline__ void dec_attn_merge_wave(const float* sm, const float* sl, const float (*so)[64], float* sm2, float* sl2,
                                                    float (*so2)[64], int w, int lane) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    float mm = -INFINITY;
#pragma unroll
    for (int g = 0; g < 4; ++g) mm = fmaxf(mm, sm[4 * w + g]);
    const float mg = sm[4 * w + (lane & 3)];
    const float fl = (mg == -INFINITY) ? 0.f : __expf(mg - mm);   // lane g < 4: the weight of group g
    float num = 0.f, den = 0.f;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float f = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, fl), g));
        num += so[4 * w + g][lane] * f; den += sl[4 * w + g] * f;
    }
    so2[w][lane] = num;
    if (lane == 0) { sm2[w] = mm; sl2[w] = den; }
}
// Level 2, by every lane of a wave (col = the lane's value column): mm / den are the same in all lanes.
__device__ __forceinline__ void dec_attn_merge_block(const float* sm2, const float* sl2, const float (*so2)[64], int col, int lane,
                                                     float& mm, float& num, float& den) {
    mm = -INFINITY;
#pragma unroll
    for (int q = 0; q < 8; ++q) mm = fmaxf(mm, sm2[q]);
    const float mq = sm2[lane & 7];
    const float fl = (mq == -INFINITY) ? 0.f : __expf(mq - mm);   // lane q < 8: the weight of wave record q
    num = 0.f; den = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const float f = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, fl), q));
        num += so2[q][col] * f; den += sl2[q] * f;
    }
}
static_assert(DEC_G == 32, "the two-level merge is written for eight waves of four lane groups");
