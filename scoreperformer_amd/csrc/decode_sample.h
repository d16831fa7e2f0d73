// The multinomial draw of the sampling LM head, shared by dec_head_kernel<true> (decode.hip, spn_dec_head_sample) and the head phase of the
// persistent launch (decode_layer.hip): ONE definition of how the kept weights are summed, so that both paths draw the same token.
//
// wg[0 .. V) in LDS: the weight of every id (0 = filtered out).  Lane l of the calling wave owns the ids [l C, (l + 1) C), C = ceil(V / 64):
// it adds its ids in order; the lanes' sums are then added in lane order to the total and -- with target = u01 * total -- walked again in
// lane order to the lane whose range holds the target, whose ids are walked in order to the first one that takes the running sum past the
// target.  (The form this replaces walked all V ids on one thread: V dependent LDS reads, ~7 us at V = 165 on the critical path of a note.)
// Every lane of the wave must call it (wave-uniform arguments); returns the picked id in every lane.  V <= 1024.
#pragma once
#include "common.h"

__device__ __forceinline__ int dec_sample_pick(const float* wg, int V, float u01, int lane) {
    // no FMA contraction in the sums below, whatever the including file's setting: both callers must add the same values in the same way
    // (scoped to this function; decode.hip / decode_layer.hip set their own file-wide pragma explicitly)
#pragma clang fp contract(off)
    const int C = (V + 63) >> 6;
    float s = 0.f;
    int last_pos = -1;                      // the last id with a positive weight in this lane's range
    for (int c = 0; c < C; ++c) {
        const int id = lane * C + c;
        const float wv = id < V ? wg[id] : 0.f;
        s += wv;
        if (wv > 0.f) last_pos = id;
    }
    float total = 0.f;
    for (int l = 0; l < 64; ++l) total += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, s), l));
    const float target = u01 * total;
    float run = 0.f, before = 0.f;
    int pl = -1;
    for (int l = 0; l < 64; ++l) {
        const float sl = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, s), l));
        if (pl < 0 && run + sl > target) { pl = l; before = run; }
        run += sl;
    }
    // the overall last positive id: the fallback when rounding leaves the target at or past the total
    int lp = last_pos;
    for (int o = 32; o > 0; o >>= 1) lp = max(lp, __shfl_xor(lp, o));
    if (pl < 0) return max(lp, 0);
    float cum = before;
    int pick = -1, lastpos = 0;
    for (int c = 0; c < C; ++c) {
        const int id = pl * C + c;
        const float wv = id < V ? wg[id] : 0.f;
        if (wv > 0.f) { lastpos = id; cum += wv; if (pick < 0 && cum > target) pick = id; }
    }
    return pick >= 0 ? pick : lastpos;
}
