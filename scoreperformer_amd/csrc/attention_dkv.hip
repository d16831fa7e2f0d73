// dK / dV of the fused attention (see attention.hip).  grid (ceil(nk/128), kv_heads, b), 256 threads.
// Wave w owns 32 keys (two 16-key blocks) j0 + 32w .. +31 and loops over all query tiles of all heads that share its K/V head
// (multi-query attention: 8 heads), so dK/dV are reduced over heads in registers, without atomics.
// Orientation S = Q K^T (lane = one key column): P and dS in C-layout are the B operands of dV^T = dO^T P and dK^T = Q^T dS.
// 32 keys per wave halve the LDS fragment traffic per MFMA relative to 16 (each Q / dO fragment feeds two key blocks).
#include "attention_common.h"

namespace spn_attn {
namespace {

// P' (= P / keep_prob) masked by keep bit `bit` of `word`
__device__ __forceinline__ float keep_bits(float p, uint32_t word, int bit) {
    int m = __builtin_amdgcn_sbfe((int)word, bit, 1);
    asm("" : "+v"(m));   // opaque: v_bfe_i32 + v_and_b32; knowing that m is a sign-extended bit hipcc emits v_and + v_cmp + v_cndmask
    return __int_as_float(__float_as_int(p) & m);
}

// One (head, 64-row query tile) against this wave's 32 keys.  Row statistics come from LDS, staged once per tile by the block:
//   rc_s  [64]: LINEAR tiles -- the whole row part of the exponent, (-lse2_i + log2(1 / keep_prob) -+ slope2 i) / c1, in the variant of
//               this tile's side (left: -, right: +): it is the C INPUT of the score MFMA, so a score costs one fma (acc * c1 +- slope2 j)
//               before its exp; GENERAL tiles (diagonal / masked / ragged) -- n_s = -lse2_i + log2(1 / keep_prob)
//   dl_s  [64]: delta_i * keep_prob
template <int CLS, bool DROP>
__device__ __forceinline__ void dkv_tile(const char* q_tile, const char* qt_tile, const char* do_tile, const char* dot_tile,
                                         const float* rc_s, const float* dl_s, const bf16x8 (&kf)[2][2], const bf16x8 (&vf)[2][2],
                                         f32x4 (&dk)[4][2], f32x4 (&dv)[4][2], const float (&jf)[2], const bool (&key_ok)[2],
                                         int ioff, float c1, float slope2, bool causal, int lane, int g,
                                         const char* bits_lane, int boff0) {
    const float sgn = CLS == T_LEFT ? 1.f : -1.f;
    const float ekey[2] = {sgn * slope2 * jf[0], sgn * slope2 * jf[1]};
#pragma unroll
    for (int u = 0; u < 2; ++u) {               // two halves of 32 query rows
        uint2 pp[2][2], dd[2][2];               // [qq][kb]: P' and dS of 4 rows x this lane's key, rounded to bf16 as soon as they exist
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) {
            const int qb = 2 * u + qq;
            const bf16x8 qa0 = frag_rows(q_tile, 16 * qb, 0, lane), qa1 = frag_rows(q_tile, 16 * qb, 1, lane);
            const bf16x8 da0 = frag_rows(do_tile, 16 * qb, 0, lane), da1 = frag_rows(do_tile, 16 * qb, 1, lane);
            // dropout: P' = P / keep_prob (the log2 shift rides on the row term), dS = P' * (keep * dP - delta * keep_prob)
            const f32x4 n4 = *reinterpret_cast<const f32x4*>(rc_s + 16 * qb + 4 * g);
            const f32x4 d4 = *reinterpret_cast<const f32x4*>(dl_s + 16 * qb + 4 * g);
            // keep bits of query block qb against this lane's key column: 4 halfwords (rows r) of the forward's words, from the LDS copy
            uint2 bwq = make_uint2(0u, 0u);
            if (DROP) {   // shifted once so that the 8 extractions below take immediate bit positions (no position registers)
                bwq = *reinterpret_cast<const uint2*>(bits_lane + 256 * qb);
                bwq.x >>= boff0; bwq.y >>= boff0;
            }
            const float ib = (float)(ioff + 16 * qb + 4 * g);
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
                f32x4 acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa0, kf[kb][0], CLS == T_GEN ? zero : n4, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa1, kf[kb][1], acc, 0, 0, 0);
                f32x4 acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(da0, vf[kb][0], zero, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(da1, vf[kb][1], acc2, 0, 0, 0);
                float pq[4], dq_[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float e;   // log2-domain score minus lse
                    if (CLS != T_GEN) e = fmaf(acc[r], c1, ekey[kb]);
                    else {
                        const float i_f = ib + (float)r;
                        const bool ok = key_ok[kb] && (!causal || jf[kb] <= i_f);
                        const float t = fmaf(-slope2, fabsf(jf[kb] - i_f), acc[r] * c1);
                        e = (ok ? t : NEG_FILL) + n4[r];
                    }
                    const float pv = fast_exp2(e);
                    if (DROP) {   // row r of this lane's key column: halfword r of bwq, bit boff0 + 4 kb (see attention_common.h)
                        const float pm = keep_bits(pv, r < 2 ? bwq.x : bwq.y, 4 * kb + 16 * (r & 1));          // P_dropped feeds dV
                        pq[r] = pm;
                        dq_[r] = fmaf(pm, acc2[r], -(pv * d4[r]));   // = pv * (keep ? dP : 0) - pv * delta: one AND less
                    } else {
                        pq[r] = pv;
                        dq_[r] = pv * (acc2[r] - d4[r]);
                    }
                }
                pp[qq][kb] = make_uint2(pack_bf2(pq[0], pq[1]), pack_bf2(pq[2], pq[3]));
                dd[qq][kb] = make_uint2(pack_bf2(dq_[0], dq_[1]), pack_bf2(dq_[2], dq_[3]));
            }
        }
        bf16x8 pf[2], dsf[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            pf[kb] = __builtin_bit_cast(bf16x8, make_uint4(pp[0][kb].x, pp[0][kb].y, pp[1][kb].x, pp[1][kb].y));
            dsf[kb] = __builtin_bit_cast(bf16x8, make_uint4(dd[0][kb].x, dd[0][kb].y, dd[1][kb].x, dd[1][kb].y));
        }
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            const bf16x8 dot = frag_cols_t(dot_tile, 16 * db, u, lane);
            const bf16x8 qt = frag_cols_t(qt_tile, 16 * db, u, lane);
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                dv[db][kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dot, pf[kb], dv[db][kb], 0, 0, 0);
                dk[db][kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qt, dsf[kb], dk[db][kb], 0, 0, 0);
            }
        }
    }
}

template <bool DROP>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_kernel(AttnArgs a) {
    // per (head, query tile): Q "a" (A operand of S), Q "t" (Q^T, A operand of dK^T), dO "a" (A operand of dP), dO "t" (dO^T, A operand
    // of dV^T) -- four 8 KiB images by LDS DMA into two alternating stages, one barrier per tile (see attn_fwd_kernel); the XOR swizzles
    // sit on the source address, rows past nq lie outside the buffer resource and read as zero
    __shared__ __attribute__((aligned(16))) char smem[2 * 32768 + 2 * 1024 + 2 * 1024];
    // [2][row term, left variant | right variant | general variant | delta * keep_prob] of the tile's 64 rows (dkv_tile); dead rows: -1e30 (p = 0)
    float* stat_s = reinterpret_cast<float*>(smem + 65536);
    // [2][4 query blocks of 16][128 forward words]: the keep bits of (64 rows x the block's 128 keys), 1 KiB per tile, fetched ONCE per
    // block by LDS DMA (wave w: query block w) -- read straight into registers, each lane of each wave fetched 32 bytes to use 8 bits
    // of them (8x the buffer through the L2 per launch) and held the next tile's words in 8 registers across the whole tile
    char* bits_s = smem + 65536 + 2048;

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g = lane >> 4, c = lane & 15;
    int bi = blockIdx.z, kh = blockIdx.y, jt = blockIdx.x;
    if (!causal_order(a, false, bi, kh, jt) && !a.causal) xcd_batch_coords(a, bi, kh, jt);
    const int j0 = jt * 128;
    const int off = a.nk - a.nq;
    const int heads_per_kv = a.h / a.kvh;
    const bf16_t* kp = a.k + bi * a.k_bs + kh * a.k_hs;
    const bf16_t* vp = a.v + bi * a.v_bs + kh * a.v_hs;
    const float c1 = a.scale * LOG2E, inv_c1 = 1.f / c1;
    int jcol[2];
    float jf[2];
    bool key_ok[2];
    bf16x8 kf[2][2], vf[2][2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        jcol[kb] = j0 + 32 * w + 16 * kb + c;
        jf[kb] = (float)jcol[kb];
        key_ok[kb] = (jcol[kb] < a.nk) && (a.kmask ? a.kmask[(long)bi * a.nk + jcol[kb]] != 0 : true);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            kf[kb][ks] = load_row_frag(kp, a.k_ns, jcol[kb], a.nk, ks, lane);
            vf[kb][ks] = load_row_frag(vp, a.v_ns, jcol[kb], a.nk, ks, lane);
        }
    }
    const bool keys_full = __all(key_ok[0] && key_ok[1]);
    const bool keys_none = !__any(key_ok[0] || key_ok[1]);   // this wave's 32 keys are all masked: P = 0 for every live row, dK = dV = 0
    const int wv = __builtin_amdgcn_readfirstlane(w);   // the wave index as a scalar: tile classes must be wave-uniform FOR THE COMPILER
    const int jw_lo = j0 + 32 * wv, jw_hi = jw_lo + 31;

    f32x4 dk[4][2], dv[4][2];
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) { dk[db][kb] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[db][kb] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    const int nqt = (a.nq + 63) / 64;
    int t_first = 0;
    if (a.causal) {  // query i sees key j iff j <= i + off  ->  first useful row i = j0 - off
        const int i_min = j0 - off;
        t_first = i_min <= 0 ? 0 : i_min / 64;
    }
    // a block whose 128 keys are all masked (the padding of a ragged batch) walks nothing and stores zeros
    const int n_iter = ((nqt - t_first) > 0 && !__syncthreads_and(keys_none)) ? (nqt - t_first) * heads_per_kv : 0;

    // keep bits of the forward: this lane's key column c of key block kb sits in forward lanes (c>>2)*16 + 4g + r (r = its 4 rows),
    // bit 4*kb_f + (c&3) with kb_f = the 16-key block index inside the forward's 64-key tile
    const long bstride = (long)a.nkt64 * 64;
    const int boff0 = 8 * (w & 1) + (c & 3);   // + 4 kb: bit of this lane's key column inside a forward word's halfword
    const int bit_lane_bytes = ((w >> 1) * 64 + (c >> 2) * 16 + 4 * g) * 2;   // this lane's 4 halfwords inside a query block's 128 words
    const float log2_inv_keep = DROP ? __builtin_log2f(a.inv_keep) : 0.f, keep_prob = DROP ? 1.f / a.inv_keep : 1.f;

    // this wave's two 1 KiB pieces of each image: LDS chunk <- inverse-swizzled source chunk.  The second piece lies 8 rows further
    // down with the same swizzle (both keys repeat every 8 rows), so it shares the lane offset and takes 8 rows on the SCALAR offset
    uint32_t voQa, voQt, voDa, voDt;
    {
        const int L = wv * 128 + lane, row = L >> 3, ch = L & 7;
        const int ca = (ch ^ (row & 7)) << 3, ct = (ch ^ (((row >> 1) & 3) << 1)) << 3;
        voQa = (uint32_t)(((long)row * a.q_ns + ca) * 2); voQt = (uint32_t)(((long)row * a.q_ns + ct) * 2);
        voDa = (uint32_t)(((long)row * a.o_ns + ca) * 2); voDt = (uint32_t)(((long)row * a.o_ns + ct) * 2);
    }
    float lreg = 0.f, dreg = 0.f;   // raw lse / delta of the requested tile's row (threads 0..63): two registers across the tile, not four
    int n_issued = 0;   // stage of the next issue = n_issued & 1
    auto issue = [&](int it) {
        // pointers and strides of the tile request: read from the kernel-argument segment here, once per tile (attention_common.h)
        AttnKernargPtr ai = attn_kernarg();
        asm volatile("" : "+s"(ai));
        const int hh = kh * heads_per_kv + it / (nqt - t_first);
        const int i0 = (t_first + it % (nqt - t_first)) * 64;
        if (DROP) {
            // query block wv of the tile: words [kt64 = j0 / 64, j0 / 64 + 1][64 lanes] are contiguous (256 bytes = one DMA instruction)
            const uint16_t* hb = ai->dropbits + (long)(bi * ai->h + hh) * ai->nqt16 * bstride;
            const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)hb, 0, (int)((long)ai->nqt16 * bstride * 2), 0x00020000);
            const uint32_t sb = (uint32_t)(((long)(i0 / 16 + wv) * bstride + (long)(j0 / 64) * 64) * 2);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (__attribute__((address_space(3))) void*)(bits_s + (n_issued & 1) * 1024 + wv * 256), 4,
                                                     (uint32_t)(lane * 4), sb, 0, 0);
        }
        const long q_ns = ai->q_ns, o_ns = ai->o_ns;
        const int nq = ai->nq;
        const __amdgpu_buffer_rsrc_t rsQ = __builtin_amdgcn_make_buffer_rsrc((void*)(ai->q + bi * ai->q_bs + hh * ai->q_hs), 0,
                                                                              (int)(((long)(nq - 1) * q_ns + 64) * 2), 0x00020000);
        const __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc((void*)(ai->d_o + bi * ai->o_bs + hh * ai->o_hs), 0,
                                                                              (int)(((long)(nq - 1) * o_ns + 64) * 2), 0x00020000);
        char* base = smem + (n_issued & 1) * 32768 + wv * 2048;
        const uint32_t sq = (uint32_t)i0 * (uint32_t)q_ns * 2u, sd = (uint32_t)i0 * (uint32_t)o_ns * 2u;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const uint32_t sqi = sq + (uint32_t)(i * 8) * (uint32_t)q_ns * 2u, sdi = sd + (uint32_t)(i * 8) * (uint32_t)o_ns * 2u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsQ, (__attribute__((address_space(3))) void*)(base + i * 1024), 16, voQa, sqi, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsQ, (__attribute__((address_space(3))) void*)(base + 8192 + i * 1024), 16, voQt, sqi, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsD, (__attribute__((address_space(3))) void*)(base + 16384 + i * 1024), 16, voDa, sdi, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsD, (__attribute__((address_space(3))) void*)(base + 24576 + i * 1024), 16, voDt, sdi, 0, 0);
        }
        ++n_issued;
        if (tid < 64) {
            const int i = i0 + tid;
            const long si = ((long)bi * ai->h + hh) * nq + i;
            lreg = i < nq ? ai->lse[si] : NEG_FILL;
            dreg = i < nq ? ai->delta[si] : 0.f;
        }
    };
    // ALiBi band (attention.hip): a (head, query tile) whose reach ends before this block's 128 keys is not visited.  Which iterations
    // are live is decided ONCE, by all 256 threads in parallel, into a bit mask in LDS: decided tile by tile inside the loop, the two
    // dependent global loads of band_reach (and ~110 scalar / vector instructions around them) sat in front of every tile request.
    constexpr int LIVE_WORDS = 16;   // up to 1024 (head, query tile) iterations; longer walks decide on the fly as before
    __shared__ unsigned long long live_mask[LIVE_WORDS];
    // ... and the same question per WAVE (its 32 keys instead of the block's 128): the block walks the union, a wave sits out the
    // iterations only its neighbours need (steep heads: a third of them)
    __shared__ unsigned long long live_wave[4][LIVE_WORDS];
    auto reach_of = [&](int it, float& r_lo) {
        const int hh = kh * heads_per_kv + it / (nqt - t_first);
        const int i0 = (t_first + it % (nqt - t_first)) * 64;
        r_lo = (float)(i0 + off);
        const float reach = band_reach(a, bi, hh, kh, i0 / 64, 1, c1, a.slopes ? a.slopes[hh] * LOG2E : 0.f);
        return reach < 1.0e9f ? reach : 3.0e38f;
    };
    auto is_live = [&](int it) {
        float r_lo;
        const float reach = reach_of(it, r_lo);
        return (float)j0 <= r_lo + 63.f + reach && (float)(j0 + 127) >= r_lo - reach;
    };
    const bool masked_walk = n_iter <= 64 * LIVE_WORDS;
    // query tiles of padding rows only (the forward's qmask: zero output rows, lse = NEG_FILL) are not requested at all: one bit per
    // 64-row tile, wave by wave from the tile's 64 mask bytes.  (Padding rows inside a live tile are dead through their lse.)
    __shared__ unsigned long long qtile_live[LIVE_WORDS];
    __shared__ int wave_keys_none[4];   // a wave whose 32 keys are all masked sits out every tile: its live_wave bits stay zero
    const bool q_tiles_known = a.qmask && nqt <= 64 * LIVE_WORDS && n_iter > 0;
    if (n_iter > 0) {
        if (tid < LIVE_WORDS) qtile_live[tid] = 0ull;
        if (lane == 0) wave_keys_none[w] = keys_none ? 1 : 0;
        __syncthreads();
        if (q_tiles_known) {   // 16 mask bytes per thread, one trip for up to 4096 rows
            const uint8_t* qm = a.qmask + (long)bi * a.nq;
            const bool vec = ((reinterpret_cast<uintptr_t>(qm) | (uintptr_t)a.nq) & 15) == 0;
            for (int ch = t_first * 4 + tid; ch * 16 < a.nq; ch += 256) {
                bool any = false;
                if (vec) {
                    const uint4 x = *reinterpret_cast<const uint4*>(qm + ch * 16);
                    any = (x.x | x.y | x.z | x.w) != 0u;
                } else {
                    for (int e = 0; e < 16; ++e) any |= ch * 16 + e < a.nq && qm[ch * 16 + e] != 0;
                }
                if (any) atomicOr(&qtile_live[ch >> 8], 1ull << ((ch >> 2) & 63));
            }
        }
        __syncthreads();
    }
    auto qtile_ok = [&](int it) {
        if (!q_tiles_known) return true;
        const int tq = t_first + it % (nqt - t_first);
        return ((qtile_live[tq >> 6] >> (tq & 63)) & 1ull) != 0ull;
    };
    if (masked_walk) {
        for (int base = 0; base < n_iter; base += 256) {
            const int it = base + tid;
            float r_lo = 0.f;
            const float reach = it < n_iter ? reach_of(it, r_lo) : -1.f;
            const bool in = it < n_iter && qtile_ok(it);
            const unsigned long long m = __ballot(in && (float)j0 <= r_lo + 63.f + reach && (float)(j0 + 127) >= r_lo - reach);
            if (lane == 0) live_mask[(base >> 6) + w] = m;
#pragma unroll
            for (int wq = 0; wq < 4; ++wq) {
                const float jl = (float)(j0 + 32 * wq);
                const unsigned long long mw = __ballot(in && !wave_keys_none[wq] && jl <= r_lo + 63.f + reach && jl + 31.f >= r_lo - reach);
                if (lane == 0) live_wave[wq][(base >> 6) + w] = mw;
            }
        }
        __syncthreads();
    }
    auto next_live = [&](int it) {
        if (masked_walk) {
            while (it < n_iter) {
                const unsigned long long word = live_mask[it >> 6] >> (it & 63);
                if (word) { it += __builtin_ctzll(word); break; }
                it = (it | 63) + 1;
            }
            return min(it, n_iter);
        }
        for (; it < n_iter; ++it)
            if (qtile_ok(it) && is_live(it)) break;
        return it;
    };
    int it = next_live(0);
    if (it < n_iter) issue(it);
    while (it < n_iter) {
        const int it_next = next_live(it + 1);
        const int hh = kh * heads_per_kv + it / (nqt - t_first);
        const int i0 = (t_first + it % (nqt - t_first)) * 64;
        const float slope2 = a.slopes ? a.slopes[hh] * LOG2E : 0.f;
        const int stage = (n_issued - 1) & 1;   // the stage the tile of THIS iteration was requested into
        const char* q_tile = smem + stage * 32768;
        const char* qt_tile = q_tile + 8192;
        const char* do_tile = q_tile + 16384;
        const char* dot_tile = q_tile + 24576;
        float* rcL_s = stat_s + stage * 256;
        float* dl_s = rcL_s + 192;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces, the row statistics and the keep-bit words have landed
        if (tid < 64) {
            const bool live = lreg > -1e37f;   // rows beyond nq / fully masked rows: p = exp2(-huge) = 0
            const float nl = log2_inv_keep - lreg * LOG2E;
            const float lin = slope2 * (float)(i0 + tid + off);
            rcL_s[tid] = live ? (nl - lin) * inv_c1 : -1e30f;
            rcL_s[64 + tid] = live ? (nl + lin) * inv_c1 : -1e30f;
            rcL_s[128 + tid] = live ? nl : NEG_FILL;
            dl_s[tid] = dreg * keep_prob;
        }
        const char* bits_lane = bits_s + stage * 1024 + bit_lane_bytes;
        __syncthreads();   // tile visible to all; everybody is done with the other stage
        if (it_next < n_iter) issue(it_next);

        // tile class of this wave's 32 keys against the 64 rows (key coordinates i + off)
        const int r_lo = i0 + off, r_hi = r_lo + 63;
        int cls = T_GEN;
        if (a.causal && jw_lo > r_hi) cls = T_SKIP;
        else if (masked_walk && !((__builtin_amdgcn_readfirstlane((int)(live_wave[wv][it >> 6] >> (it & 63))) & 1))) cls = T_SKIP;   // outside this wave's own band
        else if (keys_full && jw_hi <= r_lo) cls = T_LEFT;              // j - i <= 0 everywhere
        else if (keys_full && !a.causal && jw_lo >= r_hi) cls = T_RIGHT;
        if (cls == T_SKIP) { it = it_next; continue; }

        if (cls == T_LEFT) dkv_tile<T_LEFT, DROP>(q_tile, qt_tile, do_tile, dot_tile, rcL_s, dl_s, kf, vf, dk, dv, jf, key_ok, i0 + off, c1, slope2, a.causal, lane, g, bits_lane, boff0);
        else if (cls == T_RIGHT) dkv_tile<T_RIGHT, DROP>(q_tile, qt_tile, do_tile, dot_tile, rcL_s + 64, dl_s, kf, vf, dk, dv, jf, key_ok, i0 + off, c1, slope2, a.causal, lane, g, bits_lane, boff0);
        else dkv_tile<T_GEN, DROP>(q_tile, qt_tile, do_tile, dot_tile, rcL_s + 128, dl_s, kf, vf, dk, dv, jf, key_ok, i0 + off, c1, slope2, a.causal, lane, g, bits_lane, boff0);
        it = it_next;
    }
    AttnKernargPtr ae = attn_kernarg();   // the output pointers and strides are not held in registers across the tile loop
    asm volatile("" : "+s"(ae));
    const float out_scale = ae->scale;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        if (jcol[kb] >= ae->nk) continue;
        bf16_t* pk_ = ae->dk + bi * ae->dk_bs + (long)jcol[kb] * ae->dk_ns + kh * ae->dk_hs;
        bf16_t* pv_ = ae->dv + bi * ae->dv_bs + (long)jcol[kb] * ae->dv_ns + kh * ae->dv_hs;
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            uint2 x, y;
            x.x = pack_bf2(dk[db][kb][0] * out_scale, dk[db][kb][1] * out_scale);
            x.y = pack_bf2(dk[db][kb][2] * out_scale, dk[db][kb][3] * out_scale);
            y.x = pack_bf2(dv[db][kb][0], dv[db][kb][1]); y.y = pack_bf2(dv[db][kb][2], dv[db][kb][3]);
            *reinterpret_cast<uint2*>(pk_ + 16 * db + 4 * g) = x;
            *reinterpret_cast<uint2*>(pv_ + 16 * db + 4 * g) = y;
        }
    }
}

}  // namespace

int launch_attn_dkv(const AttnArgs& a, hipStream_t stream) {
    if (a.drop_on) hipLaunchKernelGGL(attn_bwd_dkv_kernel<true>, dim3(cdiv(a.nk, 128), a.kvh, a.b), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(attn_bwd_dkv_kernel<false>, dim3(cdiv(a.nk, 128), a.kvh, a.b), dim3(256), 0, stream, a);
    return 0;
}

}  // namespace spn_attn
