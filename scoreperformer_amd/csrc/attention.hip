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
#include "attention_common.h"

namespace spn_attn { int launch_attn_dkv(const AttnArgs& a, hipStream_t stream); }

namespace {
using namespace spn_attn;

// ==========================================================================================================
// forward: grid (ceil(nq/128), h, b), 256 threads; wave w owns query rows q0 + 32w .. +31
// ==========================================================================================================
// Dropout of one probability with NO mask register: `word` holds the keep bits of the scores still to come, the next one in bit 31.
// v_add_co_u32 shifts the word left by one and leaves the bit that falls out, for all 64 lanes, in VCC; v_cndmask_b32 applies it.  Two
// VALU slots per score like v_bfe_i32 + v_and_b32, but no live mask per score (hipcc hoists all 32 extractions of a tile to its top:
// +16 registers, which costs this kernel its third wave per SIMD), and no v_and + v_cmp + v_cndmask (what hipcc makes of a bit test).
// Both instructions sit in one asm statement, the add first: p comes out of v_exp_f32 and a transcendental result may not be read by the
// very next VALU instruction -- hipcc pads nothing in front of an asm statement, the add (which does not read p) is that padding.
__device__ __forceinline__ float drop_next(float p, uint32_t& word) {
    float r;
    asm("v_add_co_u32 %1, vcc, %1, %1\n\tv_cndmask_b32 %0, 0, %2, vcc" : "=v"(r), "+v"(word) : "v"(p) : "vcc");
    return r;
}

// general tile (diagonal / masked / ragged): per-score masks and |d|
template <bool DROP>
__device__ __forceinline__ void fwd_softmax_gen(f32x4 (&s)[4][2], f32x4 (&o)[4][2], float (&m_run)[2], float (&l_run)[2],
                                                const float (&i_f)[2], float c1, float slope2, float j0f, int g, const uint8_t* m_tile,
                                                bool causal, uint32_t keep32, uint16_t* bitp, long bstride) {
    uint32_t mbits[4];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) mbits[kb] = *reinterpret_cast<const uint32_t*>(m_tile + 16 * kb + 4 * g);
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        float tmax = NEG_FILL;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float jf = j0f + (float)(4 * g) + (float)(16 * kb + r);
                const bool ok = ((((mbits[kb] >> (8 * r)) & 0xff) != 0) && (!causal || jf <= i_f[qb]));
                const float t = ok ? fmaf(-slope2, fabsf(jf - i_f[qb]), s[kb][qb][r] * c1) : NEG_FILL;
                s[kb][qb][r] = t;
                tmax = fmaxf(tmax, t);
            }
        // The running maximum only has to be the SAME for the 4 lanes of a row and within 2^RESCALE_THR of the true one.  So the test uses
        // each lane's OWN maximum (no cross-lane traffic); the two swizzles of the row reduction -- LDS-crossbar round trips in the
        // middle of the dependency chain max -> exp -> P V -- run only in the rare iteration that actually raises a maximum.
        if (__any(tmax > m_run[qb] + RESCALE_THR)) {   // wave-uniform: rescale only when some row's max really grew
            tmax = group_max(tmax);
            const float m_new = fmaxf(m_run[qb], tmax);
            const float alpha = fast_exp2(m_run[qb] - m_new);
            m_run[qb] = m_new;
            l_run[qb] *= alpha;
#pragma unroll
            for (int db = 0; db < 4; ++db) o[db][qb] *= alpha;
        }
        const float mm = m_run[qb];
        float psum = 0.f;
        uint32_t word = qb == 1 ? keep32 : keep32 << 16;   // bit 16 qb + 4 kb + r: taken from the top, so (kb, r) run downwards
#pragma unroll
        for (int kb = 3; kb >= 0; --kb)
#pragma unroll
            for (int r = 3; r >= 0; --r) {
                const float p = fast_exp2(s[kb][qb][r] - mm);
                psum += p;   // the softmax normaliser is that of the un-dropped probabilities
                s[kb][qb][r] = DROP ? drop_next(p, word) : p;
            }
        l_run[qb] += psum;
        if (DROP) bitp[qb * bstride] = (uint16_t)((keep32 >> (16 * qb)) & 0xffffu);
    }
}

// linear tile (all keys on one side of the wave's rows: |d| = sigma (i - j)).  The scores arrive from the MFMA with the key part of the
// ALiBi term already in them (C input kq0 = sigma slope2 (4g + r) / c1, attn_fwd_kernel) and are taken RELATIVE TO THE RUNNING MAXIMUM
// by the row term rowck = sigma slope2 (j0 + 16 kb - i) - m_base: one fma per score in front of its exp (was: packed fma + subtract).
// m_base is the running maximum, or 0 while the row has none yet (a row term of 1.7e38 would swallow the score).
template <bool DROP>
__device__ __forceinline__ void fwd_softmax_lin(f32x4 (&s)[4][2], f32x4 (&o)[4][2], float (&m_run)[2], float (&l_run)[2],
                                                const float (&i_f)[2], float c1, float sg_slope2, float j0f, uint32_t keep32,
                                                uint16_t* bitp, long bstride) {
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const float m_base = m_run[qb] > -1e37f ? m_run[qb] : 0.f;
        const float rc = fmaf(sg_slope2, j0f - i_f[qb], -m_base);
        float tmax = NEG_FILL;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            const float rck = rc + sg_slope2 * (float)(16 * kb);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float t = fmaf(s[kb][qb][r], c1, rck);
                s[kb][qb][r] = t;
                tmax = fmaxf(tmax, t);
            }
        }
        if (__any(tmax > (m_run[qb] - m_base) + RESCALE_THR)) {   // see fwd_softmax_gen; everything here is relative to m_base
            tmax = group_max(tmax);
            const float m_new = fmaxf(m_run[qb], m_base + tmax);
            const float alpha = fast_exp2(m_run[qb] - m_new);
            const float shift = m_new - m_base;
            m_run[qb] = m_new;
            l_run[qb] *= alpha;
#pragma unroll
            for (int db = 0; db < 4; ++db) o[db][qb] *= alpha;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) s[kb][qb] -= shift;
        }
        float psum = 0.f;
        uint32_t word = qb == 1 ? keep32 : keep32 << 16;   // bit 16 qb + 4 kb + r: taken from the top, so (kb, r) run downwards
#pragma unroll
        for (int kb = 3; kb >= 0; --kb)
#pragma unroll
            for (int r = 3; r >= 0; --r) {
                const float p = fast_exp2(s[kb][qb][r]);
                psum += p;
                s[kb][qb][r] = DROP ? drop_next(p, word) : p;
            }
        l_run[qb] += psum;
        if (DROP) bitp[qb * bstride] = (uint16_t)((keep32 >> (16 * qb)) & 0xffffu);
    }
}

template <bool DROP>
__global__ __launch_bounds__(256, 3) void attn_fwd_kernel(AttnArgs a) {
    // K / V tiles arrive by LDS DMA (buffer_load ... lds: no VGPR round trip, no ds_write, no address arithmetic in the loop) into
    // two alternating stages, so ONE barrier per key tile both publishes tile t and proves that the stage tile t+1 is about to
    // overwrite is no longer read.  The DMA writes LDS linearly (wave base + lane * 16): the XOR swizzles of the "a" / "t" layouts sit
    // on the SOURCE address.  Rows past the end of the sequence lie outside the buffer resource and read as zero.
    __shared__ __attribute__((aligned(16))) char smem[2 * 16384 + 2 * 64 + 16];
    char* k_stage = smem;             // [2][8 KiB] "a" layout
    char* v_stage = smem + 16384;     // [2][8 KiB] "t" layout
    uint8_t* m_stage = reinterpret_cast<uint8_t*>(smem + 32768);    // [2][64] key mask bytes
    int* full_flags = reinterpret_cast<int*>(smem + 32768 + 128);   // [2]

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g = lane >> 4, c = lane & 15;
    int bi = blockIdx.z, hi = blockIdx.y, qt;
    if (!causal_order(a, true, bi, hi, qt) && (a.causal || !xcd_batch_coords(a, bi, hi, qt))) qt = qtile_of(blockIdx.x, gridDim.x, a.causal);
    const int q0 = qt * 128;
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
    // The key padding of a ragged batch is not walked: the live key tiles come from the band pre-pass (band_key_tiles), or, without a
    // band buffer (no ALiBi: cross-attention), from a scan of the mask row that starts here
    const KeyScan kscan = key_scan_begin(a.band ? nullptr : mp, a.nk, lane);
    // Padding rows (qmask == 0): a block of nothing else walks no tile, a wave of nothing else sits out every tile (it gets an empty band
    // below) -- in a right-padded ragged batch these rows would otherwise walk EVERY key tile (no ALiBi band without a live own key).
    // Every wave reads the block's 128 mask bytes itself (two coalesced byte loads, behind the q loads so that all of the prologue's
    // loads are in flight together): no barrier, nothing live across the tile loop (the epilogue reads its rows' bytes again; parking them
    // in LDS or in a register cost more in spills than the two loads), and no branch (an early return put a wait for these loads in
    // front of the band loads and the first tile request: +7 % on a full-length batch).
    bool wave_dead = false, block_dead = false;
    if (a.qmask) {
        const uint8_t* qm = a.qmask + (long)bi * a.nq;
        const int ia = q0 + lane, ib = q0 + 64 + lane;
        const uint8_t ma = ia < a.nq ? qm[ia] : 0, mb = ib < a.nq ? qm[ib] : 0;
        const unsigned long long la = __ballot(ma != 0), lb = __ballot(mb != 0);
        block_dead = (la | lb) == 0ull;   // walks no tile at all (nt = t_lo below); the epilogue stores its zeros and dead lse
        wave_dead = (uint32_t)((w < 2 ? la : lb) >> (32 * (w & 1))) == 0u;
    }

    const int wv = __builtin_amdgcn_readfirstlane(w);   // the wave index as a scalar: tile classes must be wave-uniform FOR THE COMPILER
    const int i_lo = q0 + 32 * wv + off, i_hi = i_lo + 31;
    uint32_t rowc[2] = {0, 0};
    if (DROP) {
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
            rowc[qb] = drop_row_const(a.seed, bi * a.h + hi, a.nq, q0 + 32 * w + 16 * qb + c);
    }

    const long bstride = (long)a.nkt64 * 64;   // keep-bit words between consecutive 16-query tiles
    uint16_t* bitbase = DROP ? a.dropbits + ((long)(bi * a.h + hi) * a.nqt16 + (q0 + 32 * wv) / 16) * bstride + lane : nullptr;

    f32x4 o[4][2];
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) o[db][qb] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run[2] = {NEG_FILL, NEG_FILL}, l_run[2] = {0.f, 0.f};

    int nt = (a.nk + 63) / 64, t_lo = 0;
    if (a.causal) {
        const int last = q0 + 127 + off;  // largest key index any row of this block may see
        nt = last < 0 ? 0 : min(nt, last / 64 + 1);
    }
    int wave_lo = -0x40000000, wave_hi = 0x40000000;   // keys this WAVE's 32 rows can reach (the bound holds for every row of the block)
    {   // ALiBi band: key tiles further than the reach of this block's 128 rows contribute < 2^-band_log2 per probability
        const float reach = band_reach(a, bi, hi, kh, q0 / 64, 2, c1, slope2);
        if (reach < 1.0e9f) {
            const int d = (int)reach + 1;
            t_lo = max(0, q0 + off - d) / 64;
            nt = min(nt, (q0 + 127 + off + d) / 64 + 1);
            wave_lo = i_lo - d; wave_hi = i_hi + d;     // the block walks the union of its four waves' ranges; a wave sits out the rest
        }
    }
    if (wave_dead) { wave_lo = 0x40000000; wave_hi = -0x40000000; }   // padding rows only: every tile is "outside this wave's band"
    if (a.band) band_key_tiles(a, bi, t_lo, nt); else key_scan_finish(kscan, mp, a.nk, lane, t_lo, nt);
    if (block_dead) nt = t_lo;

    const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc((void*)kp, 0, (int)(((long)(a.nk - 1) * a.k_ns + 64) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc((void*)vp, 0, (int)(((long)(a.nk - 1) * a.v_ns + 64) * 2), 0x00020000);
    // C input of the score MFMAs on linear tiles: the key part of the ALiBi term, sigma slope2 (4g + r) / c1 (re-signed when the walk
    // crosses the diagonal, once per block)
    float sigma = 1.f;
    f32x4 kq0;
#pragma unroll
    for (int r = 0; r < 4; ++r) kq0[r] = (slope2 / c1) * (float)(4 * g + r);
    // this wave's two 1 KiB pieces of a tile: LDS chunk L = piece * 64 + lane <- source chunk (inverse swizzle).  The second piece lies 8
    // rows further down with the same swizzle (both keys repeat every 8 rows): same lane offset, 8 rows on the SCALAR offset
    uint32_t voK, voV;
    {
        const int L = wv * 128 + lane, row = L >> 3, ch = L & 7;
        voK = (uint32_t)(((long)row * a.k_ns + ((ch ^ (row & 7)) << 3)) * 2);
        voV = (uint32_t)(((long)row * a.v_ns + ((ch ^ (((row >> 1) & 3) << 1)) << 3)) * 2);
    }
    auto issue_tile = [&](int t) {
        char* kd = k_stage + (t & 1) * 8192 + wv * 2048;
        char* vd = v_stage + (t & 1) * 8192 + wv * 2048;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const uint32_t sk = (uint32_t)(t * 64 + 8 * i) * (uint32_t)a.k_ns * 2u, sv = (uint32_t)(t * 64 + 8 * i) * (uint32_t)a.v_ns * 2u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsK, (__attribute__((address_space(3))) void*)(kd + i * 1024), 16, voK, sk, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsV, (__attribute__((address_space(3))) void*)(vd + i * 1024), 16, voV, sv, 0, 0);
        }
    };
    uint8_t mreg = 1;
    if (nt > t_lo) {
        issue_tile(t_lo);
        if (tid < 64) { const int j = t_lo * 64 + tid; mreg = (j < a.nk) ? (mp ? mp[j] : 1) : 0; }
    }
    for (int t = t_lo; t < nt; ++t) {
        const int j0 = t * 64;
        const char* k_tile = k_stage + (t & 1) * 8192;
        const char* v_tile = v_stage + (t & 1) * 8192;
        const uint8_t* m_tile = m_stage + (t & 1) * 64;
        const int* full_flag = full_flags + (t & 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of tile t (requested a whole iteration ago) have landed
        if (tid < 64) {
            m_stage[(t & 1) * 64 + tid] = mreg;
            const unsigned long long all = __ballot(mreg != 0);
            if (tid == 0) full_flags[t & 1] = (all == ~0ull) ? 1 : 0;
        }
        __syncthreads();   // tile t visible to all; everybody is done with the other stage (read in iteration t - 1)
        if (t + 1 < nt) {
            issue_tile(t + 1);
            if (tid < 64) { const int j = j0 + 64 + tid; mreg = (j < a.nk) ? (mp ? mp[j] : 1) : 0; }
        }
        // wave-uniform by construction, but read through LDS: say so, or every branch on it becomes an exec-masked region
        const int cls = classify(j0, i_lo, i_hi, __builtin_amdgcn_readfirstlane(*full_flag) != 0, a.causal != 0);
        if (cls == T_SKIP || j0 > wave_hi || j0 + 63 < wave_lo) continue;   // causal future, or outside this wave's own band (padding rows: all)
        const float j0f = (float)j0;
        uint32_t thr_t = a.thr8;
        if (DROP && a.thr_frac) {   // this block's threshold: thr8 + Bernoulli(frac16 / 65536), all-scalar (set_dropout)
            const uint32_t blk = (uint32_t)__builtin_amdgcn_readfirstlane((int)(((bi * a.h + hi) * a.nqt16 + (q0 + 32 * wv) / 16) * a.nkt64 + t));
            thr_t += ((block_mix(blk ^ a.seed) & 0xffffu) < a.thr_frac) ? 1u : 0u;
        }
        // one counter per (row c of the block, key tile, lane group g): rowc[0] is the row's constant, the tile index comes in here
        const uint32_t keep32 = DROP ? drop_keep32(rowc[0] + __umul24((uint32_t)((j0 >> 4) + g), 0xEBCA77u), thr_t) : 0xffffffffu;

        // S^T = K Q^T
        f32x4 s[4][2];
        if (cls == T_GEN) {
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                bf16x8 kf0 = frag_rows(k_tile, 16 * kb, 0, lane), kf1 = frag_rows(k_tile, 16 * kb, 1, lane);
#pragma unroll
                for (int qb = 0; qb < 2; ++qb) {
                    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf0, qf[qb][0], acc, 0, 0, 0);
                    s[kb][qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf1, qf[qb][1], acc, 0, 0, 0);
                }
            }
            fwd_softmax_gen<DROP>(s, o, m_run, l_run, i_f, c1, slope2, j0f, g, m_tile, a.causal, keep32, bitbase + t * 64, bstride);
        } else {
            const float sg = cls == T_LEFT ? 1.f : -1.f;
            if (sg != sigma) { sigma = sg; kq0 = -kq0; }
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                bf16x8 kf0 = frag_rows(k_tile, 16 * kb, 0, lane), kf1 = frag_rows(k_tile, 16 * kb, 1, lane);
#pragma unroll
                for (int qb = 0; qb < 2; ++qb) {
                    f32x4 acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf0, qf[qb][0], kq0, 0, 0, 0);
                    s[kb][qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf1, qf[qb][1], acc, 0, 0, 0);
                }
            }
            fwd_softmax_lin<DROP>(s, o, m_run, l_run, i_f, c1, sg * slope2, j0f, keep32, bitbase + t * 64, bstride);
        }

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

    AttnKernargPtr ae = attn_kernarg();   // the mask pointer is not held in registers across the tile loop (attention_common.h)
    asm volatile("" : "+s"(ae));
    const uint8_t* qm_e = ae->qmask;
    bool row_live[2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const int i = q0 + 32 * w + 16 * qb + c;
        row_live[qb] = !qm_e || (i < a.nq && qm_e[(long)bi * a.nq + i] != 0);
    }
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const int i = q0 + 32 * w + 16 * qb + c;
        const float l = group_sum(l_run[qb]);
        const float inv = (l > 0.f && row_live[qb] ? 1.f / l : 0.f) * (DROP ? a.inv_keep : 1.f);
        if (i < a.nq) {
            bf16_t* op = a.o + bi * a.o_bs + (long)i * a.o_ns + hi * a.o_hs;
#pragma unroll
            for (int db = 0; db < 4; ++db) {
                uint2 pk;
                pk.x = pack_bf2(o[db][qb][0] * inv, o[db][qb][1] * inv);
                pk.y = pack_bf2(o[db][qb][2] * inv, o[db][qb][3] * inv);
                *reinterpret_cast<uint2*>(op + 16 * db + 4 * g) = pk;
            }
            if (g == 0) a.lse[((long)bi * a.h + hi) * a.nq + i] = row_live[qb] ? (m_run[qb] + log2f(l)) * LN2 : NEG_FILL;
        }
    }
}

// ==========================================================================================================
// dQ (+ d slope): same decomposition as the forward
// ==========================================================================================================
// The kernel is bound by VALU ISSUE, not by the matrix pipe (tools/issue_probe.hip: a SIMD issues one VALU instruction per ~3.5
// cycles at two waves, v_exp_f32 costs two such slots, and the MFMAs of a key tile hide completely under ~450 VALU instructions).
// So everything that can ride on an MFMA does:
//   * linear tiles (all keys on one side of the wave's 32 rows: |d| = sigma (i - j), sigma = +-1): the key part of the ALiBi term,
//     sigma slope2 (4g + r) / c1, is the C INPUT of the score MFMA (a loop-invariant register quad, re-signed when the walk crosses
//     the diagonal), the row part sigma slope2 (j0 + 16 kb - i) - lse_i one scalar per (row block, key block) and tile: a score costs
//     ONE fma before its exp (was: add + fma + sub);
//   * -delta_i (times the keep probability under dropout) is the C input of the dP MFMA, so dS = P' * select(keep, dP - delta', -delta')
//     is one v_bfi_b32 + one multiply (was: and + fma + multiply); 1 / keep_prob rides on the row term of the exponent;
//   * d slope_h = sum_ij dS_ij (-|d_ij|) needs the row sums sum_j dS_ij, sum_j dS_ij |d_ij| and sum_j P_ij |d_ij|: with |d| linear in
//     the key index they are three more ROWS of the dQ^T = K^T dS^T product -- an A fragment holding (sigma, sigma * jrel, 1) against the
//     bf16 dS^T (and P^T) B fragments that exist anyway: two MFMAs per 512 scores instead of three accumulations per score; the
//     per-tile results are folded into per-row accumulators with the tile's distance (i - j0) once per tile.
// Tiles with a masked key, the causal diagonal or the sequence end take the general per-score path (T_GEN) as before.
//
// d slope_h = sum_ij dS_ij * (-|j - i - off|).  delta is computed from the bf16-rounded O, so each row's dS carries a
// common error -P_ij * eps_i; since sum_j dS_ij must be 0, the measured row sum r_i = -eps_i gives the exact
// correction  + r_i * sum_j P_ij |d_ij|  (otherwise the error is amplified by the mean attended distance).
__device__ __forceinline__ float sel_bits(int m, float a, float b) {   // m all-ones: a, m zero: b  -> one v_bfi_b32
    // The mask is laundered through an EMPTY asm statement: knowing that it is a sign-extended bit, hipcc rewrites (b & ~m) as shift +
    // compare + cndmask and the whole select as four VALU instructions.  (The instruction itself must NOT be written as asm: `a` comes
    // straight out of an MFMA, and hipcc inserts the MFMA-write -> VALU-read wait states only for instructions it schedules itself.)
    asm("" : "+v"(m));
    return __int_as_float((__float_as_int(a) & m) | (__float_as_int(b) & ~m));
}

// general tile: per-score masks and |d| (diagonal / masked / ragged tiles)
template <bool SLOPE_GRAD, bool DROP>
__device__ __forceinline__ void dq_scores_gen(int u, f32x4 (&s)[2][2], const f32x4 (&dp)[2][2], const float (&l2)[2], const float (&dl)[2],
                                              const float (&i_f)[2], float c1, float slope2, float j0f, int g, const uint8_t* m_tile,
                                              bool causal, float (&acc_d)[2], float (&acc_r)[2], float (&acc_p)[2],
                                              const uint32_t (&kw)[2], float inv_keep) {
    uint32_t mbits[2];
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2) mbits[k2] = *reinterpret_cast<const uint32_t*>(m_tile + 16 * (2 * u + k2) + 4 * g);
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float jf = j0f + (float)(4 * g) + (float)(16 * (2 * u + k2) + r);
                const bool ok = (((mbits[k2] >> (8 * r)) & 0xff) != 0) && (!causal || jf <= i_f[qb]);
                const float ad = fabsf(jf - i_f[qb]);
                const float t = ok ? fmaf(-slope2, ad, s[k2][qb][r] * c1) : NEG_FILL;
                const float p = fast_exp2(t - l2[qb]);
                float ds;
                if (DROP) {   // keep bit 4*kb + r of the forward's word -> all-ones / zero mask
                    const int keepm = __builtin_amdgcn_sbfe((int)kw[qb], 4 * (2 * u + k2) + r, 1);
                    const float dpm = __uint_as_float(__float_as_uint(dp[k2][qb][r]) & (uint32_t)keepm);
                    ds = p * fmaf(dpm, inv_keep, -dl[qb]);
                } else {
                    ds = p * (dp[k2][qb][r] - dl[qb]);
                }
                if (SLOPE_GRAD) { acc_d[qb] = fmaf(ds, ad, acc_d[qb]); acc_r[qb] += ds; acc_p[qb] = fmaf(p, ad, acc_p[qb]); }
                s[k2][qb][r] = ds;
            }
        }
    }
}

template <bool DROP, bool SLOPE_GRAD>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(AttnArgs a) {
    // three images per key tile -- K "a" (A operand of S^T), K "t" (K^T, A operand of dQ^T), V "a" (A operand of dP^T = V dO^T) -- by LDS
    // DMA into two alternating stages, one barrier per tile (see attn_fwd_kernel); K is fetched twice (it comes from the L2)
    __shared__ __attribute__((aligned(16))) char smem[2 * 24576 + 2 * 64 + 16];
    uint8_t* m_stage = reinterpret_cast<uint8_t*>(smem + 49152);
    int* full_flags = reinterpret_cast<int*>(smem + 49152 + 128);

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g = lane >> 4, c = lane & 15;
    int bi = blockIdx.z, hi = blockIdx.y, qt;
    if (!causal_order(a, true, bi, hi, qt) && (a.causal || !xcd_batch_coords(a, bi, hi, qt))) qt = qtile_of(blockIdx.x, gridDim.x, a.causal);
    const int q0 = qt * 128;
    const int kh = (a.kvh == 1) ? 0 : hi;
    const int off = a.nk - a.nq;
    const bf16_t* qp = a.q + bi * a.q_bs + hi * a.q_hs;
    const bf16_t* dop = a.d_o + bi * a.o_bs + hi * a.o_hs;
    const bf16_t* kp = a.k + bi * a.k_bs + kh * a.k_hs;
    const bf16_t* vp = a.v + bi * a.v_bs + kh * a.v_hs;
    const uint8_t* mp = a.kmask ? a.kmask + (long)bi * a.nk : nullptr;
    const float slope2 = a.slopes ? a.slopes[hi] * LOG2E : 0.f;
    const float c1 = a.scale * LOG2E;
    const float keep_prob = DROP ? 1.f / a.inv_keep : 1.f;

    // the block's 128 lse values, requested by every wave itself (two coalesced loads, in flight with the row loads below): see the
    // dead-block test behind them
    float lse_a, lse_b;
    {
        const float* lb = a.lse + ((long)bi * a.h + hi) * a.nq;
        const int ia = q0 + lane, ib = q0 + 64 + lane;
        lse_a = ia < a.nq ? lb[ia] : NEG_FILL; lse_b = ib < a.nq ? lb[ib] : NEG_FILL;
    }
    const KeyScan kscan = key_scan_begin(a.band ? nullptr : mp, a.nk, lane);   // see attn_fwd_kernel
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
        // Dead rows -- outside the problem, padding rows of the forward's qmask (lse = NEG_FILL), and rows whose keys were ALL masked
        // (lse ~ -1e38: degenerate uniform attention whose output the caller zeroes) -- get p = 0
        const float lse_i = i < a.nq ? a.lse[si] : NEG_FILL;
        l2[qb] = lse_i > -1e37f ? lse_i * LOG2E : 1e30f;
        // delta_i = sum_d O[i, d] * dO[i, d] (bf16 inputs, fp32 sum): this lane holds 16 of the row's 64 dO values, its three
        // lane-group peers the rest.  Computed here, once per row, and stored for the dK/dV kernel that runs after this one
        // (a separate 47 us pass over O and dO per layer before).
        float dsum = 0.f;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const bf16x8 of = load_row_frag(a.o + bi * a.o_bs + hi * a.o_hs, a.o_ns, i, a.nq, ks, lane);
            const uint4 x = __builtin_bit_cast(uint4, of), y = __builtin_bit_cast(uint4, dof[qb][ks]);
            const uint32_t xs[4] = {x.x, x.y, x.z, x.w}, ys[4] = {y.x, y.y, y.z, y.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                dsum += bf2f(xs[e] & 0xffff) * bf2f(ys[e] & 0xffff);
                dsum += bf2f(xs[e] >> 16) * bf2f(ys[e] >> 16);
            }
        }
        dl[qb] = group_sum(dsum);
        if (g == 0 && i < a.nq) const_cast<float*>(a.delta)[si] = dl[qb];
    }
    // A block of dead rows only walks no tile at all (nt = t_lo below; its accumulators store zeros), a wave of dead rows only sits out
    // every tile (it gets an empty band).  No early return: a branch here would put a wait for the loads above in front of the band
    // loads and the first tile request (measured: +7 % on the full-length batch).
    const bool block_dead = !__any(lse_a > -1e37f || lse_b > -1e37f);
    const bool wave_dead = !__any(l2[0] < 1e29f || l2[1] < 1e29f);
    const int wv = __builtin_amdgcn_readfirstlane(w);       // the wave index as a scalar: tile classes must be wave-uniform FOR THE COMPILER
    const int i_lo = q0 + 32 * wv + off, i_hi = i_lo + 31;
    const long bstride = (long)a.nkt64 * 64;
    const uint16_t* bitbase = DROP ? a.dropbits + ((long)(bi * a.h + hi) * a.nqt16 + (q0 + 32 * w) / 16) * bstride + lane : nullptr;
    uint32_t kw[2] = {0, 0}, kwn[2] = {0, 0};   // keep-bit words of the current / next key tile
    f32x4 dq[4][2];
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) dq[db][qb] = f32x4{0.f, 0.f, 0.f, 0.f};
    float acc_d[2] = {0.f, 0.f}, acc_r[2] = {0.f, 0.f}, acc_p[2] = {0.f, 0.f};

    // ---- loop-invariant operands of the linear tiles (see the header of this section)
    const float sc = slope2 / c1;
    float sigma = 1.f;                                       // the walk starts left of the diagonal (or stays there: causal)
    f32x4 kq0;                                               // C input of the score MFMAs: sigma * sc * (4g + r)
#pragma unroll
    for (int r = 0; r < 4; ++r) kq0[r] = sc * (float)(4 * g + r);
    f32x4 ndl[2];                                            // C input of the dP MFMAs: -delta' (delta' = delta * keep_prob)
    float l2x[2];                                            // lse in log2 units minus log2(1 / keep_prob): exp2 gives P / keep_prob
    const float log2_inv = DROP ? __builtin_log2f(a.inv_keep) : 0.f;
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const float nd = -dl[qb] * keep_prob;
        ndl[qb] = f32x4{nd, nd, nd, nd};
        l2x[qb] = l2[qb] - log2_inv;
    }
    // A fragment of the three extra product rows: row 0 = sigma, row 1 = sigma * jrel, row 2 = 1; element e of lane (p, g) multiplies
    // key 32u + 16(e >> 2) + 4g + (e & 3) of the tile -- the enumeration of the dS^T / P^T B fragments (pack8 of two 16-key blocks)
    bf16x8 wfr[2];
    uint32_t wflip = 0u;                                     // sign bits of the rows that carry sigma
    if (SLOPE_GRAD) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float jrel = (float)(32 * u + 16 * (e >> 2) + 4 * g + (e & 3));
                wfr[u][e] = (__bf16)(c == 0 ? 1.f : (c == 1 ? jrel : (c == 2 ? 1.f : 0.f)));
            }
        wflip = c < 2 ? 0x80008000u : 0u;
    }

    int nt = (a.nk + 63) / 64, t_lo = 0;
    if (a.causal) {
        const int last = q0 + 127 + off;
        nt = last < 0 ? 0 : min(nt, last / 64 + 1);
    }
    int wave_lo = -0x40000000, wave_hi = 0x40000000;
    {   // same ALiBi band as the forward (same bound inputs, same arithmetic, same per-wave range)
        const float reach = band_reach(a, bi, hi, kh, q0 / 64, 2, c1, slope2);
        if (reach < 1.0e9f) {
            const int d = (int)reach + 1;
            t_lo = max(0, q0 + off - d) / 64;
            nt = min(nt, (q0 + 127 + off + d) / 64 + 1);
            wave_lo = i_lo - d; wave_hi = i_hi + d;
        }
    }
    if (wave_dead) { wave_lo = 0x40000000; wave_hi = -0x40000000; }   // dead rows only: every tile is "outside this wave's band"
    // masked keys only: P = exp2(NEG_FILL - lse) = 0 for every live row, as in the forward
    if (a.band) band_key_tiles(a, bi, t_lo, nt); else key_scan_finish(kscan, mp, a.nk, lane, t_lo, nt);
    if (block_dead) nt = t_lo;
    const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc((void*)kp, 0, (int)(((long)(a.nk - 1) * a.k_ns + 64) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc((void*)vp, 0, (int)(((long)(a.nk - 1) * a.v_ns + 64) * 2), 0x00020000);
    uint32_t voKa[2], voKt[2], voV[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int L = (wv * 2 + i) * 64 + lane, row = L >> 3, ch = L & 7;
        voKa[i] = (uint32_t)(((long)row * a.k_ns + ((ch ^ (row & 7)) << 3)) * 2);
        voKt[i] = (uint32_t)(((long)row * a.k_ns + ((ch ^ (((row >> 1) & 3) << 1)) << 3)) * 2);
        voV[i] = (uint32_t)(((long)row * a.v_ns + ((ch ^ (row & 7)) << 3)) * 2);
    }
    auto issue_tile = [&](int t) {
        char* base = smem + (t & 1) * 24576 + wv * 2048;
        const uint32_t sk = (uint32_t)(t * 64) * (uint32_t)a.k_ns * 2u, sv = (uint32_t)(t * 64) * (uint32_t)a.v_ns * 2u;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsK, (__attribute__((address_space(3))) void*)(base + i * 1024), 16, voKa[i], sk, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsK, (__attribute__((address_space(3))) void*)(base + 8192 + i * 1024), 16, voKt[i], sk, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsV, (__attribute__((address_space(3))) void*)(base + 16384 + i * 1024), 16, voV[i], sv, 0, 0);
        }
    };
    uint8_t mreg = 1;
    if (nt > t_lo) {
        issue_tile(t_lo);
        if (tid < 64) { const int j = t_lo * 64 + tid; mreg = (j < a.nk) ? (mp ? mp[j] : 1) : 0; }
        if (DROP) { kwn[0] = bitbase[t_lo * 64]; kwn[1] = bitbase[bstride + t_lo * 64]; }
    }
    for (int t = t_lo; t < nt; ++t) {
        const int j0 = t * 64;
        const char* k_tile = smem + (t & 1) * 24576;
        const char* kt_tile = k_tile + 8192;
        const char* v_tile = k_tile + 16384;
        const uint8_t* m_tile = m_stage + (t & 1) * 64;
        const int* full_flag = full_flags + (t & 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // tile t's pieces (and its keep-bit words) have landed
        kw[0] = kwn[0]; kw[1] = kwn[1];
        if (tid < 64) {
            m_stage[(t & 1) * 64 + tid] = mreg;
            const unsigned long long all = __ballot(mreg != 0);
            if (tid == 0) full_flags[t & 1] = (all == ~0ull) ? 1 : 0;
        }
        __syncthreads();
        if (t + 1 < nt) {
            issue_tile(t + 1);
            if (tid < 64) { const int j = j0 + 64 + tid; mreg = (j < a.nk) ? (mp ? mp[j] : 1) : 0; }
            if (DROP) { kwn[0] = bitbase[(t + 1) * 64]; kwn[1] = bitbase[bstride + (t + 1) * 64]; }
        }
        // rows beyond nq hold zero fragments and are never written; the last q block may straddle nq: use the general path
        const bool rows_ok = q0 + 32 * wv + 31 < a.nq;
        // wave-uniform by construction, but read through LDS: say so, or every branch on it becomes an exec-masked region
        int cls = classify(j0, i_lo, i_hi, __builtin_amdgcn_readfirstlane(*full_flag) != 0, a.causal != 0);
        if (cls == T_SKIP || j0 > wave_hi || j0 + 63 < wave_lo) continue;   // causal future, or outside this wave's own band (dead rows: all)
        if (!rows_ok && SLOPE_GRAD) cls = T_GEN;
        const float j0f = (float)j0;

        if (cls == T_GEN) {
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
                dq_scores_gen<SLOPE_GRAD, DROP>(u, s, dp, l2, dl, i_f, c1, slope2, j0f, g, m_tile, a.causal, acc_d, acc_r, acc_p, kw, a.inv_keep);
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
            continue;
        }

        // ---- linear tile
        const float sg = cls == T_LEFT ? 1.f : -1.f;
        if (sg != sigma) {   // the walk crossed the diagonal (once per block): re-sign the key part and the weight rows
            sigma = sg;
            kq0 = -kq0;
            if (SLOPE_GRAD) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    uint4 x = __builtin_bit_cast(uint4, wfr[u]);
                    x.x ^= wflip; x.y ^= wflip; x.z ^= wflip; x.w ^= wflip;
                    wfr[u] = __builtin_bit_cast(bf16x8, x);
                }
            }
        }
        float rowck[2][4];   // sigma slope2 (j0 + 16 kb - i) - lse2_i  (+ log2 keep_prob^-1)
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            const float rc = fmaf(sg * slope2, j0f - i_f[qb], -l2x[qb]);
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) rowck[qb][kb] = rc + sg * slope2 * (float)(16 * kb);
        }
        f32x4 ws[2], wp[2];   // rows 0..2 of (weights x dS^T) and (weights x P^T) of this tile, per row block
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
                    f32x4 acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf0, qf[qb][0], kq0, 0, 0, 0);
                    s[k2][qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf1, qf[qb][1], acc, 0, 0, 0);
                    f32x4 acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf0, dof[qb][0], ndl[qb], 0, 0, 0);
                    dp[k2][qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf1, dof[qb][1], acc2, 0, 0, 0);
                }
            }
            f32x4 pr[2][2];   // P / keep_prob
#pragma unroll
            for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float p = fast_exp2(fmaf(s[k2][qb][r], c1, rowck[qb][2 * u + k2]));
                        float y = dp[k2][qb][r];                                  // dP - delta'
                        if (DROP) y = sel_bits(__builtin_amdgcn_sbfe((int)kw[qb], 4 * (2 * u + k2) + r, 1), y, ndl[qb][0]);
                        s[k2][qb][r] = p * y;
                        if (SLOPE_GRAD) pr[k2][qb][r] = p;
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
            if (SLOPE_GRAD) {
#pragma unroll
                for (int qb = 0; qb < 2; ++qb) {
                    const bf16x8 pf = pack8(pr[0][qb], pr[1][qb]);
                    const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
                    ws[qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfr[u], dsf[qb], u == 0 ? z : ws[qb], 0, 0, 0);
                    wp[qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfr[u], pf, u == 0 ? z : wp[qb], 0, 0, 0);
                }
            }
        }
        if (SLOPE_GRAD) {
            // lanes of group 0 hold rows 0..2 of this lane's query column: sigma sum dS, sigma sum dS jrel, sum dS (and the same of P');
            // |d| = sigma (i - j0 - jrel).  The other lane groups hold all-zero rows and add nothing.
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                const float dist = i_f[qb] - j0f;
                acc_d[qb] += fmaf(dist, ws[qb][0], -ws[qb][1]);
                acc_r[qb] += ws[qb][2];
                acc_p[qb] = fmaf(keep_prob, fmaf(dist, wp[qb][0], -wp[qb][1]), acc_p[qb]);
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
    if (SLOPE_GRAD) {
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
// ALiBi band skipping.  ALiBi subtracts slope*|i - j| from the score, so for the steep heads everything far from the diagonal
// is numerically dead: with B >= |q.k|*scale*log2e the probability of key j relative to the row's largest one is at most
// 2^(2B - slope2*|d|) (attention_common.h, band_reach).  Tiles where that is < 2^-band_log2 (default 30: 2048 such terms sum to
// < 2^-19 of the row normaliser, three orders of magnitude below the 2^-9 rounding of the bf16 probabilities that enter P V; 40 puts the
// sum below fp32 resolution) are not visited -- in the forward AND in both backward kernels, from
// the same bound, so the three stay consistent.  The bound comes from this pre-pass: per 64-row query tile max ||q||^2 and the smallest
// score of a row's OWN key (a lower bound of that row's maximum), and max ||k||^2 per (batch, kv head) (Cauchy-Schwarz for every
// other score; band_reach, attention_common.h).  A query tile containing a row whose own key is masked or out of range has no
// such lower bound on its row maximum and is marked +inf = never skipped.  Learned slopes <= 0 disable skipping for that head.
// ==========================================================================================================
__global__ __launch_bounds__(64) void attn_band_kernel(AttnArgs a, float* __restrict__ band) {
    // One wave per 64-row tile; EIGHT LANES per row (one 16-byte chunk each: a wave load covers 8 whole 128-byte rows), eight passes.
    // (One lane per row made every wave load touch 64 cache lines for 16 useful bytes each.)
    const int lane = threadIdx.x, bi = blockIdx.z, hi = blockIdx.y, x = blockIdx.x;
    const int off = a.nk - a.nq;
    const bool is_q = x < a.nqt64;
    if (!is_q && hi >= a.kvh) return;
    if (is_q && band_head_unbounded(a, a.slopes[hi] * LOG2E)) return;   // this head never skips: band_reach will not read its entries
    const int n = is_q ? a.nq : a.nk;
    const long nq_part = (long)a.b * a.h * a.nqt64;
    const int ch = lane & 7;
    float vmax = 0.f, dmin = __builtin_inff();   // max |row|^2; query rows: min q_i . k_i' with the row's own key i' = i + off
    // all eight passes' rows are requested before the first one is reduced (rows past the end re-read the last row and are not counted):
    // pass by pass, every load was waited for before the next was issued -- eight serial trips to memory per 64-row tile
    uint4 u[8], uk[8];
    bool own[8], qdead[8];
    // the tile's 64 mask bytes -- of the rows' own keys, and of the rows themselves -- by one coalesced load each, handed to the eight
    // lanes of a row by shuffle (a byte load per lane and pass put 16 dependent loads in front of the row loads)
    int km_l = 0, qm_l = 1;
    if (is_q) {
        const int r64 = x * 64 + lane, jd = r64 + off;
        if (r64 < a.nq && jd >= 0 && jd < a.nk) km_l = a.kmask ? a.kmask[(long)bi * a.nk + jd] : 1;
        if (a.qmask && r64 < a.nq) qm_l = a.qmask[(long)bi * a.nq + r64];
    }
#pragma unroll
    for (int pass = 0; pass < 8; ++pass) {
        const int row = min((is_q ? x : x - a.nqt64) * 64 + pass * 8 + (lane >> 3), n - 1);
        const bf16_t* p = is_q ? a.q + bi * a.q_bs + (long)row * a.q_ns + hi * a.q_hs : a.k + bi * a.k_bs + (long)row * a.k_ns + hi * a.k_hs;
        const int jd = row + off;   // the row's own key: its score bounds the row maximum from below
        qdead[pass] = __shfl(qm_l, pass * 8 + (lane >> 3), 64) == 0;   // padding row: no output, no bound to hold
        own[pass] = __shfl(km_l, pass * 8 + (lane >> 3), 64) != 0;     // rows past nq: 0 (they re-read the last row and are not counted)
        const bf16_t* pk = own[pass] ? a.k + bi * a.k_bs + (long)jd * a.k_ns + (a.kvh == 1 ? 0 : hi) * a.k_hs : p;
        u[pass] = *reinterpret_cast<const uint4*>(p + ch * 8);
        uk[pass] = *reinterpret_cast<const uint4*>(pk + ch * 8);
    }
#pragma unroll
    for (int pass = 0; pass < 8; ++pass) {
        const int row = (is_q ? x : x - a.nqt64) * 64 + pass * 8 + (lane >> 3);
        float v = 0.f, d = 0.f;
        const bool own_ok = own[pass];
        {
            const uint32_t w[4] = {u[pass].x, u[pass].y, u[pass].z, u[pass].w}, wk[4] = {uk[pass].x, uk[pass].y, uk[pass].z, uk[pass].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float lo = bf2f(w[e] & 0xffff), hi_ = bf2f(w[e] >> 16);
                v = fmaf(lo, lo, fmaf(hi_, hi_, v));
                d = fmaf(lo, bf2f(wk[e] & 0xffff), fmaf(hi_, bf2f(wk[e] >> 16), d));
            }
        }
        // the row's eight partial sums (lanes 8r .. 8r + 7)
        v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64);
        d += __shfl_xor(d, 1, 64); d += __shfl_xor(d, 2, 64); d += __shfl_xor(d, 4, 64);
        if (row < n && !qdead[pass]) {
            if (is_q && !own_ok) v = __builtin_inff();   // no lower bound on this row's maximum: the tile is never skipped
            vmax = fmaxf(vmax, v);
            if (own_ok) dmin = fminf(dmin, d);
        }
    }
    const float v = wave_max(vmax);
    const float dot = is_q ? -wave_max(-dmin) : 0.f;   // minimum over the tile's live rows (none: +inf next to v = 0 -> reach 0, band_reach)
    if (lane == 0) {
        if (is_q) {
            band[((long)(bi * a.h + hi)) * a.nqt64 + x] = v;
            band[nq_part + ((long)(bi * a.h + hi)) * a.nqt64 + x] = dot;
        } else {
            atomicMax(reinterpret_cast<unsigned int*>(band + 2 * nq_part + bi * a.kvh + hi), __float_as_uint(v));   // v >= 0
        }
    }
    if (!is_q && hi == 0 && a.kmask) {   // the live key tiles of this batch row (band_key_tiles): this tile has a live key -> widen the range
        const int kt = x - a.nqt64, j = kt * 64 + lane;
        const bool any = __any(j < a.nk && a.kmask[(long)bi * a.nk + j] != 0);
        if (any && lane == 0) {
            unsigned int* kr = reinterpret_cast<unsigned int*>(band + 2 * nq_part + (long)a.b * a.kvh) + 2 * bi;
            atomicMax(kr, (unsigned)(kt + 1));
            atomicMax(kr + 1, (unsigned)((a.nk + 63) / 64 - kt));
        }
    }
}

// fills a.band for this problem.  The band buffer is the CALLER's (spn_attn_band_elems floats): the forward fills it (`reuse` = false),
// the backward of the same q / k / mask reads it back (`reuse` = true).  Without a buffer (or without ALiBi slopes, or with the knob
// at 0) every tile is visited; with it, skipped tiles contribute < 2^-19 of a row's normaliser at the default threshold (include/spn.h).  Nothing is
// allocated or synchronised here.
int prepare_band(AttnArgs& a, hipStream_t stream, float* own, bool reuse) {
    const float band_log2 = (float)spn_tune(SPN_TUNE_ATTN_BAND);
    a.band = nullptr; a.band_log2 = band_log2; a.nqt64 = (a.nq + 63) / 64;
    a.order = spn_tune_i(SPN_TUNE_ATTN_ORDER);
    if (!a.slopes || band_log2 <= 0.f || !own) return SPN_OK;
    a.band = own;
    if (reuse) return SPN_OK;
    const size_t nq_part = (size_t)a.b * a.h * a.nqt64;
    (void)hipMemsetAsync(own + 2 * nq_part, 0, ((size_t)a.b * a.kvh + 2 * (size_t)a.b) * 4, stream);
    hipLaunchKernelGGL(attn_band_kernel, dim3(a.nqt64 + (a.nk + 63) / 64, a.h, a.b), dim3(64), 0, stream, a, own);
    return SPN_OK;
}

// Drop probability p as an 8-bit threshold PLUS a 16-bit fraction: a wave's 32x64 block of scores uses thr8 or thr8 + 1, chosen by a
// (scalar) hash of the block's coordinates with probability frac16 / 65536, so every score is dropped with probability
// (thr8 + frac16 / 65536) / 256 = p to within 2^-24 -- F.dropout(p)'s rate (attend.py:122), where a plain 8-bit threshold would
// turn p = 0.1 into 26/256 = 0.1016 -- at no VALU cost: the bit-sliced Bernoulli draw below still folds eight random words.
// Kept probabilities are scaled by 1 / (1 - p).  (Scores of one block share the choice: pairwise correlation < 2e-4.)
void set_dropout(AttnArgs& a, float p_drop, unsigned seed, void* dropbits, int nq, int nk) {
    const double t = (double)p_drop * 256.0;
    if (t <= 0.0) { a.thr8 = 0u; a.thr_frac = 0u; a.inv_keep = 1.f; }
    else if (t >= 255.0) { a.thr8 = 255u; a.thr_frac = 0u; a.inv_keep = 256.f; }
    else {
        a.thr8 = (uint32_t)t;
        a.thr_frac = (uint32_t)((t - (double)a.thr8) * 65536.0 + 0.5);
        if (a.thr_frac > 65535u) { a.thr_frac = 0u; a.thr8 += 1u; }
        a.inv_keep = (float)(1.0 / (1.0 - ((double)a.thr8 + (double)a.thr_frac / 65536.0) / 256.0));
    }
    a.seed = seed;
    a.drop_on = (a.thr8 | a.thr_frac) ? 1 : 0;
    a.dropbits = (uint16_t*)dropbits;
    a.nqt16 = dropbits_nqt16(nq);
    a.nkt64 = dropbits_nkt64(nk);
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
// probabilities below 2^-log2_threshold of their row's largest one may be skipped by the ALiBi band (0 = visit everything).
// Default 30.  Process-wide (the "attn_band" knob of spn_set_tuning); meant for tests and ablations.
extern "C" int spn_set_tuning(const char* name, double value);
extern "C" void spn_attn_set_band(float log2_threshold) { spn_set_tuning("attn_band", log2_threshold < 0.f ? 0.f : log2_threshold); }

// floats of a caller-owned band buffer (spn_attn_fwd fills it, spn_attn_bwd of the same problem reuses it instead of recomputing)
extern "C" long spn_attn_band_elems(int b, int h, int kvh, int nq) { return 2l * b * h * ((nq + 63) / 64) + (long)b * kvh + 2l * b; }

// uint16 words of the dropout keep-bit buffer for a [b, h, nq, nk] attention (1 bit per score, whole 128x128 blocks)
extern "C" long spn_attn_dropbits_elems(int b, int h, int nq, int nk) {
    return (long)b * h * dropbits_nqt16(nq) * dropbits_nkt64(nk) * 64;
}

extern "C" int spn_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse, const uint8_t* kmask, const uint8_t* qmask,
                            const float* slopes, int b, int h, int kvh, int nq, int nk, int causal, float scale,
                            const long* strides, float p_drop, unsigned seed, void* dropbits, float* band, hipStream_t stream) {
    AttnArgs a;
    memset(&a, 0, sizeof(a));
    set_dropout(a, p_drop, seed, dropbits, nq, nk);
    SPN_REQUIRE(!a.drop_on || dropbits, "spn_attn_fwd: dropout needs the keep-bit buffer (spn_attn_dropbits_elems uint16 words)");
    a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.o = (bf16_t*)o; a.lse = lse;
    a.kmask = kmask; a.qmask = qmask; a.slopes = slopes; a.b = b; a.h = h; a.kvh = kvh; a.nq = nq; a.nk = nk; a.causal = causal;
    a.scale = scale;
    a.q_bs = strides[0]; a.q_ns = strides[1]; a.q_hs = strides[2];
    a.k_bs = strides[3]; a.k_ns = strides[4]; a.k_hs = strides[5];
    a.v_bs = strides[6]; a.v_ns = strides[7]; a.v_hs = strides[8];
    a.o_bs = strides[9]; a.o_ns = strides[10]; a.o_hs = strides[11];
    int rc = check_common(a);
    if (rc) return rc;
    SPN_REQUIRE(o && lse, "spn_attn_fwd: null output");
    SPN_REQUIRE((a.o_ns % 4) == 0 && (a.o_hs % 4) == 0 && (a.o_bs % 4) == 0, "spn_attn_fwd: o strides must be multiples of 4");
    rc = prepare_band(a, stream, band, false);
    if (rc) return rc;
    dim3 grid(cdiv(nq, 128), h, b);
    if (a.drop_on) hipLaunchKernelGGL(attn_fwd_kernel<true>, grid, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(attn_fwd_kernel<false>, grid, dim3(256), 0, stream, a);
    SPN_LAUNCH_CHECK();
    return SPN_OK;
}

// strides: 21 longs = forward's 12 followed by {dq_bs,dq_ns,dq_hs, dk_bs,dk_ns,dk_hs, dv_bs,dv_ns,dv_hs}.
// delta: workspace of b*h*nq floats.  dslope: [h] fp32, accumulated with atomics (zero it first), or null.
extern "C" int spn_attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* d_o,
                            const float* lse, float* delta, void* dq, void* dk, void* dv, float* dslope,
                            const uint8_t* kmask, const uint8_t* qmask, const float* slopes, int b, int h, int kvh, int nq, int nk,
                            int causal, float scale, const long* strides, float p_drop, const void* dropbits, const float* band,
                            hipStream_t stream) {
    AttnArgs a;
    memset(&a, 0, sizeof(a));
    set_dropout(a, p_drop, 0, const_cast<void*>(dropbits), nq, nk);
    SPN_REQUIRE(!a.drop_on || dropbits, "spn_attn_bwd: dropout needs the keep bits written by spn_attn_fwd");
    a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.o = (bf16_t*)o; a.d_o = (const bf16_t*)d_o;
    a.lse = const_cast<float*>(lse); a.delta = delta; a.dq = (bf16_t*)dq; a.dk = (bf16_t*)dk; a.dv = (bf16_t*)dv;
    a.dslope = dslope; a.kmask = kmask; a.qmask = qmask; a.slopes = slopes; a.b = b; a.h = h; a.kvh = kvh; a.nq = nq; a.nk = nk;
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
    rc = prepare_band(a, stream, const_cast<float*>(band), true);   // the bounds the forward computed for this q / k / mask (null: visit all)
    if (rc) return rc;
    // dQ first: it computes delta = rowsum(O * dO) in its prologue and stores it for dK/dV
    const dim3 gq(cdiv(nq, 128), h, b);
    if (a.drop_on && dslope) hipLaunchKernelGGL((attn_bwd_dq_kernel<true, true>), gq, dim3(256), 0, stream, a);
    else if (a.drop_on) hipLaunchKernelGGL((attn_bwd_dq_kernel<true, false>), gq, dim3(256), 0, stream, a);
    else if (dslope) hipLaunchKernelGGL((attn_bwd_dq_kernel<false, true>), gq, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((attn_bwd_dq_kernel<false, false>), gq, dim3(256), 0, stream, a);
    launch_attn_dkv(a, stream);
    SPN_LAUNCH_CHECK();
    return SPN_OK;
}
