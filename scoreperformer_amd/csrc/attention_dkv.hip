// dK / dV of the fused attention (see attention.hip).  grid (ceil(nk/128), kv_heads, b), 256 threads.
// Wave w owns 32 keys (two 16-key blocks) j0 + 32w .. +31 and loops over all query tiles of all heads that share its K/V head
// (multi-query attention: 8 heads), so dK/dV are reduced over heads in registers, without atomics.
// Orientation S = Q K^T (lane = one key column): P and dS in C-layout are the B operands of dV^T = dO^T P and dK^T = Q^T dS.
// 32 keys per wave halve the LDS fragment traffic per MFMA relative to 16 (each Q / dO fragment feeds two key blocks).
#include "attention_common.h"

namespace spn_attn {
namespace {

template <int CLS, bool DROP>
__device__ __forceinline__ void dkv_tile(const char* q_tile, const char* qt_tile, const char* do_tile, const char* dot_tile,
                                         const float* nl2_s, const float* dl_s, const bf16x8 (&kf)[2][2], const bf16x8 (&vf)[2][2],
                                         f32x4 (&dk)[4][2], f32x4 (&dv)[4][2], const float (&jf)[2], const bool (&key_ok)[2],
                                         int ioff, float c1, float slope2, bool causal, int lane, int g,
                                         const uint2 (&bw)[4], const int (&boff)[2], float log2_inv_keep, float keep_prob) {
    const float sjf[2] = {slope2 * jf[0], slope2 * jf[1]};
#pragma unroll
    for (int u = 0; u < 2; ++u) {               // two halves of 32 query rows
        f32x4 p[2][2], ds[2][2];                // [qq][kb]
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) {
            const int qb = 2 * u + qq;
            const bf16x8 qa0 = frag_rows(q_tile, 16 * qb, 0, lane), qa1 = frag_rows(q_tile, 16 * qb, 1, lane);
            const bf16x8 da0 = frag_rows(do_tile, 16 * qb, 0, lane), da1 = frag_rows(do_tile, 16 * qb, 1, lane);
            // dropout: P' = P / keep_prob (the log2 shift rides on -lse), dS = P' * (keep * dP - delta * keep_prob)
            f32x4 n4 = *reinterpret_cast<const f32x4*>(nl2_s + 16 * qb + 4 * g);
            f32x4 d4 = *reinterpret_cast<const f32x4*>(dl_s + 16 * qb + 4 * g);
            if (DROP) { n4 += log2_inv_keep; d4 *= keep_prob; }
            const float ib = (float)(ioff + 16 * qb + 4 * g);
            // linear-bias tiles: slope2*(j - i) - lse = [slope2*j] + [-slope2*i - lse]: the row part once per 4 rows (shared by both
            // key blocks), the key part once per lane -- one add per score instead of a subtract and an fma
            f32x4 rowt = n4;
            if (CLS == T_LEFT) rowt -= slope2 * (ib + f32x4{0.f, 1.f, 2.f, 3.f});
            else if (CLS == T_RIGHT) rowt += slope2 * (ib + f32x4{0.f, 1.f, 2.f, 3.f});
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f}, acc2 = f32x4{0.f, 0.f, 0.f, 0.f};
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa0, kf[kb][0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa1, kf[kb][1], acc, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(da0, vf[kb][0], acc2, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(da1, vf[kb][1], acc2, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float i_f = ib + (float)r;
                    float e;   // log2-domain score minus lse
                    if (CLS == T_LEFT) e = fmaf(acc[r], c1, sjf[kb] + rowt[r]);
                    else if (CLS == T_RIGHT) e = fmaf(acc[r], c1, rowt[r] - sjf[kb]);
                    else {
                        const bool ok = key_ok[kb] && (!causal || jf[kb] <= i_f);
                        const float t = fmaf(-slope2, fabsf(jf[kb] - i_f), acc[r] * c1);
                        e = (ok ? t : NEG_FILL) + n4[r];
                    }
                    const float pv = fast_exp2(e);
                    if (DROP) {   // row r of this lane's key column: halfword r of bw[qb], bit boff[kb] (see attention_common.h)
                        const uint32_t keepm = (uint32_t)__builtin_amdgcn_sbfe((int)(r < 2 ? bw[qb].x : bw[qb].y), boff[kb] + 16 * (r & 1), 1);
                        const float pm = __uint_as_float(__float_as_uint(pv) & keepm);                     // P_dropped feeds dV
                        p[qq][kb][r] = pm;
                        ds[qq][kb][r] = fmaf(pm, acc2[r], -(pv * d4[r]));   // = pv * (keep ? dP : 0) - pv * delta: one AND less
                    } else {
                        p[qq][kb][r] = pv;
                        ds[qq][kb][r] = pv * (acc2[r] - d4[r]);
                    }
                }
            }
        }
        bf16x8 pf[2], dsf[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) { pf[kb] = pack8(p[0][kb], p[1][kb]); dsf[kb] = pack8(ds[0][kb], ds[1][kb]); }
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
    __shared__ __attribute__((aligned(16))) char smem[2 * 32768 + 2 * 512];
    float* stat_s = reinterpret_cast<float*>(smem + 65536);   // [2][-lse * log2e per row (NEG_FILL for dead rows: p = 0) | delta per row]

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g = lane >> 4, c = lane & 15;
    int bi = blockIdx.z, kh = blockIdx.y, jt = blockIdx.x;
    if (!causal_order(a, false, bi, kh, jt) && !a.causal) xcd_batch_coords(a, bi, kh, jt);
    const int j0 = jt * 128;
    const int off = a.nk - a.nq;
    const int heads_per_kv = a.h / a.kvh;
    const bf16_t* kp = a.k + bi * a.k_bs + kh * a.k_hs;
    const bf16_t* vp = a.v + bi * a.v_bs + kh * a.v_hs;
    const float c1 = a.scale * LOG2E;
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
    const int jw_lo = j0 + 32 * w, jw_hi = jw_lo + 31;

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
    const int n_iter = (nqt - t_first) > 0 ? (nqt - t_first) * heads_per_kv : 0;

    // keep bits of the forward: this lane's key column c of key block kb sits in forward lanes (c>>2)*16 + 4g + r (r = its 4 rows),
    // bit 4*kb_f + (c&3) with kb_f = the 16-key block index inside the forward's 64-key tile
    const long bstride = (long)a.nkt64 * 64;
    const int boff[2] = {4 * (2 * (w & 1)) + (c & 3), 4 * (2 * (w & 1) + 1) + (c & 3)};
    const long bit_lane = (long)((j0 + 32 * w) / 64) * 64 + (c >> 2) * 16 + 4 * g;
    const float log2_inv_keep = DROP ? __builtin_log2f(a.inv_keep) : 0.f, keep_prob = DROP ? 1.f / a.inv_keep : 1.f;
    uint2 bw[4], bwn[4];
#pragma unroll
    for (int qb = 0; qb < 4; ++qb) { bw[qb] = make_uint2(0, 0); bwn[qb] = make_uint2(0, 0); }

    const int wv = __builtin_amdgcn_readfirstlane(w);
    uint32_t voQa[2], voQt[2], voDa[2], voDt[2];   // this wave's two 1 KiB pieces of each image: LDS chunk <- inverse-swizzled source chunk
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int L = (wv * 2 + i) * 64 + lane, row = L >> 3, ch = L & 7;
        const int ca = (ch ^ (row & 7)) << 3, ct = (ch ^ (((row >> 1) & 3) << 1)) << 3;
        voQa[i] = (uint32_t)(((long)row * a.q_ns + ca) * 2); voQt[i] = (uint32_t)(((long)row * a.q_ns + ct) * 2);
        voDa[i] = (uint32_t)(((long)row * a.o_ns + ca) * 2); voDt[i] = (uint32_t)(((long)row * a.o_ns + ct) * 2);
    }
    float lreg = 0.f, dreg = 0.f;
    int n_issued = 0;   // stage of the next issue = n_issued & 1
    auto issue = [&](int it) {
        // pointers and strides of the tile request: read from the kernel-argument segment here, once per tile (attention_common.h)
        AttnKernargPtr ai = attn_kernarg();
        asm volatile("" : "+s"(ai));
        const int hh = kh * heads_per_kv + it / (nqt - t_first);
        const int i0 = (t_first + it % (nqt - t_first)) * 64;
        if (DROP) {
            const uint16_t* bp = ai->dropbits + ((long)(bi * ai->h + hh) * ai->nqt16 + i0 / 16) * bstride + bit_lane;
#pragma unroll
            for (int qb = 0; qb < 4; ++qb) bwn[qb] = *reinterpret_cast<const uint2*>(bp + qb * bstride);
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
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsQ, (__attribute__((address_space(3))) void*)(base + i * 1024), 16, voQa[i], sq, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsQ, (__attribute__((address_space(3))) void*)(base + 8192 + i * 1024), 16, voQt[i], sq, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsD, (__attribute__((address_space(3))) void*)(base + 16384 + i * 1024), 16, voDa[i], sd, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsD, (__attribute__((address_space(3))) void*)(base + 24576 + i * 1024), 16, voDt[i], sd, 0, 0);
        }
        ++n_issued;
        if (tid < 64) {
            const int i = i0 + tid;
            const long si = ((long)bi * ai->h + hh) * nq + i;
            const float lse_i = i < nq ? ai->lse[si] : NEG_FILL;
            lreg = lse_i > -1e37f ? -lse_i * LOG2E : NEG_FILL;   // rows beyond nq / fully masked rows: p = exp2(t + NEG) = 0
            dreg = i < nq ? ai->delta[si] : 0.f;
        }
    };
    // ALiBi band (attention.hip): a (head, query tile) whose reach ends before this block's 128 keys is not visited.  Which iterations
    // are live is decided ONCE, by all 256 threads in parallel, into a bit mask in LDS: decided tile by tile inside the loop, the two
    // dependent global loads of band_reach (and ~110 scalar / vector instructions around them) sat in front of every tile request.
    constexpr int LIVE_WORDS = 16;   // up to 1024 (head, query tile) iterations; longer walks decide on the fly as before
    __shared__ unsigned long long live_mask[LIVE_WORDS];
    auto is_live = [&](int it) {
        const int hh = kh * heads_per_kv + it / (nqt - t_first);
        const int i0 = (t_first + it % (nqt - t_first)) * 64;
        const float reach = band_reach(a, bi, hh, kh, i0 / 64, 1, c1, a.slopes ? a.slopes[hh] * LOG2E : 0.f);
        if (!(reach < 1.0e9f)) return true;
        const float r_lo = (float)(i0 + off), r_hi = r_lo + 63.f;
        return (float)j0 <= r_hi + reach && (float)(j0 + 127) >= r_lo - reach;
    };
    const bool masked_walk = n_iter <= 64 * LIVE_WORDS;
    if (masked_walk) {
        for (int base = 0; base < n_iter; base += 256) {
            const int it = base + tid;
            const unsigned long long m = __ballot(it < n_iter && is_live(it));
            if (lane == 0) live_mask[(base >> 6) + w] = m;
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
            if (is_live(it)) break;
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
        float* nl2_s = stat_s + stage * 128;
        float* dl_s = nl2_s + 64;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces, the row statistics and the keep-bit words have landed
        if (tid < 64) { nl2_s[tid] = lreg; dl_s[tid] = dreg; }
#pragma unroll
        for (int qb = 0; qb < 4; ++qb) bw[qb] = bwn[qb];
        __syncthreads();   // tile visible to all; everybody is done with the other stage
        if (it_next < n_iter) issue(it_next);

        // tile class of this wave's 32 keys against the 64 rows (key coordinates i + off)
        const int r_lo = i0 + off, r_hi = r_lo + 63;
        int cls = T_GEN;
        if (a.causal && jw_lo > r_hi) cls = T_SKIP;
        else if (keys_full && jw_hi <= r_lo) cls = T_LEFT;              // j - i <= 0 everywhere
        else if (keys_full && !a.causal && jw_lo >= r_hi) cls = T_RIGHT;
        if (cls == T_SKIP) { it = it_next; continue; }

        if (cls == T_LEFT) dkv_tile<T_LEFT, DROP>(q_tile, qt_tile, do_tile, dot_tile, nl2_s, dl_s, kf, vf, dk, dv, jf, key_ok, i0 + off, c1, slope2, a.causal, lane, g, bw, boff, log2_inv_keep, keep_prob);
        else if (cls == T_RIGHT) dkv_tile<T_RIGHT, DROP>(q_tile, qt_tile, do_tile, dot_tile, nl2_s, dl_s, kf, vf, dk, dv, jf, key_ok, i0 + off, c1, slope2, a.causal, lane, g, bw, boff, log2_inv_keep, keep_prob);
        else dkv_tile<T_GEN, DROP>(q_tile, qt_tile, do_tile, dot_tile, nl2_s, dl_s, kf, vf, dk, dv, jf, key_ok, i0 + off, c1, slope2, a.causal, lane, g, bw, boff, log2_inv_keep, keep_prob);
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
