// One decoder layer pair (self-attention block + feed-forward block) of a cached decode step as ONE persistent launch.
//
// The graph engine (decode.hip, decode.py::_step_fused) runs such a pair as five dependent launches: LayerNorm + q|k|v GEMV, split-key
// attention, output projection (+ merge of the split partials, + residual), LayerNorm + gated input projection, output projection
// (+ residual).  Every one of them is a chain "kernel boundary -> input vector from memory -> a few hundred bytes of arithmetic per lane
// -> store": ~6 us each at batch 1, whatever the arithmetic.  Here the five phases live in one launch of G workgroups (512 threads,
// one per CU) and hand their vectors over INSIDE the launch:
//   * the data is the flag: every value travels as one naturally aligned 8-byte granule {tag = epoch, fp32 bits}, written by ONE
//     agent-scope (write-through, `sc1`) store and polled with agent-scope loads until every tag a workgroup needs carries the epoch of
//     this phase of this note -- no fences, no counters, no barrier (cdna_hip_programming.md Guideline 16, form R2);
//   * weight rows do not depend on the hand-off: every wave requests the rows of its NEXT phase before it starts polling, so the
//     weight stream runs under the hop instead of behind it;
//   * epoch = tick * 256 + 8 * layer + phase + 1 with `tick` a device counter that the last pair launch of a note advances: tags never
//     repeat within a render (the host zeroes the granule buffers when a render starts), also when a session re-decodes a position.
// Arithmetic, association order and rounding are those of the five kernels it replaces, through helpers both files share: one wave per
// output row, lanes own k = 4 lane + 256 c, two partial sums per lane by packed FMA (common.h dec_dot4); LayerNorm statistics per wave
// over that layout (common.h dec_ln_stats); the split-key batches of dec_attn2_kernel (decode_attn.h: 8 keys per lane group, 32 lane
// groups, two-level group merge); the 16-record split merge of dec_fused_gemv_kernel's prologue: the residual stream leaves bit-identical.
// A decode step is bound by the INSTRUCTION COUNT of these single-wave chains between the hand-offs (tools/pk_probe.hip: a wave issues
// ~one VALU instruction per 5 clocks whatever its neighbour on the SIMD does; packed fp32 issues at full rate), not by bytes or flops.
// Every poll loop is bounded: a workgroup that never sees its tags sets *err and stops polling (the launch completes with garbage and
// the host raises), it never hangs the device.
//
// Replaces (per note) the layer body of `ScorePerformerMixedLMWrapper.unmask_tokens` (wrappers.py:325-407 ->
// modules/transformer/transformer.py:159-221, attention.py:107-222, feedforward.py:13-64) for pre-norm ('a', 'f') decoders.
#include <algorithm>
#include "common.h"
#include "decode_attn.h"
#include "decode_sample.h"
#include "../../include/spn.h"   // spn_dec_pair_args

#pragma clang fp contract(off)   // as decode.hip: the same source expression must round the same way in both files

namespace {

typedef __attribute__((address_space(1))) unsigned long long gu64;
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
constexpr int NT = 512, NW = 8;                 // threads / waves per workgroup
constexpr unsigned SPIN_LIMIT = 1u << 18;        // polls per hand-off before giving up (~0.3 s)

__device__ __forceinline__ gu64* as_global(unsigned long long* p) { return (gu64*)(uintptr_t)p; }

__device__ __forceinline__ void put(unsigned long long* g, unsigned epoch, float v) {
    __hip_atomic_store(as_global(g), ((unsigned long long)epoch << 32) | __float_as_uint(v), RLX_AGENT);
}

// the counter hash of dec_head_kernel<true>'s draw (decode.hip dec_mix)
__device__ __forceinline__ unsigned dec_mix32(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// Hand-off read: thread tid polls the granule PAIRS p = tid + 512 k (granules 2p, 2p + 1; k < 2, n even and <= 2048) with one 16-byte
// agent-scope load each until both tags carry `epoch`, then leaves the values in xs.  One load per lane and pass at n <= 1024.  Measured
// against one polling wave per workgroup with 8 loads per lane (fewer pollers on the hot lines, but every pass is 8 serial loads long):
// the hops were 1-2 us SLOWER that way.
typedef unsigned int pair_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void gather(unsigned long long* g, int n, unsigned epoch, float* xs, int tid, int* err) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)g, 0, n * 8, 0x00020000);
    pair_u32x4 v[2] = {pair_u32x4{0u, 0u, 0u, 0u}, pair_u32x4{0u, 0u, 0u, 0u}};
    unsigned spins = 0;
    for (;;) {
        bool ok = true;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int p = tid + NT * k;
            if (2 * p < n) {
                // aux 16 = sc1: served past the L1, at agent scope; bit 31 = volatile: the load is re-issued in every pass (the intrinsic
                // is read-only and s_sleep touches no memory, so nothing else tells the optimiser that the granules change under it)
                v[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, p * 16, 0, (int)(16u | 0x80000000u));
                ok &= v[k][1] == epoch && v[k][3] == epoch;
            }
        }
        if (ok) break;
        if (++spins > SPIN_LIMIT) { *err = 1; break; }
        __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");   // belt and braces: a compiler barrier between two polling passes
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int p = tid + NT * k;
        if (2 * p < n) { xs[2 * p] = __uint_as_float(v[k][0]); xs[2 * p + 1] = __uint_as_float(v[k][2]); }
    }
}

// Hand-off write of a workgroup's 16 consecutive outputs (two per wave, staged in outv by lane 0 of every wave): ONE store instruction,
// 16 lanes x 8 bytes = one whole 128-byte line.  (One 8-byte store per wave made 16 partial writes of every line: the fabric serialises
// them, and the hop behind 512 such stores took 4.5 us instead of ~1.5.)
__device__ __forceinline__ void publish16(unsigned long long* g, int n0, int n, unsigned epoch, const float* outv, int tid) {
    __syncthreads();
    if (tid < 16 && n0 + tid < n) put(g + n0 + tid, epoch, outv[tid]);
}

__device__ __forceinline__ float pair_act(float g, int act) {
    return act == 0 ? g / (1.f + __expf(-g)) : 0.5f * g * (1.f + erff(g * 0.70710678118654752f));
}

// the affine parameters of a pre-norm for this thread's entries k = tid, tid + 256 (K <= 512), requested when the launch starts: read
// behind the statistics they were a dependent trip to memory (1.5-2 us: the adaptive (gamma | beta) row is written earlier in the same note)
struct NormRegs { float g[2], b[2]; };
__device__ __forceinline__ NormRegs norm_regs(int K, int mode, const float* gam, const float* bet, int tid) {
    NormRegs r;
    const float* be = mode == 2 ? gam + K : bet;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int k = tid + 256 * i;
        const bool in = tid < 256 && k < K && gam != nullptr;
        r.g[i] = in ? gam[k] : 1.f;
        r.b[i] = in ? be[k] : 0.f;
    }
    return r;
}

// LayerNorm of xs[0 .. K), K <= 512, per WAVE and without a barrier: statistics by common.h dec_ln_stats (the arithmetic of
// dec_fused_gemv_kernel's norm), the normalised vector returned in the dot_rows layout (lane owns k = 4 lane + 256 c, c < 2) with the
// affine parameters held in that layout (NormRegs4); xs itself is left as gathered.
struct NormRegs4 { f32x4 g[2], b[2]; };
__device__ __forceinline__ NormRegs4 norm_regs4(int K, int mode, const float* gam, const float* bet, int lane) {
    NormRegs4 r;
    if (mode == 2 && gam != nullptr) {
        // adaptive (gamma | beta) row: written by OTHER workgroups -- one note earlier, possibly in this very launch -- into a buffer that
        // is reused every second note: read past the L2 (agent scope, aux 16 = sc1), or a later note of the launch finds the lines its
        // XCD's L2 kept from two notes ago
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)gam, 0, 2 * K * 4, 0x00020000);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int k = lane * 4 + c * 256;
            const bool in = k < K;
            const pair_u32x4 g = __builtin_amdgcn_raw_buffer_load_b128(rs, min(k, K - 4) * 4, 0, 16);
            const pair_u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(rs, (K + min(k, K - 4)) * 4, 0, 16);
            r.g[c] = in ? __builtin_bit_cast(f32x4, g) : f32x4{1.f, 1.f, 1.f, 1.f};
            r.b[c] = in ? __builtin_bit_cast(f32x4, b) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        return r;
    }
    const float* be = bet;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int k = lane * 4 + c * 256;
        const bool in = k < K && gam != nullptr;
        r.g[c] = in ? *reinterpret_cast<const f32x4*>(gam + k) : f32x4{1.f, 1.f, 1.f, 1.f};
        r.b[c] = in ? *reinterpret_cast<const f32x4*>(be + k) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    return r;
}
__device__ __forceinline__ void wave_norm(const float* xs, int K, bool affine, const NormRegs4& nr, float eps, int lane, f32x4 (&xv)[2]) {
#pragma unroll
    for (int c = 0; c < 2; ++c) xv[c] = *reinterpret_cast<const f32x4*>(xs + lane * 4 + c * 256);
    float mu, rs;
    dec_ln_stats<2>(xv, K, eps, lane, mu, rs);
#pragma unroll
    for (int c = 0; c < 2; ++c) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v = (xv[c][e] - mu) * rs;
            if (affine) v = v * nr.g[c][e] + nr.b[c][e];
            xv[c][e] = v;
        }
    }
}

// one weight row in registers: lane owns k = 4 lane + 256 c
template <int C>
__device__ __forceinline__ void load_row(f32x4 (&r)[C], const float* w, int K, int lane) {
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int k = lane * 4 + c * 256;
        r[c] = k < K ? *reinterpret_cast<const f32x4*>(w + k) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
}
// R rows against the vector in LDS: the vector's chunks are read ONCE for all rows and every step is branch-free (a row-by-row loop with
// an exec-masked branch per chunk cost ~190 ns per row: 3.1 us for the 16 products of a feed-forward wave).  Per row the arithmetic is
// the one of dec_fused_gemv_kernel: chunks ascending through common.h dec_dot4 (two partial sums per lane by packed FMA), folded, then
// the DPP ladder of wave_sum.
template <int C, int R>
__device__ __forceinline__ void dot_rows_x(const f32x4 (&r)[R][C], const f32x4 (&xv)[C], int K, int lane, float (&out)[R]);
template <int C, int R>
__device__ __forceinline__ void dot_rows(const f32x4 (&r)[R][C], const float* xs, int K, int lane, float (&out)[R]) {
    f32x4 xv[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int k = lane * 4 + c * 256;
        xv[c] = *reinterpret_cast<const f32x4*>(xs + min(k, 2044));
    }
    dot_rows_x<C, R>(r, xv, K, lane, out);
}
// the same with the vector already in registers (lane owns k = 4 lane + 256 c): wave_norm's output
template <int C, int R>
__device__ __forceinline__ void dot_rows_x(const f32x4 (&r)[R][C], const f32x4 (&xv)[C], int K, int lane, float (&out)[R]) {
    // entries past K: the rows are zero there (load_row), the vector may hold anything -- zeroed ONCE for all rows (fma(0, 0, a) = a,
    // which is what the chunk-skipping loops of decode.hip compute)
    f32x4 xz[C];
#pragma unroll
    for (int c = 0; c < C; ++c) xz[c] = lane * 4 + c * 256 < K ? xv[c] : f32x4{0.f, 0.f, 0.f, 0.f};
    f32x2 acc[R];
#pragma unroll
    for (int i = 0; i < R; ++i) {
        acc[i] = f32x2{0.f, 0.f};
#pragma unroll
        for (int c = 0; c < C; ++c) dec_dot4(acc[i], r[i][c], xz[c]);
    }
#pragma unroll
    for (int i = 0; i < R; ++i) out[i] = wave_sum(dec_fold(acc[i]));
}

// Roles.  Workgroups 0 .. h S - 1 ("A") own the q|k|v rows (the first ceil(N1 / 16) of them, 16 rows each) and one (head, split) of the
// attention each.  Workgroups h S .. h S + ceil(d / 16) - 1 ("B") own the d-row phases: the merge of a head's partials (the first h of
// them), the attention output projection and the feed-forward output projection, 16 rows each.  The last ceil(inner / 32) workgroups
// ("C") own the gated feed-forward rows, 32 each.  Every weight row a workgroup needs for a layer is requested as soon as it has
// published its last result of the PREVIOUS layer (at the start of the launch for the first one), in the order of use: loads return in
// order, so a request in front of a poll whose data is about to land would hold every pass of that poll back by a trip to HBM -- with the
// roles split this way a workgroup's next inputs are always several hand-offs away when it requests weights.
//
// A CHAIN of n consecutive layer pairs runs in one launch (`layers`: device array): the residual stream goes from a pair's last phase to
// the next pair's first one as granules too (gxo; the projection workgroups keep their own rows as the next residual in registers), so
// the ~1.6 us between two launches and the first weight trip of every pair but the first leave the critical path.
// n_notes consecutive notes in ONE launch (round 5; needs the embed, front, tail and head phases): every role loops over the notes, position
// and tick advance in registers, and the one thing a note needs from its predecessor that used to cross a kernel boundary -- the chosen
// tokens -- travels as granules from the head's winners to the embed workgroups (spn_dec_chain_ext.gt).  Everything else one note writes
// and a later note reads from another workgroup is written through and read past the L2 (agent scope): the XCDs' L2s are only made
// coherent at kernel boundaries.  Key / value rows: written through by their one writer, and no workgroup touches the lines of row t
// before note t + 1 (at note t the row comes from the q | k | v granules).  AdaLN rows: two buffers by note parity, written through,
// read with agent-scope loads.  Saves the graph edge and the first trips of a launch (position, argument records) per note.
// pos_p / tick_p / err_p: the chain's position, tick and error word (= layers[0].pos / .tick / .err) as kernel arguments: read through the
// argument record in device memory they were a dependent second trip to memory at the start of every note.
__global__ __launch_bounds__(NT) void dec_pair_kernel(const spn_dec_pair_args* __restrict__ layers0, int n_layers,
                                                      const spn_dec_chain_ext* __restrict__ ext0, const int* __restrict__ pos_p,
                                                      const int* __restrict__ tick_p, int* __restrict__ err_p, int n_notes) {
    __shared__ __attribute__((aligned(16))) float xs[2048];
    __shared__ float red[8];
    __shared__ float sm[DEC_G], sl[DEC_G];
    __shared__ __attribute__((aligned(16))) float so[DEC_G][64];
    __shared__ float sm2[8], sl2[8];
    __shared__ float so2[8][64];
    __shared__ __attribute__((aligned(16))) float qs[192];
    __shared__ float outv[16];
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x;
    const spn_dec_pair_args* const layers = layers0;
    const spn_dec_chain_ext* const ext = ext0;
#define a0 (layers[0])
    const int d = a0.d, h = a0.h, kvh = a0.kvh, S = a0.S, inner = a0.inner;   // the same in every pair of a chain (checked by the host)
    // (position, tick and error word are requested here and first USED behind the weight requests of the role: read up front, the error
    // check alone held every workgroup's first weight load back by a trip to memory)
    const int err_in = *err_p;
    int* const err = err_p;
    // Per-phase time stamps (spn_dec_pair_args.stamps, tools/bench_dec_pair.py) are a tuning aid and compiled in with -DSPN_DEC_STAMPS only
    // (tools/build_variant.py decode_layer.hip stamps_spn.so -DSPN_DEC_STAMPS): each of the ~45 stamp sites of a note is a scalar load of
    // the record's pointer, a wait and a branch on the critical path, whether or not anybody asked for stamps.
#ifdef SPN_DEC_STAMPS
#define STAMP(k_) do { if (a.stamps && tid == 0) a.stamps[(long)b * 8 + (k_)] = (long long)__builtin_amdgcn_s_memrealtime(); } while (0)
#define STAMP_OF(rec_, k_) do { if ((rec_).stamps && tid == 0) (rec_).stamps[(long)b * 8 + (k_)] = (long long)__builtin_amdgcn_s_memrealtime(); } while (0)
    if (a0.stamps && tid == 0) a0.stamps[(long)b * 8 + (b < a0.h * a0.S ? 6 : (b < a0.h * a0.S + (a0.d + 15) / 16 ? 5 : 7))] = (long long)__builtin_amdgcn_s_memrealtime();   // kernel entry
#else
#define STAMP(k_) ((void)0)
#define STAMP_OF(rec_, k_) ((void)0)
#endif
    // embed phase: key q's first column, row width and table in lane q.  Requested HERE, by every workgroup, before the record's scalar
    // fields (which decide who runs the phase) have arrived: behind them these loads were a third dependent trip at the start of a note.
    const int kq_ = min(lane, 15);
    const int l_c0 = ext ? ext->ecol0[kq_] : 0, l_w = ext ? ext->ewidth[kq_] : 0;
    const unsigned long long l_tb = ext ? reinterpret_cast<unsigned long long>(ext->etable[kq_]) : 0ull;
    const int t_in = *pos_p;   // (requested here; made wave-uniform below, where the first user -- the AdaLN row set -- needs it anyway)
    unsigned ebase = (unsigned)(*tick_p) * 256u + 1u;
    const int N1 = (h + 2 * kvh) * 64;
    const int nA = h * S, nB = (d + 15) / 16;
    // Optional phases around the chain (spn_dec_chain_ext): in FRONT of the first pair the two input projections of the note (B
    // workgroups: x0 = Wm . xin + bm, then x = Wp . (LN?(x0) | context row | style row) + bp), BEHIND the last pair the LM head's input
    // projection e = Wh . LN(x) (A workgroups).  Their epochs use pair number 31.
    const bool front = ext && ext->Wm, tail = ext && ext->Wh, head = tail && ext->hn > 0;
    unsigned efront = ebase + 8u * 31u;
    unsigned ehead = efront + 2u;   // e (tail -> head workgroups); + 1: the per-key partial maxima
    unsigned eemb = efront + 4u;    // the projected token embeddings (attention workgroups -> front); + 1: the chosen tokens (winners -> embed)
    const bool emb = front && ext->en > 0;
    // adaptive norms read their (gamma | beta) rows from the buffer of this note's parity (spn_dec_chain_ext.ada_par)
    // (wave-uniform by construction: through readfirstlane it lives in scalar registers -- as a per-lane value it was spilled to scratch)
    int t = __builtin_amdgcn_readfirstlane(t_in);   // scalar: every address derived from the position is scalar arithmetic
    long apar = (ext && (t & 1)) ? ext->ada_par : 0;
#define ADA(mode_, ptr_) (((mode_) == 2 && (ptr_)) ? (ptr_) + apar : (ptr_))
    // the next note of this launch: position + 1, a fresh epoch block (tick + 1), the other set of AdaLN rows
    auto next_note = [&]() __attribute__((always_inline)) {
        t += 1; ebase += 256u; efront += 256u; ehead += 256u; eemb += 256u;
        apar = (ext && (t & 1)) ? ext->ada_par : 0;
    };

    if (b < nA) {
        // ================================================ A: q|k|v rows, attention split ==================================================
        const int r0 = b * 16 + 2 * w;                      // this wave's rows of phase 1: r0, r0 + 1
        const bool own1 = b * 16 < N1;
        const int hi = b / S, sp = b - hi * S;
        const int kh = kvh == 1 ? 0 : hi;
        const long cw = (long)kvh * 64;
        f32x4 wq[2][2];
        NormRegs4 n1;
        auto request = [&](const spn_dec_pair_args& a) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                wq[i][0] = wq[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (r0 + i < N1) load_row<2>(wq[i], a.Wqkv + (long)(r0 + i) * a.ld_qkv, d, lane);
            }
            n1 = norm_regs4(d, a.norm1, ADA(a.norm1, a.gam1), a.bet1, lane);
        };
        // Key / value rows of this workgroup's split, requested BEFORE the query exists: the split's key range depends on the query only
        // through the first key inside the ALiBi reach (j_lo), which is rounded down to a multiple of 256 (more keys than necessary, never
        // fewer) and therefore almost always equals the previous note's (jlo[head], written below).  The rows are then in registers when
        // q arrives (one trip to the Infinity Cache, ~1.5-2 us, off the critical path); a wrong guess falls back to the loads behind q.
        const int grp = lane >> 4, l16 = lane & 15;
        f32x4 k4[DEC_NU], v4[DEC_NU];
        bool pre = false;
        int jlo_guess = 0;
        auto prefetch = [&](const spn_dec_pair_args& a) __attribute__((always_inline)) {
            pre = false;
            if (a.jlo) {
                // jlo[] and kmax2[] are read and written by different workgroups of the SAME launch without ordering (relaxed agent-scope
                // accesses, no hand-off): benign by construction -- jlo is only a GUESS of which rows to request early (a stale or a fresh
                // value both work: a wrong guess reloads behind q, see `hit` below), and kmax2 is a running maximum into which every reader
                // folds this note's |k_new|^2 itself, so it sees an upper bound of the reach whether or not another workgroup's atomicMax
                // has landed yet; both decode paths round j_lo down to a multiple of 256, so the summation ranges do not depend on it either.
                jlo_guess = min(__hip_atomic_load(a.jlo + hi, RLX_AGENT), t);
                const int chunk = (t + 1 - jlo_guess + S - 1) / S;
                const int j0 = jlo_guess + sp * chunk, j1 = min(t + 1, j0 + chunk);
                const int jb0 = j0 + w * 4 + grp;
                pre = true;
                if (jb0 < j1) {
#pragma unroll
                    for (int u = 0; u < DEC_NU; ++u) {
                        const int j = min(min(jb0 + DEC_G * u, j1 - 1), t - 1);   // row t does not exist yet: patched from q|k|v below
                        k4[u] = *reinterpret_cast<const f32x4*>(a.kcache + (long)max(j, 0) * cw + kh * 64 + l16 * 4);
                        v4[u] = *reinterpret_cast<const f32x4*>(a.vcache + (long)max(j, 0) * cw + kh * 64 + l16 * 4);
                    }
                }
            }
        };
        for (int note = 0; note < n_notes; ++note) {
        // (the early key / value rows are only conditionally reloaded below: without this the previous note's values count as live across
        // the embed phase -- 64 registers on top of its weight rows)
#pragma unroll
        for (int u = 0; u < DEC_NU; ++u) k4[u] = v4[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        pre = false;
        if (emb && b * 8 * ext->eR < 2 * ext->eN) {
            // ---- embed: xin = We . LN(concat_k table_k[token_k]) + be for both sequences (dec_embed_proj_kernel, decode.hip): this workgroup's
            //      8 eR rows of ONE sequence; every workgroup rebuilds that sequence's embedding in LDS ----
            // The start of a note is a chain of dependent trips to memory (position -> tokens -> table rows): everything that does not depend
            // on the position is requested first -- weight rows, the norm's affine rows in the dot-product layout, and per column of this
            // thread (c = tid + 512 i) the table base, row width and key -- so that exactly those three trips remain (round 5: the
            // column -> key search and the table pointer sat inside the gather loop, behind the tokens: 5 us from the position to the rows).
            const int N = ext->eN, R = ext->eR, D = ext->eD, gr0 = (b * 8 + w) * R;
            const int seq = (b * 8 * R) / N;
            f32x4 we[2][8];
            float bev[2] = {0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 2; ++i) {
#pragma unroll
                for (int c = 0; c < 8; ++c) we[i][c] = f32x4{0.f, 0.f, 0.f, 0.f};
                const int n = gr0 + i - seq * N;
                if (i < R && gr0 + i < 2 * N) {   // weight row and bias do not depend on the tokens: requested first
                    load_row<8>(we[i], ext->We + (long)n * ext->ld_e, D, lane);
                    if (ext->be) bev[i] = ext->be[n];
                }
            }
            f32x4 eg[8], eb[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int k = lane * 4 + c * 256;
                const bool in = k < D && ext->egamma != nullptr;
                eg[c] = in ? *reinterpret_cast<const f32x4*>(ext->egamma + k) : f32x4{1.f, 1.f, 1.f, 1.f};
                eb[c] = in ? *reinterpret_cast<const f32x4*>(ext->ebeta + k) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
            // (key of every column from the keys' first columns -- lane q holds key q's fields, handed round by readlane / bpermute: indexing
            // the record with a per-lane key was one more trip to memory behind the weight rows, a scalar loop over the keys one trip per key)
            const float* cbase[4];
            int cwid[4], ckey[4];
            {
                const int en = ext->en;
                int cc[4], kk[4] = {0, 0, 0, 0};
#pragma unroll
                for (int i = 0; i < 4; ++i) cc[i] = min(tid + NT * i, D - 1);
#pragma unroll
                for (int q = 1; q < 16; ++q) {
                    const int cq = q < en ? __builtin_amdgcn_readlane(l_c0, q) : 0x7fffffff;
#pragma unroll
                    for (int i = 0; i < 4; ++i) kk[i] = cc[i] >= cq ? q : kk[i];
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int c0 = __builtin_amdgcn_ds_bpermute(kk[i] << 2, l_c0);
                    const unsigned lo = (unsigned)__builtin_amdgcn_ds_bpermute(kk[i] << 2, (int)(unsigned)l_tb);
                    const unsigned hi = (unsigned)__builtin_amdgcn_ds_bpermute(kk[i] << 2, (int)(unsigned)(l_tb >> 32));
                    ckey[i] = kk[i];
                    cwid[i] = __builtin_amdgcn_ds_bpermute(kk[i] << 2, l_w);
                    cbase[i] = reinterpret_cast<const float*>(((unsigned long long)hi << 32) | lo) + (cc[i] - c0);
                }
            }
            if (!err_in) {
                const long* tok = (seq ? ext->tok_b : ext->tok_a) + (long)(t + seq) * ext->etok_ld;
                if (n_layers > 1 && t >= 0) STAMP_OF(layers[1], 5);   // embed: position known
                int tq = 0;   // the tuple's ids: lane c holds column c, handed to the table columns by bpermute
                if (note > 0 && seq == 0) {
                    // a later note of the launch: the cells the head decodes come from its winners as granules (the PREVIOUS note's epoch
                    // block), the other columns of the row are given input
                    int qsel = -1;
                    for (int q = 0; q < ext->hn; ++q) if (ext->hdim[q] == lane) qsel = q;
                    if (lane < ext->en) {
                        if (qsel < 0) tq = (int)tok[lane];
                        else {
                            unsigned spins = 0;
                            for (;;) {
                                const unsigned long long x = __hip_atomic_load(as_global(ext->gt + qsel), RLX_AGENT);
                                tq = (int)(unsigned)x;
                                if ((unsigned)(x >> 32) == eemb - 256u + 1u) break;
                                if (++spins > SPIN_LIMIT) { *err = 3; break; }
                                __builtin_amdgcn_s_sleep(1);
                            }
                        }
                    }
                } else if (lane < ext->en) tq = (int)tok[lane];
                float val[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int tk = __builtin_amdgcn_ds_bpermute(ckey[i] << 2, tq);
                    val[i] = cbase[i][(long)tk * cwid[i]];
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) if (tid + NT * i < D) xs[tid + NT * i] = val[i];
                __syncthreads();
                if (n_layers > 1) STAMP_OF(layers[1], 6);   // embed: table rows gathered
                f32x4 xv[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const int k = lane * 4 + c * 256;
                    xv[c] = k < D ? *reinterpret_cast<const f32x4*>(xs + k) : f32x4{0.f, 0.f, 0.f, 0.f};
                }
                if (ext->egamma) {   // statistics per wave (common.h dec_ln_stats), the normalisation of dec_embed_proj_kernel in registers
                    float mu, rs;
                    dec_ln_stats<8>(xv, D, ext->eeps, lane, mu, rs);
#pragma unroll
                    for (int c = 0; c < 8; ++c)
#pragma unroll
                        for (int e = 0; e < 4; ++e) xv[c][e] = (xv[c][e] - mu) * rs * eg[c][e] + eb[c][e];
                }
                float y[2];
                dot_rows_x<8, 2>(we, xv, D, lane, y);
                if (lane == 0) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) if (i < R) outv[w * R + i] = ext->be ? y[i] + bev[i] : y[i];
                }
                __syncthreads();
                if (tid < 8 * R && b * 8 * R + tid < 2 * N) put(ext->gin + b * 8 * R + tid, eemb, outv[tid]);
                if (n_layers > 1) STAMP_OF(layers[1], 7);   // embed: rows out
                __syncthreads();   // xs / outv are reused below
            }
        }
        request(a0);
        if (own1 && !front) for (int k = tid; k < d; k += NT) xs[k] = a0.x[k];
        if (err_in) return;   // an earlier launch of this render timed out: do not wait again
        for (int l = 0; l < n_layers; ++l) {
            const spn_dec_pair_args& a = layers[l];
            const unsigned e0 = ebase + 8u * (unsigned)a.layer;   // layer < 31
            STAMP(0);
            // (first pair, workgroups with q|k|v rows: their x and weight loads were just issued and phase 1 waits for them -- loads return in
            // order, so the key / value requests go out behind phase 1 there)
            const bool late = l == 0 && own1;
            if (!late) prefetch(a);
            // slope and running max |k|^2 do not depend on this note's q either (the other workgroups' atomicMax below only folds in
            // |k_new|^2, which is folded in here anyway): read behind q they were one more trip to memory on the critical path
            const float slope = a.slopes ? a.slopes[hi] : 0.f;
            const float kmax_old = __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned int*>(a.kmax2 + kh), RLX_AGENT));
            // ---- phase 1: q | k | v = Wqkv . LN(x) ------------------------------------------------------------------------------------
            if (own1) {
                if (l > 0) gather(layers[l - 1].gxo, d, ebase + 8u * (unsigned)layers[l - 1].layer + 5u, xs, tid, err);
                else if (front) gather(ext->gxf, d, efront + 1u, xs, tid, err);
                __syncthreads();
                f32x4 xn[2];
                wave_norm(xs, d, a.gam1 != nullptr, n1, a.eps1, lane, xn);
                float y[2];
                dot_rows_x<2, 2>(wq, xn, d, lane, y);
                if (lane == 0) { outv[2 * w] = y[0]; outv[2 * w + 1] = y[1]; }
                publish16(a.gq, b * 16, N1, e0, outv, tid);
            }
            if (late) prefetch(a);
            STAMP(1);
            // ---- phase 2: split-key attention of (head hi, split sp) ----------------------------------------------------------------------
            __syncthreads();
            if (tid < 192) {   // q of the head, new key, new value: 3 x 64 granules
                const int part = tid >> 6;
                const int idx = (part == 0 ? hi * 64 : (part == 1 ? h * 64 + kh * 64 : h * 64 + kvh * 64 + kh * 64)) + lane;
                unsigned v = 0, spins = 0;
                for (;;) {
                    const unsigned long long x = __hip_atomic_load(as_global(a.gq + idx), RLX_AGENT);
                    v = (unsigned)x;
                    if ((unsigned)(x >> 32) == e0) break;
                    if (++spins > SPIN_LIMIT) { *err = 2; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
                qs[tid] = __uint_as_float(v);
            }
            __syncthreads();
            STAMP(3);
            {
                const float* knew = qs + 64;
                const float* vnew = qs + 128;
                float kn2 = knew[lane] * knew[lane], qn2 = qs[lane] * qs[lane];
                kn2 = wave_sum(kn2); qn2 = wave_sum(qn2);
                if (sp == 0 && w == 0) {
                    // (written through: a later note of the same launch reads the row from another XCD)
                    __hip_atomic_store(a.kcache + t * cw + kh * 64 + lane, knew[lane], RLX_AGENT);
                    __hip_atomic_store(a.vcache + t * cw + kh * 64 + lane, vnew[lane], RLX_AGENT);
                    if (lane == 0) atomicMax(reinterpret_cast<unsigned int*>(a.kmax2 + kh), __float_as_uint(kn2));
                }
                int j_lo = 0;
                if (slope > 0.f) {
                    const float km = fmaxf(kmax_old, kn2);
                    const float reach = (104.f + 2.f * a.scale * sqrtf(qn2 * km)) / slope;
                    if (reach < (float)t) j_lo = (t - (int)reach - 1) & ~255;   // as dec_attn2_kernel
                }
                if (sp == 0 && w == 0 && lane == 0 && a.jlo) __hip_atomic_store(a.jlo + hi, j_lo, RLX_AGENT);   // the next note's guess
                const int total = t + 1 - j_lo;
                const int chunk = (total + S - 1) / S;
                const int j0 = j_lo + sp * chunk, j1 = min(t + 1, j0 + chunk);
                const f32x4 q4 = *reinterpret_cast<const f32x4*>(qs + l16 * 4) * a.scale;
                float m = -INFINITY, lsum = 0.f;
                f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
                const bool hit = pre && j_lo == jlo_guess;   // the rows requested before q are the rows of this range (first 256 keys of it)
                // up to DEC_NU keys per lane group and batch (= 256 keys per workgroup: the whole split at L <= 4096): decode_attn.h
                bool first = true;
                const f32x4 knew4 = *reinterpret_cast<const f32x4*>(knew + l16 * 4), vnew4 = *reinterpret_cast<const f32x4*>(vnew + l16 * 4);
                for (int jb0 = j0 + w * 4 + grp; jb0 < j1; jb0 += 256) {
                    if (!(hit && first)) {
#pragma unroll
                        for (int u = 0; u < DEC_NU; ++u) {
                            const int j = min(min(jb0 + DEC_G * u, j1 - 1), t - 1);
                            k4[u] = *reinterpret_cast<const f32x4*>(a.kcache + (long)max(j, 0) * cw + kh * 64 + l16 * 4);
                            v4[u] = *reinterpret_cast<const f32x4*>(a.vcache + (long)max(j, 0) * cw + kh * 64 + l16 * 4);
                        }
                    }
                    first = false;
                    dec_attn_batch<DEC_NU>(k4, v4, knew4, vnew4, q4, slope, t, jb0, j1, m, lsum, acc);
                }
                STAMP(4);
                const int gi = w * 4 + grp;
                if (l16 == 0) { sm[gi] = m; sl[gi] = lsum; }
                *reinterpret_cast<f32x4*>(&so[gi][l16 * 4]) = acc;
                dec_attn_merge_wave(sm, sl, so, sm2, sl2, so2, w, lane);
            }
            __syncthreads();
            if (w < 2) {   // (the two waves that hold the 66 storing lanes, whole: the weights travel by readlane)
                float mm, num, den;
                dec_attn_merge_block(sm2, sl2, so2, min(tid, 63), lane, mm, num, den);
                if (tid < 66) {
                    // record (max, normaliser, 64 weighted value sums): lanes 0-63 of wave 0 store the sums, lanes 0-1 of wave 1 the two scalars
                    unsigned long long* mine = a.gp + ((long)hi * S + sp) * 66;
                    put(mine + (tid < 64 ? 2 + tid : tid - 64), e0 + 1, tid < 64 ? num : (tid == 64 ? mm : den));
                }
            }
            STAMP(2);
            if (l + 1 < n_layers) request(layers[l + 1]);
        }
        if (tail) {
            // ---- tail: e = Wh . LN(x) (the LM head's input projection: rows 16 b + 2 w, + 1), the normalised x mirrored by workgroup 0 ----
            const spn_dec_pair_args& al = layers[n_layers - 1];
            const int Nh = ext->Nh;
            f32x4 wh[2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                wh[i][0] = wh[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (r0 + i < Nh) load_row<2>(wh[i], ext->Wh + (long)(r0 + i) * ext->ld_h, d, lane);
            }
            const NormRegs4 nh = norm_regs4(d, ext->normh, ADA(ext->normh, ext->gamh), ext->beth, lane);
            if (b * 16 < Nh) {
                __syncthreads();
                gather(al.gxo, d, ebase + 8u * (unsigned)al.layer + 5u, xs, tid, err);
                __syncthreads();
                f32x4 xn[2];
                wave_norm(xs, d, ext->gamh != nullptr, nh, ext->epsh, lane, xn);
                if (b == 0 && w == 0 && ext->xn_out) {
#pragma unroll
                    for (int c = 0; c < 2; ++c)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int k = lane * 4 + c * 256 + e;
                            if (k < d) ext->xn_out[(long)t * ext->xn_ld + k] = xn[c][e];
                        }
                }
                float y[2];
                dot_rows_x<2, 2>(wh, xn, d, lane, y);
                if (lane == 0) {
                    if (r0 < Nh) ext->e_out[r0] = y[0];
                    if (r0 + 1 < Nh) ext->e_out[r0 + 1] = y[1];
                    outv[2 * w] = y[0]; outv[2 * w + 1] = y[1];
                }
                if (head) publish16(ext->ge, b * 16, Nh, ehead, outv, tid);
                STAMP_OF(al, 5);   // e out
            }
        }
        next_note();
        }   // notes
        return;
    }
    if (b >= nA + nB) {
        // ================================================ C: gated feed-forward rows =======================================================
        // 32 rows per workgroup (4 per wave: value + gate rows), so that only ceil(inner / 32) workgroups poll the x1 lines: with all h S
        // attention workgroups polling them this hop took 3.4-4.5 us, with 32-64 pollers ~1 us.  (64 rows per workgroup: 16 wave
        // reductions per wave, 2.5 us of issue time on the critical path.)
        const int bc = b - nA - nB;
        const int r0 = bc * 32 + 4 * w;
        f32x4 w1v[4][2], w1g[4][2];
        float b1v[4], b1g[4];
        NormRegs4 n2;
        auto request = [&](const spn_dec_pair_args& a) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                w1v[i][0] = w1v[i][1] = w1g[i][0] = w1g[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
                b1v[i] = b1g[i] = 0.f;
                if (r0 + i < inner) {
                    load_row<2>(w1v[i], a.W1 + (long)(r0 + i) * a.ld_1, d, lane);
                    load_row<2>(w1g[i], a.W1 + (long)(r0 + i + inner) * a.ld_1, d, lane);
                    if (a.b1) { b1v[i] = a.b1[r0 + i]; b1g[i] = a.b1[r0 + i + inner]; }
                }
            }
            n2 = norm_regs4(d, a.norm2, ADA(a.norm2, a.gam2), a.bet2, lane);
        };
        for (int note = 0; note < n_notes; ++note) {
        request(a0);
        if (err_in) return;
        if (ext && ext->rW) {
            // ---- the NEXT note's AdaLN (gamma | beta) rows: ry[n] = rW[n, :] . style row + rbias[n], the rider GEMV of dec_embed_proj_kernel
            //      (one wave per row, K <= 256: one chunk per lane), into the buffer of the next note's parity.  These workgroups idle until the
            //      first x1 arrives; the rows' first reader is the next launch ----
            const int rN = ext->rN, rK = ext->rK, nCw = ((int)gridDim.x - nA - nB) * NW;
            const float* x = ext->rx + (long)min(t + 2, ext->rx_rows - 1) * ext->rx_ld;
            float* yo = ext->ry + ((t & 1) ? 0 : ext->ada_par);
            const bool in = lane * 4 < rK;
            const f32x4 xv = in ? *reinterpret_cast<const f32x4*>(x + lane * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
            for (int n0 = bc * NW + w; n0 < rN; n0 += 8 * nCw) {
                f32x4 wv[8];
                float bz[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int n = min(n0 + u * nCw, rN - 1);
                    wv[u] = in ? *reinterpret_cast<const f32x4*>(ext->rW + (long)n * ext->r_ldw + lane * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
                    bz[u] = ext->rbias ? ext->rbias[n] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int n = n0 + u * nCw;
                    f32x2 a2 = f32x2{0.f, 0.f};
                    if (in) dec_dot4(a2, wv[u], xv);
                    float acc = wave_sum(dec_fold(a2));
                    if (lane == 0 && n < rN) __hip_atomic_store(yo + n, ext->rbias ? acc + bz[u] : acc, RLX_AGENT);   // read by other workgroups, maybe in this launch
                }
            }
        }
        for (int l = 0; l < n_layers; ++l) {
            const spn_dec_pair_args& a = layers[l];
            const unsigned e0 = ebase + 8u * (unsigned)a.layer;
            STAMP(0);
            STAMP(1);
            __syncthreads();
            gather(a.gx, d, e0 + 3, xs, tid, err);
            __syncthreads();
            STAMP(2);
            f32x4 xn[2];
            wave_norm(xs, d, a.gam2 != nullptr, n2, a.eps2, lane, xn);
            STAMP(3);
            float* outc = &so[0][0];   // 32 results of the workgroup
            float accv[4], accg[4];
            dot_rows_x<2, 4>(w1v, xn, d, lane, accv);
            dot_rows_x<2, 4>(w1g, xn, d, lane, accg);
            // lane i < 4 finishes row i (bias, activation, product): ONE evaluation of the activation per wave instead of four (erff is
            // ~60 instructions, and a wave pays for an instruction whether one lane needs it or all)
            float acc = accv[0], ag = accg[0], bv = b1v[0], bg = b1g[0];
#pragma unroll
            for (int i = 1; i < 4; ++i) {
                const bool me = lane == i;
                acc = me ? accv[i] : acc; ag = me ? accg[i] : ag; bv = me ? b1v[i] : bv; bg = me ? b1g[i] : bg;
            }
            if (a.b1) { acc += bv; ag += bg; }
            if (lane < 4) outc[4 * w + lane] = acc * pair_act(ag, a.act);
            STAMP(4);
            __syncthreads();
            if (tid < 32 && bc * 32 + tid < inner) put(a.gg + bc * 32 + tid, e0 + 4, outc[tid]);   // one store instruction: 2 whole lines
            STAMP(5);
            if (l + 1 < n_layers) request(layers[l + 1]);
        }
        if (head) {
            // ---- head: the arg-max LM head of dec_head_kernel (decode.hip) over e = Wh . LN(x), which the tail publishes as granules --------
            // Work items (key q, slab sl) with SL slabs per key go round the feed-forward workgroups; a slab's rows are v = 8 sl + w, + 8 SL, ...
            // The logit of a row is the expression of dec_head_kernel (fma chain over k = lane, lane + 64, ..., then the wave ladder), and
            // the arg-max with ties to the lower id is a total order, so how the rows are dealt out does not matter.
            const int n = ext->hn, D = ext->hD, nC = (int)gridDim.x - nA - nB;
            const int SL = min(16, max(1, nC / n)), items = n * SL;
            float* bvs = sm;                          // per-wave best value / id of the current item
            int* bis = reinterpret_cast<int*>(sl);
            // Sampling (spn_dec_chain_ext.stopk): the slabs publish every LOGIT of their rows instead of their maxima, and the key's first
            // workgroup ranks, filters, normalises and draws exactly as dec_head_kernel<true> does behind its last slab (same expressions,
            // same 256-thread partial sums, same counter hash of (seed, position, key)): the token of spn_dec_head_sample, bit for bit.
            const bool sampling = ext->stopk != nullptr;
            float* slog = &so[0][0];                  // this workgroup's logits of the current item: [wave][row of the wave]
            float rw[4][4];
            auto load_rows = [&](float (&dst)[4][4], const float* tab, int V, int W, int vb, int vstep) __attribute__((always_inline)) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float* row = tab + (long)min(vb + u * vstep, V - 1) * W;
#pragma unroll
                    for (int c = 0; c < 4; ++c) dst[u][c] = (lane + 64 * c < W) ? row[lane + 64 * c] : 0.f;
                }
            };
            // the first item's first rows and the norm's affine rows do not depend on e: requested before the poll
            const int it0 = bc;
            if (it0 < items) {
                const int q = it0 / SL, s0 = it0 - q * SL;
                if (ext->hwidth[q] <= 256 && s0 * 8 + w < ext->hV[q]) load_rows(rw, ext->htable[q], ext->hV[q], ext->hwidth[q], s0 * 8 + w, 8 * SL);
            }
            if (it0 < items) {
                __syncthreads();
                gather(ext->ge, D, ehead, xs, tid, err);
                __syncthreads();
                // LayerNorm statistics of e, the arithmetic of dec_head_kernel: per wave over the dot-product layout (common.h dec_ln_stats)
                float mu, rs;
                {
                    f32x4 xv[8];
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        const int k = lane * 4 + c * 256;
                        xv[c] = k < D ? *reinterpret_cast<const f32x4*>(xs + k) : f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                    dec_ln_stats<8>(xv, D, ext->heps, lane, mu, rs);
                }
                __syncthreads();   // every wave has read xs before the slices below are overwritten
                for (int it = it0; it < items; it += nC) {
                    const int q = it / SL, s0 = it - q * SL;
                    const int c0 = ext->hcol0[q], W = ext->hwidth[q], V = ext->hV[q];
                    const float* tab = ext->htable[q];
                    const int vstep = 8 * SL;
                    for (int k = tid; k < W; k += NT) xs[c0 + k] = (xs[c0 + k] - mu) * rs * ext->hgamma[c0 + k] + ext->hbeta[c0 + k];
                    __syncthreads();
                    float best = -INFINITY;
                    int idx = 0x7fffffff;
                    const int UM = (V + vstep - 1) / vstep;   // rows per wave of a slab (rows v = 8 s0 + w + vstep u, u < UM)
                    auto take = [&](int v, float acc) __attribute__((always_inline)) {
                        acc = wave_sum(acc);
                        if (v < 32 && ((ext->hban >> v) & 1u)) acc = -INFINITY;
                        if (sampling && lane == 0) slog[w * UM + (v - (s0 * 8 + w)) / vstep] = acc;
                        if (acc > best || (acc == best && v < idx)) { best = acc; idx = v; }
                    };
                    int vb = s0 * 8 + w;
                    if (W <= 256) {
                        if (it != it0 && vb < V) load_rows(rw, tab, V, W, vb, vstep);
                        for (; vb < V; vb += 4 * vstep) {
                            float rn[4][4];
                            const bool more = vb + 4 * vstep < V;
                            if (more) load_rows(rn, tab, V, W, vb + 4 * vstep, vstep);
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                const int v = vb + u * vstep;
                                if (v >= V) break;
                                float acc = 0.f;
#pragma unroll
                                for (int c = 0; c < 4; ++c)
                                    if (lane + 64 * c < W) acc = fmaf(rw[u][c], xs[c0 + lane + 64 * c], acc);
                                take(v, acc);
                            }
                            if (more) {
#pragma unroll
                                for (int u = 0; u < 4; ++u)
#pragma unroll
                                    for (int c = 0; c < 4; ++c) rw[u][c] = rn[u][c];
                            }
                        }
                    } else {
                        for (int v = vb; v < V; v += vstep) {
                            const float* row = tab + (long)v * W;
                            float acc = 0.f;
                            for (int k = lane; k < W; k += 64) acc = fmaf(row[k], xs[c0 + k], acc);
                            take(v, acc);
                        }
                    }
                    if (lane == 0) { bvs[w] = best; bis[w] = idx; }
                    __syncthreads();
                    if (sampling) {
                        // this slab's 8 UM logits as one contiguous run of the key's granules: entry (8 s0 + w) UM + u is row 8 s0 + w + vstep u
                        if (tid < 8 * UM) {
                            const int ww = tid / UM, u = tid - ww * UM;
                            const float lv = (s0 * 8 + ww + vstep * u < V) ? slog[tid] : -INFINITY;
                            put(ext->gl + q * 1024 + s0 * 8 * UM + tid, ehead + 1u, lv);
                        }
                        __syncthreads();   // slog is reused by this workgroup's next item
                        continue;
                    }
                    if (tid == 0) {
                        for (int r = 1; r < NW; ++r) if (bvs[r] > best || (bvs[r] == best && bis[r] < idx)) { best = bvs[r]; idx = bis[r]; }
                        outv[0] = best; outv[1] = __int_as_float(idx);
                    }
                    __syncthreads();
                    if (tid < 2) put(ext->gh + (q * 16 + s0) * 2 + tid, ehead + 1u, outv[tid]);   // one store instruction, one line
                }
                STAMP_OF(layers[n_layers - 1], 6);   // this workgroup's slab maxima out
                // the first slab's workgroup of a key picks the winner among the key's slabs and writes the token
                for (int it = it0; it < items; it += nC) {
                    const int q = it / SL, s0 = it - q * SL;
                    if (s0 != 0) continue;
                    // the cell of this key at the next position (MASK or a given token: nobody else writes it during this launch) is
                    // requested BEFORE the poll -- read behind the winner it was one more trip to memory at the end of every note
                    long* cell = ext->tokens + (long)(t + 1) * ext->tok_ld + ext->hdim[q];
                    const long cur = tid == 0 ? *cell : 0;
                    __syncthreads();
                    if (sampling) {
                        // ---- the key's logits from all its slabs, then dec_head_kernel<true>'s rank / filter / normalise / draw ----
                        const int V = ext->hV[q], vstep = 8 * SL, UM = (V + vstep - 1) / vstep;
                        gather(ext->gl + q * 1024, 8 * SL * UM, ehead + 1u, xs, tid, err);
                        __syncthreads();
                        float* lg = &so[0][0];          // [V] logits by id, [1024 + V] weights
                        for (int v = tid; v < V; v += NT) lg[v] = xs[((v % vstep) / 8 * 8 + (v & 7)) * UM + v / vstep];
                        __syncthreads();
                        const int keep_n = ext->stopk[q];
                        float mx = -INFINITY;
                        if (tid < 256) for (int v = tid; v < V; v += 256) mx = fmaxf(mx, lg[v]);
                        mx = wave_max(mx);
                        if (tid < 256 && lane == 0) red[w] = mx;
                        __syncthreads();
                        mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
                        __syncthreads();
                        float mine_sum = 0.f;
                        if (tid < 256) for (int v = tid; v < V; v += 256) {
                            const float lv = lg[v];
                            int rank = 0;
                            for (int u = 0; u < V; ++u) { const float lu = lg[u]; rank += (lu > lv) || (lu == lv && u < v); }
                            const float wgt = (rank < keep_n && lv > -INFINITY) ? __expf((lv - mx) * ext->sinv_temp) : 0.f;
                            lg[1024 + v] = wgt;
                            mine_sum += wgt;
                        }
                        (void)mine_sum;
                        __syncthreads();
                        if (w == 0) {
                            const unsigned r = dec_mix32(dec_mix32(*ext->sseed ^ 0x9e3779b9u * (unsigned)(t + 1)) + 0x85ebca6bu * (unsigned)(q + 1));
                            const int pick = dec_sample_pick(lg + 1024, V, (float)(r >> 8) * (1.f / 16777216.f), lane);   // decode_sample.h
                            if (tid == 0) {
                                const long chosen = cur == ext->mask_id ? (long)pick : cur;
                                if (cur == ext->mask_id) __hip_atomic_store(cell, chosen, RLX_AGENT);
                                if (ext->gt) __hip_atomic_store(as_global(ext->gt + q), ((unsigned long long)(eemb + 1u) << 32) | (unsigned)chosen, RLX_AGENT);
                            }
                        }
                        __syncthreads();   // lg / red are reused by this workgroup's next key
                        continue;
                    }
                    gather(ext->gh + q * 32, 2 * SL, ehead + 1u, xs, tid, err);
                    __syncthreads();
                    if (tid == 0) {
                        float best = 0.f;
                        int idx = 0;
                        for (int r = 0; r < SL; ++r) {
                            const float bvv = xs[2 * r];
                            const int bii = __float_as_int(xs[2 * r + 1]);
                            if (r == 0 || bvv > best || (bvv == best && bii < idx)) { best = bvv; idx = bii; }
                        }
                        const long chosen = cur == ext->mask_id ? (long)idx : cur;
                        if (cur == ext->mask_id) __hip_atomic_store(cell, chosen, RLX_AGENT);
                        // the cell's final value for the next note's embed phase (same launch: spn_dec_chain_ext.gt)
                        if (ext->gt) __hip_atomic_store(as_global(ext->gt + q), ((unsigned long long)(eemb + 1u) << 32) | (unsigned)chosen, RLX_AGENT);
                    }
                    STAMP_OF(layers[n_layers - 1], 7);   // token written
                }
            }
            if (note == n_notes - 1 && bc == 0 && tid == 0 && ext->pos_next) *ext->pos_next = t + 1;
        }
        next_note();
        }   // notes
        return;
    }
    // ==================================================== B: merge, output projections ====================================================
    const int bb = b - nA;
    const int r0 = bb * 16 + 2 * w;
    f32x4 wo[2][2], w2[2][8];
    float res0[2] = {0.f, 0.f}, b2v[2] = {0.f, 0.f};
    auto request = [&](const spn_dec_pair_args& a) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            wo[i][0] = wo[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (r0 + i < d) load_row<2>(wo[i], a.Wo + (long)(r0 + i) * a.ld_o, h * 64, lane);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int c = 0; c < 8; ++c) w2[i][c] = f32x4{0.f, 0.f, 0.f, 0.f};
            b2v[i] = 0.f;
            if (r0 + i < d) { load_row<8>(w2[i], a.W2 + (long)(r0 + i) * a.ld_2, inner, lane); if (a.b2) b2v[i] = a.b2[r0 + i]; }
        }
    };
    for (int note = 0; note < n_notes; ++note) {
    if (front) {
        // ---- front: x0 = Wm . xin + bm (K = Km <= 1024), then x = Wp . (LN?(x0) | ctx[t + 1] | style[t + 1]) + bp (K <= 2048) ----
        const int Km = ext->Km, cw_ = ext->ctx ? ext->ctx_w : 0, sw_ = ext->style ? ext->style_w : 0, Kc = d + cw_ + sw_;
        f32x4 wm[2][4], wp[2][8];
        float bmv[2] = {0.f, 0.f}, bpv[2] = {0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int c = 0; c < 4; ++c) wm[i][c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < 8; ++c) wp[i][c] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (r0 + i < d) {
                load_row<4>(wm[i], ext->Wm + (long)(r0 + i) * ext->ld_m, Km, lane);
                if (ext->bm) bmv[i] = ext->bm[r0 + i];
            }
        }
        if (!emb) for (int k = tid; k < Km; k += NT) xs[k] = ext->xin[k];
#pragma unroll
        for (int i = 0; i < 2; ++i)
            if (r0 + i < d) { load_row<8>(wp[i], ext->Wp + (long)(r0 + i) * ext->ld_p, Kc, lane); if (ext->bp) bpv[i] = ext->bp[r0 + i]; }
        // context / style rows of the position: independent of everything computed here (up to 4 + 4 entries per thread)
        float cx[4] = {0.f, 0.f, 0.f, 0.f}, sx[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k = tid + NT * q;
            if (k < cw_) cx[q] = ext->ctx[(long)(t + 1) * ext->ctx_ld + k];
            if (k < sw_) sx[q] = ext->style[(long)(t + 1) * ext->style_ld + k];
        }
        const NormRegs nc = norm_regs(d, 1, ext->cat_gamma, ext->cat_beta, tid);
        request(a0);
        if (err_in) return;
        if (emb) gather(ext->gin, Km, eemb, xs, tid, err);   // the embed phase of the attention workgroups
        if (n_layers > 1) STAMP_OF(layers[1], 5);   // front: embedded tokens gathered
        __syncthreads();
        float y[2];
        dot_rows<4, 2>(wm, xs, Km, lane, y);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (ext->bm) y[i] += bmv[i];
            if (lane == 0) outv[2 * w + i] = y[i];
            if (lane == 0 && r0 + i < d && ext->y2m) ext->y2m[(long)t * ext->y2m_ld + r0 + i] = y[i];
        }
        publish16(ext->gf, bb * 16, d, efront, outv, tid);
        if (n_layers > 1) STAMP_OF(layers[1], 7);   // front: x0 out
        __syncthreads();
        gather(ext->gf, d, efront, xs, tid, err);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k = tid + NT * q;
            if (k < cw_) xs[d + k] = cx[q];
            if (k < sw_) xs[d + cw_ + k] = sx[q];
        }
        __syncthreads();
        if (ext->cat_gamma) {   // dec_fused_gemv_kernel's cat prologue: statistics per wave (common.h dec_ln_stats), normalised in place
            f32x4 xv[2];
#pragma unroll
            for (int c = 0; c < 2; ++c) xv[c] = lane * 4 + c * 256 < d ? *reinterpret_cast<const f32x4*>(xs + lane * 4 + c * 256) : f32x4{0.f, 0.f, 0.f, 0.f};
            float mu, rs;
            dec_ln_stats<2>(xv, d, ext->cat_eps, lane, mu, rs);
            __syncthreads();
            if (tid < 256) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int k = tid + 256 * i;
                    if (k < d) xs[k] = (xs[k] - mu) * rs * nc.g[i] + nc.b[i];
                }
            }
            __syncthreads();
        }
        dot_rows<8, 2>(wp, xs, Kc, lane, y);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (ext->bp) y[i] += bpv[i];
            res0[i] = y[i];
            if (lane == 0) outv[2 * w + i] = y[i];
            if (lane == 0 && r0 + i < d && ext->y2p) ext->y2p[(long)t * ext->y2p_ld + r0 + i] = y[i];
        }
        publish16(ext->gxf, bb * 16, d, efront + 1u, outv, tid);
        STAMP_OF(a0, 7);   // the note's input vector out
    } else {
        request(a0);
#pragma unroll
        for (int i = 0; i < 2; ++i) if (r0 + i < d) res0[i] = a0.x[r0 + i];
        if (err_in) return;
    }
    for (int l = 0; l < n_layers; ++l) {
        const spn_dec_pair_args& a = layers[l];
        const unsigned e0 = ebase + 8u * (unsigned)a.layer;
        STAMP(0);
        // ---- phase 2b: the first h of these workgroups merge the S partials of head bb ---------------------------------------------------
        if (bb < h) {
            __syncthreads();
            gather(a.gp + (long)bb * S * 66, S * 66, e0 + 1, xs, tid, err);     // S <= 16: 1056 granules
            __syncthreads();
            if (tid < 64) {
                const int dcol = 2 + tid;
                float mv[16], lv[16], nv[16];
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const bool in = q < S;
                    mv[q] = in ? xs[q * 66] : -INFINITY;
                    lv[q] = in ? xs[q * 66 + 1] : 0.f;
                    nv[q] = in ? xs[q * 66 + dcol] : 0.f;
                }
                float mm = -INFINITY;
#pragma unroll
                for (int q = 0; q < 16; ++q) mm = fmaxf(mm, mv[q]);
                float num = 0.f, den = 0.f;
                // split q's factor: computed by lane q, broadcast through a scalar register (as in the attention workgroups' own merge)
                const int ql = lane & 15;
                const float mq = ql < S ? xs[ql * 66] : -INFINITY;
                const float fl = (mq == -INFINITY) ? 0.f : __expf(mq - mm);
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    if (q < S) {
                        const float f = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, fl), q));
                        num += nv[q] * f; den += lv[q] * f;
                    }
                }
                put(a.go + bb * 64 + tid, e0 + 2, num / den);
            }
        }
        STAMP(3);
        // ---- phase 3: x1 = x + Wo . o ------------------------------------------------------------------------------------------------------
        float x1[2];
        __syncthreads();
        gather(a.go, h * 64, e0 + 2, xs, tid, err);
        __syncthreads();
        dot_rows<2, 2>(wo, xs, h * 64, lane, x1);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            x1[i] = x1[i] + res0[i];
            if (lane == 0) outv[2 * w + i] = x1[i];
        }
        publish16(a.gx, bb * 16, d, e0 + 3, outv, tid);
        STAMP(4);
        // ---- phase 5: x = x1 + W2 . g + b2 ---------------------------------------------------------------------------------------------------
        __syncthreads();
        gather(a.gg, inner, e0 + 4, xs, tid, err);
        __syncthreads();
        float y5[2];
        dot_rows<8, 2>(w2, xs, inner, lane, y5);
        const bool last = l + 1 == n_layers;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float acc = y5[i];
            if (a.b2) acc += b2v[i];
            acc += x1[i];
            res0[i] = acc;                                   // this wave's rows of the stream: the next pair's residual
            if (lane == 0) outv[2 * w + i] = acc;
            if (lane == 0 && r0 + i < d) {
                if (last) a.x[r0 + i] = acc;
                if (a.y2) a.y2[(long)t * a.y2_ld + r0 + i] = acc;
            }
        }
        if (!last) {
            publish16(a.gxo, bb * 16, d, e0 + 5, outv, tid);
            STAMP(6);
            request(layers[l + 1]);
        } else {
            if (tail) publish16(a.gxo, bb * 16, d, e0 + 5, outv, tid);
            STAMP(6);
            // Everybody has read the tick by now: this workgroup's inputs needed every C workgroup's gated rows, those needed every B
            // workgroup's x1 rows, those every A workgroup's partials, and every workgroup reads the tick before its first phase.
            if (a.bump && note == n_notes - 1 && bb == 0 && tid == 0) *a.tick = *a.tick + n_notes;
        }
    }
    next_note();
    }   // notes
#undef a0
}

}  // namespace

// number of workgroups the launch uses for this shape (h S attention + ceil(d / 16) projection + ceil(inner / 32) feed-forward
// workgroups: all resident at once, one per CU), 0 = shape not supported
extern "C" int spn_dec_pair_groups(int d, int h, int kvh, int inner, int S) {
    if (!(d % 4 == 0 && d >= 64 && d <= 512 && h >= 1 && h * 64 <= 512 && inner % 4 == 0 && inner >= 16 && inner <= 2048 && S >= 1 && S <= 16 &&
          (kvh == 1 || kvh == h)))
        return 0;
    const int nA = h * S, nB = (d + 15) / 16, nC = (inner + 31) / 32, N1 = (h + 2 * kvh) * 64;
    if ((N1 + 15) / 16 > nA || h > nB || nA + nB + nC > 256) return 0;
    return nA + nB + nC;
}

extern "C" int spn_dec_struct_size(int which) { return which == 0 ? (int)sizeof(spn_dec_pair_args) : (which == 1 ? (int)sizeof(spn_dec_chain_ext) : -1); }

// `host`: the n argument records (validated here); `dev`: the same n records in DEVICE memory (the launch reads them there: a chain of
// pairs does not fit the kernel-argument segment).  The caller keeps both alive and identical; nothing is copied or allocated here.
// the kernel reads the norms' affine rows (and the adaptive (gamma | beta) rows) with 16-byte loads in the dot-product layout
static bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
static bool norm_rows_aligned(const spn_dec_pair_args& a) {
    return al16(a.gam1) && al16(a.bet1) && al16(a.gam2) && al16(a.bet2) && (a.d % 4) == 0;
}
// every workgroup of the launch polls results of the others: all of them must be resident at once, one per CU
static bool device_holds(int groups) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return false;
    return cus >= groups;
}

extern "C" int spn_dec_pairs(const spn_dec_pair_args* host, const spn_dec_pair_args* dev, int n, hipStream_t s) {
    SPN_REQUIRE(host && dev && n >= 1 && n <= 32, "spn_dec_pairs: 1 to 32 argument records, on the host and on the device");
    const spn_dec_pair_args& f = host[0];
    const int G = spn_dec_pair_groups(f.d, f.h, f.kvh, f.inner, f.S);
    SPN_REQUIRE(G > 0, "spn_dec_pairs: shape not supported (spn_dec_pair_groups)");
    SPN_REQUIRE(device_holds(G), "spn_dec_pairs: the device has fewer compute units than the launch has workgroups (all must be resident at once)");
    for (int l = 0; l < n; ++l) {
        const spn_dec_pair_args& a = host[l];
        SPN_REQUIRE(a.d == f.d && a.h == f.h && a.kvh == f.kvh && a.inner == f.inner && a.S == f.S, "spn_dec_pairs: the pairs of a chain must have one shape");
        SPN_REQUIRE(a.pos == f.pos && a.tick == f.tick && a.err == f.err, "spn_dec_pairs: one position, tick and error word per chain");
        SPN_REQUIRE(a.layer >= 0 && a.layer < 32 && (l == 0 || a.layer > host[l - 1].layer), "spn_dec_pairs: layer numbers must ascend below 32");
        SPN_REQUIRE(a.bump == 0 || l == n - 1, "spn_dec_pairs: only the last pair of a chain may advance the tick");
        SPN_REQUIRE(a.Wqkv && a.Wo && a.W1 && a.W2 && a.x && a.kcache && a.vcache && a.kmax2 && a.pos && a.tick && a.err && a.gq && a.gp && a.go &&
                    a.gx && a.gg && (a.gxo || l == n - 1), "spn_dec_pairs: null operand");
        SPN_REQUIRE(a.x == f.x, "spn_dec_pairs: one residual stream per chain");
        SPN_REQUIRE((a.ld_qkv % 4) == 0 && (a.ld_o % 4) == 0 && (a.ld_1 % 4) == 0 && (a.ld_2 % 4) == 0, "spn_dec_pairs: weight rows must be 16-byte aligned");
        SPN_REQUIRE(norm_rows_aligned(a), "spn_dec_pairs: the norms' affine rows must be 16-byte aligned");
    }
    hipLaunchKernelGGL(dec_pair_kernel, dim3(G), dim3(NT), 0, s, dev, n, (const spn_dec_chain_ext*)nullptr, f.pos, (const int*)f.tick, f.err, 1);
    SPN_LAUNCH_CHECK();
    return SPN_OK;
}

// The same with the note's input projections in front of the first pair and / or the LM head's input projection behind the last one
// (spn_dec_chain_ext: either part may be absent).  ext_host / ext_dev: the record on the host (validated) and in device memory.
static int dec_pairs_ext_notes(const spn_dec_pair_args* host, const spn_dec_pair_args* dev, int n, const spn_dec_chain_ext* ext_host,
                               const spn_dec_chain_ext* ext_dev, int notes, hipStream_t s) {
    SPN_REQUIRE(host && dev && ext_host && ext_dev && n >= 1 && n <= 31, "spn_dec_pairs_ext / _notes: 1 to 31 argument records and the extension record, on the host and on the device");
    const spn_dec_pair_args& f = host[0];
    const spn_dec_chain_ext& e = *ext_host;
    const int G = spn_dec_pair_groups(f.d, f.h, f.kvh, f.inner, f.S);
    SPN_REQUIRE(G > 0, "spn_dec_pairs_ext: shape not supported (spn_dec_pair_groups)");
    SPN_REQUIRE(device_holds(G), "spn_dec_pairs_ext: the device has fewer compute units than the launch has workgroups (all must be resident at once)");
    for (int l = 0; l < n; ++l) {
        const spn_dec_pair_args& a = host[l];
        SPN_REQUIRE(a.d == f.d && a.h == f.h && a.kvh == f.kvh && a.inner == f.inner && a.S == f.S, "spn_dec_pairs_ext: the pairs of a chain must have one shape");
        SPN_REQUIRE(a.pos == f.pos && a.tick == f.tick && a.err == f.err && a.x == f.x, "spn_dec_pairs_ext: one position, tick, error word and stream per chain");
        SPN_REQUIRE(a.layer >= 0 && a.layer < 31 && (l == 0 || a.layer > host[l - 1].layer), "spn_dec_pairs_ext: layer numbers must ascend below 31");
        SPN_REQUIRE(a.bump == 0 || l == n - 1, "spn_dec_pairs_ext: only the last pair of a chain may advance the tick");
        SPN_REQUIRE(a.Wqkv && a.Wo && a.W1 && a.W2 && a.x && a.kcache && a.vcache && a.kmax2 && a.gq && a.gp && a.go && a.gx && a.gg &&
                    (a.gxo || (l == n - 1 && !e.Wh)), "spn_dec_pairs_ext: null operand");
        SPN_REQUIRE((a.ld_qkv % 4) == 0 && (a.ld_o % 4) == 0 && (a.ld_1 % 4) == 0 && (a.ld_2 % 4) == 0, "spn_dec_pairs_ext: weight rows must be 16-byte aligned");
        SPN_REQUIRE(norm_rows_aligned(a), "spn_dec_pairs_ext: the norms' affine rows must be 16-byte aligned");
    }
    if (e.Wm) {
        const int Kc = f.d + (e.ctx ? e.ctx_w : 0) + (e.style ? e.style_w : 0);
        SPN_REQUIRE(e.Wp && e.xin && e.gf && e.gxf && e.Km >= 4 && e.Km <= 1024 && e.Km % 4 == 0 && Kc <= 2048 && Kc % 4 == 0 &&
                    (e.ld_m % 4) == 0 && (e.ld_p % 4) == 0 && (!e.ctx || e.ctx_w <= 2048) && (!e.style || e.style_w <= 2048),
                    "spn_dec_pairs_ext: front projections: Km <= 1024, d + context + style <= 2048, 16-byte aligned rows");
    }
    SPN_REQUIRE(al16(e.gamh) && al16(e.beth) && al16(e.egamma) && al16(e.ebeta) && (e.ada_par % 4) == 0,
                "spn_dec_pairs_ext: affine rows (tail norm, embedding norm) must be 16-byte aligned, ada_par a multiple of 4");
    if (e.Wh) SPN_REQUIRE(e.e_out && e.Nh >= 1 && e.Nh <= 16 * f.h * f.S && (e.ld_h % 4) == 0, "spn_dec_pairs_ext: tail projection: at most 16 h S rows");
    if (e.en) {
        SPN_REQUIRE(e.Wm && e.en >= 1 && e.en <= 16 && e.eD >= 4 && e.eD <= 2048 && e.eD % 4 == 0 && e.eN >= 1 && 2 * e.eN == e.Km && (e.eR == 1 || e.eR == 2) &&
                    e.eN % (8 * e.eR) == 0 && 2 * e.eN <= f.h * f.S * 8 * e.eR && e.tok_a && e.tok_b && e.We && (e.ld_e % 4) == 0 && e.gin &&
                    (!e.egamma || e.ebeta),
                    "spn_dec_pairs_ext: embed phase: needs the front with Km = 2 eN, eD <= 2048 in whole float4, eN in whole workgroups of 8 eR rows that the h S attention workgroups cover");
        int col = 0;
        for (int q = 0; q < e.en; ++q) {
            SPN_REQUIRE(e.etable[q] && e.ewidth[q] >= 1 && e.ecol0[q] == col, "spn_dec_pairs_ext: embed phase: key columns must be consecutive");
            col += e.ewidth[q];
        }
        SPN_REQUIRE(col == e.eD, "spn_dec_pairs_ext: embed phase: key widths must add up to eD");
    }
    if (e.rW) SPN_REQUIRE(e.rN >= 1 && e.rK >= 4 && e.rK <= 256 && e.rK % 4 == 0 && (e.r_ldw % 4) == 0 && e.rx && (e.rx_ld % 4) == 0 && e.rx_rows >= 1 && e.ry && e.ada_par >= e.rN,
                          "spn_dec_pairs_ext: rider: K <= 256 in whole float4, two row buffers ada_par >= rN floats apart");
    if (e.stopk) {
        SPN_REQUIRE(e.hn > 0 && e.gl && e.sseed && e.sinv_temp > 0.f, "spn_dec_pairs_ext: sampling needs the head phase, the logit granules gl and the seed");
        const int nC = G - f.h * f.S - (f.d + 15) / 16, SL = std::min(16, std::max(1, nC / e.hn));
        for (int q = 0; q < e.hn; ++q) {
            const int UM = (e.hV[q] + 8 * SL - 1) / (8 * SL);
            SPN_REQUIRE(e.hV[q] <= 1024 && 8 * UM <= 512 && 8 * SL * UM <= 1024,
                        "spn_dec_pairs_ext: sampling: vocabularies up to 1024 ids, at most 64 rows per wave of a slab");
        }
    }
    if (e.hn) {
        SPN_REQUIRE(e.Wh && e.hn >= 1 && e.hn <= 16 && e.hD == e.Nh && e.hD <= 2048 && e.hD % 4 == 0 && e.hgamma && e.hbeta && e.tokens && e.ge && e.gh,
                    "spn_dec_pairs_ext: head phase: needs the tail, 1 to 16 keys, e of a width in whole float4 <= 2048, norm, tokens and granule buffers");
        for (int q = 0; q < e.hn; ++q)
            SPN_REQUIRE(e.htable[q] && e.hV[q] >= 1 && e.hwidth[q] >= 1 && e.hcol0[q] >= 0 && e.hcol0[q] + e.hwidth[q] <= e.hD && e.hdim[q] >= 0,
                        "spn_dec_pairs_ext: head phase: bad key record");
    }
    SPN_REQUIRE(notes >= 1 && notes <= 64, "spn_dec_pairs_notes: 1 to 64 notes per launch");
    if (notes > 1)
        SPN_REQUIRE(e.Wm && e.Wh && e.hn > 0 && e.en > 0 && e.gt && e.pos_next && host[n - 1].bump,
                    "spn_dec_pairs_notes: several notes per launch need the whole note in the launch (embed, front, tail, head phases, the "
                    "token granules gt, pos_next, and the last pair advancing the tick)");
    hipLaunchKernelGGL(dec_pair_kernel, dim3(G), dim3(NT), 0, s, dev, n, ext_dev, f.pos, (const int*)f.tick, f.err, notes);
    SPN_LAUNCH_CHECK();
    return SPN_OK;
}

extern "C" int spn_dec_pairs_ext(const spn_dec_pair_args* host, const spn_dec_pair_args* dev, int n, const spn_dec_chain_ext* ext_host,
                                 const spn_dec_chain_ext* ext_dev, hipStream_t s) {
    return dec_pairs_ext_notes(host, dev, n, ext_host, ext_dev, 1, s);
}

// `notes` consecutive notes (positions *pos .. *pos + notes - 1) in ONE launch: the note's phases repeat inside the kernel, the chosen
// tokens reach the next note's embed phase as granules (ext.gt), *pos_next = *pos + notes and the tick advances by notes at the end.
// The caller guarantees that all those positions are to be decoded (the launch does not look at a length).
extern "C" int spn_dec_pairs_notes(const spn_dec_pair_args* host, const spn_dec_pair_args* dev, int n, const spn_dec_chain_ext* ext_host,
                                   const spn_dec_chain_ext* ext_dev, int notes, hipStream_t s) {
    return dec_pairs_ext_notes(host, dev, n, ext_host, ext_dev, notes, s);
}
