// Shared definitions of the attention kernels (see attention.hip for the design notes).
#pragma once
#include "common.h"
#include "tuning.h"

namespace spn_attn {

constexpr float NEG_FILL = -1.7014118e38f;  // -finfo(float32).max // 2   (attend.py:102)
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;
constexpr float RESCALE_THR = 8.f;          // log2 units

struct AttnArgs {
    const bf16_t* q; const bf16_t* k; const bf16_t* v;
    bf16_t* o; float* lse;
    const bf16_t* d_o; const float* delta;   // backward only
    bf16_t* dq; bf16_t* dk; bf16_t* dv; float* dslope;
    const uint8_t* kmask;   // [b, nk] or null
    const uint8_t* qmask;   // [b, nq] or null: rows with 0 are padding -- output rows of zeros, lse = NEG_FILL (dead: no gradient either way)
    const float* slopes;    // [h] or null
    int b, h, kvh, nq, nk, causal;
    long q_bs, q_ns, q_hs;
    long k_bs, k_ns, k_hs;
    long v_bs, v_ns, v_hs;
    long o_bs, o_ns, o_hs;     // o and d_o share strides
    long dq_bs, dq_ns, dq_hs;
    long dk_bs, dk_ns, dk_hs;
    long dv_bs, dv_ns, dv_hs;
    float scale;
    // attention dropout (attend.py:122 `dropout_p`): keep iff the 8-bit field of hash(seed, b, h, i>>1, j>>1) selected by
    // (i&1, j&1) is >= thr8;  kept probabilities are scaled by inv_keep = 1 / (1 - thr8/256).  thr8 == 0: no dropout.
    // The forward hashes and ALSO writes the keep bits (1 bit per score, dropbits_elems() uint16 words); both backward kernels
    // read them back instead of re-hashing (they are VALU-bound; a bit test is 3x cheaper per element than hash + compare).
    // Word [bh][query tile of 16][key tile of 64][forward lane]: bit 4*kb + r = key 16*kb + 4*(lane>>4) + r of query lane&15,
    // i.e. exactly the forward's (and dQ's) S^T register layout.
    uint32_t thr8, thr_frac, seed; int drop_on; float inv_keep;   // p = (thr8 + thr_frac / 65536) / 256: see set_dropout (attention.hip)
    uint16_t* dropbits; int nqt16, nkt64;
    // ALiBi band (attention.hip, "band skipping"): per 64-row query tile max ||q_i||^2 (+inf if the tile must never be skipped) and
    // min q_i.k_i' (the rows' own keys), then max ||k_j||^2 per (batch, kv head): layout at band_reach below.  null / band_log2 <= 0: off.
    const float* band; int nqt64; float band_log2;
    int order;   // block -> (batch, head, tile) remaps below (XCD locality, causal longest-first); SPN_ATTN_ORDER=0: plain grid order
};

// Largest |j - i - off| that can still matter for a query tile.  Log2-domain scores: t_ij = c1 q_i.k_j - slope2 |d|.  Every score obeys
// t_ij <= c1 |q_i| max|k| - slope2 |d|, and the row maximum is at least the score of the row's OWN key (d = 0): m_i >= c1 q_i.k_i'.
// So t_ij - m_i <= c1 (|q_i| max|k| - q_i.k_i') - slope2 |d|, and beyond D = (band_log2 + c1 (max|q| max|k| - min q_i.k_i')) / slope2
// every probability of the tile's rows is below 2^-band_log2 of its row's largest one.  (Until round 2 the row maximum was bounded by
// -c1 |q_i| max|k| instead of the actual diagonal score: 2 c1 max|q| max|k| in the numerator, a ~1.7x longer reach at initialisation.)
// band layout: [b*h*nqt64] max |q_i|^2 per 64-row tile (+inf: never skip), [b*h*nqt64] min q_i.k_i' per tile, [b*kvh] max |k_j|^2,
// [b][2] uint32 the live key tiles of the batch row (band_key_tiles below).
// A head whose slope is so shallow that even the numerator's floor (band_log2: Cauchy-Schwarz makes the rest non-negative) reaches past
// every distance of the problem can never skip a tile: the pre-pass does not compute its row norms (round 6: 2 of 8 heads at the
// initial slopes, a quarter of the pre-pass's pass over q) and band_reach answers "unbounded" without reading them -- the same test on
// both sides, so the three kernels and the pre-pass stay consistent.
__device__ __forceinline__ bool band_head_unbounded(const AttnArgs& a, float slope2) {
    return !(slope2 > 0.f) || a.band_log2 >= slope2 * (float)(a.nq > a.nk ? a.nq : a.nk);   // |j - i - (nk - nq)| < max(nq, nk)
}
__device__ __forceinline__ float band_reach(const AttnArgs& a, int bi, int hi, int kh, int qtile64, int ntiles, float c1, float slope2) {
    if (!a.band || band_head_unbounded(a, slope2)) return 3.0e38f;
    const long nq_part = (long)a.b * a.h * a.nqt64;
    const float* qt = a.band + ((long)(bi * a.h + hi)) * a.nqt64 + qtile64;
    float qm = qt[0], dm = qt[nq_part];
    for (int t = 1; t < ntiles; ++t)
        if (qtile64 + t < a.nqt64) { qm = fmaxf(qm, qt[t]); dm = fminf(dm, qt[nq_part + t]); }
    const float km = a.band[2 * nq_part + bi * a.kvh + kh];
    const float D = (a.band_log2 + c1 * (sqrtf(qm * km) - dm)) / slope2;
    return D < 1.0e9f ? fmaxf(D, 0.f) : 3.0e38f;   // +inf / nan (never-skip tiles) -> unbounded; -inf (padding rows only: qm = 0, dm = +inf) -> 0
}

// The live key tiles of batch row bi, as the band pre-pass left them ([2 bi] = 1 + the last 64-key tile with a live key, [2 bi + 1] =
// number of tiles - the first one; both zero: no mask, or no live key at all -- the walk stays unclamped, see key_scan_finish): two
// scalar loads next to band_reach's, instead of every wave scanning the mask row.
__device__ __forceinline__ void band_key_tiles(const AttnArgs& a, int bi, int& kt_lo, int& kt_hi) {
    const uint32_t* kr = reinterpret_cast<const uint32_t*>(a.band + 2 * (long)a.b * a.h * a.nqt64 + (long)a.b * a.kvh) + 2 * bi;
    const uint32_t hi = kr[0], lo = kr[1];
    if (hi) { kt_lo = max(kt_lo, (a.nk + 63) / 64 - (int)lo); kt_hi = min(kt_hi, (int)hi); }
}

// Key tiles (64 keys) outside [kt_lo, kt_hi) hold masked keys only -- the padding of a ragged batch, at either end -- and add
// exp(NEG_FILL - m) = 0 to every row that has a live key: the forward and dQ walks are clamped to the range.  Without a band buffer
// (no ALiBi: cross-attention) every wave scans the mask row itself (32 bytes per lane and 2048-key chunk, one ballot, scalar bit scans): no barrier, no LDS, and nothing added to the tile
// loop.  In two halves, so that the loads of the first chunk are in flight together with the prologue's other loads (q rows, band
// bounds) instead of in front of them.  Needs 16-byte aligned mask rows (nk % 16 == 0); otherwise, and for a row without a single live
// key (whose degenerate uniform average the full walk keeps, attend.py:102), the range stays [0, number of tiles).
struct KeyScan { uint4 x, y; bool on; };
__device__ __forceinline__ void key_scan_chunk(const uint8_t* mp, int nk, int at, uint4& x, uint4& y) {
    x = uint4{0u, 0u, 0u, 0u}; y = x;
    if (at < nk) x = *reinterpret_cast<const uint4*>(mp + at);
    if (at + 16 < nk) y = *reinterpret_cast<const uint4*>(mp + at + 16);
}
__device__ __forceinline__ KeyScan key_scan_begin(const uint8_t* mp, int nk, int lane) {
    KeyScan ks;
    ks.on = mp && ((reinterpret_cast<uintptr_t>(mp) | (uintptr_t)nk) & 15) == 0;
    ks.x = uint4{0u, 0u, 0u, 0u}; ks.y = ks.x;
    if (ks.on) key_scan_chunk(mp, nk, 32 * lane, ks.x, ks.y);
    return ks;
}
__device__ __forceinline__ void key_scan_finish(const KeyScan& ks, const uint8_t* mp, int nk, int lane, int& kt_lo, int& kt_hi) {
    if (!ks.on) return;
    int lo = 0x7fffffff, hi = -1;
    uint4 x = ks.x, y = ks.y;
    for (int base = 0; base < nk; base += 2048) {
        if (base) key_scan_chunk(mp, nk, base + 32 * lane, x, y);
        const unsigned long long segs = __ballot(((x.x | x.y | x.z | x.w) | (y.x | y.y | y.z | y.w)) != 0u);   // bit l: keys [32 l, 32 l + 32) of the chunk
        if (segs) {
            lo = min(lo, base / 64 + (__builtin_ctzll(segs) >> 1));
            hi = max(hi, base / 64 + ((63 - __builtin_clzll(segs)) >> 1));
        }
    }
    if (hi >= 0) { kt_lo = max(kt_lo, lo); kt_hi = min(kt_hi, hi + 1); }
}

// The kernel arguments again, straight from the kernel-argument segment (constant address space: scalar loads through the constant
// cache).  A kernel that takes its arguments from the by-value struct keeps every field it ever uses in an SGPR from the first
// instruction on -- ~90 of them here -- and hipcc parks what does not fit in VGPR lanes, to be fetched back with v_readlane (a VALU
// slot each, in VALU-bound loops).  Code that runs once (an epilogue) or once per tile reads this copy instead, behind an opaque
// `asm volatile("" : "+s"(ptr))` so that the loads are issued where they are used and not hoisted back to the top.
typedef const __attribute__((address_space(4))) AttnArgs* AttnKernargPtr;
__device__ __forceinline__ AttnKernargPtr attn_kernarg() { return (AttnKernargPtr)__builtin_amdgcn_kernarg_segment_ptr(); }

// geometry of the keep-bit buffer: whole 128-query blocks and 128-key blocks, so no kernel needs bounds checks
__host__ __device__ inline int dropbits_nqt16(int nq) { return 8 * ((nq + 127) / 128); }
__host__ __device__ inline int dropbits_nkt64(int nk) { return 2 * ((nk + 127) / 128); }

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// [64 rows][64 cols] bf16 tile (128-byte rows, 8 chunks of 16 B) ------------------------------------------
// "a" layout: chunk ^ (row & 7)           -> conflict-light ds_read_b128 of (row = lane&15, chunk = lane>>4)
// "t" layout: chunk ^ (((row>>1)&3) << 1) -> the 8 rows touched by two lane groups of a transpose read fall
//                                            into 8 distinct 32-byte windows
__device__ __forceinline__ int a_off(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }
__device__ __forceinline__ int t_off(int row, int chunk) { return row * 128 + ((chunk ^ (((row >> 1) & 3) << 1)) << 4); }

// A-operand fragment, rows = tile rows r_base + (lane&15), k = 32*ks + (lane>>4)*8 + e  (row-major "a" tile)
__device__ __forceinline__ bf16x8 frag_rows(const char* tile, int r_base, int ks, int lane) {
    return *reinterpret_cast<const bf16x8*>(tile + a_off(r_base + (lane & 15), ks * 4 + (lane >> 4)));
}

// A-operand fragment of the TRANSPOSED tile: rows = tile columns c_base + (lane&15); contraction index e of lane
// group g enumerates tile rows  32*u + 16*(e>>2) + 4*g + (e&3)   ("t" tile) -- the same enumeration that the
// C-layout registers of two stacked 16x16 blocks give when used as a B operand.
__device__ __forceinline__ bf16x8 frag_cols_t(const char* tile, int c_base, int u, int lane) {
    const int g = lane >> 4, p = lane & 15;
    const int col_byte = (c_base + 4 * (p & 3)) * 2;
    const int chunk = col_byte >> 4, within = col_byte & 15;
    const int r0 = 32 * u + 4 * g + (p >> 2);
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(tile + t_off(r0, chunk) + within));
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(tile + t_off(r0 + 16, chunk) + within));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

__device__ __forceinline__ bf16x8 pack8(const f32x4& a, const f32x4& b) {
    uint4 u;
    u.x = pack_bf2(a[0], a[1]); u.y = pack_bf2(a[2], a[3]);
    u.z = pack_bf2(b[0], b[1]); u.w = pack_bf2(b[2], b[3]);
    return __builtin_bit_cast(bf16x8, u);
}

// global -> registers for a [64][64] tile: 512 chunks of 16 B, 2 per thread (256 threads)
struct TileRegs {
    uint4 r[2];
    __device__ __forceinline__ void load(const bf16_t* base, long row_stride, int row0, int nrows, int tid) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = tid + 256 * i, row = c >> 3, ch = c & 7;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (row0 + row < nrows) v = *reinterpret_cast<const uint4*>(base + (long)(row0 + row) * row_stride + ch * 8);
            r[i] = v;
        }
    }
    template <bool T>
    __device__ __forceinline__ void store(char* tile, int tid) const {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = tid + 256 * i, row = c >> 3, ch = c & 7;
            *reinterpret_cast<uint4*>(tile + (T ? t_off(row, ch) : a_off(row, ch))) = r[i];
        }
    }
};

__device__ __forceinline__ bf16x8 load_row_frag(const bf16_t* base, long row_stride, int row, int nrows, int ks, int lane) {
    uint4 v = make_uint4(0, 0, 0, 0);
    if (row < nrows) v = *reinterpret_cast<const uint4*>(base + (long)row * row_stride + ks * 32 + (lane >> 4) * 8);
    return __builtin_bit_cast(bf16x8, v);
}

__device__ __forceinline__ float group_max(float v) {  // across the 4 lane groups (same lane&15)
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float group_sum(float v) {
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
}

// dropout bits of the 2x2 block (i>>1, j>>1): `row_const` = drop_row_const(...) of the row pair, `j_half` = j >> 1
__device__ __forceinline__ uint32_t drop_row_const(uint32_t seed, int bh, int nq_half, int i_half) {
    return ((uint32_t)(bh * nq_half + i_half)) * 0x9E3779B1u + seed;
}
// Mixer built from full-rate VALU ops only (v_mul_u32_u24 / v_mad_u32_u24; v_mul_lo_u32 is quarter rate): two rounds of
// fold + 24-bit multiply-add, final fold.  Checked offline on the (row_const, j_half) lattice: per-byte chi^2 ~ 1, keep-rate and
// field / row / column / diagonal correlations at sampling noise, avalanche 0.49-0.51.
__device__ __forceinline__ uint32_t drop_hash(uint32_t x) {
    x ^= x >> 11; x = __umul24(x, 0xD35A2Du) + (x >> 8);
    x ^= x >> 13; x = __umul24(x, 0x9E3B35u) + (x >> 9);
    return x ^ (x >> 15);
}
__device__ __forceinline__ uint32_t drop_bits(uint32_t row_const, int j_half) { return drop_hash(row_const + __umul24((uint32_t)j_half, 0xEBCA77u)); }
// Forward keep mask of a lane's 2 x 16 scores of one key tile (rows c and c + 16 of the wave's block, keys 16 kb + 4 g + r) as ONE
// 32-bit word: bit 16 qb + 4 kb + r.  Eight random words (one strong mix of the (row, key group) counter, seven light ones chained
// from it) are folded digit by digit along the binary expansion of thr8 / 256 -- AND for a 0 digit, OR for a 1 digit, least
// significant first -- which leaves every bit set with probability exactly thr8 / 256 (the drop mask).  ~2 VALU ops per score
// instead of a byte compare per score; the backward kernels read the bits the forward stored, so only this function defines the mask.
// scalar mixer for wave-uniform counters (plain 32-bit multiplies: stays on the SALU)
__device__ __forceinline__ uint32_t block_mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
// the seven cheaper words of a keep mask, chained from the strong one: w' = (w >> 8) * C + rot(w, 13) -- v_lshrrev + v_alignbit +
// v_mad_u32_u24, three full-rate VALU slots (round 4: add, shift, xor, shift, mad).  The 24-bit multiply sees the word's HIGH bits, the
// rotated addend puts its well-mixed middle bits under the product's weak low ones.  tools/dropmask_quality.py (numpy emulation on the
// kernel's counter lattice, 2 M words): drop rate, per-bit rates, every in-word bit pair, key / row / diagonal neighbours at sampling
// noise for thr8 = 1 .. 255, same as the five-slot chain; (w >> 8) * C + w, two slots, correlates bit pairs at 0.1 - 0.3 and is out.
__device__ __forceinline__ uint32_t drop_light(uint32_t w) { return __umul24(w >> 8, 0xD35A2Du) + __builtin_amdgcn_alignbit(w, w, 13); }
__device__ __forceinline__ uint32_t drop_keep32(uint32_t counter, uint32_t thr8) {
    uint32_t w = drop_hash(counter), acc = 0u;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        // digit k of thr8 (wave-uniform, as a 0 / ~0 mask d in an SGPR): 1 -> acc | w, 0 -> acc & w  =  majority(acc, w, d): ONE
        // v_bitop3_b32 (truth table 0xe8; the last digit writes the complement, 0x17, which is the keep mask).  Written as asm: from
        // the C expression hipcc makes and + or + v_cndmask (three slots), or two bit operations when the mask is laundered.
        const uint32_t d = 0u - ((thr8 >> k) & 1u);
        if (k < 7) asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0xe8" : "=v"(acc) : "v"(acc), "v"(w), "s"(d));
        else asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x17" : "=v"(acc) : "v"(acc), "v"(w), "s"(d));
        if (k < 7) w = drop_light(w);
    }
    return acc;
}
// Lanes l and l^1 hold neighbouring rows (or columns) of the same 2x2 blocks and need the same two hashes: each computes one
// and they swap through DPP (quad_perm [1,0,3,2]), halving the hash count.
__device__ __forceinline__ uint32_t lane_swap1(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false); }
__device__ __forceinline__ bool drop_keep(uint32_t bits, int i_odd, int j_odd, uint32_t thr8) {
    return ((bits >> (8 * (2 * i_odd + j_odd))) & 0xffu) >= thr8;
}

// Causal work per query tile grows linearly with its index, and consecutive blockIdx.x land on consecutive XCDs
// (block id % 8): map x -> tile so that XCD k owns tiles {k, k+8, ...} of the first half and their mirror images of the
// second half, i.e. every XCD gets the same total number of key tiles (a speed-only remap; any mapping is correct).
__device__ __forceinline__ int qtile_of(int x, int n, int causal) {
    if (!causal || n < 16 || (n & 15)) return x;
    const int half = n >> 1;
    return x < half ? x : (n - 1 - (x - half));
}

// Block -> (batch, head, tile) of the attention launches (grid = tiles x heads x batch).  Any bijection is correct; these are
// speed-only remaps (SPN_ATTN_ORDER=0 turns them off).  Workgroups go to the 8 XCDs round-robin (linear id % 8), each XCD has its
// own L2, and a free CU slot takes the next workgroup in linear order.
//  * xcd_batch_coords: XCD k owns the batch elements k, k+8, ... entirely.  Every block of a batch element streams the same K / V
//    (fwd, dQ; with MQA all heads share them) or the same Q / dO tiles (dK/dV): one L2 serves them all instead of eight
//    (bidirectional b=64 h=8 n=2048: fwd 0.86 -> 0.80 ms, bwd 2.50 -> 2.36 ms).
//  * causal_order: work per query tile grows with its index (per key block it shrinks).  With heavy and light tiles interleaved
//    the launch ends on a few late heavy blocks (fwd: ~8 %), and dK/dV -- 1024 blocks, 4 per CU -- gave XCD 0 2.4x the work of
//    XCD 7.  Tiles are issued heaviest first (longest-processing-time order): within an XCD's batch elements when the batch is a
//    multiple of 8, else XCD k takes tiles {k + 8m} and their mirror images {n-1-k-8m} (equal totals).  `heavy_high`: work grows
//    with the tile index (query tiles) or shrinks (key blocks).  Causal b=64 h=8 n=2048: fwd 0.66 -> 0.48 ms, bwd 2.00 -> 1.42 ms.
__device__ __forceinline__ bool xcd_batch_coords(const AttnArgs& a, int& bi, int& hi, int& tile) {
    const int nx = gridDim.x, ny = gridDim.y, nz = gridDim.z;
    if (!a.order || (nz & 7)) return false;
    const int L = blockIdx.x + nx * (blockIdx.y + ny * blockIdx.z), xcd = L & 7, idx = L >> 3;
    tile = idx % nx;
    const int rest = idx / nx;
    // heads from the LAST one down: ALiBi slopes fall with the head index, so the last heads have the widest band and the longest
    // blocks (32 key tiles against 3-5 for the steepest head at n = 2048); started last, they were the tail of every launch
    hi = ny - 1 - rest % ny;
    bi = (rest / ny) * 8 + xcd;
    return true;
}

__device__ __forceinline__ bool causal_order(const AttnArgs& a, bool heavy_high, int& bi, int& hi, int& tile) {
    const int nx = gridDim.x, ny = gridDim.y, nz = gridDim.z;
    if (!a.causal || !a.order) return false;
    const int L = blockIdx.x + nx * (blockIdx.y + ny * blockIdx.z), k = L & 7, idx = L >> 3;
    if ((nz & 7) == 0) {   // XCD k: its batch elements, tile by tile from the heaviest, heads fastest
        const int per = ny * (nz >> 3), seq = idx / per, rest = idx - seq * per;
        hi = rest % ny; bi = (rest / ny) * 8 + k;
        tile = heavy_high ? nx - 1 - seq : seq;
        return true;
    }
    if (nx & 15) return false;
    const int per_tile = ny * nz, seq = idx / per_tile, rest = idx - seq * per_tile, half = nx >> 4;
    hi = rest % ny; bi = rest / ny;
    const int light_first = seq < half ? k + 8 * seq : nx - 1 - k - 8 * (2 * half - 1 - seq);   // ascending tile index
    tile = heavy_high ? nx - 1 - light_first : light_first;
    return true;
}

// Tile classes for a wave's block of query rows [i_lo, i_hi] (in key coordinates, i + nk - nq) against keys [j0, j0+63]:
enum { T_GEN = 0, T_LEFT = 1, T_RIGHT = 2, T_SKIP = 3 };
__device__ __forceinline__ int classify(int j0, int i_lo, int i_hi, bool full, bool causal) {
    if (causal && j0 > i_hi) return T_SKIP;            // every key is in the future of every row of this wave
    if (!full) return T_GEN;
    if (j0 + 63 <= i_lo) return T_LEFT;                // all distances j - i <= 0: causal-clean, |d| = i - j
    if (j0 >= i_hi && !causal) return T_RIGHT;         // all distances >= 0: |d| = j - i
    return T_GEN;
}

// log2-domain scores of one (kb, qb) 16x16 block column for this lane.  Returns the value relative to the per-row offset u:
//   LEFT : t = s*c1 + slope2*(j - i)  = [s*c1 + slope2*j] + u,  u = -slope2*i
//   RIGHT: t = s*c1 - slope2*(j - i)  = [s*c1 - slope2*j] + u,  u = +slope2*i
//   GEN  : t = s*c1 - slope2*|j - i| (u = 0), masked entries -> NEG_FILL
template <int MODE>
__device__ __forceinline__ float score(float s, float c1, float slope2, float sj, float jf, float i_f, bool ok) {
    if (MODE == T_LEFT) return fmaf(s, c1, sj);
    if (MODE == T_RIGHT) return fmaf(s, c1, -sj);
    const float t = fmaf(-slope2, fabsf(jf - i_f), s * c1);
    return ok ? t : NEG_FILL;
}

}  // namespace spn_attn
