// Latent stage of the hierarchical MMD-VAE heads (models/scoreperformer/mmd_transformer.py:232-283,505-542), the part BEHIND the head
// projections: what the reference does with ~60 small tensor ops per level -- boolean gathers, randperm subsets, the deadpan MSE,
// rand-and-gather dropout masks, concatenations -- as three kernels.  HBM-bound on a few MB; what matters is the launch count (the
// stage was ~240 of the ~420 ATen launches of a train step).
//
//   spn_latent_select    one level: a uniform random subset of at most K of the VALID latents (MMDLoss.forward, mmd:511-517:
//                        `latents[mask]`, then `randperm(N)[:max_num_latents]` when more than 4096 are left) packed into y [K, D] with 0/1
//                        row weights, plus the deadpan sums of the level (mmd:232-237,268-273).  No host read: the count of valid
//                        latents never leaves the device.
//   spn_latent_unselect  its backward: the gradient of the packed rows back to their latents, plus the deadpan term.
//   spn_latent_drop      all levels: latent dropout masks (mmd:249-253,351-354,537-542: one draw per latent, scattered to its notes,
//                        inclusive across levels) and the masked style embeddings in one pass over [b, n, sum L].
#include "common.h"

namespace {

constexpr int SEL_THREADS = 1024;

constexpr int SEL_MAX_N = 262144;          // the validity bitmask of a level lives in LDS (32 KB)
constexpr int SEL_MAX_K = 4096;            // so does the slot -> latent map (16 KB)

// random key of latent i: never 0 for a valid latent (0 = "not a candidate")
__device__ __forceinline__ uint32_t sel_key(uint32_t seed, uint32_t i, const uint32_t* bits) {
    return ((bits[i >> 5] >> (i & 31u)) & 1u) ? (spn_hash32(seed + i * 0x9E3779B1u) | 1u) : 0u;   // (one finaliser round: the key is recomputed in every pass)
}

// exclusive prefix sums of (a, b) over the 1024 threads of the block, in thread order; totals in ta / tb.  scratch: 2 x 16 ints.
__device__ __forceinline__ void block_excl_scan2(int a, int b, int& ea, int& eb, int& ta, int& tb, int* scratch) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int ia = a, ib = b;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int ua = __shfl_up(ia, o), ub = __shfl_up(ib, o);
        if (lane >= o) { ia += ua; ib += ub; }
    }
    if (lane == 63) { scratch[wv] = ia; scratch[16 + wv] = ib; }
    __syncthreads();
    int wa = 0, wb = 0;
    ta = tb = 0;
    for (int k = 0; k < SEL_THREADS / 64; ++k) {
        const int sa = scratch[k], sb = scratch[16 + k];
        if (k < wv) { wa += sa; wb += sb; }
        ta += sa; tb += sb;
    }
    ea = wa + ia - a; eb = wb + ib - b;
    __syncthreads();
}

// One workgroup.  lat [N, D] fp32, valid [N] (0/1), dead_b [B] (0/1, deadpan flag of the latent's batch element; N = B * S).
// Outputs: y [K, D], w [K] (1 for the first min(#valid, K) rows, else 0; unused rows of y are zero), slot [N] (row of y the latent went to, -1 =
// not selected), dead[3] = (sum of lat^2 over valid latents of deadpan elements, their count, 1 if any such square is non-zero).
// A single workgroup on purpose: the whole level is a few hundred KB, the work is a chain of block-wide scans, and the launch sits on
// the critical path of nothing wide -- what it replaces is ~25 dependent launches (rand, where, topk's sort passes, index_select ...).
__global__ __launch_bounds__(SEL_THREADS) void latent_select_kernel(const float* __restrict__ lat, const uint8_t* __restrict__ valid,
                                                                    const uint8_t* __restrict__ dead_b, int N, int S, int D, int K,
                                                                    uint32_t seed, float* __restrict__ y, float* __restrict__ w,
                                                                    int* __restrict__ slot, float* __restrict__ dead) {
    __shared__ uint32_t bits[SEL_MAX_N / 32 + 2];
    __shared__ int inv[SEL_MAX_K];
    __shared__ int hist[256];
    __shared__ int scratch[32];
    __shared__ uint32_t sh_prefix;
    __shared__ int sh_rem, sh_anydead;
    __shared__ float fred[3 * (SEL_THREADS / 64)];
    const int tid = threadIdx.x, lane = tid & 63;
    // ---- the validity bytes -> a bitmask in LDS.  16 bytes per lane and load, every load of a thread requested before the first is
    // packed (a byte per lane and trip was 68 dependent round trips to memory for the onset level: 60 of the kernel's 80 us)
    if (tid == 0) sh_anydead = 0;
    uint16_t* bits16 = reinterpret_cast<uint16_t*>(bits);
    const int nvec = N >> 4;                                  // whole 16-latent groups (valid is 16-byte aligned: host wrapper)
    for (int v0 = tid; v0 < nvec; v0 += 4 * SEL_THREADS) {
        uint4 u[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int v = v0 + k * SEL_THREADS;
            u[k] = v < nvec ? reinterpret_cast<const uint4*>(valid)[v] : uint4{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int v = v0 + k * SEL_THREADS;
            if (v >= nvec) continue;
            const uint32_t w4[4] = {u[k].x, u[k].y, u[k].z, u[k].w};
            uint32_t m = 0u;
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) m |= ((w4[q] >> (8 * e)) & 0xffu) ? 1u << (4 * q + e) : 0u;
            bits16[v] = (uint16_t)m;
        }
    }
    if (tid == 0) {                                          // the last, partial group
        uint32_t m = 0u;
        for (int i = nvec << 4; i < N; ++i) m |= valid[i] ? 1u << (i & 15) : 0u;
        bits16[nvec] = (uint16_t)m;
        if ((nvec & 1) == 0) bits16[nvec + 1] = 0;
    }
    __syncthreads();
    const int B = N / S;
    if (dead_b && tid < B && dead_b[tid]) sh_anydead = 1;     // (B <= 1024 checked by the host wrapper)
    // contiguous chunk of latents per thread: ranks in index order come out of one block scan
    const int per = (N + SEL_THREADS - 1) / SEL_THREADS;
    const int i0 = min(tid * per, N), i1 = min(i0 + per, N);
    int nv = 0;
    for (int i = i0; i < i1; ++i) nv += (bits[i >> 5] >> (i & 31)) & 1;
    int e_nv, e_unused, total_valid, t_unused;
    block_excl_scan2(nv, 0, e_nv, e_unused, total_valid, t_unused, scratch);

    // ---- deadpan sums (mmd_transformer.py:232-237,268-273): only when some batch element is flagged
    float dsq = 0.f, dcnt = 0.f, dany = 0.f;
    if (sh_anydead) {
        for (long e = tid; e < (long)N * D; e += SEL_THREADS) {
            const int i = (int)(e / D);
            if (((bits[i >> 5] >> (i & 31)) & 1) && dead_b[i / S]) {
                const float v = lat[e];
                dsq = fmaf(v, v, dsq);
                if (v * v != 0.f) dany = 1.f;
                if (e % D == 0) dcnt += 1.f;
            }
        }
    }
    dsq = wave_sum(dsq); dcnt = wave_sum(dcnt); dany = wave_max(dany);
    if (lane == 0) { fred[tid >> 6] = dsq; fred[16 + (tid >> 6)] = dcnt; fred[32 + (tid >> 6)] = dany; }
    __syncthreads();
    if (tid == 0) {
        float a = 0.f, b = 0.f, c = 0.f;
        for (int k = 0; k < SEL_THREADS / 64; ++k) { a += fred[k]; b += fred[16 + k]; c = fmaxf(c, fred[32 + k]); }
        dead[0] = a; dead[1] = b; dead[2] = c;
    }

    // ---- the K-th largest key by a 4-pass radix select (only when more than K latents are valid)
    uint32_t thr = 0u;       // selected: key > thr, and the first `rem` (index order) with key == thr
    int rem = 0;
    if (total_valid > K) {
        uint32_t prefix = 0u, pmask = 0u;
        rem = K;
        for (int shift = 24; shift >= 0; shift -= 8) {
            if (tid < 256) hist[tid] = 0;
            __syncthreads();
            for (int i = tid; i < N; i += SEL_THREADS) {
                const uint32_t k = sel_key(seed, (uint32_t)i, bits);
                if (k != 0u && (k & pmask) == prefix) atomicAdd(&hist[(k >> shift) & 255u], 1);
            }
            __syncthreads();
            // the bin that holds the rem-th largest key: suffix counts over the 256 bins by ONE wave (4 bins per lane, a shuffle scan
            // over the lanes) -- 16 block barriers per digit as a Hillis-Steele scan in LDS
            if (tid < 64) {
                const int h0 = hist[4 * tid], h1 = hist[4 * tid + 1], h2 = hist[4 * tid + 2], h3 = hist[4 * tid + 3];
                const int sl = h0 + h1 + h2 + h3;
                int suf = sl;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const int u = __shfl_down(suf, o);
                    if (tid + o < 64) suf += u;
                }
                int above = suf - sl;                          // keys in the bins of higher lanes
                const int hh[4] = {h0, h1, h2, h3};
#pragma unroll
                for (int q = 3; q >= 0; --q) {                 // bins 4 tid + 3 down to 4 tid
                    if (above < rem && above + hh[q] >= rem) { sh_prefix = prefix | ((uint32_t)(4 * tid + q) << shift); sh_rem = rem - above; }
                    above += hh[q];
                }
            }
            __syncthreads();
            prefix = sh_prefix; rem = sh_rem; pmask |= 255u << shift;
            __syncthreads();
        }
        thr = prefix;
    }

    // ---- ranks in index order -> slots
    int gt = 0, eq = 0;
    for (int i = i0; i < i1; ++i) {
        const uint32_t k = sel_key(seed, (uint32_t)i, bits);
        gt += (k > thr); eq += (k == thr && k != 0u);
    }
    int egt, eeq, tgt, teq;
    block_excl_scan2(gt, eq, egt, eeq, tgt, teq, scratch);
    const int G = tgt;                                   // keys above the threshold: slots [0, G); ties fill [G, G + rem)
    const int total = G + min(rem, teq);
    for (int i = i0; i < i1; ++i) {
        const uint32_t k = sel_key(seed, (uint32_t)i, bits);
        int s = -1;
        if (k > thr) s = egt++;
        else if (k == thr && k != 0u) { if (eeq < rem) s = G + eeq; ++eeq; }
        slot[i] = s;
        if (s >= 0) inv[s] = i;
    }
    __syncthreads();
    // w, the packed rows (coalesced over y), zero rows where no latent lands (a short level: fewer than K valid latents)
    for (int r = tid; r < K; r += SEL_THREADS) w[r] = r < total ? 1.f : 0.f;
#pragma unroll 4
    for (long e = tid; e < (long)K * D; e += SEL_THREADS) {
        const int r = (int)(e / D);
        y[e] = r < total ? lat[(long)inv[r] * D + (e % D)] : 0.f;
    }
}

// dlat[i, :] = (slot[i] >= 0 ? dy[slot[i], :] : 0) + coef_dead * lat[i, :] * (valid[i] & dead_b[i / S])
// g_dead (device scalar or null): dL/d(deadpan loss); the loss is sum / max(count * D, 1), so coef = 2 g / max(dead[1] * D, 1)
__global__ void latent_unselect_kernel(const float* __restrict__ dy, const int* __restrict__ slot, const float* __restrict__ lat,
                                       const uint8_t* __restrict__ valid, const uint8_t* __restrict__ dead_b, const float* __restrict__ dead,
                                       const float* __restrict__ g_dead, int N, int S, int D, float* __restrict__ dlat) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long)N * D) return;
    const int i = (int)(e / D), c = (int)(e % D);
    const int s = slot[i];
    float v = (s >= 0 && dy) ? dy[(long)s * D + c] : 0.f;
    if (g_dead && dead_b && valid[i] && dead_b[i / S]) v += 2.f * g_dead[0] / fmaxf(dead[1] * (float)D, 1.f) * lat[e];
    dlat[e] = v;
}

// (mmd, deadpan loss, deadpan flag) of a level from the MMD sums and the deadpan sums; weight = loss_weight of the MMD term (mmd:266)
__global__ void latent_scalars_kernel(const float* __restrict__ sums, float Z, const float* __restrict__ dead, float D, float weight,
                                      float* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const float n = fmaxf(sums[3], 1.f);
    out[0] = weight * (sums[0] / (Z * Z) + sums[1] / (n * n) - 2.f * sums[2] / (Z * n));
    out[1] = dead[0] / fmaxf(dead[1] * D, 1.f);
    out[2] = dead[2];
}

struct DropLevel {
    const long* seg;          // [b, n] segment id of every note, or null
    const uint8_t* lmask;     // [b, S] validity of the level's latents
    int S;                    // latents per batch element (1: sequence mean; n with seg == null: one latent per note)
    int col0, col1;           // the level's columns of the style embedding
    uint32_t thr24;           // drop a latent when (hash >> 8) < thr24  (0: the level draws nothing)
    const uint8_t* given;     // [b, S] explicit drop mask (test hook), or null
};
struct DropArgs {
    DropLevel lv[8];
    int nl, inclusive;
};

// One block = 64 notes of one sequence.  Phase 1: one thread per note decides, level by level, whether the note's latent is dropped (the
// dependent chain segment id -> latent mask -> hash runs ONCE per note, its result -- a bit per level -- goes to LDS); phase 2: all threads
// stream the block's [64, W] slab of embeddings, four columns each.  emb / out [b * n, W] fp32, drop [b * n, W] bytes (bool).
constexpr int DROP_ROWS = 64;
__global__ __launch_bounds__(256) void latent_drop_kernel(DropArgs a, const float* __restrict__ emb, const uint8_t* __restrict__ mask,
                                                          const uint8_t* __restrict__ deadpan, int n, int W, uint32_t seed,
                                                          float* __restrict__ out, uint8_t* __restrict__ drop) {
    __shared__ uint32_t hits[DROP_ROWS];
    const int bi = blockIdx.y, t0 = blockIdx.x * DROP_ROWS;
    const int rows = min(DROP_ROWS, n - t0);
    if (threadIdx.x < rows) {
        const int t = t0 + threadIdx.x;
        const long r = (long)bi * n + t;
        uint32_t bits = 0u;
        if (mask[r] && !(deadpan && deadpan[bi])) {
            bool prior = false;
            for (int l = 0; l < a.nl; ++l) {
                const DropLevel& L = a.lv[l];
                bool d = false;
                if (L.given || L.thr24) {
                    const long s = L.seg ? L.seg[r] : (L.S == 1 ? 0 : t);
                    if (s >= 0 && s < L.S) {
                        const long li = (long)bi * L.S + s;
                        if (L.given) d = L.given[li] != 0;
                        else d = L.lmask[li] && (spn_hash32(seed + (uint32_t)l * 0x632BE5ABu + spn_hash32((uint32_t)li * 0x9E3779B1u + 0x51ED27u)) >> 8) < L.thr24;
                    }
                }
                if (a.inclusive) { prior = prior || d; d = prior; }
                bits |= d ? 1u << l : 0u;
            }
        }
        hits[threadIdx.x] = bits;
    }
    __syncthreads();
    const int per = (W + 3) / 4;
    const long base = ((long)bi * n + t0) * W;
    for (int q = threadIdx.x; q < rows * per; q += blockDim.x) {
        const int e = q / per, c0 = (q - e * per) * 4;
        const uint32_t bits = hits[e];
        for (int c = c0; c < min(c0 + 4, W); ++c) {
            bool d = false;
            for (int l = 0; l < a.nl; ++l) if (c >= a.lv[l].col0 && c < a.lv[l].col1) d = (bits >> l) & 1u;
            const long o = base + (long)e * W + c;
            drop[o] = d ? 1 : 0;
            out[o] = d ? 0.f : emb[o];
        }
    }
}

__global__ void latent_drop_bwd_kernel(const float* __restrict__ g, const uint8_t* __restrict__ drop, long total, float* __restrict__ dx) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < total) dx[e] = drop[e] ? 0.f : g[e];
}

}  // namespace

// replaces: MMDLoss.forward's boolean gather + randperm subset (mmd_transformer.py:511-517) and the deadpan reduction (mmd:232-237)
extern "C" int spn_latent_select(const float* lat, const uint8_t* valid, const uint8_t* dead_b, int N, int S, int D, int K, unsigned seed,
                                 float* y, float* w, int* slot, float* dead, hipStream_t s) {
    SPN_REQUIRE(lat && valid && y && w && slot && dead && N > 0 && S > 0 && D > 0 && K > 0 && N % S == 0, "spn_latent_select: bad arguments");
    SPN_REQUIRE(N <= SEL_MAX_N && K <= SEL_MAX_K && N / S <= SEL_THREADS, "spn_latent_select: at most 262144 latents, 4096 selected, 1024 batch elements");
    SPN_REQUIRE(((uintptr_t)valid & 15) == 0, "spn_latent_select: the validity bytes must be 16-byte aligned");
    hipLaunchKernelGGL(latent_select_kernel, dim3(1), dim3(SEL_THREADS), 0, s, lat, valid, dead_b, N, S, D, K, (uint32_t)seed, y, w, slot, dead);
    SPN_LAUNCH_CHECK();
    return SPN_OK;
}

extern "C" int spn_latent_unselect(const float* dy, const int* slot, const float* lat, const uint8_t* valid, const uint8_t* dead_b,
                                   const float* dead, const float* g_dead, int N, int S, int D, float* dlat, hipStream_t s) {
    SPN_REQUIRE(slot && lat && valid && dlat && N > 0 && S > 0 && D > 0, "spn_latent_unselect: bad arguments");
    SPN_REQUIRE(!g_dead || (dead_b && dead), "spn_latent_unselect: a deadpan gradient needs the deadpan flags and sums");
    const long total = (long)N * D;
    hipLaunchKernelGGL(latent_unselect_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, dy, slot, lat, valid, dead_b, dead, g_dead, N, S, D, dlat);
    SPN_LAUNCH_CHECK();
    return SPN_OK;
}

extern "C" int spn_latent_scalars(const float* sums, int Z, const float* dead, int D, float weight, float* out, hipStream_t s) {
    SPN_REQUIRE(sums && dead && out && Z > 0 && D > 0, "spn_latent_scalars: bad arguments");
    hipLaunchKernelGGL(latent_scalars_kernel, dim3(1), dim3(64), 0, s, sums, (float)Z, dead, (float)D, weight, out);
    SPN_LAUNCH_CHECK();
    return SPN_OK;
}

// replaces: dropout_latent_mask + the per-level scatter / OR / concat / multiply of mmd_transformer.py:249-253,275-283,351-354,537-542.
// Per level (host arrays of nl entries): seg (int64 [b, n] or null), lmask (uint8 [b, S]), S, first column, p (drop probability; 0 = no
// draw), given (uint8 [b, S] explicit drop mask or null).
extern "C" int spn_latent_drop(int nl, const long* const* seg, const uint8_t* const* lmask, const int* S, const int* col0, const float* p,
                               const uint8_t* const* given, int inclusive, const float* emb, const uint8_t* mask, const uint8_t* deadpan,
                               int b, int n, int W, unsigned seed, float* out, uint8_t* drop, hipStream_t s) {
    SPN_REQUIRE(nl > 0 && nl <= 8 && seg && lmask && S && col0 && p && emb && mask && out && drop && b > 0 && n > 0 && W > 0,
                "spn_latent_drop: bad arguments (at most 8 levels)");
    DropArgs a;
    memset(&a, 0, sizeof(a));
    a.nl = nl; a.inclusive = inclusive;
    for (int l = 0; l < nl; ++l) {
        a.lv[l].seg = seg[l]; a.lv[l].lmask = lmask[l]; a.lv[l].S = S[l]; a.lv[l].col0 = col0[l];
        a.lv[l].col1 = l + 1 < nl ? col0[l + 1] : W;
        const float t = p[l] * 16777216.f;
        a.lv[l].thr24 = t <= 0.f ? 0u : (t >= 16777215.f ? 16777215u : (uint32_t)(t + 0.5f));
        a.lv[l].given = given ? given[l] : nullptr;
        SPN_REQUIRE(a.lv[l].lmask || a.lv[l].given || a.lv[l].thr24 == 0, "spn_latent_drop: a drawing level needs its latent mask");
    }
    hipLaunchKernelGGL(latent_drop_kernel, dim3(cdiv(n, DROP_ROWS), b), dim3(256), 0, s, a, emb, mask, deadpan, n, W, (uint32_t)seed, out, drop);
    SPN_LAUNCH_CHECK();
    return SPN_OK;
}

extern "C" int spn_latent_drop_bwd(const float* g, const uint8_t* drop, long total, float* dx, hipStream_t s) {
    SPN_REQUIRE(g && drop && dx && total > 0, "spn_latent_drop_bwd: bad arguments");
    hipLaunchKernelGGL(latent_drop_bwd_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, g, drop, total, dx);
    SPN_LAUNCH_CHECK();
    return SPN_OK;
}
