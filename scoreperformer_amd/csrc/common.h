// Common device/host helpers for the spn (ScorePerformer-native) HIP kernels.  gfx950 (CDNA4) only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

typedef uint16_t bf16_t;  // raw bfloat16 bits
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint16_t u16x4 __attribute__((ext_vector_type(4)));

#define SPN_OK 0
#define SPN_ERR_ARG -1
#define SPN_ERR_HIP -2

extern "C" void spn_set_error(const char* msg);

#define SPN_REQUIRE(cond, msg)                                                   \
    do {                                                                         \
        if (!(cond)) {                                                           \
            spn_set_error(msg);                                                  \
            return SPN_ERR_ARG;                                                  \
        }                                                                        \
    } while (0)

#define SPN_LAUNCH_CHECK()                                                       \
    do {                                                                         \
        hipError_t e__ = hipGetLastError();                                      \
        if (e__ != hipSuccess) {                                                 \
            spn_set_error(hipGetErrorString(e__));                               \
            return SPN_ERR_HIP;                                                  \
        }                                                                        \
    } while (0)

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// fp32 -> bf16 round-to-nearest-even through the native type: hipcc emits v_cvt_pk_bf16_f32 (one VALU op per PAIR);
// a hand-written integer RNE costs ~8 VALU ops and a divergent NaN branch per element.
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }

__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
    bf16x2_t v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(uint32_t, v);
}

// counter-based random bits for dropout masks (murmur3 finaliser): the same (seed, counter) gives the same bits in the
// forward and in every backward kernel, so masks are recomputed instead of stored
__device__ __forceinline__ uint32_t spn_hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    return x;
}

// Dropout bits of the feed-forward (nn.Dropout behind the activation, feedforward.py:57-60): element (row t, column c) is kept iff the
// 16-bit field (c & 1) of ffn_drop_bits(ffn_drop_rowc(t, seed), c >> 1) is >= thr16.  One definition for spn_act_fwd / spn_act_bwd and
// the gated GEMM epilogues (forward and backward must see the same mask).  The mixer uses full-rate 24-bit multiplies only
// (v_mul_u32_u24 / v_mad_u32_u24; a 32-bit v_mul_lo_u32 issues at quarter rate and the gated epilogues are VALU-bound): two rounds of
// fold + multiply-add and a final fold, the mixer of the attention dropout (attention_common.h: per-byte chi^2 ~ 1, avalanche ~0.5).
__device__ __forceinline__ uint32_t ffn_drop_rowc(long t, uint32_t seed) { return (uint32_t)t * 0x9E3779B1u + seed; }
__device__ __forceinline__ uint32_t ffn_drop_bits(uint32_t rowc, uint32_t pair) {
    uint32_t x = rowc + __umul24(pair, 0xEBCA77u);
    x ^= x >> 11; x = __umul24(x, 0xD35A2Du) + (x >> 8);
    x ^= x >> 13; x = __umul24(x, 0x9E3B35u) + (x >> 9);
    return x ^ (x >> 15);
}

// activations of the feed-forward (one definition: the fused GEMM epilogue and the stand-alone kernels must agree bit for bit)
// v_rcp_f32 (1 ulp) instead of the IEEE divide sequence: the GEMM epilogue that applies it is VALU-bound (-70 us per FFN at C3)
__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float silu_grad(float x) {
    const float s = __builtin_amdgcn_rcpf(1.f + __expf(-x));   // the sigmoid of silu_f: one exp + one rcp serve both in a gated backward
    return s * (1.f + x * (1.f - s));
}
__device__ __forceinline__ float gelu_grad(float x) {
    const float cdf = 0.5f * (1.f + erff(x * 0.70710678118654752f));
    const float pdf = 0.3989422804014327f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}

// 64-lane reductions through DPP (data-parallel primitives: a lane permutation folded into the VALU instruction), result in every lane.
// __shfl_xor compiles to ds_bpermute_b32 -- a round trip through the LDS crossbar per step, six dependent ones per reduction (~700
// cycles); the DPP ladder (two quad permutes, row_half_mirror, row_mirror, row_bcast:15, row_bcast:31, one v_readlane) is ~40.
// Same association order for every lane, so the result is wave-uniform by construction.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float spn_dpp(float old, float src) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += spn_dpp<0xB1, 0xf>(v, v);     // quad_perm [1,0,3,2]
    v += spn_dpp<0x4E, 0xf>(v, v);     // quad_perm [2,3,0,1]
    v += spn_dpp<0x141, 0xf>(v, v);    // row_half_mirror
    v += spn_dpp<0x140, 0xf>(v, v);    // row_mirror: every lane holds its 16-lane row's sum
    v += spn_dpp<0x142, 0xa>(0.f, v);  // row_bcast:15 into rows 1 and 3
    v += spn_dpp<0x143, 0xc>(0.f, v);  // row_bcast:31 into rows 2 and 3: lane 63 holds the total
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// Dot-product arithmetic of the decode kernels (round 5; decode.hip, decode_layer.hip, decode_attn.h -- every GEMV and score of both
// decode paths goes through these two, which is what keeps them bit-identical to each other): a lane accumulates TWO partial sums, the
// even and the odd elements of its k range, by packed FMA (v_pk_fma_f32: two fp32 FMAs per instruction at the issue cost of one --
// tools/pk_probe.hip), chunks ascending, and folds them lo + hi before the wave reduction.  A decode step is bound by the instruction
// count of its single-wave chains (4 multiplies + 4 adds per chunk became 2 instructions), not by bytes or flops.
__device__ __forceinline__ void dec_dot4(f32x2& a, const f32x4 w, const f32x4 x) {
    a = __builtin_elementwise_fma(f32x2{w[0], w[1]}, f32x2{x[0], x[1]}, a);
    a = __builtin_elementwise_fma(f32x2{w[2], w[3]}, f32x2{x[2], x[3]}, a);
}
__device__ __forceinline__ float dec_fold(const f32x2 a) { return a[0] + a[1]; }

// LayerNorm statistics of the decode kernels (round 5), by ONE wave over the vector in the dot-product layout (lane owns the chunks
// k = 4 lane + 256 c .. + 3, c < C; K a multiple of 4, entries past K ignored): two-pass, every lane folds its own entries first (packed
// adds / FMAs, chunks ascending), then ONE wave_sum per pass.  Every wave that needs the normalised vector computes the statistics itself:
// no barrier, no partial sums through LDS (the 256-thread form -- four wave partials per pass exchanged through LDS, three barriers --
// cost 0.76 us per call on the critical path of a note, twice per layer pair; recomputing those four partials in every wave 0.64).
template <int C>
__device__ __forceinline__ void dec_ln_stats(const f32x4 (&xv)[C], int K, float eps, int lane, float& mu, float& rs) {
    f32x2 s2 = f32x2{0.f, 0.f};
#pragma unroll
    for (int c = 0; c < C; ++c) {
        if (lane * 4 + c * 256 < K) { s2 += f32x2{xv[c][0], xv[c][1]}; s2 += f32x2{xv[c][2], xv[c][3]}; }
    }
    mu = wave_sum(dec_fold(s2)) / (float)K;
    f32x2 q2 = f32x2{0.f, 0.f};
#pragma unroll
    for (int c = 0; c < C; ++c) {
        if (lane * 4 + c * 256 < K) {
            const f32x2 ta = f32x2{xv[c][0], xv[c][1]} - mu, tb = f32x2{xv[c][2], xv[c][3]} - mu;
            q2 = __builtin_elementwise_fma(ta, ta, q2);
            q2 = __builtin_elementwise_fma(tb, tb, q2);
        }
    }
    rs = rsqrtf(wave_sum(dec_fold(q2)) / (float)K + eps);
}

// sum over each aligned group of 16 lanes (one DPP row), result in all 16 lanes
__device__ __forceinline__ float row16_sum(float v) {
    v += spn_dpp<0xB1, 0xf>(v, v);
    v += spn_dpp<0x4E, 0xf>(v, v);
    v += spn_dpp<0x141, 0xf>(v, v);
    return v + spn_dpp<0x140, 0xf>(v, v);
}

__device__ __forceinline__ float wave_max(float v) {
    const float ninf = -__builtin_inff();
    v = fmaxf(v, spn_dpp<0xB1, 0xf>(v, v));
    v = fmaxf(v, spn_dpp<0x4E, 0xf>(v, v));
    v = fmaxf(v, spn_dpp<0x141, 0xf>(v, v));
    v = fmaxf(v, spn_dpp<0x140, 0xf>(v, v));
    v = fmaxf(v, spn_dpp<0x142, 0xa>(ninf, v));
    v = fmaxf(v, spn_dpp<0x143, 0xc>(ninf, v));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// ---- LDS DMA through inline asm ----------------------------------------------------------------------------------------------
// `buffer_load_dwordx4 ... offen lds` copies 16 bytes per lane from a buffer resource straight into LDS (wave-uniform LDS base in M0
// + lane * 16).  Issued through the builtin (__builtin_amdgcn_raw_ptr_buffer_load_lds) hipcc tracks the copy as a pending LDS write on
// the VM counter and puts `s_waitcnt vmcnt(0)` in front of every later ds_read_b64_tr_b16 (and in front of LDS traffic of an epilogue):
// a software pipeline that keeps several tiles in flight is drained once per phase.  Issued through asm the compiler knows nothing
// about it: completion is counted by hand (`s_waitcnt vmcnt(N)` in asm, then a barrier, then the reads), which these kernels did
// anyway.  M0 is written in the statement that uses it.  The leading s_nop covers a resource / offset operand that the compiler has
// just moved into SGPRs with v_readlane / v_readfirstlane (VALU write of an SGPR -> VMEM read: 5 wait states; hipcc pads nothing
// inside or in front of an asm statement).
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x4 spn_buffer_rsrc(const void* base, uint32_t bytes) {
    const uint64_t a = (uint64_t)base;
    u32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((uint32_t)a);
    r[1] = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32) & 0xffffu);   // stride 0, no swizzle
    r[2] = __builtin_amdgcn_readfirstlane(bytes);                           // raw buffer: bounds check in bytes, out of range reads 0
    r[3] = 0x00020000u;
    return r;
}
__device__ __forceinline__ uint32_t spn_lds_addr(const void* p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}
// one 1 KiB piece: LDS [lds .. lds + 1023] <- lane l: 16 bytes at resource offset voff(l) + soff
__device__ __forceinline__ void spn_dma16(const u32x4 rs, uint32_t lds, uint32_t voff, uint32_t soff) {
    asm volatile("s_nop 4\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :: "s"(lds), "v"(voff), "s"(rs), "s"(soff) : "memory");
}
// the same without the leading s_nop 4: for call sites whose resource, LDS address and scalar offset all come from SALU instructions
// (no VALU-written SGPR in reach); M0 is still written one wait state in front of its use
__device__ __forceinline__ void spn_dma16_lean(const u32x4 rs, uint32_t lds, uint32_t voff, uint32_t soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :: "s"(lds), "v"(voff), "s"(rs), "s"(soff) : "memory");
}
// two pieces into consecutive KiB of LDS (the second piece's per-lane offset in voff1)
__device__ __forceinline__ void spn_dma16x2(const u32x4 rs, uint32_t lds, uint32_t voff0, uint32_t voff1, uint32_t soff) {
    asm volatile("s_nop 4\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, %4 offen lds\n\t"
                 "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds"
                 :: "s"(lds), "v"(voff0), "v"(voff1), "s"(rs), "s"(soff) : "memory", "scc");
}

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
