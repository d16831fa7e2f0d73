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

// activations of the feed-forward (one definition: the fused GEMM epilogue and the stand-alone kernels must agree bit for bit)
// v_rcp_f32 (1 ulp) instead of the IEEE divide sequence: the GEMM epilogue that applies it is VALU-bound (-70 us per FFN at C3)
__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
