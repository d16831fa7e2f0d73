// Fused optimizer step over the flat parameter arena (HBM-bound, ~28 B/param):
//   global L2 grad norm -> clip coefficient -> AdamW -> refresh of the bf16 compute copy.
// Replaces `Optimizer.step` = clip_grad_norm_(2.0) + torch.optim.AdamW (experiments/optimizers.py:151-169,
// recipes/default.yaml:79-89) over 218 separate tensors with two launches over one contiguous buffer.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, long n, float* __restrict__ out) {
    __shared__ float red[4];
    float acc = 0.f;
    const long n4 = n / 4;
    const f32x4* g4 = reinterpret_cast<const f32x4*>(g);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const f32x4 v = g4[i];
        acc += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const float v = g[n4 * 4 + threadIdx.x]; acc += v * v; }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, red[0] + red[1] + red[2] + red[3]);
}

// The same sum with a FIXED order (round 5): data-parallel replicas clip with the norm of the all-reduced gradient, which is bit-identical on
// every rank -- but a sum of block partials by float atomicAdd arrives in a different order on every rank and every run, the clip
// coefficient then differs in its last bit and the replicas drift apart one ulp per step (found by the first two-process run,
// tests/test_dp_gpu.py).  Pass 1: block b writes its partial to ws[b] (same elements, same order every time); pass 2: ONE wave adds the
// partials in index order (lane l: b = l, l + 64, ...; then the DPP ladder of wave_sum, the same association for every launch).
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ g, long n, float* __restrict__ ws) {
    __shared__ float red[4];
    float acc = 0.f;
    const long n4 = n / 4;
    const f32x4* g4 = reinterpret_cast<const f32x4*>(g);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const f32x4 v = g4[i];
        acc += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const float v = g[n4 * 4 + threadIdx.x]; acc += v * v; }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) ws[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(64) void sumsq_final_kernel(const float* __restrict__ ws, int nb, float* __restrict__ out) {
    float acc = 0.f;
    for (int b = threadIdx.x; b < nb; b += 64) acc += ws[b];
    acc = wave_sum(acc);
    if (threadIdx.x == 0) out[0] += acc;
}

// normsq: device scalar (sum of squares of ALL grads, already reduced across whatever the caller wants);
// grad_scale multiplies g before clipping (1/world_size for data-parallel sums, 1/loss_scale ...).
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, bf16_t* __restrict__ shadow,
                                                    const uint8_t* __restrict__ slot_mask, long n,
                                                    const float* __restrict__ normsq, float max_norm, float grad_scale, float lr,
                                                    float beta1, float beta2, float eps, float wd, float bc1, float bc2_sqrt) {
    float coef = grad_scale;
    if (max_norm > 0.f) {
        const float total = sqrtf(*normsq) * grad_scale;
        coef *= fminf(1.f, max_norm / (total + 1e-6f));
    }
    const long n4 = n / 4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        // torch.optim.AdamW skips a parameter whose grad is None (frozen by Model.freeze, or never used): no decay, no moments
        if (slot_mask && slot_mask[i >> 1] == 0) continue;
        f32x4 pv = reinterpret_cast<f32x4*>(p)[i], mv = reinterpret_cast<f32x4*>(m)[i], vv = reinterpret_cast<f32x4*>(v)[i];
        const f32x4 gv = reinterpret_cast<const f32x4*>(g)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float gg = gv[e] * coef;
            float pe = pv[e] * (1.f - lr * wd);
            mv[e] = beta1 * mv[e] + (1.f - beta1) * gg;
            vv[e] = beta2 * vv[e] + (1.f - beta2) * gg * gg;
            const float denom = sqrtf(vv[e]) / bc2_sqrt + eps;
            pe -= (lr / bc1) * (mv[e] / denom);
            pv[e] = pe;
        }
        reinterpret_cast<f32x4*>(p)[i] = pv; reinterpret_cast<f32x4*>(m)[i] = mv; reinterpret_cast<f32x4*>(v)[i] = vv;
        if (shadow) {
            uint2 pk; pk.x = pack_bf2(pv[0], pv[1]); pk.y = pack_bf2(pv[2], pv[3]);
            reinterpret_cast<uint2*>(shadow)[i] = pk;
        }
    }
}

inline int grid_for(long total, int block = 256) { long g = (total + block - 1) / block; return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g)); }

}  // namespace

// out[0] += sum(g^2)
extern "C" int spn_sumsq(const float* g, long n, float* out, hipStream_t s) {
    SPN_REQUIRE(g && out && n > 0 && (((uintptr_t)g) & 15) == 0, "spn_sumsq: bad arguments (16-byte aligned)");
    hipLaunchKernelGGL(sumsq_kernel, dim3(grid_for(n / 4 + 1)), dim3(256), 0, s, g, n, out);
    SPN_LAUNCH_CHECK();
    return SPN_OK;
}

// out[0] += sum(g^2), the same bits for the same g on every launch and every device (see sumsq_partial_kernel).  ws: caller-owned scratch
// of spn_sumsq_det_ws_floats() floats.
extern "C" int spn_sumsq_det_ws_floats(void) { return 2048; }
extern "C" int spn_sumsq_det(const float* g, long n, float* out, float* ws, hipStream_t s) {
    SPN_REQUIRE(g && out && ws && n > 0 && (((uintptr_t)g) & 15) == 0, "spn_sumsq_det: bad arguments (16-byte aligned, workspace required)");
    const int nb = grid_for(n / 4 + 1);
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(nb), dim3(256), 0, s, g, n, ws);
    SPN_LAUNCH_CHECK();
    hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(64), 0, s, ws, nb, out);
    SPN_LAUNCH_CHECK();
    return SPN_OK;
}

// n must be a multiple of 4 (the arena pads); step >= 1.
// slot_mask: optional uint8 [n / 8], one flag per 8-element slot (the arena aligns every parameter to 8 elements): 0 = the slot belongs
// to a parameter torch.optim.AdamW would skip this step (grad is None: frozen or unused) and is left untouched; null = update all.
extern "C" int spn_adamw_step(float* p, const float* g, float* m, float* v, void* shadow_bf16, const uint8_t* slot_mask, long n,
                              const float* normsq,
                              float max_norm, float grad_scale, float lr, float beta1, float beta2, float eps, float weight_decay,
                              int step, hipStream_t s) {
    SPN_REQUIRE(p && g && m && v && n > 0 && n % 4 == 0 && step >= 1, "spn_adamw_step: bad arguments (n multiple of 4)");
    SPN_REQUIRE(!slot_mask || n % 8 == 0, "spn_adamw_step: a slot mask needs n to be a multiple of 8");
    SPN_REQUIRE(max_norm <= 0.f || normsq, "spn_adamw_step: normsq required when clipping");
    const float bc1 = 1.f - powf(beta1, (float)step);
    const float bc2_sqrt = sqrtf(1.f - powf(beta2, (float)step));
    hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(n / 4)), dim3(256), 0, s, p, g, m, v, (bf16_t*)shadow_bf16, slot_mask, n, normsq, max_norm,
                       grad_scale, lr, beta1, beta2, eps, weight_decay, bc1, bc2_sqrt);
    SPN_LAUNCH_CHECK();
    return SPN_OK;
}
