// Hop probe 2 (round 5): ONE-WAY latency of a hand-off as a function of how many workgroups poll the same lines, and of a two-level
// form -- one RELAY workgroup per XCD polls the global lines (agent scope) and re-publishes them with plain stores into a buffer of its
// XCD, which the other consumers of that XCD poll with workgroup-scope loads (they hit that XCD's L2: tools/hop_probe.hip, 0.2 us).
// NP producers publish 16 granules each ({epoch, low word of the 100 MHz clock}); a consumer's latency is its clock at the moment all
// tags match minus the newest stamp it read.  Consumers acknowledge through one counter; producers start the next round when all have.
// Block b is assumed to run on XCD b % 8 (checked: XCC_ID is reported).
//   build: hipcc --offload-arch=gfx950 -O3 -o tools/_bin/hop_probe2 tools/hop_probe2.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int TWO_LEVEL>
__global__ __launch_bounds__(512) void fan(unsigned long long* g, unsigned long long* loc, unsigned* ack, int NP, int NC, int iters, long long* out) {
    const int b = blockIdx.x, tid = threadIdx.x;
    const bool producer = b < NP;
    const int c = b - NP;                       // consumer number
    if (!producer && c >= NC) return;
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void*)g, 0, NP * 128, 0x00020000);
    const int xcd = b & 7;
    const bool relay = TWO_LEVEL && !producer && c < 8;     // consumers NP .. NP + 7 sit on eight different XCDs
    unsigned long long* mine = loc + (long)xcd * NP * 16;
    const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc((void*)mine, 0, NP * 128, 0x00020000);
    __shared__ unsigned lat_s;
    long long lat_sum = 0, lat_max = 0;
    for (int i = 1; i <= iters; ++i) {
        if (producer) {
            // wait for every consumer's acknowledgement of the previous round
            if (tid == 0) while (__hip_atomic_load(ack, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)((i - 1) * NC)) __builtin_amdgcn_s_sleep(2);
            __syncthreads();
            if (tid < 16) {
                const unsigned now = (unsigned)__builtin_amdgcn_s_memrealtime();
                __builtin_amdgcn_raw_buffer_store_b64(u32x2{now, (unsigned)i}, rg, b * 128 + tid * 8, 0, 16);
            }
            continue;
        }
        if (tid == 0) lat_s = 0;
        __syncthreads();
        const bool global_poll = !TWO_LEVEL || relay;
        unsigned newest = 0;
        for (unsigned spins = 0; spins < (1u << 22); ++spins) {
            bool ok = true;
            u32x4 v = u32x4{0, 0, 0, 0};
            if (tid < 8 * NP) {
                v = global_poll ? __builtin_amdgcn_raw_buffer_load_b128(rg, tid * 16, 0, (int)(16u | 0x80000000u))
                                : __builtin_amdgcn_raw_buffer_load_b128(rl, tid * 16, 0, (int)(1u | 0x80000000u));
                ok = v[1] == (unsigned)i && v[3] == (unsigned)i;
            }
            if (ok) { newest = v[0] > v[2] ? v[0] : v[2]; if (relay && tid < 8 * NP) __builtin_amdgcn_raw_buffer_store_b128(v, rl, tid * 16, 0, 0); break; }
            __builtin_amdgcn_s_sleep(1);
            asm volatile("" ::: "memory");
        }
        const unsigned now = (unsigned)__builtin_amdgcn_s_memrealtime();
        if (tid < 8 * NP) atomicMax(&lat_s, now - newest);
        __syncthreads();
        if (tid == 0) {
            if (i > 8) { lat_sum += lat_s; lat_max = lat_s > lat_max ? lat_s : lat_max; }
            atomicAdd(ack, 1u);
        }
    }
    if (!producer && tid == 0) { out[c * 2] = lat_sum; out[c * 2 + 1] = lat_max; if (c < 8) out[1024 + c] = __builtin_amdgcn_s_getreg((31 << 11) | 20); }
}

int main() {
    unsigned long long *g, *loc; unsigned* ack; long long* out;
    hipMalloc(&g, 1 << 16); hipMalloc(&loc, 1 << 20); hipMalloc(&ack, 64); hipMalloc(&out, 2048 * 8);
    const int iters = 600;
    for (int two = 0; two < 2; ++two)
        for (int NP : {8, 32})
            for (int NC : {1, 8, 32, 64, 128}) {
                if (two && NC < 16) continue;
                hipMemset(g, 0, 1 << 16); hipMemset(loc, 0, 1 << 20); hipMemset(ack, 0, 64); hipMemset(out, 0, 2048 * 8);
                if (two) hipLaunchKernelGGL((fan<1>), dim3(NP + NC), dim3(512), 0, 0, g, loc, ack, NP, NC, iters, out);
                else hipLaunchKernelGGL((fan<0>), dim3(NP + NC), dim3(512), 0, 0, g, loc, ack, NP, NC, iters, out);
                if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
                long long h[2048]; hipMemcpy(h, out, 2048 * 8, hipMemcpyDeviceToHost);
                double mean = 0, worst = 0, relay_mean = 0;
                for (int c = 0; c < NC; ++c) { mean += h[2 * c] * 10.0 / (iters - 8); worst = h[2 * c + 1] * 10.0 > worst ? h[2 * c + 1] * 10.0 : worst; }
                for (int c = 0; c < 8 && c < NC; ++c) relay_mean += h[2 * c] * 10.0 / (iters - 8);
                printf("%-10s %2d producers -> %3d consumers: mean %6.0f ns (first eight consumers %6.0f), worst %6.0f ns   XCC of consumers 0..7:", two ? "two-level" : "direct", NP, NC,
                       mean / NC, relay_mean / (NC < 8 ? NC : 8), worst);
                for (int c = 0; c < 8 && c < NC; ++c) printf(" %lld", h[1024 + c] & 15);
                printf("\n");
            }
    return 0;
}
