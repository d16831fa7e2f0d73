// Hop probe (round 5): what does ONE hand-off between two workgroups of a persistent launch cost, and does it get cheaper when producer
// and consumer sit on the same XCD and the consumer's poll is allowed to hit that XCD's L2?
// dec_pair_kernel (csrc/decode_layer.hip) hands every vector over as {epoch, value} granules: agent-scope store (sc1: written through
// to the fabric) + agent-scope poll loads (sc1: served past the L2).  Each XCD has its own L2; a store -- written through or not -- leaves
// its line in the producer's L2, so a consumer on the SAME XCD could poll with a workgroup-scope load (sc0: past the L1 only).
// Ping-pong between block 0 and block P (P = 8: same XCD under round-robin dispatch; P = 1: the neighbouring XCD), one granule each way;
// the blocks report their XCC_ID.  Modes: store / load cache bits (aux of the raw buffer intrinsics: 1 = sc0, 16 = sc1).
//   build: hipcc --offload-arch=gfx950 -O3 -o tools/_bin/hop_probe tools/hop_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int ST, int LD>
__global__ __launch_bounds__(64) void pingpong(unsigned long long* a2b, unsigned long long* b2a, int P, int iters, long long* out) {
    const int b = blockIdx.x;
    if (b != 0 && b != P) return;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)a2b, 0, 64, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)b2a, 0, 64, 0x00020000);
    const bool ping = b == 0;
    const __amdgpu_buffer_rsrc_t mine = ping ? ra : rb, theirs = ping ? rb : ra;
    long long t0 = 0;
    for (int i = 1; i <= iters + 16; ++i) {
        if (i == 17) t0 = __builtin_amdgcn_s_memrealtime();
        if (ping && threadIdx.x == 0) __builtin_amdgcn_raw_buffer_store_b64(u32x2{(unsigned)i, (unsigned)i}, mine, 0, 0, ST);
        for (unsigned spins = 0; spins < (1u << 22); ++spins) {
            const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(theirs, 0, 0, (int)((unsigned)LD | 0x80000000u));
            if (__builtin_amdgcn_readfirstlane(v[1]) == (unsigned)i) break;
            __builtin_amdgcn_s_sleep(1);
            asm volatile("" ::: "memory");
        }
        if (!ping && threadIdx.x == 0) __builtin_amdgcn_raw_buffer_store_b64(u32x2{(unsigned)i, (unsigned)i}, mine, 0, 0, ST);
    }
    const long long t1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        out[ping ? 0 : 2] = t1 - t0;
        out[ping ? 1 : 3] = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // XCC_ID
    }
}

// Fan-in / fan-out as in the decoder: NP producer blocks publish 16 granules each (one 128-byte line, one store instruction), NC consumer
// blocks (512 threads) poll all 16 NP granules with one 16-byte load per lane, then the roles swap (consumers publish, producers poll) --
// one iteration = two hops.  `local`: producers and consumers are chosen on ONE XCD (block ids = xcd + 8 k) and poll with LD bits; else
// spread over all XCDs.
template <int ST, int LD>
__global__ __launch_bounds__(512) void fan(unsigned long long* g0, unsigned long long* g1, int NP, int NC, int local, int iters, long long* out) {
    int b = blockIdx.x;
    int role = -1, idx = 0;   // 0 producer, 1 consumer
    if (local) {
        if ((b & 7) == 0) { const int k = b >> 3; if (k < NP) { role = 0; idx = k; } else if (k < NP + NC) { role = 1; idx = k - NP; } }
    } else {
        if (b < NP) { role = 0; idx = b; } else if (b < NP + NC) { role = 1; idx = b - NP; }
    }
    if (role < 0) return;
    const int tid = threadIdx.x;
    const int nmine = role == 0 ? NP : NC, ntheirs = role == 0 ? NC : NP;
    unsigned long long* mine = role == 0 ? g0 : g1;
    unsigned long long* theirs = role == 0 ? g1 : g0;
    const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc((void*)mine, 0, nmine * 128, 0x00020000);
    const __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc((void*)theirs, 0, ntheirs * 128, 0x00020000);
    long long t0 = 0;
    for (int i = 1; i <= iters + 16; ++i) {
        if (i == 17) t0 = __builtin_amdgcn_s_memrealtime();
        if (role == 0) {
            __syncthreads();
            if (tid < 16) __builtin_amdgcn_raw_buffer_store_b64(u32x2{(unsigned)tid, (unsigned)i}, rm, idx * 128 + tid * 8, 0, ST);
        }
        // poll all granules of the other side: pair p = tid (16 bytes), 8 ntheirs pairs in all (<= 512)
        for (unsigned spins = 0; spins < (1u << 22); ++spins) {
            bool ok = true;
            if (tid < 8 * ntheirs) {
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rt, tid * 16, 0, (int)((unsigned)LD | 0x80000000u));
                ok = v[1] == (unsigned)i && v[3] == (unsigned)i;
            }
            if (__syncthreads_and(ok)) break;
            __builtin_amdgcn_s_sleep(1);
            asm volatile("" ::: "memory");
        }
        if (role == 1) {
            __syncthreads();
            if (tid < 16) __builtin_amdgcn_raw_buffer_store_b64(u32x2{(unsigned)tid, (unsigned)i}, rm, idx * 128 + tid * 8, 0, ST);
        }
    }
    const long long t1 = __builtin_amdgcn_s_memrealtime();
    if (tid == 0 && idx == 0) {
        out[role == 0 ? 0 : 2] = t1 - t0;
        out[role == 0 ? 1 : 3] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    }
}

unsigned long long *g0, *g1;
long long* out;

template <int ST, int LD>
void run_pp(const char* what, int P) {
    const int iters = 2000;
    hipMemset(g0, 0, 1 << 16); hipMemset(g1, 0, 1 << 16); hipMemset(out, 0, 64);
    hipLaunchKernelGGL((pingpong<ST, LD>), dim3(16), dim3(64), 0, 0, g0, g1, P, iters, out);
    if (hipDeviceSynchronize() != hipSuccess) { printf("%s: launch failed\n", what); exit(1); }
    long long h[4];
    hipMemcpy(h, out, 32, hipMemcpyDeviceToHost);
    printf("ping-pong %-34s block 0 (XCC %lld) <-> block %d (XCC %lld): %7.0f ns per hop\n", what, h[1] & 15, P, h[3] & 15, h[0] * 10.0 / iters / 2);
}

template <int ST, int LD>
void run_fan(const char* what, int NP, int NC, int local) {
    const int iters = 1000;
    hipMemset(g0, 0, 1 << 16); hipMemset(g1, 0, 1 << 16); hipMemset(out, 0, 64);
    hipLaunchKernelGGL((fan<ST, LD>), dim3(256), dim3(512), 0, 0, g0, g1, NP, NC, local, iters, out);
    if (hipDeviceSynchronize() != hipSuccess) { printf("%s: launch failed\n", what); exit(1); }
    long long h[4];
    hipMemcpy(h, out, 32, hipMemcpyDeviceToHost);
    printf("fan %2d -> %2d -> %2d %-8s %-30s (XCC %lld / %lld): %7.0f ns per hop\n", NP, NC, NP, local ? "one XCD" : "spread", what, h[1] & 15, h[3] & 15,
           h[0] * 10.0 / iters / 2);
}

int main() {
    hipMalloc(&g0, 1 << 16); hipMalloc(&g1, 1 << 16); hipMalloc(&out, 64);
    for (int rep = 0; rep < 2; ++rep) {
        run_pp<16, 16>("store sc1, load sc1 (shipped)", 1);
        run_pp<16, 16>("store sc1, load sc1 (shipped)", 8);
        run_pp<16, 1>("store sc1, load sc0", 8);
        run_pp<1, 1>("store sc0, load sc0", 8);
        run_pp<0, 1>("store plain, load sc0", 8);
        run_pp<17, 17>("store sc0 sc1, load sc0 sc1", 1);
    }
    // the decoder's fan shapes: 16 splits -> 1 merge workgroup; 4 q-row workgroups -> 16 splits; 32 -> 32 as the d-row phases
    run_fan<16, 16>("sc1 / sc1 (shipped)", 16, 1, 0);
    run_fan<16, 16>("sc1 / sc1 (shipped)", 16, 1, 1);
    run_fan<16, 1>("store sc1, load sc0", 16, 1, 1);
    run_fan<0, 1>("store plain, load sc0", 16, 1, 1);
    run_fan<16, 16>("sc1 / sc1 (shipped)", 4, 16, 0);
    run_fan<16, 16>("sc1 / sc1 (shipped)", 4, 16, 1);
    run_fan<16, 1>("store sc1, load sc0", 4, 16, 1);
    run_fan<0, 1>("store plain, load sc0", 4, 16, 1);
    run_fan<16, 16>("sc1 / sc1 (shipped)", 32, 32, 0);
    return 0;
}
