// How much HBM bandwidth does the gated-backward epilogue's access PATTERN allow?  u [M, 2I] bf16 is read and du [M, 2I] bf16 written in
// pieces of `SEG` bytes per row (value half and gate half of a row are I * 2 bytes apart), ROWS rows per wave step, by waves that walk
// tiles in the order of the GEMM's tile walk (column groups of 8 tiles, rows down a group).  SEG = 128 is what a wave of the shipped
// kernel touches per row (64 bf16 outputs); larger SEG = what a cooperative (LDS-transposed) epilogue could do.
//   build: hipcc --offload-arch=gfx950 -O3 -o tools/_bin/seg_bw_probe tools/seg_bw_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// one workgroup = 256 threads; a "tile" = TR rows x TC bytes of the value half (+ the same of the gate half)
template <int SEG>   // bytes per row and wave step
__global__ __launch_bounds__(256) void probe(const char* __restrict__ u, char* __restrict__ du, int M, long row_bytes, int tile_rows, int tile_bytes) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int tiles_x = (int)(row_bytes / 2 / tile_bytes), tiles_y = M / tile_rows;
    constexpr int LPR = SEG / 16;            // lanes per row
    constexpr int RPS = 64 / LPR;            // rows per wave step
    for (int tile = blockIdx.x; tile < tiles_x * tiles_y; tile += gridDim.x) {
        const int G = 8, per = G * tiles_y, c = tile / per, within = tile - c * per;
        const int tx = c * G + within % G, ty = within / G;
        // the tile's (rows x bytes) region is cut into wave pieces of RPS rows x SEG bytes; 4 waves take pieces round-robin
        const int pieces_x = tile_bytes / SEG, pieces_y = tile_rows / RPS;
        for (int p = w; p < pieces_x * pieces_y; p += 4) {
            const int px = p % pieces_x, py = p / pieces_x;
            const long row = (long)ty * tile_rows + py * RPS + lane / LPR;
            const long col = (long)tx * tile_bytes + px * SEG + (lane % LPR) * 16;
            const u32x4 a = *reinterpret_cast<const u32x4*>(u + row * row_bytes + col);
            const u32x4 g = *reinterpret_cast<const u32x4*>(u + row * row_bytes + row_bytes / 2 + col);
            u32x4 x = a ^ g, y = a + g;
            *reinterpret_cast<u32x4*>(du + row * row_bytes + col) = x;
            *reinterpret_cast<u32x4*>(du + row * row_bytes + row_bytes / 2 + col) = y;
        }
    }
}

template <int SEG>
void run(const char* u, char* du, int M, long row_bytes, int tile_rows, int tile_bytes, int blocks) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<SEG>, dim3(blocks), dim3(256), 0, 0, u, du, M, row_bytes, tile_rows, tile_bytes);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(probe<SEG>, dim3(blocks), dim3(256), 0, 0, u, du, M, row_bytes, tile_rows, tile_bytes);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double bytes = 2.0 * M * row_bytes;
    printf("tile %4d rows x %5d B, wave piece %4d B per row, %5d workgroups: %7.1f us  %5.2f TB/s\n", tile_rows, tile_bytes, SEG, blocks, ms * 1e3, bytes / (ms * 1e-3) / 1e12);
}

int main() {
    const int M = 131072; const long row_bytes = 4096 * 2;   // u / du [M, 2 I] bf16, I = 2048
    char *u, *du; hipMalloc(&u, M * row_bytes); hipMalloc(&du, M * row_bytes);
    hipMemset(u, 1, M * row_bytes);
    for (int blocks : {512, 1024, 2048}) {
        run<128>(u, du, M, row_bytes, 256, 256, blocks);      // the duo kernel's tile (256 x 128 outputs), 128-byte pieces
        run<256>(u, du, M, row_bytes, 256, 256, blocks);
        run<128>(u, du, M, row_bytes, 256, 512, blocks);      // the ping-pong kernel's tile (256 x 256 outputs)
        run<512>(u, du, M, row_bytes, 256, 512, blocks);
        run<1024>(u, du, M, row_bytes, 128, 1024, blocks);
        run<1024>(u, du, M, row_bytes, 64, 4096, blocks);     // whole half rows
    }
    return 0;
}
