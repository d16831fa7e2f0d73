// Process-wide tuning knobs of libspn.so.  The library never reads the environment: the host binding sets knobs explicitly through
// spn_set_tuning() (scoreperformer_amd/lib.py maps SPN_* environment variables onto them once, at load).  Reads are relaxed atomic
// loads of plain doubles: a knob changed while kernels are being enqueued from another thread takes effect for later launches only.
#pragma once

enum SpnTune {
    SPN_TUNE_ATTN_BAND = 0,        // log2 of the smallest probability ratio still visited by the ALiBi band (0 = visit everything); 30
    SPN_TUNE_ATTN_ORDER,           // attention block order: 1 = XCD per batch element + heaviest-first causal tiles, 0 = grid order; 1
    SPN_TUNE_GEMM_VARIANT,         // 0 = measured dispatch; 1..6 force a 128x128 variant; 9 = ping-pong wherever eligible; 0
    SPN_TUNE_GEMM_NGROUP,          // n-tiles per column group of the tile order; 8
    SPN_TUNE_GEMM_SLICE_XCD,       // split-K: an XCD runs whole K slices; 1
    SPN_TUNE_GEMM_SPLIT_BLOCKS,    // split-K block target of the ping-pong kernel (0 = one full round of 256); 0
    SPN_TUNE_GEMM_PERSIST,         // plain GEMM, bf16 output, K <= 1024: 256 persistent blocks when the grid has at least this many rounds (0 = off); 2
    SPN_TUNE_GLU_PERSIST,          // gated GEMM: the same; 2
    SPN_TUNE_EMBED_STATS_BLOCKS,   // 2048
    SPN_TUNE_EMBED_SCATTER_MFMA,   // 1 = one-hot MFMA scatter, 0 = LDS-atomic scatter; 1
    SPN_TUNE_EMBED_SCATTER_BLOCKS, // 256
    SPN_TUNE_LN_BWD_BLOCKS,        // 2048
    SPN_TUNE_GEMM_DUO,             // 0 off; 1 = two 4-wave workgroups per CU (256x128 tiles) where measured faster; 2 = wherever eligible; 1
    SPN_TUNE_GEMM_DUO_NGROUP,      // n-tiles per column group of the duo kernel's tile order; 8
    SPN_TUNE_GEMM_STAGGER,         // persistent GEMM: start delay spread over the 32 CU slots of an XCD, in units of 1024 shader cycles (0 = off); 0
    SPN_TUNE_GLU_BWD_DUO,          // gated-backward GEMM (spn_gemm_glu_bwd): 2 = two 8-wave workgroups per CU (256x128 tiles, 64x64 wave tiles),
                                   // 1 = two 4-wave workgroups per CU, 0 = ping-pong kernel; 2
    SPN_TUNE_GEMM_PERSIST_BWD,     // 1 = the persistent walk also for input-gradient GEMMs (N-contiguous B).  Only safe when no other kernel holds CUs
                                   // during the backward (a concurrent all-reduce starves the blocks that land on its CUs): the host sets it; 0
    SPN_TUNE_GEMM_F32_MFMA,        // exact-fp32 GEMM with both operands contiguous along K: 1 = v_mfma_f32_32x32x2_f32 tiles, 0 = the VALU tile kernel; 1
    SPN_TUNE_COUNT
};

double spn_tune(SpnTune k);
static inline int spn_tune_i(SpnTune k) { return (int)spn_tune(k); }

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (call site, device); `mask` is the call site's static word.
// Two host threads may both find the bit clear and both set the attribute: harmless (idempotent).
#include <atomic>
#include <hip/hip_runtime.h>
static inline void spn_lds_optin(std::atomic<unsigned>& mask, const void* fn, int bytes) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned bit = 1u << (dev & 31);
    if (mask.load(std::memory_order_acquire) & bit) return;
    (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    mask.fetch_or(bit, std::memory_order_release);
}
