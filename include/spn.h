/* spn.h — C ABI of libspn.so: hand-written HIP kernels (gfx950 / MI355X) for the ScorePerformer transformer hot path.
 *
 * The reference (ilya16/ScorePerformer) is pure Python/PyTorch and has NO FFI of its own; each entry
 * point below names the reference call site(s) it replaces (file:line under scoreperformer/).  A maintainer binds them
 * with ctypes exactly as scoreperformer_amd/lib.py does.
 *
 * Conventions
 *  - plain pointers / sizes / strides; no torch types.  All device buffers -- inputs, outputs, saved-for-backward tensors AND
 *    workspaces -- are owned by the caller; the library never allocates, frees or retains device memory (no hipMalloc / hipFree
 *    anywhere in csrc/).  Ops that want scratch take `void* workspace, size_t workspace_bytes` and have a
 *    `spn_<op>_workspace_bytes(shape...)` query (GEMM split-K) or a documented element count (attention band, dropout bits, delta).
 *    Pointer-table arguments (`const float* const*`) are HOST arrays of device pointers, copied into the kernel argument block.
 *  - every call is asynchronous on `stream`; no device/stream synchronisation, no host reads: safe under hipGraph capture
 *    (tests/test_abi_gpu.py captures a GEMM + attention + LayerNorm sequence in a fresh process), and re-entrant across host
 *    threads: the only mutable process state is the tuning table below (atomics) and the per-kernel "LDS opt-in done" bit masks.
 *  - the library never reads the environment.  Tuning knobs are set with spn_set_tuning(name, value) (names: csrc/tuning.h;
 *    spn_tuning_count / spn_tuning_name enumerate them); the Python binding maps SPN_<NAME> variables onto it at load.
 *  - returns 0 on success, <0 on error (spn_last_error() gives a thread-local message).  No C++ exceptions cross the ABI.
 *  - dtype codes: 0 = fp32, 1 = bf16.  "bf16" buffers are raw 16-bit bfloat16.  Strides/leading dimensions in ELEMENTS.
 *  - ACCUMULATED outputs (documented per function) must be zeroed by the caller first.
 */
#ifndef SPN_H
#define SPN_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* spn_stream_t; /* hipStream_t */

int spn_abi_version(void); /* 10: round 6 added the qmask argument of spn_attn_fwd / spn_attn_bwd; 9: round 6 added spn_latent_{select,unselect,scalars,drop,drop_bwd}, spn_segment_{sum,gather}_multi, the row strides of spn_segment_sum / spn_segment_gather; 8: round 5 added spn_dec_pairs_notes + spn_dec_chain_ext.gt; 7: round 5 added spn_sumsq_det (+ the gemm_ow tuning knob); 6: round 4 added spn_comm_available, spn_mmd_scalars, spn_dropout, spn_layernorm_bwd_gb16_colsum, spn_dec_struct_size, the head / embed phases of spn_dec_chain_ext; 5: round 3 added spn_adaln_*, spn_dec_xattn_dyn, spn_dec_lookup, spn_dec_pair* */
const char* spn_last_error(void);
void spn_set_error(const char* msg);
int spn_set_tuning(const char* name, double value); /* 0, or -1 for an unknown knob */
int spn_get_tuning(const char* name, double* value);
int spn_tuning_count(void);
const char* spn_tuning_name(int i);

/* ---- GEMM (nn.Linear / F.linear everywhere on the path: modules/transformer/attention.py:135-142,210-218;
 *      feedforward.py:13-21,51-64; models/scoreperformer/embeddings.py:104,139,211,255,345-349; transformer.py:131,185;
 *      modules/layers.py:37,46) and their backward contractions.
 *      C[M,N] = residual + rowmask[m] * (alpha * A.B + bias[n]);  bf16 operands, fp32 accumulate.
 *      flags: bit0 A stored [K,M] (M contiguous); bit1 B stored [K,N] (N contiguous; default is nn.Linear's [N,K]);
 *             bit2 C fp32 (else bf16); bit3 C += (fp32 only). lda/ldb multiples of 8, 16-byte aligned bases.
 *      workspace: weight-gradient shapes (fp32 C, small M x N, K = all tokens) are split over K through `workspace`
 *      (spn_gemm_workspace_bytes(M, N, K, flags, batch) bytes, 16-byte aligned; 0 = this shape is never split); with a null or
 *      smaller workspace the product runs unsplit (same result up to fp32 summation order, slower). */
size_t spn_gemm_workspace_bytes(int M, int N, int K, int flags, int batch);
int spn_gemm_bf16(const void* A, const void* B, void* C, const float* bias, const float* residual, const uint8_t* rowmask,
                  int M, int N, int K, int lda, int ldb, int ldc, int ldr, float alpha, int flags, int batch, long strideA,
                  long strideB, long strideC, void* workspace, size_t workspace_bytes, spn_stream_t stream);
/* exact fp32 GEMM with arbitrary strides: A(m,k)=a[m*sam+k*sak], B(k,n)=b[k*sbk+n*sbn]  (VAE heads
 * models/scoreperformer/mmd_transformer.py:53-56; embedding value MLP modules/transformer/embeddings.py:202-213) */
int spn_gemm_f32(const float* a, long sam, long sak, const float* b, long sbk, long sbn, float* c, long ldc, const float* bias,
                 const uint8_t* rowmask, int M, int N, int K, float alpha, int accumulate, spn_stream_t stream);

/* ---- attention core (modules/transformer/attend.py:58-126 + mask/ALiBi assembly attention.py:162-197,
 *      modules/transformer/embeddings.py:294-315).  head dim 64; q [b,nq,h,64], k/v [b,nk,kvh,64] through strides;
 *      kvh = 1 is multi-query.  strides: {q_bs,q_ns,q_hs, k_.., v_.., o_..} (+ {dq_.., dk_.., dv_..} for bwd). */
int spn_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse, const uint8_t* kmask, const uint8_t* qmask,
                 const float* slopes, int b, int h, int kvh, int nq, int nk, int causal, float scale, const long* strides,
                 float p_drop, unsigned seed, void* dropbits, float* band, spn_stream_t stream);
/* qmask [b,nq] or null: the module's query-side `mask`.  attention.py:216-218 multiplies the block's output by it, so what the core
 * computes for a row with qmask = 0 is never seen: such a row's o is written as zeros and its lse as a dead marker (-1.7e38) that
 * makes spn_attn_bwd give it (and take from it) no gradient, and 128-row blocks / 64-row tiles made of such rows only are not
 * walked at all.  Key tiles with masked keys only are not walked either (they add exp(-1.7e38 - m) = 0 to every live row).
 * In a right-padded ragged batch that is the padding's share of the attention work.  A LIVE row without a single live key in its
 * visible range (causal rows in front of a front-padded sequence when no qmask is given, a context that is masked entirely) has no
 * defined attention: the reference averages V uniformly over the masked keys (attend.py:102 fills with a finite value); this core
 * returns that average when no live key exists in the batch row at all, and zeros for rows that merely lie outside the live key
 * range -- the module zeroes such rows either way. */
/* p_drop > 0: attention dropout (attend.py:122).  The mask is a pure function of (seed, b, h, i, j); the forward also writes
 * it as keep bits (1 bit per score) into `dropbits` (spn_attn_dropbits_elems() uint16 words), which the backward reads back.
 * delta: workspace b*h*nq floats; dslope [h] ACCUMULATED (may be null) */
long spn_attn_dropbits_elems(int b, int h, int nq, int nk);
/* ALiBi band skipping: key tiles whose probabilities are provably below 2^-log2_threshold of the row maximum (Cauchy-Schwarz
 * bound on q.k plus the linear distance penalty) are not visited, forward and backward alike.  Default 30 (the "attn_band" knob of
 * spn_set_tuning); 0 = visit all.  This is an APPROXIMATION, on by default whenever a band buffer is passed: at 30 every skipped
 * probability is < 2^-30 of its row's largest one, so the skipped mass of a row of 2048 keys is < 2^-19 of the softmax normaliser --
 * far below the 2^-9 rounding of the bf16 probabilities that enter P V, but above fp32 resolution (40 puts it below). */
void spn_attn_set_band(float log2_threshold);
/* `band` (spn_attn_band_elems floats): caller-owned buffer of the band bounds; spn_attn_fwd fills it and spn_attn_bwd of the same
 * q / k / mask reads it back.  null (or no slopes, or slopes <= 0, or a row whose own key is masked): every tile is visited. */
long spn_attn_band_elems(int b, int h, int kvh, int nq);
int spn_attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* d_o, const float* lse, float* delta,
                 void* dq, void* dk, void* dv, float* dslope, const uint8_t* kmask, const uint8_t* qmask, const float* slopes,
                 int b, int h, int kvh, int nq, int nk, int causal, float scale, const long* strides, float p_drop,
                 const void* dropbits, const float* band, spn_stream_t stream);

/* ---- LayerNorm / AdaptiveLayerNorm (modules/transformer/transformer.py:106,123-125,192-193,217;
 *      modules/layers.py:31-47).  gb = [T,2D] fp32 per-token (gamma|beta).  bwd: dy bf16; dgamma/dbeta ACCUMULATED. */
int spn_layernorm_fwd(const void* x, int x_dtype, long ldx, const float* gamma, const float* beta, const float* gb, long ldgb,
                      void* y, int y_dtype, long ldy, float* mean, float* rstd, int T, int D, float eps, spn_stream_t stream);
int spn_layernorm_bwd(const void* x, int x_dtype, long ldx, const void* dy, long lddy, const float* gamma, const float* gb,
                      long ldgb, const float* mean, const float* rstd, const float* dres, long lddres, void* dx, int dx_dtype,
                      long lddx, void* dx16 /* optional bf16 copy of dx */, long lddx16, float* dgamma, float* dbeta, void* dgb,
                      long lddgb, int T, int D, spn_stream_t stream);
/* adaptive LayerNorm with the per-token (gamma | beta) rows in bf16 (modules/layers.py:41-47: gamma, beta = Linear(condition)) */
int spn_layernorm_fwd_gb16(const void* x, int x_dtype, long ldx, const void* gb16, long ldgb, void* y, int y_dtype, long ldy,
                           float* mean, float* rstd, int T, int D, float eps, spn_stream_t stream);
int spn_layernorm_bwd_gb16(const void* x, int x_dtype, long ldx, const void* dy, long lddy, const void* gb16, long ldgb,
                           const float* mean, const float* rstd, const float* dres, long lddres, void* dx, int dx_dtype, long lddx,
                           void* dx16, long lddx16, void* dgb, long lddgb, int T, int D, spn_stream_t stream);
/* the same, and the column sums of the dgb rows (= the bias gradient of the condition Linear, modules/layers.py:38) are ACCUMULATED into
 * dgb_colsum [2D] fp32 by the same pass, from the unrounded products: no second read of dgb */
int spn_layernorm_bwd_gb16_colsum(const void* x, int x_dtype, long ldx, const void* dy, long lddy, const void* gb16, long ldgb,
                                  const float* mean, const float* rstd, const float* dres, long lddres, void* dx, int dx_dtype, long lddx,
                                  void* dx16, long lddx16, void* dgb, long lddgb, float* dgb_colsum /* [2D] ACCUMULATED */, int T, int D,
                                  spn_stream_t stream);

/* AdaptiveLayerNorm FORWARD with the condition projection INSIDE the kernel (modules/layers.py:31-47: gamma | beta = Linear(condition),
 * never materialised as [T, 2D] rows by a K = 64 GEMM).  spn_adaln_ok: 1 when the fused kernel takes the shape (D = 512, C = 64);
 * otherwise project with spn_gemm_bf16 and use spn_layernorm_fwd_gb16.  x fp32 [T, D]; cond bf16 [T, C]; W bf16 [2D, C] contiguous
 * (gamma rows, then beta rows); bias fp32 [2D]; y bf16 [T, D]; mean / rstd [T]; gamma_out: optional bf16 [T, D] (ldg), the gamma rows
 * the backward needs -- spn_layernorm_bwd_gb16(gb16 = gamma_out, ldgb = ldg) reads only the gamma half of a (gamma | beta) row. */
int spn_adaln_ok(int D, int C);
int spn_adaln_fwd(const float* x, long ldx, const void* cond, long ldc, const void* W, const float* bias, void* y, long ldy,
                  void* gamma_out, long ldg, float* mean, float* rstd, int T, int D, int C, float eps, spn_stream_t s);

/* ---- gated feed-forward input projection, activation fused into the GEMM epilogue (feedforward.py:13-21 GLU.forward and the
 * nn.Dropout of feedforward.py:57-60):  u[M,2I] = x W^T + bias (bf16, value | gate, kept for the backward);
 * g[M,I] = dropout(u[:, :I] * act(u[:, I:])) with the mask of spn_act_fwd(seed): spn_act_bwd(u, dg, ...) is its backward.
 * g equals spn_act_fwd(u) bit for bit.  spn_gemm_glu_ok: 1 when the shape is taken (M >= 128, I % 128 == 0, K % 64 == 0, K >= 256) */
int spn_gemm_glu_ok(int M, int I, int K);
int spn_gemm_glu(const void* x, const void* W /* [2I, K] */, void* u, void* g, const float* bias /* [2I] or null */, int M, int I, int K,
                 int lda, int ldb, int ldu, int ldg, int act /* 0 SiLU, 1 GELU */, float p_drop, unsigned seed, spn_stream_t s);
/* backward through the same block (feedforward.py:57-64 output projection, then :13-21): the input gradient of the OUTPUT projection
 * with the activation backward in the GEMM epilogue:  dg = dy[M,K] . W2[K,I] is never stored;
 * du[M,2I] = (d * act(gate) | d * value * act'(gate)),  d = dropout_mask(seed)(bf16(dg)),  value | gate = u.
 * Equals spn_gemm_bf16 (flags bit1) + spn_act_bwd bit for bit.  colsum_partial: null, or fp32 [ceil(M/128), 2I]: row r receives the
 * column sums of du rows 128r.. (bias gradient of the input projection = the sum of the rows).
 * spn_gemm_glu_bwd_ok: 1 when the shape is taken (M >= 128, M % 8 == 0, I % 256 == 0, K % 64 == 0, K >= 256) */
int spn_gemm_glu_bwd_ok(int M, int I, int K);
int spn_gemm_glu_bwd(const void* dy, const void* W2 /* [K, I] */, const void* u /* [M, 2I] */, void* du, float* colsum_partial,
                     int M, int I, int K, int lddy, int ldw, int ldu, int lddu, int act, float p_drop, unsigned seed, spn_stream_t s);

/* ---- element-wise (feedforward.py:13-21 GLU/act; attention.py:216-218 & mmd_transformer.py:213-214 row masks) */
/* p_drop > 0: nn.Dropout on the activation output (feedforward.py:57-60); the mask is a pure function of (seed, index) */
int spn_act_fwd(const void* u, long ldu, void* out, long ldo, long T, int I, int act, int glu, float p_drop, unsigned seed,
                spn_stream_t s);
int spn_act_bwd(const void* u, long ldu, const void* dout, long lddo, void* du, long lddu, long T, int I, int act, int glu,
                float p_drop, unsigned seed, float* colsum /* optional: += column sums of du (bias gradient) */,
                float* colsum_ws /* scratch, 4096 * width floats, required with colsum */, spn_stream_t s);
/* stand-alone nn.Dropout on [rows, D] (models/scoreperformer/transformer.py:122,184 `emb_dropout`; modules/transformer/feedforward.py:58 behind
 * a post-activation LayerNorm): y = keep ? x / (1 - p) : 0, counter-based mask of (seed, row, column) as in spn_act_fwd, so the backward is
 * the same call on dy with the same seed.  dtype 0 = fp32, 1 = bf16 (x and y alike); y may alias x */
int spn_dropout(const void* x, long ldx, void* y, long ldy, int dtype, long rows, int D, float p_drop, unsigned seed, spn_stream_t s);
int spn_cast(const void* x, int x_dtype, long x_bs, long x_ts, void* y, int y_dtype, long y_bs, long y_ts,
             const uint8_t* rowmask, long B, long t_len, int D, spn_stream_t s);
int spn_colsum(const void* x, int x_dtype, long ldx, float* out /* ACCUMULATED */, long T, int N, spn_stream_t s);
int spn_mish_fwd(const float* x, float* y, long n, spn_stream_t s);
int spn_mish_bwd(const float* x, const float* dy, float* dx, long n, spn_stream_t s);
/* x[r,:] = 0 where m[r] == 0, in place, bf16 [R, D]: the row mask of a Linear's backward applied to the bf16 gradient copy that the
 * LayerNorm backward already wrote (instead of a masked cast pass over the fp32 gradient) */
int spn_zero_masked_rows(void* x, long ldx, const uint8_t* m, long R, int D, spn_stream_t s);
int spn_rows_all_nonzero(const float* x, long ldx, uint8_t* mask, long R, int D, spn_stream_t s);
int spn_mask_rows(const float* x, long ldx, const uint8_t* m, float* y, long ldy, long R, int D, int invert, spn_stream_t s);

/* ---- embedding tables + tuple gather (modules/transformer/embeddings.py:118-152,202-213;
 *      models/scoreperformer/embeddings.py:121-143).  dw0/db0/db1, dtables, dgamma/dbeta ACCUMULATED. */
int spn_table_build_fwd(int nkeys, const float* const* tv, const float* const* w0, const float* const* b0,
                        const float* const* w1, const float* const* b1, const float* const* iw, float* const* out,
                        float* const* h1, const int* V, const int* E, int dense, int discrete, unsigned ids_mask,
                        spn_stream_t stream);
int spn_table_build_bwd(int nkeys, const float* const* tv, const float* const* w0, const float* const* b0,
                        const float* const* w1, const float* const* dout, float* const* dw0, float* const* db0,
                        float* const* db1, float* const* diw, float* const* dval, const int* V, const int* E, int dense,
                        int discrete, unsigned ids_mask, spn_stream_t stream);
int spn_embed_fwd(int nkeys, const float* const* tables, const int* V, const int* E, const long* tokens, long tok_bs,
                  long tok_ts, int t_len, const float* gamma, const float* beta, void* y, long ldy, float* mean, float* rstd,
                  int T, float eps, spn_stream_t stream);
int spn_embed_bwd(int nkeys, const float* const* tables, float* const* dtables, const int* V, const int* E, const long* tokens,
                  long tok_bs, long tok_ts, int t_len, const void* dy, long lddy, const float* gamma, const float* mean,
                  const float* rstd, float* dgamma, float* dbeta, float* ws, int T, int padding_idx, spn_stream_t stream);

/* ---- losses (models/scoreperformer/wrappers.py:49-59 CE; mmd_transformer.py:325-368 segments, 505-534 MMD) */
int spn_ce_fwd(const void* logits, int dtype, long ld, const long* labels, long lab_bs, long lab_ts, int t_len,
               int ignore_index, float* lse, float* sums /* [2] ACCUMULATED */, int* argmax, long T, int V, spn_stream_t s);
/* spn_ce_fwd + the evaluator's per-key sums in the same pass (models/scoreperformer/evaluator.py:38-45,73-104): over rows with a valid
 * label, metrics[0] += #(argmax == label); metrics[1] += |tv[argmax] - tv[label]| (weighted = 0) or sum_c softmax_c |tv[label] - tv[c]|
 * (weighted = 1); token_values = fp32 [V] or null (no distance).  metrics is ACCUMULATED; the valid count is sums[1]. */
int spn_ce_fwd_eval(const void* logits, int dtype, long ld, const long* labels, long lab_bs, long lab_ts, int t_len, int ignore_index,
                    float* lse, float* sums /* [2] ACCUMULATED */, int* argmax, const float* token_values, int weighted,
                    float* metrics /* [2] ACCUMULATED */, long T, int V, spn_stream_t s);
int spn_ce_bwd(const void* logits, int dtype, long ld, const long* labels, long lab_bs, long lab_ts, int t_len,
               int ignore_index, const float* lse, const float* coef, void* dlogits, long ldd, long T, int V, int Vpad,
               spn_stream_t s);
int spn_segment_count(const long* seg, float* counts /* ACCUMULATED */, int b, int t, int S, spn_stream_t s);
int spn_segment_sum(const void* x, int dtype, long x_bs, long x_ts, const long* seg, const float* counts,
                    const uint8_t* rowmask, float* out /* ACCUMULATED (sums); ZERO on entry for means (counts given): whole runs are stored, not added; row stride out_ld >= d */, long out_ld, int b, int t, int S, int d,
                    spn_stream_t s);
int spn_segment_gather(const float* src, long src_ld /* >= d */, const long* seg, const float* counts, const uint8_t* rowmask, float* y,
                       long y_ld, int b, int t, int S, int d, int accumulate, spn_stream_t s);
/* all latent levels in ONE pass over the hidden states (mmd_transformer.py:325-340 runs once per level): out_l[b, S_l, 0:d] (row stride
 * out_ld[l], zeroed by the caller) += segment MEANS of x[b, t, 0:d] * rowmask under seg_l; host arrays of nl <= 8 entries; d % 4 == 0 */
int spn_segment_sum_multi(const void* x, int dtype, long x_bs, long x_ts, const uint8_t* rowmask, int nl, const long* const* seg,
                          const float* const* counts, float* const* out, const long* out_ld, const int* S, int b, int t, int d,
                          spn_stream_t s);
/* its backward, written once: y[b * t, 0:d] = rowmask * sum_l src_l[b, seg_l, 0:d] / max(counts_l, 1) */
int spn_segment_gather_multi(int nl, const float* const* src, const long* src_ld, const long* const* seg, const float* const* counts,
                             const int* S, const uint8_t* rowmask, float* y, long y_ld, int b, int t, int d, spn_stream_t s);
int spn_mmd_fwd(const float* z, int Z, const float* y, const float* w, int N, int D, float* sums /* [4] ACCUMULATED */,
                spn_stream_t s);
int spn_mmd_bwd(const float* z, int Z, const float* y, const float* w, int N, int D, const float* coef, float* dy,
                spn_stream_t s);
/* scalar tail of compute_mmd (mmd_transformer.py:529-534) in one launch; n = max(sums[3], 1).  g == null: out[0] = the MMD value from
 * spn_mmd_fwd's sums; g != null (device scalar dL/dmmd): out[0..1] = the two coefficients spn_mmd_bwd takes */
int spn_mmd_scalars(const float* sums /* [4] */, int Z, const float* g, float* out, spn_stream_t s);

/* ---- latent stage behind the VAE head projections (csrc/latent.hip), one launch each instead of ~60 tensor ops per level.
 * spn_latent_select: MMDLoss.forward's `latents[mask]` + `randperm(N)[:max_num_latents]` (mmd_transformer.py:511-517) without a host
 * read -- a uniform random subset of at most K of the valid latents of one level, packed into y [K, D] with 0/1 row weights w [K]
 * (rows past the number of valid latents: weight 0, zeros); slot [N]: the row of y a latent went to or -1; and the deadpan sums of the
 * level (mmd:232-237,268-273) dead[3] = (sum of lat^2 over valid latents of flagged batch elements, their count, any-non-zero flag).
 * lat [N, D] fp32, valid [N] bytes, dead_b [N / S] bytes or null.  N <= 262144, K <= 4096, N / S <= 1024. */
int spn_latent_select(const float* lat, const uint8_t* valid, const uint8_t* dead_b, int N, int S, int D, int K, unsigned seed,
                      float* y, float* w, int* slot, float* dead, spn_stream_t s);
/* backward of the above: dlat[i] = (slot[i] >= 0 ? dy[slot[i]] : 0) + 2 g_dead / max(dead[1] D, 1) * lat[i] on valid latents of flagged
 * batch elements.  dy [K, D] or null; g_dead: device scalar dL/d(deadpan loss) or null */
int spn_latent_unselect(const float* dy, const int* slot, const float* lat, const uint8_t* valid, const uint8_t* dead_b, const float* dead,
                        const float* g_dead, int N, int S, int D, float* dlat, spn_stream_t s);
/* out[3] = (weight * MMD from spn_mmd_fwd's sums (mmd:529-534, loss_weight mmd:266), deadpan loss dead[0] / max(dead[1] D, 1), dead[2]) */
int spn_latent_scalars(const float* sums /* [4] */, int Z, const float* dead /* [3] */, int D, float weight, float* out, spn_stream_t s);
/* latent dropout of all levels in one pass over the style embeddings [b, n, W] (mmd_transformer.py:249-253,275-283,351-354,537-542): one
 * draw per valid latent (probability p[l]; 0: the level draws nothing), scattered to the latent's notes through seg[l] (int64 [b, n]; null:
 * S[l] == 1 one latent per sequence, else one per note), OR-ed into the following levels when `inclusive`, never on padded notes or
 * deadpan sequences.  out = emb with dropped columns zeroed, drop [b, n, W] bytes = the reference's dropout_mask.  Host arrays of nl <= 8
 * entries; col0[l] = first column of level l; given[l] (or given == null): explicit uint8 [b, S[l]] drop masks instead of draws. */
int spn_latent_drop(int nl, const long* const* seg, const uint8_t* const* lmask, const int* S, const int* col0, const float* p,
                    const uint8_t* const* given, int inclusive, const float* emb, const uint8_t* mask, const uint8_t* deadpan, int b, int n,
                    int W, unsigned seed, float* out, uint8_t* drop, spn_stream_t s);
int spn_latent_drop_bwd(const float* g, const uint8_t* drop, long total, float* dx, spn_stream_t s);

/* ---- optimizer (experiments/optimizers.py:151-169: clip_grad_norm_ + torch.optim.AdamW) over the flat arena */
int spn_sumsq(const float* g, long n, float* out /* ACCUMULATED */, spn_stream_t s);
/* the same with a fixed summation order: identical bits for identical g on every launch and every rank (the clip coefficient of
 * data-parallel replicas must not differ in its last bit).  ws: spn_sumsq_det_ws_floats() floats of caller-owned scratch. */
int spn_sumsq_det_ws_floats(void);
int spn_sumsq_det(const float* g, long n, float* out /* ACCUMULATED */, float* ws, spn_stream_t s);
/* slot_mask: optional uint8 [n / 8], one flag per 8-element arena slot; 0 = the slot's parameter has no gradient this step (frozen or
 * unused: torch.optim.AdamW skips grad-is-None parameters -- no decay, no moments) and is left untouched; null = update everything */
int spn_adamw_step(float* p, const float* g, float* m, float* v, void* shadow_bf16, const uint8_t* slot_mask, long n,
                   const float* normsq, float max_norm, float grad_scale, float lr, float beta1, float beta2, float eps,
                   float weight_decay, int step, spn_stream_t s);

/* ---- decode, b = 1 (per-note body of ScorePerformerMixedLMWrapper.unmask_tokens, models/scoreperformer/wrappers.py:325-407;
 *      cache protocol modules/transformer/transformer.py:159-181,219-221; attention.py:155-156; sampling.py:28-59).
 *      fp32; `pos` is a DEVICE int: row addressing = base + (*pos + off) * ld, so one captured hipGraph step is replayed per note. */
int spn_dec_gemv(const float* W, long ldw, const float* x, long x_ld, int x_off, const float* bias, const float* residual, float* y,
                 long y_ld, int y_off, const int* pos, int N, int K, int kn_layout, spn_stream_t s);
int spn_dec_embed(int nkeys, const float* const* tables, const int* E, const long* tokens, long tok_ld, int row_off, const int* pos,
                  const float* gamma, const float* beta, float* y, float eps, spn_stream_t s);
/* spn_dec_embed + the embeddings' projection GEMV for both sequences of a multi-sequence decoder in one launch (bit-identical values) */
int spn_dec_embed_proj(int nkeys, const float* const* tables, const int* E, const long* tokens_a, const long* tokens_b, long tok_ld,
                       const int* pos, const float* gamma, const float* beta, float eps, const float* W, long ldw, const float* bias, float* y,
                       int N, spn_stream_t s);
/* First launch of a fused step: spn_dec_embed_proj, plus the position latch (*pos_latch = *pos: every later launch of the step reads
 * pos_latch, the LAST one -- spn_dec_head with pos_next = pos -- writes position + 1 back, so no launch reads a scalar that is written in
 * the same launch and the one-thread spn_dec_add_pos between two notes is gone), plus an optional rider GEMV in the same launch:
 * ry[n] = rW[n, :] . rx[(*pos + rx_off) * rx_ld ..] + rbias[n] (the stacked AdaLN condition projections of the step); rW = null: none. */
int spn_dec_step_begin(int nkeys, const float* const* tables, const int* E, const long* tokens_a, const long* tokens_b, long tok_ld,
                       const int* pos, int* pos_latch, const float* gamma, const float* beta, float eps, const float* W, long ldw,
                       const float* bias, float* y, int N, const float* rW, long r_ldw, int rN, int rK, const float* rx, long rx_ld,
                       int rx_off, const float* rbias, float* ry, spn_stream_t s);
int spn_dec_copy_row(const float* src, long src_ld, int src_off, float* dst, long dst_ld, int dst_off, const int* pos, int D,
                     spn_stream_t s);
int spn_dec_glu(const float* u, float* out, int I, int act, int glu, spn_stream_t s);
int spn_dec_attn(const float* qkv, float* kcache, float* vcache, const float* slopes, const int* pos, float* o, int h, int kvh,
                 float scale, spn_stream_t s);
int spn_dec_argmax_write(const float* logits, int V, unsigned ban_mask, long* tokens, long tok_ld, int dim, int mask_id, const int* pos,
                         spn_stream_t s);
int spn_dec_add_pos(int* pos, int delta, spn_stream_t s);
/* fused step kernels (a decode step is launch-latency bound): LayerNorm/AdaLN prologue + GEMV + GLU/bias/residual epilogue (+ cache
 * row mirror); (LN(x) | context row | style row) concatenation; split-key single-query attention with the ALiBi reach and a running
 * max |k|^2; LM head (LayerNorm + per-dim logits + banned ids + arg-max write) for all candidate dims in one launch */
int spn_dec_fused_gemv(const float* W, long ldw, int N, int K, const float* x, long x_ld, int x_off, int norm, const float* gamma,
                       const float* beta, float eps, const float* bias, const float* residual, float* y, long y_ld, int y_off, float* y2,
                       long y2_ld, int y2_off, float* xn_out, long xn_ld, int xn_off, int glu, int act, const int* pos, spn_stream_t s);
/* o = null in spn_dec_attn2 / spn_dec_xattn leaves only the split-key partials; this GEMV merges them in its prologue */
int spn_dec_attn_out(const float* W, long ldw, int N, const float* part, int h, int splits, const float* residual, float* y, spn_stream_t s);
int spn_dec_cat(const float* x, int d, const float* gamma, const float* beta, float eps, const float* ctx, long ctx_ld, int ctx_w,
                const float* style, long style_ld, int style_w, const int* pos, float* out, spn_stream_t s);
/* spn_dec_cat + the projection GEMV over its output in one launch: y = W . (LN?(x[0:d]) | ctx[*pos + 1] | style[*pos + 1]) + bias, y2 = row
 * *pos mirror (models/scoreperformer/transformer.py:160-181 for one position) */
int spn_dec_cat_gemv(const float* W, long ldw, int N, const float* x, int d, const float* gamma, const float* beta, float eps,
                     const float* ctx, long ctx_ld, int ctx_w, const float* style, long style_ld, int style_w, const float* bias, float* y,
                     float* y2, long y2_ld, const int* pos, spn_stream_t s);
int spn_dec_attn2(const float* qkv, float* kcache, float* vcache, const float* slopes, const int* pos, float* o, float* part, int* counter,
                  float* kmax2, int h, int kvh, float scale, int splits, spn_stream_t s);
/* cross-attention of the decoded position over a static context (decoder layer type 'c': modules/transformer/transformer.py:201 under
 * the cache protocol :159-181; ALiBi distance from the END of the context as attention.py:193-197 gives it for a single query) */
int spn_dec_xattn(const float* q, const float* kctx, const float* vctx, const float* slopes, const uint8_t* kmask, int nk, float* o,
                  float* part, int* counter, int h, int kvh, float scale, int splits, spn_stream_t s);
/* the same with the number of context rows read from device memory at run time (render sessions: the context grows under one graph) */
int spn_dec_xattn_dyn(const float* q, const float* kctx, const float* vctx, const float* slopes, const uint8_t* kmask, const int* nk_dev,
                      float* o, float* part, int* counter, int h, int kvh, float scale, int splits, spn_stream_t s);
/* out[0] = tab[*pos] */
int spn_dec_lookup(const int* tab, const int* pos, int* out, spn_stream_t s);

/* One pre-norm decoder layer pair -- self-attention block + gated feed-forward block of a cached decode step
 * (modules/transformer/transformer.py:159-221, attention.py:107-222, feedforward.py:13-64) -- as ONE persistent launch of spn_dec_pair_groups() workgroups
 * whose five phases hand their vectors over inside the launch (8-byte {epoch, value} granules, agent-scope stores and polls:
 * csrc/decode_layer.hip).  Same arithmetic and association order as spn_dec_fused_gemv + spn_dec_attn2 + spn_dec_attn_out +
 * spn_dec_fused_gemv(glu) + spn_dec_fused_gemv: x leaves bit-identical.  The granule buffers must be zeroed once per render; `tick`
 * (device int, advanced by the launch with bump = 1: the LAST pair of a note) makes the epochs unique; *err != 0 after a launch means a
 * hand-off timed out (the launch never hangs; results are then garbage).
 * RESIDENCY: every workgroup polls results of the others, so all spn_dec_pair_groups() workgroups (one per CU) must run at once: the
 * call refuses a device with fewer CUs, and nothing else may occupy CUs of the device while the launch runs (another stream's or another
 * process's kernel that holds CUs long enough makes the hand-offs time out: *err, never a hang). */
typedef struct spn_dec_pair_args {
    const float* Wqkv; long ld_qkv;            /* [(h + 2 kvh) * 64, d]: q | k | v rows */
    const float* Wo; long ld_o;                /* [d, h * 64] */
    const float* W1; long ld_1; const float* b1;   /* [2 * inner, d]: value rows, then gate rows; bias [2 * inner] or null */
    const float* W2; long ld_2; const float* b2;   /* [d, inner]; bias [d] or null */
    const float* slopes;                       /* [h] ALiBi slopes or null */
    float* kcache; float* vcache; float* kmax2;    /* [L, kvh * 64] x 2, running max |k|^2 [kvh] */
    int* jlo;                                  /* null, or [h] (zeroed once): first key inside the ALiBi reach at the previous note -- the guess
                                                  behind the early key / value requests */
    int norm1; const float* gam1; const float* bet1; float eps1;   /* 1 = LayerNorm(gamma, beta; null = plain), 2 = adaptive row (gamma | beta) */
    int norm2; const float* gam2; const float* bet2; float eps2;
    float* x;                                  /* [d] residual stream, in and out */
    float* y2; long y2_ld;                     /* null, or the hidden-cache mirror of the pair's output: y2[*pos * y2_ld + n] */
    int d, h, kvh, inner, S, act;              /* S = key splits (<= 16), act 0 = SiLU, 1 = GELU */
    float scale;
    const int* pos; int* tick; int layer, bump;
    unsigned long long* gq; unsigned long long* gp; unsigned long long* go; unsigned long long* gx; unsigned long long* gg;
    unsigned long long* gxo;                   /* granules: (h + 2 kvh) * 64, h * S * 66, h * 64, d, inner, d (gxo: the stream to the next pair of a chain) */
    int* err;
    long long* stamps;                         /* null, or [groups][8] constant-clock (100 MHz) time stamps after each phase: tuning aid */
} spn_dec_pair_args;
int spn_dec_pair_groups(int d, int h, int kvh, int inner, int S); /* workgroups of the launch for this shape; 0 = not supported */
/* A chain of n consecutive pairs (1 <= n <= 32, one shape, ascending `layer`) in ONE launch: the residual stream goes from pair to pair
 * inside the launch (gxo).  `host` = the n records (validated here), `dev` = the same records in device memory (read by the launch).  x
 * of the first record is read, x of the last one written; every pair writes its y2 mirror. */
int spn_dec_pairs(const spn_dec_pair_args* host, const spn_dec_pair_args* dev, int n, spn_stream_t s);
/* Phases around the chain, in the same launch (either part may be absent: Wm = null / Wh = null):
 *   front (in front of the first pair; its result replaces x as the chain's input):
 *       x0 = Wm . xin + bm                                                   (spn_dec_fused_gemv of the multi-sequence projection)
 *       x  = Wp . (LN?(x0) | ctx[*pos + 1] | style[*pos + 1]) + bp           (spn_dec_cat_gemv)
 *   tail (behind the last pair):  e_out = Wh . LN(x), the normalised x mirrored  (spn_dec_fused_gemv of the LM head's input projection)
 *   head (behind the tail; hn = 0: none, needs the tail): the arg-max LM head of spn_dec_head over e -- LayerNorm(e) slice . table_q^T
 *       per decoded key q < hn, banned ids, arg-max (ties to the lower id), written to tokens[(*pos + 1) * tok_ld + hdim[q]] where that
 *       cell holds mask_id; *pos_next = *pos + 1.  e travels to the feed-forward workgroups as granules (ge), their per-key partial
 *       maxima to the key's first workgroup as granules (gh).
 * Same arithmetic as those launches. */
typedef struct spn_dec_chain_ext {
    const float* Wm; long ld_m; const float* bm; const float* xin; int Km;          /* [d, Km], Km <= 1024 */
    float* y2m; long y2m_ld;                                                         /* null, or mirror of x0: y2m[*pos * ld + n] */
    const float* Wp; long ld_p; const float* bp;                                     /* [d, d + ctx_w + style_w] (<= 2048 columns) */
    const float* cat_gamma; const float* cat_beta; float cat_eps;                    /* LayerNorm of x0 (null gamma: none) */
    const float* ctx; long ctx_ld; int ctx_w; const float* style; long style_ld; int style_w;
    float* y2p; long y2p_ld;                                                         /* null, or mirror of x */
    unsigned long long* gf; unsigned long long* gxf;                                 /* granules [d] each (zeroed once per render) */
    const float* Wh; long ld_h; int Nh;                                              /* [Nh, d], Nh <= 16 h S */
    int normh; const float* gamh; const float* beth; float epsh;                     /* as norm1 of a pair */
    float* e_out; float* xn_out; long xn_ld;                                         /* [Nh]; null, or xn_out[*pos * ld + k] = LN(x)[k] */
    int hn; int hD;                                                                  /* decoded keys (<= 16), width of e (= Nh <= 2048) */
    const float* htable[16]; int hV[16]; int hwidth[16]; int hcol0[16]; int hdim[16]; /* per key: [V, width] table, slice of e, token column */
    const float* hgamma; const float* hbeta; float heps; unsigned hban;              /* the head's LayerNorm over e; bit v < 32 set = id v banned */
    long* tokens; long tok_ld; int mask_id; int* pos_next;                           /* as spn_dec_head */
    unsigned long long* ge; unsigned long long* gh;                                  /* granules [Nh], [16 * 16 * 2] (zeroed once per render) */
    /* embed phase (en = 0: none; needs the front, whose xin it replaces): the two token-tuple embeddings of the note and their projection,
     * spn_dec_embed_proj: xin[seq * eN + n] = We[n, :] . LN(concat_k etable_k[token_k]) + be[n], tokens tok_a[*pos] (seq 0) and
     * tok_b[*pos + 1] (seq 1); the attention workgroups compute it (eR <= 2 rows per wave) and hand it to the front as granules (gin). */
    int en; int eD; int eN; int eR;                                                  /* keys (<= 16), embedding width (<= 2048), rows per sequence, rows per wave */
    const float* etable[16]; int ewidth[16]; int ecol0[16];
    const long* tok_a; const long* tok_b; long etok_ld;
    const float* egamma; const float* ebeta; float eeps;                             /* LayerNorm of the embedding (null gamma: none) */
    const float* We; long ld_e; const float* be;
    unsigned long long* gin;                                                         /* granules [2 eN] */
    /* AdaLN rows of the NEXT note (rW = null: none), the rider of spn_dec_step_begin one note early: ry[par + n] = rW[n, :] .
     * rx[min(*pos + 2, rx_rows - 1), :] + rbias[n] with par = ada_par on even *pos, 0 on odd *pos; the adaptive norms (mode 2) of THIS
     * note read their (gamma | beta) rows at + ada_par when *pos is odd.  The caller computes the first note's rows (spn_dec_gemv). */
    const float* rW; long r_ldw; int rN; int rK; const float* rx; long rx_ld; int rx_rows; const float* rbias; float* ry; long ada_par;
    /* several notes per launch (spn_dec_pairs_notes): granules [16] (zeroed once per render) through which the head's winners hand the
     * final value of their token cells to the next note's embed phase; null: one note per launch only */
    unsigned long long* gt;
    /* sampling in the head phase (stopk = null: arg-max): the decision of spn_dec_head_sample -- keep the stopk[q] largest logits of key q,
     * weights exp((l - max) * sinv_temp), one draw at the counter hash of (*sseed, position, key) -- inside the launch; gl: granules
     * [16 * 1024] (zeroed once per render) through which the slabs hand their logits to the key's first workgroup */
    unsigned long long* gl; const int* stopk; float sinv_temp; const unsigned* sseed;
} spn_dec_chain_ext;
int spn_dec_pairs_ext(const spn_dec_pair_args* host, const spn_dec_pair_args* dev, int n, const spn_dec_chain_ext* ext_host,
                      const spn_dec_chain_ext* ext_dev, spn_stream_t s);
/* `notes` consecutive notes (positions *pos .. *pos + notes - 1, all of them to be decoded) in ONE launch: needs the embed, front, tail and
 * head phases and ext.gt; position and tick advance by `notes`.  Same tokens, hidden rows and cache rows as `notes` launches of
 * spn_dec_pairs_ext. */
int spn_dec_pairs_notes(const spn_dec_pair_args* host, const spn_dec_pair_args* dev, int n, const spn_dec_chain_ext* ext_host,
                        const spn_dec_chain_ext* ext_dev, int notes, spn_stream_t s);
int spn_dec_struct_size(int which); /* sizeof(spn_dec_pair_args) (0) / sizeof(spn_dec_chain_ext) (1): lets a binding check its record layout */
int spn_dec_head(int n, const float* const* tables, const int* V, const int* width, const int* col0, const int* dim, int D, const float* e,
                 const float* gamma, const float* beta, float eps, unsigned ban_mask, long* tokens, long tok_ld, int mask_id,
                 const int* pos, float* part /* n*slabs*2 */, int* counter /* n, zeroed once */, int slabs,
                 int* pos_next /* null, or: *pos_next = *pos + 1 (see spn_dec_step_begin) */, spn_stream_t s);

/* spn_dec_head with sampling instead of the arg-max (modules/sampling.py:28-59: top_k filter, temperature, one multinomial draw):
 * logits = scratch [n, ldl]; topk[n] = ids kept per key (device); seed = device scalar mixed with the position and the key. */
int spn_dec_head_sample(int n, const float* const* tables, const int* V, const int* width, const int* col0, const int* dim, int D,
                        const float* e, const float* gamma, const float* beta, float eps, unsigned ban_mask, long* tokens, long tok_ld,
                        int mask_id, const int* pos, float* part, int* counter, int slabs, float* logits, int ldl, const int* topk,
                        float temperature, const unsigned* seed, int* pos_next, spn_stream_t s);

/* batched (prefill) forms of the decode kernels: rows t0 .. t0+n-1 of a window whose tokens are known (the reference recomputes a
 * cropped window in one batched forward, inference/generators.py:184-241 -> wrappers.py:391-393); K/V rows already in the caches. */
int spn_dec_attn_rows(const float* q, long q_ld, const float* kcache, const float* vcache, const float* slopes, int t0, int n, float* o,
                      long o_ld, int h, int kvh, float scale, spn_stream_t s);
int spn_dec_glu_rows(const float* u, long u_ld, float* out, long out_ld, int n, int I, int act, int glu, spn_stream_t s);

/* ---- device-side batch builder (scoreperformer/data/collators/score_performance.py:35-115,186-234; performance.py:18-92,100-115,
 *      213-247): raw ragged int32 tokens (samples concatenated, row offsets [b+1]) -> every tensor the MixedLM collator returns:
 *      padded int64 tokens, bool masks, lengths, bar/beat/onset ids (seg_flat = [3, sum_s] or null), deadpan flags, masked_perf, labels.
 *      ignore_ids: host array (<= 16 ids; pad is always ignored); ignore_dims: bit k = token dim k is never masked. */
int spn_collate_mixlm(const int32_t* score_flat, const int32_t* perf_flat, const int32_t* seg_flat, const int32_t* score_off,
                      const int32_t* perf_off, const uint8_t* deadpan, int b, int Ks, int Kp, int Ls, int Lp, long sum_s, int pad_id,
                      int mask_id, int label_pad_id, const int* ignore_ids, int n_ignore, unsigned ignore_dims, int label_pad_ignored_dims,
                      long long* score, uint8_t* score_mask, long long* score_len, long long* perf, uint8_t* perf_mask, long long* perf_len,
                      long long* masked_perf, long long* labels, long long* bar, long long* beat, long long* onset, uint8_t* deadpan_mask,
                      spn_stream_t stream);

/* one more ragged token array of the batch, padded: the noisy performance (score_performance.py:48-51,66-69,94-95) */
int spn_collate_pad_tokens(const int32_t* flat, const int32_t* off, int b, int K, int L, int pad_id, long long* out, uint8_t* mask,
                           long long* len, spn_stream_t stream);

/* ---- data-parallel collective: thin wrappers over RCCL's ncclAllReduce on a dedicated communication stream, event-fenced against
 *      the producer / consumer streams (replaces the gradient all-reduce a data-parallel trainer.py run
 *      performs through torch.distributed; scoreperformer_amd/parallel.py launches one call per gradient bucket from inside backward).
 *      RCCL is bound at run time (dlopen, RTLD_NOLOAD first: the copy PyTorch has loaded is reused; rccl_path null = "librccl.so").
 *      id128: 128 bytes made by spn_comm_unique_id on one rank and handed to every rank out of band.  spn_comm_init is collective.
 *      spn_comm_allreduce: in-place sum of buf[0 .. count) (dtype 0 fp32, 1 bf16) ordered behind everything enqueued on
 *      producer_stream so far; returns at once.  spn_comm_wait: consumer_stream waits for every all-reduce enqueued so far.
 *      No host synchronisation in allreduce / wait; streams / events are created in init (on the device current at that call, which
 *      every later entry point re-selects for the duration of the call; a buffer on another device is rejected) and released in
 *      destroy, which -- the one exception in this library -- first waits for the communication stream to drain. */
int spn_comm_available(const char* rccl_path); /* SPN_OK when RCCL can be bound in this process; not collective, makes no id */
int spn_comm_unique_id(void* id128, const char* rccl_path);
int spn_comm_init(void** comm, int nranks, int rank, const void* id128, const char* rccl_path);
int spn_comm_allreduce(void* comm, void* buf, size_t count, int dtype, spn_stream_t producer_stream);
int spn_comm_wait(void* comm, spn_stream_t consumer_stream);
int spn_comm_destroy(void* comm);

#ifdef __cplusplus
}
#endif
#endif /* SPN_H */
