/* nr_hip.h -- C ABI of libnr_hip.so: the MI355X (gfx950) kernels behind NeighborRetr's
 * similarity / neighbour-weighting / loss head.
 *
 * The reference has no FFI layer: its hot path is a sequence of ATen calls inside two Python
 * modules.  Each entry point below replaces one such sequence; the reference
 * lines it stands in for are cited per function (paths relative to the reference checkout).
 * The binding a maintainer adds on the reference side is a ctypes stub -- see INTEGRATION.md.
 *
 * Conventions (all functions):
 *   - plain device pointers + extents, contiguous row-major, no torch types;
 *   - `stream` is a hipStream_t passed as void*; work is enqueued, never synchronised;
 *   - no allocation inside: scratch comes in through `workspace` arguments whose size the
 *     matching *_workspace_bytes() query returns;
 *   - return value: 0 ok, <0 invalid argument / unsupported shape (NR_E*), >0 a hipError_t;
 *   - never throws, never writes outside the buffers named in the signature.
 */
#ifndef NR_HIP_H
#define NR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NR_ABI_VERSION 1

/* precision of the MFMA contractions */
#define NR_PREC_BF16 0   /* one bf16 pass (training path)                                   */
#define NR_PREC_BF16X3 1 /* split-bf16, 3 passes: ~fp32 products (eval / rank-parity path)  */

/* what nr_local_level_fwd writes */
#define NR_OUT_FULL 0   /* S[A,Bv]                                                          */
#define NR_OUT_ROWSUM 1 /* partial[n_col_tiles, A]  : sum over the tile's columns (videos)  */
#define NR_OUT_COLSUM 2 /* partial[n_row_tiles, Bv] : sum over the tile's rows (texts)      */

int nr_version(void);

/* F.normalize (eps 1e-12) + mask multiply + bf16 hi/lo split of a token matrix.
 * Replaces modeling.py:495-496 and the two mask einsums :500-501 (a masked token becomes a zero
 * vector, so every product with it is exactly 0, as in the reference).
 *   x [n_tok,d] f32; mask [n_tok] f32 or NULL; normalize: 1 = L2-normalise rows, 0 = keep scale
 *   hi, lo [n_tok,d] bf16 (lo may be NULL); norm [n_tok] f32 = max(||x||,1e-12) (may be NULL);
 *   colsum_part [nr_prepare_parts(n_tok), d] f32 or NULL: per-workgroup column sums of the
 *   UNMASKED normalised rows (feeds compute_centrality_weights, modeling.py:413-424).        */
int nr_prepare_parts(int n_tok);
int nr_prepare_tokens(const float* x, const float* mask, int n_tok, int d, int normalize,
                      uint16_t* hi, uint16_t* lo, float* norm, float* colsum_part, void* stream);

/* plain f32 -> bf16 hi/lo split (MLP weight matrices). */
int nr_split_bf16(const float* x, size_t n, uint16_t* hi, uint16_t* lo, void* stream);

/* Token-weight MLP, Linear(d,H)+ReLU+Linear(H,1), on the prepared tokens (modeling.py:148-153,
 * called at :485 and :490).  logit_part[p, tok] holds the contribution of hidden units
 * [128p, 128p+128); the H/128 parts are summed by nr_token_softmax.
 *   tok_hi/lo [n_tok,d] normalised tokens, norm [n_tok] their original norms (row scale),
 *   w1_hi/lo [H,d], b1 [H], w2 [H] f32.                                                       */
int nr_token_logits_fwd(const uint16_t* tok_hi, const uint16_t* tok_lo, const float* norm, int n_tok, int d,
                        const uint16_t* w1_hi, const uint16_t* w1_lo, const float* b1, const float* w2, int H,
                        int prec, float* logit_part, void* stream);

/* masked_fill(-9e15) + softmax over the token axis (modeling.py:486-487 / :491-492).
 *   logit_part [n_parts, n_samples*N]; b2 [1] device; mask [n_samples*N] f32 or NULL;
 *   w [n_samples, N] out; logits [n_samples*N] out or NULL (pre-mask logits, kept for backward) */
int nr_token_softmax(const float* logit_part, int n_parts, const float* b2, const float* mask,
                     int n_samples, int N, float* w, float* logits, void* stream);

/* Fused local_level (modeling.py:499-512): token-token cosine products on MFMA, max-pool over
 * each token axis, weighted sums, (t2v+v2t)/2.  The [A,Bv,Nt,Nv] tensor is never materialised.
 *   t_hi/lo [A*Nt,d], v_hi/lo [Bv*Nv,d] prepared tokens; w_t [A*Nt], w_v [Bv*Nv] token weights;
 *   out per out_mode (see NR_OUT_*); arg_v [A,Bv,Nt] / arg_t [A,Bv,Nv] u8 arg-max indices or NULL.
 * nr_local_level_tiles reports the tile grid the kernel will use (sizes of the partial outputs). */
int nr_local_level_tiles(int A, int Nt, int Bv, int Nv, int* n_row_tiles, int* n_col_tiles);
int nr_local_level_fwd(const uint16_t* t_hi, const uint16_t* t_lo, const uint16_t* v_hi, const uint16_t* v_lo,
                       const float* w_t, const float* w_v, int A, int Nt, int Bv, int Nv, int d,
                       int prec, int out_mode, float* out, uint8_t* arg_v, uint8_t* arg_t, void* stream);

/* out[i] = scale * sum_p part[p, i]   (memory-bank centrality, until_module.py:181) */
int nr_reduce_parts(const float* part, int n_parts, int n, float scale, float* out, void* stream);

/* C[M,N] = A[M,K] * B[N,K]^T in exact fp32 (v_mfma_f32_16x16x4_f32): the one-global-token case
 * of global_level (modeling.py:526-537), whose logits are un-normalised and feed Sinkhorn. */
int nr_gemm_nt_f32(const float* a, const float* b, int M, int N, int K, float* c, void* stream);

/* compute_centrality_weights (modeling.py:403-430) from the prepared column sums:
 *   w[i] = exp(scale * <g_i/||g_i||, mean_tok>),  mean_tok = sum_p colsum_part[p,:] / n_tok.
 *   g [B,d] f32 global tokens; w [B]; gnorm [B] out or NULL (kept for backward);
 *   mean_out [d] out or NULL.                                                                */
int nr_centrality_weights(const float* g, int B, int d, const float* colsum_part, int n_parts, int n_tok,
                          float scale, float* w, float* gnorm, float* mean_out, void* stream);

/* Log-domain Sinkhorn targets, both directions in one launch (until_module.py:235-266):
 *   tgt_rows = beta*Q(G) + (1-beta)*I,  tgt_cols = beta*Q(G^T) + (1-beta)*I  (each [B,B],
 *   tgt_cols indexed in the transposed frame).  workspace: nr_sinkhorn_workspace_bytes(B).    */
size_t nr_sinkhorn_workspace_bytes(int B);
int nr_sinkhorn_targets(const float* G, int B, float beta, int iters, float* tgt_rows, float* tgt_cols,
                        void* workspace, void* stream);

/* Row-wise fused losses, both directions (until_module.py:303-328 centrality, :161-211 neighbour,
 * :285-289 uniform CE, :351-357 KL; orchestration modeling.py:329-401).  One wave per
 * (row, direction); direction 1 works on the transposed matrices.
 *   S, G [B,B]; tgt_rows/tgt_cols from nr_sinkhorn_targets; bank_c0 [B] = mean_m local_level(bank_text,
 *   video).T (used by the t2v loss), bank_c1 [B] = mean_m local_level(text, bank_video) (v2t loss);
 *   wc_text / wc_video [B] centrality weights; logit_scale [1] device f32.
 *   rowloss [2,4,B] out (dir, {centrality, uniform, neighbour, kl}, row).
 * nr_loss_finalize reduces rowloss to losses[5] = (total, centrality, uniform, neighbour, kl).  */
int nr_row_losses_fwd(const float* S, const float* G, const float* tgt_rows, const float* tgt_cols,
                      const float* bank_c0, const float* bank_c1, const float* wc_text, const float* wc_video,
                      const float* logit_scale, int B, int K, float temperature, float* rowloss, void* stream);
int nr_loss_finalize(const float* rowloss, int B, float uniform_weight, float neighbor_weight, float kl_weight,
                     float* losses, void* stream);

/* Backward of nr_row_losses_fwd + nr_loss_finalize for d(total): recomputes the row statistics.
 *   g_losses [5] upstream gradients of (total, centrality, uniform, neighbour, kl) (device);
 *   dS_dir [2,B,B] (direction 1 in the transposed frame), dG_dir [2,B,B], d_bank_c [2,B]
 *   (per-row contributions are reduced inside), d_wc [2,B], d_logit_scale_rows [2,B].        */
int nr_row_losses_bwd(const float* S, const float* G, const float* tgt_rows, const float* tgt_cols,
                      const float* bank_c0, const float* bank_c1, const float* wc_text, const float* wc_video,
                      const float* logit_scale, int B, int K, float temperature,
                      float uniform_weight, float neighbor_weight, float kl_weight, const float* g_losses,
                      float* dS_dir, float* dG_dir, float* d_bank_c_rows, float* d_wc, float* d_ls_rows,
                      void* stream);

/* Backward of nr_local_level_fwd w.r.t. the prepared (normalised, masked) tokens and the token
 * weights, routed through the stored arg-max indices (max-pool backward = scatter to the arg-max).
 *   dS [A,Bv] upstream (for the bank modes the caller expands d_mean/M into it);
 *   t_n/v_n: normalised masked tokens as f32 [A*Nt,d] / [Bv*Nv,d];
 *   d_tn [A*Nt,d], d_vn [Bv*Nv,d] (either may be NULL), d_wt [A*Nt], d_wv [Bv*Nv] out.       */
int nr_local_level_bwd(const float* dS, const float* t_n, const float* v_n, const float* w_t, const float* w_v,
                       const uint8_t* arg_v, const uint8_t* arg_t, int A, int Nt, int Bv, int Nv, int d,
                       float* d_tn, float* d_vn, float* d_wt, float* d_wv, void* stream);

/* Memory-bank FIFO push (modeling.py:237-249): bank <- cat(batch, bank)[:capacity] done as an
 * in-place shift; rows are `row_bytes` wide.  Requires 0 < n_new; if n_new >= capacity the bank
 * becomes the first `capacity` rows of the batch.  scratch: capacity*row_bytes bytes.          */
int nr_bank_push(void* bank, const void* batch, int capacity, int n_new, size_t row_bytes, void* scratch,
                 void* stream);

/* Rank of the diagonal in every row under the reference's tie rule (metrics.py:58-66):
 *   greater[i] = #{j : S[i,j] > S[i,i]},  equal[i] = #{j : S[i,j] == S[i,i]} (includes j=i). */
int nr_diag_ranks(const float* S, int N, int32_t* greater, int32_t* equal, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NR_HIP_H */
