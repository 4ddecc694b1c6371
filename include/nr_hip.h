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

/* bumped whenever a signature or the layout of a descriptor struct changes (2: NrCtmStageDesc gained x_hi/x_lo/out_hi/out_lo in
 * round 3, nr_stream_create / nr_stream_destroy, NrBankAbsorbDesc in round 4; 3: nr_pack_shard_convert); a binding compares nr_version() with the value it was written for */
#define NR_ABI_VERSION 5

/* precision of the MFMA contractions */
#define NR_PREC_BF16 0   /* one bf16 pass (training path)                                   */
#define NR_PREC_BF16X3 1 /* split-bf16, 3 passes: ~fp32 products (eval / rank-parity path)  */

/* what nr_local_level_fwd writes */
#define NR_OUT_FULL 0   /* S[A,Bv]                                                          */
#define NR_OUT_ROWSUM 1 /* partial[n_col_tiles, A]  : sum over the tile's columns (videos)  */
#define NR_OUT_COLSUM 2 /* partial[n_row_tiles, Bv] : sum over the tile's rows (texts)      */

int nr_version(void);
/* sizeof of one of this header's descriptor structs by name ("NrSplitItem", ...; 0: unknown name): lets a foreign-language
 * binding verify its mirror of the layout at load time. */
size_t nr_struct_size(const char* name);

/* Identity of the HIP stream capture `stream` belongs to (0 = not capturing).  No counterpart in the reference (it launches
 * eagerly, trainer.py:84-110); used by neighborretr_amd/capture_guard.py to validate the step's fork / join topology per
 * capture before handing it to the runtime.  Host-only, no launch, no sync. */
int nr_stream_capture_id(void* stream, unsigned long long* id);

/* A new non-blocking HIP stream (hipStreamCreateWithFlags) / its destruction.  No counterpart in the reference.  The host code
 * forks the step's branches onto streams created for ONE graph capture each (neighborretr_amd/streams.py): a stream object
 * that has taken part in an earlier capture is never handed to a later one.  Host-only. */
int nr_stream_create(void** stream);
int nr_stream_destroy(void* stream);

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

/* nr_prepare_tokens for TWO token matrices of the same width (the batch's text and video tokens) in one launch. */
int nr_prepare_tokens_pair(const float* x0, const float* mask0, int n_tok0, uint16_t* hi0, uint16_t* lo0, float* norm0,
                           float* colsum_part0, const float* x1, const float* mask1, int n_tok1, uint16_t* hi1, uint16_t* lo1,
                           float* norm1, float* colsum_part1, int d, int normalize, void* stream);

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

/* Backward of the token-weight MLP's hidden layer, fused (the reference's autograd does this in ~10 element-wise launches on
 * [n_tok, H] tensors behind modeling.py:148-153): recomputes h = norm * (tok W1^T) + b1 like nr_token_logits_fwd and writes
 *   dh = (h > 0) * dl[t] * w2 as bf16 pairs -- transposed into dhT [H, ldT] at columns [t0, t0 + n_tok) (the K = tokens operand
 *   of dW1 = dh^T X; columns up to the next multiple of 64, as far as ldT reaches, are written as zeros: the K padding;
 *   t0 % 8 == 0, ldT % 8 == 0; dhT_lo may be NULL when the weight-gradient GEMM is one-pass) and, when dh_hi/lo are given,
 *   row-major [n_tok, H] (operand of dX = dh W1);
 *   dw2_part / db1_part [nr_token_mlp_bwd_part_rows(n_tok, H, prec, hi_only), H] (hi_only: the call passes dhT_lo = dh_hi =
 *   dh_lo = NULL, which lets the large one-pass sets run a bigger block): partial column sums of dl * relu(h) and of dh;
 *   dl_part [the same number of rows] (may be NULL): partial sums of dl (the gradient of the output bias).
 * dl [n_tok] f32 = gradient of the logits (0 on masked tokens).                                                            */
int nr_token_mlp_bwd_row_tiles(int n_tok);
int nr_token_mlp_bwd_part_rows(int n_tok, int H, int prec, int hi_only);
int nr_token_mlp_bwd_hidden(const uint16_t* tok_hi, const uint16_t* tok_lo, const float* norm, int n_tok, int d,
                            const uint16_t* w1_hi, const uint16_t* w1_lo, const float* b1, const float* w2, int H, int prec,
                            const float* dl, uint16_t* dhT_hi, uint16_t* dhT_lo, int ldT, int t0, uint16_t* dh_hi, uint16_t* dh_lo,
                            float* dw2_part, float* db1_part, float* dl_part, void* stream);

/* masked_fill(-9e15) + softmax over the token axis (modeling.py:486-487 / :491-492).
 *   logit_part [n_parts, n_samples*N]; b2 [1] device; mask [n_samples*N] f32 or NULL;
 *   w [n_samples, N] out; logits [n_samples*N] out or NULL (pre-mask logits, kept for backward) */
int nr_token_softmax(const float* logit_part, int n_parts, const float* b2, const float* mask,
                     int n_samples, int N, float* w, float* logits, void* stream);

/* The two calls above in ONE launch: scorer MLP + masked softmax over each sample's tokens (modeling.py:485-487 /
 * :490-492).  The H/BN column blocks of a row tile add to the tile's counter once their partial logits are visible
 * device-wide; the block that arrives last sums the parts, applies mask and softmax for the tile's samples and resets the
 * counter.  counters [n_counters] u32, zero on entry (the kernel leaves them zero), n_counters >= ceil(n_samples*N / 64);
 * launches that may run concurrently need separate counters.  NR_EUNSUPPORTED when no block shape holds whole samples
 * (N must divide 96 or 192, or 64 / 128 -- the caller then issues the two calls above) or N > 256.                        */
int nr_token_weights_fwd(const uint16_t* tok_hi, const uint16_t* tok_lo, const float* norm, int n_samples, int N, int d,
                         const uint16_t* w1_hi, const uint16_t* w1_lo, const float* b1, const float* w2, const float* b2,
                         int H, int prec, const float* mask, float* logit_part, unsigned int* counters, int n_counters,
                         float* w, float* logits, void* stream);

/* Two nr_token_weights_fwd calls of one precision in ONE launch (round 4: the step's text and video tokens, from a few
 * workgroups per CU on: configs[2] / [3]; smaller sets are faster as two launches).  Fields as the arguments of
 * nr_token_weights_fwd; the two problems need separate `counters`.  NR_EUNSUPPORTED when the two do not run the same block
 * shape (the caller issues them one by one then); results are bit-identical to the single calls.                        */
typedef struct NrTokenWeightsProblem {
    const uint16_t *tok_hi, *tok_lo;
    const float* norm;
    const uint16_t *w1_hi, *w1_lo;
    const float *b1, *w2, *b2, *mask;
    float* logit_part;
    unsigned int* counters;
    float *w, *logits;
    int32_t n_samples, N, d, H, n_counters, reserved;
} NrTokenWeightsProblem;
int nr_token_weights_fwd_pair(const NrTokenWeightsProblem* a, const NrTokenWeightsProblem* b, int prec, void* stream);
/* Up to four nr_token_weights_fwd calls -- the four token sets a step scores: batch text / video (modeling.py:485-492 inside
 * local_level :283-287) and bank video / text (the two bank calls, until_module.py:170-185) -- each in its own precision
 * (precs[i]: NR_PREC_*), in ONE launch of 192 x 256 blocks; split-bf16 sets as three accumulated passes.  NR_EUNSUPPORTED
 * when a set does not fit that block (192 % N, H % 256): issue the sets one by one then.  One-pass sets: bit-identical to
 * their single launch; split-bf16 sets: another summation order (~1e-7 relative).                                          */
int nr_token_weights_fwd_group(const NrTokenWeightsProblem* probs, const int* precs, int n, void* stream);

/* Fused local_level (modeling.py:499-512): token-token cosine products on MFMA, max-pool over
 * each token axis, weighted sums, (t2v+v2t)/2.  The [A,Bv,Nt,Nv] tensor is never materialised.
 *   t_hi/lo [A*Nt,d], v_hi/lo [Bv*Nv,d] prepared tokens; w_t [A*Nt], w_v [Bv*Nv] token weights;
 *   out per out_mode (see NR_OUT_*).  Optional outputs kept for the backward pass (all or none):
 *   arg_v [A,Bv,Nt] / arg_t [A,Bv,Nv] u8 arg-max indices, pmax [A,Bv,Nt] / qmax [A,Bv,Nv] f32 the
 *   pooled maxima themselves.
 * nr_local_level_tiles reports the tile grid the kernel will use for that shape and precision (sizes of
 * the partial outputs).                                                                            */
int nr_local_level_tiles(int A, int Nt, int Bv, int Nv, int prec, int* n_row_tiles, int* n_col_tiles);
int nr_local_level_fwd(const uint16_t* t_hi, const uint16_t* t_lo, const uint16_t* v_hi, const uint16_t* v_lo,
                       const float* w_t, const float* w_v, int A, int Nt, int Bv, int Nv, int d,
                       int prec, int out_mode, float* out, uint8_t* arg_v, uint8_t* arg_t,
                       float* pmax, float* qmax, void* stream);

/* Several fused local_level products of ONE step in a single grid (modeling.py:499-512 is called three times per step:
 * batch x batch, :283-287, and the two memory-bank products of until_module.py:170-185): every product as
 * nr_local_level_fwd would compute it, without arg-max outputs, outputs in the same layout.  A CU picks up the next
 * product's block when its previous block retires instead of paying one dispatch + drain per product.
 * nr_local_level_group_kind: >= 0 if a product can join a group (24 x 12 tokens at sizes where the single launch runs the
 * 192 x 384 bf16 or the 96 x 192 split-bf16 blocks), -1 otherwise; nr_local_level_group returns NR_EUNSUPPORTED then
 * (nothing launched: call nr_local_level_fwd per product).  At most 4 products.                                        */
typedef struct NrLocalLevelProblem {
    const uint16_t *t_hi, *t_lo, *v_hi, *v_lo;   /* prepared tokens, as for nr_local_level_fwd (lo: split-bf16 only) */
    const float *w_t, *w_v;                      /* token weights [A*Nt], [Bv*Nv]                                   */
    float* out;                                  /* per out_mode                                                     */
    int A, Nt, Bv, Nv, d, prec, out_mode;
} NrLocalLevelProblem;
int nr_local_level_group_kind(int A, int Nt, int Bv, int Nv, int d, int prec);
int nr_local_level_group(int n, const NrLocalLevelProblem* problems, void* stream);

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

/* Same weights for both modalities in one launch, from the finished token means
 * (mean = nr_reduce_parts(colsum_part, 1/n_tok)).  g_text [B, n_g_text, d], g_video [B, n_g_video, d]:
 *   w[i] = mean_g exp(scale * <g_ig/||g_ig||, mean>).
 * n_g = 1 is the reference's expression (modeling.py:403-430).  n_g > 1 (ActivityNet token counts) has no reference
 * answer -- its loss fails to broadcast [B] * [B,n_g] (until_module.py:321); the mean over the global tokens is this
 * build's documented reduction (config.centrality_multi_token = "mean").
 * gnorm_* [B*n_g] and wtok_* [B*n_g] (per-token norms / weights, kept for backward) out or NULL.     */
int nr_centrality_weights_pair(const float* g_text, const float* g_video, int B, int n_g_text, int n_g_video, int d,
                               const float* mean_text, const float* mean_video, float scale, float* w_text,
                               float* w_video, float* gnorm_text, float* gnorm_video, float* wtok_text,
                               float* wtok_video, void* stream);

/* DPC-KNN cluster assignment of every token (cluster.py:453-509; index-only, no gradient):
 *   x [n_samples,N,C] f32 tokens; mask [n_samples,N] f32 (>0 = valid) or NULL; noise [n_samples,N]
 *   f32 in [0,1) (the reference's torch.rand tie-break draw, cluster.py:483); k nearest neighbours for
 *   the density; cluster_num centres; assign [n_samples,N] i64 out (cluster id in [0,cluster_num)).
 *   N <= 64.  workspace: nr_dpc_workspace_bytes(n_samples, N).                                  */
size_t nr_dpc_workspace_bytes(int n_samples, int N);
int nr_dpc_knn_assign(const float* x, const float* mask, const float* noise, int n_samples, int N, int C, int k,
                      int cluster_num, int64_t* assign, void* workspace, void* stream);

/* Forward pieces of one CTM + TCBlock stage (cluster.py:689-717, :834-888, :938-965) between the
 * library GEMMs (no gradient: the training path keeps these ops on autograd):
 *   nr_shift_concat     x [n_samples,N,C] -> out [n_samples*N, 3C] = (x[n-1] | x[n] | x[n+1]), zero at
 *                       the sample borders: the k=3, padding=1 token convolution becomes one GEMM.
 *   nr_ctm_norm_score   y [n_rows,C] (= x + conv) -> xn = LayerNorm(y), score = xn.sc_w + sc_b (masked
 *                       rows -> -inf, cluster.py:703-705), tokw = exp(score), kvn = norm1(xn).
 *   nr_merge_ln         merge_tokens (cluster.py:512-561): merged [n_samples,cnum,C] weighted cluster
 *                       means, merged_pb = merged + proj_b, qn = norm1(merged).
 *   nr_tc_attention     out [n_samples,cnum,C] = softmax_n(q.k/8 + score[n]) v over H heads of 64;
 *                       q [n_samples,cnum,C], kv [n_samples,N,2C] (k | v), score [n_samples,N].     */
int nr_shift_concat(const float* x, int n_samples, int N, int C, float* out, void* stream);
int nr_ctm_norm_score(const float* y, const float* mask, int n_rows, int C, const float* ln_w, const float* ln_b,
                      const float* sc_w, const float* sc_b, const float* n1_w, const float* n1_b, float eps,
                      float* xn, float* kvn, float* score, float* tokw, void* stream);
int nr_merge_ln(const float* xn, const int64_t* assign, const float* tokw, int n_samples, int N, int C, int cnum,
                const float* n1_w, const float* n1_b, const float* proj_b, float eps, float* merged,
                float* merged_pb, float* qn, void* stream);
int nr_tc_attention(const float* q, const float* kv, const float* score, int n_samples, int N, int C, int cnum,
                    int H, float* out, void* stream);

/* Two-launch form of the same stage pieces, one workgroup per sample (what the step actually runs):
 *   nr_ctm_front = nr_ctm_norm_score + the distance half of nr_dpc_knn_assign
 *                  (dist [n_samples,N,N], smax [n_samples] out); norm1(xn) goes either to kvn (f32) or,
 *                  when kvn_hi/kvn_lo are given, straight to the split-bf16 operand of nr_linear_x3;
 *   nr_ctm_back  = the assignment half of nr_dpc_knn_assign + nr_merge_ln (assign may be NULL).   */
int nr_ctm_front(const float* y, const float* mask, int n_samples, int N, int C, const float* ln_w,
                 const float* ln_b, const float* sc_w, const float* sc_b, const float* n1_w, const float* n1_b,
                 float eps, float* xn, float* kvn, uint16_t* kvn_hi, uint16_t* kvn_lo, float* score, float* tokw,
                 float* dist, float* smax, void* stream);
int nr_ctm_back(const float* dist, const float* smax, const float* mask, const float* noise, const float* xn,
                const float* tokw, int n_samples, int N, int C, int k, int cluster_num, const float* n1_w,
                const float* n1_b, const float* proj_b, float eps, float* merged, float* merged_pb, float* qn,
                int64_t* assign, void* stream);

/* ---- one whole clustering stage for a group of problems ------------------------------------------------
 * The step's form of the stage: CTM.forward (cluster.py:689-717) + TCBlock.forward (:938-965) of up to
 * NR_CTM_MAX_GROUP independent problems -- the text and the video tokens of modeling.py:446-481 -- in
 * seven launches that each carry the workgroups of every problem (conv GEMM, front, back, q+kv GEMMs,
 * attention, proj GEMM, preceded by the bf16 split of the token rows; the k=3 convolution reads the neighbour rows in place).  All GEMMs run split-bf16 (see nr_linear_x3); the
 * w*_hi/lo operands are the bf16 pairs (nr_split_bf16) of: the conv kernel as a [C, 3C] matrix with the
 * three taps side by side (tap k multiplies x[n+k-1]), q.weight [C,C], kv.weight [2C,C], proj.weight [C,C].
 * mask may be NULL (stage 1); conv_bias / q_bias / kv_bias may be NULL; assign (int64 [n_samples,N]) may
 * be NULL.  workspace: nr_ctm_stage_workspace_bytes(n_samples, N, C, cnum) bytes, 256-byte aligned,
 * private to the problem.  out [n_samples, cnum, C] f32.  N <= 64, C == 64*heads <= 1024.              */
#define NR_CTM_MAX_GROUP 4
typedef struct NrCtmStageDesc {
    int32_t n_samples, N, C, k, cnum, heads;
    float eps_ctm, eps_n1;
    const float* x;
    const float* mask;
    const float* noise;
    const uint16_t *wconv_hi, *wconv_lo;
    const float* conv_bias;
    const float *ln_w, *ln_b, *sc_w, *sc_b, *n1_w, *n1_b;
    const uint16_t *wq_hi, *wq_lo;
    const float* q_bias;
    const uint16_t *wkv_hi, *wkv_lo;
    const float* kv_bias;
    const uint16_t *wp_hi, *wp_lo;
    const float* proj_bias;
    void* workspace;
    float* out;
    int64_t* assign;
    /* optional (NULL): x as a bf16 pair [n_samples*N, C] (nr_split_bf16 of x) -- the stage then skips its own split launch;
     * out_hi / out_lo [n_samples*cnum, C]: `out` also written as a bf16 pair by the proj GEMM (chain two stages with them). */
    const uint16_t *x_hi, *x_lo;
    uint16_t *out_hi, *out_lo;
} NrCtmStageDesc;
size_t nr_ctm_stage_workspace_bytes(int n_samples, int N, int C, int cluster_num);
/* What the stage leaves in its workspace for a backward pass (neighborretr_amd/cluster_backward.py): byte offsets of
 *   [0] y [n,N,C] conv output + residual (LayerNorm input)   [1] xn [n,N,C] LayerNorm output   [2] score [n,N] (-inf on
 *   masked tokens)   [3] tokw [n,N] = exp(score)   [4] merged_pb [n,cnum,C] cluster means + proj bias   [5] q [n*cnum,C]
 *   [6] kv [n*N,2C]   [7] smax [n]: per-sample maximum pairwise distance, written by the front launch (launch 2 of
 *   nr_ctm_stage_fwd_range) and max-reduced over the samples by the back launch (3): a caller that clusters only a shard of
 *   the batch stores the maximum over ALL shards in smax[0] between the two (cluster.py:473-475 uses the batch-wide
 *   maximum).  All fp32.  offsets: 8 entries.                                                                          */
int nr_ctm_stage_workspace_layout(int n_samples, int N, int C, int cluster_num, size_t* offsets);
int nr_ctm_stage_fwd(const NrCtmStageDesc* problems, int n_problems, void* stream);
/* Launches [first, last) of the stage's NR_CTM_STAGE_LAUNCHES only (0 shift|split, 1 conv GEMM, 2 front, 3 back,
 * 4 q+kv GEMMs, 5 attention, 6 proj GEMM): a host that captures the step into a HIP graph interleaves them with
 * the launches of an independent branch, because the graph starts its nodes in capture order.            */
#define NR_CTM_STAGE_LAUNCHES 7
int nr_ctm_stage_fwd_range(const NrCtmStageDesc* problems, int n_problems, int first, int last, void* stream);

/* As nr_ctm_stage_workspace_layout, plus the split-bf16 operands the forward GEMMs read and the backward's weight-gradient
 * GEMMs need again: [8] kvn_hi [9] kvn_lo  [n*N, C]  norm1(xn);  [10] qn_hi [11] qn_lo  [n*cnum, C]  norm1(merged);
 * [12] att_hi [13] att_lo  [n*cnum, C]  attention output (input of proj).  offsets: 14 entries.                          */
int nr_ctm_stage_workspace_layout2(int n_samples, int N, int C, int cluster_num, size_t* offsets);

/* ---- backward of the clustering stage (no counterpart in the reference: PyTorch autograd differentiates cluster.py:453-561,
 * 689-717, 834-888 op by op; here the hand-derived backward of neighborretr_amd/cluster_backward.py runs as grouped kernels)
 *
 * nr_split_group: up to NR_SPLIT_MAX matrices in ONE launch.  mode 0: f32 src [rows, cols] -> bf16 hi/lo [rows, ld]
 * (ld >= cols, same 64-column tile count; lo may be NULL); mode 1: the same split written TRANSPOSED [cols, ld], ld >= rows,
 * entries [rows, min(ld, rows rounded up to 64)) zero (K padding of a GEMM operand; hi / lo may point INTO a wider buffer of
 * pitch ld at a 64-aligned column); mode 2: a bf16 PAIR src (hi) / src2 (lo) [rows, cols] -> transposed
 * [cols, ld]; mode 3: f32 src [rows, cols] = token rows in samples of `group` tokens -> the transposed k=3 neighbourhood
 * [3 cols, ld]: row 3 c + s holds src[r + s - 1, c] at column r (zero where r + s - 1 leaves the sample of r, and in the K
 * padding) -- the operand that makes the token-convolution weight gradient (cluster.py:664) come out of its GEMM in the
 * parameter's own [C_out, C_in, 3] order; modes 4 / 5: the two matrix forms of a k=3 convolution kernel W [C_out, C_in, 3] read
 * in place, row-major like mode 0 -- 4: dst[o, s C_in + i] = W[o, i, s] (rows C_out, cols 3 C_in, group C_in), 5: dst[i, s C_out + o]
 * = W[o, i, s] (rows C_in, cols 3 C_out, group C_out), 6: as 5 with the taps reversed, dst[i, s C_out + o] = W[o, i, 2 - s] (the
 * kernel of the TRANSPOSED convolution in nr_linear_group's in-place form).  Used for the per-step re-split of every weight matrix and for the K = token-rows
 * operands of the weight-gradient GEMMs.                                                                                   */
#define NR_SPLIT_MAX 48
typedef struct NrSplitItem {
    const void* src;
    const void* src2;
    uint16_t *hi, *lo;
    int32_t rows, cols, mode, ld;
    int32_t group, pad_;                         /* mode 3: tokens per sample; modes 4 / 5: C_in / C_out */
} NrSplitItem;                                   /* 56 bytes */
int nr_split_group(int n, const NrSplitItem* items, void* stream);

/* nr_colsum_group: dst[c] = scale * sum_r src[r, c] for up to NR_COLSUM_MAX f32 matrices in one launch (bias gradients, sums of
 * per-sample partial parameter gradients); fixed summation order.                                                           */
#define NR_COLSUM_MAX 16
typedef struct NrColsumItem {
    const float* src;
    float* dst;
    int32_t rows, cols;
    float scale;       /* dst = scale * column sums */
    int32_t pad_;
} NrColsumItem;
int nr_colsum_group(int n, const NrColsumItem* items, void* stream);

/* nr_linear_group: up to 8 independent problems Y = X W^T (+bias) (+residual) of nr_linear_x3's kind in one launch (every
 * problem tiled with the same block shape; K % 64 == 0 each).
 *   x_lo == w_lo == NULL in EVERY problem: the one-pass bf16 product of the hi halves (a third of the matrix-core work).
 *   ld > 0: X and W are K-slices of wider row-major matrices with row pitch ld elements (ld >= K, ld % 8 == 0) -- how a
 *   weight-gradient GEMM with a long K (token rows) is cut into several problems whose outputs are added afterwards
 *   (nr_slab_sum_group).  ld == 0: the rows are K long.                                                                     */
typedef struct NrLinearProblem {
    const uint16_t *x_hi, *x_lo, *w_hi, *w_lo;
    const float *bias, *residual;
    float* out;
    int32_t M, N, K, ld;
    int32_t conv_n, pad_;                        /* conv_n > 0: X is the token matrix [M, K/3] of samples of conv_n rows and the
                                                    product the k=3 token convolution read in place: sum_s X[r+s-1] W[:, s K/3..]
                                                    (rows outside r's sample read as zeros); every problem of the launch then is */
} NrLinearProblem;                               /* 80 bytes */
int nr_linear_group(int n, const NrLinearProblem* problems, void* stream);

/* Score-biased attention backward (cluster.py:868-885) per sample for up to NR_CTM_MAX_GROUP problems: from q [n*cnum,C],
 * kv [n*N,2C] (k | v), score [n,N] and d_att [n*cnum,C] (gradient of the attention output, = upstream x proj.weight) to
 * d_q [n*cnum,C], d_kv [n*N,2C] (both also as bf16 pairs, operands of the next GEMMs) and d_score [n,N] (the gradient
 * reaching the token scores through the attention bias).  N <= 64, C == 64 * heads, heads <= 16.                           */
typedef struct NrCtmAttnBwdDesc {
    int32_t n_samples, N, C, cnum, heads;
    const float *q, *kv, *score, *d_att;
    float *d_q, *d_kv, *d_score;
    uint16_t *dq_hi, *dq_lo, *dkv_hi, *dkv_lo;
} NrCtmAttnBwdDesc;
int nr_ctm_attn_bwd(int n, const NrCtmAttnBwdDesc* problems, void* stream);

/* The middle of the stage, backward, per sample: norm1 backward of the merged rows (d_qn) and of the token rows (d_kvn),
 * the block's residual (g), weighted cluster means (cluster.py:536-556; cluster ids `assign` carry no gradient), score / exp
 * (masked tokens: zero), LayerNorm(ctm) backward.  Writes d_y [n*N,C] (gradient of the conv output), the same as a bf16 pair
 * dy_hi / dy_lo [n*N,C] -- the A operand of the transposed token convolution, which nr_linear_group reads in place
 * (conv_n) -- and partial [n,6,C]: per-sample sums of d norm1.weight, d norm1.bias, d ctm.norm.weight, d ctm.norm.bias,
 * d score.weight, and d score.bias in [.,5,0].  merged_pb = cluster means + proj bias (what the forward keeps).            */
typedef struct NrCtmMidBwdDesc {
    int32_t n_samples, N, C, cnum;
    float eps_ctm, eps_n1;
    const float *d_qn, *d_kvn, *g, *merged_pb, *proj_b, *xn, *y, *tokw, *d_score, *mask, *n1_w, *ln_w, *sc_w;
    const int64_t* assign;
    float* d_y;
    uint16_t *dy_hi, *dy_lo;
    float* partial;
} NrCtmMidBwdDesc;
int nr_ctm_mid_bwd(int n, const NrCtmMidBwdDesc* problems, void* stream);

/* Y[M,N] = X[M,K] W[N,K]^T (+ bias[N]) (+ residual[M,N]) in split-bf16 on the MFMA tile engine: the big
 * fp32 GEMMs of the clustering stage (token convolution cluster.py:664, kv projection :866).
 * x_hi/x_lo [M,K], w_hi/w_lo [N,K] bf16 pairs; out [M,N] f32; K % 64 == 0.
 * nr_shift_concat_split = nr_shift_concat writing the bf16 pair directly.                        */
int nr_linear_x3(const uint16_t* x_hi, const uint16_t* x_lo, const uint16_t* w_hi, const uint16_t* w_lo,
                 const float* bias, const float* residual, int M, int N, int K, float* out, void* stream);
int nr_shift_concat_split(const float* x, int n_samples, int N, int C, uint16_t* hi, uint16_t* lo, void* stream);

/* Log-domain Sinkhorn targets, both directions in one launch (until_module.py:235-266):
 *   tgt_rows = beta*Q(G) + (1-beta)*I,  tgt_cols = beta*Q(G^T) + (1-beta)*I  (each [B,B],
 *   tgt_cols indexed in the transposed frame).  workspace: nr_sinkhorn_workspace_bytes(B).    */
size_t nr_sinkhorn_workspace_bytes(int B);
/* 128 < B <= 1024, B % 64 == 0: nr_sinkhorn_targets runs ONE launch whose B/32 workgroups per direction meet at counter
 * barriers, which needs them resident together.  nr_sinkhorn_cooperative_ok(B): 1 if the current device can hold them (kernel
 * occupancy x CUs of one XCD >= B/32) -- else, and for every other B > 128, the multi-launch form (2 iters + 3 launches) runs.
 * nr_sinkhorn_cooperative_gate: the same decision as a pure function (host-only; unit-tested without a GPU).
 * nr_sinkhorn_targets_multilaunch: the fallback form on its own (B > 128).  Should residency fail at run time all the same,
 * every spin is bounded and BOTH targets come back as NaN in full. */
int nr_sinkhorn_cooperative_ok(int B);
int nr_sinkhorn_cooperative_gate(int B, int blocks_per_cu, int n_cus, int n_xcd);
int nr_sinkhorn_targets_multilaunch(const float* G, int B, float beta, int iters, float* tgt_rows, float* tgt_cols,
                                    void* workspace, void* stream);
int nr_sinkhorn_targets(const float* G, int B, float beta, int iters, float* tgt_rows, float* tgt_cols,
                        void* workspace, void* stream);
/* The same solve emitting the uniform-regularisation ROW TERMS directly (until_module.py:285-289 on those targets):
 *   uniform_rows[dir][i] = -sum_j tgt_ij (T X_ij - LSE_j(T X_ij)),  X = G (dir 0) / G^T (dir 1)   (direction dir at
 * uniform_rows + dir * uniform_dir_stride);
 * write them as rows 1 of rowloss [2,4,B] (uniform_rows = rowloss + B, uniform_dir_stride = 4B) and run
 * nr_row_losses_fwd_no_uniform for the other terms BESIDE this launch.  tgt_rows / tgt_cols may both be NULL.
 * B <= 128, B % 4 == 0 (NR_EUNSUPPORTED otherwise: use nr_sinkhorn_targets + nr_row_losses_fwd).            */
int nr_sinkhorn_uniform_rows(const float* G, int B, float beta, int iters, float temperature, float* uniform_rows,
                             int uniform_dir_stride, float* tgt_rows, float* tgt_cols, void* workspace, void* stream);

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
/* The split tail of the loss-only step as two CONCURRENT launches that finalize themselves (B <= 128, B % 4 == 0):
 *   nr_sinkhorn_uniform_rows_final      Sinkhorn solve + uniform-CE row terms into rowloss[:,1,:]  (2 workgroups)
 *   nr_row_losses_fwd_no_uniform_final  centrality / neighbour / KL row terms into rowloss[:,(0,2,3),:], reading the
 *                                       bank centralities as the PARTIAL sums the fused local_level kernel writes
 *                                       (c0_parts [n_c0,B], c1_parts [n_c1,B], c_j = c_scale * sum_p part[p][j]:
 *                                       until_module.py:181 without a reduction launch)
 * Every workgroup of both launches adds to `counter` (one zero-initialised device word the caller keeps); the one that
 * arrives last -- nr_split_tail_workgroups(B) in all -- reduces rowloss [2,4,B] to losses[5] exactly as
 * nr_loss_finalize does and resets the counter.  Neither launch may be issued without the other.                    */
int nr_split_tail_workgroups(int B);
int nr_sinkhorn_uniform_rows_final(const float* G, int B, float beta, int iters, float temperature, float* rowloss,
                                   uint32_t* counter, float uniform_weight, float neighbor_weight, float kl_weight,
                                   float* losses, void* workspace, void* stream);
int nr_row_losses_fwd_no_uniform_final(const float* S, const float* G, const float* c0_parts, int n_c0,
                                       const float* c1_parts, int n_c1, float c_scale, const float* wc_text,
                                       const float* wc_video, const float* logit_scale, int B, int K, float temperature,
                                       float* rowloss, uint32_t* counter, float uniform_weight, float neighbor_weight,
                                       float kl_weight, float* losses, void* stream);
/* nr_row_losses_fwd_no_uniform_final computing the centrality weights itself (compute_centrality_weights, modeling.py:403-430,
 * ONE global token per sample): g_text / g_video [B,d] global tokens, mean_text / mean_video [d] means of the normalised
 * batch tokens, w_i = exp(centrality_scale * <g_i, mean> / max(|g_i|, 1e-12)) -- the arithmetic of
 * nr_centrality_weights_pair, whose launch this form saves in the loss-only step. */
int nr_row_losses_fwd_no_uniform_final_cw(const float* S, const float* G, const float* c0_parts, int n_c0,
                                          const float* c1_parts, int n_c1, float c_scale, const float* g_text,
                                          const float* g_video, const float* mean_text, const float* mean_video, int d,
                                          float centrality_scale, const float* logit_scale, int B, int K, float temperature,
                                          float* rowloss, uint32_t* counter, float uniform_weight, float neighbor_weight,
                                          float kl_weight, float* losses, void* stream);
/* nr_row_losses_fwd without the uniform term: rowloss[dir][0,2,3][i] only, no dependence on the Sinkhorn targets. */
int nr_row_losses_fwd_no_uniform(const float* S, const float* G, const float* bank_c0, const float* bank_c1,
                                 const float* wc_text, const float* wc_video, const float* logit_scale, int B, int K,
                                 float temperature, float* rowloss, void* stream);

/* Row-slab form of nr_row_losses_fwd for a loss sharded over ranks: the caller owns samples
 * [row0, row0 + n_rows) and holds S_rows = S[row0 : row0+n_rows, :] ([n_rows, B]) and S_cols = S[:, row0 : row0+n_rows]
 * ([B, n_rows]); G, the Sinkhorn targets, the bank centralities and the centrality weights are the full (replicated)
 * ones.  Only the owned rows of rowloss [2,4,B] are written: zero the buffer first, sum it over the ranks
 * (all-reduce), then nr_loss_finalize -- every rank then holds the same five losses.                    */
int nr_row_losses_fwd_slab(const float* S_rows, const float* S_cols, int row0, int n_rows, const float* G,
                           const float* tgt_rows, const float* tgt_cols, const float* bank_c0, const float* bank_c1,
                           const float* wc_text, const float* wc_video, const float* logit_scale, int B, int K,
                           float temperature, float* rowloss, void* stream);

/* Backward of nr_row_losses_fwd_slab (the sharded TRAINING loss, neighborretr_amd/sharded.py): for the rows a rank owns,
 * from upstream g_rowloss [2,4,B] (only the owned rows are read): dS_dir [2,n_rows,B] (direction 0: d S[row0+k, :],
 * direction 1: d S[:, row0+k] as a row), dG_dir [2,n_rows,B] likewise, d_c_rows [2,n_rows,B] (gradient of the bank
 * centrality vector a row's neighbour term read: direction 0 -> bank_c0, 1 -> bank_c1), d_wc / d_ls_rows [2,n_rows].       */
int nr_row_losses_bwd_slab(const float* S_rows, const float* S_cols, int row0, int n_rows, const float* G,
                           const float* tgt_rows, const float* tgt_cols, const float* bank_c0, const float* bank_c1,
                           const float* wc_text, const float* wc_video, const float* logit_scale, int B, int K,
                           float temperature, const float* g_rowloss, float* dS_dir, float* dG_dir, float* d_c_rows,
                           float* d_wc, float* d_ls_rows, void* stream);

/* nr_row_losses_fwd + nr_loss_finalize in one launch: the workgroup that finishes last reduces the row
 * terms (same arithmetic, bit-identical losses).  counter: one zero-initialised device word owned by the
 * caller; the kernel leaves it at zero again.                                                        */
int nr_row_losses_fwd_final(const float* S, const float* G, const float* tgt_rows, const float* tgt_cols,
                            const float* bank_c0, const float* bank_c1, const float* wc_text,
                            const float* wc_video, const float* logit_scale, int B, int K, float temperature,
                            float* rowloss, uint32_t* counter, float uniform_weight, float neighbor_weight,
                            float kl_weight, float* losses, void* stream);

/* Backward of nr_row_losses_fwd: recomputes the row statistics, then differentiates all four
 * terms.  g_rowloss [2,4,B] = d(objective)/d(rowloss) (for the fused objective: the constants of
 * nr_loss_finalize).  Outputs, all in the direction's own frame (direction 1 = transposed):
 *   dS_dir [2,B,B], dG_dir [2,B,B], d_c_rows [2,B,B] (row i's contribution to d bank_c[dir][j];
 *   the caller sums over i), d_wc [2,B], d_ls_rows [2,B] (summed by the caller).               */
int nr_row_losses_bwd(const float* S, const float* G, const float* tgt_rows, const float* tgt_cols,
                      const float* bank_c0, const float* bank_c1, const float* wc_text, const float* wc_video,
                      const float* logit_scale, int B, int K, float temperature, const float* g_rowloss,
                      float* dS_dir, float* dG_dir, float* d_c_rows, float* d_wc, float* d_ls_rows, void* stream);

/* g_rowloss [2,4,B] for nr_row_losses_bwd from the gradients of the five losses nr_loss_finalize returns (device scalars; NULL =
 * that loss has no gradient): coef[term] = (g_total * weight[term] + g_term) * 0.5 / B, the kl term divided by B once more.  One
 * launch in place of a dozen element-wise ones on 5-element tensors at the start of every backward pass.                    */
int nr_rowloss_coef(const float* g_total, const float* g_centrality, const float* g_uniform, const float* g_neighbor,
                    const float* g_kl, float uniform_weight, float neighbor_weight, float kl_weight, int B,
                    float* g_rowloss, void* stream);

/* out[i,j] = a[i,j] + b[j,i]  (folds the direction-1 gradients back into the row-major frame);
 * colsum variant: out[j] = sum_i a[i,j].                                                        */
int nr_add_transposed(const float* a, const float* b, int B, float* out, void* stream);
int nr_colsum(const float* a, int rows, int cols, float* out, void* stream);

/* Backward of nr_local_level_fwd through the stored arg-max indices (max-pool backward = route to
 * the arg-max) for ONE side of the product.
 *   side 0: gradients of the ROW (text) operand: d_x [A*Nt,d], d_w [A*Nt]; one workgroup per text,
 *           looping over the Bv videos.  side 1: the COLUMN (video) operand: d_x [Bv*Nv,d], d_w [Bv*Nv].
 *   dS: upstream gradient; ds_mode 0: dS[A,Bv] full, 1: dS[a,b] = vec[a] (row-mean modes),
 *       2: dS[a,b] = vec[b] (column-mean modes); `ds_scale` multiplies it (1/M for the means).
 *   o_hi/o_lo: prepared tokens of the OTHER operand (lo may be NULL: hi only);
 *   w_self / w_other: token weights of this / the other operand;
 *   d_x may be NULL (weights-only: memory-bank side, whose features get no gradient);
 *   accumulate != 0 adds into d_x / d_w instead of overwriting.
 *   workspace: nr_local_level_bwd_workspace_bytes(...) (chunk partials, reduced in fixed order). */
size_t nr_local_level_bwd_workspace_bytes(int side, int A, int Nt, int Bv, int Nv, int d);
int nr_local_level_bwd(int side, const float* dS, int ds_mode, float ds_scale,
                       const uint16_t* o_hi, const uint16_t* o_lo, const float* w_self, const float* w_other,
                       const uint8_t* arg_v, const uint8_t* arg_t, const float* pmax, const float* qmax,
                       int A, int Nt, int Bv, int Nv, int d, float* d_x, float* d_w, int accumulate,
                       void* workspace, void* stream);

/* The d_x half of nr_local_level_bwd as an MFMA GEMM: 96 x 96 blocks of the (sparse) routing matrix
 *   P[(s,n),(o,m)] = 0.5 dS(s,o) ( w_other[o,m] [scat(s,o,m) == n] + w_self[s,n] [gath(s,o,n) == m] )
 * are generated in LDS from the stored arg-max bytes and multiplied with the other operand's tokens.
 *   oT_hi / oT_lo: the other operand's prepared tokens in the order the kernel reads them (fragment-major, written by
 *   nr_sim_bwd_operand_group; ldk = the tokens they cover, whole slices of 96); oT_lo may be NULL (one pass).
 *   Token counts: multiples of 4 that divide 96 with at most 32 sample pairs per block (24 x 12, 12 x 24, 24 x 24,
 *   16 x 24, 24 x 16), d % 256 == 0 (nr_local_level_bwd_mfma_supported); d_w is not produced --
 *   nr_pool_weight_bwd_group computes it.  workspace: nr_local_level_bwd_mfma_workspace_bytes(...).      */
int nr_local_level_bwd_mfma_supported(int Nt, int Nv, int d);
size_t nr_local_level_bwd_mfma_workspace_bytes(int side, int A, int Nt, int Bv, int Nv, int d);
int nr_local_level_bwd_mfma(int side, const float* dS, int ds_mode, float ds_scale, const uint16_t* oT_hi,
                            const uint16_t* oT_lo, int ldk, const float* w_self, const float* w_other,
                            const uint8_t* arg_v, const uint8_t* arg_t, int A, int Nt, int Bv, int Nv, int d,
                            float* d_x, int accumulate, void* workspace, void* stream);

/* Prepared tokens [n_tok][d] (bf16 hi, optional lo; nr_prepare_tokens) -> fragment-major
 * [slice of 96 tokens][k-step of 32][d / 16][64 lanes][8]: element j of lane (kg, n) of block (slice, ks, dg) is token
 * 96 slice + 32 ks + 8 kg + j, dim 16 dg + n -- one wave load of the backward kernel is then one contiguous KiB.
 * out_hi / out_lo hold ceil(n_tok / 96) * 96 * d elements (tokens past n_tok are written as zeros); out_lo may be NULL.
 * d % 64 == 0, n <= 4 operands per launch.                                                                */
typedef struct NrSimBwdOperand {
    const uint16_t* hi;
    const uint16_t* lo;
    uint16_t* out_hi;
    uint16_t* out_lo;
    int32_t n_tok, d;
} NrSimBwdOperand;                               /* 40 bytes */
int nr_sim_bwd_operand_group(int n, const NrSimBwdOperand* items, void* stream);

/* Several of those products in ONE launch (+ one launch that sums the chunks): the loss step has four -- text and
 * video gradient of the batch x batch product (modeling.py:331) and of the two memory-bank products
 * (modeling.py:339-340).  Items with the same d_x are added together in item order (`accumulate` of the FIRST
 * item of a d_x says whether its previous content is kept).  All items share d and have oT_lo either all set
 * or all NULL.  n <= 4.  Fields as the arguments of nr_local_level_bwd_mfma.                            */
typedef struct NrSimBwdItem {
    const float* dS;
    const uint16_t* oT_hi;
    const uint16_t* oT_lo;
    const float* w_self;
    const float* w_other;
    const uint8_t* arg_v;
    const uint8_t* arg_t;
    float* d_x;
    float ds_scale;
    int32_t side, ds_mode, ldk, A, Nt, Bv, Nv, d, accumulate;
} NrSimBwdItem;                                  /* 104 bytes */
size_t nr_local_level_bwd_group_workspace_bytes(int n, const NrSimBwdItem* items);
int nr_local_level_bwd_group(int n, const NrSimBwdItem* items, void* workspace, size_t workspace_bytes, void* stream);

/* out[i] = (accumulate ? out[i] : 0) + sum over sources k, slabs c < n_slabs[k] of part[k][c * n + i], in that fixed order
 * (bitwise reproducible).  What nr_local_level_bwd_group ends with, exported for the weight-gradient GEMMs that are cut
 * along their long K (token rows) into several nr_linear_group problems.  n % 4 == 0; <= 4 outputs, <= 4 sources each.     */
typedef struct NrSlabSum {
    float* out;
    uint64_t n;
    int32_t accumulate, n_src;
    const float* part[4];
    int32_t n_slabs[4];
} NrSlabSum;                                     /* 72 bytes */
int nr_slab_sum_group(int n, const NrSlabSum* items, void* stream);

/* Gradient of the token weights of the fused product (modeling.py:505-512: the weighted sums of the pooled
 * maxima are linear in the weights):   d_w[s,n] = sum_o 0.5 * ds_scale * dS(s,o) * pooled[s,o,n]
 * summed over up to two products that use the same weights (batch x batch + one bank product).
 *   side 0: s = row sample a, pooled = pmax [A,Bv,N];  side 1: s = column sample b, pooled = qmax [A,Bv,N]
 *   (the arrays nr_local_level_fwd stores with want_arg).  dS / ds_mode / ds_scale as in nr_local_level_bwd.
 *   The sources of a job share the differentiated operand (side 0: same A; side 1: same Bv).  n <= 8 jobs,
 *   one launch, fixed summation order.                                                                */
typedef struct NrPoolWSrc {
    const float* dS;
    const float* pool;
    float ds_scale;
    int32_t ds_mode, A, Bv;
} NrPoolWSrc;                                    /* 32 bytes */
typedef struct NrPoolWJob {
    NrPoolWSrc src[2];
    float* d_w;
    int32_t n_src, side, N, accumulate;
} NrPoolWJob;                                    /* 88 bytes */
int nr_pool_weight_bwd_group(int n, const NrPoolWJob* jobs, void* stream);

/* Backward of F.normalize + mask + the centrality mean (nr_prepare_tokens):
 *   g = mask*d_xn + dmean/n_tok;  dx = (g - xhat <xhat,g>) / ||x||,  xhat = x/||x||.
 *   x [n_tok,d] original features, norm [n_tok], mask [n_tok] or NULL, d_xn [n_tok,d] or NULL,
 *   dmean [d] or NULL.                                                                          */
int nr_normalize_bwd(const float* x, const float* norm, const float* mask, const float* d_xn, const float* dmean,
                     int n_tok, int d, float* dx, void* stream);

/* The short head of the loss backward (everything between nr_row_losses_bwd and the two long chains) in three launches:
 *   nr_rowloss_bwd_finish: dS = dS_dir[0] + dS_dir[1]^T, dG likewise, d_c0 / d_c1 [B] = column sums of d_c_rows[0] / [1],
 *     d_ls [1] = sum of d_ls_rows [2,B]  (fixed order);
 *   nr_centrality_weights_bwd_pair: nr_centrality_weights_bwd for the text and the video global tokens together;
 *   nr_global_logits_bwd: gradient of G = gt gv^T (one global token per sample, modeling.py:333) with the centrality part added:
 *     d_gt [B,d] = dG gv + add_t,  d_gv [B,d] = dG^T gt + add_v   (fp32).                                                  */
int nr_rowloss_bwd_finish(const float* dS_dir, const float* dG_dir, const float* d_c_rows, const float* d_ls_rows, int B,
                          float* dS, float* dG, float* d_c0, float* d_c1, float* d_ls, void* stream);
int nr_centrality_weights_bwd_pair(const float* g_t, const float* gnorm_t, const float* mean_t, const float* w_t, const float* dw_t,
                                   const float* g_v, const float* gnorm_v, const float* mean_v, const float* w_v, const float* dw_v,
                                   int B, int d, float scale, float* dg_t, float* dmean_t, float* dg_v, float* dmean_v, void* stream);
int nr_global_logits_bwd(const float* dG, const float* gt, const float* gv, const float* add_t, const float* add_v, int B, int d,
                         float* d_gt, float* d_gv, void* stream);

/* Backward of nr_token_softmax: dlogit = w * (dw - sum_t w dw) per sample ([n_samples,N]). */
int nr_token_softmax_bwd(const float* w, const float* dw, int n_samples, int N, float* dlogit, void* stream);

/* Backward of nr_centrality_weights:  a_i = dw_i * w_i * scale;
 *   dg_i = a_i * (mean - ghat_i <ghat_i, mean>) / ||g_i||;   dmean = sum_i a_i ghat_i.           */
int nr_centrality_weights_bwd(const float* g, const float* gnorm, const float* mean, const float* w, const float* dw,
                              int B, int d, float scale, float* dg, float* dmean, void* stream);

/* The exchange step's packing (modeling.py:274-280: five all_gathers -> one).  nr_pack_shard copies n <= 8
 * device buffers (srcs / bytes / offsets are HOST arrays) to their offsets inside one packed record;
 * nr_unpack_gathered scatters the [world, record_bytes] all-gather result into n rank-major outputs
 * (dst_k[w*bytes_k + i] = rec_w[offset_k + i]); where u8_to_f32[k] != 0 the piece is bytes_k uint8 values per
 * rank written as fp32 (the masks become the multipliers the kernels read).  One launch each.        */
int nr_pack_shard(int n, const void* const* srcs, const size_t* bytes, const size_t* offsets, void* packed, void* stream);
/* nr_pack_shard with the masks' dtype conversion folded in (the reference gathers its int64 masks as they are,
 * modeling.py:277-278; this build's record carries them as u8): kinds[k] = 0 raw bytes, 1 = int64 elements -> u8,
 * 2 = fp32 elements -> u8 (C truncation, as torch's .to(uint8)); for a converted piece bytes[k] is its ELEMENT count (= the
 * bytes it takes in the record).  kinds == NULL: all raw.  One launch instead of two element-wise ones and the pack. */
int nr_pack_shard_convert(int n, const void* const* srcs, const size_t* bytes, const size_t* offsets, const int* kinds,
                          void* packed, void* stream);
int nr_unpack_gathered(int n, const void* gathered, int world, size_t record_bytes, const size_t* bytes,
                       const size_t* offsets, void* const* dsts, const int* u8_to_f32, void* stream);
/* n <= 12 device-to-device copies (srcs[k] -> dsts[k], bytes[k] each) in one launch.  No counterpart in the reference: the bank
 * copy in front of an overlapped owned step of the step-interleaved job (the prepared shadow, masks and noise counter as the
 * step's loss must see them: neighborretr_amd/modeling.py OwnedSlot). */
int nr_copy_group(int n, const void* const* srcs, void* const* dsts, const size_t* bytes, void* stream);

/* The whole exchange step in one call (SURVEY.md 8b minimum set): nr_pack_shard into `packed` [record_bytes], ONE RCCL
 * all-gather of record_bytes uint8 per rank over xGMI on the caller's communicator (`nccl_comm` = its ncclComm_t),
 * nr_unpack_gathered from `gathered` [world * record_bytes] into the rank-major outputs -- all on `stream`.  RCCL is
 * resolved at run time (the librccl.so.1 already loaded in the process, else the system one): NR_EUNSUPPORTED if there is
 * none, 1000 + ncclResult_t if the collective fails.  The Python host reaches the same collective through
 * torch.distributed (neighborretr_amd/dist.py), whose communicator is not exposed as a raw handle.                      */
int nr_allgather_packed(void* nccl_comm, int world, int n, const void* const* srcs, const size_t* bytes,
                        const size_t* offsets, size_t record_bytes, void* packed, void* gathered, void* const* dsts,
                        const int* u8_to_f32, void* stream);

/* Front of the step in one launch (any part may be switched off with n = 0 / NULL):
 *   out0[i] = (float)mask0[i], out1[i] = (float)mask1[i]   the loader's int64 masks as fp32 multipliers;
 *   logit_scale_exp[0] = exp(logit_scale[0])                modeling.py:289;
 *   noise[0..n_noise) uniform in [0,1)                      the torch.rand draws of cluster.py:483, from a
 *       counter-based generator: rng_state = device uint64[2] {seed, counter}; the kernel advances the
 *       counter, so replays of a captured graph draw fresh numbers;
 *   ring_head (device int32, optional): *ring_head = (*ring_head - ring_advance) mod ring_capacity -- the
 *       memory bank's ring head moves back by the batch this step will push (nr_bank_ring_push reads it).  */
int nr_step_prologue(const int64_t* mask0, int n0, float* out0, const int64_t* mask1, int n1, float* out1,
                     const float* logit_scale, float* logit_scale_exp, uint64_t* rng_state, float* noise,
                     int n_noise, int32_t* ring_head, int ring_advance, int ring_capacity, void* stream);

/* Memory-bank FIFO push (modeling.py:237-249): bank <- cat(batch, bank)[:capacity] done as an
 * in-place shift; rows are `row_bytes` wide.  Requires 0 < n_new; if n_new >= capacity the bank
 * becomes the first `capacity` rows of the batch.  scratch: capacity*row_bytes bytes.          */
int nr_bank_push(void* bank, const void* batch, int capacity, int n_new, size_t row_bytes, void* scratch,
                 void* stream);

/* The same FIFO kept as a ring (logical order L[i] = S[(head+i) mod capacity]): writes the n_new batch
 * rows of up to 12 tensors at rows [head_new, head_new+n_new) mod capacity in one launch.
 * banks / batches / row_bytes are HOST arrays of n_tensors device pointers / row sizes.  head_dev (device
 * int32, optional) overrides head_new: the head then lives on the device, advanced by nr_step_prologue, so that
 * a captured HIP graph pushes to a new place at every replay.                                      */
int nr_bank_ring_push(int n_tensors, void* const* banks, const void* const* batches, const size_t* row_bytes,
                      int capacity, int head_new, const int32_t* head_dev, int n_new, void* stream);

/* A gathered batch straight into the memory bank, ONE launch (step-interleaved multi-rank job: a step whose loss another rank
 * evaluates leaves nothing behind on this rank but the FIFO push of modeling.py:309-310 and the bank's prepared shadow).
 * Reads the RECEIVE buffer of the packed exchange step -- world records of record_bytes bytes, each holding per_rank samples'
 * text tokens f32 [per_rank, Nt, d] at off_text, video tokens at off_video, ids i64 at off_index, masks u8 [per_rank, N] at
 * off_text_mask / off_video_mask (nr_pack_shard's layout) -- and
 *   moves the ring head back by world * per_rank (as nr_step_prologue does) and writes the samples at the new head on:
 *   bank_text [capacity, Nt, d] / bank_video f32, bank_index i64 [capacity], bank_*_mask f32 [capacity, N];
 *   the same rows PREPARED (normalised x mask as bf16 hi / lo + norms: nr_prepare_tokens' arithmetic, bit for bit) into
 *   shadow_* (hi / lo [capacity * N, d], norm [capacity * N]; all six NULL: no shadow kept);
 *   rng_state (optional): the noise stream's step counter advances by one, as a step's nr_step_prologue advances it.
 * counter: nr_bank_absorb_counter_words() zeroed device words (a two-level ticket: one word per group of workgroups, 64 B
 *   apart, and the launch's own), all zero again when the launch ends.
 * world * per_rank >= capacity: the batch's first `capacity` samples become the bank and the head 0 (the reference's
 *   cat(batch, bank)[:capacity], modeling.py:244-249).  d % 256 == 0, d <= 1024. */
typedef struct NrBankAbsorbDesc {
    const void* gathered;
    uint64_t record_bytes, off_text, off_video, off_index, off_text_mask, off_video_mask;
    int32_t world, per_rank, Nt, Nv, d, capacity;
    float *bank_text, *bank_video, *bank_text_mask, *bank_video_mask;
    int64_t* bank_index;
    uint16_t *shadow_text_hi, *shadow_text_lo, *shadow_video_hi, *shadow_video_lo;
    float *shadow_text_norm, *shadow_video_norm;
    int32_t* ring_head;
    uint64_t* rng_state;
    uint32_t* counter;
} NrBankAbsorbDesc;
int nr_bank_absorb_counter_words(void);
int nr_bank_absorb_gathered(const NrBankAbsorbDesc* desc, void* stream);

/* Rank of the diagonal in every row under the reference's tie rule (metrics.py:58-66):
 *   greater[i] = #{j : S[i,j] > S[i,i]},  equal[i] = #{j : S[i,j] == S[i,i]} (includes j=i). */
int nr_diag_ranks(const float* S, int N, int32_t* greater, int32_t* equal, void* stream);

/* Sharded evaluation (evaluator.py:21-63 with the N x N matrix split into row slabs over the ranks; metrics.py:58-66):
 * rank counts from S_slab = S[row0 : row0 + n_rows, :] and the full diagonal diag [N] (gathered by the caller).
 *   greater_rows / equal_rows [n_rows]: text->video counts of the slab's rows (complete);
 *   greater_cols / equal_cols [N]: this slab's PARTIAL video->text counts per column (sum them over the ranks).      */
int nr_slab_ranks(const float* S_slab, int n_rows, int N, int row0, const float* diag, int32_t* greater_rows,
                  int32_t* equal_rows, int32_t* greater_cols, int32_t* equal_cols, void* stream);

/* Multi-sentence retrieval (several captions per video: evaluator.py:114-149,225-262; metrics.py:82-148) from a row slab
 * S_slab = S[row0 : row0 + n_rows, :] of the sentence x video matrix.  group_end [G] (device, int32, increasing): the
 * sentences of video g are the global rows [group_end[g-1], group_end[g]); the last entry is the sentence count and
 * row0 + n_rows must not exceed it (checked by the caller: the array lives on the device).
 *   greater_rows / equal_before_rows [n_rows]: #{j : S[i,j] > S[i,g(i)]} and #{j < g(i) : S[i,j] == S[i,g(i)]}; their sum
 *     is the text->video rank of sentence i (metrics.py:103-106 with a stable sort; NaN scores rank first, as torch.argsort
 *     puts them); greater = -1 marks a sentence whose own score is NaN / infinite: not ranked (metrics.py:108-111);
 *   group_max [G, V]: max over this slab's sentences of video g of S[., j], -inf where the slab has none (NaN scores
 *     ignored): MAX-reduce over the ranks, transpose, and nr_diag_ranks gives the video->text ranks (metrics.py:141-146). */
int nr_group_slab_ranks(const float* S_slab, int n_rows, int V, int row0, const int32_t* group_end, int G,
                        int32_t* greater_rows, int32_t* equal_before_rows, float* group_max, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NR_HIP_H */
