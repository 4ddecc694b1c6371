// Internal (not exported) interface of the split-bf16 linear kernels: one launch for a GROUP of
// independent problems Y_g = X_g W_g^T (+bias_g) (+residual_g) -- used by the clustering stage, which
// runs its text and video problems side by side (nr_ctm_group.hip).
#pragma once
#include "nr_common.h"

struct NrLinearArgs {
    const uint16_t *x_hi, *x_lo, *w_hi, *w_lo;
    const float *bias, *residual;
    float* out;
    int M, N, K;
    // conv_n > 0 (only in launches made with conv = true): X is the token matrix [M, K/3] and the product the k=3 token
    // convolution over samples of conv_n tokens, read in place (NrGemmTile::run_conv3)
    int conv_n = 0;
    // optional second form of the output: the bf16 pair of `out` (the A operand of a following GEMM)
    uint16_t *out_hi = nullptr, *out_lo = nullptr;
    // row pitch of X and W in elements when both are K-slices of wider matrices (0: rows are K long); not with conv
    int ld = 0;
};

#define NR_LINEAR_MAX_GROUP 8

// All problems of a group are tiled with the same tile shape; K % 64 == 0 for each.  x_lo == w_lo == nullptr in EVERY problem
// selects the one-pass bf16 product (hi halves only) instead of the three-pass split-bf16 one.
// conv = true: every problem is a token convolution (conv_n > 0 each).
int nr_linear_group_launch(const NrLinearArgs* probs, int n_probs, hipStream_t stream, bool conv = false);

// The same grouped launch with the clustering stage's DPC-KNN / merge workgroups ("back", nr_ctm_back_body) in FRONT of the GEMM
// tiles in one grid of 512-thread workgroups: the kv projection depends on the stage's front launch alone, like the back half,
// so the two run beside each other instead of one behind the other (the back half is a latency chain on a few waves per
// sample, the GEMM tiles are LDS-DMA bound: they share CUs well).  gb: the back problems (start[] = first workgroup of each,
// start[NR_CTM_MAX_GROUP] = their total); back_lds: dynamic LDS of a back workgroup when use_lds.  Tile shape fixed: 64 x 128
// on 8 waves, one-deep ring.  NR_EUNSUPPORTED when a problem does not fit that form (the caller then launches the two apart).
struct NrCtmBackArgs;
template <typename A> struct NrGroupOf;
// back_form: 0 = nr_ctm_back_body (first form); 4 / 16 = nr_ctm_back_body2 with accumulators for that many clusters (C = 512).
int nr_linear_group_launch_beside_back(const NrLinearArgs* probs, int n_probs, const NrGroupOf<NrCtmBackArgs>& gb, size_t back_lds,
                                       int use_lds, int back_form, hipStream_t stream);
