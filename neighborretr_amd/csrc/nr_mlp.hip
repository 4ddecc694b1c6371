// Token-weight scorer: Linear(d,H) + ReLU + Linear(H,1) as an MFMA GEMM with the second layer
// folded into the epilogue, then the masked softmax over each sample's tokens.
// Reference: modeling.py:148-153 (the MLP), :485-487 / :490-492 (mask -> -9e15, softmax).
//
// The GEMM consumes the NORMALISED bf16 tokens written by nr_prepare_tokens and rescales each
// accumulator row by the token's original norm (W1 x = ||x|| * W1 x_hat), so the features are
// converted to bf16 only once for both the scorer and the similarity kernel.  Rows of masked
// tokens are zero vectors; their logits are overwritten with -9e15 by the softmax anyway.
#include "nr_gemm_tile.h"
#include "../../include/nr_hip.h"

template <bool X3, int STAGES>
__global__ __launch_bounds__(256) void nr_mlp_kernel(const uint16_t* __restrict__ tok_hi, const uint16_t* __restrict__ tok_lo,
                                                     const float* __restrict__ norm, int n_tok, int d,
                                                     const uint16_t* __restrict__ w1_hi, const uint16_t* __restrict__ w1_lo,
                                                     const float* __restrict__ b1, const float* __restrict__ w2, int H,
                                                     float* __restrict__ logit_part) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using Tile = NrGemmTile<4, 4, X3, 16, 16, STAGES>;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int row0 = blockIdx.y * 128, col0 = blockIdx.x * 128;

    Tile tile;
    tile.zero();
    tile.run(tok_hi, tok_lo, row0, n_tok, w1_hi, w1_lo, col0, H, d, smem);

    float bb[4], ww[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        int c = col0 + wc * 64 + n * 16 + (lane & 15);
        bb[n] = b1[c];
        ww[n] = w2[c];
    }
    float* sPart = reinterpret_cast<float*>(smem);   // [2][128]; staging LDS is free after run()
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int r = wr * 64 + m * 16 + (lane >> 4) * 4 + j;
            int gr = min(row0 + r, n_tok - 1);
            float sc = norm[gr];
            float v = 0.f;
#pragma unroll
            for (int n = 0; n < 4; ++n) v += fmaxf(tile.acc[m][n][j] * sc + bb[n], 0.f) * ww[n];
            v += __shfl_xor(v, 1);
            v += __shfl_xor(v, 2);
            v += __shfl_xor(v, 4);
            v += __shfl_xor(v, 8);
            if ((lane & 15) == 0) sPart[wc * 128 + r] = v;
        }
    __syncthreads();
    if (tid < 128 && row0 + tid < n_tok)
        logit_part[(size_t)blockIdx.x * n_tok + row0 + tid] = sPart[tid] + sPart[128 + tid];
}

extern "C" int nr_token_logits_fwd(const uint16_t* tok_hi, const uint16_t* tok_lo, const float* norm, int n_tok, int d,
                                   const uint16_t* w1_hi, const uint16_t* w1_lo, const float* b1, const float* w2,
                                   int H, int prec, float* logit_part, void* stream) {
    if (!tok_hi || !norm || !w1_hi || !b1 || !w2 || !logit_part) return NR_EINVAL;
    if (n_tok <= 0 || d <= 0 || (d % 64) != 0 || H <= 0 || (H % 128) != 0) return NR_EINVAL;
    if (prec != NR_PREC_BF16 && prec != NR_PREC_BF16X3) return NR_EINVAL;
    if (prec == NR_PREC_BF16X3 && (!tok_lo || !w1_lo)) return NR_EINVAL;
    dim3 grid(H / 128, (n_tok + 127) / 128);
    hipStream_t st = (hipStream_t)stream;
    const bool x3 = prec == NR_PREC_BF16X3;
    const int stages = nr_pick_stages((long)grid.x * grid.y);
#define NR_MLP_CASE(X3_, ST_)                                                                                          \
    if (x3 == X3_ && stages == ST_) {                                                                                  \
        size_t lds = NrGemmTile<4, 4, X3_, 16, 16, ST_>::RING_BYTES;                                                   \
        if (lds < 1024) lds = 1024;                                                                                    \
        if (lds > 64 * 1024) {                                                                                         \
            hipError_t e = hipFuncSetAttribute((const void*)nr_mlp_kernel<X3_, ST_>,                                   \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                  \
            if (e != hipSuccess) return (int)e;                                                                        \
        }                                                                                                              \
        hipLaunchKernelGGL((nr_mlp_kernel<X3_, ST_>), grid, dim3(256), lds, st, tok_hi, tok_lo, norm, n_tok, d, w1_hi, \
                           w1_lo, b1, w2, H, logit_part);                                                              \
    }
    NR_MLP_CASE(true, 1) NR_MLP_CASE(true, 2) NR_MLP_CASE(false, 1) NR_MLP_CASE(false, 2)
#undef NR_MLP_CASE
    NR_LAUNCH_CHECK();
    return NR_OK;
}

// one wave per sample; N <= 256 tokens
__global__ __launch_bounds__(256) void nr_token_softmax_kernel(const float* __restrict__ logit_part, int n_parts,
                                                               const float* __restrict__ b2, const float* __restrict__ mask,
                                                               int n_samples, int N, float* __restrict__ w,
                                                               float* __restrict__ logits) {
    const int lane = threadIdx.x & 63;
    const int s = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s >= n_samples) return;
    const size_t n_tok = (size_t)n_samples * N;
    const float bias = b2[0];
    float x[4];
    float mx = -INFINITY;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        int t = e * 64 + lane;
        x[e] = -INFINITY;
        if (t < N) {
            size_t idx = (size_t)s * N + t;
            float v = bias;
            for (int p = 0; p < n_parts; ++p) v += logit_part[(size_t)p * n_tok + idx];
            if (logits) logits[idx] = v;
            if (mask && mask[idx] == 0.f) v = NR_NEG_BIG;     // masked_fill_(-9e15)
            x[e] = v;
            mx = fmaxf(mx, v);
        }
    }
    mx = nr_wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        x[e] = (e * 64 + lane < N) ? expf(x[e] - mx) : 0.f;
        sum += x[e];
    }
    sum = nr_wave_sum(sum);
    float inv = 1.0f / sum;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        int t = e * 64 + lane;
        if (t < N) w[(size_t)s * N + t] = x[e] * inv;
    }
}

extern "C" int nr_token_softmax(const float* logit_part, int n_parts, const float* b2, const float* mask, int n_samples,
                                int N, float* w, float* logits, void* stream) {
    if (!logit_part || !b2 || !w || n_parts <= 0 || n_samples <= 0 || N <= 0) return NR_EINVAL;
    if (N > 256) return NR_EUNSUPPORTED;
    hipLaunchKernelGGL(nr_token_softmax_kernel, dim3((n_samples + 3) / 4), dim3(256), 0, (hipStream_t)stream,
                       logit_part, n_parts, b2, mask, n_samples, N, w, logits);
    NR_LAUNCH_CHECK();
    return NR_OK;
}
