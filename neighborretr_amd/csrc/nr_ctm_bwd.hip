// Backward of one token-clustering stage (CTM + TCBlock; reference cluster.py:453-561, 689-717, 834-888, differentiated by
// hand in neighborretr_amd/cluster_backward.py, which is checked against autograd in fp64) as GROUPED HIP kernels: like the
// forward (nr_ctm_group.hip) every launch carries the workgroups of all problems of the group -- the text and the video
// tokens of a stage -- and the GEMMs run split-bf16 on the MFMA tile engine (nr_linear_group).  Per stage, for both
// modalities together:
//   nr_split_group      upstream gradient -> bf16 hi/lo (and every TRANSPOSED operand the weight-gradient GEMMs need)
//   nr_linear_group     d_att = g Wp
//   nr_ctm_attn_bwd     score-biased attention backward per sample: d_q, d_k | d_v, d_score      (cluster.py:868-885)
//   nr_linear_group     d_qn = d_q Wq,  d_kvn = d_kv Wkv
//   nr_ctm_mid_bwd      per sample: both LayerNorm(norm1) backwards, residual, weighted cluster means, score / exp,
//                       LayerNorm(ctm) backward -> d_y, written as the three shifted split-bf16 column blocks the
//                       transposed convolution reads, + per-sample partial sums of the LayerNorm / score parameter grads
//   nr_linear_group     d_x0 = d_y + [d_y[n+1] | d_y[n] | d_y[n-1]] Wbt                          (conv k=3 + residual)
//   nr_split_group      transposes of d_q, d_kv, d_y
//   nr_linear_group     dWproj, dWq, dWkv, dWconv  (K = token rows)
//   nr_colsum_group     bias gradients and the per-sample partial sums
// The host side (neighborretr_amd/cluster_backward_hip.py) strings these together; every entry point here is stateless.
#include "nr_ctm_bodies.h"
#include "nr_linear.h"
#include "../../include/nr_hip.h"

// ---- grouped (transpose-)split: fp32 [rows, cols] -> bf16 hi/lo, row-major [rows, ld] or transposed [cols, ld] -----------
struct NrSplitGroup {
    NrSplitItem it[NR_SPLIT_MAX];
    int start[NR_SPLIT_MAX + 1];     // first workgroup (64 x 64 tile) of every item
    int n;
};

// Four consecutive destination columns r .. r + 3 of one transposed row as ONE 8-byte store per half (the rows of the transposed
// operands are K-contiguous: 16 lanes x 8 B = a whole 128-byte line per row and wave store; the 2-byte stores this replaces put
// a quarter of the bytes into four times the store instructions).  Falls back to element stores at a ragged end.
__device__ __forceinline__ void nr_store_t4(uint16_t* __restrict__ hi, uint16_t* __restrict__ lo, size_t o, int r, int ld,
                                            const uint16_t (&h)[4], const uint16_t (&l)[4]) {
    if (r + 3 < ld && ((o | (size_t)ld) & 3) == 0 && ((reinterpret_cast<uintptr_t>(hi) | reinterpret_cast<uintptr_t>(lo)) & 7) == 0) {
        *reinterpret_cast<uint2*>(hi + o) = uint2{(uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16)};
        if (lo) *reinterpret_cast<uint2*>(lo + o) = uint2{(uint32_t)l[0] | ((uint32_t)l[1] << 16), (uint32_t)l[2] | ((uint32_t)l[3] << 16)};
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (r + j < ld) {
                hi[o + j] = h[j];
                if (lo) lo[o + j] = l[j];
            }
    }
}

__global__ __launch_bounds__(256) void nr_split_group_kernel(NrSplitGroup g) {
    __shared__ float smem3[66 * 65];
    float (*t)[65] = reinterpret_cast<float (*)[65]>(smem3);
    const int wg = blockIdx.x;
    int gi = 0;
    for (int i = 1; i < g.n; ++i)
        if (wg >= g.start[i]) gi = i;
    const NrSplitItem it = g.it[gi];
    const int tl = wg - g.start[gi];
    const int tcn = (it.cols + 63) / 64;
    const int tr = tl / tcn, tc = tl - tr * tcn;
    const int c = threadIdx.x & 63, r0 = threadIdx.x >> 6;
    const float* src = static_cast<const float*>(it.src);
    if (it.mode == 2) {                     // a bf16 pair, transposed: two u16 tiles through the same LDS tile
        const uint16_t* sh = static_cast<const uint16_t*>(it.src);
        const uint16_t* sl = static_cast<const uint16_t*>(it.src2);
        uint32_t (*tu)[65] = reinterpret_cast<uint32_t (*)[65]>(t);
#pragma unroll 4
        for (int i = 0; i < 16; ++i) {
            const int r = tr * 64 + r0 + 4 * i, cc = tc * 64 + c;
            const bool on = r < it.rows && cc < it.cols;
            const size_t o = (size_t)r * it.cols + cc;
            tu[r0 + 4 * i][c] = on ? ((uint32_t)sh[o] | ((uint32_t)(sl ? sl[o] : 0) << 16)) : 0u;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int cl = (threadIdx.x >> 4) + 16 * i, cc = tc * 64 + cl, rl = 4 * (threadIdx.x & 15), r = tr * 64 + rl;
            if (cc < it.cols && r < it.ld) {
                uint16_t h[4], l[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t v = tu[rl + j][cl];
                    h[j] = (uint16_t)(v & 0xFFFFu);
                    l[j] = (uint16_t)(v >> 16);
                }
                nr_store_t4(it.hi, it.lo, (size_t)cc * it.ld + r, r, it.ld, h, l);
            }
        }
        return;
    }
    if (it.mode == 3) {                     // transposed k=3 neighbourhood: tile rows with a one-row halo on both sides
        float (*th)[65] = reinterpret_cast<float (*)[65]>(smem3);
        for (int e = threadIdx.x; e < 66 * 64; e += 256) {
            const int k = e >> 6, cl = e & 63;
            const int r = tr * 64 - 1 + k, cc = tc * 64 + cl;
            th[k][cl] = (r >= 0 && r < it.rows && cc < it.cols) ? src[(size_t)r * it.cols + cc] : 0.f;
        }
        __syncthreads();
        const int rl = 4 * (threadIdx.x & 15), r = tr * 64 + rl;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int cl = (threadIdx.x >> 4) + 16 * i, cc = tc * 64 + cl;
            if (cc >= it.cols || r >= it.ld) continue;
#pragma unroll
            for (int sft = 0; sft < 3; ++sft) {
                uint16_t h[4], l[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int rj = r + j, rr = rj + sft - 1;
                    const bool on = rj < it.rows && rr >= 0 && rr < it.rows && rr / it.group == rj / it.group;
                    const float v = on ? th[rl + j + sft][cl] : 0.f;
                    h[j] = nr_f2bf(v);
                    l[j] = nr_f2bf(v - nr_bf2f(h[j]));
                }
                nr_store_t4(it.hi, it.lo, (size_t)(3 * cc + sft) * it.ld + r, r, it.ld, h, l);
            }
        }
        return;
    }
    if (it.mode == 0 || it.mode >= 4) {
        // modes 4 / 5: the two matrix forms of a k=3 convolution kernel W [C_out, C_in, 3] read in place --
        //   4: dst[o, s C_in + i] = W[o, i, s]   (group = C_in)        5: dst[i, s C_out + o] = W[o, i, s]   (group = C_out)
        //   6: dst[i, s C_out + o] = W[o, i, 2 - s]: the TRANSPOSED convolution's kernel for the in-place form (tap s meets row n+s-1)
#pragma unroll 4
        for (int i = 0; i < 16; ++i) {
            const int r = tr * 64 + r0 + 4 * i, cc = tc * 64 + c;
            if (r < it.rows && cc < it.ld) {
                size_t o = (size_t)r * it.cols + cc;
                if (it.mode == 4) o = (size_t)r * it.cols + (size_t)(cc % it.group) * 3 + cc / it.group;
                else if (it.mode == 5) o = (size_t)(cc % it.group) * (3 * (size_t)it.rows) + (size_t)r * 3 + cc / it.group;
                else if (it.mode == 6) o = (size_t)(cc % it.group) * (3 * (size_t)it.rows) + (size_t)r * 3 + (2 - cc / it.group);
                const float v = cc < it.cols ? src[o] : 0.f;
                const uint16_t h = nr_f2bf(v);
                it.hi[(size_t)r * it.ld + cc] = h;
                if (it.lo) it.lo[(size_t)r * it.ld + cc] = nr_f2bf(v - nr_bf2f(h));
            }
        }
        return;
    }
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
        const int r = tr * 64 + r0 + 4 * i, cc = tc * 64 + c;
        t[r0 + 4 * i][c] = (r < it.rows && cc < it.cols) ? src[(size_t)r * it.cols + cc] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int cl = (threadIdx.x >> 4) + 16 * i, cc = tc * 64 + cl, rl = 4 * (threadIdx.x & 15), r = tr * 64 + rl;
        if (cc < it.cols && r < it.ld) {                                    // rows past `rows` (up to ld) are zeros
            uint16_t h[4], l[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float v = t[rl + j][cl];
                h[j] = nr_f2bf(v);
                l[j] = nr_f2bf(v - nr_bf2f(h[j]));
            }
            nr_store_t4(it.hi, it.lo, (size_t)cc * it.ld + r, r, it.ld, h, l);
        }
    }
}

extern "C" int nr_split_group(int n, const NrSplitItem* items, void* stream) {
    if (!items || n <= 0) return NR_EINVAL;
    if (n > NR_SPLIT_MAX) return NR_EUNSUPPORTED;
    NrSplitGroup g;
    g.n = n;
    int total = 0;
    for (int i = 0; i < n; ++i) {
        const NrSplitItem& it = items[i];
        if (!it.src || !it.hi || it.rows <= 0 || it.cols <= 0 || it.mode < 0 || it.mode > 6) return NR_EINVAL;
        if (it.mode >= 3 && it.group <= 0) return NR_EINVAL;
        if (it.mode >= 4 && it.cols != 3 * it.group) return NR_EINVAL;
        const bool transposed = it.mode >= 1 && it.mode <= 3;
        if (transposed ? it.ld < it.rows : it.ld < it.cols) return NR_EINVAL;
        g.it[i] = it;
        g.start[i] = total;
        // transposed: the item's row tiles cover [0, rows rounded up to 64) of its destination columns -- the K padding of a
        // GEMM operand, written as zeros (clipped at ld); a caller that packs several items side by side into one wide buffer
        // gives each a 64-aligned start
        const int row_tiles = (it.rows + 63) / 64;
        const int col_tiles = transposed ? (it.cols + 63) / 64 : (it.ld + 63) / 64;
        total += row_tiles * col_tiles;
    }
    for (int i = n; i <= NR_SPLIT_MAX; ++i) g.start[i] = total;
    // the kernel derives the column-tile count from `cols`: for the row-major form ld == cols rounded up to 64 at most
    for (int i = 0; i < n; ++i)
        if ((items[i].mode == 0 || items[i].mode >= 4) && (items[i].ld + 63) / 64 != (items[i].cols + 63) / 64) return NR_EUNSUPPORTED;
    hipLaunchKernelGGL(nr_split_group_kernel, dim3(total), dim3(256), 0, (hipStream_t)stream, g);
    NR_LAUNCH_CHECK();
    return NR_OK;
}

// ---- grouped column sums: dst[c] = sum_r src[r, c], fixed order (deterministic) ----------------------------------------------
struct NrColsumGroup {
    NrColsumItem it[NR_COLSUM_MAX];
    int start[NR_COLSUM_MAX + 1];    // first workgroup (64 columns) of every item
    int n;
};

__global__ __launch_bounds__(1024) void nr_colsum_group_kernel(NrColsumGroup g) {
    __shared__ float red[16][64];
    const int wg = blockIdx.x;
    int gi = 0;
    for (int i = 1; i < g.n; ++i)
        if (wg >= g.start[i]) gi = i;
    const NrColsumItem it = g.it[gi];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = (wg - g.start[gi]) * 64 + lane;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (c < it.cols) {
        int r = wave;
        for (; r + 48 < it.rows; r += 64) {           // four independent loads in flight per wave
            s0 += it.src[(size_t)r * it.cols + c];
            s1 += it.src[(size_t)(r + 16) * it.cols + c];
            s2 += it.src[(size_t)(r + 32) * it.cols + c];
            s3 += it.src[(size_t)(r + 48) * it.cols + c];
        }
        for (; r < it.rows; r += 16) s0 += it.src[(size_t)r * it.cols + c];
    }
    red[wave][lane] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (wave == 0 && c < it.cols) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) s += red[w][lane];
        it.dst[c] = s * it.scale;
    }
}

extern "C" int nr_colsum_group(int n, const NrColsumItem* items, void* stream) {
    if (!items || n <= 0) return NR_EINVAL;
    if (n > NR_COLSUM_MAX) return NR_EUNSUPPORTED;
    NrColsumGroup g;
    g.n = n;
    int total = 0;
    for (int i = 0; i < n; ++i) {
        if (!items[i].src || !items[i].dst || items[i].rows <= 0 || items[i].cols <= 0) return NR_EINVAL;
        g.it[i] = items[i];
        g.start[i] = total;
        total += (items[i].cols + 63) / 64;
    }
    for (int i = n; i <= NR_COLSUM_MAX; ++i) g.start[i] = total;
    hipLaunchKernelGGL(nr_colsum_group_kernel, dim3(total), dim3(1024), 0, (hipStream_t)stream, g);
    NR_LAUNCH_CHECK();
    return NR_OK;
}

// ---- grouped split-bf16 GEMMs behind the C ABI (the forward uses the same launcher internally) ---------------------------------
extern "C" int nr_linear_group(int n, const NrLinearProblem* probs, void* stream) {
    if (!probs || n <= 0) return NR_EINVAL;
    if (n > NR_LINEAR_MAX_GROUP) return NR_EUNSUPPORTED;
    NrLinearArgs a[NR_LINEAR_MAX_GROUP];
    bool conv = false;
    for (int i = 0; i < n; ++i) {
        a[i] = NrLinearArgs{probs[i].x_hi, probs[i].x_lo, probs[i].w_hi, probs[i].w_lo, probs[i].bias, probs[i].residual,
                            probs[i].out, probs[i].M, probs[i].N, probs[i].K};
        a[i].ld = probs[i].ld;
        if (probs[i].conv_n < 0) return NR_EINVAL;
        a[i].conv_n = probs[i].conv_n;
        conv = conv || probs[i].conv_n > 0;
    }
    return nr_linear_group_launch(a, n, (hipStream_t)stream, conv);      // conv: EVERY problem must be one (checked there)
}

// ---- score-biased attention backward, one workgroup per sample, one wave per head ------------------------------------------------
// Forward (cluster.py:868-885): logit[h,ci,n] = scale q[ci,h].k[n,h] + score[n];  p = softmax_n;  att[ci,h] = sum_n p v[n,h].
// Lane n of a head's wave holds the logit / probability of token n (N <= 64); lane j holds channel j of the head (dh = 64)
// in the row operations.  k and v rows of the head live in registers, 32 tokens at a time.
template <typename A>
struct NrBwdGroupOf {
    A p[NR_CTM_MAX_GROUP];
    int start[NR_CTM_MAX_GROUP + 1];
    int n;
    __device__ __forceinline__ int find(int wg) const {
        int g = 0;
#pragma unroll
        for (int i = 1; i < NR_CTM_MAX_GROUP; ++i)
            if (i < n && wg >= start[i]) g = i;
        return g;
    }
};

struct NrAttnBwdArgs {
    const float *q, *kv, *score, *d_att;
    int N, C, cnum, heads;
    float scale;
    float *d_q, *d_kv, *d_score;
    uint16_t *dq_hi, *dq_lo, *dkv_hi, *dkv_lo;
};

__device__ __forceinline__ float nr_rl(float v, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}

#define NR_ATT_LD 129
// NCH token chunks of 32 (N <= 32 * NCH).  NCH == 1: the head's k | v rows are staged in the wave's own LDS slice first
// ([n][128] floats: every later read is a ds_read at an immediate offset); read straight from global memory they cost a
// 64-bit address pair per row and the kernel spilled.  NCH == 2 (ActivityNet token counts: N up to 64) keeps the global reads of
// k | v -- its slices do not fit the LDS (N x 4 KiB per head) -- and splits the work in two phases so that the 4 x 32 d_k / d_v
// accumulators of a lane are never live together with the logit pass (154 spilled registers in round 3): phase 1 runs the
// queries (logits, softmax backward, d_q) and parks every query's d_logit / p row in the wave's LDS slice ([c][128] floats);
// phase 2 accumulates d_k / d_v over the queries one 32-token chunk at a time from those rows (LDS broadcast reads).  Same
// arithmetic in the same order as the one-phase form.
template <int NCH>
__device__ __forceinline__ void nr_attn_bwd_body(const NrAttnBwdArgs& a, const int b, float* s_ds /* [heads][64] */, float* s_kv) {
    const int lane = threadIdx.x & 63, h = threadIdx.x >> 6;
    const int N = a.N, C = a.C, c = a.cnum;
    const size_t kvb = (size_t)b * N * 2 * C + (size_t)h * 64 + lane;
    // NCH == 1: this wave's LDS slice = N rows of k | v (NR_ATT_LD floats apart: an odd stride, so that lanes that hold
    // different TOKENS read a column without bank conflicts) + q and d_att of the current query (128 floats)
    float* slice = s_kv + (size_t)h * (N * NR_ATT_LD + 128);
    float* sk = slice + lane;
    float* sqd = slice + N * NR_ATT_LD;
    if constexpr (NCH == 1) {
        for (int n = 0; n < N; ++n) {
            sk[n * NR_ATT_LD] = a.kv[kvb + (size_t)n * 2 * C];
            sk[n * NR_ATT_LD + 64] = a.kv[kvb + (size_t)n * 2 * C + C];
        }
        // (the slice is private to this wave: no barrier)
    }
    auto K_ = [&](int n) -> float { if constexpr (NCH == 1) return sk[n * NR_ATT_LD]; else return a.kv[kvb + (size_t)n * 2 * C]; };
    auto V_ = [&](int n) -> float { if constexpr (NCH == 1) return sk[n * NR_ATT_LD + 64]; else return a.kv[kvb + (size_t)n * 2 * C + C]; };
    constexpr int NACC = NCH == 1 ? 1 : 0;            // one-phase form: accumulators live through the query loop
    float dk[NACC ? NCH : 1][32], dv[NACC ? NCH : 1][32];
    if constexpr (NCH == 1) {
#pragma unroll
        for (int n = 0; n < 32; ++n) dk[0][n] = dv[0][n] = 0.f;
    }
    float* s_rows = s_kv + (size_t)h * c * 128;       // NCH > 1: [c][d_logit 64 | p 64] of this wave
    const float sc_n = lane < N ? a.score[(size_t)b * N + lane] : -INFINITY;
    float dsc = 0.f;                       // lane n: sum over queries of d_logit[n] (this head)
    for (int ci = 0; ci < c; ++ci) {
        const size_t qo = ((size_t)b * c + ci) * C + (size_t)h * 64 + lane;
        const float qv = a.q[qo], da = a.d_att[qo];
        float lg = -INFINITY, dp = 0.f;    // lane n: logit / d_p of token n
        if constexpr (NCH == 1) {
            // lane = (token, half of the head's 64 dims): 32 multiply-adds per lane and one exchange between the halves instead
            // of two 64-lane reductions per token (48 reductions per query at 24 tokens: that was most of this kernel's time)
            sqd[lane] = qv;
            sqd[64 + lane] = da;
            const int tok = lane & 31, d0 = (lane >> 5) * 32;
            float l_ = 0.f, d_ = 0.f;
            if (tok < N) {
                const float* kr = slice + tok * NR_ATT_LD + d0;
#pragma unroll 8
                for (int j = 0; j < 32; ++j) {
                    l_ = __builtin_fmaf(sqd[d0 + j], kr[j], l_);
                    d_ = __builtin_fmaf(sqd[64 + d0 + j], kr[64 + j], d_);
                }
            }
            l_ += __shfl_xor(l_, 32);
            d_ += __shfl_xor(d_, 32);
            if (lane < N) { lg = l_ * a.scale; dp = d_; }
        } else {
            // (a rolled loop: fully unrolled, the 2 x 64 row loads of a query are hoisted together and the kernel spills)
            for (int n = 0; n < N; ++n) {
                const float l_ = nr_wave_sum(qv * K_(n)) * a.scale;
                const float d_ = nr_wave_sum(da * V_(n));
                if (lane == n) { lg = l_; dp = d_; }
            }
        }
        lg += sc_n;                         // masked tokens: -inf -> p = 0
        const float mx = nr_wave_max(lg);
        const float e = lane < N ? __expf(lg - mx) : 0.f;
        const float p = e / nr_wave_sum(e);
        const float dl = p * (dp - nr_wave_sum(p * dp));       // softmax backward
        dsc += dl;
        float dq = 0.f;
        if constexpr (NCH == 1) {
#pragma unroll
            for (int n = 0; n < 32; ++n) {
                if (n < N) {
                    const float dln = nr_rl(dl, n);
                    dq += dln * K_(n);
                    dk[0][n] += dln * qv;
                    dv[0][n] += nr_rl(p, n) * da;
                }
            }
        } else {
            for (int n = 0; n < N; ++n) dq += nr_rl(dl, n) * K_(n);
            s_rows[ci * 128 + lane] = dl;
            s_rows[ci * 128 + 64 + lane] = p;
        }
        dq *= a.scale;
        a.d_q[qo] = dq;
        const uint16_t hb = nr_f2bf(dq);
        a.dq_hi[qo] = hb;
        a.dq_lo[qo] = nr_f2bf(dq - nr_bf2f(hb));
    }
    auto store_chunk = [&](const float (&dkc)[32], const float (&dvc)[32], int u) {
#pragma unroll
        for (int n = 0; n < 32; ++n)
            if (u * 32 + n < N) {
                const size_t o = kvb + (size_t)(u * 32 + n) * 2 * C;
                const float k_ = dkc[n] * a.scale, v_ = dvc[n];
                a.d_kv[o] = k_;
                a.d_kv[o + C] = v_;
                uint16_t hb = nr_f2bf(k_);
                a.dkv_hi[o] = hb;
                a.dkv_lo[o] = nr_f2bf(k_ - nr_bf2f(hb));
                hb = nr_f2bf(v_);
                a.dkv_hi[o + C] = hb;
                a.dkv_lo[o + C] = nr_f2bf(v_ - nr_bf2f(hb));
            }
    };
    if constexpr (NCH == 1) {
        store_chunk(dk[0], dv[0], 0);
    } else {
        __syncthreads();                    // (every wave reads only its own rows; the barrier orders its LDS stores before them)
        for (int u = 0; u < NCH; ++u) {
            if (u * 32 >= N) break;
            float dkc[32], dvc[32];
#pragma unroll
            for (int n = 0; n < 32; ++n) dkc[n] = dvc[n] = 0.f;
            for (int ci = 0; ci < c; ++ci) {
                const size_t qo = ((size_t)b * c + ci) * C + (size_t)h * 64 + lane;
                const float qv = a.q[qo], da = a.d_att[qo];
                const float* row = s_rows + ci * 128 + u * 32;
#pragma unroll
                for (int n = 0; n < 32; ++n) {
                    dkc[n] += row[n] * qv;          // (tokens >= N hold d_logit = p = 0)
                    dvc[n] += row[64 + n] * da;
                }
            }
            store_chunk(dkc, dvc, u);
        }
    }
    // d_score[n] = sum over heads and queries of d_logit (the score biases every head and query): heads meet in LDS
    s_ds[h * 64 + lane] = dsc;
    __syncthreads();
    if (h == 0 && lane < N) {
        float s = 0.f;
        for (int w = 0; w < a.heads; ++w) s += s_ds[w * 64 + lane];
        a.d_score[(size_t)b * N + lane] = s;
    }
}

template <int NCH>
__global__ __launch_bounds__(512) void nr_attn_bwd_group_kernel(NrBwdGroupOf<NrAttnBwdArgs> g) {   // <= 8 heads: 256 VGPRs per lane
    __shared__ float s_ds[8 * 64];
    extern __shared__ __attribute__((aligned(16))) float s_kv[];        // NCH == 1: [heads][N * NR_ATT_LD + 128]
    const int gi = g.find(blockIdx.x);
    nr_attn_bwd_body<NCH>(g.p[gi], blockIdx.x - g.start[gi], s_ds, s_kv);
}

extern "C" int nr_ctm_attn_bwd(int n, const NrCtmAttnBwdDesc* d, void* stream) {
    if (!d || n <= 0 || n > NR_CTM_MAX_GROUP) return NR_EINVAL;
    NrBwdGroupOf<NrAttnBwdArgs> g;
    g.n = n;
    int total = 0, heads = 0, nmax = 0;
    for (int i = 0; i < n; ++i) {
        const NrCtmAttnBwdDesc& s = d[i];
        if (!s.q || !s.kv || !s.score || !s.d_att || !s.d_q || !s.d_kv || !s.d_score || !s.dq_hi || !s.dq_lo || !s.dkv_hi || !s.dkv_lo)
            return NR_EINVAL;
        if (s.n_samples <= 0 || s.N <= 0 || s.cnum <= 0 || s.heads <= 0) return NR_EINVAL;
        if (s.N > 64 || s.C != s.heads * 64 || s.heads > 8) return NR_EUNSUPPORTED;
        if (heads && heads != s.heads) return NR_EUNSUPPORTED;          // one block size for the launch
        heads = s.heads;
        nmax = s.N > nmax ? s.N : nmax;
        g.p[i] = NrAttnBwdArgs{s.q, s.kv, s.score, s.d_att, s.N, s.C, s.cnum, s.heads, 1.0f / sqrtf(64.0f),
                               s.d_q, s.d_kv, s.d_score, s.dq_hi, s.dq_lo, s.dkv_hi, s.dkv_lo};
        g.start[i] = total;
        total += s.n_samples;
    }
    for (int i = n; i <= NR_CTM_MAX_GROUP; ++i) g.start[i] = total;
    if (nmax <= 32) {
        const size_t lds = (size_t)heads * (nmax * NR_ATT_LD + 128) * sizeof(float);   // <= 133 KiB
        if (lds > 48 * 1024) {
            hipError_t e = hipFuncSetAttribute((const void*)nr_attn_bwd_group_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return (int)e;
        }
        hipLaunchKernelGGL(nr_attn_bwd_group_kernel<1>, dim3(total), dim3(64 * heads), lds, (hipStream_t)stream, g);
    } else {
        int cmax = 0;
        for (int i = 0; i < n; ++i) cmax = d[i].cnum > cmax ? d[i].cnum : cmax;
        const size_t lds = (size_t)heads * cmax * 128 * sizeof(float);                 // every query's d_logit | p row, per head
        if (lds > 150 * 1024) return NR_EUNSUPPORTED;
        if (lds > 48 * 1024) {
            hipError_t e = hipFuncSetAttribute((const void*)nr_attn_bwd_group_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return (int)e;
        }
        hipLaunchKernelGGL(nr_attn_bwd_group_kernel<2>, dim3(total), dim3(64 * heads), lds, (hipStream_t)stream, g);
    }
    NR_LAUNCH_CHECK();
    return NR_OK;
}

// ---- the middle of the stage, backward, one workgroup per sample, one wave per token row ----------------------------------------
struct NrMidBwdArgs {
    const float *d_qn, *d_kvn, *g, *merged_pb, *proj_b, *xn, *y, *tokw, *d_score, *mask, *n1_w, *ln_w, *sc_w;
    const int64_t* assign;
    int N, C, cnum;
    float eps_ctm, eps_n1;
    float* d_y;                   // [B*N, C]
    uint16_t *dy_hi, *dy_lo;      // [B*N, C]: d_y as a bf16 pair (A operand of the transposed convolution, read in place)
    float* partial;               // [B, 6, C]: d norm1.weight, d norm1.bias, d ctm.norm.weight, d ctm.norm.bias, d score.weight, [d score.bias, 0...]
};

// LayerNorm backward of one row held CPL channels per lane: x the LayerNorm INPUT, dyv the gradient of its OUTPUT, gamma
// the weight.  Returns d x in dyv; adds the row's contributions to dgam / dbet.
template <int CPL>
__device__ __forceinline__ void nr_ln_bwd_row(const float (&x)[CPL], float (&dyv)[CPL], const float (&gamma)[CPL], float (&dgam)[CPL],
                                              float (&dbet)[CPL], const int cpl, const int C, const float eps) {
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < CPL; ++q) s += q < cpl ? x[q] : 0.f;
    const float mu = nr_wave_sum(s) / (float)C;
    float var = 0.f;
#pragma unroll
    for (int q = 0; q < CPL; ++q)
        if (q < cpl) { const float d = x[q] - mu; var += d * d; }
    const float rstd = rsqrtf(nr_wave_sum(var) / (float)C + eps);
    float m1 = 0.f, m2 = 0.f;
    float xh[CPL];
#pragma unroll
    for (int q = 0; q < CPL; ++q) {
        xh[q] = q < cpl ? (x[q] - mu) * rstd : 0.f;
        if (q < cpl) {
            dgam[q] += dyv[q] * xh[q];
            dbet[q] += dyv[q];
            dyv[q] *= gamma[q];
            m1 += dyv[q];
            m2 += dyv[q] * xh[q];
        }
    }
    m1 = nr_wave_sum(m1) / (float)C;
    m2 = nr_wave_sum(m2) / (float)C;
#pragma unroll
    for (int q = 0; q < CPL; ++q)
        if (q < cpl) dyv[q] = rstd * (dyv[q] - m1 - xh[q] * m2);
}

#define MB_THREADS 512      // 8 waves: the row state (three rows + five gradient accumulators + three parameter vectors, CPL
                            // registers each) needs the 256-register budget of two waves per SIMD
// THREADS: MB_THREADS for C <= 512 (CPL = 8); C > 512 (CPL = 16: eleven CPL-sized register vectors) runs 256 threads -- one wave per
// SIMD, the whole 512-register file of a lane (at two waves per SIMD the kernel spilled 903 registers, round 3).
template <int CPL, int THREADS = MB_THREADS>
__device__ __forceinline__ void nr_mid_bwd_body(const NrMidBwdArgs& a, const int b, float* sm /* dynamic LDS */) {
    constexpr int NW = THREADS / 64;
    const int N = a.N, C = a.C, c = a.cnum;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cpl = C / 64;
    float* s_dm = sm;                          // [c][C]  d merged (with the block's residual)
    float* s_red = sm + (size_t)c * C;         // [NW][C] cross-wave reduction of one parameter-gradient vector
    __shared__ float s_w[64], s_tot[64], s_dsh[64], s_dtot[64];
    __shared__ int s_a[64];
    __shared__ float s_bs[NW];
    float n1w[CPL], lnw[CPL], scw[CPL];
    float dg1[CPL], db1[CPL], dgc[CPL], dbc[CPL], dws[CPL];
#pragma unroll
    for (int q = 0; q < CPL; ++q) {
        const int ch = q * 64 + lane;
        const bool on = q < cpl;
        n1w[q] = on ? a.n1_w[ch] : 0.f;
        lnw[q] = on ? a.ln_w[ch] : 0.f;
        scw[q] = on ? a.sc_w[ch] : 0.f;
        dg1[q] = db1[q] = dgc[q] = dbc[q] = dws[q] = 0.f;
    }
    float dbs = 0.f;
    if (tid < N) {
        s_w[tid] = a.tokw[(size_t)b * N + tid];
        s_a[tid] = (int)a.assign[(size_t)b * N + tid];
    }
    // ---- A: norm1 backward of the merged rows (+ the residual path of the block) -> s_dm --------------------------------
    for (int ci = wave; ci < c; ci += NW) {
        const size_t row = ((size_t)b * c + ci) * C;
        float x[CPL], dyv[CPL], gr[CPL];
#pragma unroll
        for (int q = 0; q < CPL; ++q) {
            const int ch = q * 64 + lane;
            const bool on = q < cpl;
            x[q] = on ? a.merged_pb[row + ch] - a.proj_b[ch] : 0.f;
            dyv[q] = on ? a.d_qn[row + ch] : 0.f;
            gr[q] = on ? a.g[row + ch] : 0.f;
        }
        nr_ln_bwd_row<CPL>(x, dyv, n1w, dg1, db1, cpl, C, a.eps_n1);
#pragma unroll
        for (int q = 0; q < CPL; ++q)
            if (q < cpl) s_dm[ci * C + q * 64 + lane] = dyv[q] + gr[q];
    }
    __syncthreads();
    if (tid < c) {                              // all_weight of every cluster (cluster.py:536-540)
        float t = 0.f;
        for (int n = 0; n < N; ++n) t += s_a[n] == tid ? s_w[n] : 0.f;
        s_tot[tid] = t + 1e-6f;
    }
    // ---- B1: d share_n = xn_n . d merged[cluster of n] ---------------------------------------------------------------
    for (int r = wave; r < N; r += NW) {
        const size_t row = ((size_t)b * N + r) * C;
        const int cl = s_a[r];
        float d = 0.f;
#pragma unroll
        for (int q = 0; q < CPL; ++q)
            if (q < cpl) d += a.xn[row + q * 64 + lane] * s_dm[cl * C + q * 64 + lane];
        d = nr_wave_sum(d);
        if (lane == 0) s_dsh[r] = d;
    }
    __syncthreads();
    if (tid < c) {
        const float tot = s_tot[tid];
        float t = 0.f;
        for (int n = 0; n < N; ++n) t += s_a[n] == tid ? -s_dsh[n] * s_w[n] / (tot * tot) : 0.f;
        s_dtot[tid] = t;
    }
    __syncthreads();
    // ---- B2: per token row: norm1 backward (kv path), cluster means, score, LayerNorm(ctm) backward -> d_y -----------
    for (int r = wave; r < N; r += NW) {
        const size_t tok = (size_t)b * N + r;
        const size_t row = tok * C;
        const int cl = s_a[r];
        float xr[CPL], dx[CPL], yr[CPL];
#pragma unroll
        for (int q = 0; q < CPL; ++q) {
            const int ch = q * 64 + lane;
            const bool on = q < cpl;
            xr[q] = on ? a.xn[row + ch] : 0.f;
            dx[q] = on ? a.d_kvn[row + ch] : 0.f;
            yr[q] = on ? a.y[row + ch] : 0.f;
        }
        nr_ln_bwd_row<CPL>(xr, dx, n1w, dg1, db1, cpl, C, a.eps_n1);          // dx = d xn through norm1 -> kv
        const float wn = s_w[r], tot = s_tot[cl];
        const float share = wn / tot;
        const float d_w = s_dsh[r] / tot + s_dtot[cl];
        float d_sc = d_w * wn + a.d_score[tok];
        if (a.mask) d_sc *= a.mask[tok] > 0.f ? 1.f : 0.f;
#pragma unroll
        for (int q = 0; q < CPL; ++q)
            if (q < cpl) {
                dx[q] += s_dm[cl * C + q * 64 + lane] * share + d_sc * scw[q];
                dws[q] += d_sc * xr[q];
            }
        if (lane == 0) dbs += d_sc;
        nr_ln_bwd_row<CPL>(yr, dx, lnw, dgc, dbc, cpl, C, a.eps_ctm);         // dx = d y
#pragma unroll
        for (int q = 0; q < CPL; ++q)
            if (q < cpl) {
                const int ch = q * 64 + lane;
                const float v = dx[q];
                a.d_y[row + ch] = v;
                const uint16_t hb = nr_f2bf(v), lb = nr_f2bf(v - nr_bf2f(hb));
                // the bf16 pair of d_y: the transposed convolution reads it in place (three shifted row sets, like the forward's)
                a.dy_hi[row + ch] = hb;
                a.dy_lo[row + ch] = lb;
            }
    }
    // ---- C: the waves' partial parameter gradients meet in LDS, one vector at a time (fixed order) ---------------------
    float* out = a.partial + (size_t)b * 6 * C;
    auto reduce = [&](const float (&v)[CPL], int slot) {
        __syncthreads();
#pragma unroll
        for (int q = 0; q < CPL; ++q)
            if (q < cpl) s_red[wave * C + q * 64 + lane] = v[q];
        __syncthreads();
        for (int ch = tid; ch < C; ch += THREADS) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) s += s_red[w * C + ch];
            out[slot * C + ch] = s;
        }
    };
    reduce(dg1, 0);
    reduce(db1, 1);
    reduce(dgc, 2);
    reduce(dbc, 3);
    reduce(dws, 4);
    if (lane == 0) s_bs[wave] = dbs;
    __syncthreads();
    for (int ch = tid; ch < C; ch += THREADS) {
        float s = 0.f;
        if (ch == 0)
            for (int w = 0; w < NW; ++w) s += s_bs[w];
        out[5 * C + ch] = s;
    }
}

template <int CPL, int THREADS = MB_THREADS>
__global__ __launch_bounds__(THREADS) void nr_mid_bwd_group_kernel(NrBwdGroupOf<NrMidBwdArgs> g) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int gi = g.find(blockIdx.x);
    nr_mid_bwd_body<CPL, THREADS>(g.p[gi], blockIdx.x - g.start[gi], sm);
}

extern "C" int nr_ctm_mid_bwd(int n, const NrCtmMidBwdDesc* d, void* stream) {
    if (!d || n <= 0 || n > NR_CTM_MAX_GROUP) return NR_EINVAL;
    NrBwdGroupOf<NrMidBwdArgs> g;
    g.n = n;
    int total = 0;
    size_t lds = 0;
    bool small = true;
    for (int i = 0; i < n; ++i) {
        const NrCtmMidBwdDesc& s = d[i];
        if (!s.d_qn || !s.d_kvn || !s.g || !s.merged_pb || !s.proj_b || !s.xn || !s.y || !s.tokw || !s.d_score || !s.n1_w || !s.ln_w ||
            !s.sc_w || !s.assign || !s.d_y || !s.dy_hi || !s.dy_lo || !s.partial)
            return NR_EINVAL;
        if (s.n_samples <= 0 || s.N <= 0 || s.cnum <= 0 || s.cnum > s.N) return NR_EINVAL;
        if (s.N > 64 || s.C <= 0 || (s.C % 64) != 0 || s.C > 64 * CF_MAX_CPL) return NR_EUNSUPPORTED;
        const size_t need = ((size_t)s.cnum + MB_THREADS / 64) * s.C * sizeof(float);
        if (need > 150 * 1024) return NR_EUNSUPPORTED;
        lds = need > lds ? need : lds;
        small = small && s.C <= 512;
        g.p[i] = NrMidBwdArgs{s.d_qn, s.d_kvn, s.g, s.merged_pb, s.proj_b, s.xn, s.y, s.tokw, s.d_score, s.mask, s.n1_w, s.ln_w, s.sc_w,
                              s.assign, s.N, s.C, s.cnum, s.eps_ctm, s.eps_n1, s.d_y, s.dy_hi, s.dy_lo, s.partial};
        g.start[i] = total;
        total += s.n_samples;
    }
    for (int i = n; i <= NR_CTM_MAX_GROUP; ++i) g.start[i] = total;
    const void* k = small ? (const void*)nr_mid_bwd_group_kernel<8> : (const void*)nr_mid_bwd_group_kernel<CF_MAX_CPL, 256>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    if (small) hipLaunchKernelGGL(nr_mid_bwd_group_kernel<8>, dim3(total), dim3(MB_THREADS), lds, (hipStream_t)stream, g);
    else hipLaunchKernelGGL((nr_mid_bwd_group_kernel<CF_MAX_CPL, 256>), dim3(total), dim3(256), lds, (hipStream_t)stream, g);
    NR_LAUNCH_CHECK();
    return NR_OK;
}
