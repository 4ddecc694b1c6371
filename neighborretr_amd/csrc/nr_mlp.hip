// Token-weight scorer: Linear(d,H) + ReLU + Linear(H,1) as an MFMA GEMM with the second layer
// folded into the epilogue, then the masked softmax over each sample's tokens.
// Reference: modeling.py:148-153 (the MLP), :485-487 / :490-492 (mask -> -9e15, softmax).
//
// The GEMM consumes the NORMALISED bf16 tokens written by nr_prepare_tokens and rescales each
// accumulator row by the token's original norm (W1 x = ||x|| * W1 x_hat), so the features are
// converted to bf16 only once for both the scorer and the similarity kernel.  Rows of masked
// tokens are zero vectors; their logits are overwritten with -9e15 by the softmax anyway.
#include <cstdlib>
#include "nr_gemm_tile.h"
#include "../../include/nr_hip.h"
// ping-pong K loop for the 8-wave one-pass blocks: measured 16.7 vs 16.8 us for the 12288 bank text tokens (the loop is not
// what bounds a scorer launch of eight K slices: its start and its epilogue are) -- off; -DNR_MLP_PP=1 builds it in
#ifndef NR_MLP_PP
#define NR_MLP_PP 0
#endif

// Block shapes: BM = 32*MI token rows x BN = 16*WC*NI hidden units with BN a multiple of 128 (one partial-logit row
// per 128 hidden units, the granularity nr_token_softmax sums).  The main loop is bound by the operand bytes a CU
// pulls into LDS (~25 B/clk/CU by LDS-DMA), so the host picks, per launch, the shape that minimises
//     ceil(workgroups / 256 CUs) * (BM + BN)
// -- e.g. 96x128 for the 3072 batch text tokens (256 workgroups, one per CU) and 192x256 on 8 waves for the
// 12288 bank text tokens (again 256) instead of 128x128 everywhere (192 resp. 768 workgroups).
// The masked softmax over each sample's tokens folded into the scorer launch (counters != nullptr): the H/BN column
// blocks of one row tile add to the tile's counter after their partial logits are visible device-wide; whichever arrives
// last sums the parts of the tile's samples (BM % N == 0: whole samples per row tile), applies mask and softmax
// (modeling.py:485-492) and resets the counter.  One launch less per scorer call (four per step) on the local chain.
struct NrMlpSoftmax {
    unsigned int* counters;      // [row tiles], zero on entry and on exit
    const float* b2;
    const float* mask;           // [n_tok] or nullptr
    int N;                       // tokens per sample
    float* w;                    // [n_tok] softmax weights
    float* logits;               // [n_tok] or nullptr
};

// one scorer call: what a workgroup needs to know about it
struct NrMlpProblem {
    const uint16_t *tok_hi, *tok_lo;
    const float* norm;
    int n_tok, d;
    const uint16_t *w1_hi, *w1_lo;
    const float *b1, *w2;
    int H;
    float* logit_part;
    NrMlpSoftmax sm;
};

// wg: the workgroup's index INSIDE this problem's (8-padded) grid -- blockIdx.x, or blockIdx.x minus the grids of the problems
// in front of it in a paired launch (multiples of 8: the XCD a workgroup lands on is the same either way)
// ACC3 (one-pass ring only): the split-bf16 product as THREE accumulated passes over K through the ping-pong loop
// (NrGemmTile::run_pp3: Ah Bh, then Ah Bl, then Al Bh) -- what lets split-bf16 token sets share a grid, a block shape and an
// LDS footprint with one-pass sets (nr_mlp_group_kernel)
template <int MI, int NI, int WC, bool X3, int STAGES, bool ACC3 = false>
__device__ __forceinline__ void nr_mlp_body(const NrMlpProblem& q, const int wg, char* smem) {
    const uint16_t* __restrict__ tok_hi = q.tok_hi;
    const uint16_t* __restrict__ tok_lo = q.tok_lo;
    const float* __restrict__ norm = q.norm;
    const int n_tok = q.n_tok, d = q.d, H = q.H;
    const uint16_t* __restrict__ w1_hi = q.w1_hi;
    const uint16_t* __restrict__ w1_lo = q.w1_lo;
    const float* __restrict__ b1 = q.b1;
    const float* __restrict__ w2 = q.w2;
    float* __restrict__ logit_part = q.logit_part;
    const NrMlpSoftmax& sm = q.sm;
    using Tile = NrGemmTile<MI, NI, X3, 16, 16, STAGES, WC>;
    constexpr int BM = Tile::BM, BN = Tile::BN;
    constexpr int WCOLS = 16 * NI;                 // hidden units per wave
    constexpr int WPP = 128 / WCOLS;               // waves per 128-unit part
    static_assert(BN % 128 == 0 && 128 % WCOLS == 0, "a block must hold whole 128-unit parts");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave / WC, wc = wave % WC;
    // 1-D grid, tiles numbered column-fastest: the H/BN column tiles of one token-row tile are neighbours and
    // nr_xcd_chunk_tile keeps neighbours on one XCD, whose L2 then fetches those token rows once
    const int n_col = H / BN;
    const int tile_id = nr_xcd_chunk_tile(wg, n_col * ((n_tok + BM - 1) / BM));
    if (tile_id < 0) return;
    const int bx = tile_id % n_col, by = tile_id / n_col;
    const int row0 = by * BM, col0 = bx * BN;

    Tile tile;
    tile.zero();
    // 8-wave blocks on a two-deep ring walk K with the ping-pong loop of the similarity kernel (the two wave rows alternate
    // between an MFMA phase and a memory phase: nr_gemm_tile.h); NR_MLP_PLAIN=1 (tuning builds): the plain loop
    if constexpr (ACC3) {
        static_assert(WC == 4 && STAGES == 2 && !X3, "three accumulated passes: 8 waves, two-deep one-pass ring");
        tile.run_pp3(tok_hi, tok_lo, row0, n_tok, w1_hi, w1_lo, col0, H, d, smem);
    } else if constexpr (WC == 4 && STAGES == 2 && !X3) {
        if (NR_MLP_PP) tile.run_pp(tok_hi, tok_lo, row0, n_tok, w1_hi, w1_lo, col0, H, d, smem);
        else tile.run(tok_hi, tok_lo, row0, n_tok, w1_hi, w1_lo, col0, H, d, smem);
    } else {
        tile.run(tok_hi, tok_lo, row0, n_tok, w1_hi, w1_lo, col0, H, d, smem);
    }

    float bb[NI], ww[NI];
#pragma unroll
    for (int n = 0; n < NI; ++n) {
        int c = col0 + wc * WCOLS + n * 16 + (lane & 15);
        bb[n] = b1[c];
        ww[n] = w2[c];
    }
    float* sPart = reinterpret_cast<float*>(smem);   // [WC][BM]; staging LDS is free after run()
#pragma unroll
    for (int m = 0; m < MI; ++m)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int r = wr * 16 * MI + m * 16 + (lane >> 4) * 4 + j;
            int gr = min(row0 + r, n_tok - 1);
            float sc = norm[gr];
            float v = 0.f;
#pragma unroll
            for (int n = 0; n < NI; ++n) v += fmaxf(tile.acc[m][n][j] * sc + bb[n], 0.f) * ww[n];
            v += __shfl_xor(v, 1);
            v += __shfl_xor(v, 2);
            v += __shfl_xor(v, 4);
            v += __shfl_xor(v, 8);
            if ((lane & 15) == 0) sPart[wc * BM + r] = v;
        }
    __syncthreads();
    // part p of this block = hidden units [col0 + 128p, col0 + 128p + 128) = wave columns [p*WPP, (p+1)*WPP)
    for (int e = tid; e < (BN / 128) * BM; e += 64 * Tile::NW) {
        const int part = e / BM, r = e - part * BM;
        if (row0 + r >= n_tok) continue;
        float v = 0.f;
#pragma unroll
        for (int q = 0; q < WPP; ++q) v += sPart[(part * WPP + q) * BM + r];
        float* dst = logit_part + (size_t)(bx * (BN / 128) + part) * n_tok + row0 + r;
        // fused softmax: another workgroup reads these -> `sc1` stores (written through, no L2 write-back fence needed)
        if (sm.counters) __hip_atomic_store(dst, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else *dst = v;
    }
    if (sm.counters == nullptr) return;
    // ---- fused softmax: the last column block of this row tile ----------------------------------------------------
    // Hand-off without cache-wide fences (a __threadfence() = buffer_wbl2 + buffer_inv costs 3.5-6.5 us per workgroup and
    // serialises at the L2: measured +25 us per scorer launch): every byte handed off is stored `sc1` (above) and loaded `sc1`
    // (below), every storing wave drains its stores (vmcnt 0) before the workgroup barrier behind which ONE lane adds to the
    // tile's counter; the workgroup whose add returned n_col - 1 loads after a barrier that lane has joined
    // (MI355X_MICROARCH.md, hand-offs with sc1 loads in place of the acquire, first row).
    int* s_last = reinterpret_cast<int*>(smem + 8192);          // behind sPart (<= 3 KiB); the ring is free
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0)
        *s_last = __hip_atomic_fetch_add(&sm.counters[by], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(n_col - 1);
    __syncthreads();
    if (!*s_last) return;
    const int N = sm.N, n_parts = H / 128;
    const int s_end = min(row0 + BM, n_tok) / N;
    const float bias = sm.b2[0];
    for (int s_ = row0 / N + wave; s_ < s_end; s_ += Tile::NW) {            // one wave per sample; N <= 256
        float x[4];
        float mx = -INFINITY;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int t = e * 64 + lane;
            x[e] = -INFINITY;
            if (t < N) {
                const size_t idx = (size_t)s_ * N + t;
                float v = bias;
                for (int p0 = 0; p0 < n_parts; p0 += 8) {        // other workgroups wrote these: `sc1` loads, 8 in flight
                    float pv[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q)
                        pv[q] = p0 + q < n_parts ? __hip_atomic_load(logit_part + (size_t)(p0 + q) * n_tok + idx, __ATOMIC_RELAXED,
                                                                     __HIP_MEMORY_SCOPE_AGENT) : 0.f;
#pragma unroll
                    for (int q = 0; q < 8; ++q) v += pv[q];
                }
                if (sm.logits) sm.logits[idx] = v;
                if (sm.mask && sm.mask[idx] == 0.f) v = NR_NEG_BIG;          // masked_fill_(-9e15)
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
        const float inv = 1.0f / sum;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int t = e * 64 + lane;
            if (t < N) sm.w[(size_t)s_ * N + t] = x[e] * inv;
        }
    }
    if (tid == 0) __hip_atomic_store(&sm.counters[by], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int MI, int NI, int WC, bool X3, int STAGES>
__global__ __launch_bounds__(128 * WC) void nr_mlp_kernel(NrMlpProblem q) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    nr_mlp_body<MI, NI, WC, X3, STAGES>(q, blockIdx.x, smem);
}

// TWO scorer calls of one block shape in one grid (the step's text and video tokens): workgroups [0, grid_a) take `a`, the
// rest `b`.  Pays from a few workgroups per CU on -- configs[3] (512 + 512 workgroups) 517 -> 530 steps/s, configs[2] 424 -> 427;
// at configs[1] (256 + 128, one per CU with a two-deep ring) the two launches one after the other are FASTER (3725 vs 3605
// steps/s, also with a one-deep ring for the pair): the host pairs only large token sets (head.PAIR_BATCH_SCORERS_FROM).
template <int MI, int NI, int WC, bool X3, int STAGES>
__global__ __launch_bounds__(128 * WC) void nr_mlp_pair_kernel(NrMlpProblem a, NrMlpProblem b, int grid_a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if ((int)blockIdx.x < grid_a) nr_mlp_body<MI, NI, WC, X3, STAGES>(a, blockIdx.x, smem);
    else nr_mlp_body<MI, NI, WC, X3, STAGES>(b, (int)blockIdx.x - grid_a, smem);
}

// ---- ALL scorer calls of a step in one grid ---------------------------------------------------------------------------------
// Up to NR_MLP_GROUP_MAX token sets, each in its own precision, on ONE block shape (192 x 256 hidden units on 8 waves, two-deep
// one-pass ring); the split-bf16 sets run three accumulated passes.  Four launches of eight K slices each pay four starts and
// four epilogues (about a third of a launch: nr_mlp.hip header of this round's notes) and leave CUs idle behind the small
// sets; one grid pays them once and keeps every CU fed.  Workgroups of the split-bf16 sets come FIRST (three passes each).
#define NR_MLP_GROUP_MAX 4
struct NrMlpGroup {
    NrMlpProblem p[NR_MLP_GROUP_MAX];
    int x3[NR_MLP_GROUP_MAX];
    int start[NR_MLP_GROUP_MAX + 1];       // first workgroup of every problem (multiples of 8: the XCD order of a problem's tiles)
    int n;
};

template <int MI, int NI>
__global__ __launch_bounds__(512) void nr_mlp_group_kernel(NrMlpGroup g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int gi = 0;
#pragma unroll
    for (int i = 1; i < NR_MLP_GROUP_MAX; ++i)
        if (i < g.n && (int)blockIdx.x >= g.start[i]) gi = i;
    const int wg = (int)blockIdx.x - g.start[gi];
    if (g.x3[gi]) nr_mlp_body<MI, NI, 4, false, 2, true>(g.p[gi], wg, smem);
    else nr_mlp_body<MI, NI, 4, false, 2, false>(g.p[gi], wg, smem);
}

namespace {
struct MlpShape { int mi, ni, wc; };

template <int MI, int NI, int WC, bool X3, int STAGES>
int mlp_launch(const NrMlpProblem& q, hipStream_t st, const NrMlpProblem* second = nullptr) {
    using Tile = NrGemmTile<MI, NI, X3, 16, 16, STAGES, WC>;
    size_t lds = Tile::RING_BYTES;
    const size_t epi = 8192 + 16;                                 // sPart (<= 3 KiB) + the last-block flag at 8192
    static_assert((size_t)WC * Tile::BM * sizeof(float) <= 8192, "the flag sits behind sPart");
    if (lds < epi) lds = epi;
    const int grid_a = nr_xcd_chunk_grid((q.H / Tile::BN) * ((q.n_tok + Tile::BM - 1) / Tile::BM));
    if (second) {
        // (pairs pay only for crowded grids -- a one-deep ring on 4-wave blocks: the other forms are not built)
        if constexpr (!(WC == 2 && STAGES == 1)) return NR_EUNSUPPORTED;
        else {
        auto kern = nr_mlp_pair_kernel<MI, NI, WC, X3, STAGES>;
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return (int)e;
        }
        const int grid_b = nr_xcd_chunk_grid((second->H / Tile::BN) * ((second->n_tok + Tile::BM - 1) / Tile::BM));
        hipLaunchKernelGGL(kern, dim3(grid_a + grid_b), dim3(128 * WC), lds, st, q, *second, grid_a);
        NR_LAUNCH_CHECK();
        return NR_OK;
        }
    }
    auto kern = nr_mlp_kernel<MI, NI, WC, X3, STAGES>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(kern, dim3(grid_a), dim3(128 * WC), lds, st, q);
    NR_LAUNCH_CHECK();
    return NR_OK;
}
}  // namespace

// block shape (index into `cand`) and workgroup count of one scorer call; -1: none fits
static const MlpShape nr_mlp_cand[] = {{2, 4, 2}, {3, 4, 2}, {4, 4, 2}, {4, 4, 4}, {6, 4, 4}};

static int nr_mlp_pick(int n_tok, int H, bool x3, const NrMlpSoftmax& sm, long* wg_out) {
    // candidate block shapes (MI, NI, WC): 64/96/128 x 128 on 4 waves, 128/192 x 256 on 8 waves (one-pass bf16
    // only: the split operands of a 256-wide block do not fit the LDS twice)
    const MlpShape* cand = nr_mlp_cand;
    int best = -1;
    long best_cost = 0, best_wg = 0;
    for (int c = 0; c < 5; ++c) {
        const int bm = 32 * cand[c].mi, bn = 16 * cand[c].wc * cand[c].ni;
        if (H % bn) continue;
        if (x3 && cand[c].wc == 4) continue;
        if (sm.counters && (bm % sm.N) != 0) continue;            // fused softmax: whole samples per row tile
        const long wg = (long)((n_tok + bm - 1) / bm) * (H / bn);
        const long cost = ((wg + 255) / 256) * (bm + bn);
        if (best < 0 || cost < best_cost || (cost == best_cost && wg > best_wg)) { best = c; best_cost = cost; best_wg = wg; }
    }
    // Crowded grids of the 192 x 256 block (>= 3 workgroups per CU): its 142 registers a lane keep every CU at ONE 8-wave
    // workgroup, while two of the 128 x 256 block (108) share a CU and hide each other's load phases with a one-deep ring --
    // configs[3], 65536 bank tokens per scorer launch: 485 steps/s with 192 x 256 one-deep, 497 two-deep, 515 with 128 x 256
    // one-deep (tools/ab_c3.sh, A/B in one session)
    if (best == 4 && best_wg >= 3 * 256 && H % 256 == 0 && !(sm.counters && 128 % sm.N != 0)) {
        best = 3;
        best_wg = (long)((n_tok + 127) / 128) * (H / 256);
    }
    if (const char* e = nr_tune_env("NR_MLP_SHAPE")) {          // tuning hook: index into the candidate list
        int c = atoi(e);
        if (c >= 0 && c < 5 && H % (16 * cand[c].wc * cand[c].ni) == 0 && !(x3 && cand[c].wc == 4) &&
            !(sm.counters && (32 * cand[c].mi) % sm.N != 0)) {
            best = c;
            best_wg = (long)((n_tok + 32 * cand[c].mi - 1) / (32 * cand[c].mi)) * (H / (16 * cand[c].wc * cand[c].ni));
        }
    }
    if (wg_out) *wg_out = best_wg;
    return best;
}

static int nr_mlp_check(const NrMlpProblem& q, int prec) {
    if (!q.tok_hi || !q.norm || !q.w1_hi || !q.b1 || !q.w2 || !q.logit_part) return NR_EINVAL;
    if (q.n_tok <= 0 || q.d <= 0 || (q.d % 64) != 0 || q.H <= 0 || (q.H % 128) != 0) return NR_EINVAL;
    if (prec != NR_PREC_BF16 && prec != NR_PREC_BF16X3) return NR_EINVAL;
    if (prec == NR_PREC_BF16X3 && (!q.tok_lo || !q.w1_lo)) return NR_EINVAL;
    return NR_OK;
}

// one scorer call, or two of one block shape in one grid (`second`; NR_EUNSUPPORTED when their shapes differ)
static int nr_mlp_go(const NrMlpProblem& q, const NrMlpProblem* second, int prec, void* stream) {
    int rc = nr_mlp_check(q, prec);
    if (rc == NR_OK && second) rc = nr_mlp_check(*second, prec);
    if (rc != NR_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    const bool x3 = prec == NR_PREC_BF16X3;
    const MlpShape* cand = nr_mlp_cand;
    long best_wg = 0;
    const int best = nr_mlp_pick(q.n_tok, q.H, x3, q.sm, &best_wg);
    if (best < 0) return NR_EUNSUPPORTED;
    if (second) {
        long wg_b = 0;
        if (nr_mlp_pick(second->n_tok, second->H, x3, second->sm, &wg_b) != best || second->d != q.d) return NR_EUNSUPPORTED;
        best_wg += wg_b;
    }
    // ring depth: workgroups that sit alone on their CU prefetch for themselves (2 stages); crowded grids run 1
    // (the 192 x 256 block holds 142 registers a lane: never two of its 8-wave workgroups on a CU, however crowded the grid --
    // one-stage launches of it ran load and MFMA phases strictly in turn: 150 us for the 65536 bank tokens of configs[3])
    bool two = best_wg < 3 * 256 || (cand[best].mi == 6 && cand[best].wc == 4);
    if (nr_tune_env("NR_MLP_ONE_STAGE")) two = false;          // tuning hook: smallest LDS footprint
#define NR_MLP_GO(MI_, NI_, WC_)                                                                                              \
    if (cand[best].mi == MI_ && cand[best].ni == NI_ && cand[best].wc == WC_) {                                               \
        if (x3) {                                                                                                             \
            if constexpr (WC_ == 2) {                                                                                         \
                return two ? mlp_launch<MI_, NI_, WC_, true, 2>(q, st, second) : mlp_launch<MI_, NI_, WC_, true, 1>(q, st, second); \
            }                                                                                                                 \
        } else {                                                                                                              \
            return two ? mlp_launch<MI_, NI_, WC_, false, 2>(q, st, second) : mlp_launch<MI_, NI_, WC_, false, 1>(q, st, second);   \
        }                                                                                                                     \
    }
    NR_MLP_GO(2, 4, 2) NR_MLP_GO(3, 4, 2) NR_MLP_GO(4, 4, 2) NR_MLP_GO(4, 4, 4) NR_MLP_GO(6, 4, 4)
#undef NR_MLP_GO
    return NR_EUNSUPPORTED;
}

static int nr_token_logits_go(const uint16_t* tok_hi, const uint16_t* tok_lo, const float* norm, int n_tok, int d,
                              const uint16_t* w1_hi, const uint16_t* w1_lo, const float* b1, const float* w2,
                              int H, int prec, float* logit_part, void* stream, const NrMlpSoftmax& sm) {
    NrMlpProblem q{tok_hi, tok_lo, norm, n_tok, d, w1_hi, w1_lo, b1, w2, H, logit_part, sm};
    return nr_mlp_go(q, nullptr, prec, stream);
}

extern "C" int nr_token_logits_fwd(const uint16_t* tok_hi, const uint16_t* tok_lo, const float* norm, int n_tok, int d,
                                   const uint16_t* w1_hi, const uint16_t* w1_lo, const float* b1, const float* w2,
                                   int H, int prec, float* logit_part, void* stream) {
    return nr_token_logits_go(tok_hi, tok_lo, norm, n_tok, d, w1_hi, w1_lo, b1, w2, H, prec, logit_part, stream, NrMlpSoftmax{});
}

// Scorer MLP AND the masked softmax over each sample's tokens in ONE launch (see NrMlpSoftmax).  n_row_tiles_max: length
// of `counters` (zeroed by the caller once; the kernel leaves it zeroed).  NR_EUNSUPPORTED: no block shape holds whole
// samples (the caller then issues nr_token_logits_fwd + nr_token_softmax).
extern "C" int nr_token_weights_fwd(const uint16_t* tok_hi, const uint16_t* tok_lo, const float* norm, int n_samples, int N, int d,
                                    const uint16_t* w1_hi, const uint16_t* w1_lo, const float* b1, const float* w2, const float* b2,
                                    int H, int prec, const float* mask, float* logit_part, unsigned int* counters, int n_counters,
                                    float* w, float* logits, void* stream) {
    if (!counters || !b2 || !w || n_samples <= 0 || N <= 0) return NR_EINVAL;
    if (N > 256) return NR_EUNSUPPORTED;
    const long n_tok = (long)n_samples * N;
    if (n_counters < (n_tok + 63) / 64) return NR_EINVAL;           // the smallest row tile is 64 tokens
    NrMlpSoftmax sm{counters, b2, mask, N, w, logits};
    return nr_token_logits_go(tok_hi, tok_lo, norm, (int)n_tok, d, w1_hi, w1_lo, b1, w2, H, prec, logit_part, stream, sm);
}

// Two nr_token_weights_fwd calls of one precision in ONE launch (the step's text and video tokens).  NR_EUNSUPPORTED when the
// two do not run the same block shape (the caller then issues them one by one).
extern "C" int nr_token_weights_fwd_pair(const NrTokenWeightsProblem* a, const NrTokenWeightsProblem* b, int prec, void* stream) {
    if (!a || !b) return NR_EINVAL;
    NrMlpProblem q[2];
    const NrTokenWeightsProblem* src[2] = {a, b};
    for (int i = 0; i < 2; ++i) {
        const NrTokenWeightsProblem& t = *src[i];
        if (!t.counters || !t.b2 || !t.w || t.n_samples <= 0 || t.N <= 0) return NR_EINVAL;
        if (t.N > 256) return NR_EUNSUPPORTED;
        const long n_tok = (long)t.n_samples * t.N;
        if (t.n_counters < (n_tok + 63) / 64) return NR_EINVAL;
        q[i] = NrMlpProblem{t.tok_hi, t.tok_lo, t.norm, (int)n_tok, t.d, t.w1_hi, t.w1_lo, t.b1, t.w2, t.H, t.logit_part,
                            NrMlpSoftmax{t.counters, t.b2, t.mask, t.N, t.w, t.logits}};
    }
    if (a->counters == b->counters) return NR_EINVAL;               // (the two problems' row tiles count separately)
    return nr_mlp_go(q[0], &q[1], prec, stream);
}

// nr_token_weights_fwd for up to four token sets, each in its own precision, in ONE launch (192 x 256 blocks; see
// nr_mlp_group_kernel).  NR_EUNSUPPORTED when a set does not fit that block (H % 256, 192 % N, missing lo halves): the caller
// issues the sets one by one.  Split-bf16 sets come out of three accumulated passes (another summation order than the
// single-launch split tile: equal to ~1e-7 relative, not bit for bit); one-pass sets are bit-identical to their single launch.
extern "C" int nr_token_weights_fwd_group(const NrTokenWeightsProblem* probs, const int* precs, int n, void* stream) {
    if (!probs || !precs || n <= 0 || n > NR_MLP_GROUP_MAX) return NR_EINVAL;
    constexpr int MI = 6, NI = 4;
    using Tile = NrGemmTile<MI, NI, false, 16, 16, 2, 4>;
    NrMlpGroup g;
    g.n = n;
    int order[NR_MLP_GROUP_MAX], k = 0;
    for (int pass = 0; pass < 2; ++pass)                       // split-bf16 sets first
        for (int i = 0; i < n; ++i)
            if ((precs[i] == NR_PREC_BF16X3) == (pass == 0)) order[k++] = i;
    int total = 0;
    for (int j = 0; j < n; ++j) {
        const NrTokenWeightsProblem& t = probs[order[j]];
        const int prec = precs[order[j]];
        if (!t.counters || !t.b2 || !t.w || t.n_samples <= 0 || t.N <= 0) return NR_EINVAL;
        if (t.N > 256 || (Tile::BM % t.N) != 0 || (t.H % Tile::BN) != 0) return NR_EUNSUPPORTED;
        const long n_tok = (long)t.n_samples * t.N;
        if (t.n_counters < (n_tok + 63) / 64) return NR_EINVAL;
        NrMlpProblem q{t.tok_hi, t.tok_lo, t.norm, (int)n_tok, t.d, t.w1_hi, t.w1_lo, t.b1, t.w2, t.H, t.logit_part,
                       NrMlpSoftmax{t.counters, t.b2, t.mask, t.N, t.w, t.logits}};
        int rc = nr_mlp_check(q, prec);
        if (rc != NR_OK) return rc;
        for (int j2 = 0; j2 < j; ++j2)
            if (g.p[j2].sm.counters == t.counters) return NR_EINVAL;      // (the problems' row tiles count separately)
        g.p[j] = q;
        g.x3[j] = prec == NR_PREC_BF16X3 ? 1 : 0;
        g.start[j] = total;
        total += nr_xcd_chunk_grid((t.H / Tile::BN) * (int)((n_tok + Tile::BM - 1) / Tile::BM));
    }
    for (int j = n; j <= NR_MLP_GROUP_MAX; ++j) g.start[j] = total;
    size_t lds = Tile::RING_BYTES;
    if (lds < 8192 + 16) lds = 8192 + 16;
    static_assert((size_t)4 * Tile::BM * sizeof(float) <= 8192, "the flag sits behind sPart");
    auto kern = nr_mlp_group_kernel<MI, NI>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(kern, dim3(total), dim3(512), lds, (hipStream_t)stream, g);
    NR_LAUNCH_CHECK();
    return NR_OK;
}

// ---- backward of the scorer MLP, hidden layer (no counterpart in the reference: autograd differentiates modeling.py:148-153) ----
// Recomputes h = ||x|| (x_hat W1^T) + b1 exactly as the forward did (same tile engine, same precision plan) and turns the
// upstream gradient of the logits into the gradient of the hidden layer IN THE EPILOGUE -- the [n_tok, H] activations never
// exist in fp32, and nothing element-wise is left for the host:
//     dh[t, c]  = h > 0 ? dl[t] * w2[c] : 0          -> bf16 pair, TRANSPOSED [H, ldT] at column t0 + t (operand of dW1 = dh^T X,
//                                                       K = tokens) and, for the rows that need dX = dh W1, row-major [n_tok, H]
//     dW2[c]    = sum_t dl[t] * relu(h[t, c]),   db1[c] = sum_t dh[t, c]   as per-wave-row partial sums [2 * row tiles, H]
template <int MI, int NI, int WC, bool X3, int STAGES, bool LO_OUT = true>
__global__ __launch_bounds__(128 * WC) void nr_mlp_bwd_hidden_kernel(const uint16_t* __restrict__ tok_hi, const uint16_t* __restrict__ tok_lo,
                                                                     const float* __restrict__ norm, int n_tok, int d,
                                                                     const uint16_t* __restrict__ w1_hi, const uint16_t* __restrict__ w1_lo,
                                                                     const float* __restrict__ b1, const float* __restrict__ w2, int H,
                                                                     const float* __restrict__ dl, uint16_t* __restrict__ dhT_hi,
                                                                     uint16_t* __restrict__ dhT_lo, int ldT, int t0,
                                                                     uint16_t* __restrict__ dh_hi, uint16_t* __restrict__ dh_lo,
                                                                     float* __restrict__ dw2_part, float* __restrict__ db1_part,
                                                                     float* __restrict__ dl_part) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using Tile = NrGemmTile<MI, NI, X3, 16, 16, STAGES, WC>;
    constexpr int BM = Tile::BM, BN = Tile::BN;
    constexpr int WCOLS = 16 * NI;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave / WC, wc = wave % WC;
    const int n_col = H / BN;
    const int tile_id = nr_xcd_chunk_tile(blockIdx.x, n_col * ((n_tok + BM - 1) / BM));
    if (tile_id < 0) return;
    const int bx = tile_id % n_col, by = tile_id / n_col;
    const int row0 = by * BM, col0 = bx * BN;

    Tile tile;
    tile.zero();
    tile.run(tok_hi, tok_lo, row0, n_tok, w1_hi, w1_lo, col0, H, d, smem);

    float s_dl = 0.f;
    float s_w2[NI], s_b1[NI];
#pragma unroll
    for (int n = 0; n < NI; ++n) s_w2[n] = s_b1[n] = 0.f;
    // dh of this lane's 4 x MI x NI accumulator entries as bf16 pairs: [m][n] = rows rb(m) .. rb(m)+3 of column c(n)
    // LO_OUT = false (only the hi halves of dh^T are wanted): the packed values go straight into the LDS image, no register copy
    uint2 ph[LO_OUT ? MI : 1][LO_OUT ? NI : 1], pl[LO_OUT ? MI : 1][LO_OUT ? NI : 1];
    uint16_t* sT = reinterpret_cast<uint16_t*>(smem);
    constexpr int TLD = BM + 8, RLD = BN + 8;                            // row pitches of the two images (16-byte multiples)
#pragma unroll
    for (int m = 0; m < MI; ++m) {
        const int rb = row0 + wr * 16 * MI + m * 16 + (lane >> 4) * 4;          // this lane's 4 consecutive rows
        float sc[4], dlr[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool on = rb + j < n_tok;
            sc[j] = on ? norm[rb + j] : 0.f;
            dlr[j] = on ? dl[rb + j] : 0.f;
            s_dl += dlr[j];
        }
#pragma unroll
        for (int n = 0; n < NI; ++n) {
            const int c = col0 + wc * WCOLS + n * 16 + (lane & 15);
            const float bb = b1[c], ww = w2[c];
            uint16_t hb[4], lb[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float h = tile.acc[m][n][j] * sc[j] + bb;
                const float dh = h > 0.f ? dlr[j] * ww : 0.f;                   // rows past n_tok: dl = 0 -> 0
                s_w2[n] += dlr[j] * fmaxf(h, 0.f);
                s_b1[n] += dh;
                hb[j] = nr_f2bf(dh);
                lb[j] = nr_f2bf(dh - nr_bf2f(hb[j]));
            }
            const uint2 packed_hi = uint2{(uint32_t)hb[0] | ((uint32_t)hb[1] << 16), (uint32_t)hb[2] | ((uint32_t)hb[3] << 16)};
            if constexpr (LO_OUT) {
                ph[m][n] = packed_hi;
                pl[m][n] = uint2{(uint32_t)lb[0] | ((uint32_t)lb[1] << 16), (uint32_t)lb[2] | ((uint32_t)lb[3] << 16)};
            } else {
                const int cl = wc * WCOLS + n * 16 + (lane & 15), tl = wr * 16 * MI + m * 16 + (lane >> 4) * 4;
                *reinterpret_cast<uint2*>(sT + cl * TLD + tl) = packed_hi;       // (the ring is free: run() ended with a barrier)
            }
        }
    }
    // The tile leaves through LDS (the ring is free: run() ended with a barrier), one image at a time, so that global memory
    // sees whole 16-byte pieces of contiguous rows: the TRANSPOSED pair [unit][token] (a unit's 128 tokens = 256 contiguous
    // bytes of its row of dhT; written straight from the accumulator layout they were 8-byte pieces at a row pitch of tens of
    // KB, and this kernel spent more time storing than multiplying), then -- for the token set whose dX is wanted -- the
    // row-major pair [token][unit].  Columns [n_tok, n_tok rounded up to 64) of dhT get the zeros they need as K padding.
    const int lim = min((n_tok + 63) / 64 * 64, ldT - t0) - row0;        // tokens of this tile that belong to the set's columns
    auto flush = [&](uint16_t* __restrict__ dst, const bool transposed) {
        __syncthreads();                                                 // image complete
        if (transposed) {                                                // [BN units][BM tokens]
            for (int e = threadIdx.x; e < BN * (BM / 8); e += 128 * WC) {
                const int r = e / (BM / 8), k = e - r * (BM / 8);        // unit, 16-byte piece (8 tokens)
                const uint4 v = *reinterpret_cast<const uint4*>(sT + r * TLD + 8 * k);
                if (8 * k < lim) *reinterpret_cast<uint4*>(dst + (size_t)(col0 + r) * ldT + t0 + row0 + 8 * k) = v;
            }
        } else {                                                         // [BM tokens][BN units]
            for (int e = threadIdx.x; e < BM * (BN / 8); e += 128 * WC) {
                const int r = e / (BN / 8), k = e - r * (BN / 8);
                const uint4 v = *reinterpret_cast<const uint4*>(sT + r * RLD + 8 * k);
                if (row0 + r < n_tok) *reinterpret_cast<uint4*>(dst + (size_t)(row0 + r) * H + col0 + 8 * k) = v;
            }
        }
        __syncthreads();                                                 // image consumed
    };
    auto image_t = [&](const uint2 (&p)[MI][NI]) {
#pragma unroll
        for (int m = 0; m < MI; ++m)
#pragma unroll
            for (int n = 0; n < NI; ++n) {
                const int cl = wc * WCOLS + n * 16 + (lane & 15), tl = wr * 16 * MI + m * 16 + (lane >> 4) * 4;
                *reinterpret_cast<uint2*>(sT + cl * TLD + tl) = p[m][n];
            }
    };
    auto image_r = [&](const uint2 (&p)[MI][NI]) {
#pragma unroll
        for (int m = 0; m < MI; ++m)
#pragma unroll
            for (int n = 0; n < NI; ++n) {
                const int cl = wc * WCOLS + n * 16 + (lane & 15), tl = wr * 16 * MI + m * 16 + (lane >> 4) * 4;
                sT[(tl + 0) * RLD + cl] = (uint16_t)(p[m][n].x & 0xffffu);
                sT[(tl + 1) * RLD + cl] = (uint16_t)(p[m][n].x >> 16);
                sT[(tl + 2) * RLD + cl] = (uint16_t)(p[m][n].y & 0xffffu);
                sT[(tl + 3) * RLD + cl] = (uint16_t)(p[m][n].y >> 16);
            }
    };
    if constexpr (LO_OUT) image_t(ph);
    flush(dhT_hi, true);
    if constexpr (LO_OUT) {
        if (dhT_lo) {
            image_t(pl);
            flush(dhT_lo, true);
        }
        if (dh_hi) {
            image_r(ph);
            flush(dh_hi, false);
            image_r(pl);
            flush(dh_lo, false);
        }
    }
    // column sums over this wave's 16 * MI rows: lanes l, l+16, l+32, l+48 hold the same column
#pragma unroll
    for (int n = 0; n < NI; ++n) {
        float a = s_w2[n], b = s_b1[n];
        a += __shfl_xor(a, 16); a += __shfl_xor(a, 32);
        b += __shfl_xor(b, 16); b += __shfl_xor(b, 32);
        if (lane < 16) {
            const int c = col0 + wc * WCOLS + n * 16 + lane;
            dw2_part[(size_t)(2 * by + wr) * H + c] = a;
            db1_part[(size_t)(2 * by + wr) * H + c] = b;
        }
    }
    // d b2 = sum of dl: the first column block's first wave column adds up its rows (lanes 0, 16, 32, 48 hold distinct rows)
    if (dl_part && bx == 0 && wc == 0) {
        float a = (lane & 15) == 0 ? s_dl : 0.f;
        a += __shfl_xor(a, 16); a += __shfl_xor(a, 32);
        if (lane == 0) dl_part[2 * by + wr] = a;
    }
}

namespace {
template <int MI, int NI, int WC, bool X3, int STAGES, bool LO_OUT = true>
int mlp_bwd_launch(const uint16_t* tok_hi, const uint16_t* tok_lo, const float* norm, int n_tok, int d, const uint16_t* w1_hi,
                   const uint16_t* w1_lo, const float* b1, const float* w2, int H, const float* dl, uint16_t* dhT_hi, uint16_t* dhT_lo,
                   int ldT, int t0, uint16_t* dh_hi, uint16_t* dh_lo, float* dw2_part, float* db1_part, float* dl_part, hipStream_t st) {
    using Tile = NrGemmTile<MI, NI, X3, 16, 16, STAGES, WC>;
    // the epilogue's LDS images: [BN][BM + 8] (transposed) and, only with dh_hi, [BM][BN + 8] (row-major)
    size_t image = (size_t)Tile::BN * (Tile::BM + 8) * sizeof(uint16_t);
    if (dh_hi && (size_t)Tile::BM * (Tile::BN + 8) * sizeof(uint16_t) > image) image = (size_t)Tile::BM * (Tile::BN + 8) * sizeof(uint16_t);
    const size_t lds = Tile::RING_BYTES > image ? Tile::RING_BYTES : image;
    if (lds > 160 * 1024) return NR_EUNSUPPORTED;
    if (!LO_OUT && (dhT_lo || dh_hi)) return NR_EINVAL;
    auto kern = nr_mlp_bwd_hidden_kernel<MI, NI, WC, X3, STAGES, LO_OUT>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    dim3 grid(nr_xcd_chunk_grid((H / Tile::BN) * ((n_tok + Tile::BM - 1) / Tile::BM)));
    hipLaunchKernelGGL(kern, grid, dim3(128 * WC), lds, st, tok_hi, tok_lo, norm, n_tok, d, w1_hi, w1_lo, b1, w2, H, dl, dhT_hi, dhT_lo,
                       ldT, t0, dh_hi, dh_lo, dw2_part, db1_part, dl_part);
    NR_LAUNCH_CHECK();
    return NR_OK;
}
}  // namespace

// Block of nr_token_mlp_bwd_hidden for a token set: 192 x 256 on 8 waves for the large one-pass sets (the memory bank: one
// block per CU at 12 288 tokens, 43 -> 25 us against 128 x 128 on 4 waves), 128 x 128 on 4 waves otherwise.
static bool mlp_bwd_big(int n_tok, int H, int prec, bool hi_only) { return hi_only && prec == NR_PREC_BF16 && n_tok >= 4096 && (H % 256) == 0; }

// Rows of dw2_part / db1_part / dl_part that nr_token_mlp_bwd_hidden writes for n_tok tokens in precision `prec` with H hidden
// units: two (wave rows) per row block of the variant it will run (hi_only: the call will pass dhT_lo = dh_hi = dh_lo = NULL).
extern "C" int nr_token_mlp_bwd_part_rows(int n_tok, int H, int prec, int hi_only) {
    if (n_tok <= 0) return 0;
    const int bm = mlp_bwd_big(n_tok, H, prec, hi_only != 0) ? 192 : 128;
    return 2 * ((n_tok + bm - 1) / bm);
}

// (kept: the row blocks of the 128-row variant; nr_token_mlp_bwd_part_rows is what sizes the partial-sum buffers)
extern "C" int nr_token_mlp_bwd_row_tiles(int n_tok) { return n_tok > 0 ? (n_tok + 127) / 128 : 0; }

extern "C" int nr_token_mlp_bwd_hidden(const uint16_t* tok_hi, const uint16_t* tok_lo, const float* norm, int n_tok, int d,
                                       const uint16_t* w1_hi, const uint16_t* w1_lo, const float* b1, const float* w2, int H, int prec,
                                       const float* dl, uint16_t* dhT_hi, uint16_t* dhT_lo, int ldT, int t0, uint16_t* dh_hi,
                                       uint16_t* dh_lo, float* dw2_part, float* db1_part, float* dl_part, void* stream) {
    if (!tok_hi || !norm || !w1_hi || !b1 || !w2 || !dl || !dhT_hi || !dw2_part || !db1_part) return NR_EINVAL;
    if ((dh_hi == nullptr) != (dh_lo == nullptr)) return NR_EINVAL;
    if (n_tok <= 0 || d <= 0 || (d % 64) != 0 || H <= 0 || (H % 128) != 0 || t0 < 0 || ldT < t0 + n_tok) return NR_EINVAL;
    if ((t0 % 8) != 0 || (ldT % 8) != 0) return NR_EUNSUPPORTED;          // 16-byte pieces of the transposed rows
    if (prec != NR_PREC_BF16 && prec != NR_PREC_BF16X3) return NR_EINVAL;
    if (prec == NR_PREC_BF16X3 && (!tok_lo || !w1_lo)) return NR_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    // 128 x 128 blocks on 4 waves (the partial-sum layout is tied to BM = 128); two-deep ring unless the grid is crowded
    const long wg = (long)((n_tok + 127) / 128) * (H / 128);
    const bool two = wg < 3 * 256;
    if (prec == NR_PREC_BF16X3)
        return two ? mlp_bwd_launch<4, 4, 2, true, 2>(tok_hi, tok_lo, norm, n_tok, d, w1_hi, w1_lo, b1, w2, H, dl, dhT_hi, dhT_lo, ldT, t0, dh_hi, dh_lo, dw2_part, db1_part, dl_part, st)
                   : mlp_bwd_launch<4, 4, 2, true, 1>(tok_hi, tok_lo, norm, n_tok, d, w1_hi, w1_lo, b1, w2, H, dl, dhT_hi, dhT_lo, ldT, t0, dh_hi, dh_lo, dw2_part, db1_part, dl_part, st);
    if (mlp_bwd_big(n_tok, H, prec, !dhT_lo && !dh_hi))
        return mlp_bwd_launch<6, 4, 4, false, 2, false>(tok_hi, tok_lo, norm, n_tok, d, w1_hi, w1_lo, b1, w2, H, dl, dhT_hi, dhT_lo, ldT, t0, dh_hi, dh_lo, dw2_part, db1_part, dl_part, st);
    // one pass: the two-deep ring is 64 KB, and the 256 registers of the epilogue allow two workgroups per CU either way
    if (nr_tune_env("NR_MLP_BWD_ONE_STAGE") == nullptr)
        return mlp_bwd_launch<4, 4, 2, false, 2>(tok_hi, tok_lo, norm, n_tok, d, w1_hi, w1_lo, b1, w2, H, dl, dhT_hi, dhT_lo, ldT, t0, dh_hi, dh_lo, dw2_part, db1_part, dl_part, st);
    return two ? mlp_bwd_launch<4, 4, 2, false, 2>(tok_hi, tok_lo, norm, n_tok, d, w1_hi, w1_lo, b1, w2, H, dl, dhT_hi, dhT_lo, ldT, t0, dh_hi, dh_lo, dw2_part, db1_part, dl_part, st)
               : mlp_bwd_launch<4, 4, 2, false, 1>(tok_hi, tok_lo, norm, n_tok, d, w1_hi, w1_lo, b1, w2, H, dl, dhT_hi, dhT_lo, ldT, t0, dh_hi, dh_lo, dw2_part, db1_part, dl_part, st);
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
