// Y[M,N] = X[M,K] * W[N,K]^T (+ bias[N]) (+ residual[M,N]) in split-bf16 on the MFMA tile engine:
// the fp32 GEMMs of the token-clustering stage (k=3 token convolution as a [B*N, 3C] x [3C, C] product,
// cluster.py:664; the kv projection, cluster.py:866) at ~fp32 accuracy (operands carried as bf16
// hi + lo, products Ah*Bh + Ah*Bl + Al*Bh, fp32 accumulate) and bf16-pipe speed.
// X arrives already split (nr_shift_concat_split / nr_ctm_front write hi/lo directly), W is split once
// per parameter version by the host.
#include <cstdio>
#include <cstdlib>
#include "nr_gemm_tile.h"
#include "nr_linear.h"
#include "nr_ctm_bodies.h"
#include "../../include/nr_hip.h"

struct NrLinearGroup {
    NrLinearArgs p[NR_LINEAR_MAX_GROUP];
    int tile_start[NR_LINEAR_MAX_GROUP + 1];    // first workgroup of every problem
    int n;
};

template <int MI, int NI, int STAGES, int WC, bool CONV = false, bool X3 = true>
__device__ __forceinline__ void nr_linear_tile(const NrLinearArgs& p, const int tile_row, const int tile_col, char* smem) {
    using Tile = NrGemmTile<MI, NI, X3, 16, 16, STAGES, WC>;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave / WC, wc = wave % WC;
    const int row0 = tile_row * Tile::BM, col0 = tile_col * Tile::BN;
    Tile tile;
    tile.zero();
    if constexpr (CONV) tile.run_conv3(p.x_hi, p.x_lo, row0, p.M, p.w_hi, p.w_lo, col0, p.N, p.K, smem, p.conv_n);
#ifdef NR_TUNE
    else if constexpr (WC == 4 && STAGES == 2 && X3) {
        // tuning builds: 8-wave split-bf16 blocks on a two-deep ring walk K with the ping-pong loop of the similarity kernel.
        // Measured (tools/probe_linear_tiles.sh): the token convolution's shape on 128 x 128 blocks 38.5 us against 39.4 on the
        // shipped 64 x 64 one-deep blocks (144 workgroups for 256 CUs), the kv shape 35.3 against 27.7 -- not shipped
        if (p.ld == 0 || p.ld == p.K) tile.run_pp(p.x_hi, p.x_lo, row0, p.M, p.w_hi, p.w_lo, col0, p.N, p.K, smem);
        else tile.run(p.x_hi, p.x_lo, row0, p.M, p.w_hi, p.w_lo, col0, p.N, p.K, smem, 0, false, p.ld);
    }
#endif
    else tile.run(p.x_hi, p.x_lo, row0, p.M, p.w_hi, p.w_lo, col0, p.N, p.K, smem, 0, false, p.ld);
#pragma unroll
    for (int n = 0; n < NI; ++n) {
        const int c = col0 + wc * 16 * NI + n * 16 + (lane & 15);
        if (c >= p.N) continue;
        const float bv = p.bias ? p.bias[c] : 0.f;
#pragma unroll
        for (int m = 0; m < MI; ++m)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = row0 + wr * 16 * MI + m * 16 + (lane >> 4) * 4 + j;
                if (r >= p.M) continue;
                const size_t o = (size_t)r * p.N + c;
                float v = tile.acc[m][n][j] + bv;
                if (p.residual) v += p.residual[o];
                p.out[o] = v;
                if (p.out_hi) {
                    const uint16_t hb = nr_f2bf(v);
                    p.out_hi[o] = hb;
                    p.out_lo[o] = nr_f2bf(v - nr_bf2f(hb));
                }
            }
    }
}

template <int MI, int NI, int STAGES>
__global__ __launch_bounds__(256) void nr_linear_kernel(NrLinearArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    nr_linear_tile<MI, NI, STAGES, 2>(p, blockIdx.y, blockIdx.x, smem);
}

// grouped: workgroup -> (problem, tile) through the prefix table; tiles of a problem are column-fastest
template <int MI, int NI, int STAGES, int WC, bool CONV = false, bool X3 = true>
__global__ __launch_bounds__(128 * WC) void nr_linear_group_kernel(NrLinearGroup g) {
    NR_CRITICAL_PATH();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int gi = 0;
    // tiles are numbered problem by problem, column-fastest: the column tiles of one row tile (same X rows) are
    // neighbours, and nr_xcd_chunk_tile keeps neighbours on one XCD so that its L2 fetches those rows once
    const int wg = nr_xcd_chunk_tile(blockIdx.x, g.tile_start[NR_LINEAR_MAX_GROUP]);
    if (wg < 0) return;
#pragma unroll
    for (int i = 1; i < NR_LINEAR_MAX_GROUP; ++i)
        if (i < g.n && wg >= g.tile_start[i]) gi = i;
    const NrLinearArgs& p = g.p[gi];
    const int t = wg - g.tile_start[gi];
    const int ncol = (p.N + 16 * WC * NI - 1) / (16 * WC * NI);
    nr_linear_tile<MI, NI, STAGES, WC, CONV, X3>(p, t / ncol, t % ncol, smem);
}

// back workgroups of a clustering stage in front of the GEMM tiles of its kv projection, one grid (nr_linear.h)
template <int MI, int NI, int STAGES, int WC, int BACK_FORM>
__global__ __launch_bounds__(128 * WC, (256 * WC) / 256) void nr_back_beside_linear_kernel(NrGroupOf<NrCtmBackArgs> gb, int n_back_pad, int use_lds,
                                                                                         NrLinearGroup g) {
    NR_CRITICAL_PATH();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if ((int)blockIdx.x < n_back_pad) {
        if ((int)blockIdx.x >= gb.start[NR_CTM_MAX_GROUP]) return;          // padding up to a multiple of 8 (XCD order of the tiles)
        const int gi = gb.find(blockIdx.x);
        float* rows = use_lds ? reinterpret_cast<float*>(smem) : nullptr;
        if constexpr (BACK_FORM == 0) nr_ctm_back_body<false, 128 * WC>(gb.p[gi], blockIdx.x - gb.start[gi], rows);
        else nr_ctm_back_body2<false, 128 * WC, BACK_FORM, 512 / (128 * WC), BACK_FORM <= 4 ? 32 : 64>(gb.p[gi], blockIdx.x - gb.start[gi], rows, nullptr, nullptr);
        return;
    }
    const int wg = nr_xcd_chunk_tile(blockIdx.x - n_back_pad, g.tile_start[NR_LINEAR_MAX_GROUP]);
    if (wg < 0) return;
    int gi = 0;
#pragma unroll
    for (int i = 1; i < NR_LINEAR_MAX_GROUP; ++i)
        if (i < g.n && wg >= g.tile_start[i]) gi = i;
    const NrLinearArgs& p = g.p[gi];
    const int t = wg - g.tile_start[gi];
    const int ncol = (p.N + 16 * WC * NI - 1) / (16 * WC * NI);
    nr_linear_tile<MI, NI, STAGES, WC, false, true>(p, t / ncol, t % ncol, smem);
}

int nr_linear_group_launch_beside_back(const NrLinearArgs* probs, int n, const NrGroupOf<NrCtmBackArgs>& gb, size_t back_lds, int use_lds,
                                       int back_form, hipStream_t st) {
    constexpr int MI = 2, NI = 2, STAGES = 1, WC = 4;
    using Tile = NrGemmTile<MI, NI, true, 16, 16, STAGES, WC>;
    if (!probs || n <= 0 || n > NR_LINEAR_MAX_GROUP) return NR_EINVAL;
    NrLinearGroup g;
    g.n = n;
    int total = 0;
    for (int i = 0; i < n; ++i) {
        const NrLinearArgs& a = probs[i];
        if (!a.x_hi || !a.x_lo || !a.w_hi || !a.w_lo || !a.out || a.M <= 0 || a.N <= 0 || a.K <= 0) return NR_EINVAL;
        if ((a.K % 64) != 0 || a.conv_n != 0 || a.ld != 0 || a.out_hi || a.out_lo) return NR_EUNSUPPORTED;
        g.p[i] = a;
        g.tile_start[i] = total;
        total += ((a.M + Tile::BM - 1) / Tile::BM) * ((a.N + Tile::BN - 1) / Tile::BN);
    }
    for (int i = n; i <= NR_LINEAR_MAX_GROUP; ++i) g.tile_start[i] = total;
    if (back_form != 0 && back_form != 4 && back_form != 16) return NR_EINVAL;
    for (int i = 0; i < gb.n; ++i) {
        if (gb.p[i].N > 64) return NR_EUNSUPPORTED;
        // first form: the 8-wave body spreads cnum * C/128 merge jobs over 8 waves; second form: C = 512, cnum <= back_form
        if (back_form == 0 ? gb.p[i].cnum * (gb.p[i].C / 128) > 8 * BK_MAXJ : (gb.p[i].C != 512 || gb.p[i].cnum > back_form)) return NR_EUNSUPPORTED;
    }
    const int n_back = gb.start[NR_CTM_MAX_GROUP], n_back_pad = (n_back + 7) & ~7;
    size_t lds = Tile::RING_BYTES;
    if (use_lds && back_lds > lds) lds = back_lds;
    const void* k = back_form == 0 ? (const void*)nr_back_beside_linear_kernel<MI, NI, STAGES, WC, 0>
                    : back_form == 4 ? (const void*)nr_back_beside_linear_kernel<MI, NI, STAGES, WC, 4>
                                     : (const void*)nr_back_beside_linear_kernel<MI, NI, STAGES, WC, 16>;
    if (lds > 40 * 1024) {
        hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    const dim3 grid(n_back_pad + nr_xcd_chunk_grid(total)), block(128 * WC);
    if (back_form == 0) hipLaunchKernelGGL((nr_back_beside_linear_kernel<MI, NI, STAGES, WC, 0>), grid, block, lds, st, gb, n_back_pad, use_lds, g);
    else if (back_form == 4) hipLaunchKernelGGL((nr_back_beside_linear_kernel<MI, NI, STAGES, WC, 4>), grid, block, lds, st, gb, n_back_pad, use_lds, g);
    else hipLaunchKernelGGL((nr_back_beside_linear_kernel<MI, NI, STAGES, WC, 16>), grid, block, lds, st, gb, n_back_pad, use_lds, g);
    NR_LAUNCH_CHECK();
    return NR_OK;
}

template <int MI, int NI, int STAGES>
static int nr_linear_launch_s(NrLinearArgs& a, hipStream_t st) {
    using Tile = NrGemmTile<MI, NI, true, 16, 16, STAGES>;
    size_t lds = Tile::RING_BYTES;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)nr_linear_kernel<MI, NI, STAGES>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    dim3 grid((a.N + Tile::BN - 1) / Tile::BN, (a.M + Tile::BM - 1) / Tile::BM);
    hipLaunchKernelGGL((nr_linear_kernel<MI, NI, STAGES>), grid, dim3(256), lds, st, a);
    NR_LAUNCH_CHECK();
    return NR_OK;
}

template <int MI, int NI>
static int nr_linear_launch(NrLinearArgs& a, hipStream_t st) {
    long n_wg = (long)((a.N + 32 * NI - 1) / (32 * NI)) * ((a.M + 32 * MI - 1) / (32 * MI));
    if (nr_pick_stages(n_wg) == 1) return nr_linear_launch_s<MI, NI, 1>(a, st);
    return nr_linear_launch_s<MI, NI, 2>(a, st);
}

template <int MI, int NI, int STAGES, int WC = 2, bool CONV = false, bool X3 = true>
static int nr_linear_group_launch_s(const NrLinearArgs* probs, int n, hipStream_t st) {
    using Tile = NrGemmTile<MI, NI, X3, 16, 16, STAGES, WC>;
    NrLinearGroup g;
    g.n = n;
    int total = 0;
    for (int i = 0; i < n; ++i) {
        g.p[i] = probs[i];
        g.tile_start[i] = total;
        total += ((probs[i].M + Tile::BM - 1) / Tile::BM) * ((probs[i].N + Tile::BN - 1) / Tile::BN);
    }
    for (int i = n; i <= NR_LINEAR_MAX_GROUP; ++i) g.tile_start[i] = total;
    size_t lds = Tile::RING_BYTES;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)nr_linear_group_kernel<MI, NI, STAGES, WC, CONV, X3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL((nr_linear_group_kernel<MI, NI, STAGES, WC, CONV, X3>), dim3(nr_xcd_chunk_grid(total)), dim3(128 * WC), lds, st, g);
    NR_LAUNCH_CHECK();
    return NR_OK;
}

// Tile shape / ring depth of a grouped launch.  NR_LINEAR_TILE="MI,NI,STAGES" overrides (tuning only).
int nr_linear_group_launch(const NrLinearArgs* probs, int n, hipStream_t st, bool conv) {
    if (!probs || n <= 0 || n > NR_LINEAR_MAX_GROUP) return NR_EINVAL;
    for (int i = 0; i < n; ++i) {
        if (conv != (probs[i].conv_n > 0)) return NR_EINVAL;
        if (conv && (probs[i].K % 192) != 0) return NR_EINVAL;             // three segments of whole 64-wide slices
        if ((probs[i].out_hi == nullptr) != (probs[i].out_lo == nullptr)) return NR_EINVAL;
    }
    long t32x64 = 0, t128 = 0;
    const bool one_pass = !probs[0].x_lo && !probs[0].w_lo;
    for (int i = 0; i < n; ++i) {
        const NrLinearArgs& a = probs[i];
        if (!a.x_hi || !a.w_hi || !a.out || a.M <= 0 || a.N <= 0 || a.K <= 0 || (a.K % 64) != 0) return NR_EINVAL;
        if ((a.x_lo == nullptr) != one_pass || (a.w_lo == nullptr) != one_pass) return NR_EINVAL;       // one kernel variant per launch
        if (a.ld < 0 || (a.ld > 0 && (a.ld < a.K || (a.ld % 8) != 0 || conv))) return NR_EINVAL;
        t32x64 += (long)((a.M + 31) / 32) * ((a.N + 63) / 64);
        t128 += (long)((a.M + 127) / 128) * ((a.N + 127) / 128);
    }
    // Measured on MI355X over the clustering-stage shapes (tools/cluster_times.py under rocprofv3, us; XCD-aware
    // tile order):
    //   <= 256 tiles of 32x64 (proj, stage-1 conv: one workgroup per CU, latency-bound K loop): 4-deep ring
    //       <1,2,4> 7.1 / 10.3   vs <1,2,2> 8.9 / 14.0   vs <2,2,2> 12.1 / 18.1
    //   <= 640 (stage-1 q+kv):       <1,2,2> 9.2    vs <2,2,2> 10.7   vs <1,2,4> 11.9
    //   <= 1536 (stage-0 conv):      <2,2,1> 34.5   vs 64x128 on 8 waves 38.9   vs <1,2,2> 41.7   vs <2,4,1> 47.7
    //   larger (stage-0 q+kv):       64x128 on 8 waves, 1 stage 22.6   vs <2,4,1> 25.7   vs <2,2,1> 26.5   vs <1,2,2> 30.7
    int mi, ni, stg, wcols = 2;
    if (t32x64 <= 256) { mi = 1; ni = 2; stg = 4; }
    else if (t32x64 <= 640) { mi = 1; ni = 2; stg = 2; }
    else if (t32x64 <= 1536) { mi = 2; ni = 2; stg = 1; }
    else if (t128 < 1024) { mi = 2; ni = 2; stg = 1; wcols = 4; }
    else { mi = 4; ni = 4; stg = 1; }
    if (const char* ov = nr_tune_env("NR_LINEAR_TILE")) {
        int a_, b_, c_, d_ = 2;
        if (sscanf(ov, "%d,%d,%d,%d", &a_, &b_, &c_, &d_) >= 3) { mi = a_; ni = b_; stg = c_; wcols = d_; }
    }
    if (one_pass) {
        // hi halves only (a third of the MFMAs, half the operand bytes): 64 x 128 blocks on 8 waves, or 64 x 64 on 4 for small
        // grids; two-deep ring unless the grid brings >= 2 workgroups per CU.  (tools/scorer_bwd_times.py, NR_LINEAR_TILE)
        if (conv) return NR_EUNSUPPORTED;
        const long t64x128 = t32x64 / 4;
        if (!nr_tune_env("NR_LINEAR_TILE")) {
            mi = 2; ni = 2;
            wcols = t64x128 >= 256 ? 4 : 2;
            stg = (wcols == 4 ? t64x128 : t32x64 / 2) >= 512 ? 1 : 2;
        }
        if (const char* ov = nr_tune_env("NR_LINEAR_TILE1")) {       // the one-pass launches only
            int a_, b_, c_, d_;
            if (sscanf(ov, "%d,%d,%d,%d", &a_, &b_, &c_, &d_) == 4) { mi = a_; ni = b_; stg = c_; wcols = d_; }
        }
#define NR_LG1_CASE(MI_, NI_, ST_, WC_) if (mi == MI_ && ni == NI_ && stg == ST_ && wcols == WC_) return nr_linear_group_launch_s<MI_, NI_, ST_, WC_, false, false>(probs, n, st)
        NR_LG1_CASE(2, 2, 1, 4); NR_LG1_CASE(2, 2, 2, 4); NR_LG1_CASE(2, 2, 1, 2); NR_LG1_CASE(2, 2, 2, 2);
#ifdef NR_TUNE      // shapes only the NR_LINEAR_TILE1 hook of a tuning build asks for
        NR_LG1_CASE(4, 2, 1, 4); NR_LG1_CASE(4, 2, 2, 4); NR_LG1_CASE(2, 4, 2, 2);
#endif
#undef NR_LG1_CASE
        return NR_EUNSUPPORTED;
    }
    if (conv) {                      // token convolutions: the two shapes the clustering stages use (and their neighbours)
        if (wcols == 4 || mi > 2) { mi = 2; ni = 2; stg = 1; }
        if (const char* ov = nr_tune_env("NR_LINEAR_TILE_CONV")) {          // the convolutions only: "MI,NI,STAGES"
            int a_, b_, c_;
            if (sscanf(ov, "%d,%d,%d", &a_, &b_, &c_) == 3 && t32x64 > 256) { mi = a_; ni = b_; stg = c_; }
        }
#define NR_LGC_CASE(MI_, NI_, ST_) if (mi == MI_ && ni == NI_ && stg == ST_) return nr_linear_group_launch_s<MI_, NI_, ST_, 2, true>(probs, n, st)
        NR_LGC_CASE(2, 2, 1); NR_LGC_CASE(1, 2, 2); NR_LGC_CASE(1, 2, 4);
#ifdef NR_TUNE
        NR_LGC_CASE(2, 2, 2);
        NR_LGC_CASE(3, 4, 1); NR_LGC_CASE(3, 4, 2); NR_LGC_CASE(3, 2, 1); NR_LGC_CASE(3, 2, 2); NR_LGC_CASE(4, 4, 1); NR_LGC_CASE(4, 4, 2);
        NR_LGC_CASE(4, 2, 1); NR_LGC_CASE(4, 2, 2); NR_LGC_CASE(2, 4, 1); NR_LGC_CASE(2, 4, 2); NR_LGC_CASE(6, 2, 1); NR_LGC_CASE(6, 2, 2);
#endif
#undef NR_LGC_CASE
        return NR_EUNSUPPORTED;
    }
    if (wcols == 4) {
#define NR_LG8_CASE(MI_, NI_, ST_) if (mi == MI_ && ni == NI_ && stg == ST_) return nr_linear_group_launch_s<MI_, NI_, ST_, 4>(probs, n, st)
        NR_LG8_CASE(2, 2, 1);                             // 64 x 128 on 8 waves (the shape the table above picks)
#ifdef NR_TUNE      // the other 8-wave shapes: NR_LINEAR_TILE of a tuning build only
        NR_LG8_CASE(4, 2, 1); NR_LG8_CASE(4, 2, 2);       // 128 x 128
        NR_LG8_CASE(2, 2, 2);
        NR_LG8_CASE(4, 4, 1);                             // 128 x 256
        NR_LG8_CASE(2, 4, 1); NR_LG8_CASE(2, 4, 2);       // 64 x 256
#endif
#undef NR_LG8_CASE
        return NR_EUNSUPPORTED;
    }
#define NR_LG_CASE(MI_, NI_, ST_) if (mi == MI_ && ni == NI_ && stg == ST_) return nr_linear_group_launch_s<MI_, NI_, ST_>(probs, n, st)
    NR_LG_CASE(4, 4, 1);
    NR_LG_CASE(2, 2, 1);
    NR_LG_CASE(1, 2, 2); NR_LG_CASE(1, 2, 4);
#ifdef NR_TUNE
    NR_LG_CASE(2, 4, 1); NR_LG_CASE(2, 4, 2); NR_LG_CASE(2, 2, 2);
#endif
#undef NR_LG_CASE
    return NR_EUNSUPPORTED;
}

extern "C" int nr_linear_x3(const uint16_t* x_hi, const uint16_t* x_lo, const uint16_t* w_hi, const uint16_t* w_lo,
                            const float* bias, const float* residual, int M, int N, int K, float* out, void* stream) {
    if (!x_hi || !x_lo || !w_hi || !w_lo || !out || M <= 0 || N <= 0 || K <= 0 || (K % 64) != 0) return NR_EINVAL;
    NrLinearArgs a{x_hi, x_lo, w_hi, w_lo, bias, residual, out, M, N, K};
    hipStream_t st = (hipStream_t)stream;
    // enough workgroups to cover the 256 CUs: 128x128 tiles when that already gives >= 192 of them
    long t128 = (long)((M + 127) / 128) * ((N + 127) / 128);
    if (t128 >= 192) return nr_linear_launch<4, 4>(a, st);
    return nr_linear_launch<2, 4>(a, st);
}

// x[n-1] | x[n] | x[n+1] written directly as bf16 hi / lo (operand of the conv GEMM); body in nr_ctm_bodies.h
__global__ __launch_bounds__(256) void nr_shift_concat_split_kernel(NrShiftArgs a) { nr_shift_split_body(a, blockIdx.x); }

extern "C" int nr_shift_concat_split(const float* x, int n_samples, int N, int C, uint16_t* hi, uint16_t* lo, void* stream) {
    if (!x || !hi || !lo || n_samples <= 0 || N <= 0 || C <= 0 || (C % 4) != 0) return NR_EINVAL;
    NrShiftArgs a{x, N, C, hi, lo, 0};
    hipLaunchKernelGGL(nr_shift_concat_split_kernel, dim3(n_samples * N), dim3(256), 0, (hipStream_t)stream, a);
    NR_LAUNCH_CHECK();
    return NR_OK;
}
