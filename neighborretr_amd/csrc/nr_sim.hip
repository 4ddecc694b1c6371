// Fused local_level forward (reference: NeighborRetr/models/modeling.py:499-512).
//
// rows  = text tokens  (A samples x Nt tokens),  cols = video tokens (Bv samples x Nv tokens).
// One workgroup computes a BM x BN block of token-token cosine products that contains only WHOLE
// (text, video) pairs -- TA = BM/Nt texts x TB = BN/Nv videos -- so both max-pools (over the
// video tokens of a pair, over the text tokens of a pair) stay inside the block.  The block is
// produced on MFMA (nr_gemm_tile.h), dropped into LDS once, and reduced there:
//   phase A: P[row, video]  = max_v C[row, video*Nv+v]   (x w_t[row])   + arg-max
//   phase B: Q[text, col]   = max_t C[text*Nt+t, col]    (x w_v[col])   + arg-max
//   phase C: S[text, video] = 0.5 * (sum_t P + sum_v Q)   (J lanes per pair, shuffle-reduced)
//   phase D: optional row / column sums of S over the block (memory-bank centrality).
// Masked tokens were zeroed by nr_prepare_tokens, so their products are exactly 0 and take part
// in the max like the reference's mask multiply (modeling.py:500-501).
#include <stdlib.h>
#include "nr_gemm_tile.h"
#include "../../include/nr_hip.h"

struct NrSimArgs {
    const uint16_t *t_hi, *t_lo, *v_hi, *v_lo;
    const float *w_t, *w_v;
    float* out;
    uint8_t *arg_v, *arg_t;
    float *pmax, *qmax;
    int A, Nt, Bv, Nv, K;
    int TA, TB;
    int out_mode;
    int ldc;      // floats per LDS C row
    int J;        // lanes per pair in phase C (power of two, <= 64)
    int off_pw, off_qw, off_sp, off_wt;   // byte offsets of the LDS scratch arrays
};

template <int MI, int NI, bool X3>
__global__ __launch_bounds__(256) void nr_sim_kernel(NrSimArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using Tile = NrGemmTile<MI, NI, X3>;
    constexpr int BM = Tile::BM, BN = Tile::BN;
    const int tid = threadIdx.x;
    const int bx = blockIdx.x, by = blockIdx.y;
    const int Nt = p.Nt, Nv = p.Nv, TA = p.TA, TB = p.TB;
    const int row0 = by * TA * Nt;
    const int col0 = bx * TB * Nv;
    const int nrow = TA * Nt;     // tile rows that belong to whole texts
    const int ncol = TB * Nv;

    float* sC = reinterpret_cast<float*>(smem);
    float* sPW = reinterpret_cast<float*>(smem + p.off_pw);
    float* sQW = reinterpret_cast<float*>(smem + p.off_qw);
    float* sSP = reinterpret_cast<float*>(smem + p.off_sp);
    float* sWT = reinterpret_cast<float*>(smem + p.off_wt);     // token weights of the tile's rows / columns,
    float* sWV = sWT + BM;                                      // 0 for rows / columns outside the problem

    // token weights into LDS up front: their global latency hides under the main loop
    for (int r = tid; r < BM; r += 256) {
        int ag = by * TA + r / Nt;
        sWT[r] = (r < nrow && ag < p.A) ? p.w_t[row0 + r] : 0.f;
    }
    for (int c = tid; c < BN; c += 256) {
        int bg = bx * TB + c / Nv;
        sWV[c] = (c < ncol && bg < p.Bv) ? p.w_v[col0 + c] : 0.f;
    }

    Tile tile;
    tile.zero();
    tile.run(p.t_hi, p.t_lo, row0, p.A * Nt, p.v_hi, p.v_lo, col0, p.Bv * Nv, p.K, smem);

    const int ldc = p.ldc;
    tile.store_lds(sC, ldc);
    __syncthreads();

    // ---- phase A: max over the video tokens of each (row, video) ----------------------------
    const bool vec_a = (Nv & 3) == 0;
    for (int i = tid; i < nrow * TB; i += 256) {
        int r = i / TB, bl = i - r * TB;
        const float* c = sC + r * ldc + bl * Nv;
        float m;
        int am = 0;
        if (vec_a) {
            f32x4_t x = *reinterpret_cast<const f32x4_t*>(c);
            m = x[0];
            if (x[1] > m) { m = x[1]; am = 1; }
            if (x[2] > m) { m = x[2]; am = 2; }
            if (x[3] > m) { m = x[3]; am = 3; }
            for (int v = 4; v < Nv; v += 4) {
                x = *reinterpret_cast<const f32x4_t*>(c + v);
                if (x[0] > m) { m = x[0]; am = v; }
                if (x[1] > m) { m = x[1]; am = v + 1; }
                if (x[2] > m) { m = x[2]; am = v + 2; }
                if (x[3] > m) { m = x[3]; am = v + 3; }
            }
        } else {
            m = c[0];
            for (int v = 1; v < Nv; ++v) {
                float x = c[v];
                if (x > m) { m = x; am = v; }
            }
        }
        sPW[i] = m * sWT[r];
        if (p.arg_v) {
            int al = r / Nt, t = r - al * Nt;
            int ag = by * TA + al, bg = bx * TB + bl;
            if (ag < p.A && bg < p.Bv) {
                size_t o = ((size_t)ag * p.Bv + bg) * Nt + t;
                p.arg_v[o] = (uint8_t)am;
                p.pmax[o] = m;
            }
        }
    }
    // ---- phase B: max over the text tokens of each (text, col); 4 adjacent columns per thread ----
    const int ngrp = (ncol + 3) >> 2;
    for (int i = tid; i < TA * ngrp; i += 256) {
        int al = i / ngrp, cg = i - al * ngrp;
        const float* col = sC + (al * Nt) * ldc + cg * 4;
        f32x4_t m = *reinterpret_cast<const f32x4_t*>(col);
        int am0 = 0, am1 = 0, am2 = 0, am3 = 0;
        for (int t = 1; t < Nt; ++t) {
            f32x4_t x = *reinterpret_cast<const f32x4_t*>(col + t * ldc);
            if (x[0] > m[0]) { m[0] = x[0]; am0 = t; }
            if (x[1] > m[1]) { m[1] = x[1]; am1 = t; }
            if (x[2] > m[2]) { m[2] = x[2]; am2 = t; }
            if (x[3] > m[3]) { m[3] = x[3]; am3 = t; }
        }
        const int am[4] = {am0, am1, am2, am3};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            int c = cg * 4 + e;
            if (c >= ncol) break;
            sQW[al * ncol + c] = m[e] * sWV[c];
            if (p.arg_t) {
                int bl = c / Nv, v = c - bl * Nv;
                int ag = by * TA + al, bg = bx * TB + bl;
                if (ag < p.A && bg < p.Bv) {
                    size_t o = ((size_t)ag * p.Bv + bg) * Nv + v;
                    p.arg_t[o] = (uint8_t)am[e];
                    p.qmax[o] = m[e];
                }
            }
        }
    }
    __syncthreads();

    // ---- phase C: weighted sums per pair ----------------------------------------------------
    const int J = p.J;
    const int npair = TA * TB;
    const int groups = 256 / J;
    const int g = tid / J, j = tid - g * J;
    for (int base = 0; base < npair; base += groups) {
        int pr = base + g;
        float s = 0.f;
        int al = 0, bl = 0;
        if (pr < npair) {
            al = pr / TB;
            bl = pr - al * TB;
            for (int t = j; t < Nt; t += J) s += sPW[(al * Nt + t) * TB + bl];
            for (int v = j; v < Nv; v += J) s += sQW[al * ncol + bl * Nv + v];
        }
        for (int o = J >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (j == 0 && pr < npair) {
            int ag = by * TA + al, bg = bx * TB + bl;
            bool ok = (ag < p.A) && (bg < p.Bv);
            float val = 0.5f * s;
            if (p.out_mode == NR_OUT_FULL) {
                if (ok) p.out[(size_t)ag * p.Bv + bg] = val;
            } else {
                sSP[pr] = ok ? val : 0.f;
            }
        }
    }
    if (p.out_mode == NR_OUT_FULL) return;
    __syncthreads();
    // ---- phase D: block-level sums (fixed order => deterministic) ---------------------------
    if (p.out_mode == NR_OUT_ROWSUM) {
        if (tid < TA) {
            int ag = by * TA + tid;
            float s = 0.f;
            for (int bl = 0; bl < TB; ++bl) s += sSP[tid * TB + bl];
            if (ag < p.A) p.out[(size_t)bx * p.A + ag] = s;
        }
    } else {
        if (tid < TB) {
            int bg = bx * TB + tid;
            float s = 0.f;
            for (int al = 0; al < TA; ++al) s += sSP[al * TB + tid];
            if (bg < p.Bv) p.out[(size_t)by * p.Bv + bg] = s;
        }
    }
}

int nr_sim_reg_dispatch(const uint16_t* t_hi, const uint16_t* t_lo, const uint16_t* v_hi, const uint16_t* v_lo,
                        const float* w_t, const float* w_v, int A, int Nt, int Bv, int Nv, int d, int prec, int out_mode,
                        float* out, uint8_t* arg_v, uint8_t* arg_t, float* pmax, float* qmax, hipStream_t st);
extern "C" int nr_sim_reg_tile(int A, int Nt, int Bv, int Nv, int prec, int* TA, int* TB);

static bool nr_sim_force_generic() {
    const char* e = nr_tune_env("NR_SIM_GENERIC");
    return e && atoi(e);
}

// ---- host side -------------------------------------------------------------------------------
// pick the tile extent (64 / 96 / 128 rows) that wastes the fewest MFMA rows on padding
static int nr_pick_extent(int n_tok, int* mi_out) {
    int best_mi = 0;
    double best_eff = -1.0;
    for (int mi = 2; mi <= 4; ++mi) {
        int ext = 32 * mi;
        int t = ext / n_tok;
        if (t <= 0) continue;
        double eff = (double)(t * n_tok) / ext;
        if (eff > best_eff + 1e-9 || (eff > best_eff - 1e-9 && mi > best_mi)) {
            best_eff = eff;
            best_mi = mi;
        }
    }
    if (best_mi == 0) return 0;
    *mi_out = best_mi;
    return (32 * best_mi) / n_tok;
}

extern "C" int nr_local_level_tiles(int A, int Nt, int Bv, int Nv, int prec, int* n_row_tiles, int* n_col_tiles) {
    int mi, ni;
    if (A <= 0 || Bv <= 0 || Nt <= 0 || Nv <= 0) return NR_EINVAL;
    int TA = nr_pick_extent(Nt, &mi), TB = nr_pick_extent(Nv, &ni);
    if (TA == 0 || TB == 0) return NR_EUNSUPPORTED;   // more than 128 tokens per sample
    if (TA > A) TA = A > 0 ? A : 1;
    if (TB > Bv) TB = Bv > 0 ? Bv : 1;
    if (!nr_sim_force_generic()) nr_sim_reg_tile(A, Nt, Bv, Nv, prec, &TA, &TB);    // the register-epilogue kernel's blocks
    if (n_row_tiles) *n_row_tiles = (A + TA - 1) / TA;
    if (n_col_tiles) *n_col_tiles = (Bv + TB - 1) / TB;
    return NR_OK;
}

template <int MI, int NI, bool X3>
static int nr_sim_launch(NrSimArgs& a, dim3 grid, size_t lds, hipStream_t st) {
    auto kern = nr_sim_kernel<MI, NI, X3>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, a);
    NR_LAUNCH_CHECK();
    return NR_OK;
}

extern "C" int nr_local_level_fwd(const uint16_t* t_hi, const uint16_t* t_lo, const uint16_t* v_hi,
                                  const uint16_t* v_lo, const float* w_t, const float* w_v, int A, int Nt, int Bv,
                                  int Nv, int d, int prec, int out_mode, float* out, uint8_t* arg_v,
                                  uint8_t* arg_t, float* pmax, float* qmax, void* stream) {
    if (!t_hi || !v_hi || !w_t || !w_v || !out) return NR_EINVAL;
    if (A <= 0 || Bv <= 0 || Nt <= 0 || Nv <= 0 || d <= 0 || (d % 64) != 0) return NR_EINVAL;
    if (prec != NR_PREC_BF16 && prec != NR_PREC_BF16X3) return NR_EINVAL;
    if (prec == NR_PREC_BF16X3 && (!t_lo || !v_lo)) return NR_EINVAL;
    if (out_mode < 0 || out_mode > 2) return NR_EINVAL;
    if ((arg_v || arg_t || pmax || qmax) && !(arg_v && arg_t && pmax && qmax)) return NR_EINVAL;
    if (Nt > 255 || Nv > 255) return NR_EUNSUPPORTED;   // arg-max indices are u8
    // token counts with a register-level epilogue (nr_sim_reg.hip); NR_SIM_GENERIC=1 forces this file's
    // LDS epilogue (test hook)
    {
        if (!nr_sim_force_generic()) {
            int rc = nr_sim_reg_dispatch(t_hi, t_lo, v_hi, v_lo, w_t, w_v, A, Nt, Bv, Nv, d, prec, out_mode, out, arg_v, arg_t,
                                         pmax, qmax, (hipStream_t)stream);
            if (rc != NR_EUNSUPPORTED) return rc;
        }
    }
    int mi = 0, ni = 0;
    int TA = nr_pick_extent(Nt, &mi), TB = nr_pick_extent(Nv, &ni);
    if (TA == 0 || TB == 0) return NR_EUNSUPPORTED;
    if (TA > A) TA = A;
    if (TB > Bv) TB = Bv;

    NrSimArgs a;
    a.t_hi = t_hi; a.t_lo = t_lo; a.v_hi = v_hi; a.v_lo = v_lo;
    a.w_t = w_t; a.w_v = w_v; a.out = out; a.arg_v = arg_v; a.arg_t = arg_t; a.pmax = pmax; a.qmax = qmax;
    a.A = A; a.Nt = Nt; a.Bv = Bv; a.Nv = Nv; a.K = d; a.TA = TA; a.TB = TB; a.out_mode = out_mode;
    const int BM = 32 * mi, BN = 32 * ni;
    a.ldc = BN + 4;
    int npair = TA * TB;
    int J = 64;
    while (J > 1 && J * npair > 256) J >>= 1;
    a.J = J;
    size_t stage = (size_t)2 * (BM + BN) * 128 * (prec == NR_PREC_BF16X3 ? 2 : 1);   // DMA ring
    size_t cbytes = (size_t)BM * a.ldc * 4;
    size_t base = stage > cbytes ? stage : cbytes;
    base = (base + 15) & ~(size_t)15;
    a.off_pw = (int)base;
    size_t pw = ((size_t)TA * Nt * TB * 4 + 15) & ~(size_t)15;
    a.off_qw = (int)(base + pw);
    size_t qw = ((size_t)TA * TB * Nv * 4 + 15) & ~(size_t)15;
    a.off_sp = (int)(base + pw + qw);
    size_t sp = ((size_t)npair * 4 + 15) & ~(size_t)15;
    a.off_wt = (int)(base + pw + qw + sp);
    size_t lds = base + pw + qw + sp + (size_t)(BM + BN) * 4;
    if (lds > 160 * 1024) return NR_EUNSUPPORTED;

    dim3 grid((Bv + TB - 1) / TB, (A + TA - 1) / TA);
    hipStream_t st = (hipStream_t)stream;
    const bool x3 = prec == NR_PREC_BF16X3;
#define NR_SIM_CASE(M_, N_)                                                       \
    if (mi == M_ && ni == N_)                                                     \
        return x3 ? nr_sim_launch<M_, N_, true>(a, grid, lds, st) : nr_sim_launch<M_, N_, false>(a, grid, lds, st);
    NR_SIM_CASE(2, 2) NR_SIM_CASE(2, 3) NR_SIM_CASE(2, 4)
    NR_SIM_CASE(3, 2) NR_SIM_CASE(3, 3) NR_SIM_CASE(3, 4)
    NR_SIM_CASE(4, 2) NR_SIM_CASE(4, 3) NR_SIM_CASE(4, 4)
#undef NR_SIM_CASE
    return NR_EUNSUPPORTED;
}
