// Backward of the fused local_level product with respect to one operand's (normalised) tokens, as an MFMA GEMM.
// Reference semantics: autograd of modeling.py:499-512 -- a max-pool routes its gradient to the arg-max, so with
// g(s,o) = 0.5 * dS(s,o) the gradient of token n of "self" sample s is
//     d_x[(s,n), :] = sum over other samples o, tokens m of   P[(s,n),(o,m)] * x_other[(o,m), :]
//     P[(s,n),(o,m)] = g(s,o) * ( w_other[o,m] * [scat(s,o,m) == n]  +  w_self[s,n] * [gath(s,o,n) == m] )
// (scat / gath = the arg-max indices stored by the forward kernel).  P is a [n_self_tokens, n_other_tokens]
// matrix with <= 2 non-zeros per (s, o, n) -- nr_sim_bwd_kernel walks it entry by entry (466 us per bank product
// on MI355X); here 96 x 96 blocks of P (whole samples on both sides: 96 = 4 x 24 = 8 x 12 tokens) are GENERATED
// in LDS as bf16 from the index bytes and multiplied on the matrix cores with the other operand's tokens.
//
// Workgroup = 8 waves; block = 96 self-token rows x 256 feature dims x one chunk of the other operand's samples.
// Wave w owns dims [32 w, 32 w + 32) of the block: its B fragments (x_other^T, [dim][token] bf16, k-contiguous)
// are private, so they go straight from global memory to registers (prefetched one slice ahead); only P is
// shared through LDS.  The chunks' partial sums are reduced in fixed order by nr_sum_chunks_kernel.
#include "nr_common.h"
#include "../../include/nr_hip.h"

struct NrBwdMfmaArgs {
    const float* dS;
    int ds_mode;
    float ds_scale;
    const uint16_t* oT;                 // other operand's prepared tokens, TRANSPOSED: [d][ldk] bf16
    const uint16_t* oT_lo;              // optional low halves (same layout) or nullptr
    int ldk;
    const float *w_self, *w_other;
    const uint8_t *gath, *scat;         // [A][Bv][Ns] and [A][Bv][No]
    int side, A, Bv, Ns, No, d;
    int n_self, n_other;                // samples
    int slices_per_chunk, n_slices;
    float* part;                        // [n_chunks][n_self*Ns][d]
    size_t part_stride;
};

#define BW_ROWS 96
#define BW_K 96
#define BW_LDA 104                      // bf16 elements per P row in LDS: 208 B = 52 dwords -> conflict-free b128 reads
#define BW_THREADS 512

__global__ __launch_bounds__(BW_THREADS) void nr_sim_bwd_mfma_kernel(NrBwdMfmaArgs p) {
    __shared__ __attribute__((aligned(16))) uint16_t sP[BW_ROWS * BW_LDA];
    __shared__ uint8_t s_gath[32 * 24], s_scat[32 * 24];      // [pair][token] (pairs = TS * TO <= 32, tokens <= 24)
    __shared__ float s_g[32], s_wo[BW_K], s_ws[BW_ROWS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int Ns = p.Ns, No = p.No;
    const int TS = BW_ROWS / Ns, TO = BW_K / No;              // samples per block row range / per k slice
    const int s0 = blockIdx.x * TS;                           // first self sample of the block
    const int dim0 = blockIdx.y * 256 + wave * 32;
    const int chunk = blockIdx.z;
    const int sl_begin = chunk * p.slices_per_chunk;
    const int sl_end = min(sl_begin + p.slices_per_chunk, p.n_slices);

    for (int e = tid; e < BW_ROWS; e += BW_THREADS) {
        const int s = s0 + e / Ns;
        s_ws[e] = s < p.n_self ? p.w_self[(size_t)s * Ns + e % Ns] : 0.f;
    }
    f32x4_t acc[6][2];
#pragma unroll
    for (int m = 0; m < 6; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[m][n] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // B fragments of one slice: [ks 0..2][ni 0..1], lane -> dim = dim0 + 16 ni + (lane & 15), k = k0 + 32 ks + 8 (lane >> 4)
    auto load_b = [&](int slice, bf16x8_t (&fr)[3][2], const uint16_t* base) {
        const int k0 = slice * BW_K;
#pragma unroll
        for (int ks = 0; ks < 3; ++ks)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const int dim = min(dim0 + 16 * ni + (lane & 15), p.d - 1);
                const int k = min(k0 + 32 * ks + 8 * (lane >> 4), p.ldk - 8);
                fr[ks][ni] = *reinterpret_cast<const bf16x8_t*>(base + (size_t)dim * p.ldk + k);
            }
    };
    bf16x8_t bcur[3][2], bnext[3][2], lcur[3][2], lnext[3][2];
    if (sl_begin < sl_end) {
        load_b(sl_begin, bcur, p.oT);
        if (p.oT_lo) load_b(sl_begin, lcur, p.oT_lo);
    }

    for (int sl = sl_begin; sl < sl_end; ++sl) {
        const int o0 = sl * TO;                               // first other sample of the slice
        if (sl + 1 < sl_end) {
            load_b(sl + 1, bnext, p.oT);
            if (p.oT_lo) load_b(sl + 1, lnext, p.oT_lo);
        }
        // ---- stage the slice's index bytes, pair gradients and other-token weights ---------------------------
        __syncthreads();                                      // previous slice's P fully consumed
        const int n_pairs = TS * TO;
        for (int e = tid; e < n_pairs * Ns; e += BW_THREADS) {
            const int pr = e / Ns, n = e - pr * Ns;
            const int s = s0 + pr / TO, o = o0 + pr % TO;
            const bool ok = s < p.n_self && o < p.n_other;
            const size_t pair = p.side == 0 ? (size_t)s * p.Bv + o : (size_t)o * p.Bv + s;
            s_gath[pr * 24 + n] = ok ? p.gath[pair * Ns + n] : 255;
        }
        for (int e = tid; e < n_pairs * No; e += BW_THREADS) {
            const int pr = e / No, m = e - pr * No;
            const int s = s0 + pr / TO, o = o0 + pr % TO;
            const bool ok = s < p.n_self && o < p.n_other;
            const size_t pair = p.side == 0 ? (size_t)s * p.Bv + o : (size_t)o * p.Bv + s;
            s_scat[pr * 24 + m] = ok ? p.scat[pair * No + m] : 255;
        }
        if (tid < n_pairs) {
            const int s = s0 + tid / TO, o = o0 + tid % TO;
            float g = 0.f;
            if (s < p.n_self && o < p.n_other) {
                const int a = p.side == 0 ? s : o, b = p.side == 0 ? o : s;
                if (p.ds_mode == 0) g = p.dS[(size_t)a * p.Bv + b];
                else if (p.ds_mode == 1) g = p.dS[a];
                else g = p.dS[b];
                g *= 0.5f * p.ds_scale;
            }
            s_g[tid] = g;
        }
        for (int e = tid; e < BW_K; e += BW_THREADS) {
            const int o = o0 + e / No;
            s_wo[e] = o < p.n_other ? p.w_other[(size_t)o * No + e % No] : 0.f;
        }
        __syncthreads();
        // ---- generate P: one (pair, self token) row segment of No entries per work item -----------------------
        for (int e = tid; e < n_pairs * Ns; e += BW_THREADS) {
            const int pr = e / Ns, n = e - pr * Ns;
            const int si = pr / TO, oi = pr - si * TO;
            const float g = s_g[pr];
            const float gws = g * s_ws[si * Ns + n];
            const int gm = s_gath[pr * 24 + n];
            uint16_t* row = sP + (si * Ns + n) * BW_LDA + oi * No;
            for (int m = 0; m < No; m += 2) {
                float v0 = (s_scat[pr * 24 + m] == n ? g * s_wo[oi * No + m] : 0.f) + (gm == m ? gws : 0.f);
                float v1 = (s_scat[pr * 24 + m + 1] == n ? g * s_wo[oi * No + m + 1] : 0.f) + (gm == m + 1 ? gws : 0.f);
                *reinterpret_cast<uint32_t*>(row + m) = (uint32_t)nr_f2bf(v0) | ((uint32_t)nr_f2bf(v1) << 16);
            }
        }
        __syncthreads();
        // ---- P (96 x 96) times the slice's tokens ---------------------------------------------------------------
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
            bf16x8_t fa[6];
#pragma unroll
            for (int m = 0; m < 6; ++m)
                fa[m] = *reinterpret_cast<const bf16x8_t*>(sP + (16 * m + (lane & 15)) * BW_LDA + 32 * ks + 8 * (lane >> 4));
#pragma unroll
            for (int m = 0; m < 6; ++m)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    acc[m][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[m], bcur[ks][ni], acc[m][ni], 0, 0, 0);
                    if (p.oT_lo) acc[m][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[m], lcur[ks][ni], acc[m][ni], 0, 0, 0);
                }
        }
#pragma unroll
        for (int ks = 0; ks < 3; ++ks)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                bcur[ks][ni] = bnext[ks][ni];
                lcur[ks][ni] = lnext[ks][ni];
            }
    }
    // ---- partial result of this chunk ---------------------------------------------------------------------------
    float* out = p.part + (size_t)chunk * p.part_stride;
    const size_t n_rows = (size_t)p.n_self * Ns;
#pragma unroll
    for (int m = 0; m < 6; ++m)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const size_t r = (size_t)s0 * Ns + 16 * m + 4 * (lane >> 4) + j;
                const int dim = dim0 + 16 * ni + (lane & 15);
                if (r < n_rows && dim < p.d) out[r * p.d + dim] = acc[m][ni][j];
            }
}

// out[i] = (accumulate ? out[i] : 0) + sum_c part[c][i], fixed order (same job as nr_sum_chunks_kernel in nr_backward.hip)
__global__ __launch_bounds__(256) void nr_bwd_mfma_sum_kernel(const float* __restrict__ part, int n_chunks, size_t n,
                                                              float* __restrict__ out, int accumulate) {
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    f32x4_t s = accumulate ? *reinterpret_cast<const f32x4_t*>(out + i) : f32x4_t{0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < n_chunks; ++c) s += *reinterpret_cast<const f32x4_t*>(part + (size_t)c * n + i);
    *reinterpret_cast<f32x4_t*>(out + i) = s;
}

static int bwd_mfma_chunks(int row_tiles, int dim_tiles, int n_slices) {
    int c = 512 / (row_tiles * dim_tiles);          // aim at ~2 workgroups per CU
    if (c < 1) c = 1;
    if (c > n_slices) c = n_slices;
    return c > 32 ? 32 : c;
}

extern "C" int nr_local_level_bwd_mfma_supported(int Nt, int Nv, int d) {
    return (BW_ROWS % Nt) == 0 && (BW_ROWS % Nv) == 0 && Nt <= 24 && Nv <= 24 && (Nt % 2) == 0 && (Nv % 2) == 0 && (d % 256) == 0;
}

extern "C" size_t nr_local_level_bwd_mfma_workspace_bytes(int side, int A, int Nt, int Bv, int Nv, int d) {
    if (!nr_local_level_bwd_mfma_supported(Nt, Nv, d)) return 0;
    const int Ns = side == 0 ? Nt : Nv, No = side == 0 ? Nv : Nt;
    const int n_self = side == 0 ? A : Bv, n_other = side == 0 ? Bv : A;
    const int TS = BW_ROWS / Ns, TO = BW_K / No;
    const int row_tiles = (n_self + TS - 1) / TS, n_slices = (n_other + TO - 1) / TO;
    const int nch = bwd_mfma_chunks(row_tiles, d / 256, n_slices);
    return (size_t)nch * n_self * Ns * d * sizeof(float) + 256;
}

extern "C" int nr_local_level_bwd_mfma(int side, const float* dS, int ds_mode, float ds_scale, const uint16_t* oT_hi,
                                       const uint16_t* oT_lo, int ldk, const float* w_self, const float* w_other,
                                       const uint8_t* arg_v, const uint8_t* arg_t, int A, int Nt, int Bv, int Nv, int d,
                                       float* d_x, int accumulate, void* workspace, void* stream) {
    if (!dS || !oT_hi || !w_self || !w_other || !arg_v || !arg_t || !d_x || !workspace) return NR_EINVAL;
    if (side < 0 || side > 1 || ds_mode < 0 || ds_mode > 2 || A <= 0 || Bv <= 0) return NR_EINVAL;
    if (!nr_local_level_bwd_mfma_supported(Nt, Nv, d)) return NR_EUNSUPPORTED;
    NrBwdMfmaArgs p;
    p.dS = dS; p.ds_mode = ds_mode; p.ds_scale = ds_scale; p.oT = oT_hi; p.oT_lo = oT_lo; p.ldk = ldk;
    p.w_self = w_self; p.w_other = w_other; p.side = side; p.A = A; p.Bv = Bv; p.d = d;
    if (side == 0) { p.Ns = Nt; p.No = Nv; p.gath = arg_v; p.scat = arg_t; p.n_self = A; p.n_other = Bv; }
    else           { p.Ns = Nv; p.No = Nt; p.gath = arg_t; p.scat = arg_v; p.n_self = Bv; p.n_other = A; }
    const int TS = BW_ROWS / p.Ns, TO = BW_K / p.No;
    if (ldk < ((p.n_other * p.No + 7) / 8) * 8 || (ldk % 8) != 0) return NR_EINVAL;
    const int row_tiles = (p.n_self + TS - 1) / TS;
    p.n_slices = (p.n_other + TO - 1) / TO;
    const int nch = bwd_mfma_chunks(row_tiles, d / 256, p.n_slices);
    p.slices_per_chunk = (p.n_slices + nch - 1) / nch;
    const int nch_used = (p.n_slices + p.slices_per_chunk - 1) / p.slices_per_chunk;
    p.part = reinterpret_cast<float*>(workspace);
    p.part_stride = (size_t)p.n_self * p.Ns * d;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(nr_sim_bwd_mfma_kernel, dim3(row_tiles, d / 256, nch_used), dim3(BW_THREADS), 0, st, p);
    const size_t n = p.part_stride;
    hipLaunchKernelGGL(nr_bwd_mfma_sum_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, st, p.part, nch_used, n, d_x,
                       accumulate);
    NR_LAUNCH_CHECK();
    return NR_OK;
}
