// Backward kernels of the similarity path: arg-max-routed gradient of the fused local_level,
// normalisation / mask / centrality-mean backward, token-softmax backward, centrality-weight
// backward, and two small layout helpers.
//
// local_level backward (reference autograd of modeling.py:499-512): with P = max_v R, Q = max_t R,
//   dR[a,b,t,argv] += 0.5 dS[a,b] w_t[a,t],  dR[a,b,argt,v] += 0.5 dS[a,b] w_v[b,v],
// so every (text, video) pair touches only Nt + Nv token-token products.  One workgroup owns one
// sample of the differentiated operand and a slice of the feature dimension and walks over all
// samples of the other operand; each thread owns ONE feature column, keeps the other sample's
// tokens and its own accumulators in a private LDS column (dynamic token index, no bank conflicts,
// no barriers), and never needs atomics: results are bitwise reproducible.
#include "nr_common.h"
#include "../../include/nr_hip.h"

struct NrSimBwdArgs {
    const float* dS; int ds_mode; float ds_scale;
    const uint16_t *o_hi, *o_lo;
    const float *w_self, *w_other;
    const uint8_t *gath_arg, *scat_arg;   // [pair, Ns] -> other token ; [pair, No] -> self token
    const float* pool;                    // pooled maxima [pair, Ns]
    int side, A, Bv, Ns, No, d, n_loop, chunk_len;
    float *d_x, *d_w;          // partial outputs: [n_chunks][n_self*Ns*d] and [n_chunks][n_self*Ns]
    size_t x_stride, w_stride;
};

__global__ void nr_sim_bwd_kernel(NrSimBwdArgs p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int DS = blockDim.x;
    const int k = threadIdx.x;
    const int self = blockIdx.x;              // sample of the differentiated operand
    const int dim = blockIdx.y * DS + k;
    const bool live = dim < p.d;
    float* acc = lds;                          // [Ns][DS]
    float* ov = lds + (size_t)p.Ns * DS;       // [No][DS]
    const int Ns = p.Ns, No = p.No;
    for (int n = 0; n < Ns; ++n) acc[n * DS + k] = 0.f;
    float dw = 0.f;                            // thread n < Ns of slice 0 accumulates d_w[self, n]
    const bool do_w = (blockIdx.y == 0) && (k < Ns) && p.d_w;

    const int o_begin = blockIdx.z * p.chunk_len;
    const int o_end = min(o_begin + p.chunk_len, p.n_loop);
    for (int o = o_begin; o < o_end; ++o) {
        const int a = p.side == 0 ? self : o;
        const int b = p.side == 0 ? o : self;
        const size_t pair = (size_t)a * p.Bv + b;
        float g;
        if (p.ds_mode == 0) g = p.dS[pair];
        else if (p.ds_mode == 1) g = p.dS[a];
        else g = p.dS[b];
        g *= 0.5f * p.ds_scale;
        if (do_w) dw += g * p.pool[pair * Ns + k];
        if (!p.d_x) continue;
        const uint8_t* ga = p.gath_arg + pair * Ns;
        const uint8_t* sa = p.scat_arg + pair * No;
        const float* wo = p.w_other + (size_t)o * No;
        const size_t obase = (size_t)o * No * p.d + dim;
        // scatter part while streaming the other sample's tokens through the private column
        for (int m = 0; m < No; ++m) {
            float val = 0.f;
            if (live) {
                val = nr_bf2f(p.o_hi[obase + (size_t)m * p.d]);
                if (p.o_lo) val += nr_bf2f(p.o_lo[obase + (size_t)m * p.d]);
            }
            ov[m * DS + k] = val;
            int tgt = sa[m];
            acc[tgt * DS + k] += g * wo[m] * val;
        }
        // gather part
        const float* ws = p.w_self + (size_t)self * Ns;
        for (int n = 0; n < Ns; ++n) acc[n * DS + k] += g * ws[n] * ov[(int)ga[n] * DS + k];
    }
    if (p.d_x && live) {
        float* px = p.d_x + (size_t)blockIdx.z * p.x_stride;
        for (int n = 0; n < Ns; ++n) px[((size_t)self * Ns + n) * p.d + dim] = acc[n * DS + k];
    }
    if (do_w) p.d_w[(size_t)blockIdx.z * p.w_stride + (size_t)self * Ns + k] = dw;
}

// out[i] = (accumulate ? out[i] : 0) + sum_c part[c][i]   (fixed order: deterministic)
__global__ __launch_bounds__(256) void nr_sum_chunks_kernel(const float* __restrict__ part, int n_chunks, size_t n,
                                                            float* __restrict__ out, int accumulate) {
    size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    if (i + 3 < n && (n & 3) == 0) {         // rows of every chunk stay 16-byte aligned
        f32x4_t s = accumulate ? *reinterpret_cast<const f32x4_t*>(out + i) : f32x4_t{0.f, 0.f, 0.f, 0.f};
        for (int c = 0; c < n_chunks; ++c) s += *reinterpret_cast<const f32x4_t*>(part + (size_t)c * n + i);
        *reinterpret_cast<f32x4_t*>(out + i) = s;
    } else {
        for (size_t e = i + 4; i < n && i < e; ++i) {
            float s = accumulate ? out[i] : 0.f;
            for (int c = 0; c < n_chunks; ++c) s += part[(size_t)c * n + i];
            out[i] = s;
        }
    }
}

static int nr_bwd_chunks(int n_loop) {
    int c = n_loop / 8;
    if (c < 1) c = 1;
    return c > 16 ? 16 : c;
}

extern "C" size_t nr_local_level_bwd_workspace_bytes(int side, int A, int Nt, int Bv, int Nv, int d) {
    size_t n_self = side == 0 ? (size_t)A * Nt : (size_t)Bv * Nv;
    int nch = nr_bwd_chunks(side == 0 ? Bv : A);
    return (size_t)nch * (n_self * d + n_self) * sizeof(float) + 256;
}

extern "C" int nr_local_level_bwd(int side, const float* dS, int ds_mode, float ds_scale, const uint16_t* o_hi,
                                  const uint16_t* o_lo, const float* w_self, const float* w_other, const uint8_t* arg_v,
                                  const uint8_t* arg_t, const float* pmax, const float* qmax, int A, int Nt, int Bv, int Nv,
                                  int d, float* d_x, float* d_w, int accumulate, void* workspace, void* stream) {
    if (!dS || !w_self || !w_other || !arg_v || !arg_t || !pmax || !qmax || !workspace) return NR_EINVAL;
    if (side < 0 || side > 1 || ds_mode < 0 || ds_mode > 2 || A <= 0 || Bv <= 0 || Nt <= 0 || Nv <= 0 || d <= 0) return NR_EINVAL;
    if (d_x && !o_hi) return NR_EINVAL;
    if (!d_x && !d_w) return NR_EINVAL;
    if ((d % 4) != 0) return NR_EUNSUPPORTED;
    NrSimBwdArgs p;
    p.dS = dS; p.ds_mode = ds_mode; p.ds_scale = ds_scale; p.o_hi = o_hi; p.o_lo = o_lo;
    p.w_self = w_self; p.w_other = w_other; p.side = side; p.A = A; p.Bv = Bv; p.d = d;
    if (side == 0) { p.Ns = Nt; p.No = Nv; p.gath_arg = arg_v; p.scat_arg = arg_t; p.pool = pmax; p.n_loop = Bv; }
    else           { p.Ns = Nv; p.No = Nt; p.gath_arg = arg_t; p.scat_arg = arg_v; p.pool = qmax; p.n_loop = A; }
    // the walk over the other operand's samples is cut into chunks handled by different workgroups
    // (partial sums in the workspace, then one fixed-order reduction): 16x the parallelism
    const int nch = nr_bwd_chunks(p.n_loop);
    p.chunk_len = (p.n_loop + nch - 1) / nch;
    const size_t n_self = (size_t)(side == 0 ? A : Bv) * p.Ns;
    float* ws_x = reinterpret_cast<float*>(workspace);
    float* ws_w = ws_x + (size_t)nch * n_self * d;
    p.x_stride = n_self * d;
    p.w_stride = n_self;
    p.d_x = d_x ? ws_x : nullptr;
    p.d_w = d_w ? ws_w : nullptr;
    int DS = 256;
    while (DS > 64 && (size_t)(p.Ns + p.No) * DS * 4 > 150 * 1024) DS >>= 1;
    if (DS < p.Ns) return NR_EUNSUPPORTED;                 // d_w needs one thread per token
    size_t lds = (size_t)(p.Ns + p.No) * DS * 4;
    if (lds > 160 * 1024) return NR_EUNSUPPORTED;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)nr_sim_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(side == 0 ? A : Bv, d_x ? (d + DS - 1) / DS : 1, nch);
    hipLaunchKernelGGL(nr_sim_bwd_kernel, grid, dim3(DS), lds, st, p);
    if (d_x) {
        size_t n = n_self * d;
        hipLaunchKernelGGL(nr_sum_chunks_kernel, dim3((unsigned)((n / 4 + 255) / 256 + 1)), dim3(256), 0, st, ws_x, nch, n, d_x, accumulate);
    }
    if (d_w)
        hipLaunchKernelGGL(nr_sum_chunks_kernel, dim3((unsigned)((n_self / 4 + 255) / 256 + 1)), dim3(256), 0, st, ws_w, nch, n_self, d_w, accumulate);
    NR_LAUNCH_CHECK();
    return NR_OK;
}

// ---- F.normalize + mask (+ centrality mean) backward ---------------------------------------------
__global__ __launch_bounds__(256) void nr_normalize_bwd_kernel(const float* __restrict__ x, const float* __restrict__ norm,
                                                               const float* __restrict__ mask, const float* __restrict__ d_xn,
                                                               const float* __restrict__ dmean, int n_tok, int d,
                                                               float* __restrict__ dx) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_tok) return;
    const float inv = 1.0f / norm[row];
    const float mk = mask ? mask[row] : 1.0f;
    const float invn = 1.0f / (float)n_tok;
    const float* xr = x + (size_t)row * d;
    float dot = 0.f;
    for (int c = lane; c < d; c += 64) {
        float g = (d_xn ? mk * d_xn[(size_t)row * d + c] : 0.f) + (dmean ? dmean[c] * invn : 0.f);
        dot += g * xr[c] * inv;
    }
    dot = nr_wave_sum(dot);
    for (int c = lane; c < d; c += 64) {
        float g = (d_xn ? mk * d_xn[(size_t)row * d + c] : 0.f) + (dmean ? dmean[c] * invn : 0.f);
        dx[(size_t)row * d + c] = (g - xr[c] * inv * dot) * inv;
    }
}

extern "C" int nr_normalize_bwd(const float* x, const float* norm, const float* mask, const float* d_xn, const float* dmean,
                                int n_tok, int d, float* dx, void* stream) {
    if (!x || !norm || !dx || n_tok <= 0 || d <= 0 || (!d_xn && !dmean)) return NR_EINVAL;
    hipLaunchKernelGGL(nr_normalize_bwd_kernel, dim3((n_tok + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, norm, mask, d_xn,
                       dmean, n_tok, d, dx);
    NR_LAUNCH_CHECK();
    return NR_OK;
}

// ---- token softmax backward ------------------------------------------------------------------------
__global__ __launch_bounds__(256) void nr_token_softmax_bwd_kernel(const float* __restrict__ w, const float* __restrict__ dw,
                                                                   int n_samples, int N, float* __restrict__ dlogit) {
    const int lane = threadIdx.x & 63;
    const int s = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s >= n_samples) return;
    float dot = 0.f;
    for (int t = lane; t < N; t += 64) dot += w[(size_t)s * N + t] * dw[(size_t)s * N + t];
    dot = nr_wave_sum(dot);
    for (int t = lane; t < N; t += 64) {
        size_t i = (size_t)s * N + t;
        dlogit[i] = w[i] * (dw[i] - dot);
    }
}

extern "C" int nr_token_softmax_bwd(const float* w, const float* dw, int n_samples, int N, float* dlogit, void* stream) {
    if (!w || !dw || !dlogit || n_samples <= 0 || N <= 0) return NR_EINVAL;
    hipLaunchKernelGGL(nr_token_softmax_bwd_kernel, dim3((n_samples + 3) / 4), dim3(256), 0, (hipStream_t)stream, w, dw,
                       n_samples, N, dlogit);
    NR_LAUNCH_CHECK();
    return NR_OK;
}

// ---- centrality weights backward ---------------------------------------------------------------------
__global__ __launch_bounds__(256) void nr_centrality_bwd_dg_kernel(const float* __restrict__ g, const float* __restrict__ gnorm,
                                                                   const float* __restrict__ mean, const float* __restrict__ w,
                                                                   const float* __restrict__ dw, int B, int d, float scale,
                                                                   float* __restrict__ dg) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= B) return;
    const float inv = 1.0f / gnorm[i];
    const float a = dw[i] * w[i] * scale;
    const float* gi = g + (size_t)i * d;
    float dot = 0.f;
    for (int c = lane; c < d; c += 64) dot += gi[c] * inv * mean[c];
    dot = nr_wave_sum(dot);
    for (int c = lane; c < d; c += 64) dg[(size_t)i * d + c] = a * (mean[c] - gi[c] * inv * dot) * inv;
}

// 64 columns per workgroup; wave w sums the rows w, w+4, ... with four independent loads in flight, the four partial sums
// meet in LDS in a fixed order (the shape of nr_reduce_parts).  One thread per column walking all B rows one after the
// other took 30 us of dependent loads for 128 rows.
__global__ __launch_bounds__(256) void nr_centrality_bwd_dmean_kernel(const float* __restrict__ g, const float* __restrict__ gnorm,
                                                                      const float* __restrict__ w, const float* __restrict__ dw,
                                                                      int B, int d, float scale, float* __restrict__ dmean) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (c < d) {
        auto term = [&](int i) { return dw[i] * w[i] * scale * g[(size_t)i * d + c] / gnorm[i]; };
        int i = wave;
        for (; i + 12 < B; i += 16) {
            s0 += term(i);
            s1 += term(i + 4);
            s2 += term(i + 8);
            s3 += term(i + 12);
        }
        for (; i < B; i += 4) s0 += term(i);
    }
    red[wave][lane] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (wave == 0 && c < d) dmean[c] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}

extern "C" int nr_centrality_weights_bwd(const float* g, const float* gnorm, const float* mean, const float* w, const float* dw,
                                         int B, int d, float scale, float* dg, float* dmean, void* stream) {
    if (!g || !gnorm || !mean || !w || !dw || !dg || !dmean || B <= 0 || d <= 0) return NR_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(nr_centrality_bwd_dg_kernel, dim3((B + 3) / 4), dim3(256), 0, st, g, gnorm, mean, w, dw, B, d, scale, dg);
    hipLaunchKernelGGL(nr_centrality_bwd_dmean_kernel, dim3((d + 63) / 64), dim3(256), 0, st, g, gnorm, w, dw, B, d, scale, dmean);
    NR_LAUNCH_CHECK();
    return NR_OK;
}

// ---- the short head of the loss backward in three launches ----------------------------------------------------------------
// (1) what follows nr_row_losses_bwd: both direction frames folded back (dS = dS_dir[0] + dS_dir[1]^T, same for dG), the bank
//     column sums d_c0 / d_c1 and the logit-scale gradient -- was four launches (2 x add-transposed, grouped column sum, reduce).
__global__ __launch_bounds__(256) void nr_rowloss_bwd_finish_kernel(const float* __restrict__ dS_dir, const float* __restrict__ dG_dir,
                                                                    const float* __restrict__ dC_rows, const float* __restrict__ dls_rows,
                                                                    int B, float* __restrict__ dS, float* __restrict__ dG,
                                                                    float* __restrict__ d_c0, float* __restrict__ d_c1,
                                                                    float* __restrict__ d_ls) {
    __shared__ float t[32][33];
    __shared__ float red[4][64];
    const int T = (B + 31) / 32, CB = (B + 63) / 64;
    int blk = blockIdx.x;
    const size_t BB = (size_t)B * B;
    if (blk < 2 * T * T) {                       // out = a + b^T, tile (ti, tj) of out needs tile (tj, ti) of b
        const bool second = blk >= T * T;
        if (second) blk -= T * T;
        const float* a = second ? dG_dir : dS_dir;
        const float* b = a + BB;
        float* out = second ? dG : dS;
        const int ti = blk / T, tj = blk - ti * T;
        const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
        for (int r = ty; r < 32; r += 8) {
            const int bi = tj * 32 + r, bj = ti * 32 + tx;
            if (bi < B && bj < B) t[r][tx] = b[(size_t)bi * B + bj];
        }
        __syncthreads();
        for (int r = ty; r < 32; r += 8) {
            const int i = ti * 32 + r, j = tj * 32 + tx;
            if (i < B && j < B) out[(size_t)i * B + j] = a[(size_t)i * B + j] + t[tx][r];
        }
        return;
    }
    blk -= 2 * T * T;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (blk < 2 * CB) {                          // d_c[dir][j] = sum_i dC_rows[dir][i][j]
        const int dir = blk / CB, c = (blk - dir * CB) * 64 + lane;
        const float* a = dC_rows + (size_t)dir * BB;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        if (c < B) {
            int r = wave;
            for (; r + 12 < B; r += 16) {
                s0 += a[(size_t)r * B + c];
                s1 += a[(size_t)(r + 4) * B + c];
                s2 += a[(size_t)(r + 8) * B + c];
                s3 += a[(size_t)(r + 12) * B + c];
            }
            for (; r < B; r += 4) s0 += a[(size_t)r * B + c];
        }
        red[wave][lane] = (s0 + s1) + (s2 + s3);
        __syncthreads();
        if (wave == 0 && c < B) (dir == 0 ? d_c0 : d_c1)[c] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
        return;
    }
    float s = 0.f;                               // d logit_scale = sum of the per-row terms of both directions
    for (int i = threadIdx.x; i < 2 * B; i += 256) s += dls_rows[i];
    s = nr_wave_sum(s);
    if (lane == 0) red[0][wave] = s;
    __syncthreads();
    if (threadIdx.x == 0) d_ls[0] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
}

extern "C" int nr_rowloss_bwd_finish(const float* dS_dir, const float* dG_dir, const float* d_c_rows, const float* d_ls_rows, int B,
                                     float* dS, float* dG, float* d_c0, float* d_c1, float* d_ls, void* stream) {
    if (!dS_dir || !dG_dir || !d_c_rows || !d_ls_rows || !dS || !dG || !d_c0 || !d_c1 || !d_ls || B <= 0) return NR_EINVAL;
    const int T = (B + 31) / 32, CB = (B + 63) / 64;
    hipLaunchKernelGGL(nr_rowloss_bwd_finish_kernel, dim3(2 * T * T + 2 * CB + 1), dim3(256), 0, (hipStream_t)stream, dS_dir, dG_dir,
                       d_c_rows, d_ls_rows, B, dS, dG, d_c0, d_c1, d_ls);
    NR_LAUNCH_CHECK();
    return NR_OK;
}

// (2) nr_centrality_weights_bwd for the text AND the video tokens in one launch (was four)
struct NrCentBwdPair {
    const float *g[2], *gnorm[2], *mean[2], *w[2], *dw[2];
    float *dg[2], *dmean[2];
    int B, d;
    float scale;
};

__global__ __launch_bounds__(256) void nr_centrality_bwd_pair_kernel(NrCentBwdPair p) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int RB = (p.B + 3) / 4, DB = (p.d + 63) / 64;
    int blk = blockIdx.x;
    if (blk < 2 * RB) {
        const int m = blk / RB, i = (blk - m * RB) * 4 + wave;
        if (i >= p.B) return;
        const float inv = 1.0f / p.gnorm[m][i];
        const float a = p.dw[m][i] * p.w[m][i] * p.scale;
        const float* gi = p.g[m] + (size_t)i * p.d;
        const float* mean = p.mean[m];
        float dot = 0.f;
        for (int c = lane; c < p.d; c += 64) dot += gi[c] * inv * mean[c];
        dot = nr_wave_sum(dot);
        for (int c = lane; c < p.d; c += 64) p.dg[m][(size_t)i * p.d + c] = a * (mean[c] - gi[c] * inv * dot) * inv;
        return;
    }
    blk -= 2 * RB;
    const int m = blk / DB, c = (blk - m * DB) * 64 + lane;
    const float *g = p.g[m], *gnorm = p.gnorm[m], *w = p.w[m], *dw = p.dw[m];
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (c < p.d) {
        auto term = [&](int i) { return dw[i] * w[i] * p.scale * g[(size_t)i * p.d + c] / gnorm[i]; };
        int i = wave;
        for (; i + 12 < p.B; i += 16) {
            s0 += term(i);
            s1 += term(i + 4);
            s2 += term(i + 8);
            s3 += term(i + 12);
        }
        for (; i < p.B; i += 4) s0 += term(i);
    }
    red[wave][lane] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (wave == 0 && c < p.d) p.dmean[m][c] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}

extern "C" int nr_centrality_weights_bwd_pair(const float* g_t, const float* gnorm_t, const float* mean_t, const float* w_t,
                                              const float* dw_t, const float* g_v, const float* gnorm_v, const float* mean_v,
                                              const float* w_v, const float* dw_v, int B, int d, float scale, float* dg_t,
                                              float* dmean_t, float* dg_v, float* dmean_v, void* stream) {
    if (!g_t || !gnorm_t || !mean_t || !w_t || !dw_t || !g_v || !gnorm_v || !mean_v || !w_v || !dw_v || !dg_t || !dmean_t || !dg_v ||
        !dmean_v || B <= 0 || d <= 0)
        return NR_EINVAL;
    NrCentBwdPair p{{g_t, g_v}, {gnorm_t, gnorm_v}, {mean_t, mean_v}, {w_t, w_v}, {dw_t, dw_v}, {dg_t, dg_v}, {dmean_t, dmean_v}, B, d, scale};
    hipLaunchKernelGGL(nr_centrality_bwd_pair_kernel, dim3(2 * ((B + 3) / 4) + 2 * ((d + 63) / 64)), dim3(256), 0, (hipStream_t)stream, p);
    NR_LAUNCH_CHECK();
    return NR_OK;
}

// (3) gradient of the global logits G = gt gv^T with the centrality part added in:  d_gt = dG gv + add_t,  d_gv = dG^T gt + add_v
//     (fp32 FMA; B x B x d is a few MFLOP -- was two library GEMMs and two copies)
__global__ __launch_bounds__(256) void nr_global_logits_bwd_kernel(const float* __restrict__ dG, const float* __restrict__ gt,
                                                                   const float* __restrict__ gv, const float* __restrict__ add_t,
                                                                   const float* __restrict__ add_v, int B, int d,
                                                                   float* __restrict__ d_gt, float* __restrict__ d_gv) {
    __shared__ float sA[32][33], sX[32][33];
    const int TB = (B + 31) / 32, TD = (d + 31) / 32;
    int blk = blockIdx.x;
    const bool video = blk >= TB * TD;
    if (video) blk -= TB * TD;
    const int ti = blk / TD, tc = blk - ti * TD;
    const float* X = video ? gt : gv;
    const float* add = video ? add_v : add_t;
    float* out = video ? d_gv : d_gt;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;         // thread: column tx, rows ty, ty+8, ty+16, ty+24
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < B; k0 += 32) {
        for (int r = ty; r < 32; r += 8) {
            const int i = ti * 32 + r, k = k0 + tx;
            // A[i][k] = dG[i][k] (text) or dG[k][i] (video): staged as sA[row][k]
            float a = 0.f;
            if (i < B && k < B) a = video ? dG[(size_t)k * B + i] : dG[(size_t)i * B + k];
            sA[r][tx] = a;
            const int kk = k0 + r, c = tc * 32 + tx;
            sX[r][tx] = (kk < B && c < d) ? X[(size_t)kk * d + c] : 0.f;
        }
        __syncthreads();
#pragma unroll 8
        for (int k = 0; k < 32; ++k) {
            const float x = sX[k][tx];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_fmaf(sA[ty + 8 * j][k], x, acc[j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = ti * 32 + ty + 8 * j, c = tc * 32 + tx;
        if (i < B && c < d) out[(size_t)i * d + c] = acc[j] + add[(size_t)i * d + c];
    }
}

extern "C" int nr_global_logits_bwd(const float* dG, const float* gt, const float* gv, const float* add_t, const float* add_v, int B,
                                    int d, float* d_gt, float* d_gv, void* stream) {
    if (!dG || !gt || !gv || !add_t || !add_v || !d_gt || !d_gv || B <= 0 || d <= 0) return NR_EINVAL;
    const int TB = (B + 31) / 32, TD = (d + 31) / 32;
    hipLaunchKernelGGL(nr_global_logits_bwd_kernel, dim3(2 * TB * TD), dim3(256), 0, (hipStream_t)stream, dG, gt, gv, add_t, add_v, B, d,
                       d_gt, d_gv);
    NR_LAUNCH_CHECK();
    return NR_OK;
}

// ---- layout helpers ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void nr_add_transposed_kernel(const float* __restrict__ a, const float* __restrict__ b, int B,
                                                                float* __restrict__ out) {
    __shared__ float t[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    // tile (blockIdx.y, blockIdx.x) of `out`; needs tile (blockIdx.x, blockIdx.y) of b
    for (int r = ty; r < 32; r += 8) {
        int bi = blockIdx.x * 32 + r, bj = blockIdx.y * 32 + tx;
        if (bi < B && bj < B) t[r][tx] = b[(size_t)bi * B + bj];
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        int i = blockIdx.y * 32 + r, j = blockIdx.x * 32 + tx;
        if (i < B && j < B) out[(size_t)i * B + j] = a[(size_t)i * B + j] + t[tx][r];
    }
}

extern "C" int nr_add_transposed(const float* a, const float* b, int B, float* out, void* stream) {
    if (!a || !b || !out || B <= 0) return NR_EINVAL;
    dim3 grid((B + 31) / 32, (B + 31) / 32);
    hipLaunchKernelGGL(nr_add_transposed_kernel, grid, dim3(256), 0, (hipStream_t)stream, a, b, B, out);
    NR_LAUNCH_CHECK();
    return NR_OK;
}

__global__ __launch_bounds__(256) void nr_colsum_kernel(const float* __restrict__ a, int rows, int cols, float* __restrict__ out) {
    __shared__ float red[4][64];                 // same shape as nr_centrality_bwd_dmean_kernel above
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (c < cols) {
        int r = wave;
        for (; r + 12 < rows; r += 16) {
            s0 += a[(size_t)r * cols + c];
            s1 += a[(size_t)(r + 4) * cols + c];
            s2 += a[(size_t)(r + 8) * cols + c];
            s3 += a[(size_t)(r + 12) * cols + c];
        }
        for (; r < rows; r += 4) s0 += a[(size_t)r * cols + c];
    }
    red[wave][lane] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (wave == 0 && c < cols) out[c] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}

extern "C" int nr_colsum(const float* a, int rows, int cols, float* out, void* stream) {
    if (!a || !out || rows <= 0 || cols <= 0) return NR_EINVAL;
    hipLaunchKernelGGL(nr_colsum_kernel, dim3((cols + 63) / 64), dim3(256), 0, (hipStream_t)stream, a, rows, cols, out);
    NR_LAUNCH_CHECK();
    return NR_OK;
}
