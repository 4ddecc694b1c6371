// Two-launch form of the non-GEMM part of one CTM stage (reference cluster.py:689-717 with
// cluster_dpc_knn :453-509 and merge_tokens :512-561), one workgroup per sample:
//   nr_ctm_front   rows of y = x + conv(x):  LayerNorm -> score (+mask) -> exp -> the block's norm1,
//                  then -- with the sample's normalised tokens still in LDS -- all pairwise
//                  distances and the sample's maximum           (= nr_ctm_norm_score + nr_dpc_dist)
//   nr_ctm_back    DPC-KNN assignment from the distances (needs the GLOBAL maximum, hence the launch
//                  boundary) followed by the weighted cluster means and their norm1
//                                                               (= nr_dpc_assign + nr_merge_ln)
// Same arithmetic as the four separate kernels (nr_ctm.hip, nr_cluster.hip), which stay exported and
// tested; the fusion only removes two launches and two global round trips per stage from the step's
// critical path.
#include "nr_common.h"
#include "../../include/nr_hip.h"

#define CF_THREADS 1024
#define CF_MAX_CPL 16      // C <= 1024

struct NrCtmFrontArgs {
    const float *y, *mask, *ln_w, *ln_b, *sc_w, *sc_b, *n1_w, *n1_b;
    float eps, inv_sqrt_c;
    int N, C;
    float *xn, *kvn, *score, *tokw, *dist, *smax;
    uint16_t *kvn_hi, *kvn_lo;       // when set, norm1(xn) is written split-bf16 (operand of the kv GEMM) instead of f32
};

__global__ __launch_bounds__(CF_THREADS) void nr_ctm_front_kernel(NrCtmFrontArgs p) {
    extern __shared__ __attribute__((aligned(16))) float sx[];      // [N][C] normalised tokens
    __shared__ float s_wmax[CF_THREADS / 64];
    const int b = blockIdx.x, N = p.N, C = p.C;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int NW = CF_THREADS / 64;
    const int cpl = C / 64;
    // ---- phase 1: one wave per token row ------------------------------------------------------------
    for (int r = wave; r < N; r += NW) {
        const size_t row = (size_t)b * N + r;
        const float* yr = p.y + row * C;
        float v[CF_MAX_CPL];
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < CF_MAX_CPL; ++q) {
            v[q] = q < cpl ? yr[q * 64 + lane] : 0.f;
            s += v[q];
        }
        const float mu = nr_wave_sum(s) / (float)C;
        float var = 0.f;
#pragma unroll
        for (int q = 0; q < CF_MAX_CPL; ++q)
            if (q < cpl) { float dlt = v[q] - mu; var += dlt * dlt; }
        const float rstd = rsqrtf(nr_wave_sum(var) / (float)C + p.eps);
        float dot = 0.f, s2 = 0.f;
#pragma unroll
        for (int q = 0; q < CF_MAX_CPL; ++q)
            if (q < cpl) {
                int c = q * 64 + lane;
                v[q] = (v[q] - mu) * rstd * p.ln_w[c] + p.ln_b[c];
                p.xn[row * C + c] = v[q];
                sx[r * C + c] = v[q];
                dot += v[q] * p.sc_w[c];
                s2 += v[q];
            }
        float sc = nr_wave_sum(dot) + p.sc_b[0];
        if (p.mask && p.mask[row] == 0.f) sc = -INFINITY;
        if (lane == 0) {
            p.score[row] = sc;
            p.tokw[row] = expf(sc);
        }
        const float mu2 = nr_wave_sum(s2) / (float)C;
        float var2 = 0.f;
#pragma unroll
        for (int q = 0; q < CF_MAX_CPL; ++q)
            if (q < cpl) { float dlt = v[q] - mu2; var2 += dlt * dlt; }
        const float rstd2 = rsqrtf(nr_wave_sum(var2) / (float)C + p.eps);
#pragma unroll
        for (int q = 0; q < CF_MAX_CPL; ++q)
            if (q < cpl) {
                int c = q * 64 + lane;
                float kv = (v[q] - mu2) * rstd2 * p.n1_w[c] + p.n1_b[c];
                if (p.kvn_hi) {
                    uint16_t h = nr_f2bf(kv);
                    p.kvn_hi[row * C + c] = h;
                    p.kvn_lo[row * C + c] = nr_f2bf(kv - nr_bf2f(h));
                } else {
                    p.kvn[row * C + c] = kv;
                }
            }
    }
    __syncthreads();
    // ---- phase 2: pairwise distances, wave per row of the upper triangle -------------------------------
    float wmax = 0.f;
    float* db = p.dist + (size_t)b * N * N;
    for (int i = wave; i < N; i += NW) {
        float xi[CF_MAX_CPL];
#pragma unroll
        for (int q = 0; q < CF_MAX_CPL; ++q) xi[q] = q < cpl ? sx[i * C + q * 64 + lane] : 0.f;
        if (lane == 0) db[i * N + i] = 0.f;
        for (int j = i + 1; j < N; j += 2) {
            const bool two = j + 1 < N;
            const float* xj0 = sx + j * C;
            const float* xj1 = sx + (two ? j + 1 : j) * C;
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int q = 0; q < CF_MAX_CPL; ++q)
                if (q < cpl) {
                    int c = q * 64 + lane;
                    float d0 = xi[q] - xj0[c], d1 = xi[q] - xj1[c];
                    s0 += d0 * d0;
                    s1 += d1 * d1;
                }
            s0 = nr_wave_sum(s0);
            s1 = nr_wave_sum(s1);
            float dv0 = sqrtf(s0) * p.inv_sqrt_c, dv1 = sqrtf(s1) * p.inv_sqrt_c;
            wmax = fmaxf(wmax, dv0);
            if (two) wmax = fmaxf(wmax, dv1);
            if (lane == 0) {
                db[i * N + j] = dv0;
                db[j * N + i] = dv0;
                if (two) {
                    db[i * N + j + 1] = dv1;
                    db[(j + 1) * N + i] = dv1;
                }
            }
        }
    }
    if (lane == 0) s_wmax[wave] = wmax;
    __syncthreads();
    if (tid == 0) {
        float m = 0.f;
        for (int w = 0; w < NW; ++w) m = fmaxf(m, s_wmax[w]);
        p.smax[b] = m;
    }
}

extern "C" int nr_ctm_front(const float* y, const float* mask, int n_samples, int N, int C, const float* ln_w,
                            const float* ln_b, const float* sc_w, const float* sc_b, const float* n1_w, const float* n1_b,
                            float eps, float* xn, float* kvn, uint16_t* kvn_hi, uint16_t* kvn_lo, float* score, float* tokw,
                            float* dist, float* smax, void* stream) {
    if (!y || !ln_w || !ln_b || !sc_w || !sc_b || !n1_w || !n1_b || !xn || !score || !tokw || !dist || !smax) return NR_EINVAL;
    if (!kvn && !(kvn_hi && kvn_lo)) return NR_EINVAL;
    if (n_samples <= 0 || N <= 0 || N > 64 || C <= 0 || (C % 64) != 0 || C > 64 * CF_MAX_CPL) return NR_EUNSUPPORTED;
    size_t lds = (size_t)N * C * sizeof(float);
    if (lds > 150 * 1024) return NR_EUNSUPPORTED;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)nr_ctm_front_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    NrCtmFrontArgs p{y, mask, ln_w, ln_b, sc_w, sc_b, n1_w, n1_b, eps, 1.0f / sqrtf((float)C), N, C, xn, kvn, score, tokw, dist, smax,
                     kvn_hi, kvn_lo};
    hipLaunchKernelGGL(nr_ctm_front_kernel, dim3(n_samples), dim3(CF_THREADS), lds, (hipStream_t)stream, p);
    NR_LAUNCH_CHECK();
    return NR_OK;
}

// ---- assignment + merge -------------------------------------------------------------------------------------
struct NrCtmBackArgs {
    const float *dist, *smax, *mask, *noise, *xn, *tokw, *n1_w, *n1_b, *proj_b;
    int n_samples, N, C, k, cnum;
    float eps;
    float *merged, *merged_pb, *qn;
    int64_t* assign;
};

__global__ __launch_bounds__(256) void nr_ctm_back_kernel(NrCtmBackArgs p) {
    __shared__ float sd[64][65];
    __shared__ float s_density[64], s_score[64], s_share[64], s_tot[64];
    __shared__ int s_centre[64], s_assign[64];
    __shared__ float s_red[2][4];
    const int b = blockIdx.x, N = p.N, C = p.C, cnum = p.cnum;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // global maximum over all samples (cluster.py:473-475)
    float g = 0.f;
    for (int i = tid; i < p.n_samples; i += 256) g = fmaxf(g, p.smax[i]);
    g = nr_wave_max(g);
    if (lane == 0) s_red[0][wave] = g;
    __syncthreads();
    const float far = fmaxf(fmaxf(s_red[0][0], s_red[0][1]), fmaxf(s_red[0][2], s_red[0][3])) + 1.0f;
    const float* db = p.dist + (size_t)b * N * N;
    const float* mb = p.mask ? p.mask + (size_t)b * N : nullptr;
    float lmax = 0.f;
    for (int e = tid; e < N * N; e += 256) {
        int i = e / N, j = e - i * N;
        float dv = db[e];
        if (mb && !(mb[j] > 0.f)) dv = far;
        sd[i][j] = dv;
        lmax = fmaxf(lmax, dv);
    }
    lmax = nr_wave_max(lmax);
    if (lane == 0) s_red[1][wave] = lmax;
    __syncthreads();
    const float dmax = fmaxf(fmaxf(s_red[1][0], s_red[1][1]), fmaxf(s_red[1][2], s_red[1][3]));
    for (int i = wave; i < N; i += 4) {               // local density
        float v = lane < N ? sd[i][lane] : INFINITY;
        float acc = 0.f;
        for (int r = 0; r < p.k; ++r) {
            float m = v;
            int idx = lane;
            nr_wave_argmin(m, idx);
            acc += m * m;
            if (lane == idx) v = INFINITY;
        }
        if (lane == 0) {
            float dens = expf(-acc / (float)p.k) + p.noise[(size_t)b * N + i] * 1e-6f;
            if (mb) dens *= (mb[i] > 0.f) ? 1.0f : 0.0f;
            s_density[i] = dens;
        }
    }
    __syncthreads();
    for (int i = wave; i < N; i += 4) {               // distance to the nearest denser token; score
        float di = s_density[i];
        float v = (lane < N && s_density[lane] > di) ? sd[i][lane] : dmax;
        v = nr_wave_min(v);
        if (lane == 0) s_score[i] = v * di;
    }
    __syncthreads();
    if (wave == 0) {                                  // top-cnum centres
        float v = lane < N ? s_score[lane] : -INFINITY;
        for (int c = 0; c < cnum; ++c) {
            float m = v;
            int idx = lane;
            nr_wave_argmax(m, idx);
            if (lane == 0) s_centre[c] = idx;
            if (lane == idx) v = -INFINITY;
        }
    }
    __syncthreads();
    if (tid < N) {                                    // nearest centre; centres join themselves
        float best = INFINITY;
        int bc = 0;
        for (int c = 0; c < cnum; ++c) {
            float dv = sd[s_centre[c]][tid];
            if (dv < best) { best = dv; bc = c; }
        }
        for (int c = 0; c < cnum; ++c)
            if (s_centre[c] == tid) bc = c;
        s_assign[tid] = bc;
        if (p.assign) p.assign[(size_t)b * N + tid] = bc;
    }
    __syncthreads();
    // ---- merge_tokens + norm1 ---------------------------------------------------------------------------
    if (tid < cnum) {
        float t = 0.f;
        for (int n = 0; n < N; ++n)
            if (s_assign[n] == tid) t += p.tokw[(size_t)b * N + n];
        s_tot[tid] = t + 1e-6f;
    }
    __syncthreads();
    if (tid < N) s_share[tid] = p.tokw[(size_t)b * N + tid] / s_tot[s_assign[tid]];
    __syncthreads();
    const float* xb = p.xn + (size_t)b * N * C;
    for (int cl = 0; cl < cnum; ++cl) {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int n = 0; n < N; ++n) {
            if (s_assign[n] != cl) continue;
            const float sh = s_share[n];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                int c = q * 256 + tid;
                if (c < C) acc[q] += xb[(size_t)n * C + c] * sh;
            }
        }
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) s += (q * 256 + tid < C) ? acc[q] : 0.f;
        s = nr_wave_sum(s);
        if (lane == 0) s_red[0][wave] = s;
        __syncthreads();
        const float mu = (s_red[0][0] + s_red[0][1] + s_red[0][2] + s_red[0][3]) / (float)C;
        float var = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (q * 256 + tid < C) { float dlt = acc[q] - mu; var += dlt * dlt; }
        var = nr_wave_sum(var);
        if (lane == 0) s_red[1][wave] = var;
        __syncthreads();
        const float rstd = rsqrtf((s_red[1][0] + s_red[1][1] + s_red[1][2] + s_red[1][3]) / (float)C + p.eps);
        const size_t o = ((size_t)b * cnum + cl) * C;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            int c = q * 256 + tid;
            if (c < C) {
                p.merged[o + c] = acc[q];
                p.merged_pb[o + c] = acc[q] + p.proj_b[c];
                p.qn[o + c] = (acc[q] - mu) * rstd * p.n1_w[c] + p.n1_b[c];
            }
        }
        __syncthreads();
    }
}

extern "C" int nr_ctm_back(const float* dist, const float* smax, const float* mask, const float* noise, const float* xn,
                           const float* tokw, int n_samples, int N, int C, int k, int cluster_num, const float* n1_w,
                           const float* n1_b, const float* proj_b, float eps, float* merged, float* merged_pb, float* qn,
                           int64_t* assign, void* stream) {
    if (!dist || !smax || !noise || !xn || !tokw || !n1_w || !n1_b || !proj_b || !merged || !merged_pb || !qn) return NR_EINVAL;
    if (n_samples <= 0 || N <= 0 || k <= 0 || k > N || cluster_num <= 0 || cluster_num > N) return NR_EINVAL;
    if (N > 64 || C <= 0 || C > 1024) return NR_EUNSUPPORTED;
    NrCtmBackArgs p{dist, smax, mask, noise, xn, tokw, n1_w, n1_b, proj_b, n_samples, N, C, k, cluster_num, eps,
                    merged, merged_pb, qn, assign};
    hipLaunchKernelGGL(nr_ctm_back_kernel, dim3(n_samples), dim3(256), 0, (hipStream_t)stream, p);
    NR_LAUNCH_CHECK();
    return NR_OK;
}
