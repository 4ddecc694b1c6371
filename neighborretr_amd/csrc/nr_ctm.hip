// Forward kernels of the token-clustering stage (CTM + TCBlock, reference cluster.py:670-717,
// :834-888, :938-965) around the library GEMMs: everything that is not a plain matrix product is
// fused into four launches per stage instead of ~45 ATen kernels.
//   nr_shift_concat        [B,N,C] -> [B*N,3C] = (x[n-1] | x[n] | x[n+1])  (operand of the k=3 conv GEMM)
//   nr_ctm_norm_score      LayerNorm -> token score (+mask -> -inf) -> exp -> the block's norm1
//   nr_merge_ln            weighted mean of every cluster's tokens (merge_tokens) + norm1 of the result
//   nr_tc_attention        8-head attention of the merged tokens over the un-merged ones, score-biased
#include "nr_ctm_bodies.h"
#include "../../include/nr_hip.h"

// ---- x[n-1] | x[n] | x[n+1] ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void nr_shift_concat_kernel(const float* __restrict__ x, int N, int C, float* __restrict__ out) {
    const int row = blockIdx.x;                 // b*N + n
    const int n = row % N;
    float* o = out + (size_t)row * 3 * C;
    for (int k = 0; k < 3; ++k) {
        const int nn = n + k - 1;
        const bool ok = nn >= 0 && nn < N;
        const float* src = x + (size_t)(row + k - 1) * C;
        for (int c = threadIdx.x * 4; c < C; c += 1024) {
            f32x4_t v = {0.f, 0.f, 0.f, 0.f};
            if (ok) v = *reinterpret_cast<const f32x4_t*>(src + c);
            *reinterpret_cast<f32x4_t*>(o + k * C + c) = v;
        }
    }
}

extern "C" int nr_shift_concat(const float* x, int n_samples, int N, int C, float* out, void* stream) {
    if (!x || !out || n_samples <= 0 || N <= 0 || C <= 0 || (C % 4) != 0) return NR_EINVAL;
    hipLaunchKernelGGL(nr_shift_concat_kernel, dim3(n_samples * N), dim3(256), 0, (hipStream_t)stream, x, N, C, out);
    NR_LAUNCH_CHECK();
    return NR_OK;
}

// ---- LayerNorm + score + exp + norm1: one wave per token row ------------------------------------------
#define CTM_MAX_CPL 16     // C <= 1024
__global__ __launch_bounds__(256) void nr_ctm_norm_score_kernel(const float* __restrict__ y, const float* __restrict__ mask,
                                                                int n_rows, int C, const float* __restrict__ ln_w,
                                                                const float* __restrict__ ln_b, const float* __restrict__ sc_w,
                                                                const float* __restrict__ sc_b, const float* __restrict__ n1_w,
                                                                const float* __restrict__ n1_b, float eps,
                                                                float* __restrict__ xn, float* __restrict__ kvn,
                                                                float* __restrict__ score, float* __restrict__ tokw) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    const int cpl = C / 64;
    const float* yr = y + (size_t)row * C;
    float v[CTM_MAX_CPL];
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < CTM_MAX_CPL; ++q) {
        v[q] = q < cpl ? yr[q * 64 + lane] : 0.f;
        s += v[q];
    }
    const float mu = nr_wave_sum(s) / (float)C;
    float var = 0.f;
#pragma unroll
    for (int q = 0; q < CTM_MAX_CPL; ++q)
        if (q < cpl) { float dlt = v[q] - mu; var += dlt * dlt; }
    const float rstd = rsqrtf(nr_wave_sum(var) / (float)C + eps);
    float dot = 0.f, s2 = 0.f;
#pragma unroll
    for (int q = 0; q < CTM_MAX_CPL; ++q)
        if (q < cpl) {
            int c = q * 64 + lane;
            v[q] = (v[q] - mu) * rstd * ln_w[c] + ln_b[c];
            xn[(size_t)row * C + c] = v[q];
            dot += v[q] * sc_w[c];
            s2 += v[q];
        }
    float sc = nr_wave_sum(dot) + sc_b[0];
    if (mask && mask[row] == 0.f) sc = -INFINITY;           // cluster.py:703-705
    if (lane == 0) {
        score[row] = sc;
        tokw[row] = expf(sc);
    }
    // the TCBlock's norm1 applied to the un-merged tokens (cluster.py:958-960)
    const float mu2 = nr_wave_sum(s2) / (float)C;
    float var2 = 0.f;
#pragma unroll
    for (int q = 0; q < CTM_MAX_CPL; ++q)
        if (q < cpl) { float dlt = v[q] - mu2; var2 += dlt * dlt; }
    const float rstd2 = rsqrtf(nr_wave_sum(var2) / (float)C + eps);
#pragma unroll
    for (int q = 0; q < CTM_MAX_CPL; ++q)
        if (q < cpl) {
            int c = q * 64 + lane;
            kvn[(size_t)row * C + c] = (v[q] - mu2) * rstd2 * n1_w[c] + n1_b[c];
        }
}

extern "C" int nr_ctm_norm_score(const float* y, const float* mask, int n_rows, int C, const float* ln_w, const float* ln_b,
                                 const float* sc_w, const float* sc_b, const float* n1_w, const float* n1_b, float eps,
                                 float* xn, float* kvn, float* score, float* tokw, void* stream) {
    if (!y || !ln_w || !ln_b || !sc_w || !sc_b || !n1_w || !n1_b || !xn || !kvn || !score || !tokw) return NR_EINVAL;
    if (n_rows <= 0 || C <= 0 || (C % 64) != 0 || C > 64 * CTM_MAX_CPL) return NR_EUNSUPPORTED;
    hipLaunchKernelGGL(nr_ctm_norm_score_kernel, dim3((n_rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, y, mask, n_rows, C,
                       ln_w, ln_b, sc_w, sc_b, n1_w, n1_b, eps, xn, kvn, score, tokw);
    NR_LAUNCH_CHECK();
    return NR_OK;
}

// ---- merge_tokens + norm1: one workgroup per sample ------------------------------------------------------
// merged[c] = sum_{n in cluster c} x[n] * w[n] / (sum_{n in c} w[n] + 1e-6)   (cluster.py:536-547)
#define MERGE_MAX_C 64
__global__ __launch_bounds__(256) void nr_merge_ln_kernel(const float* __restrict__ xn, const int64_t* __restrict__ assign,
                                                          const float* __restrict__ tokw, int N, int C, int cnum,
                                                          const float* __restrict__ n1_w, const float* __restrict__ n1_b,
                                                          const float* __restrict__ proj_b, float eps,
                                                          float* __restrict__ merged, float* __restrict__ merged_pb,
                                                          float* __restrict__ qn) {
    __shared__ float s_share[64];
    __shared__ int s_assign[64];
    __shared__ float s_tot[MERGE_MAX_C];
    __shared__ float s_red[2][4];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < N) s_assign[tid] = (int)assign[(size_t)b * N + tid];
    __syncthreads();
    if (tid < cnum) {
        float t = 0.f;
        for (int n = 0; n < N; ++n)
            if (s_assign[n] == tid) t += tokw[(size_t)b * N + n];
        s_tot[tid] = t + 1e-6f;
    }
    __syncthreads();
    if (tid < N) s_share[tid] = tokw[(size_t)b * N + tid] / s_tot[s_assign[tid]];
    __syncthreads();
    const float* xb = xn + (size_t)b * N * C;
    for (int cl = 0; cl < cnum; ++cl) {
        // every thread owns channels tid, tid+256, ...; accumulate then LayerNorm across the workgroup
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
        const float rstd = rsqrtf((s_red[1][0] + s_red[1][1] + s_red[1][2] + s_red[1][3]) / (float)C + eps);
        const size_t o = ((size_t)b * cnum + cl) * C;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            int c = q * 256 + tid;
            if (c < C) {
                merged[o + c] = acc[q];
                merged_pb[o + c] = acc[q] + proj_b[c];
                qn[o + c] = (acc[q] - mu) * rstd * n1_w[c] + n1_b[c];
            }
        }
        __syncthreads();           // s_red reused by the next cluster
    }
}

extern "C" int nr_merge_ln(const float* xn, const int64_t* assign, const float* tokw, int n_samples, int N, int C, int cnum,
                           const float* n1_w, const float* n1_b, const float* proj_b, float eps, float* merged,
                           float* merged_pb, float* qn, void* stream) {
    if (!xn || !assign || !tokw || !n1_w || !n1_b || !proj_b || !merged || !merged_pb || !qn) return NR_EINVAL;
    if (n_samples <= 0 || N <= 0 || N > 64 || cnum <= 0 || cnum > MERGE_MAX_C || C <= 0 || C > 1024) return NR_EUNSUPPORTED;
    hipLaunchKernelGGL(nr_merge_ln_kernel, dim3(n_samples), dim3(256), 0, (hipStream_t)stream, xn, assign, tokw, N, C, cnum,
                       n1_w, n1_b, proj_b, eps, merged, merged_pb, qn);
    NR_LAUNCH_CHECK();
    return NR_OK;
}

// ---- score-biased multi-head attention (body in nr_ctm_bodies.h) ---------------------------------------------
__global__ __launch_bounds__(1024) void nr_tc_attention_kernel(NrAttnArgs a, int use_lds) {
    extern __shared__ __attribute__((aligned(16))) float skv[];
    nr_tc_attention_body(a, blockIdx.x, use_lds ? skv : nullptr);
}

extern "C" int nr_tc_attention(const float* q, const float* kv, const float* score, int n_samples, int N, int C, int cnum,
                               int H, float* out, void* stream) {
    if (!q || !kv || !score || !out || n_samples <= 0 || N <= 0 || cnum <= 0 || H <= 0) return NR_EINVAL;
    if (N > 64 || C != H * 64) return NR_EUNSUPPORTED;
    NrAttnArgs a{q, kv, score, N, C, cnum, H, 1.0f / sqrtf(64.0f), out, nullptr, nullptr};
    size_t lds = (size_t)N * (2 * C + 4) * sizeof(float);
    const int use_lds = lds <= 128 * 1024;
    if (!use_lds) lds = 0;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)nr_tc_attention_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(nr_tc_attention_kernel, dim3(n_samples), dim3(1024), lds, (hipStream_t)stream, a, use_lds);
    NR_LAUNCH_CHECK();
    return NR_OK;
}
