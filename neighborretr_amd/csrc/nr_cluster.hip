// DPC-KNN token clustering: cluster id of every token (reference: cluster_dpc_knn,
// NeighborRetr/models/cluster.py:453-509).  Index-only work under no_grad in the reference; here two
// launches instead of ~25 tiny ATen kernels per call (4 calls per step):
//   nr_dpc_dist_kernel    one workgroup per sample: tokens into LDS, all N x N Euclidean distances
//                         (/sqrt(C)), 16 waves, one row of the upper triangle per wave, plus the sample's maximum;
//   nr_dpc_assign_kernel  one workgroup per sample: mask columns to (GLOBAL max + 1) -- the reference's
//                         dist_matrix.max() runs over the whole batch --, k-NN density + tie-break
//                         noise, distance to the nearest denser token, score = dist * density, top
//                         `cluster_num` centres (ties -> lower index), nearest-centre assignment.
// N <= 64 tokens per sample (one lane per token in the row-wise steps).
#include "nr_common.h"
#include "../../include/nr_hip.h"

#define DPC_DIST_THREADS 1024
#define DPC_MAX_CPL 16      // C <= 1024: at most 16 channels per lane

__global__ __launch_bounds__(DPC_DIST_THREADS) void nr_dpc_dist_kernel(const float* __restrict__ x, int N, int C, float inv_sqrt_c,
                                                                       float* __restrict__ dist, float* __restrict__ smax) {
    extern __shared__ __attribute__((aligned(16))) float sx[];      // [N][C]
    __shared__ float s_wmax[DPC_DIST_THREADS / 64];
    const int b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int NW = DPC_DIST_THREADS / 64;
    const float* xb = x + (size_t)b * N * C;
    for (int i = tid * 4; i < N * C; i += DPC_DIST_THREADS * 4)
        *reinterpret_cast<f32x4_t*>(sx + i) = *reinterpret_cast<const f32x4_t*>(xb + i);
    __syncthreads();
    float wmax = 0.f;
    float* db = dist + (size_t)b * N * N;
    const int cpl = (C + 63) / 64;
    for (int i = wave; i < N; i += NW) {      // wave per row of the upper triangle
        float xi[DPC_MAX_CPL];
#pragma unroll
        for (int q = 0; q < DPC_MAX_CPL; ++q) {
            int c = q * 64 + lane;
            xi[q] = (q < cpl && c < C) ? sx[i * C + c] : 0.f;
        }
        if (lane == 0) db[i * N + i] = 0.f;
        for (int j = i + 1; j < N; j += 2) {
            const bool two = j + 1 < N;
            const float* xj0 = sx + j * C;
            const float* xj1 = sx + (two ? j + 1 : j) * C;
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int q = 0; q < DPC_MAX_CPL; ++q) {
                int c = q * 64 + lane;
                if (q < cpl && c < C) {
                    float d0 = xi[q] - xj0[c], d1 = xi[q] - xj1[c];
                    s0 += d0 * d0;
                    s1 += d1 * d1;
                }
            }
            s0 = nr_wave_sum(s0);
            s1 = nr_wave_sum(s1);
            float dv0 = sqrtf(s0) * inv_sqrt_c, dv1 = sqrtf(s1) * inv_sqrt_c;
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
        smax[b] = m;
    }
}

__global__ __launch_bounds__(256) void nr_dpc_assign_kernel(const float* __restrict__ dist, const float* __restrict__ smax,
                                                            int n_samples, const float* __restrict__ mask,
                                                            const float* __restrict__ noise, int N, int k, int cnum,
                                                            int64_t* __restrict__ assign) {
    __shared__ float sd[64][65];
    __shared__ float s_density[64], s_score[64];
    __shared__ int s_centre[64];
    __shared__ float s_red[4];
    const int b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // global maximum over all samples (cluster.py:473-475)
    float g = 0.f;
    for (int i = tid; i < n_samples; i += 256) g = fmaxf(g, smax[i]);
    g = nr_wave_max(g);
    if (lane == 0) s_red[wave] = g;
    __syncthreads();
    const float far = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3])) + 1.0f;
    const float* db = dist + (size_t)b * N * N;
    const float* mb = mask ? mask + (size_t)b * N : nullptr;
    float lmax = 0.f;
    for (int e = tid; e < N * N; e += 256) {
        int i = e / N, j = e - i * N;
        float dv = db[e];
        if (mb && !(mb[j] > 0.f)) dv = far;
        sd[i][j] = dv;
        lmax = fmaxf(lmax, dv);
    }
    lmax = nr_wave_max(lmax);
    __syncthreads();                 // s_red reads above are done
    if (lane == 0) s_red[wave] = lmax;
    __syncthreads();
    const float dmax = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));   // per-sample max (:493)

    // ---- local density: one wave per row, lane = column ------------------------------------------
    for (int i = wave; i < N; i += 4) {
        float v = lane < N ? sd[i][lane] : INFINITY;
        float acc = 0.f;
        for (int r = 0; r < k; ++r) {
            float m = v;
            int idx = lane;
            nr_wave_argmin(m, idx);
            acc += m * m;
            if (lane == idx) v = INFINITY;
        }
        if (lane == 0) {
            float dens = expf(-acc / (float)k) + noise[(size_t)b * N + i] * 1e-6f;
            if (mb) dens *= (mb[i] > 0.f) ? 1.0f : 0.0f;
            s_density[i] = dens;
        }
    }
    __syncthreads();
    // ---- distance to the nearest denser token; score -----------------------------------------------
    for (int i = wave; i < N; i += 4) {
        float di = s_density[i];
        float v = (lane < N && s_density[lane] > di) ? sd[i][lane] : dmax;
        v = nr_wave_min(v);
        if (lane == 0) s_score[i] = v * di;
    }
    __syncthreads();
    // ---- top-cnum centres by score (descending, ties -> lower index); wave 0 ----------------------
    if (wave == 0) {
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
    // ---- nearest centre per token (argmin, first occurrence); centres join themselves --------------
    if (tid < N) {
        float best = INFINITY;
        int bc = 0;
        for (int c = 0; c < cnum; ++c) {
            float dv = sd[s_centre[c]][tid];
            if (dv < best) { best = dv; bc = c; }
        }
        for (int c = 0; c < cnum; ++c)
            if (s_centre[c] == tid) bc = c;
        assign[(size_t)b * N + tid] = bc;
    }
}

extern "C" size_t nr_dpc_workspace_bytes(int n_samples, int N) {
    return ((size_t)n_samples * N * N + n_samples) * sizeof(float);
}

extern "C" int nr_dpc_knn_assign(const float* x, const float* mask, const float* noise, int n_samples, int N, int C, int k,
                                 int cluster_num, int64_t* assign, void* workspace, void* stream) {
    if (!x || !noise || !assign || !workspace || n_samples <= 0 || N <= 0 || C <= 0) return NR_EINVAL;
    if (k <= 0 || k > N || cluster_num <= 0 || cluster_num > N) return NR_EINVAL;   // torch.topk raises the same
    if (N > 64 || (C % 4) != 0 || C > 64 * DPC_MAX_CPL) return NR_EUNSUPPORTED;
    size_t lds = (size_t)N * C * sizeof(float);
    if (lds > 150 * 1024) return NR_EUNSUPPORTED;
    float* dist = reinterpret_cast<float*>(workspace);
    float* smax = dist + (size_t)n_samples * N * N;
    hipStream_t st = (hipStream_t)stream;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)nr_dpc_dist_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(nr_dpc_dist_kernel, dim3(n_samples), dim3(DPC_DIST_THREADS), lds, st, x, N, C, 1.0f / sqrtf((float)C), dist, smax);
    hipLaunchKernelGGL(nr_dpc_assign_kernel, dim3(n_samples), dim3(256), 0, st, dist, smax, n_samples, mask, noise, N, k,
                       cluster_num, assign);
    NR_LAUNCH_CHECK();
    return NR_OK;
}
