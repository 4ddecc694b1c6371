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
#include "nr_ctm_bodies.h"
#include "../../include/nr_hip.h"

// (C > 512: sixteen channels per lane need more than the 128 registers of a 1024-thread workgroup: 512 threads there)
template <int CPL, int THREADS = CF_THREADS>
__global__ __launch_bounds__(THREADS) void nr_ctm_front_kernel(NrCtmFrontArgs p) {
    extern __shared__ __attribute__((aligned(16))) float sx[];      // [N][C] normalised tokens
    nr_ctm_front_body<CPL, THREADS>(p, blockIdx.x, sx);
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
    const bool small = C <= 512;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(small ? (const void*)nr_ctm_front_kernel<8> : (const void*)nr_ctm_front_kernel<CF_MAX_CPL, 512>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    NrCtmFrontArgs p{y, mask, ln_w, ln_b, sc_w, sc_b, n1_w, n1_b, eps, 1.0f / sqrtf((float)C), N, C, xn, kvn, score, tokw, dist, smax,
                     kvn_hi, kvn_lo};
    if (small) hipLaunchKernelGGL(nr_ctm_front_kernel<8>, dim3(n_samples), dim3(CF_THREADS), lds, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((nr_ctm_front_kernel<CF_MAX_CPL, 512>), dim3(n_samples), dim3(512), lds, (hipStream_t)stream, p);
    NR_LAUNCH_CHECK();
    return NR_OK;
}

__global__ __launch_bounds__(BK_THREADS) void nr_ctm_back_kernel(NrCtmBackArgs p, int use_lds) {
    extern __shared__ __attribute__((aligned(16))) float sxn[];
    nr_ctm_back_body(p, blockIdx.x, use_lds ? sxn : nullptr);
}

extern "C" int nr_ctm_back(const float* dist, const float* smax, const float* mask, const float* noise, const float* xn,
                           const float* tokw, int n_samples, int N, int C, int k, int cluster_num, const float* n1_w,
                           const float* n1_b, const float* proj_b, float eps, float* merged, float* merged_pb, float* qn,
                           int64_t* assign, void* stream) {
    if (!dist || !smax || !noise || !xn || !tokw || !n1_w || !n1_b || !proj_b || !merged || !merged_pb || !qn) return NR_EINVAL;
    if (n_samples <= 0 || N <= 0 || k <= 0 || k > N || cluster_num <= 0 || cluster_num > N) return NR_EINVAL;
    if (N > 64 || C <= 0 || C > 1024 || (C % 128) != 0 || cluster_num * (C / 128) > 16 * BK_MAXJ) return NR_EUNSUPPORTED;
    NrCtmBackArgs p{dist, smax, mask, noise, xn, tokw, n1_w, n1_b, proj_b, n_samples, N, C, k, cluster_num, eps,
                    merged, merged_pb, qn, assign, nullptr, nullptr};
    size_t lds = (size_t)N * C * sizeof(float);
    const int use_lds = lds <= 96 * 1024 && (N * C) % 256 == 0;
    if (!use_lds) lds = 0;
    if (lds > 40 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)nr_ctm_back_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(nr_ctm_back_kernel, dim3(n_samples), dim3(BK_THREADS), lds, (hipStream_t)stream, p, use_lds);
    NR_LAUNCH_CHECK();
    return NR_OK;
}
