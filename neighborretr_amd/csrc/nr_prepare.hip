// Token preparation: F.normalize + mask + bf16 hi/lo split (+ column sums for the centrality
// weights).  Reference: modeling.py:495-496 (normalize), :500-501 (mask multiplies),
// :413-424 (mean of the normalised tokens, padding included).
// HBM-bound streaming kernel: one wave per token row, 16-byte loads, 8-byte bf16x4 stores.
#include "nr_common.h"
#include "../../include/nr_hip.h"

#ifndef NR_PREP_MAX_PARTS
#define NR_PREP_MAX_PARTS 256    // column-sum partials = workgroups of a launch that wants them: 16 tokens each up to this many (64 left a
                                 // 3072-token set with 12 dependent row round trips per wave: 14 us; 256: 4)
#endif
#define NR_PREP_MAX_GRID 2048
#define NR_PREP_MAX_CHUNKS 4   // d <= 1024

extern "C" int nr_prepare_parts(int n_tok) {
    int p = (n_tok + 15) / 16;
    if (p < 1) p = 1;
    return p > NR_PREP_MAX_PARTS ? NR_PREP_MAX_PARTS : p;
}

// One tensor's share of a launch: workgroup `bid` of `nblocks`.
template <int CH>   // CH = d / 256 float4 chunks per lane
__device__ __forceinline__ void nr_prepare_body(const float* __restrict__ x, const float* __restrict__ mask,
                                                int n_tok, int d, int normalize, uint16_t* __restrict__ hi,
                                                uint16_t* __restrict__ lo, float* __restrict__ norm_out,
                                                float* __restrict__ colsum_part, const int bid, const int nblocks) {
    __shared__ float s_col[4][CH * 256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float csum[CH][4];
#pragma unroll
    for (int c = 0; c < CH; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) csum[c][e] = 0.f;

    for (int row = bid * 4 + wave; row < n_tok; row += nblocks * 4) {
        const float* xr = x + (size_t)row * d;
        f32x4_t v[CH];
        float ss = 0.f;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            v[c] = *reinterpret_cast<const f32x4_t*>(xr + (c * 64 + lane) * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) ss += v[c][e] * v[c][e];
        }
        ss = nr_wave_sum(ss);
        float nrm = fmaxf(sqrtf(ss), 1e-12f);          // F.normalize: x / max(||x||, eps)
        float inv = normalize ? 1.0f / nrm : 1.0f;
        float mk = mask ? mask[row] : 1.0f;
        if (norm_out && lane == 0) norm_out[row] = nrm;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            uint16_t h[4], l[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float xn = v[c][e] * inv;
                csum[c][e] += xn;                       // unmasked (padding tokens count)
                float y = xn * mk;
                h[e] = nr_f2bf(y);
                l[e] = nr_f2bf(y - nr_bf2f(h[e]));
            }
            size_t o = (size_t)row * d + (c * 64 + lane) * 4;
            uint2 ph = make_uint2((uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16));
            *reinterpret_cast<uint2*>(hi + o) = ph;
            if (lo) {
                uint2 pl = make_uint2((uint32_t)l[0] | ((uint32_t)l[1] << 16), (uint32_t)l[2] | ((uint32_t)l[3] << 16));
                *reinterpret_cast<uint2*>(lo + o) = pl;
            }
        }
    }
    if (colsum_part) {
#pragma unroll
        for (int c = 0; c < CH; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) s_col[wave][(c * 64 + lane) * 4 + e] = csum[c][e];
        __syncthreads();
        for (int k = threadIdx.x; k < d; k += 256)
            colsum_part[(size_t)bid * d + k] = (s_col[0][k] + s_col[1][k]) + (s_col[2][k] + s_col[3][k]);
    }
}

template <int CH>
__global__ __launch_bounds__(256) void nr_prepare_kernel(const float* __restrict__ x, const float* __restrict__ mask,
                                                         int n_tok, int d, int normalize, uint16_t* __restrict__ hi,
                                                         uint16_t* __restrict__ lo, float* __restrict__ norm_out,
                                                         float* __restrict__ colsum_part) {
    nr_prepare_body<CH>(x, mask, n_tok, d, normalize, hi, lo, norm_out, colsum_part, blockIdx.x, gridDim.x);
}

// Two tensors (the batch's text and video tokens) in ONE launch: workgroups [0, nb0) take the first, the rest the second.
struct NrPrepareOne {
    const float *x, *mask;
    int n_tok;
    uint16_t *hi, *lo;
    float *norm, *colsum_part;
};
template <int CH>
__global__ __launch_bounds__(256) void nr_prepare_pair_kernel(NrPrepareOne a, NrPrepareOne b, int d, int normalize, int nb0) {
    const bool second = (int)blockIdx.x >= nb0;
    const NrPrepareOne& p = second ? b : a;
    nr_prepare_body<CH>(p.x, p.mask, p.n_tok, d, normalize, p.hi, p.lo, p.norm, p.colsum_part, second ? blockIdx.x - nb0 : blockIdx.x,
                        second ? gridDim.x - nb0 : nb0);
}

extern "C" int nr_prepare_tokens(const float* x, const float* mask, int n_tok, int d, int normalize, uint16_t* hi,
                                 uint16_t* lo, float* norm, float* colsum_part, void* stream) {
    if (!x || !hi || n_tok <= 0 || d <= 0) return NR_EINVAL;
    if ((d % 256) != 0 || d / 256 > NR_PREP_MAX_CHUNKS) return NR_EUNSUPPORTED;
    // column sums wanted: one partial per workgroup, so few workgroups (the batch tensors are small);
    // otherwise (memory bank) enough workgroups to stream at HBM rate
    int grid = nr_prepare_parts(n_tok);
    if (!colsum_part) {
        grid = (n_tok + 15) / 16;
        if (grid > NR_PREP_MAX_GRID) grid = NR_PREP_MAX_GRID;
    }
    hipStream_t st = (hipStream_t)stream;
    switch (d / 256) {
        case 1: hipLaunchKernelGGL(nr_prepare_kernel<1>, dim3(grid), dim3(256), 0, st, x, mask, n_tok, d, normalize, hi, lo, norm, colsum_part); break;
        case 2: hipLaunchKernelGGL(nr_prepare_kernel<2>, dim3(grid), dim3(256), 0, st, x, mask, n_tok, d, normalize, hi, lo, norm, colsum_part); break;
        case 3: hipLaunchKernelGGL(nr_prepare_kernel<3>, dim3(grid), dim3(256), 0, st, x, mask, n_tok, d, normalize, hi, lo, norm, colsum_part); break;
        default: hipLaunchKernelGGL(nr_prepare_kernel<4>, dim3(grid), dim3(256), 0, st, x, mask, n_tok, d, normalize, hi, lo, norm, colsum_part); break;
    }
    NR_LAUNCH_CHECK();
    return NR_OK;
}

extern "C" int nr_prepare_tokens_pair(const float* x0, const float* mask0, int n_tok0, uint16_t* hi0, uint16_t* lo0, float* norm0,
                                      float* colsum_part0, const float* x1, const float* mask1, int n_tok1, uint16_t* hi1,
                                      uint16_t* lo1, float* norm1, float* colsum_part1, int d, int normalize, void* stream) {
    if (!x0 || !hi0 || !x1 || !hi1 || n_tok0 <= 0 || n_tok1 <= 0 || d <= 0) return NR_EINVAL;
    if ((d % 256) != 0 || d / 256 > NR_PREP_MAX_CHUNKS) return NR_EUNSUPPORTED;
    auto blocks = [](int n_tok, const float* cs) {
        if (cs) return nr_prepare_parts(n_tok);
        int g = (n_tok + 15) / 16;
        return g > NR_PREP_MAX_GRID ? NR_PREP_MAX_GRID : g;
    };
    const int nb0 = blocks(n_tok0, colsum_part0), nb1 = blocks(n_tok1, colsum_part1);
    NrPrepareOne a{x0, mask0, n_tok0, hi0, lo0, norm0, colsum_part0}, b{x1, mask1, n_tok1, hi1, lo1, norm1, colsum_part1};
    hipStream_t st = (hipStream_t)stream;
    switch (d / 256) {
        case 1: hipLaunchKernelGGL(nr_prepare_pair_kernel<1>, dim3(nb0 + nb1), dim3(256), 0, st, a, b, d, normalize, nb0); break;
        case 2: hipLaunchKernelGGL(nr_prepare_pair_kernel<2>, dim3(nb0 + nb1), dim3(256), 0, st, a, b, d, normalize, nb0); break;
        case 3: hipLaunchKernelGGL(nr_prepare_pair_kernel<3>, dim3(nb0 + nb1), dim3(256), 0, st, a, b, d, normalize, nb0); break;
        default: hipLaunchKernelGGL(nr_prepare_pair_kernel<4>, dim3(nb0 + nb1), dim3(256), 0, st, a, b, d, normalize, nb0); break;
    }
    NR_LAUNCH_CHECK();
    return NR_OK;
}

__global__ __launch_bounds__(256) void nr_split_kernel(const float* __restrict__ x, size_t n, uint16_t* __restrict__ hi,
                                                       uint16_t* __restrict__ lo) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float v = x[i];
        uint16_t h = nr_f2bf(v);
        hi[i] = h;
        if (lo) lo[i] = nr_f2bf(v - nr_bf2f(h));
    }
}

extern "C" int nr_split_bf16(const float* x, size_t n, uint16_t* hi, uint16_t* lo, void* stream) {
    if (!x || !hi || n == 0) return NR_EINVAL;
    size_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(nr_split_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, n, hi, lo);
    NR_LAUNCH_CHECK();
    return NR_OK;
}
