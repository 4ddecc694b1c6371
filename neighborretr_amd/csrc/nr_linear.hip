// Y[M,N] = X[M,K] * W[N,K]^T (+ bias[N]) (+ residual[M,N]) in split-bf16 on the MFMA tile engine:
// the fp32 GEMMs of the token-clustering stage (k=3 token convolution as a [B*N, 3C] x [3C, C] product,
// cluster.py:664; the kv projection, cluster.py:866) at ~fp32 accuracy (operands carried as bf16
// hi + lo, products Ah*Bh + Ah*Bl + Al*Bh, fp32 accumulate) and bf16-pipe speed.
// X arrives already split (nr_shift_concat_split / nr_ctm_front write hi/lo directly), W is split once
// per parameter version by the host.
#include "nr_gemm_tile.h"
#include "../../include/nr_hip.h"

struct NrLinearArgs {
    const uint16_t *x_hi, *x_lo, *w_hi, *w_lo;
    const float *bias, *residual;
    float* out;
    int M, N, K;
};

template <int MI, int NI, int STAGES>
__global__ __launch_bounds__(256) void nr_linear_kernel(NrLinearArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using Tile = NrGemmTile<MI, NI, true, 16, 16, STAGES>;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int row0 = blockIdx.y * Tile::BM, col0 = blockIdx.x * Tile::BN;
    Tile tile;
    tile.zero();
    tile.run(p.x_hi, p.x_lo, row0, p.M, p.w_hi, p.w_lo, col0, p.N, p.K, smem);
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
            }
    }
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

// x[n-1] | x[n] | x[n+1] written directly as bf16 hi / lo (operand of the conv GEMM)
__global__ __launch_bounds__(256) void nr_shift_concat_split_kernel(const float* __restrict__ x, int N, int C,
                                                                    uint16_t* __restrict__ hi, uint16_t* __restrict__ lo) {
    const int row = blockIdx.x;                 // b*N + n
    const int n = row % N;
    for (int k = 0; k < 3; ++k) {
        const int nn = n + k - 1;
        const bool ok = nn >= 0 && nn < N;
        const float* src = x + (size_t)(row + k - 1) * C;
        const size_t o = (size_t)row * 3 * C + (size_t)k * C;
        for (int c = threadIdx.x * 4; c < C; c += 1024) {
            f32x4_t v = {0.f, 0.f, 0.f, 0.f};
            if (ok) v = *reinterpret_cast<const f32x4_t*>(src + c);
            uint16_t h[4], l[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                h[e] = nr_f2bf(v[e]);
                l[e] = nr_f2bf(v[e] - nr_bf2f(h[e]));
            }
            *reinterpret_cast<uint2*>(hi + o + c) = make_uint2((uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16));
            *reinterpret_cast<uint2*>(lo + o + c) = make_uint2((uint32_t)l[0] | ((uint32_t)l[1] << 16), (uint32_t)l[2] | ((uint32_t)l[3] << 16));
        }
    }
}

extern "C" int nr_shift_concat_split(const float* x, int n_samples, int N, int C, uint16_t* hi, uint16_t* lo, void* stream) {
    if (!x || !hi || !lo || n_samples <= 0 || N <= 0 || C <= 0 || (C % 4) != 0) return NR_EINVAL;
    hipLaunchKernelGGL(nr_shift_concat_split_kernel, dim3(n_samples * N), dim3(256), 0, (hipStream_t)stream, x, N, C, hi, lo);
    NR_LAUNCH_CHECK();
    return NR_OK;
}
