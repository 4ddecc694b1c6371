// MFMA tile engine shared by the token-token similarity kernel and the token-weight MLP kernel.
//
// C[BM x BN] = A[BM x K] * B[BN x K]^T with both operands row-major bf16 ("NT" product), fp32
// accumulate on v_mfma_f32_16x16x32_bf16.  256 threads = 4 waves laid out 2 x 2; each wave owns
// MI x NI sub-tiles of 16 x 16 (BM = 32*MI, BN = 32*NI).  K is walked in BK = 64 slices:
// global -> registers (16-byte loads, issued one slice ahead so HBM/L2 latency hides under the
// MFMAs of the current slice) -> LDS (XOR-swizzled 128-byte rows, conflict-free ds_read_b128)
// -> fragments.
//
// X3 = split-bf16 mode: every operand is carried as hi + lo (two bf16 arrays) and the product is
// accumulated as Ah*Bh + Ah*Bl + Al*Bh (the ~2^-18 lo*lo term is dropped), giving ~fp32-grade
// products on the bf16 MFMA pipe for the rank-exact / golden-parity paths.
#pragma once
#include "nr_common.h"

template <int MI, int NI, bool X3>
struct NrGemmTile {
    static constexpr int BM = 32 * MI;
    static constexpr int BN = 32 * NI;
    static constexpr int BK = 64;
    static constexpr int A_BYTES = BM * BK * 2;
    static constexpr int B_BYTES = BN * BK * 2;
    static constexpr int STAGE_BYTES = (A_BYTES + B_BYTES) * (X3 ? 2 : 1);

    f32x4_t acc[MI][NI];

    // byte offset of 16-byte chunk `kc` (0..7) of row `r` inside a [rows][64] bf16 LDS image
    __device__ static __forceinline__ int lds_off(int r, int kc) { return r * 128 + ((kc ^ (r & 7)) << 4); }

    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int m = 0; m < MI; ++m)
#pragma unroll
            for (int n = 0; n < NI; ++n) acc[m][n] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }

    // Runs the whole K loop.  a_hi/a_lo: [a_rows, K]; rows [a_row0, a_row0+BM) are used, clamped to
    // a_rows-1 (the caller discards results of clamped rows).  Same for B.  K % 64 == 0.
    __device__ __forceinline__ void run(const uint16_t* __restrict__ a_hi, const uint16_t* __restrict__ a_lo,
                                        int a_row0, int a_rows,
                                        const uint16_t* __restrict__ b_hi, const uint16_t* __restrict__ b_lo,
                                        int b_row0, int b_rows, int K, char* smem) {
        const int tid = threadIdx.x;
        const int lane = tid & 63;
        const int wave = tid >> 6;
        const int wr = wave >> 1, wc = wave & 1;

        char* sAh = smem;
        char* sBh = smem + A_BYTES;
        char* sAl = smem + A_BYTES + B_BYTES;
        char* sBl = sAl + A_BYTES;

        // per-thread staging slots: chunk c = tid + i*256 -> row c>>3, k-chunk c&7
        u32x4_t ra_h[MI], rb_h[NI], ra_l[X3 ? MI : 1], rb_l[X3 ? NI : 1];
        const uint16_t* pa_h[MI];
        const uint16_t* pb_h[NI];
        const uint16_t* pa_l[X3 ? MI : 1];
        const uint16_t* pb_l[X3 ? NI : 1];
        int oa[MI], ob[NI];
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            int c = tid + i * 256, r = c >> 3, kc = c & 7;
            int gr = min(a_row0 + r, a_rows - 1);
            pa_h[i] = a_hi + (size_t)gr * K + kc * 8;
            if constexpr (X3) pa_l[i] = a_lo + (size_t)gr * K + kc * 8;
            oa[i] = lds_off(r, kc);
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            int c = tid + i * 256, r = c >> 3, kc = c & 7;
            int gr = min(b_row0 + r, b_rows - 1);
            pb_h[i] = b_hi + (size_t)gr * K + kc * 8;
            if constexpr (X3) pb_l[i] = b_lo + (size_t)gr * K + kc * 8;
            ob[i] = lds_off(r, kc);
        }
        auto gload = [&](int k0) {
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                ra_h[i] = *reinterpret_cast<const u32x4_t*>(pa_h[i] + k0);
                if constexpr (X3) ra_l[i] = *reinterpret_cast<const u32x4_t*>(pa_l[i] + k0);
            }
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                rb_h[i] = *reinterpret_cast<const u32x4_t*>(pb_h[i] + k0);
                if constexpr (X3) rb_l[i] = *reinterpret_cast<const u32x4_t*>(pb_l[i] + k0);
            }
        };

        // fragment addresses (lane-constant): row within the wave's strip, k-chunk lane>>4
        const int frow = lane & 15, fq = lane >> 4;

        gload(0);
        const int KT = K / BK;
        for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                *reinterpret_cast<u32x4_t*>(sAh + oa[i]) = ra_h[i];
                if constexpr (X3) *reinterpret_cast<u32x4_t*>(sAl + oa[i]) = ra_l[i];
            }
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                *reinterpret_cast<u32x4_t*>(sBh + ob[i]) = rb_h[i];
                if constexpr (X3) *reinterpret_cast<u32x4_t*>(sBl + ob[i]) = rb_l[i];
            }
            __syncthreads();
            if (kt + 1 < KT) gload((kt + 1) * BK);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8_t fa_h[MI], fb_h[NI], fa_l[X3 ? MI : 1], fb_l[X3 ? NI : 1];
#pragma unroll
                for (int m = 0; m < MI; ++m) {
                    int r = wr * 16 * MI + m * 16 + frow;
                    int off = lds_off(r, ks * 4 + fq);
                    fa_h[m] = *reinterpret_cast<const bf16x8_t*>(sAh + off);
                    if constexpr (X3) fa_l[m] = *reinterpret_cast<const bf16x8_t*>(sAl + off);
                }
#pragma unroll
                for (int n = 0; n < NI; ++n) {
                    int r = wc * 16 * NI + n * 16 + frow;
                    int off = lds_off(r, ks * 4 + fq);
                    fb_h[n] = *reinterpret_cast<const bf16x8_t*>(sBh + off);
                    if constexpr (X3) fb_l[n] = *reinterpret_cast<const bf16x8_t*>(sBl + off);
                }
#pragma unroll
                for (int m = 0; m < MI; ++m)
#pragma unroll
                    for (int n = 0; n < NI; ++n) {
                        if constexpr (X3) {
                            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa_l[m], fb_h[n], acc[m][n], 0, 0, 0);
                            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa_h[m], fb_l[n], acc[m][n], 0, 0, 0);
                        }
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa_h[m], fb_h[n], acc[m][n], 0, 0, 0);
                    }
            }
            __syncthreads();
        }
    }

    // C element (m, n, j) of this lane sits at tile row/col:
    //   row = wr*16*MI + m*16 + (lane>>4)*4 + j,   col = wc*16*NI + n*16 + (lane&15)
    __device__ __forceinline__ void store_lds(float* sC, int ldc) const {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int wr = wave >> 1, wc = wave & 1;
#pragma unroll
        for (int m = 0; m < MI; ++m)
#pragma unroll
            for (int n = 0; n < NI; ++n)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    int r = wr * 16 * MI + m * 16 + (lane >> 4) * 4 + j;
                    int c = wc * 16 * NI + n * 16 + (lane & 15);
                    sC[r * ldc + c] = acc[m][n][j];
                }
    }
};
