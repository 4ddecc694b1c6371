// MFMA tile engine shared by the token-token similarity kernel and the token-weight MLP kernel.
//
// C[BM x BN] = A[BM x K] * B[BN x K]^T with both operands row-major bf16 ("NT" product), fp32
// accumulate on v_mfma_f32_16x16x32_bf16.  2 x WC waves (256 threads at WC = 2); each wave owns
// MI x NI sub-tiles of 16 x 16 (BM = 32*MI, BN = 16*WC*NI).  K is walked in BK = 64 slices through a
// STAGES-deep LDS ring (default 3) filled by LDS-DMA (global_load_lds_dwordx4: HBM/L2 -> LDS with no VGPR staging and
// no ds_write pass -- the store side of the LDS, ~79 B/clk/CU, was the bottleneck of the register-
// staged version).  One DMA instruction writes 1 KiB of LDS linearly (wave base + lane*16), i.e.
// 8 rows x 128 B; the bank-conflict swizzle (16-B chunk ^= row&7, conflict-free ds_read_b128 of
// the fragments) is applied on the per-lane SOURCE address, the LDS image stays linear per wave.
// Later slices are in flight while slice kt is multiplied, one raw s_barrier per slice, counted
// s_waitcnt vmcnt (a __syncthreads() would drain the DMA).
//
// X3 = split-bf16 mode: every operand is carried as hi + lo (two bf16 arrays) and the product is
// accumulated as Ah*Bh + Ah*Bl + Al*Bh (the ~2^-18 lo*lo term is dropped), giving ~fp32-grade
// products on the bf16 MFMA pipe for the rank-exact / golden-parity paths.
#pragma once
#include "nr_common.h"

#include <type_traits>
#include <utility>

// compile-time loop: f(std::integral_constant<int, I>) for I in [B, E)
template <int B, int E, typename F>
__device__ __forceinline__ void nr_static_for(F&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        nr_static_for<B + 1, E>(std::forward<F>(f));
    }
}

#ifdef NR_STAMP
// In-kernel timing (diagnostic builds only: NR_EXTRA_FLAGS=-DNR_STAMP): every wave keeps cycle sums of the K
// loop's phases in registers (a global store per stamp would itself sit in vmcnt and be waited for by the
// loop's own s_waitcnt); wave 0 of workgroup 0 writes them out once at the end.
static __device__ unsigned long long nr_stamp_buf[256];      // one copy per translation unit
struct NrPhaseClock {
    unsigned long long t, wait_dma, barrier, compute, slices;
    __device__ __forceinline__ void start() { t = __builtin_readcyclecounter(); wait_dma = barrier = compute = slices = 0; }
    __device__ __forceinline__ unsigned long long lap() {
        unsigned long long n = __builtin_readcyclecounter(), d = n - t;
        t = n;
        return d;
    }
    __device__ __forceinline__ void flush(int base) {
        if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
            nr_stamp_buf[base + 0] = wait_dma; nr_stamp_buf[base + 1] = barrier;
            nr_stamp_buf[base + 2] = compute;  nr_stamp_buf[base + 3] = slices;
        }
    }
};
#endif

typedef __attribute__((address_space(3))) void* nr_lds_ptr_t;
typedef const __attribute__((address_space(1))) void* nr_glb_ptr_t;

// Ring depth is the caller's choice.  Measured on MI355X: with >= ~2 workgroups per CU in the grid a
// SINGLE stage wins (smallest LDS footprint -> more resident workgroups, which hide the DMA latency of
// each other: split-bf16 bank product 79 -> 52 us, bank scorer 27 -> 23 us); small grids (one workgroup
// per CU) need the 2-deep ring to overlap their own loads (batch scorer 19.5 vs 25 us); 3 stages lose to 2
// everywhere (26.5 -> 29.9 us).
constexpr int nr_pick_stages(long n_workgroups) { return n_workgroups >= 512 ? 1 : 2; }

// WC = waves along the columns: the workgroup is 2 x WC waves (256 threads at WC = 2, 512 at WC = 4).
template <int MI, int NI, bool X3, int TPS_A = 16, int TPS_B = 16, int STAGES = 2, int WC = 2>
struct NrGemmTile {
    static constexpr int NW = 2 * WC;                  // waves per workgroup
    static constexpr int BM = 32 * MI;
    static constexpr int BN = 16 * WC * NI;
    static constexpr int BK = 64;
    static constexpr int A_BYTES = BM * BK * 2;
    static constexpr int B_BYTES = BN * BK * 2;
    static constexpr int STAGE_BYTES = (A_BYTES + B_BYTES) * (X3 ? 2 : 1);
    static constexpr int RING_BYTES = STAGES * STAGE_BYTES;
    static constexpr int PA = (BM / 8) / NW;           // 8-row DMA pieces per wave, A and B
    static constexpr int PB = (BN / 8) / NW;
    static_assert((BM / 8) % NW == 0 && (BN / 8) % NW == 0, "tile rows must split evenly over the waves");
    static constexpr int DMA_PER_STAGE = (PA + PB) * (X3 ? 2 : 1);   // per wave

    f32x4_t acc[MI][NI];

    // Fragment row `f` (0..15) of sub-tile `m` of a wave strip -> row of the strip in memory order.
    // TPS = 16: identity.  TPS = 8 / 4: the strip holds 16/TPS samples of MI*TPS tokens each and a
    // sub-tile takes TPS consecutive tokens of every sample, so that (sample, token) splits cleanly
    // over (lane group, register) in the accumulator layout -- see nr_sim_reg.hip.
    template <int TPS, int XI>
    __device__ static __forceinline__ int strip_row(int m, int f) {
        if constexpr (TPS == 16) return m * 16 + f;
        else return (f / TPS) * (TPS * XI) + m * TPS + (f % TPS);
    }

    // Swizzle key of tile row r (0..7): the row's index INSIDE the 16-row MFMA fragment that reads it, mod 8.
    // A fragment ds_read_b128 touches 16 rows x one 16-byte chunk; row r sits in LDS at 128*r, i.e. on the
    // half of the 256-byte bank row selected by r&1, and chunk kc is stored at slot kc ^ key.  With
    // key = fragment index & 7 (and TPS even, so r&1 == fragment index & 1) the 16 lanes of every lane group
    // land on 16 different 16-byte slots for ANY token permutation strip_row<TPS,XI> -- the same pattern as
    // 16 consecutive rows.  (A key of r&7 is only conflict-free for TPS = 8 / 16 and 4-frame strips; the
    // 192 x 192 blocks, TPS = 4 / FPS = 2, read with 4-way conflicts under it.)
    template <int TPS, int XI>
    __device__ static __forceinline__ int row_key(int r) {
        if constexpr (TPS == 16) return r & 7;
        else {
            const int rl = r % (16 * XI);               // row inside the wave strip
            const int a = rl / (TPS * XI), b = rl % TPS;
            return (a * TPS + b) & 7;
        }
    }
    // byte offset of 16-byte chunk `kc` (0..7) of row `r` (swizzle key `key`) inside a [rows][64] bf16 LDS image
    __device__ static __forceinline__ int lds_off(int r, int kc, int key) { return r * 128 + ((kc ^ key) << 4); }

    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int m = 0; m < MI; ++m)
#pragma unroll
            for (int n = 0; n < NI; ++n) acc[m][n] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }

    // Runs the whole K loop.  a_hi/a_lo: [a_rows, K]; rows [a_row0, a_row0+BM) are used, clamped to
    // a_rows-1 (the caller discards results of clamped rows).  Same for B.  K % 64 == 0.
    // smem: RING_BYTES of dynamic LDS (the ONLY __shared__ object of the kernel).  On return all
    // waves have passed the final barrier: smem may be reused.
    __device__ __forceinline__ void run(const uint16_t* __restrict__ a_hi, const uint16_t* __restrict__ a_lo,
                                        int a_row0, int a_rows,
                                        const uint16_t* __restrict__ b_hi, const uint16_t* __restrict__ b_lo,
                                        int b_row0, int b_rows, int K, char* smem) {
        const int tid = threadIdx.x;
        const int lane = tid & 63;
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int wr = wave / WC, wc = wave % WC;

        // DMA piece i of an operand = LDS rows [8i, 8i+8); waves take pieces i = wave, wave+NW, ...
        // lane -> (row 8i + lane/8, LDS slot lane%8) which must receive global chunk slot ^ key(row)
        const char* ga_h[PA];
        const char* gb_h[PB];
        const char* ga_l[X3 ? PA : 1];
        const char* gb_l[X3 ? PB : 1];
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            int r = (wave + NW * i) * 8 + (lane >> 3);
            int kc = (lane & 7) ^ row_key<TPS_A, MI>(r);
            int gr = min(a_row0 + r, a_rows - 1);
            ga_h[i] = reinterpret_cast<const char*>(a_hi + (size_t)gr * K + kc * 8);
            if constexpr (X3) ga_l[i] = reinterpret_cast<const char*>(a_lo + (size_t)gr * K + kc * 8);
        }
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            int r = (wave + NW * i) * 8 + (lane >> 3);
            int kc = (lane & 7) ^ row_key<TPS_B, NI>(r);
            int gr = min(b_row0 + r, b_rows - 1);
            gb_h[i] = reinterpret_cast<const char*>(b_hi + (size_t)gr * K + kc * 8);
            if constexpr (X3) gb_l[i] = reinterpret_cast<const char*>(b_lo + (size_t)gr * K + kc * 8);
        }
        // DMA instruction `idx` (0 .. DMA_PER_STAGE-1) of slice kt: A pieces first (hi, then lo), then B pieces
        auto issue_one = [&](int kt, auto idx_c) {
            constexpr int idx = decltype(idx_c)::value;
            constexpr int per = X3 ? 2 : 1;
            constexpr bool is_a = idx < PA * per;
            constexpr int j = is_a ? idx : idx - PA * per;
            constexpr int piece = j / per;
            constexpr bool lo = (j % per) == 1;
            char* st = smem + (kt % STAGES) * STAGE_BYTES;
            const int kb = kt * BK * 2;                 // byte offset along K
            char* dst = st + (is_a ? 0 : A_BYTES) + (wave + NW * piece) * 1024 + (lo ? A_BYTES + B_BYTES : 0);
            const char* src;
            if constexpr (is_a) src = lo ? ga_l[X3 ? piece : 0] : ga_h[piece];
            else src = lo ? gb_l[X3 ? piece : 0] : gb_h[piece];
            __builtin_amdgcn_global_load_lds((nr_glb_ptr_t)(src + kb), (nr_lds_ptr_t)dst, 16, 0, 0);
        };
        auto issue = [&](int kt) { nr_static_for<0, DMA_PER_STAGE>([&](auto i) { issue_one(kt, i); }); };

        // fragment addresses (lane-constant): row within the wave's strip, k-chunk lane>>4
        const int frow = lane & 15, fq = lane >> 4;
        const int KT = K / BK;
        // STAGES-deep ring, ONE barrier per slice: slices kt+1 .. kt+STAGES-2 are in flight while slice kt is
        // multiplied.  The barrier at the top of iteration kt (a) publishes slice kt (every wave waited for
        // its own share first) and (b) proves every wave has finished reading slice kt-1, whose stage is the
        // one refilled right after it.
#pragma unroll
        for (int s = 0; s < STAGES - 1; ++s)
            if (s < KT) issue(s);
#ifdef NR_STAMP
        NrPhaseClock clk;
        clk.start();
#endif
        for (int kt = 0; kt < KT; ++kt) {
#ifdef NR_STAMP
            clk.compute += clk.lap();
            clk.slices += 1;
#endif
            if constexpr (STAGES == 1) {
                // single stage: refill after everyone is done with the previous slice, no prefetch; the
                // other resident workgroups of the CU cover the DMA latency (smallest LDS footprint)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                issue(kt);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            } else {
                // this wave's share of slice kt has landed once at most min(STAGES-2, KT-1-kt) younger groups are pending
                const int younger = min(STAGES - 2, KT - 1 - kt);
                if (younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * DMA_PER_STAGE) : "memory");
                else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA_PER_STAGE) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // fragment reads of slice kt-1 are complete
#ifdef NR_STAMP
                clk.wait_dma += clk.lap();
#endif
                __builtin_amdgcn_s_barrier();
#ifdef NR_STAMP
                clk.barrier += clk.lap();
#endif
            }
            // The DMA instructions of the slice to prefetch are issued BETWEEN this slice's MFMAs, a few MFMAs
            // apart (an LDS-DMA instruction holds its wave's issue for ~60-180 cycles: in one burst ahead of the
            // MFMAs that is ~1k cycles per slice in which this wave feeds the matrix pipe nothing).
            const bool prefetch = STAGES > 1 && kt + STAGES - 1 < KT;
            const int kt_pf = kt + STAGES - 1;
            const char* sAh = smem + (kt % STAGES) * STAGE_BYTES;
            const char* sBh = sAh + A_BYTES;
            const char* sAl = sAh + A_BYTES + B_BYTES;
            const char* sBl = sAl + A_BYTES;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8_t fa_h[MI], fb_h[NI], fa_l[X3 ? MI : 1], fb_l[X3 ? NI : 1];
#pragma unroll
                for (int m = 0; m < MI; ++m) {
                    int r = wr * 16 * MI + strip_row<TPS_A, MI>(m, frow);
                    int off = lds_off(r, ks * 4 + fq, frow & 7);
                    fa_h[m] = *reinterpret_cast<const bf16x8_t*>(sAh + off);
                    if constexpr (X3) fa_l[m] = *reinterpret_cast<const bf16x8_t*>(sAl + off);
                }
#pragma unroll
                for (int n = 0; n < NI; ++n) {
                    int r = wc * 16 * NI + strip_row<TPS_B, NI>(n, frow);
                    int off = lds_off(r, ks * 4 + fq, frow & 7);
                    fb_h[n] = *reinterpret_cast<const bf16x8_t*>(sBh + off);
                    if constexpr (X3) fb_l[n] = *reinterpret_cast<const bf16x8_t*>(sBl + off);
                }
                nr_static_for<0, MI * NI>([&](auto g_c) {
                    constexpr int g = decltype(g_c)::value;
                    constexpr int m = g / NI, n = g % NI;
                    if constexpr (X3) {
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa_l[m], fb_h[n], acc[m][n], 0, 0, 0);
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa_h[m], fb_l[n], acc[m][n], 0, 0, 0);
                    }
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa_h[m], fb_h[n], acc[m][n], 0, 0, 0);
                    if constexpr (STAGES > 1) {
                        // the DMA instructions go out between the MFMA groups of the FIRST HALF of the slice: apart
                        // enough not to stall the matrix pipe, early enough to have landed when the slice ends
                        // (spread over the whole slice, the last pieces kept every wave waiting ~640 cycles at
                        // the next slice's vmcnt -- in-kernel stamps, 192 x 384 blocks)
                        constexpr int G = (2 * MI * NI + 1) / 2;
                        nr_static_for<0, DMA_PER_STAGE>([&](auto i_c) {
                            constexpr int i = decltype(i_c)::value;
                            // instruction i belongs to group floor(i * G / DMA_PER_STAGE)
                            constexpr int home = (i * G) / DMA_PER_STAGE;
                            if (prefetch && ks * MI * NI + g == home) issue_one(kt_pf, i_c);
                        });
                    }
                });
            }
        }
#ifdef NR_STAMP
        clk.compute += clk.lap();
        clk.flush(0);
#endif
        // every wave is done with the ring before the caller reuses the LDS
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }

    // C element (m, n, j) of this lane sits at tile row/col:
    //   row = wr*16*MI + m*16 + (lane>>4)*4 + j,   col = wc*16*NI + n*16 + (lane&15)
    __device__ __forceinline__ void store_lds(float* sC, int ldc) const {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int wr = wave / WC, wc = wave % WC;
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
