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

// a row of zeros for operand rows that do not exist (the k=3 token convolution read in place: run_conv3)
static __device__ uint16_t nr_zero_row[1024];

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
                                        int b_row0, int b_rows, int K, char* smem, int rot = 0, bool dma_front = false, int ld = 0) {
        run_impl<false>(a_hi, a_lo, a_row0, a_rows, b_hi, b_lo, b_row0, b_rows, K, smem, rot, dma_front, 0, ld);
    }

    // The k=3 token convolution read IN PLACE (cluster.py:664): A is the token matrix itself, [a_rows, K/3] (hi / lo), and
    // the product is  sum_s A[row + s - 1, :] . B[:, s K/3 : (s+1) K/3]  with rows outside the sample of `conv_n` tokens
    // reading as zeros -- the K loop walks three segments, each with its own (shifted, or zero-row) source pointers.
    // No [rows, 3 K/3] shifted copy of the tokens exists any more: a third of the A bytes, one launch less.
    __device__ __forceinline__ void run_conv3(const uint16_t* __restrict__ a_hi, const uint16_t* __restrict__ a_lo,
                                              int a_row0, int a_rows,
                                              const uint16_t* __restrict__ b_hi, const uint16_t* __restrict__ b_lo,
                                              int b_row0, int b_rows, int K, char* smem, int conv_n) {
        run_impl<true>(a_hi, a_lo, a_row0, a_rows, b_hi, b_lo, b_row0, b_rows, K, smem, 0, false, conv_n);
    }

    template <bool CONV3>
    __device__ __forceinline__ void run_impl(const uint16_t* __restrict__ a_hi, const uint16_t* __restrict__ a_lo,
                                             int a_row0, int a_rows,
                                             const uint16_t* __restrict__ b_hi, const uint16_t* __restrict__ b_lo,
                                             int b_row0, int b_rows, int K, char* smem, int rot, bool dma_front, int conv_n,
                                             int ld = 0) {
        static_assert((BM / 8) % NW == 0 && (BN / 8) % NW == 0, "tile rows must split evenly over the waves");
        // ld: row pitch of BOTH operands in elements when they are K-slices of wider matrices (0: the rows are K long)
        const int LD = ld > 0 ? ld : K;
        const int tid = threadIdx.x;
        const int lane = tid & 63;
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int wr = wave / WC, wc = wave % WC;

        // DMA piece i of an operand = LDS rows [8i, 8i+8); waves take pieces i = wave, wave+NW, ...
        // lane -> (row 8i + lane/8, LDS slot lane%8) which must receive global chunk slot ^ key(row)
        constexpr int NSEG_A = CONV3 ? 3 : 1;                // source pointer sets of operand A
        const char* ga_h[NSEG_A][PA];
        const char* gb_h[PB];
        const char* ga_l[NSEG_A][X3 ? PA : 1];
        const char* gb_l[X3 ? PB : 1];
        const int CA = CONV3 ? K / 3 : LD;                   // row pitch of operand A
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            int r = (wave + NW * i) * 8 + (lane >> 3);
            int kc = (lane & 7) ^ row_key<TPS_A, MI>(r);
            if constexpr (CONV3) {
                const int gr = a_row0 + r, t = gr % conv_n;
#pragma unroll
                for (int sg = 0; sg < 3; ++sg) {
                    const bool on = gr < a_rows && t + sg - 1 >= 0 && t + sg - 1 < conv_n;
                    const size_t o = (size_t)(gr + sg - 1) * CA + kc * 8;
                    ga_h[sg][i] = reinterpret_cast<const char*>(on ? a_hi + o : nr_zero_row + kc * 8);
                    if constexpr (X3) ga_l[sg][i] = reinterpret_cast<const char*>(on ? a_lo + o : nr_zero_row + kc * 8);
                }
            } else {
                int gr = min(a_row0 + r, a_rows - 1);
                ga_h[0][i] = reinterpret_cast<const char*>(a_hi + (size_t)gr * CA + kc * 8);
                if constexpr (X3) ga_l[0][i] = reinterpret_cast<const char*>(a_lo + (size_t)gr * CA + kc * 8);
            }
        }
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            int r = (wave + NW * i) * 8 + (lane >> 3);
            int kc = (lane & 7) ^ row_key<TPS_B, NI>(r);
            int gr = min(b_row0 + r, b_rows - 1);
            gb_h[i] = reinterpret_cast<const char*>(b_hi + (size_t)gr * LD + kc * 8);
            if constexpr (X3) gb_l[i] = reinterpret_cast<const char*>(b_lo + (size_t)gr * LD + kc * 8);
        }
        const int KT_ = K / BK;
        // DMA instruction `idx` (0 .. DMA_PER_STAGE-1) of slice kt: A pieces first (hi, then lo), then B pieces
        auto issue_one = [&](int kt, auto idx_c) {
            constexpr int idx = decltype(idx_c)::value;
            constexpr int per = X3 ? 2 : 1;
            constexpr bool is_a = idx < PA * per;
            constexpr int j = is_a ? idx : idx - PA * per;
            constexpr int piece = j / per;
            constexpr bool lo = (j % per) == 1;
            char* st = smem + (kt % STAGES) * STAGE_BYTES;
            // byte offset along K.  `rot` rotates the order in which this workgroup walks the K slices (slice kt of
            // the loop is slice (kt + rot) mod KT of the operands): workgroups that share operand rows through one L2
            // then first-touch DIFFERENT slices and find the others' already there (see nr_sim_reg.hip)
            const int kr = kt + rot;
            int kb = (kr >= KT_ ? kr - KT_ : kr) * BK * 2;
            char* dst = st + (is_a ? 0 : A_BYTES) + (wave + NW * piece) * 1024 + (lo ? A_BYTES + B_BYTES : 0);
            const char* src;
            if constexpr (is_a) {
                if constexpr (CONV3) {
                    const int ks_ = KT_ / 3, sg = kt >= 2 * ks_ ? 2 : (kt >= ks_ ? 1 : 0);      // wave-uniform
                    kb = (kt - sg * ks_) * BK * 2;
                    const char* p0 = lo ? ga_l[0][X3 ? piece : 0] : ga_h[0][piece];
                    const char* p1 = lo ? ga_l[1][X3 ? piece : 0] : ga_h[1][piece];
                    const char* p2 = lo ? ga_l[2][X3 ? piece : 0] : ga_h[2][piece];
                    src = sg == 0 ? p0 : (sg == 1 ? p1 : p2);
                } else {
                    src = lo ? ga_l[0][X3 ? piece : 0] : ga_h[0][piece];
                }
            } else {
                src = lo ? gb_l[X3 ? piece : 0] : gb_h[piece];
            }
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
                            // dma_front (A/B hook): the slice's pieces go out in two bursts right behind the fragment
                            // reads of each k-step instead of between the MFMAs
                            const bool here = dma_front ? (g == 0 && (2 * i) / DMA_PER_STAGE == ks) : (ks * MI * NI + g == home);
                            if (prefetch && here) issue_one(kt_pf, i_c);
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

    // ---- ping-pong K loop (8-wave workgroups: WC == 4, two-stage ring) ------------------------------------------------
    // The plain loop above lets both waves of a SIMD walk the same instruction mix in lockstep: they fetch their
    // fragments at the same time, queue their LDS-DMA instructions on the CU's one texture-address unit at the same
    // time (a piece is held ~100-185 cycles behind the other seven waves' pieces) and compete for the matrix pipe at
    // the same time; in-kernel stamps of the 192 x 384 block show the older wave of a SIMD finishing a slice in 2092
    // cycles and then waiting 1170 at the barrier for the younger one, 3302 cycles for 2304 cycles of MFMA work.
    // Here the two wave rows are two GROUPS (waves w and w+4 share a SIMD) that alternate roles phase by phase:
    //   phase:       A(kt)          B(kt)          C(kt)          D(kt)
    //   group 0:  MFMA k-step 0 |  mem          | MFMA k-step 1 |  mem
    //   group 1:  mem           |  MFMA k-step 0|  mem          |  MFMA k-step 1
    // "mem" = the group's fragment reads for its NEXT MFMA phase plus its share of the LDS-DMA of a later slice; the
    // matrix pipe always belongs to one group, whose MFMAs go out back to back from registers, while the other
    // group's memory instructions use the LDS and the address unit alone.  One s_barrier per phase.
    // Ring discipline (2 stages): slice kt+1 is issued in D(kt-1) and B(kt) by group 0 and in A(kt) by group 1, into
    // the stage slice kt-1 left (its last reader, group 1's k-step 1, ended with C(kt-1)); every wave waits for its
    // own pieces (vmcnt 0) before the barrier that opens D(kt), where group 0 starts reading slice kt+1.
    // LDS-DMA pieces (1 KiB = 8 rows x 128 B) per stage: NA_P of operand A, NB_P of operand B (BS per column strip).
    // The memory side of the loop -- ~72 KB of LDS-DMA (about 21 cycles of the CU's address unit per piece) plus
    // 192 KB of fragment reads per slice of the 192 x 384 block -- is as long as its MFMA side, so the pieces are dealt
    // EVENLY over the four phases of a slice:
    //   C(kt-1)  group 1, PC1 pieces per wave: B pieces of the wave's OWN column strip of slice kt+1 -- the stage is
    //            still being read in that phase, but that strip only by this wave (its group-0 twin read it in B), so
    //            the wave may overwrite it as soon as its own reads have returned;
    //   D(kt-1)  group 0, PD0 per wave;   A(kt)  group 1, PA1 per wave (own strip again);   B(kt)  group 0, PB0 per wave.
    // Group 0's waves own the A pieces and what is left of their column strip.
    static constexpr int NA_P = BM / 8, NB_P = BN / 8, BS = NB_P / 4;
    static constexpr int PQ = (NA_P + NB_P + 8) / 16;                  // ~ a quarter of the pieces, per wave
    static constexpr int PC1 = PQ < BS ? PQ : BS;
    static constexpr int PA1 = (BS - PC1) < PQ + 1 ? (BS - PC1) : PQ + 1;
    static constexpr int PB_0 = BS - PC1 - PA1;                         // B pieces left for the strip's group-0 wave
    static constexpr int P0 = NA_P / 4 + PB_0;                          // pieces per group-0 wave
    static constexpr int PB0 = (P0 + 1) / 2, PD0 = P0 - PB0;
    static_assert(NA_P % 4 == 0 && NB_P % 4 == 0, "piece counts must split over four waves");
    static_assert(P0 <= 21 && PC1 + PA1 <= 21, "chunk indices are packed 3 bits per slot into 64 bits");

    // One operand pair of the ping-pong loop.  Two segments CHAIN two tiles through one loop: the K slices of the second
    // tile follow the first tile's in the ring (its first two slices are requested while the first tile's last slice is
    // multiplied), `between(0)` -- the first tile's epilogue; the caller zeroes nothing: the loop does -- runs at the seam.
    struct Seg {
        const uint16_t *a_hi, *a_lo;
        int a_row0, a_rows;
        const uint16_t *b_hi, *b_lo;
        int b_row0, b_rows;
    };

    __device__ __forceinline__ void run_pp(const uint16_t* __restrict__ a_hi, const uint16_t* __restrict__ a_lo,
                                           int a_row0, int a_rows,
                                           const uint16_t* __restrict__ b_hi, const uint16_t* __restrict__ b_lo,
                                           int b_row0, int b_rows, int K, char* smem) {
        const Seg sg[1] = {{a_hi, a_lo, a_row0, a_rows, b_hi, b_lo, b_row0, b_rows}};
        run_pp_segs<1>(sg, K, smem, [](int) {});
    }

    // Split-bf16 product on a ONE-PASS tile (X3 = false: the ring holds hi halves only, so the block can be as large as the
    // one-pass blocks): three passes over K through one loop, accumulated -- Ah Bh over all slices, then Ah Bl, then Al Bh
    // (the three products the X3 tile adds up slice by slice; another summation order, the same ~16 mantissa bits).
    __device__ __forceinline__ void run_pp3(const uint16_t* __restrict__ a_hi, const uint16_t* __restrict__ a_lo,
                                            int a_row0, int a_rows,
                                            const uint16_t* __restrict__ b_hi, const uint16_t* __restrict__ b_lo,
                                            int b_row0, int b_rows, int K, char* smem) {
        static_assert(!X3, "three passes: on the one-pass tile");
        const Seg sg[1] = {{a_hi, a_lo, a_row0, a_rows, b_hi, b_lo, b_row0, b_rows}};
        run_pp_segs<1, true>(sg, K, smem, [](int) {});
    }

    template <int NSEG, bool ACC3 = false, typename Between>
    __device__ __forceinline__ void run_pp_segs(const Seg (&sg)[NSEG], int K, char* smem, Between&& between) {
        static_assert(WC == 4 && STAGES == 2, "ping-pong loop: 2 x 4 waves, two-stage ring");
        static_assert(NSEG == 1 || NSEG == 2, "one tile, or two chained tiles");
        static_assert(!ACC3 || (NSEG == 1 && !X3), "three accumulated passes: one tile, one-pass ring");
        const int tid = threadIdx.x;
        const int lane = tid & 63;
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int grp = wave / WC, wc = wave % WC;        // group = wave row
        const int KS = K / BK;                            // slices per tile
        const int KT = (ACC3 ? 3 : NSEG) * KS;            // slices of the whole loop

        // The pieces go out as buffer loads to LDS: per lane ONE row offset ((lane / 8) rows) and the piece's swizzled
        // 16-byte chunk (3 bits per slot, packed); operand, first row of the piece, K offset of the slice and LDS
        // destination are wave-uniform (SGPRs).  Rows past the operand's end read as zeros (buffer bounds): no clamping.
        constexpr int SEC = NSEG - 1;                     // index of the second segment (the first again when there is one)
        const __amdgpu_buffer_rsrc_t rs_ah = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(sg[0].a_hi), 0, sg[0].a_rows * K * 2, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_bh = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(sg[0].b_hi), 0, sg[0].b_rows * K * 2, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_al = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(X3 ? sg[0].a_lo : sg[0].a_hi), 0, sg[0].a_rows * K * 2, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_bl = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(X3 ? sg[0].b_lo : sg[0].b_hi), 0, sg[0].b_rows * K * 2, 0x00020000);
        const __amdgpu_buffer_rsrc_t rt_ah = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(sg[SEC].a_hi), 0, sg[SEC].a_rows * K * 2, 0x00020000);
        const __amdgpu_buffer_rsrc_t rt_bh = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(sg[SEC].b_hi), 0, sg[SEC].b_rows * K * 2, 0x00020000);
        const __amdgpu_buffer_rsrc_t rt_al = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(X3 ? sg[SEC].a_lo : sg[SEC].a_hi), 0, sg[SEC].a_rows * K * 2, 0x00020000);
        const __amdgpu_buffer_rsrc_t rt_bl = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(X3 ? sg[SEC].b_lo : sg[SEC].b_hi), 0, sg[SEC].b_rows * K * 2, 0x00020000);
        // (three passes: the lo halves as operands of their own, slices [KS, 2 KS) = Ah Bl, [2 KS, 3 KS) = Al Bh)
        const __amdgpu_buffer_rsrc_t r3_al = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(ACC3 ? sg[0].a_lo : sg[0].a_hi), 0, sg[0].a_rows * K * 2, 0x00020000);
        const __amdgpu_buffer_rsrc_t r3_bl = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(ACC3 ? sg[0].b_lo : sg[0].b_hi), 0, sg[0].b_rows * K * 2, 0x00020000);
        const int a_row0 = sg[0].a_row0, b_row0 = sg[0].b_row0, a_row1 = sg[SEC].a_row0, b_row1 = sg[SEC].b_row0;
        const int row_off = (lane >> 3) * K * 2;
        // slot s of a group-1 wave: B piece BS wc + s (s < PC1 + PA1); of a group-0 wave: s < NA_P/4: A piece
        // (NA_P/4) wc + s, else B piece BS wc + PC1 + PA1 + (s - NA_P/4)
        constexpr int SA0 = NA_P / 4;
        unsigned long long kc_pack = 0;                    // this wave's slots, 3 bits each
        {
            constexpr int SMAX = P0 > PC1 + PA1 ? P0 : PC1 + PA1;
#pragma unroll
            for (int s_ = 0; s_ < SMAX; ++s_) {
                const bool is_a = grp == 0 && s_ < SA0;
                const int piece = grp == 1 ? BS * wc + s_ : (is_a ? SA0 * wc + s_ : BS * wc + PC1 + PA1 + (s_ - SA0));
                const int r = piece * 8 + (lane >> 3);
                const int kc = (lane & 7) ^ (is_a ? row_key<TPS_A, MI>(r % BM) : row_key<TPS_B, NI>(r % BN));
                kc_pack |= (unsigned long long)kc << (3 * s_);
            }
        }
        // slots [S0, S1) of group G (compile time) for slice kt
        auto issue = [&](auto g_c, auto s0_c, auto s1_c, int kt) {
            constexpr int G = decltype(g_c)::value, S0 = decltype(s0_c)::value, S1 = decltype(s1_c)::value;
            char* st = smem + (kt & 1) * STAGE_BYTES;
            const bool second = NSEG > 1 && kt >= KS;         // slice of the second tile (wave-uniform)
            const int pass = ACC3 ? (kt >= 2 * KS ? 2 : (kt >= KS ? 1 : 0)) : 0;      // (wave-uniform)
            const int kb = (ACC3 ? kt - pass * KS : (second ? kt - KS : kt)) * BK * 2;
            const int ar0 = second ? a_row1 : a_row0, br0 = second ? b_row1 : b_row0;
            const __amdgpu_buffer_rsrc_t r_ah = ACC3 ? (pass == 2 ? r3_al : rs_ah) : (second ? rt_ah : rs_ah);
            const __amdgpu_buffer_rsrc_t r_bh = ACC3 ? (pass == 1 ? r3_bl : rs_bh) : (second ? rt_bh : rs_bh);
            const __amdgpu_buffer_rsrc_t r_al = second ? rt_al : rs_al, r_bl = second ? rt_bl : rs_bl;
            nr_static_for<S0, S1>([&](auto s_c) {
                constexpr int s_ = decltype(s_c)::value;
                constexpr bool is_a = G == 0 && s_ < SA0;
                const int piece = G == 1 ? BS * wc + s_ : (is_a ? SA0 * wc + s_ : BS * wc + PC1 + PA1 + (s_ - SA0));
                const int voff = row_off + (int)(((kc_pack >> (3 * s_)) & 7ull) << 4);
                const int so = ((is_a ? ar0 : br0) + piece * 8) * K * 2 + kb;
                char* d = st + (is_a ? 0 : A_BYTES) + piece * 1024;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(is_a ? r_ah : r_bh, (nr_lds_ptr_t)d, 16, voff, so, 0, 0);
                if constexpr (X3)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(is_a ? r_al : r_bl, (nr_lds_ptr_t)(d + A_BYTES + B_BYTES), 16, voff, so, 0, 0);
            });
        };
#ifndef NR_PP_MV
#define NR_PP_MV 0
#endif
        // pieces per wave moved from a memory phase into the neighbouring MFMA phase (at most what every phase can spare)
        constexpr int MV_CAP = (PB0 < PA1 ? (PB0 < PC1 ? PB0 : PC1) : (PA1 < PC1 ? PA1 : PC1)) - 1;
        constexpr int MV = NR_PP_MV < MV_CAP ? NR_PP_MV : MV_CAP;
        static_assert(MV >= 0 && MV < PB0 && MV < PA1 && MV < PC1, "not that many pieces in a memory phase");
        using IMV = std::integral_constant<int, MV>;
        using IA1 = std::integral_constant<int, PC1 + PA1 - MV>;
        using IC1 = std::integral_constant<int, PC1 - MV>;
        auto nothing = []() {};
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        using IPB0 = std::integral_constant<int, PB0>;
        using IP0 = std::integral_constant<int, P0>;
        using IPC1 = std::integral_constant<int, PC1>;
        using IP1 = std::integral_constant<int, PC1 + PA1>;

        const int frow = lane & 15, fq = lane >> 4;
        bf16x8_t fa_h[MI], fb_h[NI], fa_l[X3 ? MI : 1], fb_l[X3 ? NI : 1];
        auto load_frags = [&](int step) {
            const char* sAh = smem + ((step >> 1) & 1) * STAGE_BYTES;
            const char* sBh = sAh + A_BYTES;
            const char* sAl = sAh + A_BYTES + B_BYTES;
            const char* sBl = sAl + A_BYTES;
            const int ks = step & 1;
#pragma unroll
            for (int m = 0; m < MI; ++m) {
                int r = grp * 16 * MI + strip_row<TPS_A, MI>(m, frow);
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
        };
        auto frags_in = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };
#ifdef NR_STAMP
        unsigned long long pp_t = __builtin_readcyclecounter(), pp_work = 0, pp_bar = 0, pp_mma = 0;
#endif
        // `mid`: issued a third of the way through the phase's MFMAs (NR_PP_MV > 0: a few LDS-DMA pieces moved out of the
        // wave's memory phases into its MFMA phases, where the matrix pipe still has the queued MFMAs to chew on)
#ifndef NR_PP_PRIO
#define NR_PP_PRIO 1
#endif
        // NR_PP_PRIO (A/B hook): 1 = the MFMA phase runs at raised issue priority (round 2); 0 = no priority changes; 2 = the
        // MEMORY phase of the partner wave is the one raised (its few instructions get through between the MFMAs)
        auto mma = [&](auto&& mid) {
            if constexpr (NR_PP_PRIO == 1) __builtin_amdgcn_s_setprio(1);
            if constexpr (NR_PP_PRIO == 2) __builtin_amdgcn_s_setprio(0);
            nr_static_for<0, MI * NI>([&](auto g_c) {
                constexpr int g = decltype(g_c)::value;
                constexpr int m = g / NI, n = g % NI;
                if constexpr (X3) {
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa_l[m], fb_h[n], acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa_h[m], fb_l[n], acc[m][n], 0, 0, 0);
                }
                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa_h[m], fb_h[n], acc[m][n], 0, 0, 0);
                if constexpr (g == (MI * NI) / 3) mid();
            });
            if constexpr (NR_PP_PRIO == 1) __builtin_amdgcn_s_setprio(0);
            if constexpr (NR_PP_PRIO == 2) __builtin_amdgcn_s_setprio(1);
#ifdef NR_STAMP
            __builtin_amdgcn_sched_barrier(0);
            pp_mma += __builtin_readcyclecounter() - pp_t;
#endif
        };
        auto phase = [&]() {                 // phase boundary
            __builtin_amdgcn_sched_barrier(0);
#ifdef NR_STAMP
            { unsigned long long n = __builtin_readcyclecounter(); pp_work += n - pp_t; pp_t = n; }
#endif
            __builtin_amdgcn_s_barrier();
#ifdef NR_STAMP
            { unsigned long long n = __builtin_readcyclecounter(); pp_bar += n - pp_t; pp_t = n; }
#endif
            __builtin_amdgcn_sched_barrier(0);
        };
        auto landed = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
        // group 1 in phase C: everything but the PC1 pieces (x2 with split-bf16) it has just issued for slice kt+2
        auto landed_but = [&](bool newest_issued) {
            if (newest_issued) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PC1 - MV) * (X3 ? 2 : 1)) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        };

        // prologue: slice 0 by everyone; then the phases "C(-1)" / "D(-1)": first pieces of slice 1, group 0's fragments
        if (grp == 0) issue(I0{}, I0{}, IP0{}, 0);
        else issue(I1{}, I0{}, IP1{}, 0);
        // the accumulators are cleared HERE, under the first slice's DMA latency (MI*NI*4 v_mov per wave, two waves per SIMD:
        // ~1100 cycles of a ~32k-cycle launch when done in front of the first request) -- callers do not zero()
        zero();
        landed();
        phase();
        // The seam of two chained tiles (NSEG == 2) lies behind slice KS-1: each wave runs `between(0)` -- the first tile's
        // epilogue -- right after its own last MFMA of that slice and zeroes its accumulators; group 0 takes the fragments
        // of the second tile's first k-step only afterwards (they would not fit beside the epilogue's registers).  The
        // pieces of the second tile's first two slices went out in the regular places, phases earlier.
        if (grp == 0) {
            load_frags(0);
            if (KT > 1) issue(I0{}, IPB0{}, IP0{}, 1);                     // D(-1): its PD0 pieces of slice 1
            frags_in();
            for (int kt = 0; kt < KT; ++kt) {
                const bool seam = NSEG > 1 && kt == KS - 1;
                phase();                                                   // A: MFMA k-step 0 (+ MV of B's pieces)
                mma([&]() { if (MV > 0 && kt + 1 < KT) issue(I0{}, I0{}, IMV{}, kt + 1); });
                phase();                                                   // B: fragments of k-step 1, PB0 pieces of slice kt+1
                load_frags(2 * kt + 1);
                if (kt + 1 < KT) issue(I0{}, IMV{}, IPB0{}, kt + 1);
                frags_in();
                phase();                                                   // C: MFMA k-step 1
                mma(nothing);
                landed();                                                  // this wave's pieces of slice kt+1
                phase();                                                   // D: fragments of slice kt+1, PD0 pieces of slice kt+2
                if (seam) {
                    if (kt + 2 < KT) issue(I0{}, IPB0{}, IP0{}, kt + 2);
                    between(0);
                    zero();
                    load_frags(2 * kt + 2);
                } else {
                    if (kt + 1 < KT) load_frags(2 * kt + 2);
                    if (kt + 2 < KT) issue(I0{}, IPB0{}, IP0{}, kt + 2);
                }
                frags_in();
            }
        } else {
            if (KT > 1) issue(I1{}, I0{}, IPC1{}, 1);                      // C(-1): own-strip pieces of slice 1
            for (int kt = 0; kt < KT; ++kt) {
                phase();                                                   // A: fragments of k-step 0, PA1 pieces of slice kt+1
                load_frags(2 * kt);
                if (kt + 1 < KT) issue(I1{}, IPC1{}, IA1{}, kt + 1);
                frags_in();
                phase();                                                   // B: MFMA k-step 0 (+ the last MV of A's pieces)
                mma([&]() { if (MV > 0 && kt + 1 < KT) issue(I1{}, IA1{}, IP1{}, kt + 1); });
                phase();                                                   // C: fragments of k-step 1; then, the reads of this
                load_frags(2 * kt + 1);                                    // stage done, its own strip's pieces of slice kt+2
                frags_in();
                if (kt + 2 < KT) issue(I1{}, I0{}, IC1{}, kt + 2);
                landed_but(kt + 2 < KT);                                   // pieces of slice kt+1 (the newest PC1 - MV may fly)
                phase();                                                   // D: MFMA k-step 1 (+ the last MV of C's pieces)
                mma([&]() { if (MV > 0 && kt + 2 < KT) issue(I1{}, IC1{}, IPC1{}, kt + 2); });
                if (NSEG > 1 && kt == KS - 1) {                            // the seam: see above
                    between(0);
                    zero();
                }
            }
        }
        phase();                             // every wave is done with the ring before the caller reuses the LDS
#ifdef NR_STAMP
        if (blockIdx.x == 0 && blockIdx.y == 0 && lane == 0 && wc == 0) {
            nr_stamp_buf[8 + 2 * grp] = pp_work;       // cycles between barriers (MFMA + memory phases of this wave)
            nr_stamp_buf[9 + 2 * grp] = pp_bar;        // cycles waiting at the phase barriers
            nr_stamp_buf[12 + grp] = pp_mma;           // of `work`: the MFMA phases
        }
#endif
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
