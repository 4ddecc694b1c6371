// Fused local_level forward with the max-pools done IN REGISTERS (no LDS round trip of the block).
// Reference: NeighborRetr/models/modeling.py:499-512.  Same block decomposition as nr_sim.hip
// (whole (text, video) pairs per workgroup, 2 x 2 waves), but the token -> MFMA-fragment mapping is
// permuted so that in the 16x16 accumulator layout (row = 4*(lane>>4) + reg, col = lane&15)
//   * a wave strip of 16*MI rows holds 16/TPS texts of Nt = MI*TPS tokens: sub-tile i carries tokens
//     [TPS*i, TPS*i+TPS) of each of them  ->  text = (lane>>4) / (TPS/4), token = TPS*i + 4*((lane>>4) % (TPS/4)) + reg;
//   * a wave strip of 16*NI columns holds 16/FPS videos of Nv = NI*FPS frames: sub-tile n carries frames
//     [FPS*n, FPS*n+FPS)  ->  video = (lane&15) / FPS, frame = FPS*n + (lane&15) % FPS.
// max over the frames of a video  = max over n (registers) + DPP max over FPS adjacent lanes;
// max over the tokens of a text   = max over (i, reg) (registers) + lane exchange at distance 16 / 32;
// the weighted sums follow the same two patterns.  ~150 VALU instructions per wave replace the
// store of the 96x96 block to LDS and three LDS reduction phases.
// Supported token counts: Nt = MI*TPS, Nv = NI*FPS with (MI,TPS) in {(3,8),(4,16)}, (NI,FPS) in
// {(3,4),(4,16)}: 24/64 text tokens, 12/64 frames; everything else runs nr_sim.hip.
//
// Large products at 24 x 12 tokens (the memory-bank products) run 192 x 192 blocks = 8 texts x 16 videos on
// 2 x 4 waves (512 threads, (MI,TPS,NI,FPS) = (6,4,3,4)).  The main loop is bound by the operand bytes a CU can
// pull into LDS by LDS-DMA -- ~25 B/clk/CU measured (60 GB/s), the figure that pins the 96 x 96 blocks
// (341 B per MFMA) at 27 % and their split-bf16 form (227 B per MFMA) at 42 % of the MFMA peak -- so the
// way up is fewer bytes per MFMA: doubling both block edges halves them (171 B per MFMA), and eight waves
// keep two per SIMD so one computes while the other waits on its fragments.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include "nr_gemm_tile.h"
#include "../../include/nr_hip.h"

#ifdef NR_STAMP
extern "C" int nr_debug_stamps(unsigned long long* host, int reset) {
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(host, HIP_SYMBOL(nr_stamp_buf), sizeof(unsigned long long) * 256);
    (void)reset;
    return 16;
}
#endif

struct NrSimRegArgs {
    const uint16_t *t_hi, *t_lo, *v_hi, *v_lo;
    const float *w_t, *w_v;
    float* out;
    uint8_t *arg_v, *arg_t;
    float *pmax, *qmax;
    int A, Bv, K, out_mode;
    int ntx, nty, PR, PC;      // tile grid and its partition over the 8 XCDs (PR*PC == 8, or PR == 0: none)
    int rot_x, rot_y;          // K-slice rotation of workgroup (slot_x, slot_y) of an XCD part: rot_x*slot_x + rot_y*slot_y
    int dma_front;             // A/B hook: LDS-DMA issued in bursts behind the fragment reads
};

template <int W>   // max over W adjacent lanes (W = 2, 4, 8, 16) with the index of the first maximum
__device__ __forceinline__ void nr_lanes_argmax(float& v, int& idx) {
    nr_arg_step<true, NR_DPP_XOR1, 0xF>(v, idx);
    if constexpr (W >= 4) nr_arg_step<true, NR_DPP_XOR2, 0xF>(v, idx);
    if constexpr (W >= 8) nr_arg_step<true, NR_DPP_HALF_MIRROR, 0xF>(v, idx);
    if constexpr (W >= 16) nr_arg_step<true, NR_DPP_MIRROR, 0xF>(v, idx);
}
template <int W>
__device__ __forceinline__ float nr_lanes_max(float v) {
    v = fmaxf(v, nr_dpp<NR_DPP_XOR1>(v, v));
    if constexpr (W >= 4) v = fmaxf(v, nr_dpp<NR_DPP_XOR2>(v, v));
    if constexpr (W >= 8) v = fmaxf(v, nr_dpp<NR_DPP_HALF_MIRROR>(v, v));
    if constexpr (W >= 16) v = fmaxf(v, nr_dpp<NR_DPP_MIRROR>(v, v));
    return v;
}
template <int W>
__device__ __forceinline__ float nr_lanes_sum(float v) {
    v += nr_dpp<NR_DPP_XOR1>(v, v);
    if constexpr (W >= 4) v += nr_dpp<NR_DPP_XOR2>(v, v);
    if constexpr (W >= 8) v += nr_dpp<NR_DPP_HALF_MIRROR>(v, v);
    if constexpr (W >= 16) v += nr_dpp<NR_DPP_MIRROR>(v, v);
    return v;
}

// One block of one product: the whole kernel body, callable from the single-product kernel and from the grouped one
// (nr_sim_group_kernel, below).  `bid`: the block's index inside ITS product's tile grid.
// RT > 1 (TPS = 16 only: the strip is in memory order): a wave strip of 16*MI rows holds RT texts one behind the other, text
// rt in sub-tiles [rt*MI/RT, (rt+1)*MI/RT) -- twice the rows per wave for the same columns, i.e. the operand bytes per MFMA of
// the 192 x 384 blocks at 64 x 64 tokens: 256 x 256 blocks = 4 texts x 4 videos on 2 x 4 waves of 128 x 64 (MI, NI = 8, 4).
template <int MI, int NI, int TPS, int FPS, bool X3, bool ARGS, int STAGES, int WC, bool PP = false, int RT = 1, bool P3 = false>
__device__ __forceinline__ void nr_sim_reg_body(const NrSimRegArgs& p, const int bid, char* smem) {
    static_assert(!P3 || (PP && !X3), "three accumulated passes run on the one-pass tile's ping-pong loop");
    using Tile = NrGemmTile<MI, NI, X3, TPS, FPS, STAGES, WC>;
    static_assert(MI % RT == 0 && (RT == 1 || TPS == 16), "texts stacked in a wave strip: whole sub-tiles each, strip in memory order");
    constexpr int MIE = MI / RT;                      // sub-tiles per text
    constexpr int Nt = MIE * TPS, Nv = NI * FPS;
    constexpr int TXS = 16 / TPS;                     // texts side by side in a 16-row sub-tile
    constexpr int TAW = RT * TXS, TBW = 16 / FPS;     // texts / videos per wave
    constexpr int TA = 2 * TAW, TB = WC * TBW;        // per workgroup
    constexpr int GX = TPS / 4;                       // lane groups (of 16) that share a text
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave / WC, wc = wave % WC;
#ifdef NR_STAMP
    const unsigned long long t_start = __builtin_readcyclecounter();
#endif
    // XCD-aware tile order: workgroups b and b+8 land on the same XCD (round-robin dispatch), so XCD x is
    // given one contiguous PR x PC part of the tile grid and its private L2 pulls only that part's
    // operand rows.  Pure speed: any placement computes the same tiles.
    int bx, by, rot = 0;
    {
        if (p.PR > 0) {
            const int xcd = bid & 7, slot = bid >> 3;
            const int sub_w = p.ntx / p.PC, sub_h = p.nty / p.PR;
            bx = (xcd % p.PC) * sub_w + slot % sub_w;
            by = (xcd / p.PC) * sub_h + slot / sub_w;
            (void)sub_h;
            rot = (p.rot_x * (slot % sub_w) + p.rot_y * (slot / sub_w)) % (p.K / 64);
        } else {
            bx = bid % p.ntx;
            by = bid / p.ntx;
        }
    }
    const int row0 = by * TA * Nt, col0 = bx * TB * Nv;

    const int g = lane >> 4, kap = lane & 15;
    const int al = g / GX, tau0 = 4 * (g % GX);       // text within the wave, first token of this lane's regs
    const int bl = kap / FPS, phi = kap % FPS;        // video within the wave, frame offset
    int ag[RT], agc[RT];
    bool ok[RT];
    const int bg = bx * TB + wc * TBW + bl;
    const int bgc = min(bg, p.Bv - 1);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        ag[rt] = by * TA + wr * TAW + rt * TXS + al;
        ok[rt] = ag[rt] < p.A && bg < p.Bv;
        agc[rt] = min(ag[rt], p.A - 1);
    }
    // this lane's token weights: fetched before the main loop so their latency hides under it -- except in the
    // largest blocks, whose accumulators leave no registers to park them in: there the workgroup's TA*Nt + TB*Nv
    // weights are parked in the LDS behind the ring (LDS-DMA, 64 floats per instruction, issued ahead of the loop's
    // first slice and landed with it) and picked up after the loop -- a global load there would expose its whole
    // latency (~1.5 us of a 16 us launch) in front of the epilogue.
    constexpr bool LATE_W = MI * NI >= 32;
    float wt[MI][4], wv[NI];
    float* w_lds = reinterpret_cast<float*>(smem + Tile::RING_BYTES);
    constexpr int WT_N = TA * Nt, WV_N = TB * Nv;
    static_assert(!LATE_W || (WT_N % 64 == 0 && WV_N % 64 == 0), "weights go to the LDS 64 floats per instruction");
    auto load_weights = [&]() {
        if constexpr (LATE_W) {
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                f32x4_t q = *reinterpret_cast<const f32x4_t*>(w_lds + (wr * TAW + (i / MIE) * TXS + al) * Nt + TPS * (i % MIE) + tau0);
                wt[i][0] = q[0]; wt[i][1] = q[1]; wt[i][2] = q[2]; wt[i][3] = q[3];
            }
#pragma unroll
            for (int n = 0; n < NI; ++n) wv[n] = w_lds[WT_N + (wc * TBW + bl) * Nv + FPS * n + phi];
        } else {
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                f32x4_t q = *reinterpret_cast<const f32x4_t*>(p.w_t + (size_t)agc[i / MIE] * Nt + TPS * (i % MIE) + tau0);
                wt[i][0] = q[0]; wt[i][1] = q[1]; wt[i][2] = q[2]; wt[i][3] = q[3];
            }
#pragma unroll
            for (int n = 0; n < NI; ++n) wv[n] = p.w_v[(size_t)bgc * Nv + FPS * n + phi];
        }
    };
    if constexpr (!LATE_W) load_weights();
    else {
        // pieces of 64 floats: [0, WT_N/64) text weights, then the video weights; piece q goes out from wave q mod waves
        constexpr int NPIECE = (WT_N + WV_N) / 64;
#pragma unroll
        for (int q0 = 0; q0 < NPIECE; q0 += 2 * WC) {
            const int q = q0 + wave;
            if (q < NPIECE) {
                const bool is_t = q < WT_N / 64;
                const int e = (is_t ? q : q - WT_N / 64) * 64 + lane;           // element inside the block's weights
                const long src = is_t ? min((long)by * WT_N + e, (long)p.A * Nt - 1) : min((long)bx * WV_N + e, (long)p.Bv * Nv - 1);
                const float* gp = (is_t ? p.w_t : p.w_v) + src;
                __builtin_amdgcn_global_load_lds((nr_glb_ptr_t)gp, (nr_lds_ptr_t)(w_lds + q * 64), 4, 0, 0);
            }
        }
    }

    Tile tile;
    if constexpr (!PP) tile.zero();          // the ping-pong loop clears the accumulators itself, under its first DMA
#ifdef NR_STAMP
    if (blockIdx.x == 0 && threadIdx.x == 0) nr_stamp_buf[5] = __builtin_readcyclecounter() - t_start;
#endif
    float t2v[RT], v2t[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) t2v[rt] = v2t[rt] = 0.f;
    // The pooling epilogue (registers only: max-pools, weighted sums; in the ARGS form also the arg-max stores).  In the
    // ping-pong kernels the waves of group 0 (wave row 0) run it inside the K loop's last phase, beside group 1's final MFMAs.
    // (explicit fma in the weighted sums: S must not depend on how the compiler contracts them in one instantiation or
    // another; tests/test_fullsize_gpu.py holds row / column permutations to bit equality)
    auto pool = [&]() {
    if constexpr (LATE_W) load_weights();
    if constexpr (!ARGS) {
        // Loss-only / evaluation form: only the pooled VALUES are needed, so the pools are plain max chains
        // (v_max3_f32) -- a fifth of the instructions of the arg-tracking form below, which spent ~1000 VALU
        // instructions per wave (compare + select per element) and with them a third of the kernel's time.
        // Same values: max is exact, the weighted sums run in the same order.
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float m = tile.acc[i][0][j];
#pragma unroll
                for (int n = 1; n < NI; ++n) m = fmaxf(m, tile.acc[i][n][j]);
                m = nr_lanes_max<FPS>(m);
                t2v[i / MIE] = __builtin_fmaf(m, wt[i][j], t2v[i / MIE]);
            }
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            if constexpr (GX >= 2) t2v[rt] += __shfl_xor(t2v[rt], 16);
            if constexpr (GX >= 4) t2v[rt] += __shfl_xor(t2v[rt], 32);
        }
#pragma unroll
        for (int n = 0; n < NI; ++n)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                float m = tile.acc[rt * MIE][n][0];
#pragma unroll
                for (int i = rt * MIE; i < (rt + 1) * MIE; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) m = fmaxf(m, tile.acc[i][n][j]);
                if constexpr (GX >= 2) m = fmaxf(m, __shfl_xor(m, 16));
                if constexpr (GX >= 4) m = fmaxf(m, __shfl_xor(m, 32));
                v2t[rt] = __builtin_fmaf(m, wv[n], v2t[rt]);
            }
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) v2t[rt] = nr_lanes_sum<FPS>(v2t[rt]);
    } else {
    // ---- t2v: P[t] = max over the video's frames; sum_t w_t[t] * P[t] -------------------------------
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int t = TPS * (i % MIE) + tau0 + j;
            float m = tile.acc[i][0][j];
            int av = phi;
#pragma unroll
            for (int n = 1; n < NI; ++n) {
                float x = tile.acc[i][n][j];
                if (x > m) { m = x; av = FPS * n + phi; }
            }
            if constexpr (ARGS) {
                nr_lanes_argmax<FPS>(m, av);
                if (ok[i / MIE] && phi == 0) {
                    size_t o = ((size_t)ag[i / MIE] * p.Bv + bg) * Nt + t;
                    p.arg_v[o] = (uint8_t)av;
                    p.pmax[o] = m;
                }
            } else {
                m = nr_lanes_max<FPS>(m);
            }
            t2v[i / MIE] = __builtin_fmaf(m, wt[i][j], t2v[i / MIE]);
        }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        if constexpr (GX >= 2) t2v[rt] += __shfl_xor(t2v[rt], 16);
        if constexpr (GX >= 4) t2v[rt] += __shfl_xor(t2v[rt], 32);
    }

    // ---- v2t: Q[v] = max over the text's tokens; sum_v w_v[v] * Q[v] --------------------------------
#pragma unroll
    for (int n = 0; n < NI; ++n)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const int v = FPS * n + phi;
        float m = tile.acc[rt * MIE][n][0];
        int at = tau0;
#pragma unroll
        for (int i = rt * MIE; i < (rt + 1) * MIE; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (i == rt * MIE && j == 0) continue;
                float x = tile.acc[i][n][j];
                if (x > m) { m = x; at = TPS * (i - rt * MIE) + tau0 + j; }
            }
        if constexpr (GX >= 2) {
            float om = __shfl_xor(m, 16);
            int oa = __shfl_xor(at, 16);
            if (om > m || (om == m && oa < at)) { m = om; at = oa; }
        }
        if constexpr (GX >= 4) {
            float om = __shfl_xor(m, 32);
            int oa = __shfl_xor(at, 32);
            if (om > m || (om == m && oa < at)) { m = om; at = oa; }
        }
        if constexpr (ARGS) {
            if (ok[rt] && (g % GX) == 0) {
                size_t o = ((size_t)ag[rt] * p.Bv + bg) * Nv + v;
                p.arg_t[o] = (uint8_t)at;
                p.qmax[o] = m;
            }
        }
        v2t[rt] = __builtin_fmaf(m, wv[n], v2t[rt]);
      }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) v2t[rt] = nr_lanes_sum<FPS>(v2t[rt]);

    }
    };      // pool

    // (group 0's pooling inside the loop's last phase, beside group 1's final MFMAs, was built and measured: 16.2 / 15.8 us
    // against 16.1 / 15.4 for the bank products -- nothing; the epilogue stays behind the loop.)
    if constexpr (P3) tile.run_pp3(p.t_hi, p.t_lo, row0, p.A * Nt, p.v_hi, p.v_lo, col0, p.Bv * Nv, p.K, smem);
    else if constexpr (PP) tile.run_pp(p.t_hi, p.t_lo, row0, p.A * Nt, p.v_hi, p.v_lo, col0, p.Bv * Nv, p.K, smem);
    else tile.run(p.t_hi, p.t_lo, row0, p.A * Nt, p.v_hi, p.v_lo, col0, p.Bv * Nv, p.K, smem, rot, p.dma_front != 0);
    pool();
#ifdef NR_STAMP
    if (blockIdx.x == 0 && threadIdx.x == 0) nr_stamp_buf[6] = __builtin_readcyclecounter() - t_start;
#endif

    float S[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) S[rt] = 0.5f * (t2v[rt] + v2t[rt]);
#ifdef NR_STAMP
    if (blockIdx.x == 0 && threadIdx.x == 0) nr_stamp_buf[4] = __builtin_readcyclecounter() - t_start;
#endif
    const bool writer = (phi == 0) && ((g % GX) == 0);
    if (p.out_mode == NR_OUT_FULL) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
            if (writer && ok[rt]) p.out[(size_t)ag[rt] * p.Bv + bg] = S[rt];
        return;
    }
    // ---- block-level row / column sums of S (fixed order => deterministic) ---------------------------
    float* sSP = reinterpret_cast<float*>(smem);       // the ring is free: run() ended with a barrier
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
        if (writer) sSP[(wr * TAW + rt * TXS + al) * TB + wc * TBW + bl] = ok[rt] ? S[rt] : 0.f;
    __syncthreads();
    if (p.out_mode == NR_OUT_ROWSUM) {
        if (tid < TA) {
            float s = 0.f;
            for (int b = 0; b < TB; ++b) s += sSP[tid * TB + b];
            int a = by * TA + tid;
            if (a < p.A) p.out[(size_t)bx * p.A + a] = s;
        }
    } else {
        if (tid < TB) {
            float s = 0.f;
            for (int a = 0; a < TA; ++a) s += sSP[a * TB + tid];
            int b = bx * TB + tid;
            if (b < p.Bv) p.out[(size_t)by * p.Bv + b] = s;
        }
    }
}

template <int MI, int NI, int TPS, int FPS, bool X3, bool ARGS, int STAGES, int WC, bool PP = false, int RT = 1, bool P3 = false>
__global__ __launch_bounds__(128 * WC) void nr_sim_reg_kernel(NrSimRegArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    nr_sim_reg_body<MI, NI, TPS, FPS, X3, ARGS, STAGES, WC, PP, RT, P3>(p, blockIdx.x, smem);
}

// ---- two chained tiles per workgroup (the step's two bank products) ---------------------------------------------------
// A launch of the 192 x 384 block spends ~29 % of its time outside the K loop, INSIDE every block (tools/sim_ksweep.py,
// tools/stamps.py: set-up, first-slice latency, epilogue) -- a grouped grid does not touch that.  Here block b computes tile
// b of product 0 and then tile b of product 1 through ONE ping-pong loop (NrGemmTile::run_pp_segs<2>): the second tile's
// first two K slices are requested while the first tile's last slice is multiplied, the first tile's epilogue runs at the
// seam -- group 0's part of it beside group 1's last MFMA phase --, and set-up / dispatch are paid once.  Both products
// must have the same tile count and K.  LDS: ring + both tiles' token weights + the block-sum staging row (the ring is
// never free at the seam).
template <int MI, int NI, int TPS, int FPS>
__device__ __forceinline__ void nr_sim_pair_body(const NrSimRegArgs& p0, const NrSimRegArgs& p1, const int bid, char* smem) {
    constexpr int WC = 4;
    using Tile = NrGemmTile<MI, NI, false, TPS, FPS, 2, WC>;
    constexpr int Nt = MI * TPS, Nv = NI * FPS;
    constexpr int TAW = 16 / TPS, TBW = 16 / FPS;
    constexpr int TA = 2 * TAW, TB = WC * TBW;
    constexpr int GX = TPS / 4;
    constexpr int WT_N = TA * Nt, WV_N = TB * Nv;
    static_assert(WT_N % 64 == 0 && WV_N % 64 == 0, "weights go to the LDS 64 floats per instruction");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave / WC, wc = wave % WC;
    const int g = lane >> 4, kap = lane & 15;
    const int al = g / GX, tau0 = 4 * (g % GX);
    const int bl = kap / FPS, phi = kap % FPS;
    float* w_lds = reinterpret_cast<float*>(smem + Tile::RING_BYTES);
    float* sSP = w_lds + 2 * (WT_N + WV_N);

    auto coords = [&](const NrSimRegArgs& p, int& bx, int& by) {
        if (p.PR > 0) {
            const int xcd = bid & 7, slot = bid >> 3;
            const int sub_w = p.ntx / p.PC, sub_h = p.nty / p.PR;
            bx = (xcd % p.PC) * sub_w + slot % sub_w;
            by = (xcd / p.PC) * sub_h + slot / sub_w;
        } else {
            bx = bid % p.ntx;
            by = bid / p.ntx;
        }
    };
    int bx0, by0, bx1, by1;
    coords(p0, bx0, by0);
    coords(p1, bx1, by1);

    // both tiles' token weights -> LDS (pieces of 64 floats, dealt over the waves)
    auto park_weights = [&](const NrSimRegArgs& p, int bx, int by, float* dst) {
        constexpr int NPIECE = (WT_N + WV_N) / 64;
#pragma unroll
        for (int q0 = 0; q0 < NPIECE; q0 += 2 * WC) {
            const int q = q0 + wave;
            if (q < NPIECE) {
                const bool is_t = q < WT_N / 64;
                const int e = (is_t ? q : q - WT_N / 64) * 64 + lane;
                const long src = is_t ? min((long)by * WT_N + e, (long)p.A * Nt - 1) : min((long)bx * WV_N + e, (long)p.Bv * Nv - 1);
                const float* gp = (is_t ? p.w_t : p.w_v) + src;
                __builtin_amdgcn_global_load_lds((nr_glb_ptr_t)gp, (nr_lds_ptr_t)(dst + q * 64), 4, 0, 0);
            }
        }
    };
    park_weights(p0, bx0, by0, w_lds);
    park_weights(p1, bx1, by1, w_lds + WT_N + WV_N);

    Tile tile;                                // cleared inside run_pp_segs
    // the loss-only epilogue of nr_sim_reg_body (pooled values by max chains), on the weights parked at `wl`
    auto epilogue = [&](const NrSimRegArgs& p, int bx, int by, const float* wl) {
        const int ag = by * TA + wr * TAW + al;
        const int bg = bx * TB + wc * TBW + bl;
        const bool ok = ag < p.A && bg < p.Bv;
        float wt[MI][4], wv[NI];
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            f32x4_t q = *reinterpret_cast<const f32x4_t*>(wl + (wr * TAW + al) * Nt + TPS * i + tau0);
            wt[i][0] = q[0]; wt[i][1] = q[1]; wt[i][2] = q[2]; wt[i][3] = q[3];
        }
#pragma unroll
        for (int n = 0; n < NI; ++n) wv[n] = wl[WT_N + (wc * TBW + bl) * Nv + FPS * n + phi];
        float t2v = 0.f, v2t = 0.f;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float m = tile.acc[i][0][j];
#pragma unroll
                for (int n = 1; n < NI; ++n) m = fmaxf(m, tile.acc[i][n][j]);
                m = nr_lanes_max<FPS>(m);
                t2v += m * wt[i][j];
            }
        if constexpr (GX >= 2) t2v += __shfl_xor(t2v, 16);
        if constexpr (GX >= 4) t2v += __shfl_xor(t2v, 32);
#pragma unroll
        for (int n = 0; n < NI; ++n) {
            float m = tile.acc[0][n][0];
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) m = fmaxf(m, tile.acc[i][n][j]);
            if constexpr (GX >= 2) m = fmaxf(m, __shfl_xor(m, 16));
            if constexpr (GX >= 4) m = fmaxf(m, __shfl_xor(m, 32));
            v2t += m * wv[n];
        }
        v2t = nr_lanes_sum<FPS>(v2t);
        const float S = 0.5f * (t2v + v2t);
        const bool writer = (phi == 0) && ((g % GX) == 0);
        if (p.out_mode == NR_OUT_FULL) {
            if (writer && ok) p.out[(size_t)ag * p.Bv + bg] = S;
            return;
        }
        if (writer) sSP[(wr * TAW + al) * TB + wc * TBW + bl] = ok ? S : 0.f;
        __syncthreads();
        if (p.out_mode == NR_OUT_ROWSUM) {
            if (tid < TA) {
                float s_ = 0.f;
                for (int b = 0; b < TB; ++b) s_ += sSP[tid * TB + b];
                int a = by * TA + tid;
                if (a < p.A) p.out[(size_t)bx * p.A + a] = s_;
            }
        } else {
            if (tid < TB) {
                float s_ = 0.f;
                for (int a = 0; a < TA; ++a) s_ += sSP[a * TB + tid];
                int b = bx * TB + tid;
                if (b < p.Bv) p.out[(size_t)by * p.Bv + b] = s_;
            }
        }
        __syncthreads();                       // the staging row is free again (the next epilogue is thousands of cycles away anyway)
    };
    const typename Tile::Seg sg[2] = {
        {p0.t_hi, p0.t_lo, by0 * TA * Nt, p0.A * Nt, p0.v_hi, p0.v_lo, bx0 * TB * Nv, p0.Bv * Nv},
        {p1.t_hi, p1.t_lo, by1 * TA * Nt, p1.A * Nt, p1.v_hi, p1.v_lo, bx1 * TB * Nv, p1.Bv * Nv}};
    tile.template run_pp_segs<2>(sg, p0.K, smem, [&](int) { epilogue(p0, bx0, by0, w_lds); });
    epilogue(p1, bx1, by1, w_lds + WT_N + WV_N);
}

__global__ __launch_bounds__(512) void nr_sim_pair_kernel(NrSimRegArgs p0, NrSimRegArgs p1) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    nr_sim_pair_body<6, 6, 4, 2>(p0, p1, blockIdx.x, smem);
}

// ---- grouped launch: several products of ONE step in one grid --------------------------------------------------------
// The step's three products (two bank products on 192 x 384 bf16 blocks, the batch x batch product on 96 x 192
// split-bf16 blocks) all run 8-wave workgroups on a 144 KB ring, one per CU, 256 blocks each.  Launched one by one every
// product pays its own dispatch, first-slice latency and drain (t0 of `tools/sim_ksweep.py`: ~5 us of a 16 us launch);
// in one grid a CU picks up its next block the moment the previous one retires.  Block b of the grid works on product
// g.first[k] <= b < g.first[k+1], block b - g.first[k] of its tile grid (each product keeps its own XCD-aware order
// as long as the block counts ahead of it are multiples of 8 -- block index mod 8 is the XCD; after a ragged product the
// following ones land on shifted XCDs: slower L2 reuse, same results).
#define NR_SIM_GROUP_MAX 4
struct NrSimGroup {
    NrSimRegArgs p[NR_SIM_GROUP_MAX];
    int first[NR_SIM_GROUP_MAX + 1];
    int kind[NR_SIM_GROUP_MAX];       // 0: bf16, 192 x 384 blocks; 1: split-bf16, 96 x 192 blocks
    int n;
};

__global__ __launch_bounds__(512) void nr_sim_group_kernel(NrSimGroup g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int b = blockIdx.x;
    int k = 0;
#pragma unroll
    for (int i = 1; i < NR_SIM_GROUP_MAX; ++i)
        if (i < g.n && b >= g.first[i]) k = i;
    k = __builtin_amdgcn_readfirstlane(k);
    if (g.kind[k] == 0) nr_sim_reg_body<6, 6, 4, 2, false, false, 2, 4, true>(g.p[k], b - g.first[k], smem);
    else nr_sim_reg_body<3, 3, 8, 4, true, false, 2, 4, true>(g.p[k], b - g.first[k], smem);
}

template <int MI, int NI, int TPS, int FPS, bool X3, bool ARGS, int STAGES, int WC, bool PP = false, int RT = 1, bool P3 = false>
static int nr_sim_reg_launch_s(NrSimRegArgs& a, hipStream_t st) {
    using Tile = NrGemmTile<MI, NI, X3, TPS, FPS, STAGES, WC>;
    auto kern = nr_sim_reg_kernel<MI, NI, TPS, FPS, X3, ARGS, STAGES, WC, PP, RT, P3>;
    size_t lds = Tile::RING_BYTES;
    if constexpr (MI * NI >= 32)          // the block's token weights, parked behind the ring (LATE_W in the kernel)
        lds += sizeof(float) * (2 * (16 / TPS) * MI * TPS + WC * (16 / FPS) * NI * FPS);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(kern, dim3(a.ntx * a.nty), dim3(128 * WC), lds, st, a);
    NR_LAUNCH_CHECK();
    return NR_OK;
}

// tile grid of a product and its partition over the 8 XCDs
template <int MI, int NI, int TPS, int FPS, int WC, int RT = 1>
static void nr_sim_reg_plan(NrSimRegArgs& a) {
    constexpr int TA = 2 * RT * (16 / TPS), TB = WC * (16 / FPS);
    a.ntx = (a.Bv + TB - 1) / TB;
    a.nty = (a.A + TA - 1) / TA;
    // partition of the tile grid over the 8 XCDs that minimises the operand bytes each L2 has to hold
    a.PR = a.PC = 0;
    {
        const int cand[4][2] = {{2, 4}, {4, 2}, {1, 8}, {8, 1}};
        double best = 1e300;
        for (int c = 0; c < 4; ++c) {
            int pr = cand[c][0], pc = cand[c][1];
            if (a.nty % pr || a.ntx % pc) continue;
            double cost = (double)a.nty * TA * (MI / RT * TPS) / pr + (double)a.ntx * TB * (NI * FPS) / pc;
            if (cost < best) { best = cost; a.PR = pr; a.PC = pc; }
        }
    }
    if (const char* e = nr_tune_env("NR_SIM_XCD")) {          // tuning hook: "0" = plain row-major order, "RxC" = forced partition
        int pr = 0, pc = 0;
        if (sscanf(e, "%dx%d", &pr, &pc) == 2 && pr * pc == 8 && a.nty % pr == 0 && a.ntx % pc == 0) { a.PR = pr; a.PC = pc; }
        else a.PR = a.PC = 0;
    }
    a.rot_x = a.rot_y = 0;
    a.dma_front = 0;
}

template <int MI, int NI, int TPS, int FPS, bool X3, bool ARGS, int WC = 2>
static int nr_sim_reg_launch(NrSimRegArgs& a, hipStream_t st) {
    nr_sim_reg_plan<MI, NI, TPS, FPS, WC>(a);
    if (const char* e = nr_tune_env("NR_SIM_DMA_FRONT")) a.dma_front = atoi(e);
    if (const char* e = nr_tune_env("NR_SIM_ROT")) {          // tuning hook: "XxY" K-slice rotation per slot column / row
        int rx = 0, ry = 0;
        if (sscanf(e, "%dx%d", &rx, &ry) == 2) { a.rot_x = rx; a.rot_y = ry; }
    }
    // 192 x 192 blocks (8 waves, one workgroup per CU): nobody else hides their DMA latency, so they run the
    // 2-deep ring -- except split-bf16, whose two stages (196 KB) exceed the LDS
    constexpr bool big = WC == 4;
    constexpr bool mid = WC == 2 && MI * NI >= 18;          // 96 x 192 split-bf16: 144 KB for two stages, one workgroup per CU
    // 96 x 192 split-bf16 on 2 x 4 waves (48 x 48 per wave): the two-stage ring fits (144 KB) -> ping-pong K loop
    if constexpr (big && X3 && MI * NI == 9) return nr_sim_reg_launch_s<MI, NI, TPS, FPS, X3, ARGS, 2, WC, true>(a, st);
    else {
    if constexpr (!(big && !X3)) {          // (8-wave one-pass blocks always prefetch for themselves: no one-stage form of them)
        const bool one_stage = big ? X3 : (mid ? false : nr_pick_stages((long)a.ntx * a.nty) == 1);
        if (one_stage) return nr_sim_reg_launch_s<MI, NI, TPS, FPS, X3, ARGS, 1, WC>(a, st);
    }
    if constexpr (big && X3) return NR_EUNSUPPORTED;
    else {
#ifdef NR_TUNE
        if constexpr (big && NI == 3) {          // 192 x 192: three stages fit (144 KB)
            const char* e = nr_tune_env("NR_SIM_STAGES");
            if (e && atoi(e) == 3) return nr_sim_reg_launch_s<MI, NI, TPS, FPS, X3, ARGS, 3, WC>(a, st);
        }
#endif
        if constexpr (big) {
            // 8-wave blocks on the two-stage ring: ping-pong K loop (NrGemmTile::run_pp); NR_SIM_PP=0 = the plain loop (A/B)
            const char* e = nr_tune_env("NR_SIM_PP");
            if (!(e && atoi(e) == 0) && a.K >= 128) return nr_sim_reg_launch_s<MI, NI, TPS, FPS, X3, ARGS, 2, WC, true>(a, st);
        }
        return nr_sim_reg_launch_s<MI, NI, TPS, FPS, X3, ARGS, 2, WC>(a, st);
    }
    }
}

// tile shape this path uses: texts / videos per workgroup; 0 if the token count is not covered.
// NR_SIM_BIG=0 keeps the 96 x 96 blocks everywhere (A/B hook).
// 0: 96 x 96 blocks; 1: 192 x 192 (8 texts x 16 videos); 2: 192 x 384 (8 x 32); 3: 96 x 192 (4 x 16, split-bf16
// only).  The largest block that still gives every CU a workgroup: the main loop is bound by the LDS-DMA bytes
// per CU.  NR_SIM_BIG=0/1/2 caps it, NR_SIM_BIG=3 forces level 3 for split-bf16 (A/B hooks).
static int nr_sim_reg_big(int A, int Nt, int Bv, int Nv, int prec) {
    if (Nt == 64 && Nv == 64) {       // 4: 128 x 256 blocks (2 texts x 4 videos) on 8 waves, once they fill the chip
        const char* e64 = nr_tune_env("NR_SIM_BIG");
        if (e64 && atoi(e64) == 0) return 0;
        // split-bf16: its own tile would run a 1-stage ring on the 128 x 256 block (measured slower) -- but the ONE-PASS 256 x 256
        // block takes it as three accumulated passes over K (NrGemmTile::run_pp3).  NR_SIM_BIG=4: the small blocks (A/B hook)
        if (prec == NR_PREC_BF16X3)
            return (!(e64 && atoi(e64) == 4) && (A % 4) == 0 && (long)(A / 4) * ((Bv + 3) / 4) >= 256) ? 5 : 0;
        // 5: 256 x 256 blocks (4 texts x 4 videos, two texts per wave strip) once THEY fill the chip: the operand bytes per
        // MFMA of the 192 x 384 blocks of the 24 x 12 shape.  NR_SIM_BIG=4 keeps the 128 x 256 blocks (A/B hook)
        if (!(e64 && atoi(e64) == 4) && (A % 4) == 0 && (long)(A / 4) * ((Bv + 3) / 4) >= 256) return 5;
        return (long)((A + 1) / 2) * ((Bv + 3) / 4) >= 256 ? 4 : 0;
    }
    if (Nt != 24 || Nv != 12) return 0;
    const bool x3 = prec == NR_PREC_BF16X3;
    int cap = x3 ? 1 : 2;       // split-bf16 fragments do not fit beside 144 accumulators
    const char* e = nr_tune_env("NR_SIM_BIG");
    const int env = e ? atoi(e) : -1;
    if (env >= 0 && env <= 2) cap = std::min(cap, env);
    const long wg3 = (long)((A + 3) / 4) * ((Bv + 15) / 16);
    // split-bf16 once the 192 x 384 blocks fill the chip: those ONE-PASS blocks, three accumulated passes over K (run_pp3) -- the
    // split tile's own largest block there (192 x 192) runs a one-deep ring.  NR_SIM_BIG=1 keeps the latter (A/B hook)
    if (x3 && env < 0 && (long)((A + 7) / 8) * ((Bv + 31) / 32) >= 256) return 2;
    if (x3 && env == 3 && wg3 >= 256) return 3;
    // (96 x 192 on the two-stage ping-pong loop only while the 192 x 192 blocks would not fill the chip: at B = 1024 the latter win
    // although they can only run a ONE-stage ring -- batch x batch product of configs[2] 1372 us against ~1510, step 2.638 vs
    // 2.661 ms, two A/B pairs in one session, profiles/r04_c2_ab.txt)
    if (x3 && env < 0 && wg3 >= 256 && (long)((A + 7) / 8) * ((Bv + 15) / 16) < 256) return 3;
    if (cap >= 2 && (long)((A + 7) / 8) * ((Bv + 31) / 32) >= 256) return 2;
    if (cap >= 1 && (long)((A + 7) / 8) * ((Bv + 15) / 16) >= 256) return 1;
    return 0;
}

extern "C" int nr_sim_reg_tile(int A, int Nt, int Bv, int Nv, int prec, int* TA, int* TB) {
    int ta = Nt == 24 ? 4 : (Nt == 64 ? 2 : 0);
    int tb = Nv == 12 ? 8 : (Nv == 64 ? 2 : 0);
    if (!ta || !tb) return 0;
    const int big = nr_sim_reg_big(A, Nt, Bv, Nv, prec);
    if (big == 5) { ta = 4; tb = 4; }
    else if (big == 4) { ta = 2; tb = 4; }
    else if (big == 3) { ta = 4; tb = 16; }
    else if (big) { ta = 8; tb = big == 2 ? 32 : 16; }
    if (TA) *TA = ta;
    if (TB) *TB = tb;
    return 1;
}

// returns NR_EUNSUPPORTED when the shape is not covered (the caller falls back to nr_sim.hip)
int nr_sim_reg_dispatch(const uint16_t* t_hi, const uint16_t* t_lo, const uint16_t* v_hi, const uint16_t* v_lo,
                        const float* w_t, const float* w_v, int A, int Nt, int Bv, int Nv, int d, int prec, int out_mode,
                        float* out, uint8_t* arg_v, uint8_t* arg_t, float* pmax, float* qmax, hipStream_t st) {
    if (!nr_sim_reg_tile(A, Nt, Bv, Nv, prec, nullptr, nullptr)) return NR_EUNSUPPORTED;
    NrSimRegArgs a{t_hi, t_lo, v_hi, v_lo, w_t, w_v, out, arg_v, arg_t, pmax, qmax, A, Bv, d, out_mode};
    const bool x3 = prec == NR_PREC_BF16X3, args = arg_v != nullptr;
#define NR_REG_CASE(MI_, TPS_, NI_, FPS_)                                                              \
    if (Nt == MI_ * TPS_ && Nv == NI_ * FPS_) {                                                        \
        if (x3) return args ? nr_sim_reg_launch<MI_, NI_, TPS_, FPS_, true, true>(a, st)               \
                            : nr_sim_reg_launch<MI_, NI_, TPS_, FPS_, true, false>(a, st);             \
        return args ? nr_sim_reg_launch<MI_, NI_, TPS_, FPS_, false, true>(a, st)                      \
                    : nr_sim_reg_launch<MI_, NI_, TPS_, FPS_, false, false>(a, st);                    \
    }
    if (nr_sim_reg_big(A, Nt, Bv, Nv, prec) == 5) {     // 64 x 64 tokens, 256 x 256 blocks on 2 x 4 waves
        nr_sim_reg_plan<8, 4, 16, 16, 4, 2>(a);
        if (x3)       // split-bf16 as three accumulated passes on the one-pass tile
            return args ? nr_sim_reg_launch_s<8, 4, 16, 16, false, true, 2, 4, true, 2, true>(a, st)
                        : nr_sim_reg_launch_s<8, 4, 16, 16, false, false, 2, 4, true, 2, true>(a, st);
        return args ? nr_sim_reg_launch_s<8, 4, 16, 16, false, true, 2, 4, true, 2>(a, st)
                    : nr_sim_reg_launch_s<8, 4, 16, 16, false, false, 2, 4, true, 2>(a, st);
    }
    if (nr_sim_reg_big(A, Nt, Bv, Nv, prec) == 4) {     // 64 x 64 tokens, 128 x 256 blocks on 2 x 4 waves
        if (x3) return args ? nr_sim_reg_launch<4, 4, 16, 16, true, true, 4>(a, st) : nr_sim_reg_launch<4, 4, 16, 16, true, false, 4>(a, st);
        return args ? nr_sim_reg_launch<4, 4, 16, 16, false, true, 4>(a, st) : nr_sim_reg_launch<4, 4, 16, 16, false, false, 4>(a, st);
    }
    if (nr_sim_reg_big(A, Nt, Bv, Nv, prec) == 3) {     // split-bf16, 96 x 192 blocks (4 texts x 16 videos), 2-deep ring
        // 2 x 4 waves of 48 x 48 on the ping-pong K loop (13.7 -> 13.2 us at B = 128); NR_SIM_X3PP=0: the 2 x 2-wave form (A/B)
        if (const char* e = nr_tune_env("NR_SIM_X3PP"); !(e && atoi(e) == 0) && d >= 128)
            return args ? nr_sim_reg_launch<3, 3, 8, 4, true, true, 4>(a, st) : nr_sim_reg_launch<3, 3, 8, 4, true, false, 4>(a, st);
        return args ? nr_sim_reg_launch<3, 6, 8, 2, true, true, 2>(a, st) : nr_sim_reg_launch<3, 6, 8, 2, true, false, 2>(a, st);
    }
    if (nr_sim_reg_big(A, Nt, Bv, Nv, prec) == 2 && x3) {     // ... split-bf16: three accumulated passes on the one-pass blocks
        if (d < 128) return NR_EUNSUPPORTED;
        nr_sim_reg_plan<6, 6, 4, 2, 4>(a);
        return args ? nr_sim_reg_launch_s<6, 6, 4, 2, false, true, 2, 4, true, 1, true>(a, st)
                    : nr_sim_reg_launch_s<6, 6, 4, 2, false, false, 2, 4, true, 1, true>(a, st);
    }
    if (nr_sim_reg_big(A, Nt, Bv, Nv, prec) == 2) {     // 24 x 12 tokens, 192 x 384 blocks on 2 x 4 waves
        return args ? nr_sim_reg_launch<6, 6, 4, 2, false, true, 4>(a, st) : nr_sim_reg_launch<6, 6, 4, 2, false, false, 4>(a, st);
    }
    if (nr_sim_reg_big(A, Nt, Bv, Nv, prec)) {     // 24 x 12 tokens, 192 x 192 blocks on 2 x 4 waves
        if (x3) return args ? nr_sim_reg_launch<6, 3, 4, 4, true, true, 4>(a, st) : nr_sim_reg_launch<6, 3, 4, 4, true, false, 4>(a, st);
        return args ? nr_sim_reg_launch<6, 3, 4, 4, false, true, 4>(a, st) : nr_sim_reg_launch<6, 3, 4, 4, false, false, 4>(a, st);
    }
    NR_REG_CASE(3, 8, 3, 4)      // 24 text tokens x 12 frames (MSR-VTT)
    NR_REG_CASE(4, 16, 4, 16)    // 64 x 64 (ActivityNet)
    NR_REG_CASE(3, 8, 4, 16)     // 24 x 64
    NR_REG_CASE(4, 16, 3, 4)     // 64 x 12
#undef NR_REG_CASE
    return NR_EUNSUPPORTED;
}

// ---- grouped launch (see nr_sim_group_kernel) -------------------------------------------------------------------------
// kind of block a product runs inside a group: 0 = bf16 on 192 x 384 blocks, 1 = split-bf16 on 96 x 192 blocks; -1: the
// product is not one a group takes (other token counts, or a size at which the single launch picks another block shape --
// the partial-sum outputs then have another layout).
extern "C" int nr_local_level_group_kind(int A, int Nt, int Bv, int Nv, int d, int prec) {
    if (A <= 0 || Bv <= 0 || d < 128 || (d % 64) != 0) return -1;
    const int big = nr_sim_reg_big(A, Nt, Bv, Nv, prec);
    if (big == 2 && prec == NR_PREC_BF16) return 0;
    if (big == 3 && prec == NR_PREC_BF16X3) return 1;
    return -1;
}

extern "C" int nr_local_level_group(int n, const NrLocalLevelProblem* probs, void* stream) {
    if (!probs || n <= 0) return NR_EINVAL;
    if (n > NR_SIM_GROUP_MAX) return NR_EUNSUPPORTED;
    NrSimGroup g;
    g.n = n;
    g.first[0] = 0;
    for (int i = 0; i < n; ++i) {
        const NrLocalLevelProblem& q = probs[i];
        if (!q.t_hi || !q.v_hi || !q.w_t || !q.w_v || !q.out) return NR_EINVAL;
        if (q.out_mode < 0 || q.out_mode > 2) return NR_EINVAL;
        if (q.prec == NR_PREC_BF16X3 && (!q.t_lo || !q.v_lo)) return NR_EINVAL;
        const int kind = nr_local_level_group_kind(q.A, q.Nt, q.Bv, q.Nv, q.d, q.prec);
        if (kind < 0) return NR_EUNSUPPORTED;
        NrSimRegArgs a{q.t_hi, q.t_lo, q.v_hi, q.v_lo, q.w_t, q.w_v, q.out, nullptr, nullptr, nullptr, nullptr,
                       q.A, q.Bv, q.d, q.out_mode};
        if (kind == 0) nr_sim_reg_plan<6, 6, 4, 2, 4>(a);
        else nr_sim_reg_plan<3, 3, 8, 4, 4>(a);
        g.p[i] = a;
        g.kind[i] = kind;
        g.first[i + 1] = g.first[i] + a.ntx * a.nty;
    }
    using TileA = NrGemmTile<6, 6, false, 4, 2, 2, 4>;
    using TileB = NrGemmTile<3, 3, true, 8, 4, 2, 4>;
    // two bf16 products of equal tile count and K at the front (the step's two bank products): chained tile pairs, one
    // launch; what is left of the list goes out grouped behind it
    if (n >= 2 && g.kind[0] == 0 && g.kind[1] == 0 && g.p[0].K == g.p[1].K &&
        g.p[0].ntx * g.p[0].nty == g.p[1].ntx * g.p[1].nty && !nr_tune_env("NR_SIM_NOPAIR")) {
        constexpr size_t lds_pair = TileA::RING_BYTES + sizeof(float) * (2 * (8 * 24 + 32 * 12) + 8 * 32);
        hipError_t e = hipFuncSetAttribute((const void*)nr_sim_pair_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_pair);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(nr_sim_pair_kernel, dim3(g.p[0].ntx * g.p[0].nty), dim3(512), lds_pair, (hipStream_t)stream, g.p[0], g.p[1]);
        NR_LAUNCH_CHECK();
        if (n == 2) return NR_OK;
        return nr_local_level_group(n - 2, probs + 2, stream);
    }
    for (int i = n; i < NR_SIM_GROUP_MAX; ++i) { g.p[i] = g.p[0]; g.kind[i] = g.kind[0]; g.first[i + 1] = g.first[n]; }
    constexpr size_t lds_a = TileA::RING_BYTES + sizeof(float) * (8 * 24 + 32 * 12);
    constexpr size_t lds_b = TileB::RING_BYTES;
    constexpr size_t lds = lds_a > lds_b ? lds_a : lds_b;
    hipError_t e = hipFuncSetAttribute((const void*)nr_sim_group_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(nr_sim_group_kernel, dim3(g.first[n]), dim3(512), lds, (hipStream_t)stream, g);
    NR_LAUNCH_CHECK();
    return NR_OK;
}
