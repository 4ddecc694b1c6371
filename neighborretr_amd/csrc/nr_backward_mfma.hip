// Backward of the fused local_level product with respect to one operand's (normalised) tokens, as an MFMA GEMM.
// Reference semantics: autograd of modeling.py:499-512 -- a max-pool routes its gradient to the arg-max, so with
// g(s,o) = 0.5 * dS(s,o) the gradient of token n of "self" sample s is
//     d_x[(s,n), :] = sum over other samples o, tokens m of   P[(s,n),(o,m)] * x_other[(o,m), :]
//     P[(s,n),(o,m)] = g(s,o) * ( w_other[o,m] * [scat(s,o,m) == n]  +  w_self[s,n] * [gath(s,o,n) == m] )
// (scat / gath = the arg-max indices stored by the forward kernel).  P is a [n_self_tokens, n_other_tokens]
// matrix with <= 2 non-zeros per (s, o, n) -- nr_sim_bwd_kernel walks it entry by entry (466 us per bank product
// on MI355X); here 96 x 96 blocks of P (whole samples on both sides: 96 = 4 x 24 = 8 x 12 tokens) are GENERATED
// in LDS as bf16 from the index bytes and multiplied on the matrix cores with the other operand's tokens.
//
// Workgroup = 8 waves; block = 96 self-token rows x (128 NI) feature dims x one chunk of the other operand's
// samples, walked in slices of 96 tokens.  The slice loop is a two-stage pipeline with ONE barrier per slice:
//     index bytes / pair gradients / token weights of slice i+2   global -> registers -> LDS staging buffer
//     P of slice i+1 generated from its staging buffer into the other P buffer                 (VALU + LDS)
//     P of slice i times the slice's tokens                                                     (MFMA)
// Waves 0-3 generate first and multiply second, waves 4-7 the other way round, so that the two waves of a SIMD
// keep its vector ALU and its matrix core busy at the same time.  Wave w owns dims [16 NI w, 16 NI (w+1)) of the
// block: its B fragments (x_other^T, [dim][token] bf16, k-contiguous) are private, so they go straight from global
// memory to registers, each k-step's fragments reloaded for the NEXT slice as soon as its MFMAs are issued.
// Several products run in one launch (descriptor table); chunks' partial sums are reduced in fixed order by
// nr_bwd_sum_group_kernel -- products that feed the same gradient are simply more chunks of it.
//
// The weights' gradient d_w[s,n] = sum_o g(s,o) pooled_max[s,o,n] (a matrix-vector product over the stored
// pooled maxima) has its own bandwidth-shaped kernel at the end of the file.
#include "nr_common.h"
#include "../../include/nr_hip.h"
#include <stddef.h>
#include <stdlib.h>
#include <initializer_list>
#include <mutex>

struct NrBwdMfmaArgs {
    const float* dS;
    int ds_mode;
    float ds_scale;
    const uint16_t* oT;                 // other operand's prepared tokens, fragment-major (nr_sim_bwd_operand_group)
    const uint16_t* oT_lo;              // optional low halves (same layout) or nullptr
    int ldk;                            // tokens covered by oT: whole slices of 96
    const float *w_self, *w_other;
    const uint8_t *gath, *scat;         // [A][Bv][Ns] and [A][Bv][No]
    int side, A, Bv, Ns, No, d;
    int n_self, n_other;                // samples
    int slices_per_chunk, n_slices, n_chunks;
    int row_tiles, dim_tiles;
    float* part;                        // [n_chunks][n_self*Ns][d]
    size_t part_stride;
};

#define BW_MAX_GROUP 4
struct NrBwdMfmaGroup {
    NrBwdMfmaArgs p[BW_MAX_GROUP];
    int tile_start[BW_MAX_GROUP + 1];
    int n;
    int dbg;                            // -DNR_TUNE builds only (NR_BWD_DBG): 1 skip generation, 2 skip MFMAs, 4 skip the stores
};

#define BW_ROWS 96
#define BW_K 96
#define BW_LDA 104                      // bf16 elements per P row in LDS: 208 B = 52 dwords -> conflict-free b128 reads
#define BW_THREADS 512
#define BW_PAIRS 32                     // (self sample, other sample) pairs of one 96 x 96 block, at most
#define BW_UNITS (BW_ROWS * BW_K / 4)   // generation work items: 4 consecutive columns of one row

struct NrBwdStage {                     // what one slice's P is generated from
    uint8_t gath[BW_PAIRS * 24];        // [pair][self token]  -> other token
    uint8_t scat[BW_PAIRS * 24];        // [pair][other token] -> self token
    float g[BW_PAIRS];                  // 0.5 * ds_scale * dS of the pair (0 outside the problem)
    float wo[BW_K];                     // other-token weights of the slice
};

template <int NI, bool LO>
__global__ __launch_bounds__(BW_THREADS) void nr_sim_bwd_mfma_kernel(NrBwdMfmaGroup grp) {
    __shared__ __attribute__((aligned(16))) uint16_t sP[2][BW_ROWS * BW_LDA];
    __shared__ __attribute__((aligned(16))) NrBwdStage s_st[2];
    __shared__ float s_ws[BW_ROWS];
    __shared__ uint16_t s_row[BW_ROWS], s_col[BW_K / 4];
    const int tile = nr_xcd_chunk_tile(blockIdx.x, grp.tile_start[grp.n]);
    if (tile < 0) return;
    int gi = 0;
    while (gi + 1 < grp.n && tile >= grp.tile_start[gi + 1]) ++gi;
    const NrBwdMfmaArgs& p = grp.p[gi];
    // chunk-major tile order: the workgroups of one chunk read the same tokens of the other operand and are neighbours
    // in tile order, i.e. on one XCD (nr_xcd_chunk_tile)
    int lt = tile - grp.tile_start[gi];
    const int per_chunk = p.row_tiles * p.dim_tiles;
    const int chunk = lt / per_chunk;
    lt -= chunk * per_chunk;
    const int dim_tile = lt / p.row_tiles, row_tile = lt - dim_tile * p.row_tiles;

#ifdef NR_TUNE
    const int dbg = grp.dbg;
#else
    constexpr int dbg = 0;
#endif
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int Ns = p.Ns, No = p.No;
    const int TS = BW_ROWS / Ns, TO = BW_K / No;              // samples per block row range / per k slice
    const int n_pairs = TS * TO;
    const int s0 = row_tile * TS;                             // first self sample of the block
    const int dim0 = dim_tile * (128 * NI) + wave * (16 * NI);
    const int sl_begin = chunk * p.slices_per_chunk;
    const int n_loc = min(sl_begin + p.slices_per_chunk, p.n_slices) - sl_begin;

    // ---- per-thread role in the staging step: ONE 32-bit word of one of the four staged arrays ------------------
    const char* st_src = nullptr;       // address of the word for slice 0 (nullptr: this thread stages nothing)
    size_t st_step = 0;                 // bytes per slice
    int st_oi = 0, st_dst = 0;          // other-sample offset inside the slice (validity), byte offset in NrBwdStage
    uint32_t st_fill = 0;
    float st_mul = 1.f;
    {
        const bool s_side = p.side == 0;
        if (tid < 384) {                                      // index bytes, four at a time (token counts are multiples of 4)
            const bool is_g = tid < 192;
            const int N = is_g ? Ns : No;
            const int e = 4 * (is_g ? tid : tid - 192);
            if (e < n_pairs * N) {
                const int pr = e / N, n = e - pr * N;
                const int si = pr / TO, oi = pr - si * TO;
                const int s = s0 + si;
                if (s < p.n_self) {
                    const uint8_t* base = is_g ? p.gath : p.scat;
                    const size_t pair0 = s_side ? (size_t)s * p.Bv + oi : (size_t)oi * p.Bv + s;
                    st_src = reinterpret_cast<const char*>(base + pair0 * N + n);
                    st_step = (s_side ? (size_t)TO : (size_t)TO * p.Bv) * N;
                }
                st_oi = oi;
                st_dst = (is_g ? (int)offsetof(NrBwdStage, gath) : (int)offsetof(NrBwdStage, scat)) + pr * 24 + n;
                st_fill = 0xffffffffu;
                if (!st_src) st_oi = 1 << 30;                 // never valid: the fill value is stored
            } else {
                st_dst = -1;
            }
        } else if (tid < 384 + BW_PAIRS) {                    // pair gradients
            const int pr = tid - 384;
            if (pr < n_pairs) {
                const int si = pr / TO, oi = pr - si * TO;
                const int s = s0 + si;
                st_oi = s < p.n_self ? oi : (1 << 30);
                // a = text sample, b = video sample of the pair; the slice advances the OTHER sample
                size_t idx0, step;
                if (p.ds_mode == 0) { idx0 = s_side ? (size_t)s * p.Bv + oi : (size_t)oi * p.Bv + s; step = s_side ? TO : (size_t)TO * p.Bv; }
                else if (p.ds_mode == 1) { idx0 = s_side ? s : oi; step = s_side ? 0 : TO; }
                else { idx0 = s_side ? oi : s; step = s_side ? TO : 0; }
                if (s < p.n_self) st_src = reinterpret_cast<const char*>(p.dS + idx0);
                st_step = step * sizeof(float);
                st_dst = (int)offsetof(NrBwdStage, g) + 4 * pr;
                st_mul = 0.5f * p.ds_scale;
            } else {
                st_dst = -1;
            }
        } else {                                              // other-token weights: 96 consecutive floats per slice
            const int e = tid - (384 + BW_PAIRS);
            st_oi = e / No;
            st_src = reinterpret_cast<const char*>(p.w_other + e);
            st_step = (size_t)BW_K * sizeof(float);
            st_dst = (int)offsetof(NrBwdStage, wo) + 4 * e;
        }
    }
    auto stage_load = [&](int sl) -> uint32_t {
        uint32_t v = st_fill;
        if (st_dst >= 0 && sl * TO + st_oi < p.n_other) v = *reinterpret_cast<const uint32_t*>(st_src + (size_t)sl * st_step);
        return v;
    };
    auto stage_store = [&](int buf, uint32_t v) {
        if (st_dst < 0) return;
        if (st_mul != 1.f) v = __float_as_uint(__uint_as_float(v) * st_mul);
        *reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(&s_st[buf]) + st_dst) = v;
    };

    // ---- slice-invariant tables ------------------------------------------------------------------------------------
    if (tid < BW_ROWS) {
        const int si = tid / Ns, n = tid - si * Ns;
        const int s = s0 + si;
        s_ws[tid] = s < p.n_self ? p.w_self[(size_t)s * Ns + n] : 0.f;
        s_row[tid] = (uint16_t)(((si * TO) << 8) | n);
    } else if (tid < BW_ROWS + BW_K / 4) {
        const int col = 4 * (tid - BW_ROWS);
        const int oi = col / No;
        s_col[tid - BW_ROWS] = (uint16_t)((oi << 8) | (col - oi * No));
    }

    f32x4_t acc[6][NI];
#pragma unroll
    for (int m = 0; m < 6; ++m)
#pragma unroll
        for (int n = 0; n < NI; ++n) acc[m][n] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // B fragments of one k-step.  The other operand arrives FRAGMENT-MAJOR (nr_sim_bwd_operand_group): the 16 dims x 32 tokens
    // of (slice, k-step, dim group) are one contiguous KiB in lane order, so a wave's load is eight whole cache lines
    // (lane = 16 (k / 8) + dim: dim = 16 group + (lane & 15), tokens 96 slice + 32 ks + 8 (lane >> 4) .. + 8).
    bf16x8_t bh[3][NI], bl[LO ? 3 : 1][LO ? NI : 1];
    const size_t b_lane = ((size_t)(dim0 / 16) * 64 + lane) * 8;
    const size_t b_step = (size_t)(p.d / 16) * 512;           // elements per (slice, k-step)
    auto load_b = [&](int slice, int ks) {
        const size_t o = (size_t)(slice * 3 + ks) * b_step + b_lane;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            bh[ks][ni] = *reinterpret_cast<const bf16x8_t*>(p.oT + o + ni * 512);
            if constexpr (LO) bl[ks][ni] = *reinterpret_cast<const bf16x8_t*>(p.oT_lo + o + ni * 512);
        }
    };

    // P of one slice: work item = 4 consecutive columns of one row (one 8-byte LDS store)
    auto generate = [&](int buf) {
        const NrBwdStage& S = s_st[buf];
        uint16_t* P = sP[buf];
#pragma unroll
        for (int r = 0; r < (BW_UNITS + BW_THREADS - 1) / BW_THREADS; ++r) {
            const int u = tid + BW_THREADS * r;
            if (u < BW_UNITS) {
                const int row = u / (BW_K / 4), cu = u - row * (BW_K / 4);
                const uint32_t ri = s_row[row], ci = s_col[cu];
                const int n = ri & 255, m = ci & 255;
                const int pair = (ri >> 8) + (ci >> 8);
                const float g = S.g[pair];
                const int gm = S.gath[pair * 24 + n] - m;                        // 0..3: the gathered token is in this item
                const uint32_t sc = *reinterpret_cast<const uint32_t*>(&S.scat[pair * 24 + m]);
                const f32x4_t wo = *reinterpret_cast<const f32x4_t*>(&S.wo[4 * cu]);
                const float gws = g * s_ws[row];
                const float v0 = ((sc & 255u) == (uint32_t)n ? g * wo[0] : 0.f) + (gm == 0 ? gws : 0.f);
                const float v1 = (((sc >> 8) & 255u) == (uint32_t)n ? g * wo[1] : 0.f) + (gm == 1 ? gws : 0.f);
                const float v2 = (((sc >> 16) & 255u) == (uint32_t)n ? g * wo[2] : 0.f) + (gm == 2 ? gws : 0.f);
                const float v3 = ((sc >> 24) == (uint32_t)n ? g * wo[3] : 0.f) + (gm == 3 ? gws : 0.f);
                uint2 pk;
                pk.x = (uint32_t)nr_f2bf(v0) | ((uint32_t)nr_f2bf(v1) << 16);
                pk.y = (uint32_t)nr_f2bf(v2) | ((uint32_t)nr_f2bf(v3) << 16);
                *reinterpret_cast<uint2*>(P + row * BW_LDA + 4 * cu) = pk;
            }
        }
    };

    // P (96 x 96) of buffer `buf` times the current slice's tokens; each k-step's B registers are reloaded for slice
    // `next` right after its MFMAs are issued
    auto multiply = [&](int buf, bool more, int next) {
        const uint16_t* P = sP[buf];
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
            bf16x8_t fa[6];
#pragma unroll
            for (int m = 0; m < 6; ++m)
                fa[m] = *reinterpret_cast<const bf16x8_t*>(P + (16 * m + (lane & 15)) * BW_LDA + 32 * ks + 8 * (lane >> 4));
#pragma unroll
            for (int m = 0; m < 6; ++m)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    acc[m][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[m], bh[ks][ni], acc[m][ni], 0, 0, 0);
            if constexpr (LO) {                               // the low halves afterwards: no back-to-back dependent MFMAs
#pragma unroll
                for (int m = 0; m < 6; ++m)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        acc[m][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[m], bl[ks][ni], acc[m][ni], 0, 0, 0);
            }
            if (more) load_b(next, ks);
        }
    };

    // ---- prologue: slice 0 staged and generated, slice 1 staged ---------------------------------------------------
    stage_store(0, stage_load(sl_begin));
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) load_b(sl_begin, ks);
    __syncthreads();
    generate(0);
    if (n_loc > 1) stage_store(1, stage_load(sl_begin + 1));
    __syncthreads();
    for (int i = 0; i < n_loc; ++i) {
        const int cur = i & 1;
        const bool has1 = i + 1 < n_loc, has2 = i + 2 < n_loc;
        uint32_t staged = 0;
        if (has2) staged = stage_load(sl_begin + i + 2);
        const bool gen_first = (dbg & 8) ? (wave & 1) == 0 : ((dbg & 16) ? true : wave < 4);
        if (gen_first) {
            if (has1 && !(dbg & 1)) generate(cur ^ 1);
            if (!(dbg & 2)) multiply(cur, has1, sl_begin + i + 1);
        } else {
            if (!(dbg & 2)) multiply(cur, has1, sl_begin + i + 1);
            if (has1 && !(dbg & 1)) generate(cur ^ 1);
        }
        if (has2) stage_store(cur, staged);                   // this buffer's slice was generated one iteration ago
        __syncthreads();
    }
    // ---- partial result of this chunk ---------------------------------------------------------------------------
    float* out = p.part + (size_t)chunk * p.part_stride;
    const size_t n_rows = (size_t)p.n_self * Ns;
#pragma unroll
    for (int m = 0; m < 6; ++m)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const size_t r = (size_t)s0 * Ns + 16 * m + 4 * (lane >> 4) + j;
                const int dim = dim0 + 16 * ni + (lane & 15);
                if (r < n_rows && dim < p.d && !(dbg & 4)) out[r * p.d + dim] = acc[m][ni][j];
            }
}

// out = (accumulate ? out : 0) + the chunks of every product that feeds it, in fixed (product, chunk) order
struct NrBwdSumGroup {
    struct Out {
        float* out;
        size_t n;
        int accumulate, n_src;
        const float* part[BW_MAX_GROUP];
        int n_chunks[BW_MAX_GROUP];
    } o[BW_MAX_GROUP];
    int n;
};

__global__ __launch_bounds__(256) void nr_bwd_sum_group_kernel(NrBwdSumGroup g) {
    const NrBwdSumGroup::Out& o = g.o[blockIdx.y];
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= o.n) return;
    f32x4_t s = o.accumulate ? *reinterpret_cast<const f32x4_t*>(o.out + i) : f32x4_t{0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < o.n_src; ++k)
        for (int c = 0; c < o.n_chunks[k]; ++c) s += *reinterpret_cast<const f32x4_t*>(o.part[k] + (size_t)c * o.n + i);
    *reinterpret_cast<f32x4_t*>(o.out + i) = s;
}

// ---- host side -----------------------------------------------------------------------------------------------------
extern "C" int nr_local_level_bwd_mfma_supported(int Nt, int Nv, int d) {
    if (Nt <= 0 || Nv <= 0 || Nt > 24 || Nv > 24 || (Nt % 4) != 0 || (Nv % 4) != 0) return 0;
    if ((BW_ROWS % Nt) != 0 || (BW_ROWS % Nv) != 0 || d <= 0 || (d % 256) != 0) return 0;
    return (BW_ROWS / Nt) * (BW_K / Nv) <= BW_PAIRS;       // 24 x 12, 12 x 24, 24 x 24, 16 x 24, 24 x 16
}

struct BwdPlan {
    NrBwdMfmaArgs a;
    size_t part_floats;
};

static int bwd_plan(const NrSimBwdItem& it, bool wide, BwdPlan& out) {
    if (!it.dS || !it.oT_hi || !it.w_self || !it.w_other || !it.arg_v || !it.arg_t || !it.d_x) return NR_EINVAL;
    if (it.side < 0 || it.side > 1 || it.ds_mode < 0 || it.ds_mode > 2 || it.A <= 0 || it.Bv <= 0) return NR_EINVAL;
    if (!nr_local_level_bwd_mfma_supported(it.Nt, it.Nv, it.d)) return NR_EUNSUPPORTED;
    NrBwdMfmaArgs& p = out.a;
    p.dS = it.dS; p.ds_mode = it.ds_mode; p.ds_scale = it.ds_scale; p.oT = it.oT_hi; p.oT_lo = it.oT_lo; p.ldk = it.ldk;
    p.w_self = it.w_self; p.w_other = it.w_other; p.side = it.side; p.A = it.A; p.Bv = it.Bv; p.d = it.d;
    if (it.side == 0) { p.Ns = it.Nt; p.No = it.Nv; p.gath = it.arg_v; p.scat = it.arg_t; p.n_self = it.A; p.n_other = it.Bv; }
    else              { p.Ns = it.Nv; p.No = it.Nt; p.gath = it.arg_t; p.scat = it.arg_v; p.n_self = it.Bv; p.n_other = it.A; }
    const int TS = BW_ROWS / p.Ns, TO = BW_K / p.No;
    if (it.ldk < ((p.n_other + TO - 1) / TO) * BW_K || (it.ldk % BW_K) != 0) return NR_EINVAL;    // whole slices
    p.row_tiles = (p.n_self + TS - 1) / TS;
    p.dim_tiles = it.d / (wide ? 512 : 256);
    p.n_slices = (p.n_other + TO - 1) / TO;
    p.part_stride = (size_t)p.n_self * p.Ns * it.d;
    return NR_OK;
}

// Chunks of each product: about `target` slices per workgroup, the same target for every product of the launch.  One
// workgroup is resident per CU (8 waves at > 128 VGPRs), so what matters is how the workgroups pack into rounds: the target is
// chosen by dealing each candidate's workgroups, in launch order, to the CUs and taking the shortest finish time, with the
// traffic of the partial sums (written once, read once) as the tie-breaker.  Measured at B=128 / M=512 on MI355X: 512
// workgroups of <= 11 slices (two rounds) 99.6 us, 256 workgroups of <= 22 slices (one round) 84.7 us.
static void bwd_apply_target(BwdPlan* pl, int n, long target) {
    for (int i = 0; i < n; ++i) {
        NrBwdMfmaArgs& p = pl[i].a;
        int ch = (int)((p.n_slices + target / 2) / target);
        if (ch < 1) ch = 1;
        if (ch > 32) ch = 32;
        p.slices_per_chunk = (p.n_slices + ch - 1) / ch;
        p.n_chunks = (p.n_slices + p.slices_per_chunk - 1) / p.slices_per_chunk;
        pl[i].part_floats = (size_t)p.n_chunks * p.part_stride;
    }
}

static int bwd_cu_count() {
    static int cus = 0;
    if (!cus) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
        else cus = 256;
    }
    return cus;
}

static double bwd_finish_time(const BwdPlan* pl, int n, int slots) {
    // earliest-free-CU dealing with a binary min-heap of the CUs' busy times
    static thread_local double heap[1024];
    for (int c = 0; c < slots; ++c) heap[c] = 0;
    double finish = 0;
    for (int i = 0; i < n; ++i) {
        const NrBwdMfmaArgs& p = pl[i].a;
        const int per_chunk = p.row_tiles * p.dim_tiles;
        for (int c = 0; c < p.n_chunks; ++c) {
            const int sl = (c + 1 < p.n_chunks ? p.slices_per_chunk : p.n_slices - c * p.slices_per_chunk);
            for (int w = 0; w < per_chunk; ++w) {
                const double t = heap[0] + sl + 2.0;          // + prologue / epilogue, in slice times
                finish = t > finish ? t : finish;
                int k = 0;                                    // replace the root, sift down
                for (;;) {
                    int l = 2 * k + 1, r = l + 1, m = k;
                    double mv = t;
                    if (l < slots && heap[l] < mv) { m = l; mv = heap[l]; }
                    if (r < slots && heap[r] < mv) { m = r; }
                    if (m == k) break;
                    heap[k] = heap[m];
                    k = m;
                }
                heap[k] = t;
            }
        }
    }
    return finish;
}

static void bwd_chunks(BwdPlan* pl, int n) {
    if (const char* ov = nr_tune_env("NR_BWD_SLICES")) {
        if (atoi(ov) > 0) { bwd_apply_target(pl, n, atoi(ov)); return; }
    }
    // the choice depends on the tile counts only: remembered for the shapes seen last (a training run has one or two)
    struct Memo { unsigned long long key; long target; };
    static Memo memo[8];
    static int memo_next = 0;
    static std::mutex memo_lock;
    unsigned long long key = 1469598103934665603ull;
    for (int i = 0; i < n; ++i)
        for (int v : {pl[i].a.row_tiles, pl[i].a.dim_tiles, pl[i].a.n_slices}) key = (key ^ (unsigned long long)v) * 1099511628211ull;
    {
        std::lock_guard<std::mutex> hold(memo_lock);
        for (const Memo& m : memo)
            if (m.key == key && m.target) { bwd_apply_target(pl, n, m.target); return; }
    }
    const int cus = bwd_cu_count();
    const int slots = cus < 1024 ? cus : 1024;
    int max_slices = 1;
    for (int i = 0; i < n; ++i) max_slices = pl[i].a.n_slices > max_slices ? pl[i].a.n_slices : max_slices;
    double best = 0;
    long best_target = 0;
    for (long target = 4; target <= 64 && target <= 2L * max_slices; ++target) {
        bwd_apply_target(pl, n, target);
        double part_bytes = 0;
        for (int i = 0; i < n; ++i) part_bytes += 2.0 * sizeof(float) * pl[i].part_floats;
        const double cost = bwd_finish_time(pl, n, slots) * 2.6 + part_bytes / 5.5e6;   // us: 2.6 us per slice, sums at 5.5 TB/s
        if (!best_target || cost < best) { best = cost; best_target = target; }
    }
    if (!best_target) best_target = 8;
    {
        std::lock_guard<std::mutex> hold(memo_lock);
        memo[memo_next] = Memo{key, best_target};
        memo_next = (memo_next + 1) % 8;
    }
    bwd_apply_target(pl, n, best_target);
}

static bool bwd_wide(const NrSimBwdItem* items, int n) {
    for (int i = 0; i < n; ++i)
        if (items[i].oT_lo || (items[i].d % 512) != 0) return false;
    return true;
}

extern "C" size_t nr_local_level_bwd_group_workspace_bytes(int n, const NrSimBwdItem* items) {
    if (!items || n <= 0 || n > BW_MAX_GROUP) return 0;
    BwdPlan pl[BW_MAX_GROUP];
    const bool wide = bwd_wide(items, n);
    for (int i = 0; i < n; ++i)
        if (bwd_plan(items[i], wide, pl[i]) != NR_OK) return 0;
    bwd_chunks(pl, n);
    size_t total = 0;
    for (int i = 0; i < n; ++i) total += pl[i].part_floats;
    return total * sizeof(float) + 256;
}

extern "C" int nr_local_level_bwd_group(int n, const NrSimBwdItem* items, void* workspace, size_t workspace_bytes, void* stream) {
    if (!items || n <= 0 || n > BW_MAX_GROUP || !workspace) return NR_EINVAL;
    BwdPlan pl[BW_MAX_GROUP];
    const bool wide = bwd_wide(items, n);
    const bool lo = items[0].oT_lo != nullptr;
    for (int i = 0; i < n; ++i) {
        if ((items[i].oT_lo != nullptr) != lo || items[i].d != items[0].d) return NR_EINVAL;   // one kernel variant per launch
        const int rc = bwd_plan(items[i], wide, pl[i]);
        if (rc != NR_OK) return rc;
    }
    bwd_chunks(pl, n);
    NrBwdMfmaGroup g;
    NrBwdSumGroup sg;
    g.n = n;
    g.dbg = 0;
    if (const char* ov = nr_tune_env("NR_BWD_DBG")) g.dbg = atoi(ov);
    sg.n = 0;
    size_t used = 0, n_max = 0;
    int total = 0;
    for (int i = 0; i < n; ++i) {
        pl[i].a.part = reinterpret_cast<float*>(workspace) + used;
        used += pl[i].part_floats;
        g.p[i] = pl[i].a;
        g.tile_start[i] = total;
        total += pl[i].a.row_tiles * pl[i].a.dim_tiles * pl[i].a.n_chunks;
        // products that write the same gradient are summed together, in item order
        int o = 0;
        while (o < sg.n && sg.o[o].out != items[i].d_x) ++o;
        if (o == sg.n) {
            sg.o[o].out = items[i].d_x;
            sg.o[o].n = pl[i].a.part_stride;
            sg.o[o].accumulate = items[i].accumulate;
            sg.o[o].n_src = 0;
            ++sg.n;
        } else if (sg.o[o].n != pl[i].a.part_stride) {
            return NR_EINVAL;
        }
        sg.o[o].part[sg.o[o].n_src] = pl[i].a.part;
        sg.o[o].n_chunks[sg.o[o].n_src++] = pl[i].a.n_chunks;
        if (pl[i].a.part_stride > n_max) n_max = pl[i].a.part_stride;
    }
    for (int i = n; i <= BW_MAX_GROUP; ++i) g.tile_start[i] = total;
    if (used * sizeof(float) > workspace_bytes) return NR_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(nr_xcd_chunk_grid(total));
    if (wide) hipLaunchKernelGGL((nr_sim_bwd_mfma_kernel<4, false>), grid, dim3(BW_THREADS), 0, st, g);
    else if (lo) hipLaunchKernelGGL((nr_sim_bwd_mfma_kernel<2, true>), grid, dim3(BW_THREADS), 0, st, g);
    else hipLaunchKernelGGL((nr_sim_bwd_mfma_kernel<2, false>), grid, dim3(BW_THREADS), 0, st, g);
    hipLaunchKernelGGL(nr_bwd_sum_group_kernel, dim3((unsigned)((n_max / 4 + 255) / 256), sg.n), dim3(256), 0, st, sg);
    NR_LAUNCH_CHECK();
    return NR_OK;
}

static NrSimBwdItem bwd_single_item(int side, const float* dS, int ds_mode, float ds_scale, const uint16_t* oT_hi,
                                    const uint16_t* oT_lo, int ldk, const float* w_self, const float* w_other,
                                    const uint8_t* arg_v, const uint8_t* arg_t, int A, int Nt, int Bv, int Nv, int d, float* d_x,
                                    int accumulate) {
    NrSimBwdItem it;
    it.dS = dS; it.oT_hi = oT_hi; it.oT_lo = oT_lo; it.w_self = w_self; it.w_other = w_other; it.arg_v = arg_v; it.arg_t = arg_t;
    it.d_x = d_x; it.ds_scale = ds_scale; it.side = side; it.ds_mode = ds_mode; it.ldk = ldk; it.A = A; it.Nt = Nt; it.Bv = Bv;
    it.Nv = Nv; it.d = d; it.accumulate = accumulate;
    return it;
}

extern "C" size_t nr_local_level_bwd_mfma_workspace_bytes(int side, int A, int Nt, int Bv, int Nv, int d) {
    if (!nr_local_level_bwd_mfma_supported(Nt, Nv, d) || A <= 0 || Bv <= 0) return 0;
    // the widest variant any call of this shape can take (one pass, 512 dims per workgroup) has the fewest chunks; size
    // for the narrow one
    static const float dummy_f = 0.f;
    static const uint16_t dummy_h = 0;
    static const uint8_t dummy_b = 0;
    const int n_other_tok = side == 0 ? Bv * Nv : A * Nt;
    NrSimBwdItem it = bwd_single_item(side, &dummy_f, 0, 1.f, &dummy_h, &dummy_h, (n_other_tok + BW_K - 1) / BW_K * BW_K, &dummy_f, &dummy_f,
                                      &dummy_b, &dummy_b, A, Nt, Bv, Nv, d, const_cast<float*>(&dummy_f), 0);
    size_t narrow = nr_local_level_bwd_group_workspace_bytes(1, &it);
    it.oT_lo = nullptr;
    size_t wide = nr_local_level_bwd_group_workspace_bytes(1, &it);
    return narrow > wide ? narrow : wide;
}

extern "C" int nr_local_level_bwd_mfma(int side, const float* dS, int ds_mode, float ds_scale, const uint16_t* oT_hi,
                                       const uint16_t* oT_lo, int ldk, const float* w_self, const float* w_other,
                                       const uint8_t* arg_v, const uint8_t* arg_t, int A, int Nt, int Bv, int Nv, int d,
                                       float* d_x, int accumulate, void* workspace, void* stream) {
    if (!workspace) return NR_EINVAL;
    NrSimBwdItem it = bwd_single_item(side, dS, ds_mode, ds_scale, oT_hi, oT_lo, ldk, w_self, w_other, arg_v, arg_t, A, Nt, Bv, Nv,
                                      d, d_x, accumulate);
    return nr_local_level_bwd_group(1, &it, workspace, nr_local_level_bwd_group_workspace_bytes(1, &it), stream);
}

// The fixed-order slab sum on its own (weight-gradient GEMMs cut along K write one slab per cut)
extern "C" int nr_slab_sum_group(int n, const NrSlabSum* items, void* stream) {
    if (!items || n <= 0 || n > BW_MAX_GROUP) return NR_EINVAL;
    NrBwdSumGroup sg;
    sg.n = n;
    size_t n_max = 0;
    for (int i = 0; i < n; ++i) {
        const NrSlabSum& it = items[i];
        if (!it.out || it.n == 0 || (it.n % 4) != 0 || it.n_src < 1 || it.n_src > BW_MAX_GROUP) return NR_EINVAL;
        sg.o[i].out = it.out;
        sg.o[i].n = (size_t)it.n;
        sg.o[i].accumulate = it.accumulate;
        sg.o[i].n_src = it.n_src;
        for (int k = 0; k < it.n_src; ++k) {
            if (!it.part[k] || it.n_slabs[k] < 1) return NR_EINVAL;
            sg.o[i].part[k] = it.part[k];
            sg.o[i].n_chunks[k] = it.n_slabs[k];
        }
        if ((size_t)it.n > n_max) n_max = (size_t)it.n;
    }
    hipLaunchKernelGGL(nr_bwd_sum_group_kernel, dim3((unsigned)((n_max / 4 + 255) / 256), n), dim3(256), 0, (hipStream_t)stream, sg);
    NR_LAUNCH_CHECK();
    return NR_OK;
}

// ---- the "other" operand in the order the kernel reads it ---------------------------------------------------------------------
// Prepared tokens [n_tok][d] (bf16 hi, optional lo) -> fragment-major [slice][k-step][d / 16][64 lanes][8]: element j of lane
// (kg, n) of block (slice, ks, dg) is token 96 slice + 32 ks + 8 kg + j, dim 16 dg + n; tokens past n_tok are zeros.
// Workgroup = one slice x 64 dims: 96 coalesced row pieces in, twelve contiguous KiB blocks (16 bytes per lane) out.
struct NrBwdOperandGroup {
    NrSimBwdOperand it[BW_MAX_GROUP];
    int start[BW_MAX_GROUP + 1];
    int n;
};

__global__ __launch_bounds__(256) void nr_sim_bwd_operand_kernel(NrBwdOperandGroup g) {
    __shared__ uint32_t t[BW_K][65];
    int gi = 0;
    while (gi + 1 < g.n && (int)blockIdx.x >= g.start[gi + 1]) ++gi;
    const NrSimBwdOperand& it = g.it[gi];
    const int lt = blockIdx.x - g.start[gi];
    const int dt = it.d / 64;
    const int slice = lt / dt, dtile = lt - slice * dt;
    const int tid = threadIdx.x;
    for (int e = tid; e < BW_K * 64; e += 256) {
        const int r = e >> 6, c = e & 63;
        const int tok = slice * BW_K + r;
        uint32_t v = 0;
        if (tok < it.n_tok) {
            const size_t o = (size_t)tok * it.d + dtile * 64 + c;
            v = (uint32_t)it.hi[o] | ((uint32_t)(it.lo ? it.lo[o] : 0) << 16);
        }
        t[r][c] = v;
    }
    __syncthreads();
    for (int w = tid; w < 12 * 64; w += 256) {
        const int blk = w >> 6, lane = w & 63;
        const int ks = blk >> 2, dgl = blk & 3, n = lane & 15, kg = lane >> 4;
        uint32_t v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = t[32 * ks + 8 * kg + j][16 * dgl + n];
        const size_t o = ((((size_t)slice * 3 + ks) * (it.d / 16) + dtile * 4 + dgl) * 64 + lane) * 8;
        uint4 h, l;
        h.x = (v[0] & 0xffffu) | (v[1] << 16); h.y = (v[2] & 0xffffu) | (v[3] << 16);
        h.z = (v[4] & 0xffffu) | (v[5] << 16); h.w = (v[6] & 0xffffu) | (v[7] << 16);
        *reinterpret_cast<uint4*>(it.out_hi + o) = h;
        if (it.out_lo) {
            l.x = (v[0] >> 16) | (v[1] & 0xffff0000u); l.y = (v[2] >> 16) | (v[3] & 0xffff0000u);
            l.z = (v[4] >> 16) | (v[5] & 0xffff0000u); l.w = (v[6] >> 16) | (v[7] & 0xffff0000u);
            *reinterpret_cast<uint4*>(it.out_lo + o) = l;
        }
    }
}

extern "C" int nr_sim_bwd_operand_group(int n, const NrSimBwdOperand* items, void* stream) {
    if (!items || n <= 0 || n > BW_MAX_GROUP) return NR_EINVAL;
    NrBwdOperandGroup g;
    g.n = n;
    int total = 0;
    for (int i = 0; i < n; ++i) {
        const NrSimBwdOperand& it = items[i];
        if (!it.hi || !it.out_hi || it.n_tok <= 0 || it.d <= 0 || (it.d % 64) != 0) return NR_EINVAL;
        if (it.out_lo && !it.lo) return NR_EINVAL;
        g.it[i] = it;
        g.start[i] = total;
        total += ((it.n_tok + BW_K - 1) / BW_K) * (it.d / 64);
    }
    for (int i = n; i <= BW_MAX_GROUP; ++i) g.start[i] = total;
    hipLaunchKernelGGL(nr_sim_bwd_operand_kernel, dim3(total), dim3(256), 0, (hipStream_t)stream, g);
    NR_LAUNCH_CHECK();
    return NR_OK;
}

// ---- d_w: gradient of the token weights ----------------------------------------------------------------------------
// d_w[s, n] = sum over other samples o of 0.5 * scale * dS(s,o) * pooled[s,o,n]   (modeling.py:505-512: the weighted sums
// of the pooled maxima are linear in the weights), summed over the products that use the weights.
//   side 0 (self = row sample a, pooled = pmax [A,Bv,N]):  one workgroup per a; thread (q, n) walks b = q, q+per, ...
//                                                          (coalesced over (b, n)), then a fixed-order LDS reduction.
//   side 1 (self = column sample b, pooled = qmax [A,Bv,N]): one workgroup per 64 columns (b, m); wave w walks
//                                                          a = w, w+16, ... (rows of 256 B), fixed-order LDS reduction.
#define PW_MAX_JOBS 8
struct NrPoolWGroup {
    NrPoolWJob j[PW_MAX_JOBS];
    int start[PW_MAX_JOBS + 1];
    int n;
};

__device__ __forceinline__ float nr_pw_g(const NrPoolWSrc& s, int a, int b) {
    return s.ds_mode == 0 ? s.dS[(size_t)a * s.Bv + b] : (s.ds_mode == 1 ? s.dS[a] : s.dS[b]);
}

__global__ __launch_bounds__(1024) void nr_pool_weight_bwd_kernel(NrPoolWGroup g) {
    __shared__ float s_red[1024];
    int gi = 0;
    while (gi + 1 < g.n && (int)blockIdx.x >= g.start[gi + 1]) ++gi;
    const NrPoolWJob& J = g.j[gi];
    const int lb = blockIdx.x - g.start[gi];
    const int tid = threadIdx.x, N = J.N;
    float acc = 0.f;
    if (J.side == 0) {
        const int a = lb;
        const int per = 1024 / N;
        const int n = tid % N, q = tid / N;
        if (q < per) {
            for (int k = 0; k < J.n_src; ++k) {
                const NrPoolWSrc& s = J.src[k];
                const float mul = 0.5f * s.ds_scale;
                const float* row = s.pool + (size_t)a * s.Bv * N + n;
                float part = 0.f;
#pragma unroll 4
                for (int b = q; b < s.Bv; b += per) part += nr_pw_g(s, a, b) * row[(size_t)b * N];
                acc += mul * part;
            }
        }
        s_red[tid] = acc;
        __syncthreads();
        if (tid < N) {
            float sum = J.accumulate ? J.d_w[(size_t)a * N + tid] : 0.f;
            for (int r = 0; r < per; ++r) sum += s_red[r * N + tid];
            J.d_w[(size_t)a * N + tid] = sum;
        }
    } else {
        const int lane = tid & 63, wave = tid >> 6;
        const int cols = J.src[0].Bv * N;
        const int c = lb * 64 + lane;
        if (c < cols) {
            const int b = c / N;
            for (int k = 0; k < J.n_src; ++k) {
                const NrPoolWSrc& s = J.src[k];
                const float mul = 0.5f * s.ds_scale;
                float part = 0.f;
#pragma unroll 4
                for (int a = wave; a < s.A; a += 16) part += nr_pw_g(s, a, b) * s.pool[(size_t)a * cols + c];
                acc += mul * part;
            }
        }
        s_red[tid] = acc;
        __syncthreads();
        if (tid < 64 && c < cols) {
            float sum = J.accumulate ? J.d_w[c] : 0.f;
            for (int w = 0; w < 16; ++w) sum += s_red[w * 64 + tid];
            J.d_w[c] = sum;
        }
    }
}

extern "C" int nr_pool_weight_bwd_group(int n, const NrPoolWJob* jobs, void* stream) {
    if (!jobs || n <= 0 || n > PW_MAX_JOBS) return NR_EINVAL;
    NrPoolWGroup g;
    g.n = n;
    int total = 0;
    for (int i = 0; i < n; ++i) {
        const NrPoolWJob& J = jobs[i];
        if (!J.d_w || J.n_src < 1 || J.n_src > 2 || J.side < 0 || J.side > 1 || J.N <= 0 || J.N > 1024) return NR_EINVAL;
        for (int k = 0; k < J.n_src; ++k) {
            const NrPoolWSrc& s = J.src[k];
            if (!s.dS || !s.pool || s.ds_mode < 0 || s.ds_mode > 2 || s.A <= 0 || s.Bv <= 0) return NR_EINVAL;
            // the sources of one job share the differentiated operand
            if (J.side == 0 ? s.A != J.src[0].A : s.Bv != J.src[0].Bv) return NR_EINVAL;
        }
        g.j[i] = J;
        g.start[i] = total;
        total += J.side == 0 ? J.src[0].A : (J.src[0].Bv * J.N + 63) / 64;
    }
    for (int i = n; i <= PW_MAX_JOBS; ++i) g.start[i] = total;
    hipLaunchKernelGGL(nr_pool_weight_bwd_kernel, dim3(total), dim3(1024), 0, (hipStream_t)stream, g);
    NR_LAUNCH_CHECK();
    return NR_OK;
}
