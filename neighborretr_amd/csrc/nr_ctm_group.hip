// One CTM + TCBlock stage (reference cluster.py:670-717 + :938-965, no-grad forward) for a GROUP of
// independent problems -- in the training step: the text and the video tokens -- in SEVEN launches:
//   shift|split -> conv GEMM (+x) -> front (LayerNorm, score, norm1, distances) -> back (DPC-KNN, merge, norm1)
//   -> q and kv GEMMs (one grouped launch) -> score-biased attention -> proj GEMM (+merged +bias)
// Every launch carries the workgroups of all problems of the group (workgroup -> problem through a prefix
// table), so the two modalities advance in lockstep inside the same grids instead of competing from two
// streams: measured on MI355X, two clustering branches on two streams overlap to only ~0.68 of their
// summed time, while a grid that holds both costs max(text, video).  All GEMMs run split-bf16 on the MFMA
// tile engine (nr_linear.hip); their A operands are written hi/lo directly by the producing kernel.
#include <cstdlib>
#include "nr_ctm_bodies.h"
#include "nr_linear.h"
#include "../../include/nr_hip.h"

// plain rows as bf16 pairs: NR_SHIFT_ROWS rows per 256-thread workgroup, a float4 (one 8-byte pair store each) per thread and turn
#define NR_SHIFT_ROWS 8
__global__ __launch_bounds__(256) void nr_group_shift_kernel(NrGroupOf<NrShiftArgs> g) {
    NR_CRITICAL_PATH();
    const int gi = g.find(blockIdx.x);
    const NrShiftArgs& a = g.p[gi];
    const size_t e0 = (size_t)(blockIdx.x - g.start[gi]) * NR_SHIFT_ROWS * a.C, e1 = (size_t)a.N * a.C;      // (a.N: ALL rows of the problem here)
    for (size_t e = e0 + 4 * threadIdx.x; e < e0 + (size_t)NR_SHIFT_ROWS * a.C && e < e1; e += 1024) {
        const f32x4_t v = *reinterpret_cast<const f32x4_t*>(a.x + e);
        uint32_t h0, l0, h1, l1;
        nr_split_pk(v[0], v[1], h0, l0);
        nr_split_pk(v[2], v[3], h1, l1);
        *reinterpret_cast<uint2*>(a.hi + e) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(a.lo + e) = make_uint2(l0, l1);
    }
}

template <int CPL, int THREADS = CF_THREADS>
__global__ __launch_bounds__(THREADS) void nr_group_front_kernel(NrGroupOf<NrCtmFrontArgs> g) {
    NR_CRITICAL_PATH();
    extern __shared__ __attribute__((aligned(16))) float sx[];
    const int gi = g.find(blockIdx.x);
    nr_ctm_front_body<CPL, THREADS>(g.p[gi], blockIdx.x - g.start[gi], sx);
}

// front + back in one launch for stages without a mask (stage 1 of the step): the back half needs nothing from
// other workgroups then, and the normalised rows it merges are the ones the front half left in LDS
template <int CPL, int THREADS = CF_THREADS>
__global__ __launch_bounds__(THREADS) void nr_group_front_back_kernel(NrGroupOf<NrCtmFrontArgs> gf, NrGroupOf<NrCtmBackArgs> gb) {
    NR_CRITICAL_PATH();
    extern __shared__ __attribute__((aligned(16))) float sx[];
    const int gi = gf.find(blockIdx.x);
    const int b = blockIdx.x - gf.start[gi];
    nr_ctm_front_body<CPL, THREADS>(gf.p[gi], b, sx);
    __threadfence_block();                 // the distances / token weights this workgroup just stored are read back
    __syncthreads();
    nr_ctm_back_body<true, THREADS>(gb.p[gi], b, sx);
}

// second form of the bodies (nr_ctm_bodies.h): C % 256 == 0, 512-thread workgroups (256 for a handful of tokens)
template <int V4, int THREADS>
__global__ __launch_bounds__(THREADS, THREADS / 128) void nr_group_front2_kernel(NrGroupOf<NrCtmFrontArgs> g) {
    NR_CRITICAL_PATH();
    extern __shared__ __attribute__((aligned(16))) float sx[];
    const int gi = g.find(blockIdx.x);
    nr_ctm_front_body2<V4, THREADS>(g.p[gi], blockIdx.x - g.start[gi], sx);
}

// (the 16-cluster / 64-token form keeps a 64-entry distance row per lane: one workgroup per CU, up to 256 registers)
template <int V4, int THREADS, int MAXC, int MAXN>
__global__ __launch_bounds__(THREADS, MAXC <= 4 ? THREADS / 128 : THREADS / 256) void nr_group_front_back2_kernel(NrGroupOf<NrCtmFrontArgs> gf, NrGroupOf<NrCtmBackArgs> gb) {
    NR_CRITICAL_PATH();
    extern __shared__ __attribute__((aligned(16))) float sx[];
    const int gi = gf.find(blockIdx.x);
    const int b = blockIdx.x - gf.start[gi];
    const NrCtmFrontArgs& f = gf.p[gi];
    // the back half's tie-break draws: requested now, in LDS by the time it starts (a round trip less behind the front half)
    if ((int)threadIdx.x < f.N) sx[(size_t)f.N * f.C + (size_t)f.N * f.N + 64 + threadIdx.x] = gb.p[gi].noise[(size_t)b * f.N + threadIdx.x];
    nr_ctm_front_body2<V4, THREADS>(f, b, sx);
    __syncthreads();
    nr_ctm_back_body2<true, THREADS, MAXC, (256 * V4) / THREADS, MAXN>(gb.p[gi], b, sx, sx + (size_t)f.N * f.C, sx + (size_t)f.N * f.C + (size_t)f.N * f.N);
}

template <int THREADS, int MAXC, int CPT>
__global__ __launch_bounds__(THREADS, MAXC <= 4 ? THREADS / 128 : THREADS / 256) void nr_group_back2_kernel(NrGroupOf<NrCtmBackArgs> g, int use_lds) {
    NR_CRITICAL_PATH();
    extern __shared__ __attribute__((aligned(16))) float sxn[];
    const int gi = g.find(blockIdx.x);
    nr_ctm_back_body2<false, THREADS, MAXC, CPT, MAXC <= 4 ? 32 : 64>(g.p[gi], blockIdx.x - g.start[gi], use_lds ? sxn : nullptr, nullptr, nullptr);
}

template <int THREADS = BK_THREADS>
__global__ __launch_bounds__(THREADS) void nr_group_back_kernel(NrGroupOf<NrCtmBackArgs> g, int use_lds) {
    NR_CRITICAL_PATH();
    extern __shared__ __attribute__((aligned(16))) float sxn[];
    const int gi = g.find(blockIdx.x);
    nr_ctm_back_body<false, THREADS>(g.p[gi], blockIdx.x - g.start[gi], use_lds ? sxn : nullptr);
}

__global__ __launch_bounds__(1024) void nr_group_attention_kernel(NrGroupOf<NrAttnArgs> g, int use_lds) {
    NR_CRITICAL_PATH();
    extern __shared__ __attribute__((aligned(16))) float skv[];
    const int gi = g.find(blockIdx.x);
    nr_tc_attention_body(g.p[gi], blockIdx.x - g.start[gi], use_lds ? skv : nullptr);
}

// a workgroup per (sample, pair of heads): see nr_tc_attention_heads_body.  g.start[] counts WORKGROUPS here.
#define NR_ATTN_HG 2
__global__ __launch_bounds__(256) void nr_group_attention_heads_kernel(NrGroupOf<NrAttnArgs> g) {
    NR_CRITICAL_PATH();
    extern __shared__ __attribute__((aligned(16))) float skv[];
    const int gi = g.find(blockIdx.x);
    const int idx = blockIdx.x - g.start[gi], per_sample = g.p[gi].H / NR_ATTN_HG;
    nr_tc_attention_heads_body<NR_ATTN_HG>(g.p[gi], idx / per_sample, idx % per_sample, skv);
}

#ifdef NR_STAMP
extern "C" int nr_debug_front_stamps(unsigned long long* host) {
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(host, HIP_SYMBOL(nr_front_stamps), sizeof(unsigned long long) * 16);
    return 5;
}
extern "C" int nr_debug_back_stamps(unsigned long long* host) {
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(host, HIP_SYMBOL(nr_back_stamps), sizeof(unsigned long long) * 16);
    return 7;
}
#endif

// ---- workspace carve-up --------------------------------------------------------------------------------------
namespace {
struct Carve {
    char* base;
    size_t off;
    template <typename T>
    T* take(size_t count) {
        T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
        off += (count * sizeof(T) + 255) & ~(size_t)255;
        return p;
    }
};

struct StageBuffers {
    uint16_t *cat_hi, *cat_lo, *kvn_hi, *kvn_lo, *qn_hi, *qn_lo, *att_hi, *att_lo;
    float *y, *xn, *score, *tokw, *dist, *smax, *merged_pb, *q, *kv;
    size_t bytes;
};

StageBuffers carve(void* ws, long B, long N, long C, long cnum) {
    Carve c{static_cast<char*>(ws), 0};
    StageBuffers s;
    s.cat_hi = c.take<uint16_t>(B * N * 3 * C);
    s.cat_lo = c.take<uint16_t>(B * N * 3 * C);
    s.y = c.take<float>(B * N * C);
    s.xn = c.take<float>(B * N * C);
    s.kvn_hi = c.take<uint16_t>(B * N * C);
    s.kvn_lo = c.take<uint16_t>(B * N * C);
    s.score = c.take<float>(B * N);
    s.tokw = c.take<float>(B * N);
    s.dist = c.take<float>(B * N * N);
    s.smax = c.take<float>(B);
    s.merged_pb = c.take<float>(B * cnum * C);
    s.qn_hi = c.take<uint16_t>(B * cnum * C);
    s.qn_lo = c.take<uint16_t>(B * cnum * C);
    s.q = c.take<float>(B * cnum * C);
    s.kv = c.take<float>(B * N * 2 * C);
    s.att_hi = c.take<uint16_t>(B * cnum * C);
    s.att_lo = c.take<uint16_t>(B * cnum * C);
    s.bytes = c.off;
    return s;
}
}  // namespace

extern "C" size_t nr_ctm_stage_workspace_bytes(int n_samples, int N, int C, int cluster_num) {
    if (n_samples <= 0 || N <= 0 || C <= 0 || cluster_num <= 0) return 0;
    return carve(nullptr, n_samples, N, C, cluster_num).bytes;
}

// byte offsets inside the stage workspace of what a backward pass needs: y (conv output + residual, pre-LayerNorm),
// xn (LayerNorm output), score (masked tokens: -inf), tokw (exp(score)), merged_pb (cluster means + proj bias), q, kv;
// and smax [n]: every sample's largest pairwise distance, written by the front launch and max-reduced over the samples by
// the back launch (cluster.py:473-475 fills masked columns with the maximum over the WHOLE batch + 1) -- a rank that
// clusters only its own samples puts the all-reduced maximum into smax[0] between the two launches
extern "C" int nr_ctm_stage_workspace_layout(int n_samples, int N, int C, int cluster_num, size_t* offsets) {
    if (n_samples <= 0 || N <= 0 || C <= 0 || cluster_num <= 0 || !offsets) return NR_EINVAL;
    char* const base = reinterpret_cast<char*>(4096);
    const StageBuffers s = carve(base, n_samples, N, C, cluster_num);
    const void* p[8] = {s.y, s.xn, s.score, s.tokw, s.merged_pb, s.q, s.kv, s.smax};
    for (int i = 0; i < 8; ++i) offsets[i] = (size_t)(static_cast<const char*>(p[i]) - base);
    return NR_OK;
}

extern "C" int nr_ctm_stage_workspace_layout2(int n_samples, int N, int C, int cluster_num, size_t* offsets) {
    int rc = nr_ctm_stage_workspace_layout(n_samples, N, C, cluster_num, offsets);
    if (rc != NR_OK) return rc;
    char* const base = reinterpret_cast<char*>(4096);
    const StageBuffers s = carve(base, n_samples, N, C, cluster_num);
    const void* p[6] = {s.kvn_hi, s.kvn_lo, s.qn_hi, s.qn_lo, s.att_hi, s.att_lo};
    for (int i = 0; i < 6; ++i) offsets[8 + i] = (size_t)(static_cast<const char*>(p[i]) - base);
    return NR_OK;
}

extern "C" int nr_ctm_stage_fwd(const NrCtmStageDesc* d, int n, void* stream) {
    return nr_ctm_stage_fwd_range(d, n, 0, NR_CTM_STAGE_LAUNCHES, stream);
}

// launches [first, last) of the stage's seven: lets the host interleave them with other work in capture order
extern "C" int nr_ctm_stage_fwd_range(const NrCtmStageDesc* d, int n, int first, int last, void* stream) {
    if (!d || n <= 0 || n > NR_CTM_MAX_GROUP || first < 0 || last > NR_CTM_STAGE_LAUNCHES || first >= last) return NR_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    StageBuffers w[NR_CTM_MAX_GROUP];
    size_t front_lds = 0;
    for (int i = 0; i < n; ++i) {
        const NrCtmStageDesc& s = d[i];
        if (!s.x || !s.noise || !s.wconv_hi || !s.wconv_lo || !s.ln_w || !s.ln_b || !s.sc_w || !s.sc_b || !s.n1_w || !s.n1_b ||
            !s.wq_hi || !s.wq_lo || !s.wkv_hi || !s.wkv_lo || !s.wp_hi || !s.wp_lo || !s.proj_bias || !s.workspace || !s.out)
            return NR_EINVAL;
        if (s.n_samples <= 0 || s.N <= 0 || s.k <= 0 || s.k > s.N || s.cnum <= 0 || s.cnum > s.N || s.heads <= 0) return NR_EINVAL;
        if (s.N > 64 || s.C <= 0 || (s.C % 128) != 0 || s.C > 1024 || s.C != s.heads * 64) return NR_EUNSUPPORTED;
        if (s.cnum * (s.C / 128) > 16 * BK_MAXJ) return NR_EUNSUPPORTED;
        w[i] = carve(s.workspace, s.n_samples, s.N, s.C, s.cnum);
        size_t lds = (size_t)s.N * s.C * sizeof(float);
        if (lds > 150 * 1024) return NR_EUNSUPPORTED;
        front_lds = lds > front_lds ? lds : front_lds;
    }
    int rc;
    // 1. the token rows as bf16 pairs [rows, C] -- for the problems that do not bring them along (x_hi / x_lo: written by
    //    the previous stage's proj GEMM).  The k=3 convolution reads its neighbour rows in place: no shifted copy.
    bool all_split = true;
    for (int i = 0; i < n; ++i) {
        if ((d[i].x_hi == nullptr) != (d[i].x_lo == nullptr)) return NR_EINVAL;
        all_split = all_split && d[i].x_hi != nullptr;
    }
    if (first <= 0 && 0 < last && !all_split) {
        NrGroupOf<NrShiftArgs> g;
        g.n = 0;
        int total = 0;
        for (int i = 0; i < n; ++i) {
            if (d[i].x_hi) continue;
            const int rows = d[i].n_samples * d[i].N;
            g.p[g.n] = NrShiftArgs{d[i].x, rows, d[i].C, w[i].cat_hi, w[i].cat_lo, 1};       // (N = all rows: see the kernel)
            g.start[g.n++] = total;
            total += (rows + NR_SHIFT_ROWS - 1) / NR_SHIFT_ROWS;
        }
        for (int i = g.n; i <= NR_CTM_MAX_GROUP; ++i) g.start[i] = total;
        hipLaunchKernelGGL(nr_group_shift_kernel, dim3(total), dim3(256), 0, st, g);
        NR_LAUNCH_CHECK();
    }
    // 2. y = x + conv(x)  (k=3 convolution as a [B*N, 3C] x [3C, C] product whose A operand is read in place)
    if (first <= 1 && 1 < last) {
        NrLinearArgs p[NR_CTM_MAX_GROUP];
        for (int i = 0; i < n; ++i) {
            p[i] = NrLinearArgs{d[i].x_hi ? d[i].x_hi : w[i].cat_hi, d[i].x_hi ? d[i].x_lo : w[i].cat_lo, d[i].wconv_hi, d[i].wconv_lo,
                                d[i].conv_bias, d[i].x, w[i].y, d[i].n_samples * d[i].N, d[i].C, 3 * d[i].C};
            p[i].conv_n = d[i].N;
        }
        if ((rc = nr_linear_group_launch(p, n, st, true)) != NR_OK) return rc;
    }
    // 3 + 4. front (LayerNorm, score, exp, norm1, pairwise distances) and back (DPC-KNN assignment, weighted cluster
    //        means, norm1).  Two launches when a problem carries a mask (the back half then needs the maximum
    //        distance over ALL samples); ONE when none does (stage 1 of the step): launch index 2 runs both, index 3
    //        is empty.
    bool fusable = true;
    for (int i = 0; i < n; ++i) fusable = fusable && d[i].mask == nullptr;
    bool small_c = true;                     // (`small` below: registers sized for C <= 512 unless a problem is wider)
    for (int i = 0; i < n; ++i) small_c = small_c && d[i].C <= 512;
    // Stages whose back half is a launch of its own (a mask: the back half needs the maximum distance over all samples): the kv
    // projection rides in that launch (launch 3), launch 4 is the q projection alone.  NR_KV_APART=1 (tuning builds): as before.
    bool kv_beside_back = !(fusable && small_c) && small_c && !nr_tune_env("NR_KV_APART");
    for (int i = 0; i < n; ++i) kv_beside_back = kv_beside_back && d[i].cnum * (d[i].C / 128) <= 8 * BK_MAXJ;
    if ((first <= 2 && 2 < last) || (first <= 3 && 3 < last)) {
        NrGroupOf<NrCtmFrontArgs> gf;
        NrGroupOf<NrCtmBackArgs> gb;
        gf.n = gb.n = n;
        int total = 0;
        for (int i = 0; i < n; ++i) {
            const NrCtmStageDesc& s = d[i];
            gf.p[i] = NrCtmFrontArgs{w[i].y, s.mask, s.ln_w, s.ln_b, s.sc_w, s.sc_b, s.n1_w, s.n1_b, s.eps_ctm, 1.0f / sqrtf((float)s.C),
                                     s.N, s.C, w[i].xn, nullptr, w[i].score, w[i].tokw, w[i].dist, w[i].smax, w[i].kvn_hi, w[i].kvn_lo};
            gb.p[i] = NrCtmBackArgs{w[i].dist, w[i].smax, s.mask, s.noise, w[i].xn, w[i].tokw, s.n1_w, s.n1_b, s.proj_bias,
                                    s.n_samples, s.N, s.C, s.k, s.cnum, s.eps_n1, nullptr, w[i].merged_pb, nullptr, s.assign,
                                    w[i].qn_hi, w[i].qn_lo};
            gf.start[i] = gb.start[i] = total;
            total += s.n_samples;
        }
        for (int i = n; i <= NR_CTM_MAX_GROUP; ++i) gf.start[i] = gb.start[i] = total;
        bool small = true;                   // registers sized for C <= 512 unless a problem is wider
        for (int i = 0; i < n; ++i) small = small && d[i].C <= 512;
        // The second form of the two bodies (nr_ctm_bodies.h): every problem 512 channels wide (the model's width), at most 16
        // clusters.  NR_CTM_V1=1 (tuning builds) keeps the first form.
        bool v2 = !nr_tune_env("NR_CTM_V1");
        int maxc = 0, maxn = 0;
        size_t lds2 = 0;
        for (int i = 0; i < n; ++i) {
            v2 = v2 && d[i].C == 512 && d[i].cnum <= 16;
            maxc = d[i].cnum > maxc ? d[i].cnum : maxc;
            maxn = d[i].N > maxn ? d[i].N : maxn;
            const size_t need = nr_ctm_front2_lds_floats(d[i].N, d[i].C) * sizeof(float);
            lds2 = need > lds2 ? need : lds2;
        }
        // the back half's registers: 4 cluster accumulators and a 32-entry distance row per token, or 16 and 64
        const int back_form = (maxc <= 4 && maxn <= 32) ? 4 : 16;
        if (v2 && first <= 2 && 2 < last) {
            bool few = maxc <= 4;                // a handful of tokens per sample (stage 1 of the step): 256-thread workgroups
            for (int i = 0; i < n; ++i) few = few && d[i].N <= 8;
            const void* k = !fusable ? (const void*)nr_group_front2_kernel<2, 512>
                            : few    ? (const void*)nr_group_front_back2_kernel<2, 256, 4, 8>
                                     : (const void*)nr_group_front_back2_kernel<2, 512, 16, 64>;
            if (lds2 > 40 * 1024) {
                hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
                if (e != hipSuccess) return (int)e;
            }
            if (!fusable) hipLaunchKernelGGL((nr_group_front2_kernel<2, 512>), dim3(total), dim3(512), lds2, st, gf);
            else if (few) hipLaunchKernelGGL((nr_group_front_back2_kernel<2, 256, 4, 8>), dim3(total), dim3(256), lds2, st, gf, gb);
            else hipLaunchKernelGGL((nr_group_front_back2_kernel<2, 512, 16, 64>), dim3(total), dim3(512), lds2, st, gf, gb);
            NR_LAUNCH_CHECK();
        }
        if (v2 && first <= 3 && 3 < last && !fusable) {
            size_t lds = 0;                      // token rows of the largest problem: in LDS when two workgroups still fit a CU
            for (int i = 0; i < n; ++i) {
                size_t need = (size_t)d[i].N * d[i].C * sizeof(float);
                lds = need > lds ? need : lds;
            }
            const int use_lds = lds <= 56 * 1024;
            if (!use_lds) lds = 0;
            if (kv_beside_back) {
                // the kv projections need the front launch alone, like the back half: one grid carries both (nr_linear.h)
                NrLinearArgs p[NR_CTM_MAX_GROUP];
                for (int i = 0; i < n; ++i) {
                    const NrCtmStageDesc& s = d[i];
                    p[i] = NrLinearArgs{w[i].kvn_hi, w[i].kvn_lo, s.wkv_hi, s.wkv_lo, s.kv_bias, nullptr, w[i].kv, s.n_samples * s.N, 2 * s.C, s.C};
                }
                rc = nr_linear_group_launch_beside_back(p, n, gb, lds, use_lds, back_form, st);
                if (rc != NR_OK) return rc;
            } else {
                const void* k = back_form == 4 ? (const void*)nr_group_back2_kernel<512, 4, 1> : (const void*)nr_group_back2_kernel<512, 16, 1>;
                if (lds > 40 * 1024) {
                    hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                    if (e != hipSuccess) return (int)e;
                }
                if (back_form == 4) hipLaunchKernelGGL((nr_group_back2_kernel<512, 4, 1>), dim3(total), dim3(512), lds, st, gb, use_lds);
                else hipLaunchKernelGGL((nr_group_back2_kernel<512, 16, 1>), dim3(total), dim3(512), lds, st, gb, use_lds);
                NR_LAUNCH_CHECK();
            }
        }
        if (!v2 && first <= 2 && 2 < last) {
            bool few = small;                    // a handful of tokens per sample (stage 1 of the step): 256-thread workgroups
            for (int i = 0; i < n; ++i) few = few && d[i].N <= 8 && d[i].cnum * (d[i].C / 128) <= 4 * BK_MAXJ;
            if (fusable && few) {
                hipLaunchKernelGGL((nr_group_front_back_kernel<8, 256>), dim3(total), dim3(256), front_lds, st, gf, gb);
            } else if (fusable && small) {
                const void* k = (const void*)nr_group_front_back_kernel<8>;
                if (front_lds > 64 * 1024) {
                    hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)front_lds);
                    if (e != hipSuccess) return (int)e;
                }
                hipLaunchKernelGGL(nr_group_front_back_kernel<8>, dim3(total), dim3(CF_THREADS), front_lds, st, gf, gb);
            } else {
                // C > 512: sixteen channels per lane of seven per-channel vectors do not fit the 128 registers a 1024-thread
                // workgroup leaves a lane (48 spilled registers, round 3) -- 512-thread workgroups, and never the fused
                // front + back form (the back launch below runs on its own)
                if (front_lds > 64 * 1024) {
                    hipError_t e = hipFuncSetAttribute(small ? (const void*)nr_group_front_kernel<8> : (const void*)nr_group_front_kernel<CF_MAX_CPL, 512>,
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)front_lds);
                    if (e != hipSuccess) return (int)e;
                }
                if (small) hipLaunchKernelGGL(nr_group_front_kernel<8>, dim3(total), dim3(CF_THREADS), front_lds, st, gf);
                else hipLaunchKernelGGL((nr_group_front_kernel<CF_MAX_CPL, 512>), dim3(total), dim3(512), front_lds, st, gf);
            }
            NR_LAUNCH_CHECK();
        }
        if (!v2 && first <= 3 && 3 < last && !(fusable && small)) {
            size_t lds = 0;                      // token rows of the largest problem, if every problem's rows fit
            bool fits = true;
            for (int i = 0; i < n; ++i) {
                size_t need = (size_t)d[i].N * d[i].C * sizeof(float);
                lds = need > lds ? need : lds;
                fits = fits && (d[i].N * d[i].C) % 256 == 0;
            }
            const int use_lds = fits && lds <= 96 * 1024;
            if (!use_lds) lds = 0;
            if (kv_beside_back) {
                // the kv projections need the front launch alone, like the back half: one grid carries both (nr_linear.h)
                NrLinearArgs p[NR_CTM_MAX_GROUP];
                for (int i = 0; i < n; ++i) {
                    const NrCtmStageDesc& s = d[i];
                    p[i] = NrLinearArgs{w[i].kvn_hi, w[i].kvn_lo, s.wkv_hi, s.wkv_lo, s.kv_bias, nullptr, w[i].kv, s.n_samples * s.N, 2 * s.C, s.C};
                }
                rc = nr_linear_group_launch_beside_back(p, n, gb, lds, use_lds, 0, st);
                if (rc != NR_OK) return rc;
            } else {
            if (lds > 40 * 1024) {
                hipError_t e = hipFuncSetAttribute((const void*)nr_group_back_kernel<BK_THREADS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
#ifdef NR_TUNE
                if (e == hipSuccess) e = hipFuncSetAttribute((const void*)nr_group_back_kernel<512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
#endif
                if (e != hipSuccess) return (int)e;
            }
#ifdef NR_TUNE
            const char* te = nr_tune_env("NR_CTM_THREADS");
            if (te && atoi(te) == 512) hipLaunchKernelGGL(nr_group_back_kernel<512>, dim3(total), dim3(512), lds, st, gb, use_lds);
            else
#endif
            hipLaunchKernelGGL(nr_group_back_kernel<BK_THREADS>, dim3(total), dim3(BK_THREADS), lds, st, gb, use_lds);
            NR_LAUNCH_CHECK();
            }
        }
    }
    // 5. q = norm1(merged) Wq^T (+b), kv = norm1(xn) Wkv^T (+b): 2n problems, one launch
    if (first <= 4 && 4 < last) {
        NrLinearArgs p[2 * NR_CTM_MAX_GROUP];
        int np = 0;
        for (int i = 0; i < n && !kv_beside_back; ++i) {
            const NrCtmStageDesc& s = d[i];
            p[np++] = NrLinearArgs{w[i].kvn_hi, w[i].kvn_lo, s.wkv_hi, s.wkv_lo, s.kv_bias, nullptr, w[i].kv, s.n_samples * s.N, 2 * s.C, s.C};
        }
        for (int i = 0; i < n; ++i) {
            const NrCtmStageDesc& s = d[i];
            p[np++] = NrLinearArgs{w[i].qn_hi, w[i].qn_lo, s.wq_hi, s.wq_lo, s.q_bias, nullptr, w[i].q, s.n_samples * s.cnum, s.C, s.C};
        }
        if ((rc = nr_linear_group_launch(p, np, st)) != NR_OK) return rc;
    }
    // 6. score-biased attention of the merged tokens over the un-merged ones
    if (first <= 5 && 5 < last) {
        NrGroupOf<NrAttnArgs> g;
        g.n = n;
        int total = 0;
        for (int i = 0; i < n; ++i) {
            const NrCtmStageDesc& s = d[i];
            g.p[i] = NrAttnArgs{w[i].q, w[i].kv, w[i].score, s.N, s.C, s.cnum, s.heads, 1.0f / sqrtf(64.0f), nullptr, w[i].att_hi, w[i].att_lo};
            g.start[i] = total;
            total += s.n_samples;
        }
        for (int i = n; i <= NR_CTM_MAX_GROUP; ++i) g.start[i] = total;
        size_t lds = 0;                      // k|v rows of the largest problem, if every problem's rows fit
        for (int i = 0; i < n; ++i) {
            size_t need = (size_t)d[i].N * (2 * d[i].C + 4) * sizeof(float);
            lds = need > lds ? need : lds;
        }
        const int use_lds = lds <= 128 * 1024;
        if (!use_lds) lds = 0;
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute((const void*)nr_group_attention_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return (int)e;
        }
        int jobs_max = 0;                    // (head, query) jobs per sample: one wave each, 8 waves when that covers them
        for (int i = 0; i < n; ++i) jobs_max = d[i].heads * d[i].cnum > jobs_max ? d[i].heads * d[i].cnum : jobs_max;
        // More than a wave's worth of jobs per pair of heads: a 256-thread workgroup per (sample, pair of heads) with only those
        // heads' k | v columns in LDS -- several share a CU (the whole-sample form: one 1024-thread workgroup with up to 128 KB),
        // and 64-token samples get their k | v rows into LDS at all.  NR_ATTN_WHOLE=1 keeps the whole-sample form (A/B hook).
        bool by_heads = jobs_max > 8 && !nr_tune_env("NR_ATTN_WHOLE");
        size_t lds_h = 0;
        for (int i = 0; i < n; ++i) {
            by_heads = by_heads && d[i].heads % NR_ATTN_HG == 0;
            size_t need = (size_t)d[i].N * (2 * 64 * NR_ATTN_HG + 4) * sizeof(float);
            lds_h = need > lds_h ? need : lds_h;
        }
        if (by_heads && lds_h <= 72 * 1024) {
            int blocks = 0;
            for (int i = 0; i < n; ++i) {
                g.start[i] = blocks;
                blocks += d[i].n_samples * (d[i].heads / NR_ATTN_HG);
            }
            for (int i = n; i <= NR_CTM_MAX_GROUP; ++i) g.start[i] = blocks;
            if (lds_h > 64 * 1024) {
                hipError_t e = hipFuncSetAttribute((const void*)nr_group_attention_heads_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_h);
                if (e != hipSuccess) return (int)e;
            }
            hipLaunchKernelGGL(nr_group_attention_heads_kernel, dim3(blocks), dim3(256), lds_h, st, g);
            NR_LAUNCH_CHECK();
        } else {
            const char* te = nr_tune_env("NR_CTM_THREADS");
            hipLaunchKernelGGL(nr_group_attention_kernel, dim3(total), dim3((jobs_max <= 8 || (te && atoi(te) == 512)) ? 512 : 1024), lds, st, g,
                               use_lds);
            NR_LAUNCH_CHECK();
        }
    }
    // 7. out = merged + proj(att) + proj.bias
    if (first <= 6 && 6 < last) {
        NrLinearArgs p[NR_CTM_MAX_GROUP];
        for (int i = 0; i < n; ++i) {
            p[i] = NrLinearArgs{w[i].att_hi, w[i].att_lo, d[i].wp_hi, d[i].wp_lo, nullptr, w[i].merged_pb, d[i].out,
                                d[i].n_samples * d[i].cnum, d[i].C, d[i].C};
            p[i].out_hi = d[i].out_hi;              // optional: the output also as a bf16 pair (the next stage's x_hi / x_lo)
            p[i].out_lo = d[i].out_lo;
        }
        if ((rc = nr_linear_group_launch(p, n, st)) != NR_OK) return rc;
    }
    return NR_OK;
}
