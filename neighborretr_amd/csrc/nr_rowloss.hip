// Row-wise fused losses of the NeighborRetr head, forward and backward.
//
// One wave owns one (row i, direction) of the B x B problem; direction 1 reads the transposed
// matrices, i.e. it is the reference's "v2t" call on S.T / G.T.  Each lane keeps NE = ceil(B/64)
// entries of every row vector in registers; all reductions are wave shuffles.  Per row:
//   centrality  (until_module.py:303-328):  -w_i * log_softmax(S*logit_scale)[i,i]
//   uniform CE  (until_module.py:285-289):  -sum_j tgt[i,j] * log_softmax(G*T)[i,j]
//   KL          (until_module.py:351-357):   sum_j p_ij (log p_ij - log_softmax(G)[i,j]),  p = softmax(S)
//   neighbour   (until_module.py:161-211):  top-K of the row without the diagonal (K rounds of wave
//               arg-max, ties -> lowest column), min/max of S and of the bank centrality over the
//               REST set (neither neighbour nor diagonal), p_j = softmax_N(T*(ns_j - nc_j)), p_i = 1,
//               -sum_{E} p_j log_softmax_E(S)_j / sum p,   E = N u {i}.
// rowloss[dir][term][i] is reduced by nr_loss_finalize (fixed order => deterministic).
#include "nr_common.h"
#include "nr_finalize.h"
#include "../../include/nr_hip.h"

struct NrRowArgs {
    const float *S, *G, *tgt_rows, *tgt_cols, *bank_c0, *bank_c1, *wc_text, *wc_video, *logit_scale;
    int B, K;
    float T;
    // Row-slab form (nr_row_losses_fwd_slab): only rows [row0, row0 + n_rows) of either direction are computed,
    // and S is given as the two slabs a rank owns -- S_cols = S[:, row0 : row0 + n_rows] as a [B, n_rows] matrix
    // (direction 1 reads its row i as column i - row0 of it), S itself = S[row0 : row0 + n_rows, :] as [n_rows, B].
    // Zero / nullptr in the full form.
    const float* S_cols;
    int row0, n_rows;
    // Bank centralities as PARTIAL SUMS (what the fused local_level kernel's row / column-sum modes write): bank_c0 is
    // [n_c0, B], bank_c1 [n_c1, B], c_j = c_scale * sum_p part[p][j] (until_module.py:181) -- the nr_reduce_parts
    // launches of the bank chains disappear.  0: bank_c0 / bank_c1 are the finished [B] vectors.
    int n_c0, n_c1;
    float c_scale;
    // Centrality weights computed HERE (one global token per sample): wc_i = exp(cw_scale * <g_i, mean> / max(|g_i|, 1e-12)),
    // the arithmetic of nr_centrality_pair_kernel (modeling.py:403-430) -- the weights' own launch disappears from the split
    // tail.  nullptr: wc_text / wc_video hold the finished weights.
    const float *cw_g_text, *cw_g_video, *cw_mean_text, *cw_mean_video;
    int cw_d;
    float cw_scale;
};

// Everything the forward and the backward need about one row, recomputed identically in both.
template <int NE>
struct NrRowState {
    float s[NE], g[NE], c[NE], tg[NE];
    bool valid[NE], sel[NE], rest[NE];
    float ls, wci, T;
    float lse_c;             // LSE_j(ls * s_j)
    float lse_s;             // LSE_j(s_j)
    float lse_g;             // LSE_j(g_j)
    float lse_u;             // LSE_j(T * g_j)
    float min_s, max_s, min_c, max_c;
    float lse_e;             // LSE over E of s
    float amax, asum;        // softmax stats of T*adj over the neighbours
    float psum;              // sum of positive weights (incl. the diagonal's 1)
    float s_ii;
    int i, B;

    __device__ __forceinline__ void load(const NrRowArgs& a, int row, int dir, int lane) {
        i = row; B = a.B; T = a.T;
        ls = a.logit_scale[0];
        if (a.cw_g_text) {
            const float* gi = (dir == 0 ? a.cw_g_text : a.cw_g_video) + (size_t)row * a.cw_d;
            const float* mean = dir == 0 ? a.cw_mean_text : a.cw_mean_video;
            float dot = 0.f, ss = 0.f;
            for (int k = lane * 4; k < a.cw_d; k += 256) {
                f32x4_t x = *reinterpret_cast<const f32x4_t*>(gi + k);
                f32x4_t m = *reinterpret_cast<const f32x4_t*>(mean + k);
                dot += x[0] * m[0] + x[1] * m[1] + x[2] * m[2] + x[3] * m[3];
                ss += x[0] * x[0] + x[1] * x[1] + x[2] * x[2] + x[3] * x[3];
            }
            dot = nr_wave_sum(dot);
            ss = nr_wave_sum(ss);
            wci = expf(dot / fmaxf(sqrtf(ss), 1e-12f) * a.cw_scale);
        } else {
            wci = dir == 0 ? a.wc_text[row] : a.wc_video[row];
        }
        const float* cvec = dir == 0 ? a.bank_c0 : a.bank_c1;
        const int n_cp = dir == 0 ? a.n_c0 : a.n_c1;
        const float* tgt = dir == 0 ? a.tgt_rows : a.tgt_cols;
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            int j = e * 64 + lane;
            valid[e] = j < B;
            int jj = valid[e] ? j : 0;
            size_t idx = dir == 0 ? (size_t)row * B + jj : (size_t)jj * B + row;
            if (a.S_cols) s[e] = dir == 0 ? a.S[(size_t)(row - a.row0) * B + jj] : a.S_cols[(size_t)jj * a.n_rows + (row - a.row0)];
            else s[e] = a.S[idx];
            g[e] = a.G[idx];
            if (n_cp > 0) {
                // four independent loads in flight (L2 latency, not bandwidth, is what this costs); fixed order
                float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
                int q = 0;
                for (; q + 3 < n_cp; q += 4) {
                    a0 += cvec[(size_t)q * B + jj];
                    a1 += cvec[(size_t)(q + 1) * B + jj];
                    a2 += cvec[(size_t)(q + 2) * B + jj];
                    a3 += cvec[(size_t)(q + 3) * B + jj];
                }
                for (; q < n_cp; ++q) a0 += cvec[(size_t)q * B + jj];
                c[e] = ((a0 + a1) + (a2 + a3)) * a.c_scale;
            } else {
                c[e] = cvec[jj];
            }
            tg[e] = tgt ? tgt[(size_t)row * B + jj] : 0.f;      // no targets: the uniform term comes from the Sinkhorn kernel
        }
    }

    __device__ __forceinline__ float lse(const float (&x)[NE], float scale) const {
        float m = -INFINITY;
#pragma unroll
        for (int e = 0; e < NE; ++e) if (valid[e]) m = fmaxf(m, x[e] * scale);
        m = nr_wave_max(m);
        float t = 0.f;
#pragma unroll
        for (int e = 0; e < NE; ++e) if (valid[e]) t += expf(x[e] * scale - m);
        t = nr_wave_sum(t);
        return m + logf(t);
    }

    __device__ __forceinline__ void stats(int K, int lane) {
        lse_c = lse(s, ls);
        lse_s = lse(s, 1.0f);
        lse_g = lse(g, 1.0f);
        lse_u = lse(g, T);
        // diagonal value
        float d = 0.f;
#pragma unroll
        for (int e = 0; e < NE; ++e) if (e * 64 + lane == i) d = s[e];
        s_ii = nr_wave_sum(d);
        // ---- top-K neighbours (until_module.py:100-129) ----
        float cand[NE];
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            int j = e * 64 + lane;
            sel[e] = false;
            cand[e] = (valid[e] && j != i) ? s[e] : -INFINITY;
        }
        for (int k = 0; k < K; ++k) {
            float bv = -INFINITY;
            int bi = 0x7fffffff;
#pragma unroll
            for (int e = 0; e < NE; ++e)
                if (cand[e] > bv) { bv = cand[e]; bi = e * 64 + lane; }
            nr_wave_argmax(bv, bi);
#pragma unroll
            for (int e = 0; e < NE; ++e)
                if (e * 64 + lane == bi) { sel[e] = true; cand[e] = -INFINITY; }
        }
        // ---- min / max over the rest set (until_module.py:65-86) ----
        float mns = NR_POS_BIG, mxs = NR_NEG_BIG, mnc = NR_POS_BIG, mxc = NR_NEG_BIG;
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            int j = e * 64 + lane;
            rest[e] = valid[e] && !sel[e] && j != i;
            if (rest[e]) {
                mns = fminf(mns, s[e]); mxs = fmaxf(mxs, s[e]);
                mnc = fminf(mnc, c[e]); mxc = fmaxf(mxc, c[e]);
            }
        }
        min_s = nr_wave_min(mns); max_s = nr_wave_max(mxs);
        min_c = nr_wave_min(mnc); max_c = nr_wave_max(mxc);
        // ---- positive weights: softmax over the neighbours of T*(ns - nc) ----
        const float rs = 1.0f / (max_s - min_s), rc = 1.0f / (max_c - min_c);
        // K == B: the reference takes the first K of the descending sort, whose LAST entry is the diagonal (-9e15): the diagonal
        // is a "neighbour" then and its adjusted similarity sits in the softmax's denominator (until_module.py:119-123, :147), before
        // fill_diagonal_(1) overwrites its weight (:157).  The picks above never take it (-inf): counted here.
        const bool diag_in = K >= B;
        float am = -INFINITY, em = -INFINITY;
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const bool diag = valid[e] && e * 64 + lane == i;
            if (sel[e] || (diag && diag_in)) {
                float adj = (s[e] - min_s) * rs - (c[e] - min_c) * rc;
                am = fmaxf(am, adj * T);
            }
            if (sel[e] || diag) em = fmaxf(em, s[e]);
        }
        amax = nr_wave_max(am);
        em = nr_wave_max(em);
        float as = 0.f, es = 0.f;
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const bool diag = valid[e] && e * 64 + lane == i;
            if (sel[e] || (diag && diag_in)) {
                float adj = (s[e] - min_s) * rs - (c[e] - min_c) * rc;
                as += expf(adj * T - amax);
            }
            if (sel[e] || diag) es += expf(s[e] - em);
        }
        asum = nr_wave_sum(as);
        lse_e = em + logf(nr_wave_sum(es));
    }

    // positive weight of entry e (0 outside the neighbour set; the diagonal is handled separately)
    __device__ __forceinline__ float posw(int e) const {
        if (!sel[e]) return 0.f;
        const float rs = 1.0f / (max_s - min_s), rc = 1.0f / (max_c - min_c);
        float adj = (s[e] - min_s) * rs - (c[e] - min_c) * rc;
        return expf(adj * T - amax) / asum;
    }
};

// Optional tail of the forward launch: the workgroup that finishes LAST reduces the row terms to the five
// losses (same arithmetic and summation order as nr_loss_finalize, so the result is bit-identical) --
// one launch less on the step's critical path.  `counter` is a zero-initialised device word that the
// last workgroup resets, so a captured graph can replay it.
struct NrRowFinal {
    unsigned int* counter;
    float wu, wn, wkl;
    float* losses;
    unsigned int total;        // workgroups that add to `counter` before the last one finalizes; 0 = this launch's grid
};

template <int NE>
__device__ __forceinline__ void nr_row_losses_fwd_rows(const NrRowArgs& a, float* __restrict__ rowloss, const int row, const int dir,
                                                       const int lane) {
    NrRowState<NE> r;
    r.load(a, row, dir, lane);
    r.stats(a.K, lane);

    float l_u = 0.f, l_kl = 0.f, num = 0.f, ps = 0.f;
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        if (!r.valid[e]) continue;
        l_u -= r.tg[e] * (r.g[e] * r.T - r.lse_u);
        float lp = r.s[e] - r.lse_s;
        l_kl += expf(lp) * (lp - (r.g[e] - r.lse_g));
        float p = r.posw(e);
        num += p * (r.s[e] - r.lse_e);
        ps += p;
    }
    l_u = nr_wave_sum(l_u);
    l_kl = nr_wave_sum(l_kl);
    num = nr_wave_sum(num) + (r.s_ii - r.lse_e);   // diagonal weight 1
    ps = nr_wave_sum(ps) + 1.0f;
    if (lane == 0) {
        const int B = a.B;
        // `sc1` stores: in the self-finalizing launches another workgroup reads these without a cache-wide fence (below)
        float* o = rowloss + (size_t)dir * 4 * B;
        __hip_atomic_store(o + 0 * B + row, -(r.s_ii * r.ls - r.lse_c) * r.wci, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (a.tgt_rows) __hip_atomic_store(o + 1 * B + row, l_u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(o + 2 * B + row, -num / ps, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(o + 3 * B + row, l_kl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <int NE>
__global__ __launch_bounds__(256) void nr_row_losses_fwd_kernel(NrRowArgs a, float* __restrict__ rowloss, NrRowFinal f) {
    const int lane = threadIdx.x & 63;
    const int local = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int row = a.row0 + local;
    if (local < (a.S_cols ? a.n_rows : a.B)) nr_row_losses_fwd_rows<NE>(a, rowloss, row, blockIdx.y, lane);
    if (f.counter == nullptr) return;
    // Hand-off by `sc1` stores / loads instead of cache-wide fences (__threadfence() = L2 write-back + invalidate, 3.5-6.5 us
    // per workgroup): every storing wave drains its stores, ONE lane adds to the counter behind the workgroup barrier, the
    // workgroup whose add came last reads (agent-scope loads in nr_loss_finalize_body<true>) behind a second barrier.
    __shared__ int s_last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned int total = f.total ? f.total : gridDim.x * gridDim.y;
        s_last = __hip_atomic_fetch_add(f.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == total - 1;
    }
    __syncthreads();
    if (!s_last) return;
    nr_loss_finalize_body<true>(rowloss, a.B, f.wu, f.wn, f.wkl, f.losses);
    if (threadIdx.x == 0) __hip_atomic_store(f.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(256) void nr_loss_finalize_kernel(const float* __restrict__ rowloss, int B, float wu, float wn,
                                                               float wkl, float* __restrict__ losses) {
    nr_loss_finalize_body<false>(rowloss, B, wu, wn, wkl, losses);
}

static int nr_row_ne(int B) {
    int ne = (B + 63) / 64;
    if (ne <= 2) return 2;
    if (ne <= 4) return 4;
    if (ne <= 8) return 8;
    if (ne <= 16) return 16;
    if (ne <= 32) return 32;
    return 0;
}

static int nr_row_losses_fwd_launch(const NrRowArgs& a, float* rowloss, const NrRowFinal& f, hipStream_t st) {
    dim3 grid(((a.S_cols ? a.n_rows : a.B) + 3) / 4, 2);
    switch (nr_row_ne(a.B)) {
        case 2: hipLaunchKernelGGL(nr_row_losses_fwd_kernel<2>, grid, dim3(256), 0, st, a, rowloss, f); break;
        case 4: hipLaunchKernelGGL(nr_row_losses_fwd_kernel<4>, grid, dim3(256), 0, st, a, rowloss, f); break;
        case 8: hipLaunchKernelGGL(nr_row_losses_fwd_kernel<8>, grid, dim3(256), 0, st, a, rowloss, f); break;
        case 16: hipLaunchKernelGGL(nr_row_losses_fwd_kernel<16>, grid, dim3(256), 0, st, a, rowloss, f); break;
        case 32: hipLaunchKernelGGL(nr_row_losses_fwd_kernel<32>, grid, dim3(256), 0, st, a, rowloss, f); break;
        default: return NR_EUNSUPPORTED;   // B > 2048
    }
    NR_LAUNCH_CHECK();
    return NR_OK;
}

extern "C" int nr_row_losses_fwd(const float* S, const float* G, const float* tgt_rows, const float* tgt_cols,
                                 const float* bank_c0, const float* bank_c1, const float* wc_text, const float* wc_video,
                                 const float* logit_scale, int B, int K, float temperature, float* rowloss, void* stream) {
    if (!S || !G || !tgt_rows || !tgt_cols || !bank_c0 || !bank_c1 || !wc_text || !wc_video || !logit_scale || !rowloss)
        return NR_EINVAL;
    if (B <= 0 || K < 0 || K > B) return NR_EINVAL;    // the reference raises IndexError for K > B
    NrRowArgs a{S, G, tgt_rows, tgt_cols, bank_c0, bank_c1, wc_text, wc_video, logit_scale, B, K, temperature, nullptr, 0, 0, 0, 0, 0.f, nullptr, nullptr, nullptr, nullptr, 0, 0.f};
    return nr_row_losses_fwd_launch(a, rowloss, NrRowFinal{nullptr, 0.f, 0.f, 0.f, nullptr, 0u}, (hipStream_t)stream);
}

// centrality, neighbour and KL terms only (rowloss[dir][0,2,3][i]); the uniform term rowloss[dir][1][i] is left to
// nr_sinkhorn_uniform_rows, so that this launch does not wait for the Sinkhorn solve
extern "C" int nr_row_losses_fwd_no_uniform(const float* S, const float* G, const float* bank_c0, const float* bank_c1,
                                            const float* wc_text, const float* wc_video, const float* logit_scale, int B, int K,
                                            float temperature, float* rowloss, void* stream) {
    if (!S || !G || !bank_c0 || !bank_c1 || !wc_text || !wc_video || !logit_scale || !rowloss) return NR_EINVAL;
    if (B <= 0 || K < 0 || K > B) return NR_EINVAL;
    NrRowArgs a{S, G, nullptr, nullptr, bank_c0, bank_c1, wc_text, wc_video, logit_scale, B, K, temperature, nullptr, 0, 0, 0, 0, 0.f, nullptr, nullptr, nullptr, nullptr, 0, 0.f};
    return nr_row_losses_fwd_launch(a, rowloss, NrRowFinal{nullptr, 0.f, 0.f, 0.f, nullptr, 0u}, (hipStream_t)stream);
}

// The split tail's row-loss launch: centrality / neighbour / KL terms from the bank products' PARTIAL sums, and the
// five losses from whichever workgroup -- of this launch or of the concurrent nr_sinkhorn_uniform_rows_final --
// finishes last (`counter`: one zero-initialised device word shared by both launches, reset by that workgroup).
extern "C" int nr_split_tail_workgroups(int B) { return B > 0 ? 2 + 2 * ((B + 3) / 4) : 0; }

extern "C" int nr_row_losses_fwd_no_uniform_final(const float* S, const float* G, const float* c0_parts, int n_c0,
                                                  const float* c1_parts, int n_c1, float c_scale, const float* wc_text,
                                                  const float* wc_video, const float* logit_scale, int B, int K,
                                                  float temperature, float* rowloss, uint32_t* counter, float uniform_weight,
                                                  float neighbor_weight, float kl_weight, float* losses, void* stream) {
    if (!S || !G || !c0_parts || !c1_parts || !wc_text || !wc_video || !logit_scale || !rowloss || !counter || !losses)
        return NR_EINVAL;
    if (B <= 0 || K < 0 || K > B || n_c0 <= 0 || n_c1 <= 0) return NR_EINVAL;
    NrRowArgs a{S, G, nullptr, nullptr, c0_parts, c1_parts, wc_text, wc_video, logit_scale, B, K, temperature, nullptr, 0, 0,
                n_c0, n_c1, c_scale, nullptr, nullptr, nullptr, nullptr, 0, 0.f};
    return nr_row_losses_fwd_launch(a, rowloss, NrRowFinal{counter, uniform_weight, neighbor_weight, kl_weight, losses,
                                                           (unsigned)nr_split_tail_workgroups(B)}, (hipStream_t)stream);
}

// The same launch computing the centrality weights itself (see NrRowArgs::cw_*): g_text / g_video [B, d] the samples' ONE
// global token, mean_text / mean_video [d] the means of the normalised batch tokens.
extern "C" int nr_row_losses_fwd_no_uniform_final_cw(const float* S, const float* G, const float* c0_parts, int n_c0,
                                                     const float* c1_parts, int n_c1, float c_scale, const float* g_text,
                                                     const float* g_video, const float* mean_text, const float* mean_video, int d,
                                                     float centrality_scale, const float* logit_scale, int B, int K,
                                                     float temperature, float* rowloss, uint32_t* counter, float uniform_weight,
                                                     float neighbor_weight, float kl_weight, float* losses, void* stream) {
    if (!S || !G || !c0_parts || !c1_parts || !g_text || !g_video || !mean_text || !mean_video || !logit_scale || !rowloss || !counter || !losses)
        return NR_EINVAL;
    if (B <= 0 || K < 0 || K > B || n_c0 <= 0 || n_c1 <= 0 || d <= 0 || (d % 4) != 0) return NR_EINVAL;
    NrRowArgs a{S, G, nullptr, nullptr, c0_parts, c1_parts, nullptr, nullptr, logit_scale, B, K, temperature, nullptr, 0, 0,
                n_c0, n_c1, c_scale, g_text, g_video, mean_text, mean_video, d, centrality_scale};
    return nr_row_losses_fwd_launch(a, rowloss, NrRowFinal{counter, uniform_weight, neighbor_weight, kl_weight, losses,
                                                           (unsigned)nr_split_tail_workgroups(B)}, (hipStream_t)stream);
}

extern "C" int nr_row_losses_fwd_slab(const float* S_rows, const float* S_cols, int row0, int n_rows, const float* G,
                                      const float* tgt_rows, const float* tgt_cols, const float* bank_c0, const float* bank_c1,
                                      const float* wc_text, const float* wc_video, const float* logit_scale, int B, int K,
                                      float temperature, float* rowloss, void* stream) {
    if (!S_rows || !S_cols || !G || !tgt_rows || !tgt_cols || !bank_c0 || !bank_c1 || !wc_text || !wc_video || !logit_scale || !rowloss)
        return NR_EINVAL;
    if (B <= 0 || K < 0 || K > B || row0 < 0 || n_rows <= 0 || row0 + n_rows > B) return NR_EINVAL;
    NrRowArgs a{S_rows, G, tgt_rows, tgt_cols, bank_c0, bank_c1, wc_text, wc_video, logit_scale, B, K, temperature, S_cols, row0, n_rows, 0, 0, 0.f, nullptr, nullptr, nullptr, nullptr, 0, 0.f};
    return nr_row_losses_fwd_launch(a, rowloss, NrRowFinal{nullptr, 0.f, 0.f, 0.f, nullptr, 0u}, (hipStream_t)stream);
}

extern "C" int nr_row_losses_fwd_final(const float* S, const float* G, const float* tgt_rows, const float* tgt_cols,
                                       const float* bank_c0, const float* bank_c1, const float* wc_text,
                                       const float* wc_video, const float* logit_scale, int B, int K, float temperature,
                                       float* rowloss, uint32_t* counter, float uniform_weight, float neighbor_weight,
                                       float kl_weight, float* losses, void* stream) {
    if (!S || !G || !tgt_rows || !tgt_cols || !bank_c0 || !bank_c1 || !wc_text || !wc_video || !logit_scale || !rowloss ||
        !counter || !losses)
        return NR_EINVAL;
    if (B <= 0 || K < 0 || K > B) return NR_EINVAL;
    NrRowArgs a{S, G, tgt_rows, tgt_cols, bank_c0, bank_c1, wc_text, wc_video, logit_scale, B, K, temperature, nullptr, 0, 0, 0, 0, 0.f, nullptr, nullptr, nullptr, nullptr, 0, 0.f};
    return nr_row_losses_fwd_launch(a, rowloss, NrRowFinal{counter, uniform_weight, neighbor_weight, kl_weight, losses, 0u},
                                    (hipStream_t)stream);
}

extern "C" int nr_loss_finalize(const float* rowloss, int B, float uniform_weight, float neighbor_weight, float kl_weight,
                                float* losses, void* stream) {
    if (!rowloss || !losses || B <= 0) return NR_EINVAL;
    hipLaunchKernelGGL(nr_loss_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, rowloss, B, uniform_weight,
                       neighbor_weight, kl_weight, losses);
    NR_LAUNCH_CHECK();
    return NR_OK;
}

// ================================= backward ======================================================
template <int NE>
__global__ __launch_bounds__(256) void nr_row_losses_bwd_kernel(NrRowArgs a, const float* __restrict__ g_rowloss,
                                                                float* __restrict__ dS_dir, float* __restrict__ dG_dir,
                                                                float* __restrict__ d_c_rows, float* __restrict__ d_wc,
                                                                float* __restrict__ d_ls_rows) {
    const int lane = threadIdx.x & 63;
    const int local = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int row = a.row0 + local;                 // slab form: rows [row0, row0 + n_rows) only (row0 = 0 in the full form)
    const int dir = blockIdx.y;
    const int B = a.B;
    const int n_out = a.S_cols ? a.n_rows : B;      // rows of every output matrix
    if (local >= n_out) return;
    NrRowState<NE> r;
    r.load(a, row, dir, lane);
    r.stats(a.K, lane);
    const float gC = g_rowloss[(size_t)(dir * 4 + 0) * B + row];
    const float gU = g_rowloss[(size_t)(dir * 4 + 1) * B + row];
    const float gN = g_rowloss[(size_t)(dir * 4 + 2) * B + row];
    const float gK = g_rowloss[(size_t)(dir * 4 + 3) * B + row];

    // ---- pass 1: row scalars -------------------------------------------------------------------
    float tsum = 0.f, kl = 0.f, pcs = 0.f, pe = 0.f, psum = 0.f;
    float mns = INFINITY, mxs = -INFINITY, mnc = INFINITY, mxc = -INFINITY;
    int imns = 0x7fffffff, imxs = 0x7fffffff, imnc = 0x7fffffff, imxc = 0x7fffffff;
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        if (!r.valid[e]) continue;
        const int j = e * 64 + lane;
        tsum += r.tg[e];
        float lp = r.s[e] - r.lse_s;
        kl += expf(lp) * (lp - (r.g[e] - r.lse_g));
        pcs += expf(r.s[e] * r.ls - r.lse_c) * r.s[e];
        float p = r.posw(e);
        pe += p * (r.s[e] - r.lse_e);
        psum += p;
        if (r.rest[e]) {
            if (r.s[e] < mns) { mns = r.s[e]; imns = j; }
            if (r.s[e] > mxs) { mxs = r.s[e]; imxs = j; }
            if (r.c[e] < mnc) { mnc = r.c[e]; imnc = j; }
            if (r.c[e] > mxc) { mxc = r.c[e]; imxc = j; }
        }
    }
    tsum = nr_wave_sum(tsum);
    kl = nr_wave_sum(kl);
    pcs = nr_wave_sum(pcs);
    pe = nr_wave_sum(pe);
    psum = nr_wave_sum(psum) + 1.0f;
    nr_wave_argmin(mns, imns);
    nr_wave_argmin(mnc, imnc);
    nr_wave_argmax(mxs, imxs);
    nr_wave_argmax(mxc, imxc);

    const float rs = 1.0f / (r.max_s - r.min_s), rc = 1.0f / (r.max_c - r.min_c);
    // neighbour: dz_j = p_j * (-1/ps) * (e_j - pe);  dadj_j = T * dz_j
    float Dmin_s = 0.f, Dmax_s = 0.f, Dmin_c = 0.f, Dmax_c = 0.f;
    float dadj[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        dadj[e] = 0.f;
        if (r.valid[e] && r.sel[e]) {
            float p = r.posw(e);
            float ej = r.s[e] - r.lse_e;
            dadj[e] = r.T * p * (-1.0f / psum) * (ej - pe);
            float ns = (r.s[e] - r.min_s) * rs, nc = (r.c[e] - r.min_c) * rc;
            Dmin_s += dadj[e] * rs * (ns - 1.0f);
            Dmax_s -= dadj[e] * ns * rs;
            Dmin_c += dadj[e] * rc * (1.0f - nc);
            Dmax_c += dadj[e] * nc * rc;
        }
    }
    Dmin_s = nr_wave_sum(Dmin_s); Dmax_s = nr_wave_sum(Dmax_s);
    Dmin_c = nr_wave_sum(Dmin_c); Dmax_c = nr_wave_sum(Dmax_c);

    // ---- pass 2: per-entry gradients --------------------------------------------------------------
    float* dSr = dS_dir + ((size_t)dir * n_out + local) * B;
    float* dGr = dG_dir + ((size_t)dir * n_out + local) * B;
    float* dCr = d_c_rows + ((size_t)dir * n_out + local) * B;
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        if (!r.valid[e]) continue;
        const int j = e * 64 + lane;
        const bool diag = j == row;
        // centrality
        float pc = expf(r.s[e] * r.ls - r.lse_c);
        float ds = gC * (-r.wci * r.ls) * ((diag ? 1.0f : 0.f) - pc);
        // uniform CE
        float pu = expf(r.g[e] * r.T - r.lse_u);
        float dg = gU * (-r.T) * (r.tg[e] - tsum * pu);
        // KL
        float lp = r.s[e] - r.lse_s, p = expf(lp);
        float qg = expf(r.g[e] - r.lse_g);
        ds += gK * p * ((lp - (r.g[e] - r.lse_g)) - kl);
        dg += gK * (qg - p);
        // neighbour
        float dn = 0.f, dc = 0.f;
        if (r.sel[e] || diag) {
            float wk = diag ? 1.0f : r.posw(e);
            dn += expf(r.s[e] - r.lse_e) - wk / psum;
        }
        if (r.sel[e]) {
            dn += rs * dadj[e];
            dc -= rc * dadj[e];
        }
        if (j == imns) dn += Dmin_s;
        if (j == imxs) dn += Dmax_s;
        if (j == imnc) dc += Dmin_c;
        if (j == imxc) dc += Dmax_c;
        ds += gN * dn;
        dSr[j] = ds;
        dGr[j] = dg;
        dCr[j] = gN * dc;
    }
    if (lane == 0) {
        d_wc[(size_t)dir * n_out + local] = gC * -(r.s_ii * r.ls - r.lse_c);
        d_ls_rows[(size_t)dir * n_out + local] = gC * -r.wci * (r.s_ii - pcs);
    }
}

// g_rowloss [2,4,B] of the fused objective from the gradients of its five outputs (total, centrality, uniform, neighbour, kl;
// any of them may be absent): the constants of nr_loss_finalize differentiated --
//   coef[term] = (g_total * weight[term] + g_term) * 0.5 / B,  the kl term divided by B once more (batchmean).
__global__ __launch_bounds__(256) void nr_rowloss_coef_kernel(const float* g0, const float* g1, const float* g2, const float* g3,
                                                              const float* g4, float wu, float wn, float wkl, int B,
                                                              float* __restrict__ coef) {
    const float t = g0 ? g0[0] : 0.f;
    const float h = 0.5f / (float)B;
    const float c[4] = {(t + (g1 ? g1[0] : 0.f)) * h, (t * wu + (g2 ? g2[0] : 0.f)) * h, (t * wn + (g3 ? g3[0] : 0.f)) * h,
                        (t * wkl + (g4 ? g4[0] : 0.f)) / (float)B * h};
    for (int i = blockIdx.x * 256 + threadIdx.x; i < 8 * B; i += gridDim.x * 256) coef[i] = c[(i / B) & 3];
}

extern "C" int nr_rowloss_coef(const float* g_total, const float* g_centrality, const float* g_uniform, const float* g_neighbor,
                               const float* g_kl, float uniform_weight, float neighbor_weight, float kl_weight, int B,
                               float* g_rowloss, void* stream) {
    if (!g_rowloss || B <= 0) return NR_EINVAL;
    hipLaunchKernelGGL(nr_rowloss_coef_kernel, dim3((8 * B + 255) / 256 > 64 ? 64 : (8 * B + 255) / 256), dim3(256), 0,
                       (hipStream_t)stream, g_total, g_centrality, g_uniform, g_neighbor, g_kl, uniform_weight, neighbor_weight,
                       kl_weight, B, g_rowloss);
    NR_LAUNCH_CHECK();
    return NR_OK;
}

extern "C" int nr_row_losses_bwd(const float* S, const float* G, const float* tgt_rows, const float* tgt_cols,
                                 const float* bank_c0, const float* bank_c1, const float* wc_text, const float* wc_video,
                                 const float* logit_scale, int B, int K, float temperature, const float* g_rowloss,
                                 float* dS_dir, float* dG_dir, float* d_c_rows, float* d_wc, float* d_ls_rows,
                                 void* stream) {
    if (!S || !G || !tgt_rows || !tgt_cols || !bank_c0 || !bank_c1 || !wc_text || !wc_video || !logit_scale || !g_rowloss ||
        !dS_dir || !dG_dir || !d_c_rows || !d_wc || !d_ls_rows)
        return NR_EINVAL;
    if (B <= 0 || K < 0 || K > B) return NR_EINVAL;
    NrRowArgs a{S, G, tgt_rows, tgt_cols, bank_c0, bank_c1, wc_text, wc_video, logit_scale, B, K, temperature, nullptr, 0, 0, 0, 0, 0.f, nullptr, nullptr, nullptr, nullptr, 0, 0.f};
    dim3 grid((B + 3) / 4, 2);
    hipStream_t st = (hipStream_t)stream;
#define NR_BWD_CASE(N_) \
    case N_: hipLaunchKernelGGL(nr_row_losses_bwd_kernel<N_>, grid, dim3(256), 0, st, a, g_rowloss, dS_dir, dG_dir, d_c_rows, d_wc, d_ls_rows); break;
    switch (nr_row_ne(B)) {
        NR_BWD_CASE(2) NR_BWD_CASE(4) NR_BWD_CASE(8) NR_BWD_CASE(16) NR_BWD_CASE(32)
        default: return NR_EUNSUPPORTED;
    }
#undef NR_BWD_CASE
    NR_LAUNCH_CHECK();
    return NR_OK;
}

// Backward of nr_row_losses_fwd_slab: the rows [row0, row0 + n_rows) a rank owns, from the two slabs of S it holds.  Outputs
// are slab-shaped, one row per owned row and direction: dS_dir [2, n_rows, B] (direction 0: d S[row0 + k, :]; direction 1:
// d S[:, row0 + k] -- the caller transposes), dG_dir and d_c_rows likewise, d_wc / d_ls_rows [2, n_rows].
extern "C" int nr_row_losses_bwd_slab(const float* S_rows, const float* S_cols, int row0, int n_rows, const float* G,
                                      const float* tgt_rows, const float* tgt_cols, const float* bank_c0, const float* bank_c1,
                                      const float* wc_text, const float* wc_video, const float* logit_scale, int B, int K,
                                      float temperature, const float* g_rowloss, float* dS_dir, float* dG_dir, float* d_c_rows,
                                      float* d_wc, float* d_ls_rows, void* stream) {
    if (!S_rows || !S_cols || !G || !tgt_rows || !tgt_cols || !bank_c0 || !bank_c1 || !wc_text || !wc_video || !logit_scale ||
        !g_rowloss || !dS_dir || !dG_dir || !d_c_rows || !d_wc || !d_ls_rows)
        return NR_EINVAL;
    if (B <= 0 || K < 0 || K > B || row0 < 0 || n_rows <= 0 || row0 + n_rows > B) return NR_EINVAL;
    NrRowArgs a{S_rows, G, tgt_rows, tgt_cols, bank_c0, bank_c1, wc_text, wc_video, logit_scale, B, K, temperature, S_cols, row0, n_rows, 0, 0, 0.f, nullptr, nullptr, nullptr, nullptr, 0, 0.f};
    dim3 grid((n_rows + 3) / 4, 2);
    hipStream_t st = (hipStream_t)stream;
#define NR_BWD_CASE(N_) \
    case N_: hipLaunchKernelGGL(nr_row_losses_bwd_kernel<N_>, grid, dim3(256), 0, st, a, g_rowloss, dS_dir, dG_dir, d_c_rows, d_wc, d_ls_rows); break;
    switch (nr_row_ne(B)) {
        NR_BWD_CASE(2) NR_BWD_CASE(4) NR_BWD_CASE(8) NR_BWD_CASE(16) NR_BWD_CASE(32)
        default: return NR_EUNSUPPORTED;
    }
#undef NR_BWD_CASE
    NR_LAUNCH_CHECK();
    return NR_OK;
}
