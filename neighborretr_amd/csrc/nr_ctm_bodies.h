// Device bodies of the clustering-stage kernels, shared by the one-problem launches (nr_ctm_fused.hip,
// nr_ctm.hip) and the grouped launches that run the text and the video problem of a stage in the same
// grid (nr_ctm_group.hip).  Reference: cluster.py:453-561 (DPC-KNN, merge_tokens), :689-717 (CTM.forward),
// :834-888 (score-biased attention).
#pragma once
#include <type_traits>
#include "nr_common.h"
#include "../../include/nr_hip.h"

// Arguments of a grouped launch: the workgroups of up to NR_CTM_MAX_GROUP independent problems in one grid
// (workgroup -> problem through the prefix table).
template <typename A>
struct NrGroupOf {
    A p[NR_CTM_MAX_GROUP];
    int start[NR_CTM_MAX_GROUP + 1];     // first workgroup of every problem
    int n;
    __device__ __forceinline__ int find(int wg) const {
        int g = 0;
#pragma unroll
        for (int i = 1; i < NR_CTM_MAX_GROUP; ++i)
            if (i < n && wg >= start[i]) g = i;
        return g;
    }
};

#define CF_THREADS 1024
#define CF_MAX_CPL 16      // C <= 1024

struct NrCtmFrontArgs {
    const float *y, *mask, *ln_w, *ln_b, *sc_w, *sc_b, *n1_w, *n1_b;
    float eps, inv_sqrt_c;
    int N, C;
    float *xn, *kvn, *score, *tokw, *dist, *smax;
    uint16_t *kvn_hi, *kvn_lo;       // when set, norm1(xn) is written split-bf16 (operand of the kv GEMM) instead of f32
};

// b = sample index; sx = N*C floats of (dynamic) LDS.  Called by all CF_THREADS threads of the workgroup.
// CPL = channels per lane the registers are sized for (C <= 64*CPL).
// Phase 1 is one global round trip per row: the five per-channel parameter vectors are loaded once up front,
// and the NEXT row of a wave is loaded before the current row's results are stored (vmcnt retires in order:
// a load issued after stores waits for their acknowledgement -- with the loads placed between the LayerNorm
// and norm1 halves of a row that was three dependent round trips per row).
// THREADS: the workgroup size the body is launched with (stages with a handful of tokens per sample -- stage 1 of the step:
// 4 and 3 -- run 256-thread workgroups: a 1024-thread workgroup needs a CU of its own and waits for one while the MFMA
// kernels of the other branch hold them, a small one slots in beside them).
#ifdef NR_STAMP
static __device__ unsigned long long nr_front_stamps[16];
#endif
template <int CPL, int THREADS = CF_THREADS>
__device__ __forceinline__ void nr_ctm_front_body(const NrCtmFrontArgs& p, const int b, float* sx) {
    __shared__ float s_wmax[THREADS / 64];
#ifdef NR_STAMP
    unsigned long long fk[8];
    int fn = 0;
    fk[fn++] = __builtin_readcyclecounter();
#define CF_LAP() fk[fn++] = __builtin_readcyclecounter()
#else
#define CF_LAP() ((void)0)
#endif
    const int N = p.N, C = p.C;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int NW = THREADS / 64;
    const int cpl = C / 64;
    const float* __restrict__ g_y = p.y;
    float* __restrict__ g_xn = p.xn;
    // ---- phase 1: one wave per token row ------------------------------------------------------------
    float lnw[CPL], lnb[CPL], scw[CPL], n1w[CPL], n1b[CPL], v[CPL], vnext[CPL];
    if (wave < N) {
#pragma unroll
        for (int q = 0; q < CPL; ++q) {
            const int c = q * 64 + lane;
            const bool on = q < cpl;
            v[q] = on ? g_y[((size_t)b * N + wave) * C + c] : 0.f;
            lnw[q] = on ? p.ln_w[c] : 0.f;
            lnb[q] = on ? p.ln_b[c] : 0.f;
            scw[q] = on ? p.sc_w[c] : 0.f;
            n1w[q] = on ? p.n1_w[c] : 0.f;
            n1b[q] = on ? p.n1_b[c] : 0.f;
        }
    }
    const float scb = p.sc_b[0];
    for (int r = wave; r < N; r += NW) {
        const size_t row = (size_t)b * N + r;
        const bool more = r + NW < N;
        const float mk = p.mask ? p.mask[row] : 1.f;
#pragma unroll
        for (int q = 0; q < CPL; ++q) vnext[q] = (more && q < cpl) ? g_y[(row + NW) * C + q * 64 + lane] : 0.f;
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < CPL; ++q) s += v[q];
        const float mu = nr_wave_sum(s) / (float)C;
        float var = 0.f;
#pragma unroll
        for (int q = 0; q < CPL; ++q)
            if (q < cpl) { float dlt = v[q] - mu; var += dlt * dlt; }
        const float rstd = rsqrtf(nr_wave_sum(var) / (float)C + p.eps);
        float dot = 0.f, s2 = 0.f;
#pragma unroll
        for (int q = 0; q < CPL; ++q)
            if (q < cpl) {
                int c = q * 64 + lane;
                v[q] = (v[q] - mu) * rstd * lnw[q] + lnb[q];
                g_xn[row * C + c] = v[q];
                sx[r * C + c] = v[q];
                dot += v[q] * scw[q];
                s2 += v[q];
            }
        float sc = nr_wave_sum(dot) + scb;
        if (p.mask && mk == 0.f) sc = -INFINITY;
        if (lane == 0) {
            p.score[row] = sc;
            p.tokw[row] = expf(sc);
        }
        const float mu2 = nr_wave_sum(s2) / (float)C;
        float var2 = 0.f;
#pragma unroll
        for (int q = 0; q < CPL; ++q)
            if (q < cpl) { float dlt = v[q] - mu2; var2 += dlt * dlt; }
        const float rstd2 = rsqrtf(nr_wave_sum(var2) / (float)C + p.eps);
#pragma unroll
        for (int q = 0; q < CPL; ++q)
            if (q < cpl) {
                int c = q * 64 + lane;
                float kv = (v[q] - mu2) * rstd2 * n1w[q] + n1b[q];
                if (p.kvn_hi) {
                    uint16_t h = nr_f2bf(kv);
                    p.kvn_hi[row * C + c] = h;
                    p.kvn_lo[row * C + c] = nr_f2bf(kv - nr_bf2f(h));
                } else {
                    p.kvn[row * C + c] = kv;
                }
            }
#pragma unroll
        for (int q = 0; q < CPL; ++q) v[q] = vnext[q];
    }
    CF_LAP();      // 1: this wave's rows normalised and stored (issue side)
    __syncthreads();
    CF_LAP();      // 2: every wave's rows
    // ---- phase 2: pairwise distances, wave per row of the upper triangle -------------------------------
    float wmax = 0.f;
    float* db = p.dist + (size_t)b * N * N;
    for (int i = wave; i < N; i += NW) {
        float xi[CPL];
#pragma unroll
        for (int q = 0; q < CPL; ++q) xi[q] = q < cpl ? sx[i * C + q * 64 + lane] : 0.f;
        if (lane == 0) db[i * N + i] = 0.f;
        for (int j = i + 1; j < N; j += 2) {
            const bool two = j + 1 < N;
            const float* xj0 = sx + j * C;
            const float* xj1 = sx + (two ? j + 1 : j) * C;
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int q = 0; q < CPL; ++q)
                if (q < cpl) {
                    int c = q * 64 + lane;
                    float d0 = xi[q] - xj0[c], d1 = xi[q] - xj1[c];
                    s0 += d0 * d0;
                    s1 += d1 * d1;
                }
            s0 = nr_wave_sum(s0);
            s1 = nr_wave_sum(s1);
            float dv0 = sqrtf(s0) * p.inv_sqrt_c, dv1 = sqrtf(s1) * p.inv_sqrt_c;
            wmax = fmaxf(wmax, dv0);
            if (two) wmax = fmaxf(wmax, dv1);
            if (lane == 0) {
                db[i * N + j] = dv0;
                db[j * N + i] = dv0;
                if (two) {
                    db[i * N + j + 1] = dv1;
                    db[(j + 1) * N + i] = dv1;
                }
            }
        }
    }
    if (lane == 0) s_wmax[wave] = wmax;
    CF_LAP();      // 3: this wave's distances
    __syncthreads();
    if (tid == 0) {
        float m = 0.f;
        for (int w = 0; w < NW; ++w) m = fmaxf(m, s_wmax[w]);
        p.smax[b] = m;
    }
    CF_LAP();      // 4: end
#ifdef NR_STAMP
    if (b == 0 && tid == 0)
        for (int i = 0; i < fn; ++i) nr_front_stamps[i] = fk[i] - fk[0];
#endif
}


// ---- assignment + merge -------------------------------------------------------------------------------------
struct NrCtmBackArgs {
    const float *dist, *smax, *mask, *noise, *xn, *tokw, *n1_w, *n1_b, *proj_b;
    int n_samples, N, C, k, cnum;
    float eps;
    float *merged, *merged_pb, *qn;
    int64_t* assign;
    uint16_t *qn_hi, *qn_lo;         // when set, norm1(merged) is written split-bf16 instead of f32
};

// b = sample index.  Called by all 256 threads of the workgroup.
// sxn: N*C floats of dynamic LDS for the sample's normalised tokens, or nullptr (rows that do not fit are read
// from global memory in the merge).  With sxn the rows are fetched by LDS-DMA at the very start and land while
// the DPC-KNN phases run; every other global operand (distances, token weights, noise, mask, the per-sample
// maxima) is also requested before the first barrier, so the kernel pays ONE global round trip, not a chain.
// 16 waves per sample: the density / score rows are spread over them, and the merge runs all clusters at once --
// job j = (cluster, 128-channel chunk) goes to wave j mod 16, the LayerNorm statistics of a cluster meet in LDS.
// (In-kernel stamps of the 4-wave version at N=24, 4 clusters: 19.6k cycles in the density loop, 22k in the
// cluster-by-cluster merge, of 61k.)
#ifdef NR_STAMP
static __device__ unsigned long long nr_back_stamps[16];
#endif
typedef __attribute__((address_space(3))) void* nr_bk_lds_ptr_t;
typedef const __attribute__((address_space(1))) void* nr_bk_glb_ptr_t;

#define BK_THREADS 1024
#define BK_MAXJ 12          // merge jobs per wave: cnum * C/128 <= 16 * BK_MAXJ

// FUSED = called right behind nr_ctm_front_body by the same workgroup (unmasked stages only: without a mask the
// global maximum of the distances is never used, so nothing has to come from other workgroups): sxn already holds
// the rows (the front body's sx), p.smax is not read.
template <bool FUSED = false, int THREADS = BK_THREADS>
__device__ __forceinline__ void nr_ctm_back_body(const NrCtmBackArgs& p, const int b, float* sxn) {
    constexpr int NW = THREADS / 64;
    __shared__ float sd[64][65];
    __shared__ float s_density[64], s_score[64], s_share[64], s_tot[64], s_tokw[64], s_noise[64], s_mask[64];
    __shared__ int s_centre[64], s_assign[64];
    __shared__ float s_red[2][BK_THREADS / 64];
    __shared__ float s_psum[(BK_THREADS / 64) * BK_MAXJ], s_pvar[(BK_THREADS / 64) * BK_MAXJ];
    const int N = p.N, C = p.C, cnum = p.cnum;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef NR_STAMP
    unsigned long long tk[10];
    int tn = 0;
    tk[tn++] = __builtin_readcyclecounter();
#define BK_LAP() tk[tn++] = __builtin_readcyclecounter()
#else
#define BK_LAP() ((void)0)
#endif
    const float* xb = p.xn + (size_t)b * N * C;
    if (sxn && !FUSED) {                              // 1 KiB per wave-instruction, linear copy
        const int chunks = N * C / 256;
        for (int k = wave; k < chunks; k += NW)
            __builtin_amdgcn_global_load_lds((nr_bk_glb_ptr_t)(xb + (size_t)k * 256 + lane * 4),
                                             (nr_bk_lds_ptr_t)(sxn + (size_t)k * 256), 16, 0, 0);
    }
    const float* db = p.dist + (size_t)b * N * N;
    constexpr int DPT = (4096 + THREADS - 1) / THREADS;   // distances per thread: N*N <= 4096
    float dreg[DPT];
#pragma unroll
    for (int u = 0; u < DPT; ++u) {
        const int e = tid + THREADS * u;
        dreg[u] = e < N * N ? db[e] : 0.f;
    }
    if (tid < N) {
        s_tokw[tid] = p.tokw[(size_t)b * N + tid];
        s_noise[tid] = p.noise[(size_t)b * N + tid];
        s_mask[tid] = p.mask ? p.mask[(size_t)b * N + tid] : 1.f;
    }
    // per-channel parameters of this wave's first merge job (the only one at the usual sizes), requested now so
    // that the merge's stores do not wait for another round trip
    const int CH = C / 128, jobs = cnum * CH;
    float pb0[2] = {0.f, 0.f}, nw0[2] = {0.f, 0.f}, nb0[2] = {0.f, 0.f};
    if (wave < jobs) {
        const int c0 = (wave % CH) * 128 + lane;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            pb0[h] = p.proj_b[c0 + 64 * h];
            nw0[h] = p.n1_w[c0 + 64 * h];
            nb0[h] = p.n1_b[c0 + 64 * h];
        }
    }
    // global maximum over all samples (cluster.py:473-475)
    float g = 0.f;
    if constexpr (!FUSED)
        for (int i = tid; i < p.n_samples; i += THREADS) g = fmaxf(g, p.smax[i]);
    g = nr_wave_max(g);
    if (lane == 0) s_red[0][wave] = g;
    __syncthreads();
    BK_LAP();      // 1: operands requested and arrived (first barrier)
    float far = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) far = fmaxf(far, s_red[0][w]);
    far += 1.0f;
    const bool masked = p.mask != nullptr;
    float lmax = 0.f;
#pragma unroll
    for (int u = 0; u < DPT; ++u) {
        const int e = tid + THREADS * u;
        if (e < N * N) {
            int i = e / N, j = e - i * N;
            float dv = dreg[u];
            if (masked && !(s_mask[j] > 0.f)) dv = far;
            sd[i][j] = dv;
            lmax = fmaxf(lmax, dv);
        }
    }
    lmax = nr_wave_max(lmax);
    if (lane == 0) s_red[1][wave] = lmax;
    __syncthreads();
    float dmax = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) dmax = fmaxf(dmax, s_red[1][w]);
    BK_LAP();      // 2: distances in LDS
    for (int i = wave; i < N; i += NW) {              // local density
        float v = lane < N ? sd[i][lane] : INFINITY;
        float acc = 0.f;
        for (int r = 0; r < p.k; ++r) {
            float m = v;
            int idx = lane;
            nr_wave_argmin(m, idx);
            acc += m * m;
            if (lane == idx) v = INFINITY;
        }
        if (lane == 0) {
            float dens = expf(-acc / (float)p.k) + s_noise[i] * 1e-6f;
            if (masked) dens *= (s_mask[i] > 0.f) ? 1.0f : 0.0f;
            s_density[i] = dens;
        }
    }
    __syncthreads();
    BK_LAP();      // 3: densities
    for (int i = wave; i < N; i += NW) {              // distance to the nearest denser token; score
        float di = s_density[i];
        float v = (lane < N && s_density[lane] > di) ? sd[i][lane] : dmax;
        v = nr_wave_min(v);
        if (lane == 0) s_score[i] = v * di;
    }
    __syncthreads();
    if (wave == 0) {                                  // top-cnum centres
        float v = lane < N ? s_score[lane] : -INFINITY;
        for (int c = 0; c < cnum; ++c) {
            float m = v;
            int idx = lane;
            nr_wave_argmax(m, idx);
            if (lane == 0) s_centre[c] = idx;
            if (lane == idx) v = -INFINITY;
        }
    }
    __syncthreads();
    if (tid < N) {                                    // nearest centre; centres join themselves
        float best = INFINITY;
        int bc = 0;
        for (int c = 0; c < cnum; ++c) {
            float dv = sd[s_centre[c]][tid];
            if (dv < best) { best = dv; bc = c; }
        }
        for (int c = 0; c < cnum; ++c)
            if (s_centre[c] == tid) bc = c;
        s_assign[tid] = bc;
        if (p.assign) p.assign[(size_t)b * N + tid] = bc;
    }
    __syncthreads();
    BK_LAP();      // 4: centres + assignment
    // ---- merge_tokens + norm1 ---------------------------------------------------------------------------
    for (int c = wave; c < cnum; c += NW) {           // all_weight of cluster c (cluster.py:536-540), one wave each
        const float t = nr_wave_sum((lane < N && s_assign[lane] == c) ? s_tokw[lane] : 0.f);
        if (lane == 0) s_tot[c] = t + 1e-6f;
    }
    __syncthreads();
    if (tid < N) s_share[tid] = s_tokw[tid] / s_tot[s_assign[tid]];
    if (sxn && !FUSED) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's share of the token rows has landed
    __syncthreads();
    BK_LAP();      // 5: shares, token rows landed
    // job j = cl * CH + ch: cluster cl, channels [128 ch, 128 ch + 128); lane owns channels 128 ch + lane and + 64.
    // Every token row is added with weight (its share, or exactly 0 when it belongs to another cluster): the same
    // fma chain in token order as a loop that skips foreign rows.
    float acc[BK_MAXJ][2];
#pragma unroll
    for (int jj = 0; jj < BK_MAXJ; ++jj) {
        acc[jj][0] = acc[jj][1] = 0.f;
        const int j = wave + NW * jj;
        if (j < jobs) {
            const int cl = j / CH, c0 = (j - cl * CH) * 128 + lane;
            // (two separate loops: a per-element "LDS or global" select makes the compiler issue BOTH loads)
            auto accumulate = [&](auto from_lds) {
                constexpr int RB = 8;
                for (int n0 = 0; n0 < N; n0 += RB) {
                    float x0[RB], x1[RB];
#pragma unroll
                    for (int u = 0; u < RB; ++u) {
                        const int n = min(n0 + u, N - 1);
                        if constexpr (decltype(from_lds)::value) {
                            x0[u] = sxn[n * C + c0];
                            x1[u] = sxn[n * C + c0 + 64];
                        } else {
                            x0[u] = xb[(size_t)n * C + c0];
                            x1[u] = xb[(size_t)n * C + c0 + 64];
                        }
                    }
#pragma unroll
                    for (int u = 0; u < RB; ++u) {
                        const int n = n0 + u;
                        if (n < N) {
                            const float w = s_assign[n] == cl ? s_share[n] : 0.f;
                            acc[jj][0] = fmaf(x0[u], w, acc[jj][0]);
                            acc[jj][1] = fmaf(x1[u], w, acc[jj][1]);
                        }
                    }
                }
            };
            if (sxn) accumulate(std::true_type{});
            else accumulate(std::false_type{});
            const float s = nr_wave_sum(acc[jj][0] + acc[jj][1]);
            if (lane == 0) s_psum[j] = s;
        }
    }
    __syncthreads();
    float mu[BK_MAXJ];
#pragma unroll
    for (int jj = 0; jj < BK_MAXJ; ++jj) {
        mu[jj] = 0.f;
        const int j = wave + NW * jj;
        if (j < jobs) {
            const int cl = j / CH;
            float t = 0.f;
            for (int ch = 0; ch < CH; ++ch) t += s_psum[cl * CH + ch];
            mu[jj] = t / (float)C;
            const float d0 = acc[jj][0] - mu[jj], d1 = acc[jj][1] - mu[jj];
            const float v = nr_wave_sum(d0 * d0 + d1 * d1);
            if (lane == 0) s_pvar[j] = v;
        }
    }
    __syncthreads();
#pragma unroll
    for (int jj = 0; jj < BK_MAXJ; ++jj) {
        const int j = wave + NW * jj;
        if (j < jobs) {
            const int cl = j / CH, c0 = (j - cl * CH) * 128 + lane;
            float t = 0.f;
            for (int ch = 0; ch < CH; ++ch) t += s_pvar[cl * CH + ch];
            const float rstd = rsqrtf(t / (float)C + p.eps);
            const size_t o = ((size_t)b * cnum + cl) * C;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int c = c0 + 64 * h;
                const float a = acc[jj][h];
                const float pbv = jj == 0 ? pb0[h] : p.proj_b[c];
                const float nwv = jj == 0 ? nw0[h] : p.n1_w[c];
                const float nbv = jj == 0 ? nb0[h] : p.n1_b[c];
                if (p.merged) p.merged[o + c] = a;
                p.merged_pb[o + c] = a + pbv;
                const float qv = (a - mu[jj]) * rstd * nwv + nbv;
                if (p.qn_hi) {                       // operand of the q GEMM, split-bf16
                    const uint16_t hh = nr_f2bf(qv);
                    p.qn_hi[o + c] = hh;
                    p.qn_lo[o + c] = nr_f2bf(qv - nr_bf2f(hh));
                } else {
                    p.qn[o + c] = qv;
                }
            }
        }
    }
    BK_LAP();      // 6: merged
#ifdef NR_STAMP
    if (b == 0 && tid == 0 && p.n_samples > 1)
        for (int i = 0; i < tn; ++i) nr_back_stamps[i] = tk[i] - tk[0];
#endif
}

// =====================================================================================================================
// Second form of the two bodies (round 5), for C % 256 == 0.  In-kernel stamps of the first form at N = 24, C = 512 (one
// 1024-thread workgroup per sample): front 38.9k cycles, of which 24.4k in the pairwise distances (a wave-wide reduction
// per pair, rows dealt unevenly: the first wave had 30 of 276 pairs) and 13k in the rows (dword / 2-byte stores, five
// dependent reductions per row); back 31.8k, of which 13k in DPC-KNN (k + 1 + cnum wave-wide arg-min / arg-max rounds per
// row) and 10k in the merge.  This form:
//   * a lane owns 16-byte chunks (channels 256 q + 4 lane ..+3): float4 loads / stores, 8-byte bf16-pair stores, LDS images
//     read and written 1 KiB per wave-instruction without bank conflicts; R rows of a wave are normalised together so that
//     their reductions interleave;
//   * distances: a wave keeps FOUR rows in registers and streams the others past them (one LDS row read and four interleaved
//     reductions per four pairs), units of work dealt over the waves in snake order; the matrix is assembled in LDS and
//     stored coalesced;
//   * DPC-KNN: ONE wave, lane = token, loops over the other tokens with no cross-lane reduction at all (k nearest by k scans
//     of the lane's own row, density of token j by v_readlane, centre = rank of the score by counting);
//   * merge: a thread owns channels, reads every token row once and adds it into the accumulator of its cluster.
// 512-thread workgroups (256 for a handful of tokens): two fit a CU beside the MFMA kernels of the step's other branch.
template <int R>
__device__ __forceinline__ void nr_wave_sum_n(float (&v)[R]) {
#pragma unroll
    for (int u = 0; u < R; ++u) v[u] += nr_dpp<NR_DPP_XOR1>(v[u], v[u]);
#pragma unroll
    for (int u = 0; u < R; ++u) v[u] += nr_dpp<NR_DPP_XOR2>(v[u], v[u]);
#pragma unroll
    for (int u = 0; u < R; ++u) v[u] += nr_dpp<NR_DPP_HALF_MIRROR>(v[u], v[u]);
#pragma unroll
    for (int u = 0; u < R; ++u) v[u] += nr_dpp<NR_DPP_MIRROR>(v[u], v[u]);
#pragma unroll
    for (int u = 0; u < R; ++u) v[u] += nr_dpp<NR_DPP_BCAST15, 0xA>(0.f, v[u]);
#pragma unroll
    for (int u = 0; u < R; ++u) v[u] += nr_dpp<NR_DPP_BCAST31, 0xC>(0.f, v[u]);
#pragma unroll
    for (int u = 0; u < R; ++u) v[u] = nr_lane63(v[u]);
}

// dynamic LDS of the second form's front body (and of the fused front + back): rows, distance matrix, token weights
__host__ __device__ __forceinline__ size_t nr_ctm_front2_lds_floats(int N, int C) { return (size_t)N * C + (size_t)N * N + 128; }     // (+ token weights, + the fused form's noise draws)

// b = sample; smem = nr_ctm_front2_lds_floats(N, C) floats.  V4 = C / 256 chunks of four channels per lane.
template <int V4, int THREADS>
__device__ __forceinline__ void nr_ctm_front_body2(const NrCtmFrontArgs& p, const int b, float* smem) {
    constexpr int NW = THREADS / 64;
    constexpr int R = V4 <= 2 ? 3 : 1;                       // rows a wave normalises together
    __shared__ float s_wmax[NW];
    const int N = p.N, C = p.C;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef NR_STAMP
    unsigned long long fk[8];
    int fn = 0;
    fk[fn++] = __builtin_readcyclecounter();
#endif
    float* sx = smem;                                        // [N][C] normalised rows
    float* s_dist = smem + (size_t)N * C;                    // [N][N]
    float* s_tokw = s_dist + (size_t)N * N;                  // [64]
    f32x4_t lnw[V4], lnb[V4], scw[V4], n1w[V4], n1b[V4];
#pragma unroll
    for (int q = 0; q < V4; ++q) {
        const int c = 256 * q + 4 * lane;
        lnw[q] = *reinterpret_cast<const f32x4_t*>(p.ln_w + c);
        lnb[q] = *reinterpret_cast<const f32x4_t*>(p.ln_b + c);
        scw[q] = *reinterpret_cast<const f32x4_t*>(p.sc_w + c);
        n1w[q] = *reinterpret_cast<const f32x4_t*>(p.n1_w + c);
        n1b[q] = *reinterpret_cast<const f32x4_t*>(p.n1_b + c);
    }
    const float scb = p.sc_b[0];
    const float fC = (float)C;
    // ---- rows: LayerNorm, score, token weight, norm1 ------------------------------------------------------------------
    for (int g = 0; wave + NW * g < N; g += R) {
        f32x4_t v[R][V4];
        bool on[R];
        int rr[R];
        float mk[R];
#pragma unroll
        for (int u = 0; u < R; ++u) {
            const int r = wave + NW * (g + u);
            on[u] = r < N;
            rr[u] = on[u] ? r : wave + NW * g;                // (a row that exists: loaded, never stored)
            const size_t row = (size_t)b * N + rr[u];
            mk[u] = p.mask ? p.mask[row] : 1.f;
#pragma unroll
            for (int q = 0; q < V4; ++q) v[u][q] = *reinterpret_cast<const f32x4_t*>(p.y + row * C + 256 * q + 4 * lane);
        }
        float s[R], var[R];
#pragma unroll
        for (int u = 0; u < R; ++u) {
            s[u] = 0.f;
#pragma unroll
            for (int q = 0; q < V4; ++q) s[u] += (v[u][q][0] + v[u][q][1]) + (v[u][q][2] + v[u][q][3]);
        }
        nr_wave_sum_n<R>(s);
#pragma unroll
        for (int u = 0; u < R; ++u) {
            s[u] = s[u] / fC;
            var[u] = 0.f;
#pragma unroll
            for (int q = 0; q < V4; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float dl = v[u][q][e] - s[u]; var[u] = fmaf(dl, dl, var[u]); }
        }
        nr_wave_sum_n<R>(var);
        float dot[R], s2[R];
#pragma unroll
        for (int u = 0; u < R; ++u) {
            const float rstd = rsqrtf(var[u] / fC + p.eps);
            dot[u] = s2[u] = 0.f;
#pragma unroll
            for (int q = 0; q < V4; ++q) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float x = (v[u][q][e] - s[u]) * rstd * lnw[q][e] + lnb[q][e];
                    v[u][q][e] = x;
                    dot[u] = fmaf(x, scw[q][e], dot[u]);
                    s2[u] += x;
                }
                if (on[u]) {
                    *reinterpret_cast<f32x4_t*>(p.xn + ((size_t)b * N + rr[u]) * C + 256 * q + 4 * lane) = v[u][q];
                    *reinterpret_cast<f32x4_t*>(sx + (size_t)rr[u] * C + 256 * q + 4 * lane) = v[u][q];
                }
            }
        }
        float red[2 * R];
#pragma unroll
        for (int u = 0; u < R; ++u) { red[u] = dot[u]; red[R + u] = s2[u]; }
        nr_wave_sum_n<2 * R>(red);
#pragma unroll
        for (int u = 0; u < R; ++u) {
            float sc = red[u] + scb;
            if (p.mask && mk[u] == 0.f) sc = -INFINITY;
            if (lane == 0 && on[u]) {
                const size_t row = (size_t)b * N + rr[u];
                const float tw = expf(sc);
                p.score[row] = sc;
                p.tokw[row] = tw;
                s_tokw[rr[u]] = tw;
            }
            s2[u] = red[R + u] / fC;
            var[u] = 0.f;
#pragma unroll
            for (int q = 0; q < V4; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float dl = v[u][q][e] - s2[u]; var[u] = fmaf(dl, dl, var[u]); }
        }
        nr_wave_sum_n<R>(var);
#pragma unroll
        for (int u = 0; u < R; ++u) {
            const float rstd2 = rsqrtf(var[u] / fC + p.eps);
            const size_t o = ((size_t)b * N + rr[u]) * C + 4 * lane;
#pragma unroll
            for (int q = 0; q < V4; ++q) {
                f32x4_t kv;
#pragma unroll
                for (int e = 0; e < 4; ++e) kv[e] = (v[u][q][e] - s2[u]) * rstd2 * n1w[q][e] + n1b[q][e];
                if (!on[u]) continue;
                if (p.kvn_hi) {
                    uint32_t h0, l0, h1, l1;
                    nr_split_pk(kv[0], kv[1], h0, l0);
                    nr_split_pk(kv[2], kv[3], h1, l1);
                    *reinterpret_cast<uint2*>(p.kvn_hi + o + 256 * q) = make_uint2(h0, h1);
                    *reinterpret_cast<uint2*>(p.kvn_lo + o + 256 * q) = make_uint2(l0, l1);
                } else {
                    *reinterpret_cast<f32x4_t*>(p.kvn + o + 256 * q) = kv;
                }
            }
        }
    }
    CF_LAP();      // 1: this wave's rows normalised and stored (issue side)
    __syncthreads();
    CF_LAP();      // 2: every wave's rows
    // ---- pairwise distances: two rows in registers against FOUR later rows at a time, one per 16-lane row of the wave ------
    // (a lane owns 16-byte chunks 64 q + 4 t of its row's channels, t = lane & 15: the sum over channels then needs a reduction
    // over 16 lanes only -- four DPP steps for the eight pairs of a step -- and the four 16-lane rows read four different token
    // rows without bank conflicts: a ds_read_b128 is served in groups of 16 lanes that between them hold every t once)
    constexpr int CH = 4 * V4;                               // chunks per lane: C / 64
    const int g4 = lane >> 4, t = lane & 15;
    const int nbk = (N + 1) >> 1;                            // blocks of two rows
    float wmax = 0.f;
    for (int k = 0;; ++k) {
        const int ib = (k & 1) ? k * NW + (NW - 1 - wave) : k * NW + wave;      // snake order: the blocks shrink with their index
        if (ib >= nbk) break;
        const int i0 = 2 * ib;
        f32x4_t xi[2][CH];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int q = 0; q < CH; ++q) xi[a][q] = *reinterpret_cast<const f32x4_t*>(sx + (size_t)min(i0 + a, N - 1) * C + 64 * q + 4 * t);
        if (lane < 2 && i0 + lane < N) s_dist[(i0 + lane) * N + i0 + lane] = 0.f;
        for (int j0 = i0 + 1; j0 < N; j0 += 4) {
            const int j = j0 + g4;
            const float* xr = sx + (size_t)min(j, N - 1) * C + 4 * t;
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int q = 0; q < CH; ++q) {
                const f32x4_t xj = *reinterpret_cast<const f32x4_t*>(xr + 64 * q);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float d0 = xi[0][q][e] - xj[e], d1 = xi[1][q][e] - xj[e];
                    s0 = fmaf(d0, d0, s0);
                    s1 = fmaf(d1, d1, s1);
                }
            }
            s0 += nr_dpp<NR_DPP_XOR1>(s0, s0);
            s1 += nr_dpp<NR_DPP_XOR1>(s1, s1);
            s0 += nr_dpp<NR_DPP_XOR2>(s0, s0);
            s1 += nr_dpp<NR_DPP_XOR2>(s1, s1);
            s0 += nr_dpp<NR_DPP_HALF_MIRROR>(s0, s0);
            s1 += nr_dpp<NR_DPP_HALF_MIRROR>(s1, s1);
            s0 += nr_dpp<NR_DPP_MIRROR>(s0, s0);
            s1 += nr_dpp<NR_DPP_MIRROR>(s1, s1);
            const int i = i0 + (t & 1);              // the SQUARED distance: its root is taken once per entry in the pass below
            if (t < 4 && j < N && i < j) s_dist[(t & 2) ? j * N + i : i * N + j] = (t & 1) ? s1 : s0;
        }
    }
    CF_LAP();      // 3: this wave's distances
    __syncthreads();
    float* db = p.dist + (size_t)b * N * N;
    for (int e = tid; e < N * N; e += THREADS) {
        const float dv = sqrtf(s_dist[e]) * p.inv_sqrt_c;
        s_dist[e] = dv;                              // (the fused front + back kernel reads the matrix from here)
        db[e] = dv;
        wmax = fmaxf(wmax, dv);
    }
    wmax = nr_wave_max(wmax);
    if (lane == 0) s_wmax[wave] = wmax;
    __syncthreads();
    if (tid == 0) {
        float m = 0.f;
        for (int w = 0; w < NW; ++w) m = fmaxf(m, s_wmax[w]);
        p.smax[b] = m;
    }
    CF_LAP();      // 4: end
#ifdef NR_STAMP
    if (b == 0 && tid == 0)
        for (int i = 0; i < fn; ++i) nr_front_stamps[i] = fk[i] - fk[0];
#endif
}

// b = sample.  sxn: N*C floats of LDS for the sample's normalised rows, or nullptr (rows read from global memory in the merge).
// FUSED: called right behind nr_ctm_front_body2 by the same workgroup (unmasked stages): sxn = the front body's rows,
// s_dist / s_tokw = its distance matrix and token weights in LDS; p.smax is not read.
// MAXC: clusters the accumulators are sized for (4 or 16; cnum <= MAXC); CPT: channels per thread (C <= THREADS * CPT);
// MAXN: tokens the DPC-KNN wave's registers are sized for (N <= MAXN <= 64).
template <bool FUSED, int THREADS, int MAXC, int CPT, int MAXN>
__device__ __forceinline__ void nr_ctm_back_body2(const NrCtmBackArgs& p, const int b, float* sxn, const float* s_dist, const float* s_tokw_in) {
    constexpr int NW = THREADS / 64;
    __shared__ float sd[64][65];
    __shared__ float s_tokw[64], s_noise[64], s_mask[64], s_share[64];
    __shared__ int s_assign[64];
    __shared__ float s_red[NW];
    __shared__ float s_part[2][NW][MAXC];
    const int N = p.N, C = p.C, cnum = p.cnum;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef NR_STAMP
    unsigned long long tk[10];
    int tn = 0;
    tk[tn++] = __builtin_readcyclecounter();
#endif
    const float* xb = p.xn + (size_t)b * N * C;
    if (sxn && !FUSED) {                              // 1 KiB per wave-instruction, linear copy; lands while DPC-KNN runs
        const int chunks = N * C / 256;
        for (int k = wave; k < chunks; k += NW)
            __builtin_amdgcn_global_load_lds((nr_bk_glb_ptr_t)(xb + (size_t)k * 256 + lane * 4),
                                             (nr_bk_lds_ptr_t)(sxn + (size_t)k * 256), 16, 0, 0);
    }
    const bool masked = p.mask != nullptr;
    // per-channel parameters of the epilogue, requested now: their round trip ends long before the merge needs them
    float pbv[CPT], nwv[CPT], nbv[CPT];
#pragma unroll
    for (int h = 0; h < CPT; ++h) {
        const int ch = tid + THREADS * h;
        pbv[h] = ch < C ? p.proj_b[ch] : 0.f;
        nwv[h] = ch < C ? p.n1_w[ch] : 0.f;
        nbv[h] = ch < C ? p.n1_b[ch] : 0.f;
    }
    constexpr int DPT = (4096 + THREADS - 1) / THREADS;
    float dreg[DPT];
    const float* db = p.dist + (size_t)b * N * N;
#pragma unroll
    for (int u = 0; u < DPT; ++u) {
        const int e = tid + THREADS * u;
        dreg[u] = e < N * N ? (FUSED ? s_dist[e] : db[e]) : 0.f;
    }
    if (tid < N) {
        s_tokw[tid] = FUSED ? s_tokw_in[tid] : p.tokw[(size_t)b * N + tid];
        s_noise[tid] = FUSED ? s_tokw_in[64 + tid] : p.noise[(size_t)b * N + tid];     // (fused form: requested before the front half ran)
        s_mask[tid] = masked ? p.mask[(size_t)b * N + tid] : 1.f;
    }
    float g = 0.f;                                    // global maximum over all samples (cluster.py:473-475)
    if constexpr (!FUSED)
        if (masked)
            for (int i = tid; i < p.n_samples; i += THREADS) g = fmaxf(g, p.smax[i]);
    g = nr_wave_max(g);
    if (lane == 0) s_red[wave] = g;
    __syncthreads();
    float far = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) far = fmaxf(far, s_red[w]);
    far += 1.0f;
    BK_LAP();      // 1: operands requested and arrived (first barrier)
#pragma unroll
    for (int u = 0; u < DPT; ++u) {
        const int e = tid + THREADS * u;
        if (e < N * N) {
            const int i = e / N, j = e - i * N;
            sd[i][j] = (masked && !(s_mask[j] > 0.f)) ? far : dreg[u];
        }
    }
    __syncthreads();
    BK_LAP();      // 2: distances in LDS
    // ---- DPC-KNN by ONE wave: lane = token with its row of distances in REGISTERS; every loop runs over the other tokens, no
    // cross-lane reduction but one max; the cluster weights and the tokens' shares come out of the same wave (v_readlane loops)
    if (wave == 0) {
        const bool tok = lane < N;
        const int i = tok ? lane : 0;
        // (branch-free on purpose: with `if`s the compiler splits these unrolled loops into one basic block per entry -- 32 LDS reads
        // each behind its own branch and wait took 4k cycles, the scans twice that)
        float row[MAXN];
#pragma unroll
        for (int j = 0; j < MAXN; ++j) row[j] = sd[i][j < N ? j : 0];
#pragma unroll
        for (int j = 0; j < MAXN; ++j) row[j] = j < N ? row[j] : INFINITY;
        const float tw = tok ? s_tokw[i] : 0.f, nz = s_noise[i], mki = s_mask[i];
        // k nearest (the token itself, at distance 0, included: cluster.py:478).  k <= 4 (the model's CTMs use 3): ONE pass that keeps
        // the four smallest entries sorted (a min / max insertion network, seven instructions per entry); larger k: k scans, each
        // taking the smallest entry behind the previous one in (value, index) order
        float acc = 0.f, rowmax = 0.f;
#pragma unroll
        for (int j = 0; j < MAXN; ++j) rowmax = fmaxf(rowmax, j < N ? row[j] : 0.f);
        if (p.k <= 4) {
            float t0 = INFINITY, t1 = INFINITY, t2 = INFINITY, t3 = INFINITY;
#pragma unroll
            for (int j = 0; j < MAXN; ++j) {
                const float v = row[j];
                const float a = fmaxf(v, t0);
                t0 = fminf(v, t0);
                const float c = fmaxf(a, t1);
                t1 = fminf(a, t1);
                const float e = fmaxf(c, t2);
                t2 = fminf(c, t2);
                t3 = fminf(e, t3);
            }
            acc = t0 * t0;                               // (ascending order, as the scans below add them)
            if (p.k > 1) acc += t1 * t1;
            if (p.k > 2) acc += t2 * t2;
            if (p.k > 3) acc += t3 * t3;
        } else {
            float pv = -1.f;
            int pj = -1;
            for (int r = 0; r < p.k; ++r) {
                float m = INFINITY;
                int mj = MAXN;
#pragma unroll
                for (int j = 0; j < MAXN; ++j) {
                    const float v = row[j];
                    const bool take = ((v > pv) | ((v == pv) & (j > pj))) & (v < m);
                    m = take ? v : m;
                    mj = take ? j : mj;
                }
                acc += m * m;
                pv = m;
                pj = mj;
            }
        }
        const float dmax = nr_wave_max(tok ? rowmax : 0.f);                       // cluster.py:493
        float dens = expf(-acc / (float)p.k) + nz * 1e-6f;                         // :479-484
        if (masked) dens *= (mki > 0.f) ? 1.0f : 0.0f;                             // :488
        float parent = dmax;                                                       // distance to the nearest denser token (:491-494)
#pragma unroll
        for (int j = 0; j < MAXN; ++j) {                 // (a lane j >= N holds a copy of token 0's density and row[j] = inf: no effect)
            const float dj = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, dens), j));
            parent = dj > dens ? fminf(parent, row[j]) : parent;
        }
        const float score = parent * dens;                                         // :497
        const float score_x = tok ? score : -INFINITY;                             // (lanes that are no token never outrank one)
        int rank = 0;                                                              // :498, ties -> the lower index first
#pragma unroll
        for (int j = 0; j < MAXN; ++j) {
            const float sj = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, score_x), j));
            rank += ((sj > score) | ((sj == score) & (j < lane))) ? 1 : 0;
        }
        // nearest centre (:501-502): token `lane` against row `centre c` of the matrix (the mask fills COLUMNS: a masked token is
        // `far` from every centre and joins the first); centres join themselves (:505-507)
        float best = INFINITY;
        int bc = 0;
        if constexpr (MAXC <= 4) {                      // all reads first, then the comparisons
            float dvc[MAXC];
#pragma unroll
            for (int c = 0; c < MAXC; ++c) {
                const unsigned long long holds = __ballot(tok & (rank == c));      // the lane whose rank is c
                const int ctr = holds ? __ffsll((long long)holds) - 1 : 0;
                dvc[c] = sd[ctr][i];
            }
#pragma unroll
            for (int c = 0; c < MAXC; ++c) {
                const bool take = (c < cnum) & (dvc[c] < best);
                best = take ? dvc[c] : best;
                bc = take ? c : bc;
            }
        } else {                                        // (sixteen more live registers beside a 64-entry row would spill)
            for (int c = 0; c < cnum; ++c) {
                const unsigned long long holds = __ballot(tok & (rank == c));
                const int ctr = holds ? __ffsll((long long)holds) - 1 : 0;
                const float dv = sd[ctr][i];
                const bool take = dv < best;
                best = take ? dv : best;
                bc = take ? c : bc;
            }
        }
        bc = rank < cnum ? rank : bc;
        // all_weight of every cluster, tokens in order (:536-540): lane c adds up cluster c; then every token's share
        float tot = 0.f;
#pragma unroll
        for (int n = 0; n < MAXN; ++n) {                 // (tw is 0 in lanes that are no token)
            const int an = __builtin_amdgcn_readlane(bc, n);
            const float wn = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, tw), n));
            tot += (an == lane) ? wn : 0.f;
        }
        tot += 1e-6f;
        const float mytot = __shfl(tot, bc);
        if (tok) {
            s_assign[lane] = bc;
            s_share[lane] = tw / mytot;
            if (p.assign) p.assign[(size_t)b * N + lane] = bc;
        }
    }
    if (sxn && !FUSED) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this wave's share of the rows has landed
    __syncthreads();
    BK_LAP();      // 3: DPC-KNN done (wave 0), everybody past the barrier
    BK_LAP();      // 4: shares; token rows landed
    float acc[MAXC][CPT];
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
#pragma unroll
        for (int h = 0; h < CPT; ++h) acc[c][h] = 0.f;
    for (int n0 = 0; n0 < N; n0 += 4) {               // four rows per turn: their LDS reads go out together
        float x[4][CPT], w4[4];
        int a4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int n = min(n0 + u, N - 1);
#pragma unroll
            for (int h = 0; h < CPT; ++h) {
                const int ch = tid + THREADS * h;
                x[u][h] = ch < C ? (sxn ? sxn[(size_t)n * C + ch] : xb[(size_t)n * C + ch]) : 0.f;
            }
            a4[u] = n0 + u < N ? s_assign[n] : -1;       // (a row past the end joins no cluster)
            w4[u] = s_share[n];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int an = __builtin_amdgcn_readfirstlane(a4[u]);
#pragma unroll
            for (int c = 0; c < MAXC; ++c) {
                if constexpr (MAXC <= 4) {
                    const float ws = an == c ? w4[u] : 0.f;   // a row of another cluster is added with weight exactly 0
#pragma unroll
                    for (int h = 0; h < CPT; ++h) acc[c][h] = fmaf(x[u][h], ws, acc[c][h]);
                } else {
                    if (an == c) {
#pragma unroll
                        for (int h = 0; h < CPT; ++h) acc[c][h] = fmaf(x[u][h], w4[u], acc[c][h]);
                    }
                }
            }
        }
    }
    // norm1 of every merged row: mean and variance over the C channels (spread over all threads)
    float red[MAXC];
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        red[c] = 0.f;
#pragma unroll
        for (int h = 0; h < CPT; ++h) red[c] += acc[c][h];           // (channels >= C hold zeros)
    }
    nr_wave_sum_n<MAXC>(red);
    if (lane == 0)
#pragma unroll
        for (int c = 0; c < MAXC; ++c) s_part[0][wave][c] = red[c];
    __syncthreads();
    float mu[MAXC];
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) t += s_part[0][w][c];
        mu[c] = t / (float)C;
        red[c] = 0.f;
#pragma unroll
        for (int h = 0; h < CPT; ++h)
            if (tid + THREADS * h < C) { const float dl = acc[c][h] - mu[c]; red[c] = fmaf(dl, dl, red[c]); }
    }
    nr_wave_sum_n<MAXC>(red);
    if (lane == 0)
#pragma unroll
        for (int c = 0; c < MAXC; ++c) s_part[1][wave][c] = red[c];
    __syncthreads();
#pragma unroll
    for (int h = 0; h < CPT; ++h) {
        const int ch = tid + THREADS * h;
        if (ch >= C) continue;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            if (c >= cnum) continue;
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) t += s_part[1][w][c];
            const float rstd = rsqrtf(t / (float)C + p.eps);
            const size_t o = ((size_t)b * cnum + c) * C + ch;
            const float a = acc[c][h];
            if (p.merged) p.merged[o] = a;
            p.merged_pb[o] = a + pbv[h];
            const float qv = (a - mu[c]) * rstd * nwv[h] + nbv[h];
            if (p.qn_hi) {                           // operand of the q GEMM, split-bf16
                const uint16_t hh = nr_f2bf(qv);
                p.qn_hi[o] = hh;
                p.qn_lo[o] = nr_f2bf(qv - nr_bf2f(hh));
            } else {
                p.qn[o] = qv;
            }
        }
    }
    BK_LAP();      // 5: merged + stored
#ifdef NR_STAMP
    if (b == 0 && tid == 0 && p.n_samples > 1)
        for (int i = 0; i < tn; ++i) nr_back_stamps[i] = tk[i] - tk[0];
#endif
}

// ---- score-biased multi-head attention: merged tokens (queries) over un-merged tokens ---------------------
// 16 waves, one per (head, query); lane = key for the logits / softmax, lane = channel for the value sum.
// head_dim = 64, N <= 64.
struct NrAttnArgs {
    const float *q, *kv, *score;
    int N, C, cnum, H;
    float scale;
    float* out;
    uint16_t *out_hi, *out_lo;       // when set, the result is written split-bf16 (operand of the proj GEMM)
};

// skv: N * (2C + 4) floats of LDS for the sample's k|v rows (row stride padded by 4 floats so that the 16 lanes
// of a ds_read_b128 group, one key row each, fall on different banks), or nullptr to read k and v from global
// memory (samples whose rows do not fit).  With the rows in LDS the kernel makes ONE global round trip for
// them instead of a dependent chain per (head, query) job.
__device__ __forceinline__ int nr_tc_attention_lds_floats(int N, int C) { return N * (2 * C + 4); }

__device__ __forceinline__ void nr_tc_attention_body(const NrAttnArgs& a, const int b, float* skv) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = (int)(blockDim.x >> 6);
    const int N = a.N, C = a.C, cnum = a.cnum;
    const float scale = a.scale;
    const float* kvb = a.kv + (size_t)b * N * 2 * C;
    const int ld = skv ? 2 * C + 4 : 2 * C;
    // the wave's first query vector (lane = channel) and the score bias: requested together with the k | v rows
    const int job0 = wave;
    float q0 = 0.f;
    if (job0 < a.H * cnum) {
        const int h = job0 / cnum, cl = job0 - h * cnum;
        q0 = a.q[((size_t)b * cnum + cl) * C + h * 64 + lane] * scale;
    }
    const float bias = lane < N ? a.score[(size_t)b * N + lane] : 0.f;
    if (skv) {
        const int per_row = 2 * C / 4;                       // float4 per row
        for (int e = threadIdx.x; e < N * per_row; e += (int)blockDim.x) {
            const int n = e / per_row, c4 = e - n * per_row;
            *reinterpret_cast<f32x4_t*>(skv + n * ld + c4 * 4) = *reinterpret_cast<const f32x4_t*>(kvb + (size_t)n * 2 * C + c4 * 4);
        }
        __syncthreads();
    }
    const float* kvs = skv ? skv : kvb;
    for (int job = wave; job < a.H * cnum; job += nwave) {
        const int h = job / cnum, cl = job - h * cnum;
        const float qs = job == job0 ? q0 : a.q[((size_t)b * cnum + cl) * C + h * 64 + lane] * scale;
        float logit = -INFINITY;
        {
            const float* kr = kvs + (size_t)min(lane, N - 1) * ld + h * 64;
            float dot = 0.f;
#pragma unroll
            for (int j = 0; j < 64; j += 4) {
                const f32x4_t kk = *reinterpret_cast<const f32x4_t*>(kr + j);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    dot = fmaf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, qs), j + e)), kk[e], dot);
            }
            if (lane < N) logit = dot + bias;
        }
        const float m = nr_wave_max(logit);
        float e = lane < N ? expf(logit - m) : 0.f;
        const float den = nr_wave_sum(e);
        const float p = e / den;
        float acc = 0.f;
        for (int n = 0; n < N; ++n) {
            const float pn = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, p), n));
            acc = fmaf(pn, kvs[(size_t)n * ld + C + h * 64 + lane], acc);
        }
        const size_t o = ((size_t)b * cnum + cl) * C + h * 64 + lane;
        if (a.out_hi) {
            const uint16_t hh = nr_f2bf(acc);
            a.out_hi[o] = hh;
            a.out_lo[o] = nr_f2bf(acc - nr_bf2f(hh));
        } else {
            a.out[o] = acc;
        }
    }
}

// The same attention for HG of a sample's heads (a workgroup per (sample, head group), 64 * HG threads or more): only those
// heads' k | v columns go to LDS -- N * (128 * HG + 4) floats instead of N * (2C + 4): 25 KB instead of 99 at N = 24, HG = 2, and
// 67 KB at N = 64, where the whole rows (263 KB) do not fit and nr_tc_attention_body reads k and v from global memory job by
// job.  Every (head, query) job is computed by one wave with the arithmetic of nr_tc_attention_body.
template <int HG>
__device__ __forceinline__ void nr_tc_attention_heads_body(const NrAttnArgs& a, const int b, const int hg, float* skv) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = (int)(blockDim.x >> 6);
    const int N = a.N, C = a.C, cnum = a.cnum, h0 = hg * HG;
    const float scale = a.scale;
    const float* kvb = a.kv + (size_t)b * N * 2 * C;
    constexpr int HW = 64 * HG;                              // floats of k (and of v) per row kept
    constexpr int ld = 2 * HW + 4;
    constexpr int per_row = 2 * HW / 4;                      // float4 per row
    constexpr int QJ = 4;                                    // (head, query) jobs whose query vectors a wave requests up front
    // this wave's first query vectors (lane = channel) and the sample's score bias, requested together with the k | v rows: one
    // global round trip for everything (the q loads used to sit inside the per-key dot product, a round trip per job)
    float qv[QJ];
#pragma unroll
    for (int u = 0; u < QJ; ++u) {
        const int job = wave + nwave * u;
        const int hl = job / cnum, cl = job - hl * cnum;
        qv[u] = job < HG * cnum ? a.q[((size_t)b * cnum + cl) * C + (h0 + hl) * 64 + lane] * scale : 0.f;
    }
    const float bias = lane < N ? a.score[(size_t)b * N + lane] : 0.f;
    for (int e = threadIdx.x; e < N * per_row; e += (int)blockDim.x) {
        const int n = e / per_row, c4 = e - n * per_row;
        const int c = c4 * 4;
        const int src = c < HW ? h0 * 64 + c : C + h0 * 64 + (c - HW);
        *reinterpret_cast<f32x4_t*>(skv + n * ld + c) = *reinterpret_cast<const f32x4_t*>(kvb + (size_t)n * 2 * C + src);
    }
    __syncthreads();
    for (int job = wave, u = 0; job < HG * cnum; job += nwave, ++u) {
        const int hl = job / cnum, cl = job - hl * cnum, h = h0 + hl;
        float qs;                                            // this job's scaled query, lane = channel
        if (u < QJ) {
            qs = 0.f;
#pragma unroll
            for (int w = 0; w < QJ; ++w) qs = u == w ? qv[w] : qs;
        } else {
            qs = a.q[((size_t)b * cnum + cl) * C + h * 64 + lane] * scale;
        }
        float logit = -INFINITY;
        {
            const float* kr = skv + (size_t)min(lane, N - 1) * ld + hl * 64;
            float dot = 0.f;
#pragma unroll
            for (int j = 0; j < 64; j += 4) {
                const f32x4_t kk = *reinterpret_cast<const f32x4_t*>(kr + j);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    dot = fmaf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, qs), j + e)), kk[e], dot);
            }
            if (lane < N) logit = dot + bias;
        }
        const float m = nr_wave_max(logit);
        float e = lane < N ? expf(logit - m) : 0.f;
        const float den = nr_wave_sum(e);
        const float p = e / den;
        float acc = 0.f;
        for (int n = 0; n < N; ++n) {
            const float pn = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, p), n));
            acc = fmaf(pn, skv[(size_t)n * ld + HW + hl * 64 + lane], acc);
        }
        const size_t o = ((size_t)b * cnum + cl) * C + h * 64 + lane;
        if (a.out_hi) {
            const uint16_t hh = nr_f2bf(acc);
            a.out_hi[o] = hh;
            a.out_lo[o] = nr_f2bf(acc - nr_bf2f(hh));
        } else {
            a.out[o] = acc;
        }
    }
}

// ---- x[n-1] | x[n] | x[n+1] of one token row, written as bf16 hi / lo (operand of the conv GEMM) ------------
struct NrShiftArgs {
    const float* x;
    int N, C;
    uint16_t *hi, *lo;
    int plain;        // 1: the row itself as a bf16 pair, [rows, C] (the conv GEMM reads its neighbours in place)
};

__device__ __forceinline__ void nr_shift_split_body(const NrShiftArgs& a, const int row) {
    const int n = row % a.N, C = a.C;
    if (a.plain) {
        const float* src = a.x + (size_t)row * C;
        const size_t o = (size_t)row * C;
        for (int c = threadIdx.x * 4; c < C; c += 1024) {
            const f32x4_t v = *reinterpret_cast<const f32x4_t*>(src + c);
            uint16_t h[4], l[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                h[e] = nr_f2bf(v[e]);
                l[e] = nr_f2bf(v[e] - nr_bf2f(h[e]));
            }
            *reinterpret_cast<uint2*>(a.hi + o + c) = make_uint2((uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16));
            *reinterpret_cast<uint2*>(a.lo + o + c) = make_uint2((uint32_t)l[0] | ((uint32_t)l[1] << 16), (uint32_t)l[2] | ((uint32_t)l[3] << 16));
        }
        return;
    }
    for (int k = 0; k < 3; ++k) {
        const int nn = n + k - 1;
        const bool ok = nn >= 0 && nn < a.N;
        const float* src = a.x + (size_t)(row + k - 1) * C;
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
            *reinterpret_cast<uint2*>(a.hi + o + c) = make_uint2((uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16));
            *reinterpret_cast<uint2*>(a.lo + o + c) = make_uint2((uint32_t)l[0] | ((uint32_t)l[1] << 16), (uint32_t)l[2] | ((uint32_t)l[3] << 16));
        }
    }
}
