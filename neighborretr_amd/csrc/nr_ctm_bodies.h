// Device bodies of the clustering-stage kernels, shared by the one-problem launches (nr_ctm_fused.hip,
// nr_ctm.hip) and the grouped launches that run the text and the video problem of a stage in the same
// grid (nr_ctm_group.hip).  Reference: cluster.py:453-561 (DPC-KNN, merge_tokens), :689-717 (CTM.forward),
// :834-888 (score-biased attention).
#pragma once
#include <type_traits>
#include "nr_common.h"

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
template <int CPL, int THREADS = CF_THREADS>
__device__ __forceinline__ void nr_ctm_front_body(const NrCtmFrontArgs& p, const int b, float* sx) {
    __shared__ float s_wmax[THREADS / 64];
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
    __syncthreads();
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
    __syncthreads();
    if (tid == 0) {
        float m = 0.f;
        for (int w = 0; w < NW; ++w) m = fmaxf(m, s_wmax[w]);
        p.smax[b] = m;
    }
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
    constexpr int DPT = 4;                            // distances per thread: N*N <= 4096 = 4 * 1024
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
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int N = a.N, C = a.C, cnum = a.cnum;
    const float scale = a.scale;
    const float* kvb = a.kv + (size_t)b * N * 2 * C;
    const int ld = skv ? 2 * C + 4 : 2 * C;
    if (skv) {
        const int per_row = 2 * C / 4;                       // float4 per row
        for (int e = threadIdx.x; e < N * per_row; e += (int)blockDim.x) {
            const int n = e / per_row, c4 = e - n * per_row;
            *reinterpret_cast<f32x4_t*>(skv + n * ld + c4 * 4) = *reinterpret_cast<const f32x4_t*>(kvb + (size_t)n * 2 * C + c4 * 4);
        }
        __syncthreads();
    }
    const float* kvs = skv ? skv : kvb;
    for (int job = wave; job < a.H * cnum; job += (int)(blockDim.x >> 6)) {
        const int h = job / cnum, cl = job - h * cnum;
        const float* qr = a.q + ((size_t)b * cnum + cl) * C + h * 64;
        float logit = -INFINITY;
        if (lane < N) {
            const float* kr = kvs + (size_t)lane * ld + h * 64;
            float dot = 0.f;
#pragma unroll
            for (int j = 0; j < 64; j += 4) {
                f32x4_t kk = *reinterpret_cast<const f32x4_t*>(kr + j);
                f32x4_t qq = *reinterpret_cast<const f32x4_t*>(qr + j);
                dot += (qq[0] * scale) * kk[0] + (qq[1] * scale) * kk[1] + (qq[2] * scale) * kk[2] + (qq[3] * scale) * kk[3];
            }
            logit = dot + a.score[(size_t)b * N + lane];
        }
        const float m = nr_wave_max(logit);
        float e = lane < N ? expf(logit - m) : 0.f;
        const float den = nr_wave_sum(e);
        const float p = e / den;
        float acc = 0.f;
        for (int n = 0; n < N; ++n) {
            float pn = __shfl(p, n);
            acc += pn * kvs[(size_t)n * ld + C + h * 64 + lane];
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
// job.  Every (head, query) job is computed by one wave with the arithmetic of nr_tc_attention_body: identical bits.
template <int HG>
__device__ __forceinline__ void nr_tc_attention_heads_body(const NrAttnArgs& a, const int b, const int hg, float* skv) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int N = a.N, C = a.C, cnum = a.cnum, h0 = hg * HG;
    const float scale = a.scale;
    const float* kvb = a.kv + (size_t)b * N * 2 * C;
    constexpr int HW = 64 * HG;                              // floats of k (and of v) per row kept
    constexpr int ld = 2 * HW + 4;
    constexpr int per_row = 2 * HW / 4;                      // float4 per row
    for (int e = threadIdx.x; e < N * per_row; e += (int)blockDim.x) {
        const int n = e / per_row, c4 = e - n * per_row;
        const int c = c4 * 4;
        const int src = c < HW ? h0 * 64 + c : C + h0 * 64 + (c - HW);
        *reinterpret_cast<f32x4_t*>(skv + n * ld + c) = *reinterpret_cast<const f32x4_t*>(kvb + (size_t)n * 2 * C + src);
    }
    __syncthreads();
    for (int job = wave; job < HG * cnum; job += (int)(blockDim.x >> 6)) {
        const int hl = job / cnum, cl = job - hl * cnum, h = h0 + hl;
        const float* qr = a.q + ((size_t)b * cnum + cl) * C + h * 64;
        float logit = -INFINITY;
        if (lane < N) {
            const float* kr = skv + (size_t)lane * ld + hl * 64;
            float dot = 0.f;
#pragma unroll
            for (int j = 0; j < 64; j += 4) {
                f32x4_t kk = *reinterpret_cast<const f32x4_t*>(kr + j);
                f32x4_t qq = *reinterpret_cast<const f32x4_t*>(qr + j);
                dot += (qq[0] * scale) * kk[0] + (qq[1] * scale) * kk[1] + (qq[2] * scale) * kk[2] + (qq[3] * scale) * kk[3];
            }
            logit = dot + a.score[(size_t)b * N + lane];
        }
        const float m = nr_wave_max(logit);
        float e = lane < N ? expf(logit - m) : 0.f;
        const float den = nr_wave_sum(e);
        const float p = e / den;
        float acc = 0.f;
        for (int n = 0; n < N; ++n) {
            float pn = __shfl(p, n);
            acc += pn * skv[(size_t)n * ld + HW + hl * 64 + lane];
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
