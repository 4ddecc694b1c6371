// Sinkhorn targets for the uniform-regularisation loss.
// Reference: UniformRegularizationLoss.sinkhorn_algorithm, until_module.py:235-266 --
//   mu = nu = -log(2B); u = v = 0; 50 x { u = mu - LSE_j(G + v);  v = nu - LSE_i(G + u) };
//   Q = exp(G + u + v - norm);  target = beta*Q + (1-beta)*I.
// The reference runs it on G and, separately, on G^T (modeling.py:440-441); both problems are
// solved in one launch (blockIdx.x = direction).
//
// B <= 128: ONE persistent workgroup of 1024 threads per direction; the 100 dependent reductions
// never leave the CU.  8 lanes share a line; every thread keeps 16 contiguous entries of its row
// (row pass) AND 16 of its column (column pass) in registers.  The first iteration runs in the log
// domain exactly as the reference (safe for any logit range); from then on the plan
// P = exp(G + u + v) itself is carried in registers and each half-iteration is the equivalent
// matrix-scaling form: P_ij = a_i K_ij b_j with K = exp(G + u1 + v1) fixed in registers and
//   a_i = e^mu / sum_j K_ij b_j,   b_j = e^nu / sum_i K_ij a_i
// (u_i <- mu - LSE_j(G_ij + v_j)  <=>  scale row i so that it sums to e^mu).  That replaces 2 x 16
// exp per thread and half-iteration by 16 FMAs: the kernel is VALU-issue-bound on one CU and this is
// what shortens the step's critical path.  After the first iteration every entry of K is <= 1/(2B)
// and a = b = 1, so nothing can overflow.
// B > 128: the matrix stays in L2/MALL; one launch per half-iteration (wave per row, coalesced),
// the column pass running on a transposed copy held in the workspace (log domain throughout).
#include <atomic>
#include <cstdlib>
#include "nr_common.h"
#include "nr_finalize.h"
#include "../../include/nr_hip.h"

// sum / max over the 128/EPT adjacent lanes that share a line (8, 4 or 2 lanes)
template <int LPL>
__device__ __forceinline__ float sk_group_sum(float v) {
    v += nr_dpp<NR_DPP_XOR1>(v, v);
    if constexpr (LPL >= 4) v += nr_dpp<NR_DPP_XOR2>(v, v);
    if constexpr (LPL >= 8) v += nr_dpp<NR_DPP_HALF_MIRROR>(v, v);
    return v;
}
template <int LPL>
__device__ __forceinline__ float sk_group_max(float v) {
    v = fmaxf(v, nr_dpp<NR_DPP_XOR1>(v, v));
    if constexpr (LPL >= 4) v = fmaxf(v, nr_dpp<NR_DPP_XOR2>(v, v));
    if constexpr (LPL >= 8) v = fmaxf(v, nr_dpp<NR_DPP_HALF_MIRROR>(v, v));
    return v;
}

// Optional tail (split tail of the loss-only step): the row terms are complete once this launch AND the concurrent
// row-loss launch have finished; every workgroup of both adds to `counter`, the one that arrives last reduces
// rowloss [2,4,B] (= uniform_rows - B) to the five losses and resets the counter.
struct NrSkFinal {
    unsigned int* counter;
    unsigned int total;
    float wu, wn, wkl;
    float* losses;
};

// SK_EPT entries of a line per thread, LPL = 128 / SK_EPT lanes per line, 128 * LPL threads.  Measured (MI355X,
// B = 128, 50 iterations): see the launcher.
template <int SK_EPT>
__global__ __launch_bounds__(128 * (128 / SK_EPT)) void nr_sinkhorn_small_kernel(const float* __restrict__ G, int B, float beta, int iters,
                                                                              float* __restrict__ tgt_rows, float* __restrict__ tgt_cols,
                                                                              float temperature, float* __restrict__ uniform_rows, int uniform_stride,
                                                                              NrSkFinal fin) {
    constexpr int LPL = 128 / SK_EPT;
    NR_CRITICAL_PATH();
    // The scaling vectors live in LDS with every SK_EPT-entry segment shifted by 4 floats: the 4 (2, 8) lanes of a line read
    // DIFFERENT segments with the same ds_read_b128, and unpadded segments start 128 B apart = on the same banks every other
    // segment (a 2-way conflict on every read: 512 of the ~650 cycles of a half-iteration were the LDS array, 64 KiB of
    // redundant reads at 2 x 4 cycles per wave-instruction).  vx(o) = slot of entry o.
    constexpr int SV_N = 128 + 4 * LPL;
    auto vx = [](int o) { return o + 4 * (o / SK_EPT); };
    __shared__ __attribute__((aligned(16))) float s_a[SV_N];
    __shared__ __attribute__((aligned(16))) float s_b[SV_N];
    const int dir = blockIdx.x;                 // 0: problem on G, 1: problem on G^T
    float* tgt = dir == 0 ? tgt_rows : tgt_cols;
    const int tid = threadIdx.x;
    const int line = tid / LPL, sub = tid % LPL;   // `line` = row in the row pass, column in the column pass
    const float norm = -logf((float)(2 * B));
    const float mass = 1.0f / (float)(2 * B);   // e^mu = e^nu
    const bool live = line < B;

    // X[i][j] = G[i][j] (dir 0) or G[j][i] (dir 1);  pr[k] = X[line][EPT sub + k],  pc[k] = X[EPT sub + k][line].
    // G is staged through LDS once: coalesced 16-byte global loads (one round trip), then the row segment and
    // the (strided) column segment of every thread come out of LDS -- 2*EPT scalar global loads per thread, half
    // of them a different cache line per lane, cost ~10 us of the kernel's fixed 20.
    extern __shared__ __attribute__((aligned(16))) float sG[];     // [B][SG_LD]
    constexpr int SG_LD = 129;
    {
        const int n4 = B * B / 4;                                   // B % 4 == 0 is checked by the launcher
        for (int e = tid; e < n4; e += 128 * LPL) {
            const f32x4_t v = *reinterpret_cast<const f32x4_t*>(G + (size_t)e * 4);
            const int i = (e * 4) / B, j = (e * 4) - i * B;
            float* d = sG + i * SG_LD + j;
            d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
        }
    }
    __syncthreads();
    float pr[SK_EPT], pc[SK_EPT];
    const int lc = min(line, B - 1);
#pragma unroll
    for (int k = 0; k < SK_EPT; ++k) {
        int o = sub * SK_EPT + k;
        int oc = min(o, B - 1);
        float xr = dir == 0 ? sG[lc * SG_LD + oc] : sG[oc * SG_LD + lc];
        float xc = dir == 0 ? sG[oc * SG_LD + lc] : sG[lc * SG_LD + oc];
        bool ok = live && o < B;
        pr[k] = ok ? xr : -INFINITY;
        pc[k] = ok ? xc : -INFINITY;
    }
    if (tid < SV_N) { s_a[tid] = 0.f; s_b[tid] = 0.f; }
    __syncthreads();

    if (iters > 0) {
        // ---- iteration 1, log domain: s_a <- u, s_b <- v ------------------------------------------
        {
            float m = -INFINITY;
#pragma unroll
            for (int k = 0; k < SK_EPT; ++k) m = fmaxf(m, pr[k]);                 // v = 0
            m = sk_group_max<LPL>(m);
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < SK_EPT; ++k) s += __expf(pr[k] - m);
            s = sk_group_sum<LPL>(s);
            if (live && sub == 0) s_a[vx(line)] = norm - (m + __logf(s));
        }
        __syncthreads();
        {
            float x[SK_EPT], m = -INFINITY;
#pragma unroll
            for (int k = 0; k < SK_EPT; ++k) {
                x[k] = pc[k] + s_a[vx(sub * SK_EPT) + k];
                m = fmaxf(m, x[k]);
            }
            m = sk_group_max<LPL>(m);
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < SK_EPT; ++k) s += __expf(x[k] - m);
            s = sk_group_sum<LPL>(s);
            if (live && sub == 0) s_b[vx(line)] = norm - (m + __logf(s));
        }
        __syncthreads();
    }
    // ---- the kernel matrix after iteration 1: K = exp(X + u + v)   (u = v = 0 when iters == 0) ---------
    // From here on the plan is P_ij = a_i K_ij b_j with scaling vectors a, b (= 1 now): K stays in
    // registers untouched, one half-iteration is  a_i = e^mu / sum_j K_ij b_j  (16 FMAs per thread).
    {
        const float ul = live ? s_a[vx(line)] : 0.f, vl = live ? s_b[vx(line)] : 0.f;
#pragma unroll
        for (int k = 0; k < SK_EPT; ++k) {
            int o = sub * SK_EPT + k;
            pr[k] = __expf(pr[k] + ul + s_b[vx(o)]);      // -inf entries -> 0
            pc[k] = __expf(pc[k] + s_a[vx(o)] + vl);
        }
    }
    __syncthreads();
    if (tid < SV_N) { s_a[tid] = 1.f; s_b[tid] = 1.f; }
    __syncthreads();
    float a_own = 1.f, b_own = 1.f;
    for (int it = 1; it < iters; ++it) {
        {   // a_i = e^mu / sum_j K_ij b_j
            float r0 = 0.f, r1 = 0.f, r2 = 0.f, r3 = 0.f;
#pragma unroll
            for (int k = 0; k < SK_EPT; k += 4) {
                f32x4_t f = *reinterpret_cast<const f32x4_t*>(&s_b[vx(sub * SK_EPT) + k]);
                r0 += pr[k] * f[0]; r1 += pr[k + 1] * f[1]; r2 += pr[k + 2] * f[2]; r3 += pr[k + 3] * f[3];
            }
            float r = sk_group_sum<LPL>((r0 + r1) + (r2 + r3));
            a_own = live ? mass * __builtin_amdgcn_rcpf(r) : 0.f;
            if (sub == 0) s_a[vx(line)] = a_own;
        }
        __syncthreads();
        {   // b_j = e^nu / sum_i K_ij a_i
            float c0 = 0.f, c1 = 0.f, c2 = 0.f, c3 = 0.f;
#pragma unroll
            for (int k = 0; k < SK_EPT; k += 4) {
                f32x4_t f = *reinterpret_cast<const f32x4_t*>(&s_a[vx(sub * SK_EPT) + k]);
                c0 += pc[k] * f[0]; c1 += pc[k + 1] * f[1]; c2 += pc[k + 2] * f[2]; c3 += pc[k + 3] * f[3];
            }
            float c = sk_group_sum<LPL>((c0 + c1) + (c2 + c3));
            b_own = live ? mass * __builtin_amdgcn_rcpf(c) : 0.f;
            if (sub == 0) s_b[vx(line)] = b_own;
        }
        __syncthreads();
    }
    // ---- Q = P / e^norm = 2B * a_i K_ij b_j;  target = beta*Q + (1-beta)*I ----------------------------
    // Optionally also the uniform-regularisation row term itself (until_module.py:285-289 on these targets):
    //   u_i = -sum_j tgt_ij * (T * X_ij - LSE_j(T * X_ij)),   X = G (direction 0) or G^T (direction 1),
    // written to uniform_rows[dir][i] -- the row-loss kernel then does not depend on the Sinkhorn solve at all
    // and runs beside it.  (tgt may then be NULL: nobody else reads the targets in the loss-only step.)
    if (uniform_rows) {
        float xg[SK_EPT], m = -INFINITY;
#pragma unroll
        for (int k = 0; k < SK_EPT; ++k) {
            const int o = min(sub * SK_EPT + k, B - 1);
            xg[k] = (dir == 0 ? sG[lc * SG_LD + o] : sG[o * SG_LD + lc]) * temperature;
            if (live && sub * SK_EPT + k < B) m = fmaxf(m, xg[k]);
        }
        m = sk_group_max<LPL>(m);
        float se = 0.f;
#pragma unroll
        for (int k = 0; k < SK_EPT; ++k)
            if (live && sub * SK_EPT + k < B) se += __expf(xg[k] - m);
        se = sk_group_sum<LPL>(se);
        const float lse_u = m + __logf(se);
        const float sc = beta * (float)(2 * B) * a_own;
        float u = 0.f;
#pragma unroll
        for (int k = 0; k < SK_EPT; ++k) {
            const int o = sub * SK_EPT + k;
            if (live && o < B) {
                const float t = sc * pr[k] * s_b[vx(o)] + (o == line ? 1.0f - beta : 0.f);
                u -= t * (xg[k] - lse_u);
            }
        }
        u = sk_group_sum<LPL>(u);
        if (live && sub == 0) __hip_atomic_store(uniform_rows + (size_t)dir * uniform_stride + line, u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // `sc1`: read by the finalizing workgroup
    }
    if (live && tgt) {
        const float sc = beta * (float)(2 * B) * a_own;
#pragma unroll
        for (int k = 0; k < SK_EPT; k += 4) {             // 16-byte stores (B % 4 == 0)
            const int o = sub * SK_EPT + k;
            if (o < B) {
                f32x4_t q;
#pragma unroll
                for (int e = 0; e < 4; ++e) q[e] = sc * pr[k + e] * s_b[vx(o) + e] + (o + e == line ? 1.0f - beta : 0.f);
                *reinterpret_cast<f32x4_t*>(tgt + (size_t)line * B + o) = q;
            }
        }
    }
    if (fin.counter == nullptr) return;
    __shared__ int s_last;                         // hand-off by sc1 stores / loads, no cache-wide fence: see nr_rowloss.hip
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) s_last = __hip_atomic_fetch_add(fin.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == fin.total - 1;
    __syncthreads();
    if (!s_last) return;
    nr_loss_finalize_body<true>(uniform_rows - B, B, fin.wu, fin.wn, fin.wkl, fin.losses);
    if (threadIdx.x == 0) __hip_atomic_store(fin.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- large-B path ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void nr_transpose_kernel(const float* __restrict__ in, int B, float* __restrict__ out) {
    __shared__ float t[32][33];
    int x = blockIdx.x * 32 + (threadIdx.x & 31), y0 = blockIdx.y * 32;
    for (int r = threadIdx.x >> 5; r < 32; r += 8)
        if (x < B && y0 + r < B) t[r][threadIdx.x & 31] = in[(size_t)(y0 + r) * B + x];
    __syncthreads();
    int ox = blockIdx.y * 32 + (threadIdx.x & 31), oy0 = blockIdx.x * 32;
    for (int r = threadIdx.x >> 5; r < 32; r += 8)
        if (ox < B && oy0 + r < B) out[(size_t)(oy0 + r) * B + ox] = t[threadIdx.x & 31][r];
}

// dual_out[dir][i] = norm - LSE_j(M_dir[i][j] + dual_in[dir][j]);  M_0 = X0, M_1 = X1.
// blockIdx.y = direction; one wave per row.
__global__ __launch_bounds__(256) void nr_sinkhorn_rowlse_kernel(const float* __restrict__ X0, const float* __restrict__ X1,
                                                                 int B, float norm, const float* __restrict__ dual_in,
                                                                 float* __restrict__ dual_out) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int dir = blockIdx.y;
    if (i >= B) return;
    const float* row = (dir == 0 ? X0 : X1) + (size_t)i * B;
    const float* din = dual_in + (size_t)dir * B;
    float m = -INFINITY, s = 0.f;
    for (int j = lane; j < B; j += 64) {
        float x = row[j] + din[j];
        float mn = fmaxf(m, x);
        s = s * __expf(m - mn) + __expf(x - mn);
        m = mn;
    }
    float mw = nr_wave_max(m);
    s *= __expf(m - mw);
    s = nr_wave_sum(s);
    if (lane == 0) dual_out[(size_t)dir * B + i] = norm - (mw + __logf(s));
}

__global__ __launch_bounds__(256) void nr_sinkhorn_plan_kernel(const float* __restrict__ G, const float* __restrict__ GT,
                                                               int B, float norm, float beta, const float* __restrict__ u,
                                                               const float* __restrict__ v, float* __restrict__ tgt_rows,
                                                               float* __restrict__ tgt_cols) {
    const int dir = blockIdx.y;
    const float* X = dir == 0 ? G : GT;
    float* tgt = dir == 0 ? tgt_rows : tgt_cols;
    size_t n = (size_t)B * B;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (size_t)gridDim.x * 256) {
        int i = (int)(idx / B), j = (int)(idx - (size_t)i * B);
        float q = __expf(X[idx] + u[(size_t)dir * B + i] + v[(size_t)dir * B + j] - norm);
        tgt[idx] = beta * q + (i == j ? 1.0f - beta : 0.f);
    }
}

// ---- 128 < B <= 1024 (B % 64 == 0): one cooperative launch instead of 2 x iters + 2 --------------------------------------------
// The large-B path above is launch-bound (103 graph nodes: 300 / 420 / 669 us at B = 256 / 512 / 1024 for 50 iterations).
// Here B/32 workgroups per direction keep their 32 rows AND their 32 columns of the kernel matrix in registers (wave w: 4
// lines, B/64 entries per lane each), exchange the scaling vectors through two [B] arrays in global memory (`sc1` stores and
// loads) and meet at a counter barrier after every half-iteration.  Only blockIdx % 8 in {0, 1} works -- direction 0 on one
// XCD, direction 1 on another (round-robin placement: the barrier's atomics and the vectors then stay inside one L2; a
// different placement is slower, never wrong).  Every spin is bounded: if the workgroups of a direction cannot become
// co-resident (something else holds the CUs for good) the kernel gives up and poisons its targets with NaN instead of hanging.
struct NrSkCoopArgs {
    const float* G;
    int B, iters;
    float beta;
    float *tgt_rows, *tgt_cols;
    float* vec;                  // [2 dir][2][B] scaling vectors
    unsigned int* counter;       // [2 dir][32] (one 128-byte line each), zero on entry
    int spread;                  // 0: a direction's workgroups on ONE XCD; 1: on two (see the launcher)
};

// EPL = entries per lane = B / 64.  PC_LDS: the column-major copy of the kernel matrix lives in dynamic LDS (4 EPL floats per
// thread, [4 EPL][512]: conflict-free) instead of registers -- from EPL = 13 (B >= 832) two register copies of 4 EPL entries plus
// the scaling-vector slice no longer fit the 256 registers a lane has at two waves per SIMD (69 spilled registers at EPL = 16
// in round 3); the b-update reads it once per iteration, next to a ~2 us barrier.
template <int EPL, bool PC_LDS = (EPL >= 13)>
__global__ __launch_bounds__(512) void nr_sinkhorn_coop_kernel(NrSkCoopArgs p) {
    extern __shared__ __attribute__((aligned(16))) float sk_pc_lds[];
    NR_CRITICAL_PATH();
    const int xcd = blockIdx.x & 7;
    if (xcd > (p.spread ? 3 : 1)) return;
    const int dir = xcd & 1;
    const int wg = p.spread ? 2 * (blockIdx.x >> 3) + (xcd >> 1) : (blockIdx.x >> 3), B = p.B, nwg = B / 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* va = p.vec + (size_t)dir * 2 * B;          // u / a
    float* vb = va + B;                               // v / b
    unsigned int* cnt = p.counter + dir * 32;
    unsigned int* dead_flag = p.counter + 16;         // shared by both directions: one timeout ends every workgroup of the launch
    __shared__ int s_dead;
    if (tid == 0) s_dead = 0;
    __syncthreads();                                  // every wave reads s_dead at the top of its first barrier(): LDS is not zeroed between workgroups
    const float norm = -logf((float)(2 * B));
    const float mass = 1.0f / (float)(2 * B);
    // X[i][j] = G[i][j] (dir 0) or G[j][i] (dir 1);  pr(q, e) = X[line q][64 e + lane],  pc(q, e) = X[64 e + lane][line q]
    const int l0 = 32 * wg + 4 * wave;
    float pr[4][EPL], pc_reg[PC_LDS ? 1 : 4][PC_LDS ? 1 : EPL];
    auto pc = [&](int q, int e) -> float& {
        if constexpr (PC_LDS) return sk_pc_lds[(q * EPL + e) * 512 + tid];
        else return pc_reg[q][e];
    };
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            const size_t rowmaj = (size_t)(l0 + q) * B + e * 64 + lane, colmaj = (size_t)(e * 64 + lane) * B + l0 + q;
            pr[q][e] = p.G[dir == 0 ? rowmaj : colmaj];
            pc(q, e) = p.G[dir == 0 ? colmaj : rowmaj];
        }
    unsigned int phase = 0;
    // All stores of this workgroup visible, then every workgroup of the direction.  The workgroups of a direction must be
    // resident together (the host gates the launch on that: nr_sinkhorn_cooperative_ok); should they not be -- a CU mask, another
    // process holding the CUs -- the spin is BOUNDED, and the first workgroup to run out of patience raises the launch-wide
    // dead flag: every other workgroup sees it at its next poll, every later barrier returns at once, and BOTH targets are
    // poisoned with NaN in full (the losses downstream turn NaN: a loud failure instead of finite garbage).
    auto barrier = [&]() {
        ++phase;
        if (s_dead) return;                           // (uniform: set in front of the previous barrier's closing __syncthreads)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned int target = phase * (unsigned)nwg;
            int spins = 0;
            while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(2);
                if ((spins & 63) == 63 && __hip_atomic_load(dead_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { s_dead = 1; break; }
                if (++spins > (1 << 20)) {            // a fraction of a second: never hang the chip
                    __hip_atomic_store(dead_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    s_dead = 1;
                    break;
                }
            }
        }
        __syncthreads();
    };
    auto put = [&](float* v, int q, float x) { if (lane == 0) __hip_atomic_store(v + l0 + q, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    auto get = [&](const float* v, int e) { return __hip_atomic_load(v + e * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    float a_own[4] = {1.f, 1.f, 1.f, 1.f};
    float u_own[4] = {0.f, 0.f, 0.f, 0.f}, v_own[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.iters > 0) {
        // iteration 1 in the log domain, exactly as the reference (until_module.py:249-258): u = norm - LSE_j(X), v = norm - LSE_i(X + u)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float m = -INFINITY;
#pragma unroll
            for (int e = 0; e < EPL; ++e) m = fmaxf(m, pr[q][e]);
            m = nr_wave_max(m);
            float t = 0.f;
#pragma unroll
            for (int e = 0; e < EPL; ++e) t += __expf(pr[q][e] - m);
            u_own[q] = norm - (m + __logf(nr_wave_sum(t)));
            put(va, q, u_own[q]);
        }
        barrier();
        {
            float mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
                const float uu = get(va, e);
#pragma unroll
                for (int q = 0; q < 4; ++q) { const float x = pc(q, e) + uu; pc(q, e) = x; mx[q] = fmaxf(mx[q], x); }      // pc <- X + u
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float m = nr_wave_max(mx[q]);
                float t = 0.f;
#pragma unroll
                for (int e = 0; e < EPL; ++e) t += __expf(pc(q, e) - m);
                v_own[q] = norm - (m + __logf(nr_wave_sum(t)));
                put(vb, q, v_own[q]);
            }
        }
        barrier();
        // the kernel matrix K = exp(X + u + v): fixed from here on, the plan is a_i K_ij b_j
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            const float vv = get(vb, e);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                pr[q][e] = __expf(pr[q][e] + u_own[q] + vv);
                pc(q, e) = __expf(pc(q, e) + v_own[q]);
            }
        }
        // a = b = 1: publish b = 1 through vb's slots only after everybody has read v from them
        barrier();
#pragma unroll
        for (int q = 0; q < 4; ++q) put(vb, q, 1.0f);
        barrier();
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < EPL; ++e) { pr[q][e] = __expf(pr[q][e]); pc(q, e) = __expf(pc(q, e)); }
#pragma unroll
        for (int q = 0; q < 4; ++q) put(vb, q, 1.0f);
        barrier();
    }
    for (int it = 1; it < p.iters; ++it) {
        {   // a_i = e^mu / sum_j K_ij b_j
            float bb[EPL];
#pragma unroll
            for (int e = 0; e < EPL; ++e) bb[e] = get(vb, e);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float r = 0.f;
#pragma unroll
                for (int e = 0; e < EPL; ++e) r += pr[q][e] * bb[e];
                a_own[q] = mass * __builtin_amdgcn_rcpf(nr_wave_sum(r));
                put(va, q, a_own[q]);
            }
        }
        barrier();
        {   // b_j = e^nu / sum_i K_ij a_i
            float aa[EPL];
#pragma unroll
            for (int e = 0; e < EPL; ++e) aa[e] = get(va, e);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float c = 0.f;
#pragma unroll
                for (int e = 0; e < EPL; ++e) c += pc(q, e) * aa[e];
                put(vb, q, mass * __builtin_amdgcn_rcpf(nr_wave_sum(c)));
            }
        }
        barrier();
    }
    // target = beta * Q + (1 - beta) I,  Q = 2B a_i K_ij b_j
    float* tgt = dir == 0 ? p.tgt_rows : p.tgt_cols;
    // (a direction that completed while the other one gave up is poisoned too: the flag is the launch's)
    const bool dead = s_dead != 0 || __hip_atomic_load(dead_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        const float bbv = get(vb, e);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int j = e * 64 + lane;
            const float t = p.beta * (float)(2 * B) * a_own[q] * pr[q][e] * bbv + (j == l0 + q ? 1.0f - p.beta : 0.f);
            tgt[(size_t)(l0 + q) * B + j] = dead ? __builtin_nanf("") : t;
        }
    }
}

extern "C" size_t nr_sinkhorn_workspace_bytes(int B) {
    if (B <= 128 && (B % 4) == 0) return 16;
    // (the cooperative form needs 4 B floats + 256 bytes of counters: less than the multi-launch form, which stays the fallback)
    return ((size_t)B * B + 4 * (size_t)B) * sizeof(float) + 64 + 256;
}

static int nr_sinkhorn_run(const float* G, int B, float beta, int iters, float* tgt_rows, float* tgt_cols, float temperature,
                           float* uniform_rows, int uniform_stride, void* workspace, void* stream,
                           NrSkFinal fin = NrSkFinal{nullptr, 0u, 0.f, 0.f, 0.f, nullptr}, bool allow_coop = true);

// dynamic LDS of the cooperative kernel: the column-major matrix copy of the wide variants (EPL >= 13), 4 EPL floats per thread
static size_t nr_sinkhorn_coop_lds(int B) { return B / 64 >= 13 ? (size_t)4 * (B / 64) * 512 * sizeof(float) : 0; }

static const void* nr_sinkhorn_coop_fn(int B) {
    switch (B / 64) {
#define NR_SKC(E_) case E_: return (const void*)nr_sinkhorn_coop_kernel<E_>;
        NR_SKC(3) NR_SKC(4) NR_SKC(5) NR_SKC(6) NR_SKC(7) NR_SKC(8) NR_SKC(9) NR_SKC(10) NR_SKC(11) NR_SKC(12) NR_SKC(13) NR_SKC(14)
        NR_SKC(15) NR_SKC(16)
#undef NR_SKC
        default: return nullptr;
    }
}

// The cooperative form's host gate, as a pure function of what the device reports (unit-tested on the CPU through
// nr_sinkhorn_cooperative_gate): B/32 workgroups per direction, one direction per XCD under round-robin placement, so the
// direction's workgroups are co-resident iff  blocks/CU x CUs of ONE XCD >= B/32; and both directions fit the chip anyway.
extern "C" int nr_sinkhorn_cooperative_gate(int B, int blocks_per_cu, int n_cus, int n_xcd) {
    if (B <= 128 || B > 1024 || (B % 64) != 0 || blocks_per_cu <= 0 || n_cus <= 0 || n_xcd <= 0) return 0;
    const int need = B / 32;
    return (blocks_per_cu * (n_cus / n_xcd) >= need && blocks_per_cu * n_cus >= 2 * need) ? 1 : 0;
}

// 1 when nr_sinkhorn_targets runs this B as ONE cooperative launch on the current device, 0 when it takes the multi-launch
// form (B outside 192..1024 / not a multiple of 64, or the device cannot hold a direction's workgroups together: a CU-masked
// or partitioned device).  Queried once per B (occupancy of the kernel variant x the device's CU count).
extern "C" int nr_sinkhorn_cooperative_ok(int B) {
    // by device and B / 64: 0 unknown, 1 no, 2 yes.  Per DEVICE: the occupancy answer and the dynamic-LDS grant
    // (hipFuncSetAttribute applies to the current device's copy of the kernel) are both per device.  Words are written
    // whole with the same value by whoever asks first: two host threads racing here compute the same answer.
    constexpr int kMaxDev = 64;
    static std::atomic<int> cache[kMaxDev][17];
    if (B <= 128 || B > 1024 || (B % 64) != 0) return 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) return 0;
    std::atomic<int>* slot = dev < kMaxDev ? &cache[dev][B / 64] : nullptr;
    int c = slot ? slot->load(std::memory_order_acquire) : 0;
    if (c == 0) {
        int cus = 0, per_cu = 0;
        const void* fn = nr_sinkhorn_coop_fn(B);
        const size_t lds = nr_sinkhorn_coop_lds(B);
        bool ok = fn && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess &&
                  (lds <= 64 * 1024 || hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess) &&
                  hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 512, lds) == hipSuccess;
        // MI355X in SPX mode: 256 CUs in 8 XCDs; a partition / mask reports fewer CUs and is priced as ONE XCD's worth per 32
        const int n_xcd = cus >= 64 ? 8 : 1;
        c = (ok && nr_sinkhorn_cooperative_gate(B, per_cu, cus, n_xcd)) ? 2 : 1;
        if (slot) slot->store(c, std::memory_order_release);
    }
    return c == 2;
}

// The multi-launch form (2 x iters + 3 launches) on its own: what nr_sinkhorn_targets falls back to when the gate says no.
extern "C" int nr_sinkhorn_targets_multilaunch(const float* G, int B, float beta, int iters, float* tgt_rows, float* tgt_cols,
                                               void* workspace, void* stream) {
    if (!tgt_rows || !tgt_cols) return NR_EINVAL;
    if (B <= 128 && (B % 4) == 0) return NR_EUNSUPPORTED;      // (the one-workgroup form owns these sizes; its workspace is 16 bytes)
    return nr_sinkhorn_run(G, B, beta, iters, tgt_rows, tgt_cols, 0.f, nullptr, 0, workspace, stream,
                           NrSkFinal{nullptr, 0u, 0.f, 0.f, 0.f, nullptr}, false);
}

extern "C" int nr_sinkhorn_targets(const float* G, int B, float beta, int iters, float* tgt_rows, float* tgt_cols,
                                   void* workspace, void* stream) {
    if (!tgt_rows || !tgt_cols) return NR_EINVAL;
    return nr_sinkhorn_run(G, B, beta, iters, tgt_rows, tgt_cols, 0.f, nullptr, 0, workspace, stream);
}

extern "C" int nr_sinkhorn_uniform_rows(const float* G, int B, float beta, int iters, float temperature, float* uniform_rows,
                                        int uniform_dir_stride, float* tgt_rows, float* tgt_cols, void* workspace, void* stream) {
    if (!uniform_rows || B > 128 || (B % 4) != 0) return uniform_rows ? NR_EUNSUPPORTED : NR_EINVAL;
    if ((tgt_rows == nullptr) != (tgt_cols == nullptr) || uniform_dir_stride < B) return NR_EINVAL;
    return nr_sinkhorn_run(G, B, beta, iters, tgt_rows, tgt_cols, temperature, uniform_rows, uniform_dir_stride, workspace, stream);
}

// nr_sinkhorn_uniform_rows writing into rowloss [2,4,B] (rows 1) and taking part in the shared finalize: see NrSkFinal
extern "C" int nr_sinkhorn_uniform_rows_final(const float* G, int B, float beta, int iters, float temperature, float* rowloss,
                                              uint32_t* counter, float uniform_weight, float neighbor_weight, float kl_weight,
                                              float* losses, void* workspace, void* stream) {
    if (!rowloss || !counter || !losses || B > 128 || (B % 4) != 0) return (rowloss && counter && losses) ? NR_EUNSUPPORTED : NR_EINVAL;
    NrSkFinal fin{counter, (unsigned)(2 + 2 * ((B + 3) / 4)), uniform_weight, neighbor_weight, kl_weight, losses};
    return nr_sinkhorn_run(G, B, beta, iters, nullptr, nullptr, temperature, rowloss + B, 4 * B, workspace, stream, fin);
}

static int nr_sinkhorn_run(const float* G, int B, float beta, int iters, float* tgt_rows, float* tgt_cols, float temperature,
                           float* uniform_rows, int uniform_stride, void* workspace, void* stream, NrSkFinal fin, bool allow_coop) {
    if (!G || B <= 0 || iters < 0) return NR_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (B <= 128 && (B % 4) == 0) {
        // Nearly the whole LDS of the CU is requested (156 of 160 KiB; the static arrays take the rest), although B x 129 floats would do: the solve
        // is a latency chain on ONE workgroup per direction, and no LDS-using workgroup of another kernel (scorer, bank
        // products run beside it in the step) can then be placed on its CU to compete for issue slots and the LDS.
        // (never more than the device grants a block, never less than the B x 129 floats the solve needs)
        static int lds_cap = -1;
        if (lds_cap < 0) {
            int dev = 0, v = 0;
            if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess) v = 64 * 1024;
            lds_cap = v;
        }
        const size_t need = (size_t)B * 129 * sizeof(float);
        size_t lds = 156 * 1024;                         // + the kernel's static arrays (< 4 KiB)
        if (lds + 4096 > (size_t)lds_cap) lds = lds_cap > 4096 ? (size_t)lds_cap - 4096 : 0;
        if (lds < need) lds = need;
        if (need + 4096 > (size_t)lds_cap) return NR_EUNSUPPORTED;
        // entries per thread: 16 (1024 threads) / 32 (512) / 64 (256); NR_SINKHORN_EPT overrides (tuning hook)
        int ept = 32;
        static size_t attr_set[3] = {0, 0, 0};             // per kernel variant: the dynamic-LDS limit already granted
#ifdef NR_TUNE
        if (const char* e = nr_tune_env("NR_SINKHORN_EPT")) ept = atoi(e);
#endif
        size_t& granted = attr_set[ept == 16 ? 0 : ept == 64 ? 2 : 1];
        if (lds > 64 * 1024 && lds > granted) {
            const void* k = (const void*)nr_sinkhorn_small_kernel<32>;
#ifdef NR_TUNE
            if (ept == 16) k = (const void*)nr_sinkhorn_small_kernel<16>;
            if (ept == 64) k = (const void*)nr_sinkhorn_small_kernel<64>;
#endif
            hipError_t er = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (er != hipSuccess) return (int)er;
            granted = lds;
        }
#ifdef NR_TUNE
        if (ept == 16) hipLaunchKernelGGL(nr_sinkhorn_small_kernel<16>, dim3(2), dim3(1024), lds, st, G, B, beta, iters, tgt_rows, tgt_cols, temperature, uniform_rows, uniform_stride, fin);
        else if (ept == 64) hipLaunchKernelGGL(nr_sinkhorn_small_kernel<64>, dim3(2), dim3(256), lds, st, G, B, beta, iters, tgt_rows, tgt_cols, temperature, uniform_rows, uniform_stride, fin);
        else
#endif
        hipLaunchKernelGGL(nr_sinkhorn_small_kernel<32>, dim3(2), dim3(512), lds, st, G, B, beta, iters, tgt_rows, tgt_cols, temperature, uniform_rows, uniform_stride, fin);
        NR_LAUNCH_CHECK();
        return NR_OK;
    }
    if (!workspace || !tgt_rows || !tgt_cols || uniform_rows) return NR_EINVAL;
    if (allow_coop && nr_sinkhorn_cooperative_ok(B)) {
        // cooperative form: vectors [2][2][B] f32, then the counters (zeroed by a memset node in front of the launch)
        float* vec = reinterpret_cast<float*>(workspace);
        unsigned int* counter = reinterpret_cast<unsigned int*>(vec + 4 * (size_t)B);
        hipError_t e = hipMemsetAsync(counter, 0, 256, st);
        if (e != hipSuccess) return (int)e;
        // A direction's B/32 workgroups on ONE XCD keep the barrier's atomics and the scaling vectors inside one L2 -- and hold
        // that XCD's CUs (131 KB of LDS each from B = 832) for the whole solve: workgroups are dispatched in order, so a kernel
        // on another queue whose next workgroup is due on a full XCD waits there with everything behind it (configs[2] step:
        // the bank chains' 285 us of kernels stood still for the solve's 320).  From 16 workgroups per direction on they are
        // spread over TWO XCDs each (16 CUs of four XCDs at B = 1024): the solve alone is slower, the step is not.
        const int spread = (B / 32) >= 16 && ((B / 32) % 2) == 0 && !nr_tune_env("NR_SK_NOSPREAD");
        NrSkCoopArgs a{G, B, iters, beta, tgt_rows, tgt_cols, vec, counter, spread};
        const dim3 grid(spread ? 8 * (B / 64) : 8 * (B / 32));
        switch (B / 64) {
#define NR_SKC(E_) case E_: hipLaunchKernelGGL(nr_sinkhorn_coop_kernel<E_>, grid, dim3(512), nr_sinkhorn_coop_lds(B), st, a); break;
            NR_SKC(3) NR_SKC(4) NR_SKC(5) NR_SKC(6) NR_SKC(7) NR_SKC(8) NR_SKC(9) NR_SKC(10) NR_SKC(11) NR_SKC(12) NR_SKC(13) NR_SKC(14)
            NR_SKC(15) NR_SKC(16)
#undef NR_SKC
            default: return NR_EUNSUPPORTED;
        }
        NR_LAUNCH_CHECK();
        return NR_OK;
    }
    float* GT = reinterpret_cast<float*>(workspace);
    float* u = GT + (size_t)B * B;     // [2][B]
    float* v = u + 2 * (size_t)B;      // [2][B]
    const float norm = -logf((float)(2 * B));
    dim3 tg((B + 31) / 32, (B + 31) / 32);
    hipLaunchKernelGGL(nr_transpose_kernel, tg, dim3(256), 0, st, G, B, GT);
    hipError_t e = hipMemsetAsync(u, 0, 4 * (size_t)B * sizeof(float), st);
    if (e != hipSuccess) return (int)e;
    dim3 rg((B + 3) / 4, 2);
    for (int it = 0; it < iters; ++it) {
        // dir 0: u from rows of G with v; dir 1: u from rows of G^T with v
        hipLaunchKernelGGL(nr_sinkhorn_rowlse_kernel, rg, dim3(256), 0, st, G, GT, B, norm, v, u);
        // dir 0: v from columns of G = rows of G^T with u; dir 1: rows of G
        hipLaunchKernelGGL(nr_sinkhorn_rowlse_kernel, rg, dim3(256), 0, st, GT, G, B, norm, u, v);
    }
    hipLaunchKernelGGL(nr_sinkhorn_plan_kernel, dim3(1024, 2), dim3(256), 0, st, G, GT, B, norm, beta, u, v, tgt_rows,
                       tgt_cols);
    NR_LAUNCH_CHECK();
    return NR_OK;
}
