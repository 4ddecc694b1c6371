// Token preparation: F.normalize + mask + bf16 hi/lo split (+ column sums for the centrality
// weights).  Reference: modeling.py:495-496 (normalize), :500-501 (mask multiplies),
// :413-424 (mean of the normalised tokens, padding included).
// HBM-bound streaming kernel: one wave per token row, 16-byte loads, 8-byte bf16x4 stores.
#include "nr_common.h"
#include "../../include/nr_hip.h"

#ifndef NR_PREP_MAX_PARTS
#define NR_PREP_MAX_PARTS 256    // column-sum partials = workgroups of a launch that wants them: 16 tokens each up to this many (64 left a
                                 // 3072-token set with 12 dependent row round trips per wave: 14 us; 256: 4)
#endif
#define NR_PREP_MAX_GRID 2048
#define NR_PREP_MAX_CHUNKS 4   // d <= 1024

extern "C" int nr_prepare_parts(int n_tok) {
    int p = (n_tok + 15) / 16;
    if (p < 1) p = 1;
    return p > NR_PREP_MAX_PARTS ? NR_PREP_MAX_PARTS : p;
}

// One tensor's share of a launch: workgroup `bid` of `nblocks`.
template <int CH>   // CH = d / 256 float4 chunks per lane
__device__ __forceinline__ void nr_prepare_body(const float* __restrict__ x, const float* __restrict__ mask,
                                                int n_tok, int d, int normalize, uint16_t* __restrict__ hi,
                                                uint16_t* __restrict__ lo, float* __restrict__ norm_out,
                                                float* __restrict__ colsum_part, const int bid, const int nblocks) {
    __shared__ float s_col[4][CH * 256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float csum[CH][4];
#pragma unroll
    for (int c = 0; c < CH; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) csum[c][e] = 0.f;

    // A wave's rows in batches of R: the loads of a whole batch go out before the first row is reduced -- one round trip to
    // memory per batch instead of one per row (3072 tokens on 192 workgroups: four dependent trips per wave, 8.4 us alone, were
    // the launch; the arithmetic and the order the column sums add rows in are unchanged)
    constexpr int R = CH <= 2 ? 4 : 2;
    const int stride = nblocks * 4;
    for (int row0 = bid * 4 + wave; row0 < n_tok; row0 += stride * R) {
        f32x4_t v[R][CH];
        float mk[R];
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const int row = row0 + i * stride;
            if (row < n_tok) {
                const float* xr = x + (size_t)row * d;
#pragma unroll
                for (int c = 0; c < CH; ++c) v[i][c] = *reinterpret_cast<const f32x4_t*>(xr + (c * 64 + lane) * 4);
                mk[i] = mask ? mask[row] : 1.0f;
            }
        }
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const int row = row0 + i * stride;
            if (row >= n_tok) break;
            float ss = 0.f;
#pragma unroll
            for (int c = 0; c < CH; ++c)
#pragma unroll
                for (int e = 0; e < 4; ++e) ss += v[i][c][e] * v[i][c][e];
            ss = nr_wave_sum(ss);
            float nrm = fmaxf(sqrtf(ss), 1e-12f);          // F.normalize: x / max(||x||, eps)
            float inv = normalize ? 1.0f / nrm : 1.0f;
            if (norm_out && lane == 0) norm_out[row] = nrm;
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                uint16_t h[4], l[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float xn = v[i][c][e] * inv;
                    csum[c][e] += xn;                       // unmasked (padding tokens count)
                    float y = xn * mk[i];
                    h[e] = nr_f2bf(y);
                    l[e] = nr_f2bf(y - nr_bf2f(h[e]));
                }
                size_t o = (size_t)row * d + (c * 64 + lane) * 4;
                uint2 ph = make_uint2((uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16));
                *reinterpret_cast<uint2*>(hi + o) = ph;
                if (lo) {
                    uint2 pl = make_uint2((uint32_t)l[0] | ((uint32_t)l[1] << 16), (uint32_t)l[2] | ((uint32_t)l[3] << 16));
                    *reinterpret_cast<uint2*>(lo + o) = pl;
                }
            }
        }
    }
    if (colsum_part) {
#pragma unroll
        for (int c = 0; c < CH; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) s_col[wave][(c * 64 + lane) * 4 + e] = csum[c][e];
        __syncthreads();
        for (int k = threadIdx.x; k < d; k += 256)
            colsum_part[(size_t)bid * d + k] = (s_col[0][k] + s_col[1][k]) + (s_col[2][k] + s_col[3][k]);
    }
}

template <int CH>
__global__ __launch_bounds__(256) void nr_prepare_kernel(const float* __restrict__ x, const float* __restrict__ mask,
                                                         int n_tok, int d, int normalize, uint16_t* __restrict__ hi,
                                                         uint16_t* __restrict__ lo, float* __restrict__ norm_out,
                                                         float* __restrict__ colsum_part) {
    nr_prepare_body<CH>(x, mask, n_tok, d, normalize, hi, lo, norm_out, colsum_part, blockIdx.x, gridDim.x);
}

// Two tensors (the batch's text and video tokens) in ONE launch: workgroups [0, nb0) take the first, the rest the second.
struct NrPrepareOne {
    const float *x, *mask;
    int n_tok;
    uint16_t *hi, *lo;
    float *norm, *colsum_part;
};
template <int CH>
__global__ __launch_bounds__(256) void nr_prepare_pair_kernel(NrPrepareOne a, NrPrepareOne b, int d, int normalize, int nb0) {
    const bool second = (int)blockIdx.x >= nb0;
    const NrPrepareOne& p = second ? b : a;
    nr_prepare_body<CH>(p.x, p.mask, p.n_tok, d, normalize, p.hi, p.lo, p.norm, p.colsum_part, second ? blockIdx.x - nb0 : blockIdx.x,
                        second ? gridDim.x - nb0 : nb0);
}

extern "C" int nr_prepare_tokens(const float* x, const float* mask, int n_tok, int d, int normalize, uint16_t* hi,
                                 uint16_t* lo, float* norm, float* colsum_part, void* stream) {
    if (!x || !hi || n_tok <= 0 || d <= 0) return NR_EINVAL;
    if ((d % 256) != 0 || d / 256 > NR_PREP_MAX_CHUNKS) return NR_EUNSUPPORTED;
    // column sums wanted: one partial per workgroup, so few workgroups (the batch tensors are small);
    // otherwise (memory bank) enough workgroups to stream at HBM rate
    int grid = nr_prepare_parts(n_tok);
    if (!colsum_part) {
        grid = (n_tok + 15) / 16;
        if (grid > NR_PREP_MAX_GRID) grid = NR_PREP_MAX_GRID;
    }
    hipStream_t st = (hipStream_t)stream;
    switch (d / 256) {
        case 1: hipLaunchKernelGGL(nr_prepare_kernel<1>, dim3(grid), dim3(256), 0, st, x, mask, n_tok, d, normalize, hi, lo, norm, colsum_part); break;
        case 2: hipLaunchKernelGGL(nr_prepare_kernel<2>, dim3(grid), dim3(256), 0, st, x, mask, n_tok, d, normalize, hi, lo, norm, colsum_part); break;
        case 3: hipLaunchKernelGGL(nr_prepare_kernel<3>, dim3(grid), dim3(256), 0, st, x, mask, n_tok, d, normalize, hi, lo, norm, colsum_part); break;
        default: hipLaunchKernelGGL(nr_prepare_kernel<4>, dim3(grid), dim3(256), 0, st, x, mask, n_tok, d, normalize, hi, lo, norm, colsum_part); break;
    }
    NR_LAUNCH_CHECK();
    return NR_OK;
}

extern "C" int nr_prepare_tokens_pair(const float* x0, const float* mask0, int n_tok0, uint16_t* hi0, uint16_t* lo0, float* norm0,
                                      float* colsum_part0, const float* x1, const float* mask1, int n_tok1, uint16_t* hi1,
                                      uint16_t* lo1, float* norm1, float* colsum_part1, int d, int normalize, void* stream) {
    if (!x0 || !hi0 || !x1 || !hi1 || n_tok0 <= 0 || n_tok1 <= 0 || d <= 0) return NR_EINVAL;
    if ((d % 256) != 0 || d / 256 > NR_PREP_MAX_CHUNKS) return NR_EUNSUPPORTED;
    auto blocks = [](int n_tok, const float* cs) {
        if (cs) return nr_prepare_parts(n_tok);
        int g = (n_tok + 15) / 16;
        return g > NR_PREP_MAX_GRID ? NR_PREP_MAX_GRID : g;
    };
    const int nb0 = blocks(n_tok0, colsum_part0), nb1 = blocks(n_tok1, colsum_part1);
    NrPrepareOne a{x0, mask0, n_tok0, hi0, lo0, norm0, colsum_part0}, b{x1, mask1, n_tok1, hi1, lo1, norm1, colsum_part1};
    hipStream_t st = (hipStream_t)stream;
    switch (d / 256) {
        case 1: hipLaunchKernelGGL(nr_prepare_pair_kernel<1>, dim3(nb0 + nb1), dim3(256), 0, st, a, b, d, normalize, nb0); break;
        case 2: hipLaunchKernelGGL(nr_prepare_pair_kernel<2>, dim3(nb0 + nb1), dim3(256), 0, st, a, b, d, normalize, nb0); break;
        case 3: hipLaunchKernelGGL(nr_prepare_pair_kernel<3>, dim3(nb0 + nb1), dim3(256), 0, st, a, b, d, normalize, nb0); break;
        default: hipLaunchKernelGGL(nr_prepare_pair_kernel<4>, dim3(nb0 + nb1), dim3(256), 0, st, a, b, d, normalize, nb0); break;
    }
    NR_LAUNCH_CHECK();
    return NR_OK;
}

__global__ __launch_bounds__(256) void nr_split_kernel(const float* __restrict__ x, size_t n, uint16_t* __restrict__ hi,
                                                       uint16_t* __restrict__ lo) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float v = x[i];
        uint16_t h = nr_f2bf(v);
        hi[i] = h;
        if (lo) lo[i] = nr_f2bf(v - nr_bf2f(h));
    }
}

extern "C" int nr_split_bf16(const float* x, size_t n, uint16_t* hi, uint16_t* lo, void* stream) {
    if (!x || !hi || n == 0) return NR_EINVAL;
    size_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(nr_split_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, n, hi, lo);
    NR_LAUNCH_CHECK();
    return NR_OK;
}

// ---- a gathered batch straight into the memory bank (step-interleaved job: a step this rank does not own) ----------------------
// What such a step leaves behind is the FIFO push of modeling.py:309-310 (ring form) and the bank's prepared shadow; nothing else of
// the gathered batch is needed.  One launch does it from the RECEIVE buffer of the packed all-gather: the ring head moves back
// by the batch (as nr_step_prologue moves it), every token row is copied into its ring slot as fp32 AND prepared (normalise, mask,
// bf16 hi / lo, norm: the arithmetic of nr_prepare_body, bit for bit) into the shadow's slot, masks (u8 -> f32) and sample ids
// follow, and the noise stream's counter advances by one step -- nr_unpack_gathered + nr_step_prologue + nr_prepare_tokens_pair
// + nr_bank_ring_push of the eager path.  The new head is published by the LAST workgroup (every workgroup has read the old
// one by then); `counter` is a zeroed word the launch leaves zeroed.
#define NR_ABSORB_GROUP 32
extern "C" int nr_bank_absorb_counter_words(void) { return 16 * (1 + (NR_PREP_MAX_GRID + NR_ABSORB_GROUP - 1) / NR_ABSORB_GROUP); }

template <int CH>
__global__ __launch_bounds__(256) void nr_bank_absorb_kernel(NrBankAbsorbDesc a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int B = a.world * a.per_rank, d = a.d;
    // B >= capacity (modeling.py:244-249: the bank becomes cat(batch, bank)[:capacity]): the batch's first `capacity` samples, head 0
    const bool whole = B >= a.capacity;
    const int Be = whole ? a.capacity : B;
    const int rows_t = Be * a.Nt, total = rows_t + Be * a.Nv;
    const int old_head = __hip_atomic_load(a.ring_head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old_head < 0 || old_head >= a.capacity) return;    // (a head outside the ring never becomes a wild write; the counter stays zero)
    int nh = whole ? 0 : (old_head - B) % a.capacity;
    if (nh < 0) nh += a.capacity;
    const char* recv = reinterpret_cast<const char*>(a.gathered);
    for (int row = blockIdx.x * 4 + wave; row < total; row += gridDim.x * 4) {
        const bool text = row < rows_t;
        const int r = text ? row : row - rows_t, N = text ? a.Nt : a.Nv;
        const int i = r / N, tok = r - i * N, w = i / a.per_rank, s = i - w * a.per_rank;
        const char* rec = recv + (size_t)w * a.record_bytes;
        const float* xr = reinterpret_cast<const float*>(rec + (text ? a.off_text : a.off_video)) + ((size_t)s * N + tok) * d;
        const unsigned char* mrow = reinterpret_cast<const unsigned char*>(rec + (text ? a.off_text_mask : a.off_video_mask)) + (size_t)s * N;
        int rr = nh + i;
        if (rr >= a.capacity) rr -= a.capacity;
        const size_t drow = (size_t)rr * N + tok;
        float* bank = (text ? a.bank_text : a.bank_video) + drow * d;
        uint16_t* hi = text ? a.shadow_text_hi : a.shadow_video_hi;
        uint16_t* lo = text ? a.shadow_text_lo : a.shadow_video_lo;
        float* nrm_out = text ? a.shadow_text_norm : a.shadow_video_norm;
        f32x4_t v[CH];
        float ss = 0.f;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            v[c] = *reinterpret_cast<const f32x4_t*>(xr + (c * 64 + lane) * 4);
            *reinterpret_cast<f32x4_t*>(bank + (c * 64 + lane) * 4) = v[c];
#pragma unroll
            for (int e = 0; e < 4; ++e) ss += v[c][e] * v[c][e];
        }
        if (tok == 0) {                                    // the sample's mask row and id travel with its first token
            float* bm = (text ? a.bank_text_mask : a.bank_video_mask) + (size_t)rr * N;
            if (lane < N) bm[lane] = (float)mrow[lane];
            if (text && lane == 0)
                a.bank_index[rr] = reinterpret_cast<const long long*>(rec + a.off_index)[s];
        }
        if (hi) {
            ss = nr_wave_sum(ss);
            const float nrm = fmaxf(sqrtf(ss), 1e-12f);
            const float inv = 1.0f / nrm, mk = (float)mrow[tok];
            if (lane == 0) nrm_out[drow] = nrm;
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                uint16_t h[4], l[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float y = (v[c][e] * inv) * mk;
                    h[e] = nr_f2bf(y);
                    l[e] = nr_f2bf(y - nr_bf2f(h[e]));
                }
                const size_t o = drow * d + (c * 64 + lane) * 4;
                *reinterpret_cast<uint2*>(hi + o) = make_uint2((uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16));
                *reinterpret_cast<uint2*>(lo + o) = make_uint2((uint32_t)l[0] | ((uint32_t)l[1] << 16), (uint32_t)l[2] | ((uint32_t)l[3] << 16));
            }
        }
    }
    __syncthreads();
    // Two-level ticket: a thousand device-scope adds to ONE word queue up at the memory side (the launch took 18.5 us, 10 of them
    // this queue); workgroups add to the word of their group of NR_ABSORB_GROUP (words 64 B apart), the last of a group to the
    // launch's word.  counter[0]: groups done; counter[16 * (1 + g)]: workgroups of group g done.  All zero again on exit.
    if (threadIdx.x == 0) {
        const unsigned int g = blockIdx.x / NR_ABSORB_GROUP, n_groups = (gridDim.x + NR_ABSORB_GROUP - 1) / NR_ABSORB_GROUP;
        const unsigned int in_group = min((unsigned int)NR_ABSORB_GROUP, gridDim.x - g * NR_ABSORB_GROUP);
        unsigned int* gw = a.counter + 16 * (1 + g);
        if (__hip_atomic_fetch_add(gw, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == in_group - 1) {
            __hip_atomic_store(gw, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__hip_atomic_fetch_add(a.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == n_groups - 1) {
                __hip_atomic_store(a.ring_head, nh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (a.rng_state) a.rng_state[1] += 1ull;
                __hip_atomic_store(a.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

extern "C" int nr_bank_absorb_gathered(const NrBankAbsorbDesc* desc, void* stream) {
    if (!desc) return NR_EINVAL;
    const NrBankAbsorbDesc& a = *desc;
    if (!a.gathered || !a.bank_text || !a.bank_video || !a.bank_index || !a.bank_text_mask || !a.bank_video_mask || !a.ring_head || !a.counter)
        return NR_EINVAL;
    if (a.world <= 0 || a.per_rank <= 0 || a.Nt <= 0 || a.Nv <= 0 || a.d <= 0 || a.capacity <= 0) return NR_EINVAL;
    const long long B = (long long)a.world * a.per_rank;
    if ((a.d % 256) != 0 || a.d / 256 > NR_PREP_MAX_CHUNKS || a.Nt > 64 || a.Nv > 64) return NR_EUNSUPPORTED;
    const bool shadow = a.shadow_text_hi != nullptr;
    if (shadow != (a.shadow_text_lo && a.shadow_text_norm && a.shadow_video_hi && a.shadow_video_lo && a.shadow_video_norm)) return NR_EINVAL;
    if ((a.off_text % 16) != 0 || (a.off_video % 16) != 0 || (a.off_index % 8) != 0 || (a.record_bytes % 16) != 0) return NR_EINVAL;
    const long long rows = (B < a.capacity ? B : (long long)a.capacity) * ((long long)a.Nt + a.Nv);
    int grid = (int)((rows + 3) / 4);
    if (grid > NR_PREP_MAX_GRID) grid = NR_PREP_MAX_GRID;
    hipStream_t st = (hipStream_t)stream;
    switch (a.d / 256) {
        case 1: hipLaunchKernelGGL(nr_bank_absorb_kernel<1>, dim3(grid), dim3(256), 0, st, a); break;
        case 2: hipLaunchKernelGGL(nr_bank_absorb_kernel<2>, dim3(grid), dim3(256), 0, st, a); break;
        case 3: hipLaunchKernelGGL(nr_bank_absorb_kernel<3>, dim3(grid), dim3(256), 0, st, a); break;
        default: hipLaunchKernelGGL(nr_bank_absorb_kernel<4>, dim3(grid), dim3(256), 0, st, a); break;
    }
    NR_LAUNCH_CHECK();
    return NR_OK;
}
