// Small supporting kernels of the loss head: partial-sum reduction, the exact-fp32 global-logit
// GEMM, centrality weights, memory-bank FIFO push, diagonal ranks for R@K.
#include <dlfcn.h>
#include "nr_common.h"
#include "../../include/nr_hip.h"
#include <string.h>

extern "C" int nr_version(void) { return NR_ABI_VERSION; }

// sizeof of a descriptor struct of include/nr_hip.h by name (0: unknown) -- a binding checks its own layout against the library's
// before the first grouped call (neighborretr_amd/hip.py does at load time; tests/test_host_cpu.py asserts it)
extern "C" size_t nr_struct_size(const char* name) {
    if (!name) return 0;
#define NR_SIZE_OF(T) if (!strcmp(name, #T)) return sizeof(T)
    NR_SIZE_OF(NrCtmStageDesc); NR_SIZE_OF(NrLocalLevelProblem); NR_SIZE_OF(NrSplitItem); NR_SIZE_OF(NrColsumItem);
    NR_SIZE_OF(NrLinearProblem); NR_SIZE_OF(NrCtmAttnBwdDesc); NR_SIZE_OF(NrCtmMidBwdDesc); NR_SIZE_OF(NrSimBwdItem);
    NR_SIZE_OF(NrSimBwdOperand); NR_SIZE_OF(NrSlabSum); NR_SIZE_OF(NrPoolWSrc); NR_SIZE_OF(NrPoolWJob); NR_SIZE_OF(NrBankAbsorbDesc);
    NR_SIZE_OF(NrTokenWeightsProblem);
#undef NR_SIZE_OF
    return 0;
}

// Identity of the stream capture `stream` takes part in: *id = the runtime's capture sequence id (unique per capture in
// this process), 0 when the stream is not capturing.  Host-only; lets the host code check its stream topology per capture.
extern "C" int nr_stream_capture_id(void* stream, unsigned long long* id) {
    if (!id) return NR_EINVAL;
    hipStreamCaptureStatus status = hipStreamCaptureStatusNone;
    unsigned long long cid = 0;
    hipError_t e = hipStreamGetCaptureInfo((hipStream_t)stream, &status, &cid);
    if (e != hipSuccess) return (int)e;
    *id = status == hipStreamCaptureStatusActive ? (cid ? cid : 1ull) : 0ull;
    return NR_OK;
}

extern "C" int nr_stream_create(void** stream) {
    if (!stream) return NR_EINVAL;
    hipStream_t s = nullptr;
    hipError_t e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    if (e != hipSuccess) return (int)e;
    *stream = (void*)s;
    return NR_OK;
}

extern "C" int nr_stream_destroy(void* stream) {
    if (!stream) return NR_EINVAL;
    hipError_t e = hipStreamDestroy((hipStream_t)stream);
    return e == hipSuccess ? NR_OK : (int)e;
}

// ---- out[i] = scale * sum_p part[p,i]  (until_module.py:181) -----------------------------------
// 64 outputs per workgroup; wave w sums the parts p = w, w+4, ... (independent loads in flight),
// the four partial sums meet in LDS -- fixed order, deterministic.
__global__ __launch_bounds__(256) void nr_reduce_parts_kernel(const float* __restrict__ part, int n_parts, int n,
                                                              float scale, float* __restrict__ out) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + lane;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (i < n) {
        int p = wave;
        for (; p + 12 < n_parts; p += 16) {
            s0 += part[(size_t)p * n + i];
            s1 += part[(size_t)(p + 4) * n + i];
            s2 += part[(size_t)(p + 8) * n + i];
            s3 += part[(size_t)(p + 12) * n + i];
        }
        for (; p < n_parts; p += 4) s0 += part[(size_t)p * n + i];
    }
    red[wave][lane] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (wave == 0 && i < n) out[i] = ((red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane])) * scale;
}

extern "C" int nr_reduce_parts(const float* part, int n_parts, int n, float scale, float* out, void* stream) {
    if (!part || !out || n_parts <= 0 || n <= 0) return NR_EINVAL;
    hipLaunchKernelGGL(nr_reduce_parts_kernel, dim3((n + 63) / 64), dim3(256), 0, (hipStream_t)stream, part, n_parts,
                       n, scale, out);
    NR_LAUNCH_CHECK();
    return NR_OK;
}

// ---- C = A * B^T in exact fp32 on v_mfma_f32_16x16x4_f32 (modeling.py:526, one global token) ---
// One WORKGROUP per 16x16 output tile, its four waves splitting K (wave w takes the 16-wide k blocks
// w, w+4, ...) and meeting in LDS: the product is tiny (B x B x d) and sits on the step's critical path,
// so the K loop is kept short and its loads independent.  Operands come straight from global/L2 as
// float4 along k; element s of a lane's float4 feeds MFMA step s, i.e. step s multiplies
// k = k0 + 4*(lane>>4) + s on both operands -- a permutation of the k order, identical for A and B.
__global__ __launch_bounds__(256) void nr_gemm_nt_f32_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                             int M, int N, int K, float* __restrict__ c) {
    NR_CRITICAL_PATH();
    __shared__ f32x4_t s_part[3][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row0 = blockIdx.y * 16, col0 = blockIdx.x * 16;
    const int ar = min(row0 + (lane & 15), M - 1);
    const int br = min(col0 + (lane & 15), N - 1);
    const float* pa = a + (size_t)ar * K + 4 * (lane >> 4);
    const float* pb = b + (size_t)br * K + 4 * (lane >> 4);
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = wave * 16; k0 < K; k0 += 256) {        // 4 k blocks of this wave per trip, loads first
        f32x4_t va[4], vb[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = k0 + u * 64;
            const bool ok = k < K;
            va[u] = ok ? *reinterpret_cast<const f32x4_t*>(pa + k) : f32x4_t{0.f, 0.f, 0.f, 0.f};
            vb[u] = ok ? *reinterpret_cast<const f32x4_t*>(pb + k) : f32x4_t{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(va[u][s], vb[u][s], acc, 0, 0, 0);
    }
    if (wave > 0) s_part[wave - 1][lane] = acc;
    __syncthreads();
    if (wave > 0) return;
    const f32x4_t p1 = s_part[0][lane], p2 = s_part[1][lane], p3 = s_part[2][lane];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int r = row0 + (lane >> 4) * 4 + j, cc = col0 + (lane & 15);
        if (r < M && cc < N) c[(size_t)r * N + cc] = (acc[j] + p1[j]) + (p2[j] + p3[j]);
    }
}

extern "C" int nr_gemm_nt_f32(const float* a, const float* b, int M, int N, int K, float* c, void* stream) {
    if (!a || !b || !c || M <= 0 || N <= 0 || K <= 0 || (K % 16) != 0) return NR_EINVAL;
    dim3 grid((N + 15) / 16, (M + 15) / 16);
    hipLaunchKernelGGL(nr_gemm_nt_f32_kernel, grid, dim3(256), 0, (hipStream_t)stream, a, b, M, N, K, c);
    NR_LAUNCH_CHECK();
    return NR_OK;
}

// ---- centrality weights (modeling.py:403-430) ---------------------------------------------------
// mean over ALL tokens of g_hat . x_hat  ==  g_hat . (mean of x_hat): a mat-vec instead of the
// reference's [B,d] x [d,B*N] GEMM.  8 samples per workgroup; every workgroup rebuilds the mean
// vector from the <=64 per-workgroup column sums written by nr_prepare_tokens (L2-resident) with
// 32 independent loads in flight per thread -- the kernel is latency-bound, not bandwidth-bound.
#define NR_CW_SAMPLES 8
__global__ __launch_bounds__(256) void nr_centrality_kernel(const float* __restrict__ g, int B, int d,
                                                            const float* __restrict__ colsum_part, int n_parts,
                                                            float inv_tok, float scale, float* __restrict__ w,
                                                            float* __restrict__ gnorm, float* __restrict__ mean_out) {
    __shared__ float s_part[4][1024];
    __shared__ float s_mean[1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // wave `wave` sums the parts p = wave, wave+4, ... for all columns (8 columns x 4 parts in flight)
    for (int c0 = 0; c0 < d; c0 += 512) {
        float acc[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] = 0.f;
        for (int p = wave; p < n_parts; p += 16) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                int pp = p + 4 * u;
                if (pp < n_parts) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        int c = c0 + q * 64 + lane;
                        if (c < d) acc[q] += colsum_part[(size_t)pp * d + c];
                    }
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            int c = c0 + q * 64 + lane;
            if (c < d) s_part[wave][c] = acc[q];
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < d; k += 256) {
        float s = ((s_part[0][k] + s_part[1][k]) + (s_part[2][k] + s_part[3][k])) * inv_tok;
        s_mean[k] = s;
        if (mean_out && blockIdx.x == 0) mean_out[k] = s;
    }
    __syncthreads();
    for (int q = wave; q < NR_CW_SAMPLES; q += 4) {
        int i = blockIdx.x * NR_CW_SAMPLES + q;
        if (i >= B) break;
        const float* gi = g + (size_t)i * d;
        float dot = 0.f, ss = 0.f;
        for (int k = lane; k < d; k += 64) {
            float x = gi[k];
            dot += x * s_mean[k];
            ss += x * x;
        }
        dot = nr_wave_sum(dot);
        ss = nr_wave_sum(ss);
        float nrm = fmaxf(sqrtf(ss), 1e-12f);
        if (lane == 0) {
            w[i] = expf(dot / nrm * scale);
            if (gnorm) gnorm[i] = nrm;
        }
    }
}

extern "C" int nr_centrality_weights(const float* g, int B, int d, const float* colsum_part, int n_parts, int n_tok,
                                     float scale, float* w, float* gnorm, float* mean_out, void* stream) {
    if (!g || !colsum_part || !w || B <= 0 || d <= 0 || d > 1024 || n_parts <= 0 || n_tok <= 0) return NR_EINVAL;
    hipLaunchKernelGGL(nr_centrality_kernel, dim3((B + NR_CW_SAMPLES - 1) / NR_CW_SAMPLES), dim3(256), 0,
                       (hipStream_t)stream, g, B, d, colsum_part, n_parts, 1.0f / (float)n_tok, scale, w, gnorm, mean_out);
    NR_LAUNCH_CHECK();
    return NR_OK;
}

// Both modalities in one launch from FINISHED token means (mean = nr_reduce_parts over the prepared
// column sums, issued on the local branch long before the global tokens exist): one wave per
// (sample, modality), blockIdx.y = modality.  A sample may carry several global tokens (ActivityNet token
// counts leave 3 text / 6 video tokens): the wave walks its sample's n_g tokens, w = mean_g exp(scale * cos_g)
// -- the documented "mean" reduction of config.centrality_multi_token (the reference has no answer there,
// until_module.py:321); with n_g = 1 this is the reference's expression.  Per-token weights / norms are kept for
// the backward pass when asked for.
__global__ __launch_bounds__(256) void nr_centrality_pair_kernel(const float* __restrict__ g0, const float* __restrict__ g1,
                                                                 int B, int ng0, int ng1, int d, const float* __restrict__ mean0,
                                                                 const float* __restrict__ mean1, float scale,
                                                                 float* __restrict__ w0, float* __restrict__ w1,
                                                                 float* __restrict__ gn0, float* __restrict__ gn1,
                                                                 float* __restrict__ wtok0, float* __restrict__ wtok1) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= B) return;
    const bool second = blockIdx.y == 1;
    const int ng = second ? ng1 : ng0;
    const float* mean = second ? mean1 : mean0;
    float* gn = second ? gn1 : gn0;
    float* wtok = second ? wtok1 : wtok0;
    float acc = 0.f;
    for (int t = 0; t < ng; ++t) {
        const float* gi = (second ? g1 : g0) + ((size_t)i * ng + t) * d;
        float dot = 0.f, ss = 0.f;
        for (int k = lane * 4; k < d; k += 256) {
            f32x4_t x = *reinterpret_cast<const f32x4_t*>(gi + k);
            f32x4_t m = *reinterpret_cast<const f32x4_t*>(mean + k);
            dot += x[0] * m[0] + x[1] * m[1] + x[2] * m[2] + x[3] * m[3];
            ss += x[0] * x[0] + x[1] * x[1] + x[2] * x[2] + x[3] * x[3];
        }
        dot = nr_wave_sum(dot);
        ss = nr_wave_sum(ss);
        const float nrm = fmaxf(sqrtf(ss), 1e-12f);
        const float wt = expf(dot / nrm * scale);
        acc += wt;
        if (lane == 0) {
            if (gn) gn[(size_t)i * ng + t] = nrm;
            if (wtok) wtok[(size_t)i * ng + t] = wt;
        }
    }
    if (lane == 0) (second ? w1 : w0)[i] = ng == 1 ? acc : acc / (float)ng;
}

extern "C" int nr_centrality_weights_pair(const float* g_text, const float* g_video, int B, int n_g_text, int n_g_video, int d,
                                          const float* mean_text, const float* mean_video, float scale, float* w_text,
                                          float* w_video, float* gnorm_text, float* gnorm_video, float* wtok_text,
                                          float* wtok_video, void* stream) {
    if (!g_text || !g_video || !mean_text || !mean_video || !w_text || !w_video || B <= 0 || d <= 0 || (d % 4) != 0)
        return NR_EINVAL;
    if (n_g_text <= 0 || n_g_video <= 0) return NR_EINVAL;
    hipLaunchKernelGGL(nr_centrality_pair_kernel, dim3((B + 3) / 4, 2), dim3(256), 0, (hipStream_t)stream, g_text, g_video, B,
                       n_g_text, n_g_video, d, mean_text, mean_video, scale, w_text, w_video, gnorm_text, gnorm_video, wtok_text,
                       wtok_video);
    NR_LAUNCH_CHECK();
    return NR_OK;
}

// ---- memory-bank FIFO (modeling.py:237-249) -----------------------------------------------------
extern "C" int nr_bank_push(void* bank, const void* batch, int capacity, int n_new, size_t row_bytes, void* scratch,
                            void* stream) {
    if (!bank || !batch || capacity <= 0 || n_new <= 0 || row_bytes == 0) return NR_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e;
    if (n_new >= capacity) {
        e = hipMemcpyAsync(bank, batch, (size_t)capacity * row_bytes, hipMemcpyDeviceToDevice, st);
        return e == hipSuccess ? NR_OK : (int)e;
    }
    if (!scratch) return NR_EINVAL;
    size_t keep = (size_t)(capacity - n_new) * row_bytes;
    e = hipMemcpyAsync(scratch, bank, keep, hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) return (int)e;
    e = hipMemcpyAsync((char*)bank + (size_t)n_new * row_bytes, scratch, keep, hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) return (int)e;
    e = hipMemcpyAsync(bank, batch, (size_t)n_new * row_bytes, hipMemcpyDeviceToDevice, st);
    return e == hipSuccess ? NR_OK : (int)e;
}

// ---- memory bank as a ring: O(batch) bytes per push instead of O(bank) --------------------------
// Logical FIFO order (newest first, modeling.py:237-249) is L[i] = S[(head + i) mod capacity]; a push
// moves head back by n_new and writes the batch there.  All tensors of the bank (ids, features,
// masks) go in ONE launch: blockIdx.y = tensor, blockIdx.x = batch row.
#define NR_RING_MAX 12
struct NrRingArgs {
    void* bank[NR_RING_MAX];
    const void* batch[NR_RING_MAX];
    unsigned long long row_bytes[NR_RING_MAX];
    int capacity, head_new, n_new;
    const int* head_dev;       // when set: the ring head lives on the device (a captured graph must not bake it in)
};

__global__ __launch_bounds__(256) void nr_bank_ring_kernel(NrRingArgs a) {
    const int t = blockIdx.y, r = blockIdx.x;
    const size_t rb = a.row_bytes[t];
    int head = a.head_dev ? *a.head_dev : a.head_new;
    if (head < 0 || head >= a.capacity) return;        // a device-resident head outside the ring never becomes a wild write
    int dr = head + r;
    if (dr >= a.capacity) dr -= a.capacity;
    const char* src = (const char*)a.batch[t] + (size_t)r * rb;
    char* dst = (char*)a.bank[t] + (size_t)dr * rb;
    if ((rb & 15) == 0 && (((size_t)src | (size_t)dst) & 15) == 0) {
        const u32x4_t* s4 = reinterpret_cast<const u32x4_t*>(src);
        u32x4_t* d4 = reinterpret_cast<u32x4_t*>(dst);
        for (size_t i = threadIdx.x; i < rb / 16; i += 256) d4[i] = s4[i];
    } else {
        for (size_t i = threadIdx.x; i < rb; i += 256) dst[i] = src[i];
    }
}

extern "C" int nr_bank_ring_push(int n_tensors, void* const* banks, const void* const* batches, const size_t* row_bytes,
                                 int capacity, int head_new, const int32_t* head_dev, int n_new, void* stream) {
    if (n_tensors <= 0 || n_tensors > NR_RING_MAX || !banks || !batches || !row_bytes) return NR_EINVAL;
    if (capacity <= 0 || n_new <= 0 || n_new > capacity) return NR_EINVAL;
    if (!head_dev && (head_new < 0 || head_new >= capacity)) return NR_EINVAL;
    NrRingArgs a;
    for (int i = 0; i < n_tensors; ++i) {
        if (!banks[i] || !batches[i] || row_bytes[i] == 0) return NR_EINVAL;
        a.bank[i] = banks[i];
        a.batch[i] = batches[i];
        a.row_bytes[i] = row_bytes[i];
    }
    a.capacity = capacity; a.head_new = head_new; a.n_new = n_new; a.head_dev = head_dev;
    hipLaunchKernelGGL(nr_bank_ring_kernel, dim3(n_new, n_tensors), dim3(256), 0, (hipStream_t)stream, a);
    NR_LAUNCH_CHECK();
    return NR_OK;
}

// ---- diagonal ranks (metrics.py:58-66) ----------------------------------------------------------
__global__ __launch_bounds__(256) void nr_diag_ranks_kernel(const float* __restrict__ S, int N, int32_t* __restrict__ greater,
                                                            int32_t* __restrict__ equal) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= N) return;
    const float* row = S + (size_t)i * N;
    const float dval = row[i];
    int g = 0, e = 0;
    for (int j = lane; j < N; j += 64) {
        float x = row[j];
        g += (x > dval);
        e += (x == dval);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        g += __shfl_xor(g, o);
        e += __shfl_xor(e, o);
    }
    if (lane == 0) {
        greater[i] = g;
        equal[i] = e;
    }
}

extern "C" int nr_diag_ranks(const float* S, int N, int32_t* greater, int32_t* equal, void* stream) {
    if (!S || !greater || !equal || N <= 0) return NR_EINVAL;
    hipLaunchKernelGGL(nr_diag_ranks_kernel, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream, S, N, greater, equal);
    NR_LAUNCH_CHECK();
    return NR_OK;
}

// ---- ranks from a ROW SLAB of the similarity matrix (sharded evaluation: evaluator.py:21-63 + metrics.py:58-66) -----
// Rank r of W holds S[row0 : row0 + n, :] (its texts against all N videos) and the full diagonal (gathered: N floats).
// Text->video ranks are row-local: greater/equal counts of row i against diag[row0 + i].  Video->text ranks need a
// whole column: the slab contributes PARTIAL counts per column j against diag[j], summed over the ranks afterwards
// (one all-reduce of 2N integers).  blockIdx.y = 0: rows (one wave per row); 1: columns (one thread per column,
// coalesced over adjacent columns, rows walked in order).
__global__ __launch_bounds__(256) void nr_slab_ranks_kernel(const float* __restrict__ S, int n, int N, int row0,
                                                            const float* __restrict__ diag, int32_t* __restrict__ g_rows,
                                                            int32_t* __restrict__ e_rows, int32_t* __restrict__ g_cols,
                                                            int32_t* __restrict__ e_cols) {
    if (blockIdx.y == 0) {
        const int lane = threadIdx.x & 63;
        const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
        if (i >= n) return;
        const float* row = S + (size_t)i * N;
        const float dval = diag[row0 + i];
        int g = 0, e = 0;
        for (int j = lane; j < N; j += 64) {
            const float x = row[j];
            g += (x > dval);
            e += (x == dval);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            g += __shfl_xor(g, o);
            e += __shfl_xor(e, o);
        }
        if (lane == 0) { g_rows[i] = g; e_rows[i] = e; }
    } else {
        const int j = blockIdx.x * 256 + threadIdx.x;
        if (j >= N) return;
        const float dval = diag[j];
        int g = 0, e = 0;
        for (int i = 0; i < n; ++i) {
            const float x = S[(size_t)i * N + j];
            g += (x > dval);
            e += (x == dval);
        }
        g_cols[j] = g;
        e_cols[j] = e;
    }
}

extern "C" int nr_slab_ranks(const float* S_slab, int n_rows, int N, int row0, const float* diag, int32_t* greater_rows,
                             int32_t* equal_rows, int32_t* greater_cols, int32_t* equal_cols, void* stream) {
    if (!S_slab || !diag || !greater_rows || !equal_rows || !greater_cols || !equal_cols) return NR_EINVAL;
    if (n_rows <= 0 || N <= 0 || row0 < 0 || row0 + n_rows > N) return NR_EINVAL;
    const int gx = max((n_rows + 3) / 4, (N + 255) / 256);
    hipLaunchKernelGGL(nr_slab_ranks_kernel, dim3(gx, 2), dim3(256), 0, (hipStream_t)stream, S_slab, n_rows, N, row0, diag,
                       greater_rows, equal_rows, greater_cols, equal_cols);
    NR_LAUNCH_CHECK();
    return NR_OK;
}

// ---- multi-sentence retrieval (evaluator.py:114-149,225-262; metrics.py:82-148) from a row slab --------------------------
// Rows are SENTENCES (all captions of video 0, then of video 1, ...: group g owns the global rows [group_end[g-1],
// group_end[g])), columns are videos.  blockIdx.y = 0: one wave per sentence row -> the position of its OWN video in the
// row's descending order, as greater[i] = #{j : S[i,j] > S[i,g] or NaN} and equal_before[i] = #{j < g : S[i,j] == S[i,g]} (the
// order a stable descending argsort gives; the reference's double argsort, metrics.py:103-106, is that up to the sort's
// choice among exact ties).  blockIdx.y = 1: one workgroup per (group, 256 columns) -> group_max[g, j] = the best score
// any of this slab's sentences of group g gives video j (-inf when the slab holds none of them): the video->text matrix
// of metrics.py:141-146 after a MAX all-reduce over the ranks.
__global__ __launch_bounds__(256) void nr_group_slab_ranks_kernel(const float* __restrict__ S, int n, int V, int row0,
                                                                  const int32_t* __restrict__ group_end, int G,
                                                                  int32_t* __restrict__ greater,
                                                                  int32_t* __restrict__ equal_before,
                                                                  float* __restrict__ group_max) {
    if (blockIdx.y == 0) {
        const int lane = threadIdx.x & 63;
        const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
        if (i >= n) return;
        const int s = row0 + i;
        int lo = 0, hi = G - 1;                 // first group whose end lies beyond s
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (group_end[mid] > s) hi = mid; else lo = mid + 1;
        }
        const int g = lo;
        const float* row = S + (size_t)i * V;
        const float own = row[g];
        int gt = 0, eb = 0;
        for (int j = lane; j < V; j += 64) {
            const float x = row[j];
            gt += (x > own) || (x != x);        // torch.argsort(descending=True) ranks NaN scores first
            eb += (x == own && j < g);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            gt += __shfl_xor(gt, o);
            eb += __shfl_xor(eb, o);
        }
        // a sentence whose own score is NaN or infinite is not ranked at all (metrics.py:108-111): greater = -1
        if (lane == 0) { greater[i] = (own != own || fabsf(own) == INFINITY) ? -1 : gt; equal_before[i] = eb; }
    } else {
        const int cols_blocks = (V + 255) / 256;
        const int g = blockIdx.x / cols_blocks;
        const int j = (blockIdx.x % cols_blocks) * 256 + threadIdx.x;
        if (g >= G || j >= V) return;
        const int begin = max((g ? group_end[g - 1] : 0) - row0, 0);
        const int end = min(group_end[g] - row0, n);
        float m = -INFINITY;
        for (int i = begin; i < end; ++i) {
            const float x = S[(size_t)i * V + j];
            m = (x != x) ? m : fmaxf(m, x);     // NaN entries count as -inf (metrics.py:141)
        }
        group_max[(size_t)g * V + j] = m;
    }
}

extern "C" int nr_group_slab_ranks(const float* S_slab, int n_rows, int V, int row0, const int32_t* group_end, int G,
                                   int32_t* greater_rows, int32_t* equal_before_rows, float* group_max, void* stream) {
    if (!S_slab || !group_end || !greater_rows || !equal_before_rows || !group_max) return NR_EINVAL;
    if (n_rows <= 0 || V <= 0 || G <= 0 || G > V || row0 < 0) return NR_EINVAL;
    const long long col_part = (long long)G * ((V + 255) / 256);
    if (col_part > 0x7fffffffLL) return NR_EINVAL;
    const int gx = max((n_rows + 3) / 4, (int)col_part);
    hipLaunchKernelGGL(nr_group_slab_ranks_kernel, dim3(gx, 2), dim3(256), 0, (hipStream_t)stream, S_slab, n_rows, V, row0,
                       group_end, G, greater_rows, equal_before_rows, group_max);
    NR_LAUNCH_CHECK();
    return NR_OK;
}

// ---- step prologue -------------------------------------------------------------------------------------------
// What the step needs before anything else can start, in ONE launch instead of six ATen kernels on the serial
// front of the critical path: the int64 masks of the loader as fp32 multipliers (modeling.py:283-287 keeps them
// int64; every kernel here reads fp32), exp(logit_scale) (modeling.py:289), and the uniform tie-break noise of the
// four DPC-KNN calls (cluster.py:483, torch.rand there).  The noise is a counter-based stream
// u = splitmix64(splitmix64(seed + counter) + i) >> 40 / 2^24 whose (seed, counter) pair lives on the device and
// is advanced by the kernel itself, so a captured HIP graph draws fresh numbers at every replay.
__device__ __forceinline__ unsigned long long nr_splitmix64(unsigned long long z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ __launch_bounds__(1024) void nr_step_prologue_kernel(const int64_t* __restrict__ m0, int n0, float* __restrict__ o0,
                                                               const int64_t* __restrict__ m1, int n1, float* __restrict__ o1,
                                                               const float* __restrict__ ls, float* __restrict__ ls_exp,
                                                               unsigned long long* __restrict__ rng, float* __restrict__ noise,
                                                               int n_noise, int* __restrict__ ring_head, int ring_advance,
                                                               int ring_capacity) {
    const int tid = threadIdx.x;
    // the memory bank's ring head moves back by the batch that this step will push (modeling.py:237-249 as a ring)
    if (ring_head && tid == 0) {
        int h = (*ring_head - ring_advance) % ring_capacity;
        *ring_head = h < 0 ? h + ring_capacity : h;
    }
    unsigned long long key = 0, ctr = 0;
    if (noise) {
        ctr = rng[1];
        key = nr_splitmix64(rng[0] + ctr);
    }
    for (int i = tid; i < n0; i += 1024) o0[i] = (float)m0[i];
    for (int i = tid; i < n1; i += 1024) o1[i] = (float)m1[i];
    if (ls && tid == 0) ls_exp[0] = expf(ls[0]);
    for (int i = tid; i < n_noise; i += 1024)
        noise[i] = (float)(nr_splitmix64(key + (unsigned long long)i) >> 40) * (1.0f / 16777216.0f);
    if (noise) {
        __syncthreads();                    // every thread has read the counter
        if (tid == 0) rng[1] = ctr + 1;
    }
}

extern "C" int nr_step_prologue(const int64_t* mask0, int n0, float* out0, const int64_t* mask1, int n1, float* out1,
                                const float* logit_scale, float* logit_scale_exp, uint64_t* rng_state, float* noise,
                                int n_noise, int32_t* ring_head, int ring_advance, int ring_capacity, void* stream) {
    if (n0 < 0 || n1 < 0 || n_noise < 0) return NR_EINVAL;
    if (ring_head && (ring_capacity <= 0 || ring_advance < 0)) return NR_EINVAL;
    if ((n0 > 0 && (!mask0 || !out0)) || (n1 > 0 && (!mask1 || !out1))) return NR_EINVAL;
    if ((logit_scale != nullptr) != (logit_scale_exp != nullptr)) return NR_EINVAL;
    if (n_noise > 0 && (!rng_state || !noise)) return NR_EINVAL;
    hipLaunchKernelGGL(nr_step_prologue_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, mask0, n0, out0, mask1, n1, out1,
                       logit_scale, logit_scale_exp, reinterpret_cast<unsigned long long*>(n_noise > 0 ? rng_state : nullptr),
                       n_noise > 0 ? noise : nullptr, n_noise, ring_head, ring_advance, ring_capacity);
    NR_LAUNCH_CHECK();
    return NR_OK;
}

// ---- exchange step: pack a rank's shard / unpack the gathered buffer -------------------------------------------
// The five tensors of the exchange (modeling.py:274-280) travel as ONE byte buffer per rank.  nr_pack_shard
// concatenates them in one launch; nr_unpack_gathered scatters the [W, stride] result of the all-gather into the
// five rank-major outputs in one launch, converting the u8 masks to the fp32 multipliers the kernels read.
struct NrPackArgs {
    const char* src[8];
    char* dst[8];
    unsigned long long bytes[8], off[8];     // per piece: size per rank, offset inside a rank's packed record
    int convert[8];                          // unpack: 1 = u8 -> f32;  pack: 1 = i64 -> u8, 2 = f32 -> u8 (the masks, as .to(uint8))
    int n, W;
    unsigned long long stride;               // packed record size
};

__global__ __launch_bounds__(256) void nr_pack_kernel(NrPackArgs a, char* __restrict__ out) {
    const int k = blockIdx.y;
    if (k >= a.n) return;
    const unsigned long long nb = a.bytes[k];
    const char* s = a.src[k];
    char* d = out + a.off[k];
    if (a.convert[k]) {                       // a mask of nb elements: int64 / fp32 -> the u8 the record carries
        const long long* s64 = reinterpret_cast<const long long*>(s);
        const float* s32 = reinterpret_cast<const float*>(s);
        for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < nb; i += (unsigned long long)gridDim.x * 256)
            d[i] = (char)(unsigned char)(a.convert[k] == 1 ? s64[i] : (long long)s32[i]);
        return;
    }
    const bool vec = ((reinterpret_cast<uintptr_t>(s) | reinterpret_cast<uintptr_t>(d) | nb) & 15) == 0;
    if (vec) {
        for (unsigned long long i = ((unsigned long long)blockIdx.x * 256 + threadIdx.x) * 16; i < nb; i += (unsigned long long)gridDim.x * 4096)
            *reinterpret_cast<uint4*>(d + i) = *reinterpret_cast<const uint4*>(s + i);
    } else {
        for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < nb; i += (unsigned long long)gridDim.x * 256) d[i] = s[i];
    }
}

__global__ __launch_bounds__(256) void nr_unpack_kernel(NrPackArgs a, const char* __restrict__ recv) {
    const int k = blockIdx.y, w = blockIdx.z;
    if (k >= a.n) return;
    const unsigned long long nb = a.bytes[k];
    const char* s = recv + (unsigned long long)w * a.stride + a.off[k];
    if (a.convert[k]) {
        float* d = reinterpret_cast<float*>(a.dst[k]) + (unsigned long long)w * nb;
        for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < nb; i += (unsigned long long)gridDim.x * 256)
            d[i] = (float)(unsigned char)s[i];
        return;
    }
    char* d = a.dst[k] + (unsigned long long)w * nb;
    const bool vec = ((reinterpret_cast<uintptr_t>(s) | reinterpret_cast<uintptr_t>(d) | nb) & 15) == 0;
    if (vec) {
        for (unsigned long long i = ((unsigned long long)blockIdx.x * 256 + threadIdx.x) * 16; i < nb; i += (unsigned long long)gridDim.x * 4096)
            *reinterpret_cast<uint4*>(d + i) = *reinterpret_cast<const uint4*>(s + i);
    } else {
        for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < nb; i += (unsigned long long)gridDim.x * 256) d[i] = s[i];
    }
}

extern "C" int nr_pack_shard_convert(int n, const void* const* srcs, const size_t* bytes, const size_t* offsets, const int* kinds,
                                     void* packed, void* stream) {
    if (n <= 0 || n > 8 || !srcs || !bytes || !offsets || !packed) return NR_EINVAL;
    NrPackArgs a{};
    a.n = n;
    size_t mx = 0;
    for (int k = 0; k < n; ++k) {
        if (!srcs[k]) return NR_EINVAL;
        if (kinds && (kinds[k] < 0 || kinds[k] > 2)) return NR_EINVAL;
        a.src[k] = static_cast<const char*>(srcs[k]);
        a.bytes[k] = bytes[k];
        a.off[k] = offsets[k];
        a.convert[k] = kinds ? kinds[k] : 0;
        mx = bytes[k] > mx ? bytes[k] : mx;
    }
    unsigned gx = (unsigned)((mx + 4095) / 4096);
    if (gx < 1) gx = 1;
    if (gx > 256) gx = 256;
    hipLaunchKernelGGL(nr_pack_kernel, dim3(gx, n), dim3(256), 0, (hipStream_t)stream, a, static_cast<char*>(packed));
    NR_LAUNCH_CHECK();
    return NR_OK;
}

extern "C" int nr_pack_shard(int n, const void* const* srcs, const size_t* bytes, const size_t* offsets, void* packed, void* stream) {
    return nr_pack_shard_convert(n, srcs, bytes, offsets, nullptr, packed, stream);
}

extern "C" int nr_unpack_gathered(int n, const void* gathered, int world, size_t record_bytes, const size_t* bytes,
                                  const size_t* offsets, void* const* dsts, const int* u8_to_f32, void* stream) {
    if (n <= 0 || n > 8 || !gathered || world <= 0 || !bytes || !offsets || !dsts || !u8_to_f32) return NR_EINVAL;
    NrPackArgs a{};
    a.n = n;
    a.W = world;
    a.stride = record_bytes;
    size_t mx = 0;
    for (int k = 0; k < n; ++k) {
        if (!dsts[k]) return NR_EINVAL;
        a.dst[k] = static_cast<char*>(dsts[k]);
        a.bytes[k] = bytes[k];
        a.off[k] = offsets[k];
        a.convert[k] = u8_to_f32[k];
        mx = bytes[k] > mx ? bytes[k] : mx;
    }
    unsigned gx = (unsigned)((mx + 4095) / 4096);
    if (gx < 1) gx = 1;
    if (gx > 128) gx = 128;
    hipLaunchKernelGGL(nr_unpack_kernel, dim3(gx, n, world), dim3(256), 0, (hipStream_t)stream, a, static_cast<const char*>(gathered));
    NR_LAUNCH_CHECK();
    return NR_OK;
}

// ---- several device-to-device copies in ONE launch (the bank copy of an overlapped owned step: modeling.OwnedSlot.take) -----------
struct NrCopyGroupArgs {
    const char* src[12];
    char* dst[12];
    unsigned long long bytes[12];
    int n;
};

__global__ __launch_bounds__(256) void nr_copy_group_kernel(NrCopyGroupArgs a) {
    const int k = blockIdx.y;
    if (k >= a.n) return;
    const unsigned long long nb = a.bytes[k];
    const char* s = a.src[k];
    char* d = a.dst[k];
    const bool vec = ((reinterpret_cast<uintptr_t>(s) | reinterpret_cast<uintptr_t>(d) | nb) & 15) == 0;
    if (vec) {
        for (unsigned long long i = ((unsigned long long)blockIdx.x * 256 + threadIdx.x) * 16; i < nb; i += (unsigned long long)gridDim.x * 4096)
            *reinterpret_cast<uint4*>(d + i) = *reinterpret_cast<const uint4*>(s + i);
    } else {
        for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < nb; i += (unsigned long long)gridDim.x * 256) d[i] = s[i];
    }
}

extern "C" int nr_copy_group(int n, const void* const* srcs, void* const* dsts, const size_t* bytes, void* stream) {
    if (n <= 0 || n > 12 || !srcs || !dsts || !bytes) return NR_EINVAL;
    NrCopyGroupArgs a{};
    a.n = n;
    size_t mx = 0;
    for (int k = 0; k < n; ++k) {
        if (!srcs[k] || !dsts[k]) return NR_EINVAL;
        a.src[k] = static_cast<const char*>(srcs[k]);
        a.dst[k] = static_cast<char*>(dsts[k]);
        a.bytes[k] = bytes[k];
        mx = bytes[k] > mx ? bytes[k] : mx;
    }
    unsigned gx = (unsigned)((mx + 16383) / 16384);          // four 16-byte pieces per thread of the largest copy
    if (gx < 1) gx = 1;
    if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(nr_copy_group_kernel, dim3(gx, n), dim3(256), 0, (hipStream_t)stream, a);
    NR_LAUNCH_CHECK();
    return NR_OK;
}

// ---- the whole exchange step behind the C ABI: pack -> ONE RCCL all-gather over xGMI -> unpack --------------------------
// (reference: 5 x all_gather + barrier per step, modeling.py:274-280 via until_module.py:367-388).  RCCL is resolved at
// run time -- the copy the caller's framework has already loaded (PyTorch-ROCm ships its own librccl.so.1), else the
// system one -- so that this library carries no link-time dependency on it and a process never ends up with two.
typedef int (*nr_nccl_allgather_t)(const void*, void*, size_t, int, void*, hipStream_t);
static nr_nccl_allgather_t nr_resolve_allgather() {
    static nr_nccl_allgather_t fn = nullptr;
    static bool tried = false;
    if (!tried) {
        tried = true;
        void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
        if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (h) fn = reinterpret_cast<nr_nccl_allgather_t>(dlsym(h, "ncclAllGather"));
    }
    return fn;
}

extern "C" int nr_allgather_packed(void* nccl_comm, int world, int n, const void* const* srcs, const size_t* bytes,
                                   const size_t* offsets, size_t record_bytes, void* packed, void* gathered,
                                   void* const* dsts, const int* u8_to_f32, void* stream) {
    if (!nccl_comm || world <= 0 || !packed || !gathered || record_bytes == 0) return NR_EINVAL;
    nr_nccl_allgather_t allgather = nr_resolve_allgather();
    if (!allgather) return NR_EUNSUPPORTED;            // no RCCL in this process and none loadable
    int rc = nr_pack_shard(n, srcs, bytes, offsets, packed, stream);
    if (rc != NR_OK) return rc;
    const int nccl_uint8 = 1;                          // ncclUint8 (rccl.h)
    const int nrc = allgather(packed, gathered, record_bytes, nccl_uint8, nccl_comm, (hipStream_t)stream);
    if (nrc != 0) return 1000 + nrc;                   // ncclResult_t of the collective, offset past hipError_t
    return nr_unpack_gathered(n, gathered, world, record_bytes, bytes, offsets, dsts, u8_to_f32, stream);
}
