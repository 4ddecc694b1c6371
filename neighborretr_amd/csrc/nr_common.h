// Shared device helpers for the NeighborRetr loss-head kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) short short8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

#define NR_WAVE 64
#define NR_NEG_BIG (-9e15f)   // the reference's "-inf" (modeling.py:486, until_module.py:111)
#define NR_POS_BIG (9e15f)

// Tuning hooks (block shapes, ring depths, tile orders forced from the environment for A/B measurements) exist only in
// -DNR_TUNE builds (NR_EXTRA_FLAGS=-DNR_TUNE python -m neighborretr_amd.build --force).  A release build never reads
// the environment: the behaviour of the shipped ABI does not depend on stray variables.
#ifdef NR_TUNE
#include <stdlib.h>
static inline const char* nr_tune_env(const char* name) { return getenv(name); }
#else
static inline const char* nr_tune_env(const char*) { return nullptr; }
#endif

// Kernels on the step's critical chain (clustering, global logits, Sinkhorn, finalize) raise their waves' issue priority:
// they share CUs with the MFMA kernels of the local branch, which only have throughput to lose.
#define NR_CRITICAL_PATH() __builtin_amdgcn_s_setprio(3)

// status codes returned by every extern "C" entry point
#define NR_OK 0
#define NR_EINVAL (-1)
#define NR_EUNSUPPORTED (-2)

// Every launch first clears the thread's sticky "last error" (the host process -- PyTorch's lazy
// runtime initialisation, for one -- may have left an unrelated one behind), so that
// NR_LAUNCH_CHECK reports only errors of our own launches.
#ifdef hipLaunchKernelGGL
#undef hipLaunchKernelGGL
#endif
#define hipLaunchKernelGGL(kernelName, numBlocks, numThreads, memPerBlock, streamId, ...)                  \
    do {                                                                                                  \
        (void)hipGetLastError();                                                                          \
        kernelName<<<(numBlocks), (numThreads), (memPerBlock), (streamId)>>>(__VA_ARGS__);                \
    } while (0)

#define NR_LAUNCH_CHECK()                              \
    do {                                               \
        hipError_t e__ = hipGetLastError();            \
        if (e__ != hipSuccess) return (int)e__;        \
    } while (0)

// round-to-nearest-even f32 -> bf16 bits: the plain cast, which hipcc lowers to v_cvt_pk_bf16_f32 on gfx950 (one instruction
// where the integer form -- u += 0x7FFF + ((u >> 16) & 1) -- took five; the same bits for every finite input, and a NaN
// stays a NaN)
__device__ __forceinline__ uint16_t nr_f2bf(float f) {
    return __builtin_bit_cast(uint16_t, (__bf16)f);
}
__device__ __forceinline__ float nr_bf2f(uint16_t h) {
    return __builtin_bit_cast(float, ((uint32_t)h) << 16);
}
// two values at once: {bf16(a) in the low half, bf16(b) in the high half} -- ONE v_cvt_pk_bf16_f32
typedef __attribute__((ext_vector_type(2))) float nr_f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 nr_bf16x2_t;
__device__ __forceinline__ uint32_t nr_f2bf_pk(float a, float b) {
    const nr_f32x2_t v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, nr_bf16x2_t));
}
// split-bf16 pair of two values: hi = packed bf16(a), bf16(b); lo = packed bf16 of the remainders
__device__ __forceinline__ void nr_split_pk(float a, float b, uint32_t& hi, uint32_t& lo) {
    hi = nr_f2bf_pk(a, b);
    lo = nr_f2bf_pk(a - __builtin_bit_cast(float, hi << 16), b - __builtin_bit_cast(float, hi & 0xFFFF0000u));
}

// ---- cross-lane reductions on DPP (data-parallel primitives): VALU-latency steps instead of the
// LDS-crossbar round trip of ds_bpermute that __shfl_xor compiles to.  Steps: quad_perm xor-1, xor-2,
// row_half_mirror (8 lanes), row_mirror (16 lanes), row_bcast15 / row_bcast31 (across the four rows);
// the wave-wide result lands in lane 63 and is broadcast with v_readlane.
#define NR_DPP_XOR1 0xB1
#define NR_DPP_XOR2 0x4E
#define NR_DPP_HALF_MIRROR 0x141
#define NR_DPP_MIRROR 0x140
#define NR_DPP_BCAST15 0x142
#define NR_DPP_BCAST31 0x143

template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float nr_dpp(float old, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v),
                                                                  CTRL, ROW_MASK, 0xF, false));
}
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ int nr_dpp(int old, int v) {
    return __builtin_amdgcn_update_dpp(old, v, CTRL, ROW_MASK, 0xF, false);
}
__device__ __forceinline__ float nr_lane63(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// reductions over aligned groups of 8 lanes (result in all 8 lanes)
__device__ __forceinline__ float nr_group8_sum(float v) {
    v += nr_dpp<NR_DPP_XOR1>(v, v);
    v += nr_dpp<NR_DPP_XOR2>(v, v);
    v += nr_dpp<NR_DPP_HALF_MIRROR>(v, v);
    return v;
}
__device__ __forceinline__ float nr_group8_max(float v) {
    v = fmaxf(v, nr_dpp<NR_DPP_XOR1>(v, v));
    v = fmaxf(v, nr_dpp<NR_DPP_XOR2>(v, v));
    v = fmaxf(v, nr_dpp<NR_DPP_HALF_MIRROR>(v, v));
    return v;
}

__device__ __forceinline__ float nr_wave_sum(float v) {
    v = nr_group8_sum(v);
    v += nr_dpp<NR_DPP_MIRROR>(v, v);
    v += nr_dpp<NR_DPP_BCAST15, 0xA>(0.f, v);
    v += nr_dpp<NR_DPP_BCAST31, 0xC>(0.f, v);
    return nr_lane63(v);
}
__device__ __forceinline__ float nr_wave_max(float v) {
    v = nr_group8_max(v);
    v = fmaxf(v, nr_dpp<NR_DPP_MIRROR>(v, v));
    v = fmaxf(v, nr_dpp<NR_DPP_BCAST15, 0xA>(v, v));
    v = fmaxf(v, nr_dpp<NR_DPP_BCAST31, 0xC>(v, v));
    return nr_lane63(v);
}
__device__ __forceinline__ float nr_wave_min(float v) {
    v = fminf(v, nr_dpp<NR_DPP_XOR1>(v, v));
    v = fminf(v, nr_dpp<NR_DPP_XOR2>(v, v));
    v = fminf(v, nr_dpp<NR_DPP_HALF_MIRROR>(v, v));
    v = fminf(v, nr_dpp<NR_DPP_MIRROR>(v, v));
    v = fminf(v, nr_dpp<NR_DPP_BCAST15, 0xA>(v, v));
    v = fminf(v, nr_dpp<NR_DPP_BCAST31, 0xC>(v, v));
    return nr_lane63(v);
}

// wave arg-max / arg-min of (value, index) pairs; ties go to the LOWER index.  Result in all lanes.
template <bool MAX, int CTRL, int ROW_MASK>
__device__ __forceinline__ void nr_arg_step(float& v, int& idx) {
    float ov = nr_dpp<CTRL, ROW_MASK>(v, v);
    int oi = nr_dpp<CTRL, ROW_MASK>(idx, idx);
    bool take = MAX ? (ov > v || (ov == v && oi < idx)) : (ov < v || (ov == v && oi < idx));
    v = take ? ov : v;
    idx = take ? oi : idx;
}
template <bool MAX>
__device__ __forceinline__ void nr_wave_arg(float& v, int& idx) {
    nr_arg_step<MAX, NR_DPP_XOR1, 0xF>(v, idx);
    nr_arg_step<MAX, NR_DPP_XOR2, 0xF>(v, idx);
    nr_arg_step<MAX, NR_DPP_HALF_MIRROR, 0xF>(v, idx);
    nr_arg_step<MAX, NR_DPP_MIRROR, 0xF>(v, idx);
    nr_arg_step<MAX, NR_DPP_BCAST15, 0xA>(v, idx);
    nr_arg_step<MAX, NR_DPP_BCAST31, 0xC>(v, idx);
    v = nr_lane63(v);
    idx = __builtin_amdgcn_readlane(idx, 63);
}
__device__ __forceinline__ void nr_wave_argmax(float& v, int& idx) { nr_wave_arg<true>(v, idx); }
__device__ __forceinline__ void nr_wave_argmin(float& v, int& idx) { nr_wave_arg<false>(v, idx); }

// XCD-aware tile order for 1-D grids.  Workgroups are dealt round-robin to the 8 XCDs (workgroup b runs on XCD
// b % 8), each with its own 4 MiB L2.  Tiles that share operand rows should therefore sit on ONE XCD: XCD x takes
// the contiguous tile range [x*chunk, (x+1)*chunk) in the caller's (operand-sharing) order.  The grid must be
// launched with 8*chunk workgroups, chunk = ceil(n_tiles / 8); returns -1 for the padding workgroups.
__device__ __forceinline__ int nr_xcd_chunk_tile(int wg, int n_tiles) {
    const int chunk = (n_tiles + 7) >> 3;
    const int t = (wg & 7) * chunk + (wg >> 3);
    return t < n_tiles && (wg >> 3) < chunk ? t : -1;
}
static inline int nr_xcd_chunk_grid(int n_tiles) { return 8 * ((n_tiles + 7) / 8); }

