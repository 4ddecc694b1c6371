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

// round-to-nearest-even f32 -> bf16 bits (finite inputs)
__device__ __forceinline__ uint16_t nr_f2bf(float f) {
    uint32_t u = __builtin_bit_cast(uint32_t, f);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
__device__ __forceinline__ float nr_bf2f(uint16_t h) {
    return __builtin_bit_cast(float, ((uint32_t)h) << 16);
}

__device__ __forceinline__ float nr_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float nr_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float nr_wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
    return v;
}
