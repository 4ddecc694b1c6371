// Reduction of the per-row loss terms rowloss[2][4][B] to the five losses -- shared by nr_loss_finalize and by the
// launches that finalize themselves (the workgroup that finishes last, nr_rowloss.hip / nr_sinkhorn.hip).
#pragma once
#include "nr_common.h"

// losses = (total, centrality, uniform, neighbour, kl)   (modeling.py:329-358).  Called by EVERY thread of a workgroup
// of >= 256 threads (it holds a barrier); waves 0..3 do the work
template <bool COHERENT>
__device__ __forceinline__ void nr_loss_finalize_body(const float* __restrict__ rowloss, int B, float wu, float wn, float wkl,
                                                      float* __restrict__ losses) {
    __shared__ float red[8][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // wave w handles term w; both directions
    float acc0 = 0.f, acc1 = 0.f;
    for (int j = lane; wave < 4 && j < B; j += 64) {
        const float* p0 = rowloss + (size_t)(0 * 4 + wave) * B + j;
        const float* p1 = rowloss + (size_t)(1 * 4 + wave) * B + j;
        if constexpr (COHERENT) {      // written by other workgroups of the same launch: read past the L1
            acc0 += __hip_atomic_load(p0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            acc1 += __hip_atomic_load(p1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            acc0 += *p0;
            acc1 += *p1;
        }
    }
    acc0 = nr_wave_sum(acc0);
    acc1 = nr_wave_sum(acc1);
    if (lane == 0 && wave < 4) { red[wave][0] = acc0; red[wave][1] = acc1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float invB = 1.0f / (float)B;
        float c = (red[0][0] * invB + red[0][1] * invB) * 0.5f;
        float u = (red[1][0] * invB + red[1][1] * invB) * 0.5f;
        float n = (red[2][0] * invB + red[2][1] * invB) * 0.5f;
        float k = (red[3][0] * invB * invB + red[3][1] * invB * invB) * 0.5f;   // kl_div 'mean' divides by B*B
        losses[0] = c + u * wu + n * wn + k * wkl;
        losses[1] = c; losses[2] = u; losses[3] = n; losses[4] = k;
    }
}

