// How fast can one CU stage operand tiles from L2 into LDS?  (standalone probe: hipcc --offload-arch=gfx950)
//   mode 0: LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction)
//   mode 1: global_load_dwordx4 -> VGPR -> ds_write_b128
//   mode 2: half the pieces by DMA, half through registers
//   mode 3: LDS-DMA of pieces of 16 rows x 64 B (a 32-element K slice: HALF of every 128-byte line per slice)
// Access pattern of the GEMM tile engine: a piece = 8 rows x 128 B, rows 1 KiB apart in memory (K = 512 bf16).
// Every workgroup re-reads its own 2 x 192 KiB region (L2-resident), `iters` slices of `PIECES` pieces each.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

template <int MODE, int NW>
__global__ __launch_bounds__(64 * NW) void fill_kernel(const char* __restrict__ g, int iters, int pieces_per_wave, int regions, int n_mfma, int n_dsread,
                                                       unsigned long long* cycles, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const char* base = g + (size_t)(blockIdx.x % regions) * (384 * 1024);
    // lane -> (row lane/8 of the piece, 16-byte chunk lane%8)
    const size_t lane_off = MODE == 3 ? (size_t)(lane >> 2) * 1024 + (lane & 3) * 16 : (size_t)(lane >> 3) * 1024 + (lane & 7) * 16;
    float acc = 0.f;
    f32x4_t macc[8];
    for (int i = 0; i < 8; ++i) macc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        const int kb = MODE == 3 ? (it & 15) * 64 : (it & 7) * 128;      // walk along K like the GEMM does
        for (int i = 0; i < pieces_per_wave; ++i) {
            const int piece = wave + NW * i;                   // 8 rows each
            const char* src = base + (size_t)piece * (MODE == 3 ? 16 : 8) * 1024 + lane_off + kb;
            char* dst = smem + (it & 1) * (NW * pieces_per_wave * 1024) + piece * 1024;
            bool dma = MODE == 0 || MODE == 3 || (MODE == 2 && (i & 1) == 0);
            if (dma) {
                __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)dst, 16, 0, 0);
            } else {
                f32x4_t v = *reinterpret_cast<const f32x4_t*>(src);
                *reinterpret_cast<f32x4_t*>(dst + lane * 16) = v;
            }
        }
        // the previous slice must have landed; this one stays in flight (counted wait needs a literal: 6 / 9 / 12)
        if (MODE == 0 || MODE == 3) {
            if (pieces_per_wave == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else if (pieces_per_wave == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        acc += *reinterpret_cast<float*>(smem + ((lane * 68 + it * 4) & 0xFFFC));   // keep the LDS writes alive
        // optional compute beside the fills: fragment reads of the OTHER buffer + MFMAs on them
        if (n_mfma > 0) {
            const char* rb = smem + ((it + 1) & 1) * (NW * pieces_per_wave * 1024);
            bf16x8_t fr[4];
            for (int q = 0; q < n_dsread; q += 4) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    fr[u] = *reinterpret_cast<const bf16x8_t*>(rb + (((q + u) * 1024 + lane * 16) & (NW * pieces_per_wave * 1024 - 16)));
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    for (int r = 0; r < n_mfma / n_dsread; ++r)
                        macc[(q + u + r) & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[u], fr[(u + 1) & 3], macc[(q + u + r) & 7], 0, 0, 0);
            }
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    for (int i = 0; i < 8; ++i) acc += macc[i][0] + macc[i][3];
    if (acc == 123.456f) sink[0] = acc;
}

template <int MODE, int NW>
void run(const char* name, const char* g, int pieces_per_wave, int regions, int n_mfma, int n_dsread, unsigned long long* d_cycles, float* d_sink) {
    const int iters = 2000, grid = 256;
    size_t lds = (size_t)2 * NW * pieces_per_wave * 1024;
    hipFuncSetAttribute((const void*)fill_kernel<MODE, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((fill_kernel<MODE, NW>), dim3(grid), dim3(64 * NW), lds, 0, g, 50, pieces_per_wave, regions, n_mfma, n_dsread, d_cycles, d_sink);
    hipEventRecord(e0);
    hipLaunchKernelGGL((fill_kernel<MODE, NW>), dim3(grid), dim3(64 * NW), lds, 0, g, iters, pieces_per_wave, regions, n_mfma, n_dsread, d_cycles, d_sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> c(grid);
    hipMemcpy(c.data(), d_cycles, grid * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double avg = 0;
    for (auto x : c) avg += (double)x;
    avg /= grid;
    double bytes_per_wg = (double)iters * NW * pieces_per_wave * 1024;
    printf("%-24s mfma %3d dsread %2d regions %3d waves %d, %3d KiB/slice: %6.1f B/clk/CU (in-kernel cycles), %6.1f GB/s/CU, chip %5.2f TB/s\n", name, n_mfma, n_dsread, regions, NW,
           NW * pieces_per_wave, bytes_per_wg / avg, bytes_per_wg / (ms * 1e-3) / 1e9, bytes_per_wg * grid / (ms * 1e-3) / 1e12);
}

int main() {
    char* g;
    size_t bytes = (size_t)256 * 384 * 1024 + (1 << 20);
    hipMalloc(&g, bytes);
    hipMemset(g, 1, bytes);
    unsigned long long* d_cycles;
    float* d_sink;
    hipMalloc(&d_cycles, 256 * sizeof(unsigned long long));
    hipMalloc(&d_sink, 4);
    for (int regions : {8, 256}) {
        for (int nm : {0, 24, 72, 144}) {
            run<0, 8>("LDS-DMA (2-deep)", g, 9, regions, nm, 24, d_cycles, d_sink);
        }
        run<0, 8>("LDS-DMA (2-deep)", g, 9, regions, 24, 8, d_cycles, d_sink);
        run<0, 8>("LDS-DMA (2-deep)", g, 9, regions, 72, 8, d_cycles, d_sink);
        // half-line pieces (K slices of 32): 36 KiB per slice on 4 waves as the tap-block convolution tried them, full-line pieces beside
        for (int nm : {0, 72}) {
            run<3, 4>("LDS-DMA 16 x 64 B pieces", g, 9, regions, nm, 24, d_cycles, d_sink);
            run<0, 4>("LDS-DMA  8 x 128 B pieces", g, 9, regions, nm, 24, d_cycles, d_sink);
            run<3, 8>("LDS-DMA 16 x 64 B pieces", g, 9, regions, nm, 24, d_cycles, d_sink);
        }
    }
    return 0;
}
