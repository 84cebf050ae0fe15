// Micro-benchmark: per-CU cost of filling LDS from an L2-warm 36-KiB region (the weight chunk every workgroup
// re-reads) by (a) LDS-DMA global_load_lds_dwordx4, (b) global_load_dwordx4 + ds_write_b128.
// build: hipcc --offload-arch=gfx950 -O3 -o dma_vs_regs dma_vs_regs.hip ; run: ./dma_vs_regs
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

template <int MODE, int NW>
__global__ __launch_bounds__(NW * 64, 1) void fill(const uint4* __restrict__ src, int region_pieces, int iters,
                                                  unsigned* out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned acc = 0;
    for (int it = 0; it < iters; ++it) {
        // every wave moves region_pieces / NW pieces of 1 KiB per iteration
        if (MODE == 0) {
            for (int p = wave; p < region_pieces; p += NW) {
                const uint4* s = src + (size_t)p * 64 + lane;
                __builtin_amdgcn_global_load_lds((glb_void*)s, (lds_void*)(smem + p * 1024), 16, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            uint4 v[8];
            for (int p0 = wave; p0 < region_pieces; p0 += NW * 8) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int p = p0 + u * NW;
                    if (p < region_pieces) v[u] = src[(size_t)p * 64 + lane];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int p = p0 + u * NW;
                    if (p < region_pieces) *reinterpret_cast<uint4*>(smem + p * 1024 + lane * 16) = v[u];
                }
            }
        }
        __syncthreads();
        acc += *reinterpret_cast<unsigned*>(smem + ((it * 64 + lane) * 16) % (region_pieces * 1024));
        __syncthreads();
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int MODE, int NW>
float run(const uint4* src, int pieces, int iters, unsigned* out, int grid) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&fill<MODE, NW>), hipFuncAttributeMaxDynamicSharedMemorySize,
                        pieces * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    fill<MODE, NW><<<grid, NW * 64, pieces * 1024>>>(src, pieces, 10, out);
    hipEventRecord(e0);
    fill<MODE, NW><<<grid, NW * 64, pieces * 1024>>>(src, pieces, iters, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    const int pieces = 36, iters = 2000, grid = 256;
    uint4* src;
    unsigned* out;
    hipMalloc(&src, pieces * 1024);
    hipMalloc(&out, 4);
    hipMemset(src, 1, pieces * 1024);
    const char* names[2] = {"lds-dma", "regs+ds_write"};
    for (int mode = 0; mode < 2; ++mode) {
        float ms4 = mode == 0 ? run<0, 4>(src, pieces, iters, out, grid) : run<1, 4>(src, pieces, iters, out, grid);
        float ms8 = mode == 0 ? run<0, 8>(src, pieces, iters, out, grid) : run<1, 8>(src, pieces, iters, out, grid);
        // per CU: pieces * iters KiB
        const double kb = (double)pieces * iters;
        printf("%-14s 4 waves: %.3f ms  %.1f ns/KiB/CU  %.1f GB/s/CU | 8 waves: %.3f ms  %.1f ns/KiB/CU  %.1f GB/s/CU\n",
               names[mode], ms4, ms4 * 1e6 / kb, kb * 1024 / (ms4 * 1e6), ms8, ms8 * 1e6 / kb, kb * 1024 / (ms8 * 1e6));
    }
    return 0;
}
