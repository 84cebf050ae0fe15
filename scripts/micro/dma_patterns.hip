// Micro-benchmark: per-CU LDS-DMA (global_load_lds_dwordx4) cost by SOURCE pattern, L2-warm 24-KiB region
// (192 pixels x 128 B), 24 pieces per iteration, 4 or 8 waves per CU, every CU busy.
//   0 contiguous      : piece = 1 KiB contiguous (8 pixels x 128 B), lanes in address order
//   1 octet gather    : piece = 16 B of each of 64 pixels (128-B stride)       [octet][pixel] LDS image
//   2 pixel-major xor : piece = 8 pixels x 128 B, 16-B chunks XOR-permuted inside each pixel row
//   3 half swap       : piece = 8 pixels x 128 B, 64-B halves swapped on odd pixel pairs (quads stay ordered)
// build: hipcc --offload-arch=gfx950 -O3 -o dma_patterns dma_patterns.hip
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

template <int PAT, int NW>
__global__ __launch_bounds__(NW * 64, 1) void fill(const unsigned char* __restrict__ src, int iters, unsigned* out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned acc = 0;
    for (int it = 0; it < iters; ++it) {
        for (int p = wave; p < 24; p += NW) {
            const unsigned char* s;
            if (PAT == 0) {
                s = src + p * 1024 + lane * 16;
            } else if (PAT == 1) {
                const int oct = p / 3, g = p % 3;
                s = src + (g * 64 + lane) * 128 + oct * 16;
            } else if (PAT == 2) {
                const int px = lane >> 3, slot = lane & 7, P = p * 8 + px;
                s = src + P * 128 + ((slot ^ ((P >> 1) & 7)) * 16);
            } else {
                const int px = lane >> 3, slot = lane & 7, P = p * 8 + px;
                s = src + P * 128 + ((slot ^ (((P >> 1) & 1) << 2)) * 16);
            }
            __builtin_amdgcn_global_load_lds((glb_void*)s, (lds_void*)(smem + p * 1024), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        acc += *reinterpret_cast<unsigned*>(smem + ((it * 64 + lane) * 16) % (24 * 1024));
        __syncthreads();
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int PAT, int NW>
float run(const unsigned char* src, int iters, unsigned* out) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fill<PAT, NW>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              24 * 1024);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    fill<PAT, NW><<<256, NW * 64, 24 * 1024>>>(src, 10, out);
    (void)hipEventRecord(e0);
    fill<PAT, NW><<<256, NW * 64, 24 * 1024>>>(src, iters, out);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    const int iters = 2000;
    unsigned char* src;
    unsigned* out;
    (void)hipMalloc(&src, 64 * 1024);
    (void)hipMalloc(&out, 4);
    (void)hipMemset(src, 1, 64 * 1024);
    const char* names[4] = {"contiguous", "octet gather", "pixel-major xor16", "half swap"};
    float ms[4][2];
    ms[0][0] = run<0, 4>(src, iters, out); ms[0][1] = run<0, 8>(src, iters, out);
    ms[1][0] = run<1, 4>(src, iters, out); ms[1][1] = run<1, 8>(src, iters, out);
    ms[2][0] = run<2, 4>(src, iters, out); ms[2][1] = run<2, 8>(src, iters, out);
    ms[3][0] = run<3, 4>(src, iters, out); ms[3][1] = run<3, 8>(src, iters, out);
    for (int m = 0; m < 4; ++m)
        printf("%-18s 4 waves: %6.1f ns/KiB/CU | 8 waves: %6.1f ns/KiB/CU\n", names[m], ms[m][0] * 1e6 / (24.0 * iters),
               ms[m][1] * 1e6 / (24.0 * iters));
    return 0;
}
