// Micro-benchmark: HBM streaming rate of y = relu(x*s+t) (bf16, 16-byte vectors, C = 64) by loop shape.
// build: hipcc --offload-arch=gfx950 -O3 -o stream_apply stream_apply.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int UNR, bool NT>
__global__ __launch_bounds__(256) void apply(const bf16x8* __restrict__ x, const float* __restrict__ sc,
                                             const float* __restrict__ sh, bf16x8* __restrict__ y, long total) {
    const long stride = (long)gridDim.x * blockDim.x;
    long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    const int cg = (int)(i % 8);  // stride is a multiple of 8: the channel group is loop-invariant
    float s[8], t[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        s[e] = sc[cg * 8 + e];
        t[e] = sh[cg * 8 + e];
    }
    for (; i < total; i += stride * UNR) {
        bf16x8 v[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u)
            if (i + u * stride < total) v[u] = NT ? __builtin_nontemporal_load(x + i + u * stride) : x[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNR; ++u)
            if (i + u * stride < total) {
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (__bf16)fmaxf((float)v[u][e] * s[e] + t[e], 0.f);
                y[i + u * stride] = o;
            }
    }
}

template <int UNR, bool NT>
void run(const char* name, const bf16x8* x, const float* sc, const float* sh, bf16x8* y, long total, int grid) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) apply<UNR, NT><<<grid, 256>>>(x, sc, sh, y, total);
    (void)hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) apply<UNR, NT><<<grid, 256>>>(x, sc, sh, y, total);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    ms /= 20;
    printf("%-10s grid %6d: %7.1f us  %5.2f TB/s\n", name, grid, ms * 1e3, 2.0 * total * 16 / (ms * 1e9));
}

int main() {
    const long total = 16L * 256 * 256 * 8;  // 16x256x256 pixels x 8 vectors (64 ch bf16) = 134 MB
    bf16x8 *x, *y;
    float *sc, *sh;
    (void)hipMalloc(&x, total * 16);
    (void)hipMalloc(&y, total * 16);
    (void)hipMalloc(&sc, 256);
    (void)hipMalloc(&sh, 256);
    (void)hipMemset(x, 0, total * 16);
    (void)hipMemset(sc, 0, 256);
    (void)hipMemset(sh, 0, 256);
    const int grids[] = {1024, 2048, 4096, 8192, 32768};
    for (int g : grids) {
        run<1, false>("unr1", x, sc, sh, y, total, g);
        run<2, false>("unr2", x, sc, sh, y, total, g);
        run<4, false>("unr4", x, sc, sh, y, total, g);
        run<8, false>("unr8", x, sc, sh, y, total, g);
        run<4, true>("unr4 nt", x, sc, sh, y, total, g);
    }
    return 0;
}
