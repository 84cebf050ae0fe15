// Shared device/host helpers for the gfx950 kernels (wave = 64 lanes, MFMA, LDS).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/hipseg.h"

typedef __bf16 bf16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------- error plumbing (host)
void hipseg_set_error(const char* fmt, ...);

#define HS_REQUIRE(cond, ...)          \
    do {                               \
        if (!(cond)) {                 \
            hipseg_set_error(__VA_ARGS__); \
            return HIPSEG_EINVAL;      \
        }                              \
    } while (0)

#define HS_LAUNCH_CHECK(name)                                                    \
    do {                                                                         \
        hipError_t e_ = hipGetLastError();                                       \
        if (e_ != hipSuccess) {                                                  \
            hipseg_set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); \
            return HIPSEG_EHIP;                                                  \
        }                                                                        \
    } while (0)

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// hipFuncSetAttribute(MaxDynamicSharedMemorySize), done once per (device, kernel) and thread-safe (pack.hip)
int hs_set_max_lds(const void* kernel, size_t bytes);
// compute units of the current device, cached per device; 256 (MI355X) when no device is visible (host-only geometry
// queries in the build container)
int device_cus();

// ---------------------------------------------------------------- per-dtype traits
template <typename T>
struct VecOf;
template <>
struct VecOf<float> {
    static constexpr int N = 4;  // elements per 16-byte vector
    typedef f32x4 type;
};
template <>
struct VecOf<bf16> {
    static constexpr int N = 8;
    typedef bf16x8 type;
};

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16 v) { return (float)v; }
template <typename T>
__device__ __forceinline__ T from_f32(float v) { return (T)v; }

// 16-byte vector load/store of VecOf<T>::N elements into/from a float array
template <typename T>
__device__ __forceinline__ void load_vec(const T* p, float (&v)[VecOf<T>::N]) {
    typename VecOf<T>::type t = *reinterpret_cast<const typename VecOf<T>::type*>(p);
#pragma unroll
    for (int i = 0; i < VecOf<T>::N; ++i) v[i] = (float)t[i];
}
template <typename T>
__device__ __forceinline__ void store_vec(T* p, const float (&v)[VecOf<T>::N]) {
    typename VecOf<T>::type t;
#pragma unroll
    for (int i = 0; i < VecOf<T>::N; ++i) t[i] = (T)v[i];
    *reinterpret_cast<typename VecOf<T>::type*>(p) = t;
}

// ---------------------------------------------------------------- wave / block reductions
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---------------------------------------------------------------- shared reduction stage
// in-place pre-reduction of a [rows][RC] partial matrix: block (column block, chunk j) sums the rows of
// chunk j (double accumulation) into the chunk's FIRST row.  Deterministic, no extra workspace.
static __global__ __launch_bounds__(256) void colreduce_inplace_kernel(float* __restrict__ part, int rows, long RC, int chunk) {
    __shared__ double sm[4][64];
    const int cl = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const long c = (long)blockIdx.x * 64 + cl;
    const int r0 = blockIdx.y * chunk;
    int r1 = r0 + chunk;
    if (r1 > rows) r1 = rows;
    double s = 0.0;
    if (c < RC)
#pragma unroll 8
        for (int r = r0 + sl; r < r1; r += 4) s += (double)part[(size_t)r * RC + c];
    sm[sl][cl] = s;
    __syncthreads();
    if (sl == 0 && c < RC) part[(size_t)r0 * RC + c] = (float)((sm[0][cl] + sm[1][cl]) + (sm[2][cl] + sm[3][cl]));
}

// ---------------------------------------------------------------- LDS-DMA from inline assembly
// With the builtins the compiler's own wait-count insertion defeats the software pipelines (ROCm 7.2, found in the ISA):
//   * after __builtin_amdgcn_global_load_lds (a FLAT-encoded instruction) every later wait on an LDS read becomes
//     `s_waitcnt lgkmcnt(0)` -- counted waits are "invalid while a flat access is pending" -- so a fragment ring
//     issued 2-3 steps ahead is drained at every use (conv3_wstat_kernel: one full LDS latency per 4 MFMAs);
//   * the transposed-read builtin (ds_read_b64_tr_b16) carries no memory operand the compiler could prove disjoint
//     from a pending LDS-DMA write, so it puts `s_waitcnt vmcnt(0)` in front of EVERY such read that follows a DMA
//     piece: the "DMA" of the next tile then completes synchronously inside the current one (wgrad_dma_kernel).
// As opaque assembly the pieces stay in flight and the compiler keeps counting its LDS reads; landing is awaited
// explicitly (counted / zero vmcnt + barrier), as the kernels already did.
typedef int v4i_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v4i_t make_rsrc(const void* base, unsigned bytes) {
    const unsigned long long p = reinterpret_cast<unsigned long long>(base);
    return v4i_t{(int)(unsigned)p, (int)((unsigned)(p >> 32) & 0xffffu), (int)bytes, 0x00020000};
}
__device__ __forceinline__ void dma_piece(v4i_t rsrc, unsigned lds_addr, unsigned voff, unsigned soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :
                 : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff)
                 : "memory", "m0");  // (m0 is written: the compiler must not keep a value of its own there)
}

// per-lane 64-bit source pointer form (global_load_lds_dwordx4)
__device__ __forceinline__ void dma_piece_ptr(const void* src, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" : : "s"(lds_addr), "v"(src) : "memory", "m0");
}


