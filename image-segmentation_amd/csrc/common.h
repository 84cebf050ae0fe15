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

