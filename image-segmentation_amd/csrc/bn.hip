// BatchNorm2d (+ReLU, +MaxPool2d(2,2)) forward/backward pieces and per-channel reductions, NHWC.
// All kernels are HBM-bound: 16-byte vector accesses (8 bf16 / 4 f32 channels per lane), fp32
// math, per-channel sums reduced lane -> LDS -> per-block partial -> fixed-order finalize.
#include <stdlib.h>

#include "common.h"

namespace {

template <typename T, int V>
__device__ __forceinline__ void ldv(const T* p, float (&v)[V]) {
    if constexpr (V == 1) {
        v[0] = (float)p[0];
    } else {
        const typename VecOf<T>::type t = *reinterpret_cast<const typename VecOf<T>::type*>(p);
#pragma unroll
        for (int e = 0; e < V; ++e) v[e] = (float)t[e];
    }
}
template <typename T, int V>
__device__ __forceinline__ void stv(T* p, const float (&v)[V]) {
    if constexpr (V == 1) {
        p[0] = (T)v[0];
    } else {
        typename VecOf<T>::type t;
#pragma unroll
        for (int e = 0; e < V; ++e) t[e] = (T)v[e];
        *reinterpret_cast<typename VecOf<T>::type*>(p) = t;
    }
}

// V consecutive per-channel fp32 constants as 16-byte loads (V = 8 / 4: channel vectors are 32- / 16-byte aligned),
// all issued before any use -- scalar loads interleaved with selects serialise into one memory round trip per element
template <int V>
__device__ __forceinline__ void ldc(const float* __restrict__ p, float (&v)[V]) {
    if constexpr (V % 4 == 0) {
#pragma unroll
        for (int q = 0; q < V / 4; ++q) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(p + 4 * q);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[4 * q + e] = t[e];
        }
    } else {
#pragma unroll
        for (int e = 0; e < V; ++e) v[e] = p[e];
    }
}

// ------------------------------------------------------------------ finalize of the conv-epilogue statistics
// stats: [mtiles][2][C]; block = 16 channels x 16 tile slices, double accumulation.
// stats: [mtiles][2][C].  One launch: block = 4 channels x 64 tile slices (each lane sums <= mtiles/64 rows,
// all loads independent), double accumulation, LDS tree over the slices, then the per-channel finish.
// One block per channel: 256 row slices, so a 2048-row statistics workspace is 8 independent loads per thread (one
// batch of L2 round trips) instead of 32 dependent-looking iterations on C / 4 blocks (the kernel is latency-bound:
// 16 blocks for a 64-channel layer took 5.9 us).  Fixed summation order (deterministic).
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ stats, int mtiles, int rstride,
                                                           int C, double count, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float eps, float momentum,
                                                           float* running_mean, float* running_var,
                                                           long long* nbt, float* mean, float* invstd, float* scale,
                                                           float* shift) {
    __shared__ double sS[256], sQ[256];
    const int sl = threadIdx.x, c = blockIdx.x;
    double S = 0.0, Q = 0.0;
#pragma unroll 8
    for (int t = sl; t < mtiles; t += 256) {
        S += (double)stats[((size_t)t * rstride * 2 + 0) * C + c];
        Q += (double)stats[((size_t)t * rstride * 2 + 1) * C + c];
    }
    sS[sl] = S;
    sQ[sl] = Q;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (sl < o) {
            sS[sl] += sS[sl + o];
            sQ[sl] += sQ[sl + o];
        }
        __syncthreads();
    }
    if (sl == 0) {
        S = sS[0];
        Q = sQ[0];
        const double m = S / count;
        double var = Q / count - m * m;
        if (var < 0.0) var = 0.0;
        const float is = (float)(1.0 / sqrt(var + (double)eps));
        const float sc = gamma[c] * is;
        mean[c] = (float)m;
        invstd[c] = is;
        scale[c] = sc;
        shift[c] = beta[c] - (float)m * sc;
        if (running_mean) {
            const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
            running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)m;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
        }
    }
    if (nbt && blockIdx.x == 0 && threadIdx.x == 0) nbt[0] += 1;
}

__global__ void bn_eval_params_kernel(const float* gamma, const float* beta, const float* rm, const float* rv, float eps,
                                      int C, float* mean, float* invstd, float* scale, float* shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) {
        const float is = 1.f / sqrtf(rv[c] + eps);
        const float sc = gamma[c] * is;
        mean[c] = rm[c];
        invstd[c] = is;
        scale[c] = sc;
        shift[c] = beta[c] - rm[c] * sc;
    }
}

// ------------------------------------------------------------------ y = [maxpool2x2](relu(x*scale + shift))
#ifndef BN_APPLY_UNR
#define BN_APPLY_UNR 1
#endif
#ifndef BN_APPLY_MAXBLK
#define BN_APPLY_MAXBLK 4096
#endif
// Grid-stride over (output pixel, channel vector) items; BN_APPLY_UNR independent items per lane and iteration, all
// their loads issued before the first use (a single 16-byte load per lane leaves too few bytes in flight per CU to
// cover the HBM latency at full bandwidth).
template <typename T, int V, bool POOL>
__global__ __launch_bounds__(256) void bn_relu_apply_kernel(const T* __restrict__ x, const float* __restrict__ scale,
                                                             const float* __restrict__ shift, T* __restrict__ y,
                                                             int B, int H, int W, int C) {
    constexpr int UNR = POOL ? 1 : BN_APPLY_UNR;
    const int CG = C / V;
    const int Ho = POOL ? H / 2 : H, Wo = POOL ? W / 2 : W;
    const long total = (long)B * Ho * Wo * CG;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i0 = blockIdx.x * (long)blockDim.x + threadIdx.x; i0 < total; i0 += stride * UNR) {
        float v[UNR][POOL ? 4 : 1][V];
        long op[UNR];
        int cg[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const long i = i0 + u * stride;
            if (i < total) {
                cg[u] = (int)(i % CG);
                op[u] = i / CG;
                if (!POOL) {
                    ldv<T, V>(x + op[u] * C + cg[u] * V, v[u][0]);
                } else {
                    const int xo = (int)(op[u] % Wo);
                    const int yo = (int)((op[u] / Wo) % Ho);
                    const long n = op[u] / ((long)Wo * Ho);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const long ip = (n * H + 2 * yo + (k >> 1)) * W + 2 * xo + (k & 1);
                        ldv<T, V>(x + ip * C + cg[u] * V, v[u][k]);
                    }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const long i = i0 + u * stride;
            if (i < total) {
                float sc[V], sh[V], o[V];
                ldc<V>(scale + cg[u] * V, sc);
                ldc<V>(shift + cg[u] * V, sh);
#pragma unroll
                for (int e = 0; e < V; ++e) o[e] = 0.f;  // relu output >= 0
#pragma unroll
                for (int k = 0; k < (POOL ? 4 : 1); ++k)
#pragma unroll
                    for (int e = 0; e < V; ++e) o[e] = fmaxf(o[e], v[u][k][e] * sc[e] + sh[e]);
                stv<T, V>(y + op[u] * C + cg[u] * V, o);
            }
        }
    }
}

// ------------------------------------------------------------------ block geometry of the channel reductions
// thread -> (channel group cg = tid % CG, pixel lane pl = tid / CG); PL = 256 / CG pixel lanes.
#ifndef BN_MAXBLK
#define BN_MAXBLK 1024
#endif
#ifndef BN_UNR
#define BN_UNR 2
#endif
struct RedGeo {
    int CG, PL, nblk;
    long ppb;  // pixels per block
};
inline RedGeo red_geo(long npix, int C, int V) {
    RedGeo g;
    g.CG = C / V;
    g.PL = 256 / g.CG;
    long nb = (npix + (long)g.PL * 8 - 1) / ((long)g.PL * 8);
    if (nb > BN_MAXBLK) nb = BN_MAXBLK;
    if (nb < 1) nb = 1;
    g.nblk = (int)nb;
    g.ppb = (npix + nb - 1) / nb;
    return g;
}
inline int vec_for(int C, int dtype) {
    const int v = dtype == HIPSEG_BF16 ? 8 : 4;
    return (C % v == 0) ? v : 1;
}

// One output pixel's worth of backward state.  issue() only issues the global loads (so several
// pixels can be in flight per lane before the first use), finish() routes dy through maxpool + relu.
// Shared by reduce and apply so both make identical decisions.
// TWO: the gradient arrives as two tensors (the block's output has two consumers -- an encoder block's pooled output
// feeds the next block AND a decoder's skip input, /root/reference/models/UNet.py:64-72 -- and each consumer's gradient
// is handed over on its own instead of being summed by a separate elementwise pass): dy := round_T(dy + dy2), the value
// that pass would have written.
template <typename T, int V, bool POOL, bool TWO = false>
struct PixelCtx {
    static constexpr int NP = POOL ? 4 : 1;
    float xh[NP][V];  // xhat, after finish()
    float g[NP][V];   // routed gradient, after finish()
    float xr[NP][V];  // raw x
    float gv[V];      // raw dy
    float gv2[TWO ? V : 1];
    long ip[NP];
    __device__ __forceinline__ void issue(const T* x, const T* dy, const T* dy2, long op, int cg, int H, int W, int C) {
        ldv<T, V>(dy + op * C + cg * V, gv);
        if constexpr (TWO) ldv<T, V>(dy2 + op * C + cg * V, gv2);
        if (!POOL) {
            ip[0] = op;
            ldv<T, V>(x + op * C + cg * V, xr[0]);
        } else {
            const int Ho = H / 2, Wo = W / 2;
            const int xo = (int)(op % Wo);
            const int yo = (int)((op / Wo) % Ho);
            const long n = op / ((long)Wo * Ho);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                ip[k] = (n * H + 2 * yo + (k >> 1)) * W + 2 * xo + (k & 1);
                ldv<T, V>(x + ip[k] * C + cg * V, xr[k]);
            }
        }
    }
    __device__ __forceinline__ void finish(const float (&mean)[V], const float (&invstd)[V], const float (&sc)[V],
                                           const float (&sh)[V]) {
        if constexpr (TWO) {
#pragma unroll
            for (int e = 0; e < V; ++e) gv[e] = to_f32(from_f32<T>(gv[e] + gv2[e]));
        }
        if (!POOL) {
#pragma unroll
            for (int e = 0; e < V; ++e) {
                xh[0][e] = (xr[0][e] - mean[e]) * invstd[e];
                g[0][e] = (xr[0][e] * sc[e] + sh[e] > 0.f) ? gv[e] : 0.f;
            }
        } else {
            float best[V];
            int bk[V];
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int e = 0; e < V; ++e) {
                    xh[k][e] = (xr[k][e] - mean[e]) * invstd[e];
                    const float yv = fmaxf(xr[k][e] * sc[e] + sh[e], 0.f);
                    if (k == 0 || yv > best[e]) {  // first maximum wins (window scan order)
                        best[e] = yv;
                        bk[e] = k;
                    }
                }
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int e = 0; e < V; ++e) g[k][e] = (bk[e] == k && best[e] > 0.f) ? gv[e] : 0.f;
        }
    }
};

template <typename T, int V, bool POOL, bool TWO>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const T* __restrict__ dy, const T* __restrict__ dy2,
                                                             const T* __restrict__ x,
                                                             const float* __restrict__ mean,
                                                             const float* __restrict__ invstd,
                                                             const float* __restrict__ scale,
                                                             const float* __restrict__ shift,
                                                             float* __restrict__ partial, long npix_out, int H, int W,
                                                             int C, int CG, int PL, long ppb) {
    __shared__ float red[2 * 2048];
    const int tid = threadIdx.x;
    const int cg = tid % CG, pl = tid / CG;
    float s1[V], s2[V];
#pragma unroll
    for (int e = 0; e < V; ++e) s1[e] = s2[e] = 0.f;
    if (pl < PL) {
        float mn[V], is[V], sc[V], sh[V];
        ldc<V>(mean + cg * V, mn);
        ldc<V>(invstd + cg * V, is);
        ldc<V>(scale + cg * V, sc);
        ldc<V>(shift + cg * V, sh);
        const long start = blockIdx.x * ppb;
        long end = start + ppb;
        if (end > npix_out) end = npix_out;
        constexpr int UNR = POOL ? 2 : BN_UNR;  // pixels in flight per lane
        for (long op0 = start + pl; op0 < end; op0 += (long)PL * UNR) {
            PixelCtx<T, V, POOL, TWO> ctx[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u)
                if (op0 + (long)u * PL < end) ctx[u].issue(x, dy, dy2, op0 + (long)u * PL, cg, H, W, C);
#pragma unroll
            for (int u = 0; u < UNR; ++u)
                if (op0 + (long)u * PL < end) {
                    ctx[u].finish(mn, is, sc, sh);
#pragma unroll
                    for (int k = 0; k < (POOL ? 4 : 1); ++k)
#pragma unroll
                        for (int e = 0; e < V; ++e) {
                            s1[e] += ctx[u].g[k][e];
                            s2[e] += ctx[u].g[k][e] * ctx[u].xh[k][e];
                        }
                }
        }
    }
    // reduce over the pixel lanes through LDS: red[r][pl][c] (PL * C <= 2048 floats per r for every vector width:
    // one round), then COLUMN-parallel sums -- thread t adds column t over the PL rows in row order (the same
    // order the former single-lane loop used, so results are bit-identical) with conflict-free consecutive reads.
    float* r1 = red;
    float* r2 = red + 2048;
    if (pl < PL) {
#pragma unroll
        for (int e = 0; e < V; ++e) {
            r1[pl * C + cg * V + e] = s1[e];
            r2[pl * C + cg * V + e] = s2[e];
        }
    }
    __syncthreads();
    for (int col = tid; col < 2 * C; col += 256) {
        const int arr = col / C, c = col - arr * C;
        const float* rr = arr ? r2 : r1;
        float acc = 0.f;
        for (int j = 0; j < PL; ++j) acc += rr[j * C + c];
        partial[((size_t)blockIdx.x * 2 + arr) * C + c] = acc;
    }
}

template <typename T, int V, bool POOL, bool TWO>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ dy, const T* __restrict__ dy2,
                                                            const T* __restrict__ x,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ invstd,
                                                            const float* __restrict__ scale,
                                                            const float* __restrict__ shift,
                                                            const float* __restrict__ sums, float inv_count, int eval,
                                                            T* __restrict__ dx, float* dbias, long npix_out, int H,
                                                            int W, int C, int CG, int PL, long ppb) {
    __shared__ float red[2048];
    const int tid = threadIdx.x;
    const int cg = tid % CG, pl = tid / CG;
    float sb[V];
#pragma unroll
    for (int e = 0; e < V; ++e) sb[e] = 0.f;
    if (pl < PL) {
        float mn[V], is[V], sc[V], sh[V], k1[V], k2[V];
        ldc<V>(mean + cg * V, mn);
        ldc<V>(invstd + cg * V, is);
        ldc<V>(scale + cg * V, sc);
        ldc<V>(shift + cg * V, sh);
        ldc<V>((eval ? mean : sums) + cg * V, k1);      // dbeta / N   (eval: any valid address, value unused)
        ldc<V>((eval ? mean : sums + C) + cg * V, k2);  // dgamma / N
#pragma unroll
        for (int e = 0; e < V; ++e) {
            k1[e] = eval ? 0.f : k1[e] * inv_count;
            k2[e] = eval ? 0.f : k2[e] * inv_count;
        }
        const long start = blockIdx.x * ppb;
        long end = start + ppb;
        if (end > npix_out) end = npix_out;
        constexpr int UNR = POOL ? 2 : BN_UNR;  // pixels in flight per lane
        for (long op0 = start + pl; op0 < end; op0 += (long)PL * UNR) {
            PixelCtx<T, V, POOL, TWO> ctx[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u)
                if (op0 + (long)u * PL < end) ctx[u].issue(x, dy, dy2, op0 + (long)u * PL, cg, H, W, C);
#pragma unroll
            for (int u = 0; u < UNR; ++u)
                if (op0 + (long)u * PL < end) {
                    ctx[u].finish(mn, is, sc, sh);
#pragma unroll
                    for (int k = 0; k < (POOL ? 4 : 1); ++k) {
                        float o[V];
#pragma unroll
                        for (int e = 0; e < V; ++e) {
                            o[e] = sc[e] * (ctx[u].g[k][e] - k1[e] - ctx[u].xh[k][e] * k2[e]);
                            sb[e] += o[e];
                        }
                        stv<T, V>(dx + ctx[u].ip[k] * C + cg * V, o);
                    }
                }
        }
    }
    if (dbias) {  // conv-bias gradient = per-channel sum of dx (uniform branch)
        const int R = 2048 / C;
        float t[V];
#pragma unroll
        for (int e = 0; e < V; ++e) t[e] = 0.f;
        for (int base = 0; base < PL; base += R) {
            __syncthreads();
            if (pl >= base && pl < base + R && pl < PL) {
#pragma unroll
                for (int e = 0; e < V; ++e) red[(pl - base) * C + cg * V + e] = sb[e];
            }
            __syncthreads();
            if (pl == 0) {
                const int lim = (PL - base) < R ? (PL - base) : R;
                for (int j = 0; j < lim; ++j)
#pragma unroll
                    for (int e = 0; e < V; ++e) t[e] += red[j * C + cg * V + e];
            }
        }
        if (pl == 0) {
#pragma unroll
            for (int e = 0; e < V; ++e) atomicAdd(dbias + cg * V + e, t[e]);
        }
    }
}

// ------------------------------------------------------------------ per-channel sum over pixels
template <typename T, int V>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ x, float* __restrict__ partial, long npix,
                                                      int C, int CG, int PL, long ppb) {
    __shared__ float red[2048];
    const int tid = threadIdx.x;
    const int cg = tid % CG, pl = tid / CG;
    float s[V];
#pragma unroll
    for (int e = 0; e < V; ++e) s[e] = 0.f;
    if (pl < PL) {
        const long start = blockIdx.x * ppb;
        long end = start + ppb;
        if (end > npix) end = npix;
        for (long p = start + pl; p < end; p += PL) {
            float v[V];
            ldv<T, V>(x + p * C + cg * V, v);
#pragma unroll
            for (int e = 0; e < V; ++e) s[e] += v[e];
        }
    }
    // PL * C <= 2048 floats: one LDS round, then column-parallel sums in row order (see bn_bwd_reduce_kernel)
    if (pl < PL) {
#pragma unroll
        for (int e = 0; e < V; ++e) red[pl * C + cg * V + e] = s[e];
    }
    __syncthreads();
    for (int col = tid; col < C; col += 256) {
        float acc = 0.f;
        for (int j = 0; j < PL; ++j) acc += red[j * C + col];
        partial[(size_t)blockIdx.x * C + col] = acc;
    }
}

// out[r][c] = sum_blk partial[blk][r][c]; block = 4 columns x 64 block-slices (a 16-byte run per slice: 2048 partial
// rows are 32 loads per thread in two batches of 16; with 16 x 16 it was 128 loads on RC / 16 = 8..64 blocks), fixed
// order, one launch.
__global__ __launch_bounds__(256) void colsum_finalize_kernel(const float* __restrict__ partial, int nblk, int RC,
                                                               float* __restrict__ out, float* __restrict__ zero_out,
                                                               int nzero) {
    if (zero_out)  // rides along: an exactly-zero gradient vector (conv bias in front of a train-mode BatchNorm)
        for (int i = blockIdx.x * 256 + threadIdx.x; i < nzero; i += gridDim.x * 256) zero_out[i] = 0.f;
    __shared__ float sm[64][5];
    const int cl = threadIdx.x & 3, sl = threadIdx.x >> 2;
    const int c = blockIdx.x * 4 + cl;
    float s = 0.f;
    if (c < RC) {
#pragma unroll 16
        for (int b = sl; b < nblk; b += 64) s += partial[(size_t)b * RC + c];
    }
    sm[sl][cl] = s;
    __syncthreads();
    if (sl == 0 && c < RC) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 64; ++i) t += sm[i][cl];
        out[c] = t;
    }
}

#define DISPATCH_TV(dtype, V, ...)                              \
    do {                                                        \
        if (dtype == HIPSEG_BF16) {                             \
            typedef bf16 T_;                                    \
            if (V == 8) {                                       \
                constexpr int V_ = 8;                           \
                __VA_ARGS__                                     \
            } else {                                            \
                constexpr int V_ = 1;                           \
                __VA_ARGS__                                     \
            }                                                   \
        } else {                                                \
            typedef float T_;                                   \
            if (V == 4) {                                       \
                constexpr int V_ = 4;                           \
                __VA_ARGS__                                     \
            } else {                                            \
                constexpr int V_ = 1;                           \
                __VA_ARGS__                                     \
            }                                                   \
        }                                                       \
    } while (0)

}  // namespace

extern "C" int hipseg_bn_finalize(float* stats, int mtiles, int C, double count, const float* gamma,
                                  const float* beta, float eps, float momentum, float* running_mean,
                                  float* running_var, int64_t* num_batches_tracked, float* mean, float* invstd,
                                  float* scale, float* shift, hipseg_stream_t stream) {
    HS_REQUIRE(stats && gamma && beta && mean && invstd && scale && shift && C > 0 && mtiles > 0 && count > 0,
               "bn_finalize: bad arguments");
    HS_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "bn_finalize: running_mean/var mismatch");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    int rows = mtiles, rstride = 1;
    if (mtiles > 2048) {  // full-resolution layers: wide in-place pre-reduction of 32-row chunks first
        rstride = 32;
        rows = cdiv(mtiles, rstride);
        hipLaunchKernelGGL(colreduce_inplace_kernel, dim3(cdiv(2 * C, 64), rows), dim3(256), 0, s, stats, mtiles,
                           (long)2 * C, rstride);
        HS_LAUNCH_CHECK("bn_finalize_stage1");
    }
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(256), 0, s, stats, rows, rstride, C, count, gamma, beta,
                       eps, momentum, running_mean, running_var, reinterpret_cast<long long*>(num_batches_tracked), mean,
                       invstd, scale, shift);
    HS_LAUNCH_CHECK("bn_finalize");
    return HIPSEG_OK;
}

extern "C" int hipseg_bn_eval_params(const float* gamma, const float* beta, const float* running_mean,
                                     const float* running_var, float eps, int C, float* mean, float* invstd,
                                     float* scale, float* shift, hipseg_stream_t stream) {
    HS_REQUIRE(gamma && beta && running_mean && running_var && mean && invstd && scale && shift && C > 0,
               "bn_eval_params: bad arguments");
    hipLaunchKernelGGL(bn_eval_params_kernel, dim3(cdiv(C, 256)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), gamma, beta, running_mean, running_var, eps, C, mean,
                       invstd, scale, shift);
    HS_LAUNCH_CHECK("bn_eval_params");
    return HIPSEG_OK;
}

// scale / shift of conv -> eval BatchNorm folded for hipseg_conv_affine_relu (one launch per layer)
__global__ void bn_fold_kernel(const float* gamma, const float* beta, const float* rm, const float* rv,
                               const float* conv_bias, float eps, int C, float* scale, float* shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) {
        const float sc = gamma[c] * (1.f / sqrtf(rv[c] + eps));
        const float sh = beta[c] - rm[c] * sc;  // exactly bn_eval_params' shift ...
        scale[c] = sc;
        shift[c] = conv_bias ? fmaf(conv_bias[c], sc, sh) : sh;  // ... + conv bias * scale
    }
}

extern "C" int hipseg_bn_fold(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                              const float* conv_bias, float eps, int C, float* scale, float* shift,
                              hipseg_stream_t stream) {
    HS_REQUIRE(gamma && beta && running_mean && running_var && scale && shift && C > 0, "bn_fold: bad arguments");
    hipLaunchKernelGGL(bn_fold_kernel, dim3(cdiv(C, 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), gamma, beta,
                       running_mean, running_var, conv_bias, eps, C, scale, shift);
    HS_LAUNCH_CHECK("bn_fold");
    return HIPSEG_OK;
}

extern "C" int hipseg_bn_relu_apply(int dtype, const void* x, const float* scale, const float* shift, void* y, int B,
                                    int H, int W, int C, int pool, hipseg_stream_t stream) {
    HS_REQUIRE(dtype == HIPSEG_F32 || dtype == HIPSEG_BF16, "bn_relu_apply: bad dtype");
    HS_REQUIRE(x && scale && shift && y && B > 0 && H > 0 && W > 0 && C > 0, "bn_relu_apply: bad arguments");
    HS_REQUIRE(!pool || (H % 2 == 0 && W % 2 == 0), "bn_relu_apply: pooling needs even H, W (got %d,%d)", H, W);
    const int V = vec_for(C, dtype);
    const long total = (long)B * (pool ? H / 2 : H) * (pool ? W / 2 : W) * (C / V);
    long g = (total + 255) / 256;
    if (!pool) g = (g + BN_APPLY_UNR - 1) / BN_APPLY_UNR;
    if (g > BN_APPLY_MAXBLK) g = BN_APPLY_MAXBLK;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    DISPATCH_TV(dtype, V, {
        if (pool)
            hipLaunchKernelGGL((bn_relu_apply_kernel<T_, V_, true>), dim3((unsigned)g), dim3(256), 0, s, (const T_*)x,
                               scale, shift, (T_*)y, B, H, W, C);
        else
            hipLaunchKernelGGL((bn_relu_apply_kernel<T_, V_, false>), dim3((unsigned)g), dim3(256), 0, s,
                               (const T_*)x, scale, shift, (T_*)y, B, H, W, C);
    });
    HS_LAUNCH_CHECK("bn_relu_apply");
    return HIPSEG_OK;
}

extern "C" int hipseg_bn_bwd_blocks(int B, int H, int W, int C, int dtype, int pool) {
    const long npix = (long)B * (pool ? H / 2 : H) * (pool ? W / 2 : W);
    return red_geo(npix, C, vec_for(C, dtype)).nblk;
}

static int check_red(const char* name, int C, int V) {
    HS_REQUIRE(C / V <= 256 && C <= 2048, "%s: unsupported channel count %d (vector width %d)", name, C, V);
    return HIPSEG_OK;
}

extern "C" int hipseg_bn_bwd_reduce(int dtype, const void* dy, const void* x, const float* mean, const float* invstd,
                                    const float* scale, const float* shift, float* partial, int B, int H, int W,
                                    int C, int pool, hipseg_stream_t stream) {
    return hipseg_bn_bwd_reduce2(dtype, dy, nullptr, x, mean, invstd, scale, shift, partial, B, H, W, C, pool, stream);
}

extern "C" int hipseg_bn_bwd_reduce2(int dtype, const void* dy, const void* dy2, const void* x, const float* mean,
                                     const float* invstd, const float* scale, const float* shift, float* partial, int B,
                                     int H, int W, int C, int pool, hipseg_stream_t stream) {
    HS_REQUIRE(dtype == HIPSEG_F32 || dtype == HIPSEG_BF16, "bn_bwd_reduce: bad dtype");
    HS_REQUIRE(dy && x && mean && invstd && scale && shift && partial && B > 0 && H > 0 && W > 0 && C > 0,
               "bn_bwd_reduce: bad arguments");
    HS_REQUIRE(!pool || (H % 2 == 0 && W % 2 == 0), "bn_bwd_reduce: pooling needs even H, W");
    const int V = vec_for(C, dtype);
    if (int rc = check_red("bn_bwd_reduce", C, V)) return rc;
    const long npix = (long)B * (pool ? H / 2 : H) * (pool ? W / 2 : W);
    const RedGeo g = red_geo(npix, C, V);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#define BN_RED_LAUNCH(POOL_, TWO_)                                                                                  \
    hipLaunchKernelGGL((bn_bwd_reduce_kernel<T_, V_, POOL_, TWO_>), dim3(g.nblk), dim3(256), 0, s, (const T_*)dy,     \
                       (const T_*)dy2, (const T_*)x, mean, invstd, scale, shift, partial, npix, H, W, C, g.CG, g.PL, \
                       g.ppb)
    DISPATCH_TV(dtype, V, {
        if (pool) {
            if (dy2) BN_RED_LAUNCH(true, true); else BN_RED_LAUNCH(true, false);
        } else {
            if (dy2) BN_RED_LAUNCH(false, true); else BN_RED_LAUNCH(false, false);
        }
    });
#undef BN_RED_LAUNCH
    HS_LAUNCH_CHECK("bn_bwd_reduce");
    return HIPSEG_OK;
}

extern "C" int hipseg_bn_bwd_apply(int dtype, const void* dy, const void* x, const float* mean, const float* invstd,
                                   const float* scale, const float* shift, const float* sums, double count, int eval,
                                   void* dx, float* dbias, int B, int H, int W, int C, int pool,
                                   hipseg_stream_t stream) {
    return hipseg_bn_bwd_apply2(dtype, dy, nullptr, x, mean, invstd, scale, shift, sums, count, eval, dx, dbias, B, H, W, C,
                                pool, stream);
}

extern "C" int hipseg_bn_bwd_apply2(int dtype, const void* dy, const void* dy2, const void* x, const float* mean,
                                    const float* invstd, const float* scale, const float* shift, const float* sums,
                                    double count, int eval, void* dx, float* dbias, int B, int H, int W, int C, int pool,
                                    hipseg_stream_t stream) {
    HS_REQUIRE(dtype == HIPSEG_F32 || dtype == HIPSEG_BF16, "bn_bwd_apply: bad dtype");
    HS_REQUIRE(dy && x && mean && invstd && scale && shift && dx && (eval || sums) && count > 0 && C > 0,
               "bn_bwd_apply: bad arguments");
    HS_REQUIRE(!pool || (H % 2 == 0 && W % 2 == 0), "bn_bwd_apply: pooling needs even H, W");
    const int V = vec_for(C, dtype);
    if (int rc = check_red("bn_bwd_apply", C, V)) return rc;
    const long npix = (long)B * (pool ? H / 2 : H) * (pool ? W / 2 : W);
    const RedGeo g = red_geo(npix, C, V);
    const float inv = (float)(1.0 / count);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#define BN_APP_LAUNCH(POOL_, TWO_)                                                                                 \
    hipLaunchKernelGGL((bn_bwd_apply_kernel<T_, V_, POOL_, TWO_>), dim3(g.nblk), dim3(256), 0, s, (const T_*)dy,     \
                       (const T_*)dy2, (const T_*)x, mean, invstd, scale, shift, sums, inv, eval, (T_*)dx, dbias,    \
                       npix, H, W, C, g.CG, g.PL, g.ppb)
    DISPATCH_TV(dtype, V, {
        if (pool) {
            if (dy2) BN_APP_LAUNCH(true, true); else BN_APP_LAUNCH(true, false);
        } else {
            if (dy2) BN_APP_LAUNCH(false, true); else BN_APP_LAUNCH(false, false);
        }
    });
#undef BN_APP_LAUNCH
    HS_LAUNCH_CHECK("bn_bwd_apply");
    return HIPSEG_OK;
}

extern "C" int hipseg_colsum_finalize(float* partial, int nblk, int rows, int C, float* out, float* zero_out,
                                      hipseg_stream_t stream) {
    HS_REQUIRE(partial && out && nblk > 0 && rows > 0 && C > 0, "colsum_finalize: bad arguments");
    const int RC = rows * C;
    hipLaunchKernelGGL(colsum_finalize_kernel, dim3(cdiv(RC, 4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       partial, nblk, RC, out, zero_out, C);
    HS_LAUNCH_CHECK("colsum_finalize");
    return HIPSEG_OK;
}

extern "C" int hipseg_colsum_blocks(long npix, int C, int dtype) { return red_geo(npix, C, vec_for(C, dtype)).nblk; }

extern "C" int hipseg_colsum(int dtype, const void* x, long npix, int C, float* partial, float* out,
                             hipseg_stream_t stream) {
    HS_REQUIRE(dtype == HIPSEG_F32 || dtype == HIPSEG_BF16, "colsum: bad dtype");
    HS_REQUIRE(x && partial && out && npix > 0 && C > 0, "colsum: bad arguments");
    const int V = vec_for(C, dtype);
    if (int rc = check_red("colsum", C, V)) return rc;
    const RedGeo g = red_geo(npix, C, V);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    DISPATCH_TV(dtype, V, {
        hipLaunchKernelGGL((colsum_kernel<T_, V_>), dim3(g.nblk), dim3(256), 0, s, (const T_*)x, partial, npix, C,
                           g.CG, g.PL, g.ppb);
    });
    HS_LAUNCH_CHECK("colsum");
    return hipseg_colsum_finalize(partial, g.nblk, 1, C, out, nullptr, stream);
}
