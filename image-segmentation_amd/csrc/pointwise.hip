// 1x1 stem / head, align-corners bilinear resize and layout helpers (all HBM-bound), NHWC.
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
inline int vec_for(int C, int dtype) {
    const int v = dtype == HIPSEG_BF16 ? 8 : 4;
    return (C % v == 0) ? v : 1;
}
inline unsigned grid_for(long total, int cap = 4096) {
    long g = (total + 255) / 256;
    return (unsigned)(g > cap ? cap : (g < 1 ? 1 : g));
}

constexpr int MAXCIN = 4;  // stem: RGB(A) images
#ifndef STEM_FWD_BLOCKS
#define STEM_FWD_BLOCKS 1024  // the per-thread weight prologue (40 loads) wants >= 16 items per thread behind it
#endif

// ------------------------------------------------------------------ stem: NCHW fp32 image -> NHWC
template <typename T, int V>
__global__ __launch_bounds__(256) void stem_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ b, T* __restrict__ y, int B, int Cin,
                                                        long HW, int Cout) {
    // index arithmetic in 32 bits (launch condition: B*HW*CG < 2^31): a 64-bit divide costs ~100 VALU instructions,
    // and three of them per 16-byte store made this kernel ALU-bound at a third of the HBM rate
    const int CG = Cout / V;
    const unsigned hw_n = (unsigned)HW;
    const unsigned total = (unsigned)B * hw_n * CG;
    const unsigned stride = gridDim.x * blockDim.x;
    const unsigned i0 = blockIdx.x * blockDim.x + threadIdx.x;
    if (Cin <= MAXCIN && stride % CG == 0) {
        // the channel group of a lane never changes along its grid-stride walk: keep its weights in registers
        // (loads are unconditional -- clamped index, then select -- so they all issue before the first wait)
        const int cg = (int)(i0 % CG);
        float wr[MAXCIN][V], br[V];
#pragma unroll
        for (int e = 0; e < V; ++e) {
            br[e] = b[cg * V + e];
#pragma unroll
            for (int ci = 0; ci < MAXCIN; ++ci) wr[ci][e] = w[(cg * V + e) * Cin + (ci < Cin ? ci : Cin - 1)];
        }
#pragma unroll
        for (int e = 0; e < V; ++e)
#pragma unroll
            for (int ci = 0; ci < MAXCIN; ++ci) wr[ci][e] = ci < Cin ? wr[ci][e] : 0.f;
        // A lane keeps its channel group and walks pixels p = i0 / CG + k * PS (PS = stride / CG); image index and
        // in-image offset advance incrementally.  (Round 4: as items (pixel, group) with `/ CG`, `/ HW` and a 64-bit
        // multiply-add per load, the loop ran 107 vector ALU instructions per 16-byte store -- 13 us of pure ALU issue at
        // this size, the kernel's 27 us were half that; now ~40.)
        constexpr int UNR = 4;  // pixels in flight per lane (the loop is latency-bound otherwise: 3 loads -> 1 store)
        const unsigned PS = stride / CG, npix = (unsigned)B * hw_n;
        const unsigned dn = PS / hw_n, dhw = PS - dn * hw_n;
        unsigned p = i0 / CG;
        unsigned n = p / hw_n, hw = p - n * hw_n;
        const size_t plane = hw_n;
        while (p < npix) {
            // unconditional loads (pixels past the end re-read the last pixel and skip their store)
            float xv[UNR][MAXCIN];
            bool ok[UNR];
            unsigned pu[UNR];
            const float* xp[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                ok[u] = p < npix;
                pu[u] = p;
                xp[u] = x + (size_t)(ok[u] ? n : (unsigned)B - 1) * Cin * plane + (ok[u] ? hw : hw_n - 1);
                p += PS;
                n += dn;
                hw += dhw;
                if (hw >= hw_n) {
                    hw -= hw_n;
                    n += 1;
                }
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u)
#pragma unroll
                for (int ci = 0; ci < MAXCIN; ++ci) xv[u][ci] = xp[u][(size_t)(ci < Cin ? ci : Cin - 1) * plane];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                float o[V];
#pragma unroll
                for (int e = 0; e < V; ++e) o[e] = br[e];
#pragma unroll
                for (int ci = 0; ci < MAXCIN; ++ci)
                    if (ci < Cin) {
#pragma unroll
                        for (int e = 0; e < V; ++e) o[e] = fmaf(wr[ci][e], xv[u][ci], o[e]);
                    }
                if (ok[u]) stv<T, V>(y + (size_t)pu[u] * Cout + cg * V, o);
            }
        }
        return;
    }
    for (unsigned i = i0; i < total; i += stride) {
        const int cg = (int)(i % CG);
        const unsigned p = i / CG;  // n*HW + pixel
        const unsigned n = p / hw_n, hw = p - n * hw_n;
        float o[V];
#pragma unroll
        for (int e = 0; e < V; ++e) o[e] = b[cg * V + e];
        for (int ci = 0; ci < Cin; ++ci) {
            const float xv = x[((size_t)n * Cin + ci) * hw_n + hw];
#pragma unroll
            for (int e = 0; e < V; ++e) o[e] = fmaf(w[(cg * V + e) * Cin + ci], xv, o[e]);
        }
        stv<T, V>(y + (size_t)p * Cout + cg * V, o);
    }
}

// partial[blk][Cin+1][Cout]: rows 0..Cin-1 = sum dy*x_ci, row Cin = sum dy
template <typename T, int V>
__global__ __launch_bounds__(256) void stem_bwd_kernel(const float* __restrict__ x, const T* __restrict__ dy,
                                                        const T* __restrict__ dy2, float* __restrict__ partial, int B,
                                                        int Cin, long HW, int Cout, int CG, int PL, long ppb) {
    __shared__ float red[2048];
    const int tid = threadIdx.x;
    const int cg = tid % CG, pl = tid / CG;
    float acc[MAXCIN + 1][V];
#pragma unroll
    for (int r = 0; r <= MAXCIN; ++r)
#pragma unroll
        for (int e = 0; e < V; ++e) acc[r][e] = 0.f;
    const long npix = (long)B * HW;
    if (pl < PL) {
        const long start = blockIdx.x * ppb;
        long end = start + ppb;
        if (end > npix) end = npix;
        // Every load of an iteration is issued UNCONDITIONALLY (tail pixels re-read the block's last pixel and are
        // selected to 0) before the first use: inside `if (p < end)` regions the compiler kept one pixel in flight per
        // lane (load, wait, add, next pixel) and the kernel ran at 2.4 TB/s of its 147 MB (round 3: 60 us).
        constexpr int UNR = 4;  // pixels in flight per lane
        const unsigned hw_n = (unsigned)HW;  // 32-bit divides (launch condition: B*HW < 2^31)
        for (long p0 = start + pl; p0 < end; p0 += (long)PL * UNR) {
            typename VecOf<T>::type gr[UNR], gr2[UNR];
            float xv[UNR][MAXCIN];
            bool ok[UNR];
            long pc[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const long p = p0 + (long)u * PL;
                ok[u] = p < end;
                pc[u] = ok[u] ? p : end - 1;
            }
            if constexpr (V > 1) {
#pragma unroll
                for (int u = 0; u < UNR; ++u) gr[u] = *reinterpret_cast<const typename VecOf<T>::type*>(dy + pc[u] * Cout + cg * V);
                if (dy2) {  // (uniform) second gradient of the same tensor (the stem output also feeds the last decoder block)
#pragma unroll
                    for (int u = 0; u < UNR; ++u)
                        gr2[u] = *reinterpret_cast<const typename VecOf<T>::type*>(dy2 + pc[u] * Cout + cg * V);
                }
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const unsigned pu = (unsigned)pc[u], n = pu / hw_n, hw = pu - n * hw_n;
                const float* xp = x + (size_t)n * Cin * hw_n + hw;  // (one 64-bit multiply-add per pixel, not per channel)
#pragma unroll
                for (int ci = 0; ci < MAXCIN; ++ci) xv[u][ci] = xp[(size_t)(ci < Cin ? ci : Cin - 1) * hw_n];
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                float g[V];
                if constexpr (V > 1) {
#pragma unroll
                    for (int e = 0; e < V; ++e) g[e] = (float)gr[u][e];
                    if (dy2) {
#pragma unroll
                        for (int e = 0; e < V; ++e) g[e] += (float)gr2[u][e];
                    }
                } else {
                    g[0] = (float)dy[pc[u] * Cout + cg];
                    if (dy2) g[0] += (float)dy2[pc[u] * Cout + cg];
                }
#pragma unroll
                for (int e = 0; e < V; ++e) g[e] = ok[u] ? g[e] : 0.f;
#pragma unroll
                for (int ci = 0; ci < MAXCIN; ++ci)
                    if (ci < Cin) {
#pragma unroll
                        for (int e = 0; e < V; ++e) acc[ci][e] = fmaf(g[e], xv[u][ci], acc[ci][e]);
                    }
#pragma unroll
                for (int e = 0; e < V; ++e) acc[MAXCIN][e] += g[e];
            }
        }
    }
    // block reduction over the pixel lanes, one accumulator row (input channel / bias) at a time: red[pl][Cout], then
    // COLUMN-parallel sums -- thread c adds column c over the PL rows in row order (conflict-free consecutive reads;
    // the former loop had the CG lanes of pl == 0 walk all rows alone: ~6 us of tail per block)
#pragma unroll
    for (int r = 0; r <= MAXCIN; ++r) {
        if (r < Cin || r == MAXCIN) {
            __syncthreads();
            if (pl < PL) {
#pragma unroll
                for (int e = 0; e < V; ++e) red[pl * Cout + cg * V + e] = acc[r][e];
            }
            __syncthreads();
            const int row = r == MAXCIN ? Cin : r;
            for (int c = tid; c < Cout; c += 256) {
                float t = 0.f;
                for (int j = 0; j < PL; ++j) t += red[j * Cout + c];
                partial[((size_t)blockIdx.x * (Cin + 1) + row) * Cout + c] = t;
            }
        }
    }
}

// dw[co][ci], db[co] from partial[blk][Cin+1][Cout]
// one 64-thread block per output element: lanes stride over the block partials, fixed-order wave sum
__global__ __launch_bounds__(64) void stem_bwd_finalize_kernel(const float* __restrict__ partial, int nblk, int Cin,
                                                                int Cout, float* __restrict__ dw, float* __restrict__ db) {
    const int i = blockIdx.x;
    const int row = i / Cout, co = i % Cout;
    float s = 0.f;
    for (int b = threadIdx.x; b < nblk; b += 64) s += partial[((size_t)b * (Cin + 1) + row) * Cout + co];
    s = wave_sum(s);
    if (threadIdx.x == 0) {
        if (row < Cin)
            dw[co * Cin + row] = s;
        else
            db[co] = s;
    }
}

// ------------------------------------------------------------------ head: NHWC -> NCHW fp32 logits
constexpr int MAXHC = 8;  // head output channels

// ONLOAD (round 4, hipseg_head_fwd_bnrelu): x is the PRE-normalisation output of the last ConvBlock's second convolution
// and the head runs over round_T(relu(x * scale[c] + shift[c])) -- that block's final BatchNorm + ReLU applied in the
// head's load path, so the activated tensor and the bn_relu_apply pass that wrote it never exist (the value is rounded
// to T exactly as that pass would have stored it: results are bit-identical to the two-kernel form).
// CV > 0: Cin = CV * V is known at compile time -- the lane then keeps UNR pixels (CV 16-byte vectors each, still packed)
// in flight; with one pixel per grid-stride step the four loads of a 32-channel pixel were all a lane had outstanding
// (3.8 TB/s).  HC bounds Cout as in head_bwd_kernel.
template <typename T, int V, bool ONLOAD = false, int CV = 0, int HC = MAXHC>
__global__ __launch_bounds__(256) void head_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ b, float* __restrict__ logits, int B,
                                                        long HW, int Cin, int Cout, const float* __restrict__ scale = nullptr,
                                                        const float* __restrict__ shift = nullptr) {
    const long npix = (long)B * HW;
    const unsigned hw_n = (unsigned)HW;
    if constexpr (CV > 0) {
        typedef typename VecOf<T>::type vec_t;
        static_assert(VecOf<T>::N == V, "vector path");
        constexpr int UNR = 4;
        const long stride = (long)gridDim.x * blockDim.x;
        for (long p0 = blockIdx.x * (long)blockDim.x + threadIdx.x; p0 < npix; p0 += stride * UNR) {
            vec_t raw[UNR][CV];
            bool ok[UNR];
            long pc[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const long p = p0 + u * stride;
                ok[u] = p < npix;
                pc[u] = ok[u] ? p : npix - 1;  // (tail: re-read the last pixel, skip the store)
#pragma unroll
                for (int c = 0; c < CV; ++c) raw[u][c] = *reinterpret_cast<const vec_t*>(x + pc[u] * (CV * V) + c * V);
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                float o[HC];
#pragma unroll
                for (int co = 0; co < HC; ++co) o[co] = b[co < Cout ? co : Cout - 1];
#pragma unroll
                for (int co = 0; co < HC; ++co) o[co] = co < Cout ? o[co] : 0.f;
#pragma unroll
                for (int c = 0; c < CV; ++c) {
                    float v[V];
#pragma unroll
                    for (int e = 0; e < V; ++e) v[e] = to_f32(raw[u][c][e]);
                    if constexpr (ONLOAD) {
#pragma unroll
                        for (int e = 0; e < V; ++e)
                            v[e] = to_f32(from_f32<T>(fmaxf(0.f, v[e] * scale[c * V + e] + shift[c * V + e])));
                    }
#pragma unroll
                    for (int co = 0; co < HC; ++co)
                        if (co < Cout) {
#pragma unroll
                            for (int e = 0; e < V; ++e) o[co] = fmaf(w[co * (CV * V) + c * V + e], v[e], o[co]);
                        }
                }
                if (ok[u]) {
                    const unsigned n = (unsigned)pc[u] / hw_n, hw = (unsigned)pc[u] - n * hw_n;  // 32-bit divide
#pragma unroll
                    for (int co = 0; co < HC; ++co)
                        if (co < Cout) logits[((size_t)n * Cout + co) * hw_n + hw] = o[co];
                }
            }
        }
        return;
    }
    for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < npix; p += (long)gridDim.x * blockDim.x) {
        float o[MAXHC];
#pragma unroll
        for (int co = 0; co < MAXHC; ++co) o[co] = b[co < Cout ? co : Cout - 1];
#pragma unroll
        for (int co = 0; co < MAXHC; ++co) o[co] = co < Cout ? o[co] : 0.f;
        for (int c = 0; c < Cin; c += V) {
            float v[V];
            ldv<T, V>(x + p * Cin + c, v);
            if constexpr (ONLOAD) {
#pragma unroll
                for (int e = 0; e < V; ++e) v[e] = to_f32(from_f32<T>(fmaxf(0.f, v[e] * scale[c + e] + shift[c + e])));
            }
#pragma unroll
            for (int co = 0; co < MAXHC; ++co)
                if (co < Cout) {
#pragma unroll
                    for (int e = 0; e < V; ++e) o[co] = fmaf(w[co * Cin + c + e], v[e], o[co]);
                }
        }
        const unsigned n = (unsigned)p / hw_n, hw = (unsigned)p - n * hw_n;  // 32-bit divide
#pragma unroll
        for (int co = 0; co < MAXHC; ++co)
            if (co < Cout) logits[((size_t)n * Cout + co) * hw_n + hw] = o[co];
    }
}

// dx[p][c] = sum_co dl[co][p] w[co][c];  partial[blk][Cout][Cin+1]: dW rows, last column = db
// ONLOAD (hipseg_head_bwd_bnrelu): x is the pre-normalisation tensor (see head_fwd_kernel) -- dW is taken against
// round_T(relu(x * scale + shift)) -- and, with x and dx both in registers, the block also reduces the BatchNorm-backward
// sums of that layer, bnp[blk][2][Cin] = [sum g | sum g * xhat] with g = round_T(dx) where x * scale + shift > 0
// (PixelCtx::finish's rule; rows in bn_bwd_reduce_kernel's layout): the bn_bwd_reduce launch that would re-read dx and x
// is not needed (hipseg_convblock_t::dout_rows).
// HC: compile-time bound of the head's output channels (4 or 8): the per-lane dW accumulators and weights are HC x V
// registers each, and the U-Nets' 3-class head at HC = 8 would carry 128 registers of zeros.
template <typename T, int V, bool ONLOAD = false, int HC = MAXHC>
__global__ __launch_bounds__(256) void head_bwd_kernel(const T* __restrict__ x, const float* __restrict__ dl,
                                                        const float* __restrict__ w, T* __restrict__ dx,
                                                        float* __restrict__ partial, int B, long HW, int Cin, int Cout,
                                                        int CG, int PL, long ppb, const float* __restrict__ mean = nullptr,
                                                        const float* __restrict__ invstd = nullptr,
                                                        const float* __restrict__ scale = nullptr,
                                                        const float* __restrict__ shift = nullptr,
                                                        float* __restrict__ bnp = nullptr) {
    __shared__ float red[2048 + 256];  // PL rows of (Cin + 1): PL * Cin <= 2048, PL <= 256
    const int tid = threadIdx.x;
    const int cg = tid % CG, pl = tid / CG;
    float aw[HC][V], ab[HC];
#pragma unroll
    for (int co = 0; co < HC; ++co) {
        ab[co] = 0.f;
#pragma unroll
        for (int e = 0; e < V; ++e) aw[co][e] = 0.f;
    }
    const long npix = (long)B * HW;
    float s1[ONLOAD ? V : 1], s2[ONLOAD ? V : 1];
#pragma unroll
    for (int e = 0; e < (ONLOAD ? V : 1); ++e) s1[e] = s2[e] = 0.f;
    if (pl < PL) {
        float mn[ONLOAD ? V : 1], is[ONLOAD ? V : 1], sc[ONLOAD ? V : 1], sh[ONLOAD ? V : 1];
        if constexpr (ONLOAD) {
#pragma unroll
            for (int e = 0; e < V; ++e) {
                mn[e] = mean[cg * V + e];
                is[e] = invstd[cg * V + e];
                sc[e] = scale[cg * V + e];
                sh[e] = shift[cg * V + e];
            }
        }
        float wv[HC][V];
#pragma unroll
        for (int co = 0; co < HC; ++co)
#pragma unroll
            for (int e = 0; e < V; ++e) wv[co][e] = w[(co < Cout ? co : Cout - 1) * Cin + cg * V + e];
#pragma unroll
        for (int co = 0; co < HC; ++co)
#pragma unroll
            for (int e = 0; e < V; ++e) wv[co][e] = co < Cout ? wv[co][e] : 0.f;
        const long start = blockIdx.x * ppb;
        long end = start + ppb;
        if (end > npix) end = npix;
        // all loads of an iteration are issued unconditionally (tail pixels re-read the block's last pixel, their
        // contributions are selected to 0 and their store is skipped) before the first use -- see stem_bwd_kernel
        constexpr int UNR = 4;  // pixels in flight per lane
        const unsigned hw_n = (unsigned)HW;  // 32-bit divides (launch condition: B*HW < 2^31)
        for (long p0 = start + pl; p0 < end; p0 += (long)PL * UNR) {
            float xv[UNR][V], g[UNR][HC];
            bool ok[UNR];
            long pc[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const long p = p0 + (long)u * PL;
                ok[u] = p < end;
                pc[u] = ok[u] ? p : end - 1;
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) ldv<T, V>(x + pc[u] * Cin + cg * V, xv[u]);
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const unsigned pu = (unsigned)pc[u], n = pu / hw_n, hw = pu - n * hw_n;
                const float* dlp = dl + (size_t)n * Cout * hw_n + hw;  // (one 64-bit multiply-add per pixel, not per class)
#pragma unroll
                for (int co = 0; co < HC; ++co) g[u][co] = dlp[(size_t)(co < Cout ? co : Cout - 1) * hw_n];
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
#pragma unroll
                for (int co = 0; co < HC; ++co) g[u][co] = (ok[u] && co < Cout) ? g[u][co] : 0.f;
                float o[V];
#pragma unroll
                for (int e = 0; e < V; ++e) o[e] = 0.f;
                float xh[ONLOAD ? V : 1];
                bool pos[ONLOAD ? V : 1];
                if constexpr (ONLOAD) {
#pragma unroll
                    for (int e = 0; e < V; ++e) {
                        const float raw = xv[u][e], y = raw * sc[e] + sh[e];
                        xh[e] = (raw - mn[e]) * is[e];
                        pos[e] = y > 0.f;
                        xv[u][e] = to_f32(from_f32<T>(fmaxf(0.f, y)));
                    }
                }
#pragma unroll
                for (int co = 0; co < HC; ++co)
                    if (co < Cout) {
                        ab[co] += g[u][co];
#pragma unroll
                        for (int e = 0; e < V; ++e) {
                            o[e] = fmaf(g[u][co], wv[co][e], o[e]);
                            aw[co][e] = fmaf(g[u][co], xv[u][e], aw[co][e]);
                        }
                    }
                if constexpr (ONLOAD && V > 1) {
                    // round once: the packed vector is what is stored AND what the BatchNorm sums are formed from
                    typename VecOf<T>::type ov;
#pragma unroll
                    for (int e = 0; e < V; ++e) ov[e] = from_f32<T>(o[e]);
                    if (ok[u]) *reinterpret_cast<typename VecOf<T>::type*>(dx + pc[u] * Cin + cg * V) = ov;
#pragma unroll
                    for (int e = 0; e < V; ++e) {
                        const float gq = (ok[u] && pos[e]) ? to_f32(ov[e]) : 0.f;  // dx as stored
                        s1[e] += gq;
                        s2[e] += gq * xh[e];
                    }
                } else {
                    if (ok[u]) stv<T, V>(dx + pc[u] * Cin + cg * V, o);
                    if constexpr (ONLOAD) {
#pragma unroll
                        for (int e = 0; e < V; ++e) {
                            const float gq = (ok[u] && pos[e]) ? to_f32(from_f32<T>(o[e])) : 0.f;  // dx as stored
                            s1[e] += gq;
                            s2[e] += gq * xh[e];
                        }
                    }
                }
            }
        }
    }
    if constexpr (ONLOAD) {
        // BatchNorm-backward rows: red[pl][Cin] per sum (PL * Cin <= 2048), column-parallel sums in row order
#pragma unroll
        for (int arr = 0; arr < 2; ++arr) {
            __syncthreads();
            if (pl < PL) {
#pragma unroll
                for (int e = 0; e < V; ++e) red[pl * Cin + cg * V + e] = arr ? s2[e] : s1[e];
            }
            __syncthreads();
            for (int c = tid; c < Cin; c += 256) {
                float t = 0.f;
                for (int j = 0; j < PL; ++j) t += red[j * Cin + c];
                bnp[((size_t)blockIdx.x * 2 + arr) * Cin + c] = t;
            }
        }
    }
    // block reduction over the pixel lanes, one output channel at a time: red[pl][Cin + 1] (last column = bias sum), then
    // column-parallel sums in row order (see stem_bwd_kernel)
#pragma unroll
    for (int co = 0; co < HC; ++co) {
        if (co < Cout) {
            __syncthreads();
            if (pl < PL) {
#pragma unroll
                for (int e = 0; e < V; ++e) red[pl * (Cin + 1) + cg * V + e] = aw[co][e];
                if (cg == 0) red[pl * (Cin + 1) + Cin] = ab[co];
            }
            __syncthreads();
            float* dst = partial + ((size_t)blockIdx.x * Cout + co) * (Cin + 1);
            for (int c = tid; c <= Cin; c += 256) {
                float t = 0.f;
                for (int j = 0; j < PL; ++j) t += red[j * (Cin + 1) + c];
                dst[c] = t;
            }
        }
    }
}

__global__ __launch_bounds__(64) void head_bwd_finalize_kernel(const float* __restrict__ partial, int nblk, int Cin,
                                                                int Cout, float* __restrict__ dw, float* __restrict__ db) {
    const int i = blockIdx.x;
    const int co = i / (Cin + 1), c = i % (Cin + 1);
    float s = 0.f;
    for (int b = threadIdx.x; b < nblk; b += 64) s += partial[(size_t)b * Cout * (Cin + 1) + i];
    s = wave_sum(s);
    if (threadIdx.x == 0) {
        if (c < Cin)
            dw[co * Cin + c] = s;
        else
            db[co] = s;
    }
}

// ------------------------------------------------------------------ bilinear, align_corners=True
// index/lambda exactly as ATen's area_pixel_compute_source_index + guard_index_and_lambda (fp32).
__device__ __forceinline__ void src_index(float scale, int dst, int in_size, int& i0, int& i1, float& l0, float& l1) {
    const float real = scale * (float)dst;
    i0 = (int)real;
    if (i0 > in_size - 1) i0 = in_size - 1;
    float lam = real - (float)i0;
    lam = fminf(fmaxf(lam, 0.f), 1.f);
    i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    l1 = lam;
    l0 = 1.f - lam;
}
__host__ __device__ __forceinline__ float ac_scale(int in_size, int out_size) {
    return out_size > 1 ? (float)(in_size - 1) / (float)(out_size - 1) : 0.f;
}

template <typename T, int V>
__global__ __launch_bounds__(256) void bilinear_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int Hi,
                                                            int Wi, int Ho, int Wo, int C, float sy, float sx) {
    const int CG = C / V;
    const long total = (long)B * Ho * Wo * CG;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int cg = (int)(i % CG);
        const long op = i / CG;
        const int ox = (int)(op % Wo), oy = (int)((op / Wo) % Ho);
        const long n = op / ((long)Wo * Ho);
        int y0, y1, x0, x1;
        float wy0, wy1, wx0, wx1;
        src_index(sy, oy, Hi, y0, y1, wy0, wy1);
        src_index(sx, ox, Wi, x0, x1, wx0, wx1);
        float a[V], b[V], c[V], d[V], o[V];
        ldv<T, V>(x + ((n * Hi + y0) * Wi + x0) * C + cg * V, a);
        ldv<T, V>(x + ((n * Hi + y0) * Wi + x1) * C + cg * V, b);
        ldv<T, V>(x + ((n * Hi + y1) * Wi + x0) * C + cg * V, c);
        ldv<T, V>(x + ((n * Hi + y1) * Wi + x1) * C + cg * V, d);
#pragma unroll
        for (int e = 0; e < V; ++e) o[e] = wy0 * (wx0 * a[e] + wx1 * b[e]) + wy1 * (wx0 * c[e] + wx1 * d[e]);
        stv<T, V>(y + op * C + cg * V, o);
    }
}

// gather form of the adjoint: each INPUT pixel sums the output pixels that sampled it (deterministic)
template <typename T, int V>
__global__ __launch_bounds__(256) void bilinear_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int B, int Hi,
                                                            int Wi, int Ho, int Wo, int C, float sy, float sx) {
    const int CG = C / V;
    const long total = (long)B * Hi * Wi * CG;
    const bool small = total < (1l << 31);  // 32-bit index arithmetic where it fits (a 64-bit divide is ~100 VALU instructions)
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int cg, ix, iy;
        long ip, n;
        if (small) {
            const unsigned iu = (unsigned)i, ipu = iu / (unsigned)CG, rowu = ipu / (unsigned)Wi;
            cg = (int)(iu - ipu * (unsigned)CG);
            ix = (int)(ipu - rowu * (unsigned)Wi);
            const unsigned nu = rowu / (unsigned)Hi;
            iy = (int)(rowu - nu * (unsigned)Hi);
            ip = ipu;
            n = nu;
        } else {
            cg = (int)(i % CG);
            ip = i / CG;
            ix = (int)(ip % Wi);
            iy = (int)((ip / Wi) % Hi);
            n = ip / ((long)Wi * Hi);
        }
        int jy0 = 0, jy1 = Ho - 1, jx0 = 0, jx1 = Wo - 1;
        // output rows / columns that can sample input row iy: floor(j * s) in {iy - 1, iy}, i.e. j in ((iy - 1) / s,
        // (iy + 1) / s); the candidates are re-checked exactly below, the window only has to CONTAIN them (a small
        // epsilon covers the rounding of the division).  A window two candidates wider on each side made this gather
        // ALU-bound: 25 index computations per input pixel instead of 9 when down-scaling by two.
        if (sy > 0.f) {
            jy0 = max(0, (int)floorf((float)(iy - 1) / sy - 1e-3f));
            jy1 = min(Ho - 1, (int)ceilf((float)(iy + 1) / sy + 1e-3f));
        }
        if (sx > 0.f) {
            jx0 = max(0, (int)floorf((float)(ix - 1) / sx - 1e-3f));
            jx1 = min(Wo - 1, (int)ceilf((float)(ix + 1) / sx + 1e-3f));
        }
        float acc[V];
#pragma unroll
        for (int e = 0; e < V; ++e) acc[e] = 0.f;
        for (int jy = jy0; jy <= jy1; ++jy) {
            int a0, a1;
            float l0, l1;
            src_index(sy, jy, Hi, a0, a1, l0, l1);
            const float wy = (a0 == iy ? l0 : 0.f) + (a1 == iy ? l1 : 0.f);
            if (wy == 0.f) continue;
            for (int jx = jx0; jx <= jx1; ++jx) {
                int b0, b1;
                float m0, m1;
                src_index(sx, jx, Wi, b0, b1, m0, m1);
                const float wx = (b0 == ix ? m0 : 0.f) + (b1 == ix ? m1 : 0.f);
                if (wx == 0.f) continue;
                float g[V];
                ldv<T, V>(dy + ((n * Ho + jy) * Wo + jx) * C + cg * V, g);
#pragma unroll
                for (int e = 0; e < V; ++e) acc[e] = fmaf(wy * wx, g[e], acc[e]);
            }
        }
        stv<T, V>(dx + ip * C + cg * V, acc);
    }
}

// ------------------------------------------------------------------ layout helpers
template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ x, T* __restrict__ y, int B, int C, long HW) {
    const long total = (long)B * HW * C;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const long p = i / C;
        const long n = p / HW, hw = p % HW;
        y[i] = (T)x[(n * C + c) * HW + hw];
    }
}
template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* __restrict__ x, float* __restrict__ y, int B, int C, long HW) {
    const long total = (long)B * HW * C;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long hw = i % HW;
        const int c = (int)((i / HW) % C);
        const long n = i / (HW * C);
        y[i] = (float)x[(n * HW + hw) * C + c];
    }
}

struct RedGeo {
    int CG, PL, nblk;
    long ppb;
};
inline RedGeo red_geo(long npix, int C, int V) {
    RedGeo g;
    g.CG = C / V;
    g.PL = 256 / g.CG;
    long nb = (npix + (long)g.PL * 8 - 1) / ((long)g.PL * 8);
    if (nb > 1024) nb = 1024;
    if (nb < 1) nb = 1;
    g.nblk = (int)nb;
    g.ppb = (npix + nb - 1) / nb;
    return g;
}

#define DISPATCH_TV(dtype, V, ...)  \
    do {                            \
        if (dtype == HIPSEG_BF16) { \
            typedef bf16 T_;        \
            if (V == 8) {           \
                constexpr int V_ = 8; \
                __VA_ARGS__         \
            } else {                \
                constexpr int V_ = 1; \
                __VA_ARGS__         \
            }                       \
        } else {                    \
            typedef float T_;       \
            if (V == 4) {           \
                constexpr int V_ = 4; \
                __VA_ARGS__         \
            } else {                \
                constexpr int V_ = 1; \
                __VA_ARGS__         \
            }                       \
        }                           \
    } while (0)

}  // namespace

extern "C" int hipseg_stem_fwd(int dtype, const float* x, const float* w, const float* b, void* y, int B, int Cin,
                               int H, int W, int Cout, hipseg_stream_t stream) {
    HS_REQUIRE(dtype == HIPSEG_F32 || dtype == HIPSEG_BF16, "stem_fwd: bad dtype");
    HS_REQUIRE(x && w && b && y && B > 0 && Cin > 0 && H > 0 && W > 0 && Cout > 0, "stem_fwd: bad arguments");
    const int V = vec_for(Cout, dtype);
    const long HW = (long)H * W;
    HS_REQUIRE((long)B * HW * (Cout / V) < (1l << 31), "stem_fwd: more than 2^31 (pixel, channel-vector) items");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    DISPATCH_TV(dtype, V, {
        hipLaunchKernelGGL((stem_fwd_kernel<T_, V_>), dim3(grid_for(B * HW * (Cout / V_), STEM_FWD_BLOCKS)), dim3(256), 0, s, x, w, b,
                           (T_*)y, B, Cin, HW, Cout);
    });
    HS_LAUNCH_CHECK("stem_fwd");
    return HIPSEG_OK;
}

extern "C" int hipseg_stem_bwd_blocks(int B, int H, int W) {
    long nb = ((long)B * H * W + 255) / 256;
    return (int)(nb > 1024 ? 1024 : (nb < 1 ? 1 : nb));
}

extern "C" int hipseg_stem_bwd(int dtype, const float* x, const void* dy, float* partial, float* dw, float* db, int B,
                               int Cin, int H, int W, int Cout, hipseg_stream_t stream) {
    return hipseg_stem_bwd2(dtype, x, dy, nullptr, partial, dw, db, B, Cin, H, W, Cout, stream);
}

extern "C" int hipseg_stem_bwd2(int dtype, const float* x, const void* dy, const void* dy2, float* partial, float* dw,
                                float* db, int B, int Cin, int H, int W, int Cout, hipseg_stream_t stream) {
    HS_REQUIRE(dtype == HIPSEG_F32 || dtype == HIPSEG_BF16, "stem_bwd: bad dtype");
    HS_REQUIRE(x && dy && partial && dw && db && B > 0 && H > 0 && W > 0, "stem_bwd: bad arguments");
    HS_REQUIRE(Cin >= 1 && Cin <= MAXCIN, "stem_bwd: in_channels %d unsupported (1..%d)", Cin, MAXCIN);
    const int V = vec_for(Cout, dtype);
    HS_REQUIRE(Cout / V <= 256 && Cout <= 2048, "stem_bwd: unsupported Cout %d", Cout);
    const long HW = (long)H * W, npix = (long)B * HW;
    HS_REQUIRE(npix < (1l << 31), "stem_bwd: more than 2^31 pixels");
    RedGeo g;
    g.CG = Cout / V;
    g.PL = 256 / g.CG;
    g.nblk = hipseg_stem_bwd_blocks(B, H, W);
    g.ppb = (npix + g.nblk - 1) / g.nblk;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    DISPATCH_TV(dtype, V, {
        hipLaunchKernelGGL((stem_bwd_kernel<T_, V_>), dim3(g.nblk), dim3(256), 0, s, x, (const T_*)dy, (const T_*)dy2, partial,
                           B, Cin, HW, Cout, g.CG, g.PL, g.ppb);
    });
    HS_LAUNCH_CHECK("stem_bwd");
    hipLaunchKernelGGL(stem_bwd_finalize_kernel, dim3((Cin + 1) * Cout), dim3(64), 0, s, partial, g.nblk, Cin, Cout, dw,
                       db);
    HS_LAUNCH_CHECK("stem_bwd_finalize");
    return HIPSEG_OK;
}

extern "C" int hipseg_head_fwd(int dtype, const void* x, const float* w, const float* b, float* logits, int B, int H,
                               int W, int Cin, int Cout, hipseg_stream_t stream) {
    HS_REQUIRE(dtype == HIPSEG_F32 || dtype == HIPSEG_BF16, "head_fwd: bad dtype");
    HS_REQUIRE(x && w && b && logits && B > 0 && H > 0 && W > 0 && Cin > 0, "head_fwd: bad arguments");
    HS_REQUIRE(Cout >= 1 && Cout <= MAXHC, "head_fwd: out_channels %d unsupported (1..%d)", Cout, MAXHC);
    const int V = vec_for(Cin, dtype);
    const long HW = (long)H * W;
    HS_REQUIRE((long)B * HW < (1l << 31), "head_fwd: more than 2^31 pixels");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    DISPATCH_TV(dtype, V, {
        if constexpr (V_ > 1) {
            if (Cin == 4 * V_ && Cout <= 4) {  // the U-Nets' head on bf16 activations (32 -> 3): pixels in flight per lane
                hipLaunchKernelGGL((head_fwd_kernel<T_, V_, false, 4, 4>), dim3(grid_for((B * HW + 3) / 4)), dim3(256), 0, s,
                                   (const T_*)x, w, b, logits, B, HW, Cin, Cout, nullptr, nullptr);
                HS_LAUNCH_CHECK("head_fwd");
                return HIPSEG_OK;
            }
        }
        hipLaunchKernelGGL((head_fwd_kernel<T_, V_>), dim3(grid_for(B * HW)), dim3(256), 0, s, (const T_*)x, w, b,
                           logits, B, HW, Cin, Cout);
    });
    HS_LAUNCH_CHECK("head_fwd");
    return HIPSEG_OK;
}

extern "C" int hipseg_head_fwd_bnrelu(int dtype, const void* raw, const float* scale, const float* shift, const float* w,
                                      const float* b, float* logits, int B, int H, int W, int Cin, int Cout,
                                      hipseg_stream_t stream) {
    HS_REQUIRE(dtype == HIPSEG_F32 || dtype == HIPSEG_BF16, "head_fwd_bnrelu: bad dtype");
    HS_REQUIRE(raw && scale && shift && w && b && logits && B > 0 && H > 0 && W > 0 && Cin > 0, "head_fwd_bnrelu: bad arguments");
    HS_REQUIRE(Cout >= 1 && Cout <= MAXHC, "head_fwd_bnrelu: out_channels %d unsupported (1..%d)", Cout, MAXHC);
    const int V = vec_for(Cin, dtype);
    const long HW = (long)H * W;
    HS_REQUIRE((long)B * HW < (1l << 31), "head_fwd_bnrelu: more than 2^31 pixels");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    DISPATCH_TV(dtype, V, {
        if constexpr (V_ > 1) {
            if (Cin == 4 * V_ && Cout <= 4) {
                hipLaunchKernelGGL((head_fwd_kernel<T_, V_, true, 4, 4>), dim3(grid_for((B * HW + 3) / 4)), dim3(256), 0, s,
                                   (const T_*)raw, w, b, logits, B, HW, Cin, Cout, scale, shift);
                HS_LAUNCH_CHECK("head_fwd_bnrelu");
                return HIPSEG_OK;
            }
        }
        hipLaunchKernelGGL((head_fwd_kernel<T_, V_, true>), dim3(grid_for(B * HW)), dim3(256), 0, s, (const T_*)raw, w, b,
                           logits, B, HW, Cin, Cout, scale, shift);
    });
    HS_LAUNCH_CHECK("head_fwd_bnrelu");
    return HIPSEG_OK;
}

extern "C" int hipseg_head_bwd_blocks(int B, int H, int W) {
    long nb = ((long)B * H * W + 255) / 256;
    return (int)(nb > 512 ? 512 : (nb < 1 ? 1 : nb));
}

extern "C" int hipseg_head_bwd(int dtype, const void* x, const float* dlogits, const float* w, void* dx,
                               float* partial, float* dw, float* db, int B, int H, int W, int Cin, int Cout,
                               hipseg_stream_t stream) {
    HS_REQUIRE(dtype == HIPSEG_F32 || dtype == HIPSEG_BF16, "head_bwd: bad dtype");
    HS_REQUIRE(x && dlogits && w && dx && partial && dw && db && B > 0 && H > 0 && W > 0, "head_bwd: bad arguments");
    HS_REQUIRE(Cout >= 1 && Cout <= MAXHC, "head_bwd: out_channels %d unsupported (1..%d)", Cout, MAXHC);
    const int V = vec_for(Cin, dtype);
    HS_REQUIRE(Cin / V <= 256 && Cin + 1 <= 2048, "head_bwd: unsupported Cin %d", Cin);
    const long HW = (long)H * W, npix = (long)B * HW;
    HS_REQUIRE(npix < (1l << 31), "head_bwd: more than 2^31 pixels");
    RedGeo g;
    g.CG = Cin / V;
    g.PL = 256 / g.CG;
    g.nblk = hipseg_head_bwd_blocks(B, H, W);
    g.ppb = (npix + g.nblk - 1) / g.nblk;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    DISPATCH_TV(dtype, V, {
        if (Cout <= 4)
            hipLaunchKernelGGL((head_bwd_kernel<T_, V_, false, 4>), dim3(g.nblk), dim3(256), 0, s, (const T_*)x, dlogits, w,
                               (T_*)dx, partial, B, HW, Cin, Cout, g.CG, g.PL, g.ppb);
        else
            hipLaunchKernelGGL((head_bwd_kernel<T_, V_, false, MAXHC>), dim3(g.nblk), dim3(256), 0, s, (const T_*)x, dlogits,
                               w, (T_*)dx, partial, B, HW, Cin, Cout, g.CG, g.PL, g.ppb);
    });
    HS_LAUNCH_CHECK("head_bwd");
    hipLaunchKernelGGL(head_bwd_finalize_kernel, dim3(Cout * (Cin + 1)), dim3(64), 0, s, partial, g.nblk, Cin, Cout, dw,
                       db);
    HS_LAUNCH_CHECK("head_bwd_finalize");
    return HIPSEG_OK;
}

extern "C" int hipseg_head_bwd_bnrelu(int dtype, const void* raw, const float* mean, const float* invstd, const float* scale,
                                      const float* shift, const float* dlogits, const float* w, void* dx, float* partial,
                                      float* dw, float* db, float* bn_partial, int B, int H, int W, int Cin, int Cout,
                                      hipseg_stream_t stream) {
    HS_REQUIRE(dtype == HIPSEG_F32 || dtype == HIPSEG_BF16, "head_bwd_bnrelu: bad dtype");
    HS_REQUIRE(raw && mean && invstd && scale && shift && dlogits && w && dx && partial && dw && db && bn_partial && B > 0 &&
                   H > 0 && W > 0,
               "head_bwd_bnrelu: bad arguments");
    HS_REQUIRE(Cout >= 1 && Cout <= MAXHC, "head_bwd_bnrelu: out_channels %d unsupported (1..%d)", Cout, MAXHC);
    const int V = vec_for(Cin, dtype);
    HS_REQUIRE(Cin / V <= 256 && Cin + 1 <= 2048, "head_bwd_bnrelu: unsupported Cin %d", Cin);
    const long HW = (long)H * W, npix = (long)B * HW;
    HS_REQUIRE(npix < (1l << 31), "head_bwd_bnrelu: more than 2^31 pixels");
    RedGeo g;
    g.CG = Cin / V;
    g.PL = 256 / g.CG;
    g.nblk = hipseg_head_bwd_blocks(B, H, W);
    g.ppb = (npix + g.nblk - 1) / g.nblk;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    DISPATCH_TV(dtype, V, {
        if (Cout <= 4)
            hipLaunchKernelGGL((head_bwd_kernel<T_, V_, true, 4>), dim3(g.nblk), dim3(256), 0, s, (const T_*)raw, dlogits, w,
                               (T_*)dx, partial, B, HW, Cin, Cout, g.CG, g.PL, g.ppb, mean, invstd, scale, shift, bn_partial);
        else
            hipLaunchKernelGGL((head_bwd_kernel<T_, V_, true, MAXHC>), dim3(g.nblk), dim3(256), 0, s, (const T_*)raw, dlogits,
                               w, (T_*)dx, partial, B, HW, Cin, Cout, g.CG, g.PL, g.ppb, mean, invstd, scale, shift,
                               bn_partial);
    });
    HS_LAUNCH_CHECK("head_bwd_bnrelu");
    hipLaunchKernelGGL(head_bwd_finalize_kernel, dim3(Cout * (Cin + 1)), dim3(64), 0, s, partial, g.nblk, Cin, Cout, dw,
                       db);
    HS_LAUNCH_CHECK("head_bwd_finalize");
    return HIPSEG_OK;
}

extern "C" int hipseg_bilinear_fwd(int dtype, const void* x, void* y, int B, int Hi, int Wi, int Ho, int Wo, int C,
                                   hipseg_stream_t stream) {
    HS_REQUIRE(dtype == HIPSEG_F32 || dtype == HIPSEG_BF16, "bilinear_fwd: bad dtype");
    HS_REQUIRE(x && y && B > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && C > 0, "bilinear_fwd: bad arguments");
    const int V = vec_for(C, dtype);
    const float sy = ac_scale(Hi, Ho), sx = ac_scale(Wi, Wo);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    DISPATCH_TV(dtype, V, {
        hipLaunchKernelGGL((bilinear_fwd_kernel<T_, V_>), dim3(grid_for((long)B * Ho * Wo * (C / V_))), dim3(256), 0,
                           s, (const T_*)x, (T_*)y, B, Hi, Wi, Ho, Wo, C, sy, sx);
    });
    HS_LAUNCH_CHECK("bilinear_fwd");
    return HIPSEG_OK;
}

extern "C" int hipseg_bilinear_bwd(int dtype, const void* dy, void* dx, int B, int Hi, int Wi, int Ho, int Wo, int C,
                                   hipseg_stream_t stream) {
    HS_REQUIRE(dtype == HIPSEG_F32 || dtype == HIPSEG_BF16, "bilinear_bwd: bad dtype");
    HS_REQUIRE(dy && dx && B > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && C > 0, "bilinear_bwd: bad arguments");
    const int V = vec_for(C, dtype);
    const float sy = ac_scale(Hi, Ho), sx = ac_scale(Wi, Wo);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    DISPATCH_TV(dtype, V, {
        hipLaunchKernelGGL((bilinear_bwd_kernel<T_, V_>), dim3(grid_for((long)B * Hi * Wi * (C / V_))), dim3(256), 0,
                           s, (const T_*)dy, (T_*)dx, B, Hi, Wi, Ho, Wo, C, sy, sx);
    });
    HS_LAUNCH_CHECK("bilinear_bwd");
    return HIPSEG_OK;
}

extern "C" int hipseg_nchw_to_nhwc(int dtype, const float* x, void* y, int B, int C, int H, int W,
                                   hipseg_stream_t stream) {
    HS_REQUIRE(dtype == HIPSEG_F32 || dtype == HIPSEG_BF16, "nchw_to_nhwc: bad dtype");
    HS_REQUIRE(x && y && B > 0 && C > 0 && H > 0 && W > 0, "nchw_to_nhwc: bad arguments");
    const long HW = (long)H * W;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == HIPSEG_BF16)
        hipLaunchKernelGGL(nchw_to_nhwc_kernel<bf16>, dim3(grid_for(B * HW * C)), dim3(256), 0, s, x, (bf16*)y, B, C,
                           HW);
    else
        hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, dim3(grid_for(B * HW * C)), dim3(256), 0, s, x, (float*)y, B, C,
                           HW);
    HS_LAUNCH_CHECK("nchw_to_nhwc");
    return HIPSEG_OK;
}

extern "C" int hipseg_nhwc_to_nchw(int dtype, const void* x, float* y, int B, int C, int H, int W,
                                   hipseg_stream_t stream) {
    HS_REQUIRE(dtype == HIPSEG_F32 || dtype == HIPSEG_BF16, "nhwc_to_nchw: bad dtype");
    HS_REQUIRE(x && y && B > 0 && C > 0 && H > 0 && W > 0, "nhwc_to_nchw: bad arguments");
    const long HW = (long)H * W;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == HIPSEG_BF16)
        hipLaunchKernelGGL(nhwc_to_nchw_kernel<bf16>, dim3(grid_for(B * HW * C)), dim3(256), 0, s, (const bf16*)x, y,
                           B, C, HW);
    else
        hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, dim3(grid_for(B * HW * C)), dim3(256), 0, s, (const float*)x, y,
                           B, C, HW);
    HS_LAUNCH_CHECK("nhwc_to_nchw");
    return HIPSEG_OK;
}
