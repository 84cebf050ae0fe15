// Per-pixel losses and the segmentation confusion matrix (models/losses.py), wave-level reductions.
// Logits are NCHW fp32 (B,C,H,W): the class planes are read coalesced along the pixel index.
#include "common.h"

namespace {

constexpr int MAXC = 16;
constexpr long IGNORE_INDEX = -100;  // nn.CrossEntropyLoss default

__device__ __forceinline__ float block_sum(float v, float* sm) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sm[w] = v;
    __syncthreads();
    float t = 0.f;
    if (threadIdx.x == 0)
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += sm[i];
    return t;  // valid on thread 0
}

// partial[blk][2] = {sum nll, count of non-ignored targets}.  V = 4: a lane handles 4 consecutive pixels of one image
// (16-byte loads from every class plane and 2 x 16 bytes of targets: one pixel per lane left too few bytes in flight --
// 13.8 us for 21 MB); V = 1 when HW is not a multiple of 4.
template <int V>
__global__ __launch_bounds__(256) void ce_fwd_kernel(const float* __restrict__ logits, const long long* __restrict__ tgt,
                                                      float* __restrict__ partial, int B, int C, long HW) {
    __shared__ float sm[4];
    const long ngrp = (long)B * HW / V;
    float nll = 0.f, cnt = 0.f;
    for (long g = blockIdx.x * (long)blockDim.x + threadIdx.x; g < ngrp; g += (long)gridDim.x * blockDim.x) {
        const long p = g * V;
        // (32-bit index arithmetic where it fits: a 64-bit divide is ~100 VALU instructions)
        long n, hw;
        if (ngrp * V < (1l << 31)) {
            const unsigned pu = (unsigned)p, hwn = (unsigned)HW, nu = pu / hwn;
            n = nu;
            hw = pu - nu * hwn;
        } else {
            n = p / HW;
            hw = p % HW;
        }
        long t[V];
        float v[MAXC][V];
#pragma unroll
        for (int i = 0; i < V; ++i) t[i] = tgt[p + i];
#pragma unroll
        for (int c = 0; c < MAXC; ++c)
            if (c < C) {
                const float* src = logits + (n * C + c) * HW + hw;
                if (V == 4) {
                    const f32x4 q = *reinterpret_cast<const f32x4*>(src);
#pragma unroll
                    for (int i = 0; i < V; ++i) v[c][i] = q[i];
                } else {
                    v[c][0] = src[0];
                }
            }
#pragma unroll
        for (int i = 0; i < V; ++i) {
            if (t[i] == IGNORE_INDEX) continue;
            float m = -INFINITY;
#pragma unroll
            for (int c = 0; c < MAXC; ++c)
                if (c < C) m = fmaxf(m, v[c][i]);
            float se = 0.f, vt = 0.f;
#pragma unroll
            for (int c = 0; c < MAXC; ++c)
                if (c < C) {
                    se += expf(v[c][i] - m);
                    if (c == (int)t[i]) vt = v[c][i];
                }
            nll += (m + logf(se)) - vt;
            cnt += 1.f;
        }
    }
    const float a = block_sum(nll, sm);
    const float b = block_sum(cnt, sm);
    if (threadIdx.x == 0) {
        partial[blockIdx.x * 2 + 0] = a;
        partial[blockIdx.x * 2 + 1] = b;
    }
}

// loss[0] = mean nll, loss[1] = count
__global__ __launch_bounds__(256) void ce_finalize_kernel(const float* __restrict__ partial, int nblk,
                                                           float* __restrict__ loss) {
    __shared__ double sm[2][4];
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < nblk; i += blockDim.x) {
        a += (double)partial[i * 2 + 0];
        b += (double)partial[i * 2 + 1];
    }
    a = wave_sum_d(a);
    b = wave_sum_d(b);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) {
        sm[0][w] = a;
        sm[1][w] = b;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        a = sm[0][0] + sm[0][1] + sm[0][2] + sm[0][3];
        b = sm[1][0] + sm[1][1] + sm[1][2] + sm[1][3];
        loss[0] = (float)(a / b);
        loss[1] = (float)b;
    }
}

template <int V>
__global__ __launch_bounds__(256) void ce_bwd_kernel(const float* __restrict__ logits, const long long* __restrict__ tgt,
                                                      const float* __restrict__ gscale,
                                                      const float* __restrict__ loss, float* __restrict__ dl, int B,
                                                      int C, long HW) {
    const long ngrp = (long)B * HW / V;
    const float k = gscale[0] / loss[1];
    for (long g = blockIdx.x * (long)blockDim.x + threadIdx.x; g < ngrp; g += (long)gridDim.x * blockDim.x) {
        const long p = g * V;
        // (32-bit index arithmetic where it fits: a 64-bit divide is ~100 VALU instructions)
        long n, hw;
        if (ngrp * V < (1l << 31)) {
            const unsigned pu = (unsigned)p, hwn = (unsigned)HW, nu = pu / hwn;
            n = nu;
            hw = pu - nu * hwn;
        } else {
            n = p / HW;
            hw = p % HW;
        }
        long t[V];
        float v[MAXC][V];
#pragma unroll
        for (int i = 0; i < V; ++i) t[i] = tgt[p + i];
#pragma unroll
        for (int c = 0; c < MAXC; ++c)
            if (c < C) {
                const float* src = logits + (n * C + c) * HW + hw;
                if (V == 4) {
                    const f32x4 q = *reinterpret_cast<const f32x4*>(src);
#pragma unroll
                    for (int i = 0; i < V; ++i) v[c][i] = q[i];
                } else {
                    v[c][0] = src[0];
                }
            }
#pragma unroll
        for (int i = 0; i < V; ++i) {
            float m = -INFINITY;
#pragma unroll
            for (int c = 0; c < MAXC; ++c)
                if (c < C) m = fmaxf(m, v[c][i]);
            float se = 0.f;
#pragma unroll
            for (int c = 0; c < MAXC; ++c)
                if (c < C) {
                    v[c][i] = expf(v[c][i] - m);
                    se += v[c][i];
                }
            const float inv = 1.f / se;
            const bool ign = t[i] == IGNORE_INDEX;
#pragma unroll
            for (int c = 0; c < MAXC; ++c)
                if (c < C) v[c][i] = ign ? 0.f : (v[c][i] * inv - (c == (int)t[i] ? 1.f : 0.f)) * k;
        }
#pragma unroll
        for (int c = 0; c < MAXC; ++c)
            if (c < C) {
                float* d = dl + (n * C + c) * HW + hw;
                if (V == 4)
                    *reinterpret_cast<f32x4*>(d) = f32x4{v[c][0], v[c][1], v[c][2], v[c][3]};
                else
                    d[0] = v[c][0];
            }
    }
}

// ------------------------------------------------------------------ BCE-with-logits + smp Dice (binary)
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// partial[blk][4] = {sum bce, sum p*t, sum p, sum t}, p = sigmoid(sigmoid(x))
__global__ __launch_bounds__(256) void bce_dice_fwd_kernel(const float* __restrict__ x, const float* __restrict__ t,
                                                            float* __restrict__ partial, long n) {
    __shared__ float sm[4];
    float a = 0.f, b = 0.f, c = 0.f, d = 0.f;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float xv = x[i], tv = t[i];
        a += fmaxf(xv, 0.f) - xv * tv + log1pf(expf(-fabsf(xv)));
        const float p = sigmoidf_(sigmoidf_(xv));
        b += p * tv;
        c += p;
        d += tv;
    }
    const float ra = block_sum(a, sm), rb = block_sum(b, sm), rc = block_sum(c, sm), rd = block_sum(d, sm);
    if (threadIdx.x == 0) {
        partial[blockIdx.x * 4 + 0] = ra;
        partial[blockIdx.x * 4 + 1] = rb;
        partial[blockIdx.x * 4 + 2] = rc;
        partial[blockIdx.x * 4 + 3] = rd;
    }
}

__global__ __launch_bounds__(256) void bce_dice_finalize_kernel(const float* __restrict__ partial, int nblk, double n,
                                                                 float* __restrict__ sums, float* __restrict__ loss) {
    __shared__ double sm[4][4];
    double v[4] = {0, 0, 0, 0};
    for (int i = threadIdx.x; i < nblk; i += blockDim.x)
        for (int k = 0; k < 4; ++k) v[k] += (double)partial[i * 4 + k];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int k = 0; k < 4; ++k) {
        v[k] = wave_sum_d(v[k]);
        if (lane == 0) sm[k][w] = v[k];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 0; k < 4; ++k) {
            v[k] = sm[k][0] + sm[k][1] + sm[k][2] + sm[k][3];
            sums[k] = (float)v[k];
        }
        const double card = v[2] + v[3];
        const double score = 2.0 * v[1] / (card > 1e-7 ? card : 1e-7);
        const double dice = (v[3] > 0.0) ? (1.0 - score) : 0.0;
        loss[0] = (float)(v[0] / n + dice);
    }
}

__global__ __launch_bounds__(256) void bce_dice_bwd_kernel(const float* __restrict__ x, const float* __restrict__ t,
                                                            const float* __restrict__ sums,
                                                            const float* __restrict__ gscale, float* __restrict__ dl,
                                                            long n) {
    const float gs = gscale[0];
    const float spt = sums[1], card = sums[2] + sums[3], st = sums[3];
    const float inv_n = 1.f / (float)n;
    const bool dice_on = st > 0.f;
    const bool clamped = !(card > 1e-7f);
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float xv = x[i], tv = t[i];
        const float s = sigmoidf_(xv), p = sigmoidf_(s);
        float g = (s - tv) * inv_n;
        if (dice_on) {
            // d(1 - 2*spt/card)/dp_i = -2*(t_i*card - spt)/card^2   (denominator clamp: -2*t_i/eps)
            const float dd = clamped ? (-2.f * tv / 1e-7f) : (-2.f * (tv * card - spt) / (card * card));
            g += dd * p * (1.f - p) * s * (1.f - s);
        }
        dl[i] = g * gs;
    }
}

// ------------------------------------------------------------------ confusion matrix of argmax vs target
__global__ __launch_bounds__(256) void confusion_kernel(const float* __restrict__ logits,
                                                         const long long* __restrict__ tgt,
                                                         unsigned long long* __restrict__ conf, int B, int C, long HW) {
    __shared__ unsigned int sc[MAXC * MAXC];
    for (int i = threadIdx.x; i < C * C; i += blockDim.x) sc[i] = 0u;
    __syncthreads();
    const long npix = (long)B * HW;
    for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < npix; p += (long)gridDim.x * blockDim.x) {
        const long n = p / HW, hw = p % HW;
        float best = -INFINITY;
        int bc = 0;
        for (int c = 0; c < C; ++c) {
            const float v = logits[(n * C + c) * HW + hw];
            if (v > best) {  // first maximum wins, as torch.argmax
                best = v;
                bc = c;
            }
        }
        const long t = tgt[p];
        if (t >= 0 && t < C) atomicAdd(&sc[(int)t * C + bc], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C * C; i += blockDim.x)
        if (sc[i]) atomicAdd(&conf[i], (unsigned long long)sc[i]);
}

__global__ void zero_u64_kernel(unsigned long long* p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0ull;
}

}  // namespace

extern "C" int hipseg_loss_blocks(long n) {
    long nb = (n + 1023) / 1024;
    return (int)(nb > 1024 ? 1024 : (nb < 1 ? 1 : nb));
}

extern "C" int hipseg_ce_fwd(const float* logits, const int64_t* target, float* partial, float* loss, int B, int C,
                             long HW, hipseg_stream_t stream) {
    HS_REQUIRE(logits && target && partial && loss && B > 0 && HW > 0, "ce_fwd: bad arguments");
    HS_REQUIRE(C >= 1 && C <= MAXC, "ce_fwd: %d classes unsupported (1..%d)", C, MAXC);
    const int nblk = hipseg_loss_blocks((long)B * HW);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (HW % 4 == 0 && (reinterpret_cast<uintptr_t>(logits) | reinterpret_cast<uintptr_t>(target)) % 16 == 0)
        hipLaunchKernelGGL(ce_fwd_kernel<4>, dim3(nblk), dim3(256), 0, s, logits, (const long long*)target, partial, B, C, HW);
    else
        hipLaunchKernelGGL(ce_fwd_kernel<1>, dim3(nblk), dim3(256), 0, s, logits, (const long long*)target, partial, B, C, HW);
    HS_LAUNCH_CHECK("ce_fwd");
    hipLaunchKernelGGL(ce_finalize_kernel, dim3(1), dim3(256), 0, s, partial, nblk, loss);
    HS_LAUNCH_CHECK("ce_finalize");
    return HIPSEG_OK;
}

extern "C" int hipseg_ce_bwd(const float* logits, const int64_t* target, const float* gscale, const float* loss2,
                             float* dlogits, int B, int C, long HW, hipseg_stream_t stream) {
    HS_REQUIRE(logits && target && gscale && loss2 && dlogits && B > 0 && HW > 0, "ce_bwd: bad arguments");
    HS_REQUIRE(C >= 1 && C <= MAXC, "ce_bwd: %d classes unsupported (1..%d)", C, MAXC);
    const int nblk = hipseg_loss_blocks((long)B * HW);
    if (HW % 4 == 0 &&
        (reinterpret_cast<uintptr_t>(logits) | reinterpret_cast<uintptr_t>(target) | reinterpret_cast<uintptr_t>(dlogits)) % 16 == 0)
        hipLaunchKernelGGL(ce_bwd_kernel<4>, dim3(nblk), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), logits,
                           (const long long*)target, gscale, loss2, dlogits, B, C, HW);
    else
        hipLaunchKernelGGL(ce_bwd_kernel<1>, dim3(nblk), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), logits,
                           (const long long*)target, gscale, loss2, dlogits, B, C, HW);
    HS_LAUNCH_CHECK("ce_bwd");
    return HIPSEG_OK;
}

extern "C" int hipseg_bce_dice_fwd(const float* logits, const float* target, float* partial, float* sums,
                                   float* loss, long n, hipseg_stream_t stream) {
    HS_REQUIRE(logits && target && partial && sums && loss && n > 0, "bce_dice_fwd: bad arguments");
    const int nblk = hipseg_loss_blocks(n);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(bce_dice_fwd_kernel, dim3(nblk), dim3(256), 0, s, logits, target, partial, n);
    HS_LAUNCH_CHECK("bce_dice_fwd");
    hipLaunchKernelGGL(bce_dice_finalize_kernel, dim3(1), dim3(256), 0, s, partial, nblk, (double)n, sums, loss);
    HS_LAUNCH_CHECK("bce_dice_finalize");
    return HIPSEG_OK;
}

extern "C" int hipseg_bce_dice_bwd(const float* logits, const float* target, const float* sums, const float* gscale,
                                   float* dlogits, long n, hipseg_stream_t stream) {
    HS_REQUIRE(logits && target && sums && gscale && dlogits && n > 0, "bce_dice_bwd: bad arguments");
    hipLaunchKernelGGL(bce_dice_bwd_kernel, dim3(hipseg_loss_blocks(n)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), logits, target, sums, gscale, dlogits, n);
    HS_LAUNCH_CHECK("bce_dice_bwd");
    return HIPSEG_OK;
}

extern "C" int hipseg_confusion(const float* logits, const int64_t* target, long long* conf, int B, int C, long HW,
                                hipseg_stream_t stream) {
    HS_REQUIRE(logits && target && conf && B > 0 && HW > 0, "confusion: bad arguments");
    HS_REQUIRE(C >= 1 && C <= MAXC, "confusion: %d classes unsupported (1..%d)", C, MAXC);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(zero_u64_kernel, dim3(1), dim3(256), 0, s, (unsigned long long*)conf, C * C);
    HS_LAUNCH_CHECK("confusion_zero");
    hipLaunchKernelGGL(confusion_kernel, dim3(hipseg_loss_blocks((long)B * HW)), dim3(256), 0, s, logits,
                       (const long long*)target, (unsigned long long*)conf, B, C, HW);
    HS_LAUNCH_CHECK("confusion");
    return HIPSEG_OK;
}
