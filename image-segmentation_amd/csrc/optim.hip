// Multi-tensor Adam step (the optimiser half of the timed train step): ONE launch updates every parameter of a group.
// Replaces: torch.optim.Adam(lr, betas, eps, weight_decay) as constructed by the reference's wrappers
// (models/model_wrappers.py:124 `optimizer_class(self.model.parameters(), **optimizer_args)`, stepped through
// GradScaler at :176) -- same arithmetic (L2 weight decay added to the gradient, bias-corrected moments):
//     g  = grad / grad_scale (+ wd * p)
//     m  = b1 m + (1 - b1) g          v = b2 v + (1 - b2) g^2
//     p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// HBM-bound: 4 reads + 3 writes of 4 B per parameter element.  A block owns one 4096-element chunk of one tensor; up
// to 64 tensor descriptors travel BY VALUE in the kernel arguments (nothing to upload, nothing to keep alive, and a
// hipGraph captures them), so a 124-tensor U-Net takes two launches; 16-byte vectors where all four pointers are
// aligned.  The step counter lives on the device (hipGraph-replayable): every block reads it, the LAST block of the
// LAST launch advances it.  `found_inf` != 0 (GradScaler) turns the whole step into a no-op, counter included.
#include "common.h"

namespace {

struct AdamDesc {
    float* p;
    const float* g;
    float* m;
    float* v;
    long long n;
};

constexpr int CHUNK = 4096;
constexpr int MAXT = 88;  // tensors per launch: 88 x 40 B + 89 x 4 B + ~60 B of kernel arguments (< 4 KiB); the U-Net's 76 fit one

struct AdamBatch {
    AdamDesc d[MAXT];
    int start[MAXT + 1];  // first block of tensor i (prefix sums of its chunk counts)
    int nt;
};

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, float inv_scale, float wd, float b1,
                                         float b2, float step_size, float inv_bc2_sqrt, float eps) {
    g *= inv_scale;
    g += wd * p;
    m = b1 * m + (1.0f - b1) * g;
    v = b2 * v + (1.0f - b2) * g * g;
    const float denom = sqrtf(v) * inv_bc2_sqrt + eps;
    p -= step_size * (m / denom);
}

__global__ __launch_bounds__(256) void adam_kernel(const AdamBatch batch, int* __restrict__ state,
                                                   const float* __restrict__ found_inf,
                                                   const float* __restrict__ grad_scale, float lr, float b1, float b2,
                                                   float eps, float wd) {
    if (found_inf && *found_inf != 0.0f) return;  // uniform over the grid: nothing moves, the step is not counted
    const int step = state[0] + 1;
    const double bc1 = 1.0 - pow((double)b1, (double)step), bc2 = 1.0 - pow((double)b2, (double)step);
    const float step_size = (float)((double)lr / bc1), inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
    const float inv_scale = grad_scale ? 1.0f / *grad_scale : 1.0f;
    int lo = 0, hi = batch.nt - 1;  // tensor of this block: last i with start[i] <= blockIdx.x
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (batch.start[mid] <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const AdamDesc& d = batch.d[lo];
    const long long base = (long long)((int)blockIdx.x - batch.start[lo]) * CHUNK;
    long long cnt = d.n - base;
    if (cnt > CHUNK) cnt = CHUNK;
    float* p = d.p + base;
    const float* g = d.g + base;
    float* m = d.m + base;
    float* v = d.v + base;
    const bool vec = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                       reinterpret_cast<uintptr_t>(v)) & 15) == 0;
    if (vec) {
        const int nv = (int)(cnt >> 2);
        for (int i = threadIdx.x; i < nv; i += 256) {
            f32x4 pp = reinterpret_cast<f32x4*>(p)[i], mm = reinterpret_cast<f32x4*>(m)[i], vv = reinterpret_cast<f32x4*>(v)[i];
            const f32x4 gg = reinterpret_cast<const f32x4*>(g)[i];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float a = pp[k], b = mm[k], c = vv[k];
                adam_one(a, gg[k], b, c, inv_scale, wd, b1, b2, step_size, inv_bc2_sqrt, eps);
                pp[k] = a, mm[k] = b, vv[k] = c;
            }
            reinterpret_cast<f32x4*>(p)[i] = pp, reinterpret_cast<f32x4*>(m)[i] = mm, reinterpret_cast<f32x4*>(v)[i] = vv;
        }
        for (int i = (nv << 2) + threadIdx.x; i < cnt; i += 256)
            adam_one(p[i], g[i], m[i], v[i], inv_scale, wd, b1, b2, step_size, inv_bc2_sqrt, eps);
    } else {
        for (int i = threadIdx.x; i < cnt; i += 256)
            adam_one(p[i], g[i], m[i], v[i], inv_scale, wd, b1, b2, step_size, inv_bc2_sqrt, eps);
    }
}

// found[0] = 1 when any gradient element of the batch is inf / NaN (the caller zeroed it): GradScaler's inf check without
// its unscale pass -- torch's `_amp_foreach_non_finite_check_and_unscale_` reads AND rewrites every gradient (x 1.0 when
// the optimizer takes the scale itself) through a multi-tensor launch: 31 us for the U-Net's 31 MB, 75 us for LargeUNet.
__global__ __launch_bounds__(256) void grads_nonfinite_kernel(const AdamBatch batch, float* __restrict__ found) {
    int lo = 0, hi = batch.nt - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (batch.start[mid] <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const AdamDesc& d = batch.d[lo];
    const long long base = (long long)((int)blockIdx.x - batch.start[lo]) * CHUNK;
    long long cnt = d.n - base;
    if (cnt > CHUNK) cnt = CHUNK;
    const float* g = d.g + base;
    bool bad = false;
    if ((reinterpret_cast<uintptr_t>(g) & 15) == 0) {
        const int nv = (int)(cnt >> 2);
        for (int i = threadIdx.x; i < nv; i += 256) {
            const f32x4 v = reinterpret_cast<const f32x4*>(g)[i];
            // (x - x is 0 for every finite x and NaN for inf / NaN)
            const float t = (v[0] - v[0]) + (v[1] - v[1]) + (v[2] - v[2]) + (v[3] - v[3]);
            bad = bad || (t != 0.f);
        }
        for (int i = (nv << 2) + threadIdx.x; i < cnt; i += 256) bad = bad || (g[i] - g[i] != 0.f);
    } else {
        for (int i = threadIdx.x; i < cnt; i += 256) bad = bad || (g[i] - g[i] != 0.f);
    }
    if (bad) found[0] = 1.0f;  // (every writer stores the same value)
}

// The step counter moves in a launch of its own, behind every adam_kernel launch of the step in stream order (they all
// read it).  Round 3 let the last block to arrive advance it (arrival count + __threadfence in EVERY block of the last
// launch): with all the U-Net's tensors in one launch that is a device-scope fence per 4096-element chunk -- measured on
// ClipUnet (80 tensors, one launch): 165 us for 247 MB = 1.5 TB/s against 6.1 TB/s for LargeUNet, whose last launch
// was a small one.
__global__ void adam_advance_kernel(int* __restrict__ state, const float* __restrict__ found_inf) {
    if (found_inf && *found_inf != 0.0f) return;
    state[0] += 1;
}

}  // namespace

extern "C" size_t hipseg_adam_desc_size(void) { return sizeof(AdamDesc); }

extern "C" int hipseg_adam_desc_fill(void* host_descs, int index, float* p, const float* g, float* m, float* v, long n) {
    HS_REQUIRE(host_descs && index >= 0 && p && g && m && v && n > 0, "adam_desc_fill: bad arguments");
    AdamDesc& d = reinterpret_cast<AdamDesc*>(host_descs)[index];
    d.p = p, d.g = g, d.m = m, d.v = v, d.n = n;
    return HIPSEG_OK;
}

extern "C" int hipseg_grads_nonfinite(const void* host_descs, int ntensors, float* found, hipseg_stream_t stream) {
    HS_REQUIRE(host_descs && found && ntensors > 0, "grads_nonfinite: null table or no tensors");
    const AdamDesc* all = reinterpret_cast<const AdamDesc*>(host_descs);
    for (int t0 = 0; t0 < ntensors; t0 += MAXT) {
        AdamBatch b;
        b.nt = ntensors - t0 < MAXT ? ntensors - t0 : MAXT;
        long nblk = 0;
        for (int i = 0; i < b.nt; ++i) {
            HS_REQUIRE(all[t0 + i].g && all[t0 + i].n > 0, "grads_nonfinite: bad descriptor %d", t0 + i);
            b.d[i] = all[t0 + i];
            b.start[i] = (int)nblk;
            nblk += (all[t0 + i].n + CHUNK - 1) / CHUNK;
        }
        b.start[b.nt] = (int)nblk;
        HS_REQUIRE(nblk < (1l << 31), "grads_nonfinite: too many elements in one launch");
        hipLaunchKernelGGL(grads_nonfinite_kernel, dim3((unsigned)nblk), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), b,
                           found);
        HS_LAUNCH_CHECK("grads_nonfinite");
    }
    return HIPSEG_OK;
}

extern "C" int hipseg_adam_step(const void* host_descs, int ntensors, int* state, const float* found_inf,
                                const float* grad_scale, float lr, float beta1, float beta2, float eps,
                                float weight_decay, hipseg_stream_t stream) {
    HS_REQUIRE(host_descs && state && ntensors > 0, "adam_step: null table or no tensors");
    HS_REQUIRE(lr >= 0.f && beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f && eps >= 0.f && weight_decay >= 0.f,
               "adam_step: bad hyper-parameters (lr %g betas %g %g eps %g wd %g)", lr, beta1, beta2, eps, weight_decay);
    const AdamDesc* all = reinterpret_cast<const AdamDesc*>(host_descs);
    for (int i = 0; i < ntensors; ++i)
        HS_REQUIRE(all[i].p && all[i].g && all[i].m && all[i].v && all[i].n > 0, "adam_step: bad descriptor %d", i);
    for (int t0 = 0; t0 < ntensors; t0 += MAXT) {
        AdamBatch b;
        b.nt = ntensors - t0 < MAXT ? ntensors - t0 : MAXT;
        long nblk = 0;
        for (int i = 0; i < b.nt; ++i) {
            b.d[i] = all[t0 + i];
            b.start[i] = (int)nblk;
            nblk += (all[t0 + i].n + CHUNK - 1) / CHUNK;
        }
        b.start[b.nt] = (int)nblk;
        HS_REQUIRE(nblk < (1l << 31), "adam_step: too many elements in one launch");
        hipLaunchKernelGGL(adam_kernel, dim3((unsigned)nblk), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), b, state,
                           found_inf, grad_scale, lr, beta1, beta2, eps, weight_decay);
        HS_LAUNCH_CHECK("adam_step");
    }
    hipLaunchKernelGGL(adam_advance_kernel, dim3(1), dim3(1), 0, reinterpret_cast<hipStream_t>(stream), state, found_inf);
    HS_LAUNCH_CHECK("adam_advance");
    return HIPSEG_OK;
}
