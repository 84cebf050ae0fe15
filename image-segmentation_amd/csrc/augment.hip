// On-device training augmentation: the per-step `DataAugmentor.forward(images, masks)` of the
// reference (models/processing_blocks.py:344-384, called at models/model_wrappers.py:165) as two HBM-bound kernels.
//
//   geometric (image + mask [+ prompt channels] together): RandomHorizontalFlip, then RandomRotation(+-90 deg, nearest,
//       zeros padding) about the image centre -- one nearest-neighbour gather;
//   colour (image only): ColorJitter(brightness, contrast, saturation, hue, in a per-call random ORDER) and a 5x5
//       Gaussian blur (reflect border, per-sample sigma);
//   every (augmentations_per_datapoint + 1)-th sample is passed through untouched (params[.][0] = keep).
//
// The arithmetic the reference gets from kornia 0.8.0 (absent here: PARITY UNPINNED) is restated from kornia's
// published algorithms: adjust_brightness_accumulative (x*f, clamp), adjust_contrast_with_mean_subtraction
// ((x - mean(gray))*f + mean, clamp), adjust_saturation_with_gray_subtraction ((x - gray)*f + gray, clamp), adjust_hue
// (RGB -> HSV, h += f, -> RGB), get_gaussian_kernel1d, warp_affine(nearest, zeros, align_corners=True).  The random
// parameters themselves are sampled by the host (torch RNG on the device, no sync) and handed over as a table.
//
// Contrast needs the mean grey level of the WHOLE image after the colour ops that precede it, so pass 1 reduces that
// (fixed-order: 64 partials per sample), pass 2 recomputes the pixel pipeline into an LDS tile with a 2-pixel halo and
// blurs out of LDS.  Traffic: pass 1 reads 12 B/pixel, pass 2 reads ~12 + 8 and writes 12 + 8 B/pixel.
#include "common.h"

namespace {

constexpr int NP = HIPSEG_AUG_NPARAM;  // floats per sample in the parameter table
constexpr int NPART = 64;              // grey-mean partials per sample
constexpr int TW = 32, TH = 16, HALO = 2;
constexpr int LW = TW + 2 * HALO, LH = TH + 2 * HALO;

struct Geo {
    float c, s, cx, cy;
    int flip, H, W;
};

__device__ __forceinline__ Geo make_geo(const float* p, int H, int W) {
    Geo g;
    g.flip = p[1] != 0.0f;
    g.c = p[2];
    g.s = p[3];
    g.cx = 0.5f * (float)(W - 1);
    g.cy = 0.5f * (float)(H - 1);
    g.H = H;
    g.W = W;
    return g;
}

// source pixel of output pixel (x, y): inverse rotation about the centre, nearest (ties to even, as grid_sample),
// then the flip; returns -1 when the source falls outside the image (zeros padding)
__device__ __forceinline__ long src_index(const Geo& g, int x, int y) {
    const float dx = (float)x - g.cx, dy = (float)y - g.cy;
    const float fx = g.c * dx - g.s * dy + g.cx;
    const float fy = g.s * dx + g.c * dy + g.cy;
    int sx = (int)nearbyintf(fx), sy = (int)nearbyintf(fy);
    if (sx < 0 || sx >= g.W || sy < 0 || sy >= g.H) return -1;
    if (g.flip) sx = g.W - 1 - sx;
    return (long)sy * g.W + sx;
}

__device__ __forceinline__ float clamp01(float v) { return fminf(fmaxf(v, 0.0f), 1.0f); }
__device__ __forceinline__ float gray_of(float r, float g, float b) { return 0.299f * r + 0.587f * g + 0.114f * b; }

__device__ __forceinline__ void hue_shift(float& r, float& g, float& b, float shift) {
    const float TWO_PI = 6.283185307179586f;
    // rgb -> hsv (h in radians)
    const float mx = fmaxf(r, fmaxf(g, b)), mn = fminf(r, fminf(g, b));
    float dc = mx - mn;
    const float v = mx, s = dc / (mx + 1e-8f);
    if (dc == 0.0f) dc = 1.0f;
    const float rc = mx - r, gc = mx - g, bc = mx - b;
    float h = (r == mx) ? (bc - gc) : ((g == mx) ? (rc - bc) + 2.0f * dc : (gc - rc) + 4.0f * dc);  // first max wins
    h = h / dc / 6.0f;
    h = h - floorf(h);  // python-style % 1
    h = TWO_PI * h;
    h = fmodf(h + shift, TWO_PI);  // torch.fmod: sign of the dividend
    // hsv -> rgb
    const float h6 = h / TWO_PI * 6.0f;
    const float m6 = h6 - 6.0f * floorf(h6 / 6.0f);  // python-style % 6
    float hi = floorf(h6);
    hi = hi - 6.0f * floorf(hi / 6.0f);
    const float f = m6 - hi;
    const float p = v * (1.0f - s), q = v * (1.0f - f * s), t = v * (1.0f - (1.0f - f) * s);
    const int k = (int)hi;
    r = (k == 0 || k == 5) ? v : ((k == 1) ? q : ((k == 4) ? t : p));
    g = (k == 1 || k == 2) ? v : ((k == 3) ? q : ((k == 0) ? t : p));
    b = (k == 3 || k == 4) ? v : ((k == 5) ? q : ((k == 2) ? t : p));
}

// colour ops order[0..n) in sequence; op ids: 0 brightness, 1 contrast, 2 saturation, 3 hue.
// `stop_at_contrast`: return before the contrast op (pass 1 wants the image the contrast op will see)
__device__ __forceinline__ void colour_ops(float& r, float& g, float& b, const float* p, const int* order, float mean,
                                           bool stop_at_contrast) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int op = order[i];
        if (op == 0) {
            const float f = p[4];
            r = clamp01(r * f), g = clamp01(g * f), b = clamp01(b * f);
        } else if (op == 1) {
            if (stop_at_contrast) return;
            const float f = p[5];
            r = clamp01((r - mean) * f + mean), g = clamp01((g - mean) * f + mean), b = clamp01((b - mean) * f + mean);
        } else if (op == 2) {
            const float f = p[6], y = gray_of(r, g, b);
            r = clamp01((r - y) * f + y), g = clamp01((g - y) * f + y), b = clamp01((b - y) * f + y);
        } else {
            hue_shift(r, g, b, p[7]);
        }
    }
}

// pass 1: partial[b][blk] = sum over this block's pixels of gray(image after the ops preceding contrast)
__global__ __launch_bounds__(256) void augment_gray_partial_kernel(const float* __restrict__ images,
                                                                   const float* __restrict__ params,
                                                                   const int* __restrict__ order,
                                                                   float* __restrict__ partial, int H, int W) {
    __shared__ float sm[4];
    const int b = blockIdx.y;
    const float* p = params + (size_t)b * NP;
    float acc = 0.0f;
    if (p[0] == 0.0f) {  // kept samples need no mean
        const long HW = (long)H * W;
        const float* img = images + (size_t)b * 3 * HW;
        const Geo geo = make_geo(p, H, W);
        int ord[4] = {order[0], order[1], order[2], order[3]};
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < HW; i += (long)NPART * 256) {
            const int y = (int)(i / W), x = (int)(i - (long)y * W);
            const long s = src_index(geo, x, y);
            float r = 0.0f, g = 0.0f, bl = 0.0f;
            if (s >= 0) r = img[s], g = img[HW + s], bl = img[2 * HW + s];
            colour_ops(r, g, bl, p, ord, 0.0f, true);
            acc += gray_of(r, g, bl);
        }
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[(size_t)b * NPART + blockIdx.x] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
}

__device__ __forceinline__ int reflect(int v, int n) {  // torch 'reflect' padding: -1 -> 1, n -> n-2
    if (v < 0) v = -v;
    if (v >= n) v = 2 * (n - 1) - v;
    return v;
}

// pass 2: one 32x16 output tile per block
__global__ __launch_bounds__(256) void augment_apply_kernel(const float* __restrict__ images,
                                                            const long long* __restrict__ masks,
                                                            const float* __restrict__ extra, int n_extra,
                                                            const float* __restrict__ params,
                                                            const int* __restrict__ order,
                                                            const float* __restrict__ partial,
                                                            float* __restrict__ out_images,
                                                            long long* __restrict__ out_masks,
                                                            float* __restrict__ out_extra, int H, int W) {
    __shared__ float tile[3][LH][LW + 1];
    __shared__ float s_mean;
    const int b = blockIdx.z;
    const long HW = (long)H * W;
    const float* p = params + (size_t)b * NP;
    const float* img = images + (size_t)b * 3 * HW;
    float* oimg = out_images + (size_t)b * 3 * HW;
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
    const int tx = threadIdx.x & (TW - 1), ty = threadIdx.x / TW;  // 32 x 8 threads, two rows each
    const bool keep = p[0] != 0.0f;
    if (keep) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int x = x0 + tx, y = y0 + ty + 8 * k;
            if (x < W && y < H) {
                const long i = (long)y * W + x;
                oimg[i] = img[i], oimg[HW + i] = img[HW + i], oimg[2 * HW + i] = img[2 * HW + i];
                if (masks) out_masks[(size_t)b * HW + i] = masks[(size_t)b * HW + i];
                for (int e = 0; e < n_extra; ++e)
                    out_extra[((size_t)b * n_extra + e) * HW + i] = extra[((size_t)b * n_extra + e) * HW + i];
            }
        }
        return;
    }
    if (threadIdx.x < 64) {  // grey mean of the sample: fixed-order butterfly over the 64 partials
        const float t = wave_sum(partial[(size_t)b * NPART + threadIdx.x]);
        if (threadIdx.x == 0) s_mean = t / (float)HW;
    }
    __syncthreads();
    const float mean = s_mean;
    const Geo geo = make_geo(p, H, W);
    int ord[4] = {order[0], order[1], order[2], order[3]};
    for (int i = threadIdx.x; i < LW * LH; i += 256) {
        const int hy = i / LW, hx = i - hy * LW;
        const int gx = reflect(x0 - HALO + hx, W), gy = reflect(y0 - HALO + hy, H);
        float r = 0.0f, g = 0.0f, bl = 0.0f;
        if (gx >= 0 && gx < W && gy >= 0 && gy < H) {  // (tiles hanging over the far edge reflect out of range: unused)
            const long s = src_index(geo, gx, gy);
            if (s >= 0) r = img[s], g = img[HW + s], bl = img[2 * HW + s];
            colour_ops(r, g, bl, p, ord, mean, false);
        }
        tile[0][hy][hx] = r, tile[1][hy][hx] = g, tile[2][hy][hx] = bl;
    }
    // 1-D Gaussian taps (kornia get_gaussian_kernel1d(5, sigma)): exp(-x^2 / (2 sigma^2)), normalised
    const float sg = p[8];
    float w[5];
    {
        const float w1 = expf(-1.0f / (2.0f * sg * sg)), w2 = expf(-4.0f / (2.0f * sg * sg));
        const float n = 1.0f + 2.0f * w1 + 2.0f * w2;
        w[0] = w[4] = w2 / n, w[1] = w[3] = w1 / n, w[2] = 1.0f / n;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int ly = ty + 8 * k;
        const int x = x0 + tx, y = y0 + ly;
        if (x >= W || y >= H) continue;
        const long i = (long)y * W + x;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float acc = 0.0f;
#pragma unroll
            for (int dy = 0; dy < 5; ++dy) {
                float row = 0.0f;
#pragma unroll
                for (int dx = 0; dx < 5; ++dx) row += w[dx] * tile[c][ly + dy][tx + dx];
                acc += w[dy] * row;
            }
            oimg[(size_t)c * HW + i] = acc;
        }
        const long s = src_index(geo, x, y);
        if (masks) out_masks[(size_t)b * HW + i] = s >= 0 ? masks[(size_t)b * HW + s] : 0ll;
        for (int e = 0; e < n_extra; ++e)
            out_extra[((size_t)b * n_extra + e) * HW + i] = s >= 0 ? extra[((size_t)b * n_extra + e) * HW + s] : 0.0f;
    }
}

// parameter table from 8 uniforms per sample (one tiny launch instead of ~20 elementwise host ops)
__global__ void augment_params_kernel(const float* __restrict__ u, float* __restrict__ params, int B, int keep_stride,
                                      float flip_p, float rotate_p, float degrees, float brightness, float contrast,
                                      float saturation, float hue, float sigma_lo, float sigma_hi) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float* r = u + (size_t)b * 8;
    float* p = params + (size_t)b * NP;
    const float theta = (r[1] < rotate_p) ? (r[2] * 2.0f - 1.0f) * degrees * 0.017453292519943295f : 0.0f;
    p[0] = (b % keep_stride == 0) ? 1.0f : 0.0f;
    p[1] = r[0] < flip_p ? 1.0f : 0.0f;
    p[2] = cosf(theta);
    p[3] = sinf(theta);
    p[4] = 1.0f + (r[3] * 2.0f - 1.0f) * brightness;
    p[5] = 1.0f + (r[4] * 2.0f - 1.0f) * contrast;
    p[6] = 1.0f + (r[5] * 2.0f - 1.0f) * saturation;
    p[7] = (r[6] * 2.0f - 1.0f) * hue * 6.283185307179586f;
    p[8] = sigma_lo + r[7] * (sigma_hi - sigma_lo);
    for (int i = 9; i < NP; ++i) p[i] = 0.0f;
}

}  // namespace

extern "C" size_t hipseg_augment_workspace_elems(int B) { return (size_t)(B > 0 ? B : 0) * NPART; }

extern "C" int hipseg_augment_params(const float* uniforms, float* params, int B, int keep_stride, float flip_p,
                                     float rotate_p, float degrees, float brightness, float contrast, float saturation,
                                     float hue, float sigma_lo, float sigma_hi, hipseg_stream_t stream) {
    HS_REQUIRE(uniforms && params && B > 0 && keep_stride >= 1, "augment_params: bad arguments");
    HS_REQUIRE(sigma_lo > 0.f && sigma_hi >= sigma_lo, "augment_params: bad sigma range (%g, %g)", sigma_lo, sigma_hi);
    hipLaunchKernelGGL(augment_params_kernel, dim3(cdiv(B, 64)), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), uniforms,
                       params, B, keep_stride, flip_p, rotate_p, degrees, brightness, contrast, saturation, hue, sigma_lo,
                       sigma_hi);
    HS_LAUNCH_CHECK("augment_params");
    return HIPSEG_OK;
}

extern "C" int hipseg_augment(const float* images, const int64_t* masks, const float* extra, int n_extra,
                              const float* params, const int* order, float* partial, float* out_images,
                              int64_t* out_masks, float* out_extra, int B, int H, int W, hipseg_stream_t stream) {
    HS_REQUIRE(images && params && order && partial && out_images, "augment: null pointer");
    HS_REQUIRE((masks == nullptr) == (out_masks == nullptr), "augment: masks/out_masks must both be given or both be null");
    HS_REQUIRE(n_extra >= 0 && n_extra <= 8 && ((n_extra == 0) || (extra && out_extra)),
               "augment: bad extra channels (n_extra=%d)", n_extra);
    HS_REQUIRE(B > 0 && B <= 65535 && H >= 3 && W >= 3, "augment: bad geometry (B=%d, %dx%d; the 5x5 reflect blur needs >= 3)",
               B, H, W);
    HS_REQUIRE(images != out_images, "augment: in-place operation is not supported (the gather reads the whole image)");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(augment_gray_partial_kernel, dim3(NPART, B), dim3(256), 0, s, images, params, order, partial, H, W);
    HS_LAUNCH_CHECK("augment(gray mean)");
    hipLaunchKernelGGL(augment_apply_kernel, dim3(cdiv(W, TW), cdiv(H, TH), B), dim3(256), 0, s, images,
                       reinterpret_cast<const long long*>(masks), extra, n_extra, params, order, partial, out_images,
                       reinterpret_cast<long long*>(out_masks), out_extra, H, W);
    HS_LAUNCH_CHECK("augment(apply)");
    return HIPSEG_OK;
}
