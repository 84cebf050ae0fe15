// ConvTranspose2d(k2, s2) forward and data gradient for the SMALL-channel decoder stages (dec3.up, dec4.up of the
// U-Nets: /root/reference/models/processing_blocks.py:102,106 with 128 -> 64 and 64 -> 32 channels), bf16, gfx950.
//
// These layers are bandwidth-bound (50-100 MB of activations, 4 GFLOP): the GEMM kernels tiled for compute ran them at
// 18-35 us against 9-18 us of traffic (measured here: 64 -> 32 forward 26.3 -> 20.5 us, its data gradient 35.3 ->
// 19.1 us = 5.2 TB/s of activations; 128 -> 64 forward 18.2 -> 16.4 us).  Here NOTHING goes through LDS:
//   forward        y[n, 2i+a, 2j+b, co] = bias[co] + sum_ci x[n,i,j,ci] W[ci,co,a,b]     GEMM M = pixels, K = Cin, N = 4 Cout
//   data gradient  dx[n,i,j,ci] = sum_{a,b,co} dy[n,2i+a,2j+b,co] W[ci,co,a,b]           GEMM M = pixels, K = 4 Cout, N = Cin
// as v_mfma_f32_16x16x32_bf16 with SWAPPED operands (see conv3_m16.hip): the A operand = weights (rows = 16 output
// channels, the whole K of the wave's NG channel groups stays in registers for the kernel's lifetime), the B operand =
// 16 pixels x 32 K values whose lane (pixel, k octet) is 16 contiguous bytes of an NHWC pixel row: one
// buffer_load_dwordx4 straight into the MFMA operand register.  Results leave as 16-byte NHWC stores from the
// accumulators (channel permutation of the A rows as in conv3_m16.hip).  A wave streams a contiguous run of 16-pixel
// blocks; latency is hidden by occupancy (2-3 waves per SIMD, 8 KiB of loads in flight each).
#include <stdlib.h>

#include "common.h"
#include "conv_args.h"

namespace {

// KS = K / 32 MFMA steps; NG = 32-channel output groups per wave; U = pixel blocks in flight per wave
// BWS (data gradient only, hipseg_convT_dgrad_bnstats): the output is dy of relu(bn(xr)) -- the ConvTranspose2d's input is
// the previous ConvBlock's activated output (/root/reference/models/processing_blocks.py:102-109), xr = p.bw_x the
// convolution output that BatchNorm normalised -- and the kernel also reduces that BatchNorm's backward sums,
// [sum g | sum g * xhat] with g = round_bf16(out) where xr * scale + shift > 0 (bn_bwd_reduce_kernel's rule, bn.hip), one
// row per workgroup into p.stats: the reduce launch that would re-read both tensors is not needed.
template <int MODE, int KS, int NG, int U, bool BWS = false>
__global__ __launch_bounds__(256) void convt_stream_kernel(ConvArgs p, int nsets, int npb, int pbw) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef bf16 T;
    constexpr bool FWD = MODE == HIPSEG_CONVT;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const int gw = blockIdx.x * 4 + wave;  // global wave: (pixel strip, channel-group set)
    const int nset = gw % nsets, strip = gw / nsets;
    const int g0 = nset * NG;  // first 32-channel group of this wave

    const int Cin_row = p.C0;                       // channels of an input pixel row (x: Cin; dy: Cout)
    const unsigned in_bytes = (unsigned)((size_t)p.B * p.Hi * p.Wi * Cin_row * sizeof(T));
    const __amdgpu_buffer_rsrc_t r_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in0), 0, (int)in_bytes, 0x00020000);
    const int ntap = FWD ? 1 : 4;
    const __amdgpu_buffer_rsrc_t r_w = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(p.wp), 0, (int)((size_t)ntap * p.Kp * p.Np * sizeof(T)), 0x00020000);

    // ---- stationary weights: fragment (k step ks, group g, block j): rows i <-> channels 32 g + (i >> 2) * 8 + 4 j + (i & 3)
    // k step ks covers K values 32 ks .. 32 ks + 31 of the GEMM: forward = input channels; data gradient = tap
    // (32 ks) / Cout, channels (32 ks) % Cout .. + 31 of that tap
    bf16x8 Wr[KS][NG][2];
    const int kgp = p.Kp / 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        int tap = 0, koct = ks * 4;
        if (!FWD) {
            tap = (ks * 32) / p.C0;
            koct = ((ks * 32) - tap * p.C0) / 8;
        }
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int n = (g0 + g) * 32 + (li >> 2) * 8 + j * 4 + (li & 3);
                const unsigned off = (unsigned)((((size_t)tap * kgp + koct + lg) * p.Np + n) * 16);
                Wr[ks][g][j] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r_w, off, 0, 0));
            }
    }
    // per-lane epilogue constants: this lane's 8 output channels of each group
    float bv[NG][8];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int n = (g0 + g) * 32 + lg * 8 + k;
            bv[g][k] = (FWD && p.bias) ? p.bias[n % p.N0] : 0.f;
        }

    // BWS: per-lane BatchNorm vectors of the lane's 8 channels of each group, and the running sums
    float mn[BWS ? NG : 1][8], is[BWS ? NG : 1][8], s2[BWS ? NG : 1][8], sh[BWS ? NG : 1][8];
    float sg[BWS ? NG : 1][8], sgx[BWS ? NG : 1][8];
    if constexpr (BWS) {
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int n = (g0 + g) * 32 + lg * 8 + k;
                mn[g][k] = p.bw_bn[n];
                is[g][k] = p.bw_bn[p.N + n];
                s2[g][k] = p.bw_bn[2 * (size_t)p.N + n];
                sh[g][k] = p.bw_bn[3 * (size_t)p.N + n];
                sg[g][k] = 0.f;
                sgx[g][k] = 0.f;
            }
    }
    const T* rx = reinterpret_cast<const T*>(p.bw_x);

    T* out = reinterpret_cast<T*>(p.out0);
    const int wblk = p.W / 16;  // 16-pixel blocks per image row (launch condition: W % 16 == 0)
    int pb = strip * pbw;
    const int pb_end = pb + pbw < npb ? pb + pbw : npb;
    for (; pb < pb_end; pb += U) {
        bf16x8 xb[U][KS];
        bf16x8 xr[BWS ? U : 1][BWS ? NG : 1];
        int oy[U], ox[U], oimg[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int b_ = pb + u < pb_end ? pb + u : pb_end - 1;  // (tail: recompute the last block, store masked)
            const int row = b_ / wblk;                              // (img * H + y)
            oimg[u] = row / p.H;
            oy[u] = row - oimg[u] * p.H;
            ox[u] = (b_ - row * wblk) * 16 + li;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                unsigned off;
                if (FWD) {
                    off = (unsigned)(((size_t)row * p.W + ox[u]) * Cin_row * 2) + (unsigned)(ks * 64 + lg * 16);
                } else {
                    const int tap = (ks * 32) / p.C0, c = (ks * 32) - tap * p.C0;
                    const size_t ipix = ((size_t)oimg[u] * p.Hi + 2 * oy[u] + (tap >> 1)) * p.Wi + 2 * ox[u] + (tap & 1);
                    off = (unsigned)(ipix * Cin_row * 2) + (unsigned)(c * 2 + lg * 16);
                }
                xb[u][ks] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r_in, off, 0, 0));
            }
            if constexpr (BWS) {  // (tail blocks re-read the last block's pixels; their contribution is masked below)
#pragma unroll
                for (int g = 0; g < NG; ++g)
                    xr[u][g] = *reinterpret_cast<const bf16x8*>(rx + ((size_t)row * p.W + ox[u]) * p.N0 + (g0 + g) * 32 + lg * 8);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            f32x4 acc[NG][2];
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                acc[g][0] = f32x4{0.f, 0.f, 0.f, 0.f};
                acc[g][1] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    acc[g][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Wr[ks][g][0], xb[u][ks], acc[g][0], 0, 0, 0);
                    acc[g][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Wr[ks][g][1], xb[u][ks], acc[g][1], 0, 0, 0);
                }
            if (pb + u < pb_end) {
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    bf16x8 o;
#pragma unroll
                    for (int k = 0; k < 8; ++k) o[k] = (T)(acc[g][k >> 2][k & 3] + bv[g][k]);
                    const int n = (g0 + g) * 32 + lg * 8;  // first of the lane's 8 channels
                    size_t opix;
                    int co;
                    if (FWD) {  // n = ab * Cout + co -> output pixel (2y + a, 2x + b)
                        const int ab = n / p.N0;
                        co = n - ab * p.N0;
                        opix = ((size_t)oimg[u] * (2 * p.H) + 2 * oy[u] + (ab >> 1)) * (2 * p.W) + 2 * ox[u] + (ab & 1);
                    } else {
                        co = n;
                        opix = ((size_t)oimg[u] * p.H + oy[u]) * p.W + ox[u];
                    }
                    *reinterpret_cast<bf16x8*>(out + opix * p.N0 + co) = o;
                    if constexpr (BWS) {
#pragma unroll
                        for (int k = 0; k < 8; ++k) {
                            const float xv = (float)xr[u][g][k];
                            const float gq = xv * s2[g][k] + sh[g][k] > 0.f ? (float)o[k] : 0.f;
                            sg[g][k] += gq;
                            sgx[g][k] += gq * ((xv - mn[g][k]) * is[g][k]);
                        }
                    }
                }
            }
        }
    }
    if constexpr (BWS) {
        // wave: the 16 pixels of a block sit in the 16 lanes li of a channel octet lg; workgroup: the waves whose channel
        // set is the same (wave % nsets: launch condition 4 % nsets == 0) are summed in wave order through LDS
        __shared__ float sm[4][2][NG * 32];
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) {
                    sg[g][k] += __shfl_xor(sg[g][k], o, 64);
                    sgx[g][k] += __shfl_xor(sgx[g][k], o, 64);
                }
                if (li == 0) {
                    sm[wave][0][g * 32 + lg * 8 + k] = sg[g][k];
                    sm[wave][1][g * 32 + lg * 8 + k] = sgx[g][k];
                }
            }
        __syncthreads();
        const int N = p.N;  // = nsets * NG * 32
        for (int idx = threadIdx.x; idx < 2 * N; idx += 256) {
            const int arr = idx / N, c = idx - arr * N;
            const int set = c / (NG * 32), cl = c - set * (NG * 32);
            float t = 0.f;
            for (int w = set; w < 4; w += nsets) t += sm[w][arr][cl];
            p.stats[((size_t)blockIdx.x * 2 + arr) * N + c] = t;
        }
    }
#else
    (void)p;
    (void)nsets;
    (void)npb;
    (void)pbw;
#endif
}

// strips of pixel blocks and workgroups of a launch (also the row count of the BatchNorm-backward epilogue)
struct StreamGrid {
    int nsets, npb, pbw;
    long grid;
};
template <int NG, int U>
StreamGrid stream_grid(int N, int B, int H, int W, int ncu, bool bws) {
    StreamGrid g;
    const int groups = N / 32;
    g.nsets = groups / NG;
    g.npb = (int)((long)B * H * W / 16);
    // ~12 waves per CU: enough loads in flight, and a wave amortises its weight fragments over >= 8 pixel blocks
    // (BatchNorm-backward epilogue: ~240 VGPRs, two waves per SIMD = 8 per CU)
    long waves = (long)ncu * (bws ? 8 : 12);
    long strips = waves / g.nsets;
    if (strips < 1) strips = 1;
    int pbw = (int)((g.npb + strips - 1) / strips);
    if (pbw < 8) pbw = 8;
    g.pbw = (pbw + U - 1) / U * U;
    const long nstrips = (g.npb + g.pbw - 1) / g.pbw;
    g.grid = (nstrips * g.nsets + 3) / 4;
    return g;
}

template <int MODE, int KS, int NG, int U>
int launch_stream(const ConvArgs& a, hipStream_t s) {
    const bool bws = MODE == HIPSEG_CONV2S2 && a.bw_x != nullptr;
    const StreamGrid g = stream_grid<NG, U>(a.N, a.B, a.H, a.W, a.ncu, bws);
    if constexpr (MODE == HIPSEG_CONV2S2) {
        if (bws) {
            HS_REQUIRE(4 % g.nsets == 0 && a.stats && a.bw_bn, "convt_stream: BatchNorm-backward epilogue operands");
            hipLaunchKernelGGL((convt_stream_kernel<MODE, KS, NG, U, true>), dim3((unsigned)g.grid), dim3(256), 0, s, a, g.nsets,
                               g.npb, g.pbw);
            HS_LAUNCH_CHECK("convt_stream(bn sums)");
            return HIPSEG_OK;
        }
    }
    hipLaunchKernelGGL((convt_stream_kernel<MODE, KS, NG, U>), dim3((unsigned)g.grid), dim3(256), 0, s, a, g.nsets, g.npb, g.pbw);
    HS_LAUNCH_CHECK("convt_stream");
    return HIPSEG_OK;
}

}  // namespace

// 0 = the streaming kernel does not take the shape
int convt_stream_applies(int dtype, int mode, int C0, int C1, int N0, int N1, int B, int H, int W) {
    static const bool off = getenv("HIPSEG_NO_CONVT_STREAM") != nullptr || getenv("HIPSEG_NO_DMA") != nullptr;
    if (off || dtype != HIPSEG_BF16 || (mode != HIPSEG_CONVT && mode != HIPSEG_CONV2S2) || C1 || N1) return 0;
    if (W % 16 || C0 % 32 || N0 % 32) return 0;
    const int K = mode == HIPSEG_CONVT ? C0 : 4 * C0, N = mode == HIPSEG_CONVT ? 4 * N0 : N0;
    if (mode == HIPSEG_CONVT && N0 % 32) return 0;
    // small, bandwidth-bound layers only: the weights of a wave's groups must fit its registers
    // (K = 256, the 128 -> 64 stage's data gradient: one pixel block in flight per wave, measured 15.7 us against the
    // GEMM kernel's 13.5 -- left there)
    if (!(K == 64 || K == 128) || N > 256 || N < 64 || N % 64) return 0;
    const size_t in_px = mode == HIPSEG_CONVT ? (size_t)B * H * W : (size_t)B * 4 * H * W;
    if (in_px * C0 * 2 > ((size_t)1 << 30)) return 0;
    return 1;
}

// rows the BatchNorm-backward epilogue of the data-gradient launch writes (0: the shape's set count does not divide the
// four waves of a workgroup)
int convt_stream_bws_rows(int C0, int N, int B, int H, int W, int ncu) {
    const int K = 4 * C0;
    const StreamGrid g = K == 64 ? stream_grid<2, 2>(N, B, H, W, ncu, true) : stream_grid<2, 2>(N, B, H, W, ncu, true);
    return 4 % g.nsets == 0 ? (int)g.grid : 0;
}

int convt_stream_launch(const ConvArgs& a, int mode, hipStream_t s) {
    const int K = mode == HIPSEG_CONVT ? a.C0 : 4 * a.C0;
    const int groups = a.N / 32;
    if (mode == HIPSEG_CONVT) {
        if (K == 64) return groups % 4 == 0 ? launch_stream<HIPSEG_CONVT, 2, 4, 2>(a, s) : launch_stream<HIPSEG_CONVT, 2, 2, 2>(a, s);
        return launch_stream<HIPSEG_CONVT, 4, 2, 2>(a, s);
    }
    if (K == 64) return launch_stream<HIPSEG_CONV2S2, 2, 2, 2>(a, s);
    return launch_stream<HIPSEG_CONV2S2, 4, 2, 2>(a, s);
}
