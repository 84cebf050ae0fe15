// Implicit-GEMM convolution for gfx950 (MFMA), NHWC activations.
//
//   M = output pixels, tiled 16x16 per workgroup (BM = 256)
//   N = output channels, tile BN in {32, 64, 128}
//   K = taps x input channels, consumed in chunks of KC channels x ALL taps
//
// Per K-chunk the workgroup stages (a) the input HALO tile (18x18 pixels for a 3x3 conv)
// once into LDS and re-uses it for all 9 taps, and (b) the packed weights of the chunk.
// LDS images are "k-group major" ([k/G][pixel or column][G], G = 8 bf16 / 1 f32) so that an MFMA
// operand fragment is one conflict-free ds_read_b128 (bf16) / ds_read_b32 (f32) per lane.
//
//   bf16: v_mfma_f32_32x32x16_bf16, fp32 accumulate
//   f32 : v_mfma_f32_32x32x2_f32 (exact fp32 fma chain) -- the 1e-4 parity mode
//
// The epilogue adds the bias, stores NHWC (optionally split over two destination tensors, or
// pixel-shuffled for ConvTranspose2d) and reduces the per-tile BatchNorm partial sums.
//
// Kernels in this file (bf16 unless noted; dispatch in hipseg_conv_igemm):
//   conv3_wstat_kernel     3x3, <= 64 channels on both sides, >= 1024 tiles: weights in registers, persistent
//   conv3_ring64_kernel    3x3, 128-wide tiles, K % 32 == 0: activation super-chunks + register-staged weights
//   conv_igemm_dma_kernel  every other vector-aligned shape / mode (ring of 16-channel chunks)
//   gemm1_kernel           ConvTranspose2d forward / data gradient with K >= 256: one-tap GEMM, pixel-major stages
//   conv_igemm_kernel      fp32 and channel counts that are not multiples of 8 (register-staged, single buffer)
#include <stdlib.h>

#include "common.h"
#include "conv_args.h"

#ifndef IGEMM_PPT
#define IGEMM_PPT 1
#endif

// 16x16x32-MFMA kernel for the >= 128-channel 3x3 layers (conv3_m16.hip)
int conv3_m16_rows(int dtype, int mode, int C0, int C1, int N0, int N1, int B, int H, int W);
int conv3_m16_stats_rows(int rows, int N, int B, int H, int W);
int conv3_m16_launch(const ConvArgs& a, int rows, hipStream_t s);
// LDS-free streaming kernel for the small-channel ConvTranspose2d stages (convt_stream.hip)
int convt_stream_applies(int dtype, int mode, int C0, int C1, int N0, int N1, int B, int H, int W);
int convt_stream_launch(const ConvArgs& a, int mode, hipStream_t s);

namespace {

constexpr int TH = 16, TW = 16, BM = TH * TW;

template <typename T>
struct KT;
template <>
struct KT<bf16> {
    static constexpr int G = 8, KC = 16;
};
template <>
struct KT<float> {
    static constexpr int G = 1, KC = 8;
};

template <int MODE>
struct Geo;
template <>
struct Geo<HIPSEG_CONV3> {
    static constexpr int HH = TH + 2, HW = TW + 2, NT = 9;
};
template <>
struct Geo<HIPSEG_CONV1> {
    static constexpr int HH = TH, HW = TW, NT = 1;
};
template <>
struct Geo<HIPSEG_CONV2S2> {
    static constexpr int HH = 2 * TH, HW = 2 * TW, NT = 4;
};
template <>
struct Geo<HIPSEG_CONVT> {
    static constexpr int HH = TH, HW = TW, NT = 1;
};


// Wave grid of a workgroup.  The generic kernel runs 4 waves; the DMA kernel runs 8 waves (two per SIMD, so
// one wave's LDS / barrier waits hide behind the other's MFMAs) on the 256x128 tile.
template <int BN, int NW, int THT = 16>
struct WG {
    // one wave spans ALL channels of a <= 64-wide tile: its epilogue then writes whole 128-byte (64 ch) /
    // 64-byte (32 ch) pixel rows instead of half rows, and a 64x64 wave tile needs 1.0 LDS reads per MFMA
    static constexpr int WN = BN >= 128 ? 2 : 1, WM = NW / WN;
    static constexpr int MT = (THT * TW / WM) / 32, NTL = (BN / WN) / 32;
};

// M sub-tile (32 MFMA rows) -> tile pixels: rows 0-15 = 16 pixels of tile row 2s, rows 16-31 = tile row 2s+1
// ROTATED by ROT pixels, ROT = halo pitch mod 16.  With the rotation the two 16-lane halves of an operand
// read land on the same 16 LDS slots modulo 16 (conflict-free ds_read_b128 / ds_read_b32); without it a
// 3x3 halo pitch of 18 gives every read a 2-way bank conflict.
template <int MODE>
struct Rot {
    static constexpr int value = (MODE == HIPSEG_CONV3) ? (Geo<MODE>::HW % 16) : 0;
};
template <int MODE>
__device__ __forceinline__ int sub_px(int rr) {  // pixel column (0..15) of sub-tile row rr
    return (rr >> 4) ? (((rr & 15) - Rot<MODE>::value) & 15) : (rr & 15);
}

template <int MODE>
__device__ __forceinline__ constexpr int tap_off(int tap) {
    if (MODE == HIPSEG_CONV3) return (tap / 3) * Geo<MODE>::HW + (tap % 3);
    if (MODE == HIPSEG_CONV2S2) return (tap >> 1) * Geo<MODE>::HW + (tap & 1);
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// Shared epilogue.  The MFMA accumulator layout puts one output CHANNEL on each lane, so direct stores
// would be 2-byte scattered stores (a dozen times slower per byte than 16-byte ones).  Each wave
// instead turns its 32-pixel x (32*NTL)-channel sub-tile through a private LDS tile (fp32) and
// writes full 16-byte channel vectors per lane: 128 contiguous bytes per pixel for a 64-channel wave tile.
// Also adds the bias and reduces the BatchNorm partial statistics (sum, sum of squares per channel).
// BWS (bf16, plain NHWC destination, gemm1_kernel's data gradient only: hipseg_convT_dgrad_bnstats): the output is dy of
// relu(bn(xr)), xr = p.bw_x; the store loop -- where a lane holds 8 channels of a pixel -- also forms the BatchNorm-backward
// sums [sum g | sum g * xhat] of its values (g = the bf16-rounded output where xr * scale + shift > 0), summed over the
// workgroup into row `mtile` of p.stats (see conv3_m16.hip's EPI = 2 for the same epilogue on the 16x16x32 kernel).
template <typename T, int MODE, int BN, int NW, int THT = 16, bool AFF = false, bool BWS = false>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& p,
                                              f32x16 (&acc)[WG<BN, NW, THT>::MT][WG<BN, NW, THT>::NTL],
                                              unsigned char* smem, int mtile, int img, int y0, int x0, int n0) {
    constexpr int WN = WG<BN, NW, THT>::WN, WM = WG<BN, NW, THT>::WM;
    constexpr int MT = WG<BN, NW, THT>::MT, NTL = WG<BN, NW, THT>::NTL;
    (void)WM;
    (void)mtile;
    static_assert(!BWS || (MODE == HIPSEG_CONV1 && sizeof(T) == 2 && 32 * NTL == 64), "BatchNorm-backward epilogue: geometry");
    constexpr int TN = 32 * NTL;            // channels of the wave tile
    constexpr int VEC = VecOf<T>::N;        // channels per 16-byte store
    constexpr int VPR = TN / VEC;           // vectors per pixel row
    constexpr int NIT = 32 * VPR / 64;      // store iterations per sub-tile (>= 1)
    static_assert(32 * VPR % 64 == 0, "whole wave iterations");
    // bf16 read-back: a lane takes 32 B (8 floats) as two ds_read_b128, i.e. only every second 16-byte slot per
    // read, and the 16-lane groups of a b128 read span rows that alias in the 256-B bank window -> 2-way conflicts
    // (measured: SQ_LDS_BANK_CONFLICT = 1/3 of the kernel's LDS cycles, all of it here).  XOR the 16-byte slot index
    // with the parity of the row's bank window: the rows of a group then use complementary slots (conflict-free).
    constexpr bool SWZ = VEC == 8 && (TN == 64 || TN == 32);
    constexpr int RPW = SWZ ? 64 / TN : 1;  // tile rows per 256-B bank window
    auto swz_col = [](int row, int col) { return SWZ ? ((((col >> 2) ^ ((row / RPW) & 1)) << 2) | (col & 3)) : col; };
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    T* out0 = reinterpret_cast<T*>(p.out0);
    T* out1 = reinterpret_cast<T*>(p.out1);

    __syncthreads();  // every wave is done with the staging buffers: LDS becomes the transpose scratch
    float* tile = reinterpret_cast<float*>(smem) + wave * (32 * TN);

    constexpr int GPW = MT / 2;  // 64-row statistics groups per wave (a 256-row tile has 4)
    float bv[NTL], sc[NTL], ssum[GPW][NTL], ssq[GPW][NTL];
    int ncol[NTL];
    constexpr bool aff = AFF;  // fused inference epilogue (compile-time: the training kernels carry none of it)
#pragma unroll
    for (int j = 0; j < NTL; ++j) {
        ncol[j] = wn * (BN / WN) + j * 32 + r;
        const int n = n0 + ncol[j];
        int co = n;
        if (MODE == HIPSEG_CONVT) co = n % p.N0;
        bv[j] = (n < p.N && p.bias) ? p.bias[co] : 0.f;
        sc[j] = (aff && n < p.N) ? p.post_scale[co] : 1.f;
#pragma unroll
        for (int g = 0; g < GPW; ++g) {
            ssum[g][j] = 0.f;
            ssq[g][j] = 0.f;
        }
    }
    // vector path needs every 8(4)-channel group to stay inside one destination tensor / tap group
    const bool vec_ok = (p.N0 % VEC == 0) && (p.N1 % VEC == 0);
    const int nw0 = n0 + wn * (BN / WN);  // first channel of this wave's tile

    // BWS: the lane's 8 channels are the same in every store iteration (64 % VPR == 0); all reads of xr go out now
    float bmn[BWS ? 8 : 1], bis[BWS ? 8 : 1], bsc[BWS ? 8 : 1], bsh[BWS ? 8 : 1], bs1[BWS ? 8 : 1], bs2[BWS ? 8 : 1];
    typename VecOf<T>::type bxr[BWS ? MT : 1][BWS ? NIT : 1];
    if constexpr (BWS) {
        const int n = nw0 + (lane % VPR) * VEC;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int nn = n + q < p.N ? n + q : p.N - 1;
            bmn[q] = p.bw_bn[nn];
            bis[q] = p.bw_bn[p.N + nn];
            bsc[q] = p.bw_bn[2 * (size_t)p.N + nn];
            bsh[q] = p.bw_bn[3 * (size_t)p.N + nn];
            bs1[q] = 0.f;
            bs2[q] = 0.f;
        }
        const T* rx = reinterpret_cast<const T*>(p.bw_x);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int rr = (it * 64 + lane) / VPR;
                const int y = y0 + 2 * (wm * MT + i) + (rr >> 4), x = x0 + sub_px<MODE>(rr);
                const bool in = y < p.H && x < p.W && n < p.N;
                const long opix = in ? ((long)img * p.H + y) * p.W + x : 0;
                bxr[i][it] = *reinterpret_cast<const typename VecOf<T>::type*>(rx + opix * p.N + (in ? n : 0));
            }
    }

#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int yb = y0 + 2 * (wm * MT + i);
#pragma unroll
        for (int j = 0; j < NTL; ++j) {
            const bool nok = n0 + ncol[j] < p.N;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int rr = (e & 3) + 8 * (e >> 2) + 4 * h;
                float v = aff ? fmaxf(fmaf(acc[i][j][e], sc[j], bv[j]), 0.f) : acc[i][j][e] + bv[j];
                tile[rr * TN + swz_col(rr, j * 32 + r)] = v;
                if (nok && yb + (rr >> 4) < p.H && x0 + sub_px<MODE>(rr) < p.W) {
                    ssum[i / 2][j] += v;
                    ssq[i / 2][j] += v * v;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int v = it * 64 + lane;
            const int rr = v / VPR, cv = v % VPR;
            const int y = yb + (rr >> 4), x = x0 + sub_px<MODE>(rr);
            const int n = nw0 + cv * VEC;
            if (y < p.H && x < p.W && n < p.N && !(p.debug & 16)) {
                const float* src = tile + rr * TN + cv * VEC;
                if (vec_ok) {
                    typename VecOf<T>::type o;
                    if (SWZ) {  // two 16-byte reads, their slots swapped on swizzled rows
                        const f32x4 lo = *reinterpret_cast<const f32x4*>(tile + rr * TN + swz_col(rr, cv * VEC));
                        const f32x4 hi = *reinterpret_cast<const f32x4*>(tile + rr * TN + swz_col(rr, cv * VEC + 4));
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            o[q] = (T)lo[q];
                            o[4 + q] = (T)hi[q];
                        }
                    } else {
#pragma unroll
                        for (int q = 0; q < VEC; ++q) o[q] = (T)src[q];
                    }
                    T* dst;
                    if (MODE == HIPSEG_CONVT) {
                        const int ab = n / p.N0, co = n - ab * p.N0;
                        const long opix = ((long)img * (2 * p.H) + 2 * y + (ab >> 1)) * (2 * p.W) + 2 * x + (ab & 1);
                        dst = out0 + opix * p.N0 + co;
                    } else {
                        const long opix = ((long)img * p.H + y) * p.W + x;
                        dst = (n < p.N0) ? out0 + opix * p.N0 + n : out1 + opix * p.N1 + (n - p.N0);
                    }
                    *reinterpret_cast<typename VecOf<T>::type*>(dst) = o;
                    if constexpr (BWS) {
#pragma unroll
                        for (int q = 0; q < 8; ++q) {
                            const float xv = (float)bxr[i][it][q];
                            const float gq = xv * bsc[q] + bsh[q] > 0.f ? (float)o[q] : 0.f;
                            bs1[q] += gq;
                            bs2[q] += gq * ((xv - bmn[q]) * bis[q]);
                        }
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < VEC; ++q) {
                        const int nn = n + q;
                        if (nn < p.N) {
                            if (MODE == HIPSEG_CONVT) {
                                const int ab = nn / p.N0, co = nn - ab * p.N0;
                                const long opix =
                                    ((long)img * (2 * p.H) + 2 * y + (ab >> 1)) * (2 * p.W) + 2 * x + (ab & 1);
                                out0[opix * p.N0 + co] = (T)tile[rr * TN + swz_col(rr, cv * VEC + q)];
                            } else {
                                const long opix = ((long)img * p.H + y) * p.W + x;
                                if (nn < p.N0)
                                    out0[opix * p.N0 + nn] = (T)tile[rr * TN + swz_col(rr, cv * VEC + q)];
                                else
                                    out1[opix * p.N1 + (nn - p.N0)] = (T)tile[rr * TN + swz_col(rr, cv * VEC + q)];
                            }
                        }
                    }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if constexpr (BWS) {
        // lanes with the same channel vector (lane % 8), then the WM waves of a channel half through LDS (behind the
        // waves' transpose tiles), in wave order
        float* red = reinterpret_cast<float*>(smem) + NW * (32 * TN);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
#pragma unroll
            for (int o = 8; o < 64; o <<= 1) {
                bs1[q] += __shfl_xor(bs1[q], o, 64);
                bs2[q] += __shfl_xor(bs2[q], o, 64);
            }
            if (lane < 8) {
                red[(wave * 2 + 0) * 64 + lane * 8 + q] = bs1[q];
                red[(wave * 2 + 1) * 64 + lane * 8 + q] = bs2[q];
            }
        }
        __syncthreads();
        for (int idx = tid; idx < 2 * WN * 64; idx += NW * 64) {
            const int arr = idx / (WN * 64), c = idx - arr * (WN * 64), wn_ = c / 64, cc = c - wn_ * 64;
            float t = 0.f;
#pragma unroll
            for (int m = 0; m < WM; ++m) t += red[((m * WN + wn_) * 2 + arr) * 64 + cc];
            const int n = n0 + wn_ * (BN / WN) + cc;
            if (n < p.N) p.stats[((size_t)mtile * 2 + arr) * p.N + n] = t;
        }
        return;
    }
    if (p.stats && !(p.debug & 32)) {
        // one statistics row per 64-row group (4 per tile): no cross-wave reduction, the finalize kernel sums rows
#pragma unroll
        for (int g = 0; g < GPW; ++g)
#pragma unroll
            for (int j = 0; j < NTL; ++j) {
                const float S = ssum[g][j] + __shfl_xor(ssum[g][j], 32, 64);
                const float Q = ssq[g][j] + __shfl_xor(ssq[g][j], 32, 64);
                const int n = n0 + ncol[j];
                // statistics rows follow the 16-row tile grid (4 rows of 64 pixels per 16x16 tile)
                const int G = wm * GPW + g, ty16 = y0 / 16 + G / 4, tiles_y16 = (p.H + 15) / 16;
                if (h == 0 && n < p.N && ty16 < tiles_y16) {
                    const size_t row = (((size_t)img * tiles_y16 + ty16) * p.tiles_x + x0 / 16) * 4 + (G & 3);
                    p.stats[(row * 2 + 0) * p.N + n] = S;
                    p.stats[(row * 2 + 1) * p.N + n] = Q;
                }
            }
    }
}

template <typename T, int MODE, int BN, bool AFF = false>
__global__ __launch_bounds__(256, 2) void conv_igemm_kernel(ConvArgs p) {
    constexpr int G = KT<T>::G, KC = KT<T>::KC, KG = KC / G;
    constexpr int HH = Geo<MODE>::HH, HW = Geo<MODE>::HW, NT = Geo<MODE>::NT, NPIX = HH * HW;
    constexpr int WN = WG<BN, 4>::WN, WM = WG<BN, 4>::WM;
    constexpr int MT = WG<BN, 4>::MT, NTL = WG<BN, 4>::NTL;
    constexpr int VEC = VecOf<T>::N;
    constexpr int CV = KC / VEC;
    typedef typename VecOf<T>::type vec_t;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T* sA = reinterpret_cast<T*>(smem);  // [KG][NPIX][G]
    T* sB = sA + KC * NPIX;              // [NT][KG][BN][G]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;

    const int bid = xcd_block(blockIdx.x, p.xcd);
    const int ntile = bid % p.ntn;
    const int mtile = bid / p.ntn;
    const int tx = mtile % p.tiles_x;
    const int ty = (mtile / p.tiles_x) % p.tiles_y;
    const int img = mtile / (p.tiles_x * p.tiles_y);
    const int y0 = ty * TH, x0 = tx * TW;
    const int n0 = ntile * BN;

    // halo origin in input coordinates
    int oy, ox;
    if (MODE == HIPSEG_CONV3) {
        oy = y0 - 1;
        ox = x0 - 1;
    } else if (MODE == HIPSEG_CONV2S2) {
        oy = 2 * y0;
        ox = 2 * x0;
    } else {
        oy = y0;
        ox = x0;
    }

    const T* in0 = reinterpret_cast<const T*>(p.in0);
    const T* in1 = reinterpret_cast<const T*>(p.in1);
    const T* wp = reinterpret_cast<const T*>(p.wp);

    f32x16 acc[MT][NTL];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTL; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // per-lane halo base index of each M sub-tile (32 pixels = 2 tile rows x 16)
    int hbase[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int py = 2 * (wm * MT + i) + (r >> 4), px = sub_px<MODE>(r);
        if (MODE == HIPSEG_CONV2S2)
            hbase[i] = (2 * py) * HW + 2 * px;
        else
            hbase[i] = py * HW + px;
    }
    int ncol[NTL];
#pragma unroll
    for (int j = 0; j < NTL; ++j) ncol[j] = wn * (BN / WN) + j * 32 + r;

    const int kgp = p.Kp / G;

    for (int c0 = 0; c0 < p.Kp; c0 += KC) {
        __syncthreads();  // previous chunk's fragment reads are done
        // ---------------- stage A: halo tile, KC channels
        constexpr int NCELL = NPIX * CV;
#pragma unroll
        for (int it = 0; it < (NCELL + 255) / 256; ++it) {
            const int cell = it * 256 + tid;
            if (cell < NCELL) {
                const int pix = cell / CV, cv = cell % CV;
                const int hy = pix / HW, hx = pix % HW;
                const int iy = oy + hy, ix = ox + hx;
                const int c = c0 + cv * VEC;
                vec_t v;
#pragma unroll
                for (int e = 0; e < VEC; ++e) v[e] = (T)0.f;
                if (iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi && c < p.K) {
                    const long pixoff = ((long)img * p.Hi + iy) * p.Wi + ix;
                    if (p.vec_ok) {
                        const T* src = (c < p.C0) ? in0 + pixoff * p.C0 + c : in1 + pixoff * p.C1 + (c - p.C0);
                        v = *reinterpret_cast<const vec_t*>(src);
                    } else {
#pragma unroll
                        for (int e = 0; e < VEC; ++e) {
                            const int cc = c + e;
                            if (cc < p.K)
                                v[e] = (cc < p.C0) ? in0[pixoff * p.C0 + cc] : in1[pixoff * p.C1 + (cc - p.C0)];
                        }
                    }
                }
                if (G == VEC) {
                    *reinterpret_cast<vec_t*>(sA + ((size_t)cv * NPIX + pix) * G) = v;
                } else {  // G == 1: channel-major scatter
#pragma unroll
                    for (int e = 0; e < VEC; ++e) sA[(cv * VEC + e) * NPIX + pix] = v[e];
                }
            }
        }
        // ---------------- stage B: packed weights of this chunk, all taps (16-byte copies)
        {
            constexpr int RV = BN * G * (int)sizeof(T) / 16;  // 16B vectors per (tap, kg) run
            constexpr int NV = NT * KG * RV;
            const int kg0 = c0 / G;
#pragma unroll
            for (int it = 0; it < (NV + 255) / 256; ++it) {
                const int v = it * 256 + tid;
                if (v < NV) {
                    const int run = v / RV, off = v % RV;
                    const int tap = run / KG, kgl = run % KG;
                    const T* src = wp + (((size_t)tap * kgp + kg0 + kgl) * p.Np + n0) * G;
                    const uint4 val = reinterpret_cast<const uint4*>(src)[off];
                    reinterpret_cast<uint4*>(sB + (size_t)run * BN * G)[off] = val;
                }
            }
        }
        __syncthreads();
        // ---------------- MFMA over all taps of the chunk
#pragma unroll
        for (int tap = 0; tap < NT; ++tap) {
            constexpr int dummy = 0;
            (void)dummy;
            const int toff = tap_off<MODE>(tap);
            if constexpr (sizeof(T) == 2) {
                bf16x8 bf[NTL];
#pragma unroll
                for (int j = 0; j < NTL; ++j)
                    bf[j] = *reinterpret_cast<const bf16x8*>(sB + ((size_t)(tap * KG + h) * BN + ncol[j]) * 8);
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const bf16x8 af =
                        *reinterpret_cast<const bf16x8*>(sA + ((size_t)h * NPIX + hbase[i] + toff) * 8);
#pragma unroll
                    for (int j = 0; j < NTL; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf[j], acc[i][j], 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int kk = 0; kk < KC / 2; ++kk) {
                    const int k = 2 * kk + h;
                    float bfv[NTL];
#pragma unroll
                    for (int j = 0; j < NTL; ++j) bfv[j] = sB[(size_t)(tap * KC + k) * BN + ncol[j]];
#pragma unroll
                    for (int i = 0; i < MT; ++i) {
                        const float af = sA[(size_t)k * NPIX + hbase[i] + toff];
#pragma unroll
                        for (int j = 0; j < NTL; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bfv[j], acc[i][j], 0, 0, 0);
                    }
                }
            }
        }
    }

    conv_epilogue<T, MODE, BN, 4, 16, AFF>(p, acc, smem, mtile, img, y0, x0, n0);
}

// ---------------------------------------------------------------------------------------------------
// bf16 fast path: the same tiling, but both LDS images are filled by LDS-DMA (global_load_lds_dwordx4,
// no VGPR round trip, no ds_write) into a DOUBLE-BUFFERED LDS ring: the DMA of chunk k+1 is in flight
// while the MFMAs of chunk k run; one barrier per chunk (the "2-phase" schedule).  An LDS-DMA writes
// wave-uniform-base + lane*16, which is exactly the [k-octet][pixel][8] / [tap][k-octet][n][8] images;
// the per-lane SOURCE address does the halo gather, padding pixels read a 16-byte zero word instead.
// Needs 16-byte aligned channel vectors (C0 % 8 == 0, C1 % 8 == 0).
__device__ uint4 g_zero16 = {0u, 0u, 0u, 0u};

template <int BN, int THT = 16>
struct DmaWaves {
    static constexpr int value = BN == 128 ? 8 : 4;
};
// LDS ring depth: 3 buffers (chunk k+2 streams in while chunk k computes: a full chunk of MFMA time, ~1 us,
// is not always enough to cover an L2/HBM round trip) where one workgroup per CU owns the LDS anyway.
template <int BN, int THT = 16>
struct DmaBufs {
    static constexpr int value = (BN == 128 && THT == 16) ? 3 : 2;
};

// THT = tile rows: 16 (256-pixel tile) or 32 (512-pixel "tall" tile: every wave owns a 128-row x 64-channel block,
// 0.75 LDS fragment reads per MFMA instead of 1.0, and the weight chunk is amortised over twice the pixels --
// the main loop is LDS-bandwidth bound: fragment reads + DMA writes share the LDS port with nothing to spare).
template <int MODE, int BN, int THT = 16, bool AFF = false>
__global__ __launch_bounds__((64 * DmaWaves<BN, THT>::value), (DmaWaves<BN, THT>::value == 8 ? 1 : 2)) void conv_igemm_dma_kernel(ConvArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)  // buffer-resource builtins exist in the device pass only
    constexpr int NW = DmaWaves<BN, THT>::value, NBUF = DmaBufs<BN, THT>::value, DIST = NBUF - 1;
    typedef bf16 T;
    constexpr int KC = 16, KG = 2;
    constexpr int HW = Geo<MODE>::HW, NT = Geo<MODE>::NT;
    constexpr int HH = MODE == HIPSEG_CONV3 ? THT + 2 : (MODE == HIPSEG_CONV2S2 ? 2 * THT : THT), NPIX = HH * HW;
    constexpr int NPIXP = (NPIX + 63) / 64 * 64;
    constexpr int WN = WG<BN, NW, THT>::WN, WM = WG<BN, NW, THT>::WM;
    constexpr int MT = WG<BN, NW, THT>::MT, NTL = WG<BN, NW, THT>::NTL;
    (void)WM;
    constexpr int A_BYTES = KG * NPIXP * 16, B_BYTES = NT * KG * BN * 16, BUF = A_BYTES + B_BYTES;
    constexpr int NA = A_BYTES / 1024, NB = B_BYTES / 1024;            // 1-KiB DMA pieces per chunk
    constexpr int NAW = (NA + NW - 1) / NW, NBW = (NB + NW - 1) / NW;  // per wave
    constexpr int RUNB = BN * 16;                            // bytes of one (tap, k-octet) weight run
    static_assert(B_BYTES % 1024 == 0 && A_BYTES % 1024 == 0, "whole DMA pieces");
    typedef __attribute__((address_space(3))) void lds_void;
    typedef const __attribute__((address_space(1))) void glb_void;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;

    const int bid = xcd_block(blockIdx.x, p.xcd);
    const int ntile = bid % p.ntn;
    const int mtile = bid / p.ntn;
    const int tx = mtile % p.tiles_x;
    const int ty = (mtile / p.tiles_x) % p.tiles_y;
    const int img = mtile / (p.tiles_x * p.tiles_y);
    const int y0 = ty * THT, x0 = tx * TW;
    const int n0 = ntile * BN;
    int oy, ox;
    if (MODE == HIPSEG_CONV3) {
        oy = y0 - 1;
        ox = x0 - 1;
    } else if (MODE == HIPSEG_CONV2S2) {
        oy = 2 * y0;
        ox = 2 * x0;
    } else {
        oy = y0;
        ox = x0;
    }
    const T* in0 = reinterpret_cast<const T*>(p.in0);
    const T* in1 = reinterpret_cast<const T*>(p.in1);
    const unsigned char* wp = reinterpret_cast<const unsigned char*>(p.wp);
    const T* zero = reinterpret_cast<const T*>(&g_zero16);

    // ---- LDS-DMA through BUFFER addressing (buffer_load_dwordx4 ... lds): a piece's per-lane byte offset never changes
    // from chunk to chunk (it is computed once, below), the chunk only moves a scalar offset, and a lane that must read
    // zeros (image border, padding pixel) carries an out-of-range offset -- the hardware bounds check writes 0 to LDS.
    // A piece therefore costs a handful of SALU instructions and NO vector ALU work: the tap loop was issue-bound
    // (per tap ~40 address / branch instructions against 4 MFMAs per wave), not MFMA- or LDS-bound.
    // Offsets: valid < 2^30 (launch condition: tensors <= 1 GiB), lane out of range = 2^31, chunk out of range = +2^30.
    constexpr unsigned OOB_LANE = 0x80000000u, OOB_CHUNK = 0x40000000u;
    const unsigned bytes0 = (unsigned)((size_t)p.B * p.Hi * p.Wi * p.C0 * sizeof(T));
    const unsigned bytes1 = (unsigned)((size_t)p.B * p.Hi * p.Wi * p.C1 * sizeof(T));
    const __amdgpu_buffer_rsrc_t r_in0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in0), 0, (int)bytes0, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_in1 =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in1 ? p.in1 : p.in0), 0, (int)bytes1, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_w = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(p.wp), 0, (int)((size_t)NT * p.Kp * p.Np * sizeof(T)), 0x00020000);
    (void)in0;
    (void)in1;
    (void)wp;
    (void)zero;

    // per-lane byte offset of the lane's source pixel for each of this wave's A pieces (piece = 64 consecutive halo
    // pixels of one k-octet), against either source tensor
    unsigned avo0[NAW], avo1[NAW];
    int aoct[NAW];
#pragma unroll
    for (int j = 0; j < NAW; ++j) {
        const int s = j * NW + wave;
        aoct[j] = s / (NPIXP / 64);
        const int pix = (s % (NPIXP / 64)) * 64 + lane;
        const int iy = oy + pix / HW, ix = ox + pix % HW;
        const bool ok = s < NA && pix < NPIX && iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi;
        const long apix = ((long)img * p.Hi + iy) * p.Wi + ix;
        avo0[j] = ok ? (unsigned)(apix * p.C0 * (long)sizeof(T)) : OOB_LANE;
        avo1[j] = ok ? (unsigned)(apix * p.C1 * (long)sizeof(T)) : OOB_LANE;
    }
    const int kgp = p.Kp / 8;
    // per-lane byte offset into the packed weights for each B piece (everything but the chunk's k-octet term)
    unsigned bvo[NBW];
#pragma unroll
    for (int j = 0; j < NBW; ++j) {
        const int s = j * NW + wave;
        const int bb = s * 1024 + lane * 16;
        const int run = bb / RUNB, off = bb % RUNB;
        const int tap = run / KG, kgl = run % KG;
        bvo[j] = s < NB ? (unsigned)((((size_t)tap * kgp + kgl) * p.Np + n0) * 16 + off) : OOB_LANE;
    }

    // The two operands take DIFFERENT roads into LDS.  LDS-DMA lands only ~12 B/clk per CU; with both operands on
    // it a 16-channel chunk (12-20 KiB of activations + 36 KiB of weights for a 128-wide tile) takes ~4000 cycles
    // against 2304 cycles of MFMA work (measured: either operand alone costs +3-4 us on a 55-us kernel, both +27 us).
    // So only the activations (per-lane halo gather, zero fill) use LDS-DMA; the weight chunk -- contiguous, L2
    // resident, shared by every workgroup -- is read into VGPRs (buffer_load_dwordx4, one chunk ahead, at the top
    // of the chunk) and written to its ring slot with ds_write_b128 after the chunk's MFMAs.
    typedef int v4i __attribute__((ext_vector_type(4)));
    v4i bst[NBW];
    auto loadB = [&](int c0) {
        const unsigned so = !(p.debug & 2) ? (unsigned)(c0 / 8) * (unsigned)p.Np * 16u : OOB_CHUNK;
#pragma unroll
        for (int j = 0; j < NBW; ++j) bst[j] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(r_w, bvo[j], so, 0));
    };
    auto writeB = [&](int buf) {
        unsigned char* base = smem + buf * BUF + A_BYTES;
#pragma unroll
        for (int j = 0; j < NBW; ++j) {
            const int s = j * NW + wave;
            if ((j + 1) * NW <= NB || s < NB) *reinterpret_cast<v4i*>(base + s * 1024 + lane * 16) = bst[j];
        }
    };
    // one 1-KiB DMA piece (64 halo pixels of one k-octet) of chunk c0 into ring buffer `buf`, issued one at a time
    // BETWEEN MFMA groups; every wave issues NAW per chunk (pad pieces go to the sink) so waits can be counted
    auto issue_piece = [&](int buf, int c0, int idx) {
        unsigned char* base = smem + buf * BUF;
        unsigned char* dummy = smem + NBUF * BUF;  // 1-KiB sink for the pad pieces of waves with fewer real ones
        const int s = idx * NW + wave;
        const bool real = s < NA;
        const int c = c0 + aoct[idx] * 8;  // wave-uniform: first channel of the piece's octet
        lds_void* dst = (lds_void*)(real ? base + s * 1024 : dummy);
        if (c < p.C0 || p.C1 == 0) {
            const unsigned so = (c < p.K && !(p.debug & 1)) ? (unsigned)c * (unsigned)sizeof(T) : OOB_CHUNK;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r_in0, dst, 16, avo0[idx], so, 0, 0);
        } else {
            const unsigned so = (c < p.K && !(p.debug & 1)) ? (unsigned)(c - p.C0) * (unsigned)sizeof(T) : OOB_CHUNK;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r_in1, dst, 16, avo1[idx], so, 0, 0);
        }
    };
    constexpr int NPCW = NAW;                       // LDS-DMA pieces per wave per chunk (activations only)
    // pieces issued per tap: at least IGEMM_PPT, so that a chunk's pieces all go out in its FIRST taps and have the
    // rest of the chunk's MFMA time to land (spread evenly, the last piece is issued just before the barrier that
    // waits for it)
    constexpr int PPT = (NPCW + NT - 1) / NT > IGEMM_PPT || NT == 1 ? (NPCW + NT - 1) / NT : IGEMM_PPT;

    f32x16 acc[MT][NTL];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTL; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    int hbase[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int py = 2 * (wm * MT + i) + (r >> 4), px = sub_px<MODE>(r);
        hbase[i] = MODE == HIPSEG_CONV2S2 ? (2 * py) * HW + 2 * px : py * HW + px;
    }
    int ncol[NTL];
#pragma unroll
    for (int j = 0; j < NTL; ++j) ncol[j] = wn * (BN / WN) + j * 32 + r;

    const int nchunks = p.Kp / KC;
    // prologue: the activation pieces of the first DIST chunks go out at once; weight chunk 0 through registers
#pragma unroll
    for (int d = 0; d < DIST; ++d)
        if (d < nchunks) {
#pragma unroll
            for (int q = 0; q < NPCW; ++q) issue_piece(d, d * KC, q);
        }
    loadB(0);
    writeB(0);
    int cur = 0;  // ring slot of chunk kc
    for (int kc = 0; kc < nchunks; ++kc) {
        // every wave issues exactly NPCW pieces per chunk, so "chunk kc has landed" is a COUNTED wait that
        // leaves the younger chunks' pieces in flight across the barrier (raw s_barrier: no vmcnt(0) drain)
        if (kc + DIST - 1 < nchunks)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DIST - 1) * NPCW) : "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's weight-chunk ds_writes
        __builtin_amdgcn_s_barrier();  // everyone's pieces landed; everyone left the slot chunk kc+DIST will fill
        __builtin_amdgcn_sched_barrier(0);
        const bool more = kc + DIST < nchunks;
        const bool moreB = kc + 1 < nchunks;
        if (moreB) loadB((kc + 1) * KC);  // next chunk's weights -> VGPRs, written to LDS after this chunk's MFMAs
        const int nbuf = (cur + DIST) % NBUF, nc0 = (kc + DIST) * KC;
        const unsigned char* sA = smem + cur * BUF;
        const unsigned char* sB = sA + A_BYTES;
        cur = (cur + 1) % NBUF;
        // fragments of tap t+1 are fetched from LDS while the MFMAs of tap t issue (register double buffer):
        // with one wave per SIMD nothing else hides the ds_read latency
        bf16x8 bf[2][NTL], af[2][MT];
        auto fetch = [&](int slot, int tap) {
            const int toff = tap_off<MODE>(tap);
#pragma unroll
            for (int j = 0; j < NTL; ++j)
                bf[slot][j] = *reinterpret_cast<const bf16x8*>(sB + ((size_t)(tap * KG + h) * BN + ncol[j]) * 16);
#pragma unroll
            for (int i = 0; i < MT; ++i)
                af[slot][i] = *reinterpret_cast<const bf16x8*>(sA + ((size_t)h * NPIXP + hbase[i] + toff) * 16);
        };
        if (p.debug & 4) {
            if (more) {
#pragma unroll
                for (int q = 0; q < NPCW; ++q) issue_piece(nbuf, nc0, q);
            }
            if (moreB) writeB(cur);
            continue;
        }
        fetch(0, 0);
#pragma unroll
        for (int tap = 0; tap < NT; ++tap) {
            // first half of this tap's MFMAs, then issue the reads of the next tap, then the second half:
            // the reads get >= MT*NTL/2 MFMA slots (32 cycles each) to land before they are consumed
            constexpr int HALF = (MT + 1) / 2;
#pragma unroll
            for (int i = 0; i < HALF; ++i)
#pragma unroll
                for (int j = 0; j < NTL; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[tap & 1][i], bf[tap & 1][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (tap + 1 < NT) fetch((tap + 1) & 1, tap + 1);
            if (more) {
#pragma unroll
                for (int q = tap * PPT; q < (tap + 1) * PPT && q < NPCW; ++q) issue_piece(nbuf, nc0, q);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = HALF; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NTL; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[tap & 1][i], bf[tap & 1][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (moreB) writeB(cur);  // `cur` already names the next chunk's slot (free since the barrier above)
    }

    if (!(p.debug & 8)) conv_epilogue<bf16, MODE, BN, NW, THT, AFF>(p, acc, smem, mtile, img, y0, x0, n0);
#else
    (void)p;
#endif
}

// ---------------------------------------------------------------------------------------------------
// One-tap GEMM kernel for the ConvTranspose2d(k2, s2) pair: forward (MODE CONVT: M = input pixels, K = Cin,
// N = 4*Cout, pixel-shuffle store) and data gradient (MODE CONV2S2 seen as ONE tap over K = 4 taps x Cout gathered
// channels, N = Cin).  The generic ring kernel stages 16 channels per barrier as an [octet][pixel][8] GATHER (64 cache
// lines touched per 1-KiB LDS-DMA piece) -- fine when 9 taps reuse the chunk, but with one tap that is 4 MFMAs per wave
// per barrier against ~8 KiB of gather-rate DMA: 0.04-0.10 of the MFMA peak (rocprof r01).  Without a halo the
// activation tile can instead keep the global pixel-major order: a stage is 256 pixels x 64 channels, every 1-KiB piece
// = 8 pixels x 128 B = 8 whole cache lines, 16 MFMAs per wave per barrier.  The 16-byte slots of a pixel row are
// XOR-swizzled (on the DMA's SOURCE side, LDS destinations stay linear) so the 16 pixel rows a ds_read_b128 half-wave
// touches fall on 16 different (bank half, slot) pairs: conflict-free fragment reads out of 128-byte rows.
// Weights: [k-octet][n][8] packed operand, chunk read into VGPRs one stage ahead and written with ds_write_b128, as in
// the ring kernels.  Needs C0 % 64 == 0 and N % 128 == 0 (every ConvT of the U-Nets but dec4's data gradient).
template <int MODE, bool BWS = false>
__global__ __launch_bounds__(512, 1) void gemm1_kernel(ConvArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef bf16 T;
    constexpr int BN = 128, NW = 8, THT = 16, NBUF = 3, DIST = NBUF - 1;
    constexpr int KCH = 64, KS = KCH / 16;  // channels per stage, k16 MFMA steps per stage
    constexpr int A_BYTES = 256 * KCH * 2, B_BYTES = (KCH / 8) * BN * 16, BUF = A_BYTES + B_BYTES;
    constexpr int NAW = A_BYTES / 1024 / NW, NBW = B_BYTES / 1024 / NW;  // 1-KiB pieces per wave and stage: 4, 2
    constexpr int EMODE = MODE == HIPSEG_CONVT ? HIPSEG_CONVT : HIPSEG_CONV1;  // epilogue: pixel shuffle / plain NHWC
    constexpr int WN = WG<BN, NW, THT>::WN, MT = WG<BN, NW, THT>::MT, NTL = WG<BN, NW, THT>::NTL;
    static_assert(NAW == 4 && NBW == 2 && MT == 2 && NTL == 2, "geometry");
    typedef __attribute__((address_space(3))) void lds_void;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const int bid = xcd_block(blockIdx.x, p.xcd);
    const int ntile = bid % p.ntn, mtile = bid / p.ntn;
    const int tx = mtile % p.tiles_x, ty = (mtile / p.tiles_x) % p.tiles_y, img = mtile / (p.tiles_x * p.tiles_y);
    const int y0 = ty * THT, x0 = tx * TW, n0 = ntile * BN;
    const int Ktot = MODE == HIPSEG_CONVT ? p.C0 : 4 * p.C0;

    constexpr unsigned OOB_LANE = 0x80000000u;
    const unsigned in_bytes = (unsigned)((size_t)p.B * p.Hi * p.Wi * p.C0 * sizeof(T));
    const __amdgpu_buffer_rsrc_t r_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in0), 0, (int)in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_w = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(p.wp), 0, (int)((size_t)Ktot * p.Np * sizeof(T)), 0x00020000);

    auto swz = [](int pix) { return (pix & 7) ^ ((pix >> 3) & 1); };
    // A pieces of this wave: piece s = j * NW + wave holds tile pixels 8s .. 8s+7 (half a tile row); lane -> (pixel, slot)
    unsigned avo[NAW];
#pragma unroll
    for (int j = 0; j < NAW; ++j) {
        const int sidx = j * NW + wave;
        const int pix = sidx * 8 + (lane >> 3);
        const int gy = y0 + (pix >> 4), gx = x0 + (pix & 15);
        const bool ok = gy < p.H && gx < p.W;
        const long ipix = MODE == HIPSEG_CONVT ? ((long)img * p.Hi + gy) * p.Wi + gx : ((long)img * p.Hi + 2 * gy) * p.Wi + 2 * gx;
        avo[j] = ok ? (unsigned)(ipix * p.C0 * (long)sizeof(T)) + (unsigned)(((lane & 7) ^ swz(pix)) * 16) : OOB_LANE;
    }
    // B pieces: piece s covers k-octet s/2, columns (s&1)*64 + lane of the 128-wide tile
    unsigned bvo[NBW];
#pragma unroll
    for (int j = 0; j < NBW; ++j) {
        const int sidx = j * NW + wave;
        bvo[j] = (unsigned)((((size_t)(sidx >> 1) * p.Np) + n0 + (sidx & 1) * 64 + lane) * 16);
    }
    typedef int v4i __attribute__((ext_vector_type(4)));
    v4i bst[NBW];
    auto loadB = [&](int c) {
        const unsigned so = (unsigned)(c * (KCH / 8)) * (unsigned)p.Np * 16u;
#pragma unroll
        for (int j = 0; j < NBW; ++j) bst[j] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(r_w, bvo[j], so, 0));
    };
    auto writeB = [&](int buf) {
        unsigned char* base = smem + buf * BUF + A_BYTES;
#pragma unroll
        for (int j = 0; j < NBW; ++j) *reinterpret_cast<v4i*>(base + (j * NW + wave) * 1024 + lane * 16) = bst[j];
    };
    auto issue_piece = [&](int buf, int c, int j) {
        const int k0 = c * KCH;
        unsigned so;
        if (MODE == HIPSEG_CONVT) {
            so = (unsigned)k0 * (unsigned)sizeof(T);
        } else {  // chunk = 64 channels of tap (a, b): input pixel (2y + a, 2x + b)
            const int tap = k0 / p.C0, co0 = k0 - tap * p.C0;
            so = (unsigned)((((tap >> 1) * p.Wi + (tap & 1)) * p.C0 + co0) * (int)sizeof(T));
        }
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r_in, (lds_void*)(smem + buf * BUF + (j * NW + wave) * 1024), 16, avo[j], so, 0,
                                                 0);
    };

    f32x16 acc[MT][NTL];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTL; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    int apix[MT], asw[MT], ncol[NTL];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        apix[i] = 32 * (wm * MT + i) + r;  // tile pixel of MFMA row r of sub-tile i (two 16-pixel tile rows)
        asw[i] = swz(apix[i]);
    }
#pragma unroll
    for (int j = 0; j < NTL; ++j) ncol[j] = wn * (BN / WN) + j * 32 + r;

    const int nchunks = Ktot / KCH;
#pragma unroll
    for (int d = 0; d < DIST; ++d)
        if (d < nchunks) {
#pragma unroll
            for (int q = 0; q < NAW; ++q) issue_piece(d, d, q);
        }
    loadB(0);
    writeB(0);
    int cur = 0;
    for (int kc = 0; kc < nchunks; ++kc) {
        if (kc + DIST - 1 < nchunks)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DIST - 1) * NAW) : "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const bool more = kc + DIST < nchunks, moreB = kc + 1 < nchunks;
        if (moreB) loadB(kc + 1);
        const int nbuf = (cur + DIST) % NBUF;
        const unsigned char* sA = smem + cur * BUF;
        const unsigned char* sB = sA + A_BYTES;
        cur = (cur + 1) % NBUF;
        bf16x8 bf[2][NTL], af[2][MT];
        auto fetch = [&](int slot, int t) {
            const int oct = 2 * t + h;
#pragma unroll
            for (int j = 0; j < NTL; ++j) bf[slot][j] = *reinterpret_cast<const bf16x8*>(sB + ((size_t)oct * BN + ncol[j]) * 16);
#pragma unroll
            for (int i = 0; i < MT; ++i)
                af[slot][i] = *reinterpret_cast<const bf16x8*>(sA + (size_t)apix[i] * (KCH * 2) + ((oct ^ asw[i]) * 16));
        };
        fetch(0, 0);
#pragma unroll
        for (int t = 0; t < KS; ++t) {
#pragma unroll
            for (int j = 0; j < NTL; ++j)
                acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[t & 1][0], bf[t & 1][j], acc[0][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (t + 1 < KS) fetch((t + 1) & 1, t + 1);
            if (more) issue_piece(nbuf, kc + DIST, t);  // NAW == KS: one piece per k16 step
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < NTL; ++j)
                acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[t & 1][1], bf[t & 1][j], acc[1][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (moreB) writeB(cur);
    }
    static_assert(NAW == KS, "one activation piece per k16 step");
    conv_epilogue<bf16, EMODE, BN, NW, THT, false, BWS>(p, acc, smem, mtile, img, y0, x0, n0);
#else
    (void)p;
#endif
}

// ---------------------------------------------------------------------------------------------------
// 3x3 ring kernel with ACTIVATION SUPER-CHUNKS (128-wide tile, >= 128 input channels).  The plain ring kernel gathers
// 16 B (one k-octet) per halo pixel per DMA lane and comes back for the neighbouring octets of the same cache line
// one, two and three 16-channel chunks later -- long after the line left the 32-KiB L1 -- so every activation line is
// requested from the L2 two to eight times, 64 requests per DMA instruction.  With the weight chunk's traffic on top
// the memory system, not the MFMA pipe, sets the chunk time (measured on 256->256 @64^2: either operand alone costs
// +3-4 us on a 55-us kernel, both together +27 us).  Here the halo tile is staged SO octets (SO * 8 channels = 64 or
// 128 B per pixel) at a time into its own 2-slot LDS ring: the SO DMA instructions that touch a line are issued back
// to back by the waves of the workgroup, one L2 request serves them all, and the K loop walks SO / 2 chunks inside a
// resident super-chunk.  Weights: one 16-channel chunk ahead through registers (see conv_igemm_dma_kernel).
template <int THT, int SO>
struct A64Geo {
    static constexpr int BN = 128, NW = 8, KC = 16, KG = 2, NT = 9, SCH = SO / 2;  // chunks per super-chunk
    static constexpr int HW = TW + 2, HH = THT + 2, NPIX = HH * HW;
    static constexpr int NGRP = (NPIX + 63) / 64, NPIXA = NGRP * 64;       // 64-pixel groups of the halo tile
    static constexpr int A_BYTES = SO * NPIXA * 16, NAP = SO * NGRP;       // DMA pieces per super-chunk
    static constexpr int NAW = (NAP + NW - 1) / NW;                        // per wave
    static constexpr int B_BYTES = NT * KG * BN * 16, NB = B_BYTES / 1024, NBW = (NB + NW - 1) / NW;
    static constexpr size_t RING = 2 * (size_t)A_BYTES + 2 * (size_t)B_BYTES + 1024;  // + sink
    static constexpr size_t SCRATCH = (size_t)NW * 32 * (BN / 2) * sizeof(float);      // epilogue transposes
    static constexpr size_t LDS = RING > SCRATCH ? RING : SCRATCH;
    static_assert(A_BYTES % 1024 == 0 && LDS <= 160 * 1024, "geometry");
};

template <int THT, int SO, bool AFF = false>
__global__ __launch_bounds__(512, 1) void conv3_ring64_kernel(ConvArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)  // buffer-resource builtins exist in the device pass only
    typedef A64Geo<THT, SO> G;
    typedef bf16 T;
    constexpr int MODE = HIPSEG_CONV3, BN = G::BN, NW = G::NW, KC = G::KC, KG = G::KG, NT = G::NT, SCH = G::SCH;
    constexpr int HW = G::HW, NPIX = G::NPIX, NPIXA = G::NPIXA, A_BYTES = G::A_BYTES, NAP = G::NAP, NAW = G::NAW;
    constexpr int B_BYTES = G::B_BYTES, NB = G::NB, NBW = G::NBW, RUNB = BN * 16;
    constexpr int WN = WG<BN, NW, THT>::WN, MT = WG<BN, NW, THT>::MT, NTL = WG<BN, NW, THT>::NTL;
    typedef __attribute__((address_space(3))) void lds_void;
    typedef int v4i __attribute__((ext_vector_type(4)));
    constexpr unsigned OOB_LANE = 0x80000000u;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const ringA = smem;
    unsigned char* const ringB = smem + 2 * A_BYTES;
    unsigned char* const sink = smem + 2 * A_BYTES + 2 * B_BYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const int bid = xcd_block(blockIdx.x, p.xcd);
    const int ntile = bid % p.ntn, mtile = bid / p.ntn;
    const int tx = mtile % p.tiles_x;
    const int ty = (mtile / p.tiles_x) % p.tiles_y;
    const int img = mtile / (p.tiles_x * p.tiles_y);
    const int y0 = ty * THT, x0 = tx * TW, n0 = ntile * BN;

    const unsigned bytes0 = (unsigned)((size_t)p.B * p.Hi * p.Wi * p.C0 * sizeof(T));
    const unsigned bytes1 = (unsigned)((size_t)p.B * p.Hi * p.Wi * p.C1 * sizeof(T));
    const __amdgpu_buffer_rsrc_t r_in0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in0), 0, (int)bytes0, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_in1 =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in1 ? p.in1 : p.in0), 0, (int)bytes1, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_w = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(p.wp), 0, (int)((size_t)NT * p.Kp * p.Np * sizeof(T)), 0x00020000);

    // A piece q = (pixel group q / SO, octet q % SO): 64 pixels of one k-octet, 1 KiB contiguous in the [octet][pixel]
    // image.  The SO pieces of a pixel group -- the SO 16-byte parts of the same cache lines -- belong to SO different
    // waves and are issued at the same step of the walk (L1 serves all but the first).  Per-lane byte offsets against
    // either source tensor; border / padding pixels are out of range (zero fill).
    unsigned avo0[NAW], avo1[NAW];
#pragma unroll
    for (int j = 0; j < NAW; ++j) {
        const int q = j * NW + wave;
        const int oct = q % SO, pix = (q / SO) * 64 + lane;
        const int iy = y0 - 1 + pix / HW, ix = x0 - 1 + pix % HW;
        const bool ok = q < NAP && pix < NPIX && iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi;
        const long apix = ((long)img * p.Hi + iy) * p.Wi + ix;
        avo0[j] = ok ? (unsigned)(apix * p.C0 * (long)sizeof(T) + oct * 16) : OOB_LANE;
        avo1[j] = ok ? (unsigned)(apix * p.C1 * (long)sizeof(T) + oct * 16) : OOB_LANE;
    }
    const int kgp = p.Kp / 8;
    unsigned bvo[NBW];
#pragma unroll
    for (int j = 0; j < NBW; ++j) {
        const int s = j * NW + wave;
        const int bb = s * 1024 + lane * 16;
        const int run = bb / RUNB, off = bb % RUNB;
        const int tap = run / KG, kgl = run % KG;
        bvo[j] = s < NB ? (unsigned)((((size_t)tap * kgp + kgl) * p.Np + n0) * 16 + off) : OOB_LANE;
    }
    v4i bst[NBW];
    auto loadB = [&](int c0) {
        const unsigned so = (unsigned)(c0 / 8) * (unsigned)p.Np * 16u;
#pragma unroll
        for (int j = 0; j < NBW; ++j) bst[j] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(r_w, bvo[j], so, 0));
    };
    auto writeB = [&](int slot) {
        unsigned char* base = ringB + slot * B_BYTES;
#pragma unroll
        for (int j = 0; j < NBW; ++j) {
            const int s = j * NW + wave;
            if ((j + 1) * NW <= NB || s < NB) *reinterpret_cast<v4i*>(base + s * 1024 + lane * 16) = bst[j];
        }
    };
    // idx-th A piece of this wave for the super-chunk starting at channel sc0 (launch condition: a super-chunk lies
    // inside ONE source tensor and inside K)
    auto pieceA = [&](int slotA, int sc0, int idx) {
        const int q = idx * NW + wave;
        lds_void* dst = (lds_void*)(((idx + 1) * NW <= NAP || q < NAP)
                                        ? ringA + slotA * A_BYTES + ((q % SO) * NPIXA + (q / SO) * 64) * 16
                                        : sink);
        if (sc0 < p.C0)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r_in0, dst, 16, avo0[idx], (unsigned)sc0 * (unsigned)sizeof(T), 0, 0);
        else
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r_in1, dst, 16, avo1[idx], (unsigned)(sc0 - p.C0) * (unsigned)sizeof(T),
                                                     0, 0);
    };

    f32x16 acc[MT][NTL];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTL; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    int hbase[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) hbase[i] = (2 * (wm * MT + i) + (r >> 4)) * HW + sub_px<MODE>(r);
    int ncol[NTL];
#pragma unroll
    for (int j = 0; j < NTL; ++j) ncol[j] = wn * (BN / WN) + j * 32 + r;

    const int nchunks = p.Kp / KC, nsuper = nchunks / SCH;
    // prologue: super-chunk 0 and weight chunk 0
#pragma unroll
    for (int q = 0; q < NAW; ++q) pieceA(0, 0, q);
    loadB(0);
    writeB(0);
    constexpr int SLOTS = SCH * NT;                  // (chunk, tap) steps inside a super-chunk
    constexpr int PSTEP = SLOTS / NAW > 0 ? SLOTS / NAW : 1;  // a DMA piece every PSTEP steps
    static_assert((NAW - 1) * PSTEP < SLOTS, "all pieces of the next super-chunk are issued inside the current one");
    for (int su = 0; su < nsuper; ++su) {
        const bool moreA = su + 1 < nsuper;
        const unsigned char* sA = ringA + (su & 1) * A_BYTES;
#pragma unroll
        for (int kl = 0; kl < SCH; ++kl) {
            const int kc = su * SCH + kl;
            // first chunk of a super-chunk: its pieces (issued during the previous super-chunk) must have landed
            if (kl == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's weight-chunk ds_writes
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            const bool moreB = kc + 1 < nchunks;
            if (moreB) loadB((kc + 1) * KC);
            const unsigned char* sB = ringB + (kc & 1) * B_BYTES;
            bf16x8 bf[2][NTL], af[2][MT];
            auto fetch = [&](int slot, int tap) {
                const int toff = tap_off<MODE>(tap);
#pragma unroll
                for (int j = 0; j < NTL; ++j)
                    bf[slot][j] = *reinterpret_cast<const bf16x8*>(sB + ((size_t)(tap * KG + h) * BN + ncol[j]) * 16);
#pragma unroll
                for (int i = 0; i < MT; ++i)
                    af[slot][i] =
                        *reinterpret_cast<const bf16x8*>(sA + ((size_t)(kl * 2 + h) * NPIXA + hbase[i] + toff) * 16);
            };
            fetch(0, 0);
#pragma unroll
            for (int tap = 0; tap < NT; ++tap) {
                constexpr int HALF = (MT + 1) / 2;
#pragma unroll
                for (int i = 0; i < HALF; ++i)
#pragma unroll
                    for (int j = 0; j < NTL; ++j)
                        acc[i][j] =
                            __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[tap & 1][i], bf[tap & 1][j], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (tap + 1 < NT) fetch((tap + 1) & 1, tap + 1);
                {
                    const int step = kl * NT + tap;
                    if (moreA && step % PSTEP == 0 && step / PSTEP < NAW)
                        pieceA((su + 1) & 1, (su + 1) * SCH * KC, step / PSTEP);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = HALF; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NTL; ++j)
                        acc[i][j] =
                            __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[tap & 1][i], bf[tap & 1][j], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (moreB) writeB((kc + 1) & 1);
        }
    }
    conv_epilogue<bf16, MODE, BN, NW, THT, AFF>(p, acc, smem, mtile, img, y0, x0, n0);
#else
    (void)p;
#endif
}

// ---------------------------------------------------------------------------------------------------
// Weights-stationary 3x3 kernel for the <= 64-channel layers (the full-resolution layers of the U-Net).
// There the implicit GEMM is short (K = 9 x 32..64) and the ring kernel spends as many LDS-DMA pieces on
// re-fetching the weight chunk for every pixel tile as on the activations -- and LDS-DMA moves only ~11 B/clk
// per CU, whatever the source.  Here every wave keeps the weights of its 32 output channels for ALL taps and
// ALL input channels in registers (9 x K/16 MFMA B-fragments = 144 VGPRs at K = 64), loaded once per workgroup;
// LDS holds nothing but a 3-deep ring of whole-K halo tiles (8 x 16 output pixels, 10 x 18 halo), filled by
// LDS-DMA with per-lane constant source offsets.  Workgroups are persistent (2 x 4 waves per CU, independent
// barriers, so one workgroup's epilogue stores run under the other's MFMAs) and walk a strided tile list;
// one s_barrier per 128-pixel tile; the epilogue is wave-private (bf16 LDS transpose, 16-byte stores, BN
// partial statistics in the same row convention as conv_epilogue).
template <int KCH, int NB>
struct WsGeo {
    static constexpr int THS = 8, HWs = TW + 2, HHs = THS + 2, NPIX = HHs * HWs, NPIXP = (NPIX + 63) / 64 * 64;
    static constexpr int NG = NPIXP / 64;                           // 64-pixel groups of the halo tile
    static constexpr int NOCT = 2 * KCH, NW = 4, OPW = NOCT / NW;   // k-octets; octets per wave
    static constexpr int A_BYTES = NOCT * NPIXP * 16, NAW = OPW * NG;  // DMA pieces per wave per tile
    static constexpr int NBUF = 3, DIST = NBUF - 1;
    static constexpr int WM = NW / NB, MT = (THS / 2) / WM;         // NB = 2: 2 x 2 waves, MT 2;  NB = 1: 4 x 1, MT 1
    static constexpr int NACC = MT == 1 ? 2 : MT;                   // MT 1: two accumulators break the MFMA chain
    static constexpr int DEPTH = MT == 1 ? 4 : 3;                   // fragment prefetch ring (steps)
    static constexpr int SCR = 2048;                                // per-wave: two bf16 [16 px][32 ch] transpose tiles
    static constexpr size_t LDS = (size_t)NBUF * A_BYTES + NW * SCR;
    static constexpr int NST = 2 * MT;                              // output stores per wave per tile
    static_assert(NOCT % NW == 0 && LDS <= 80 * 1024, "geometry");
};

#ifdef WS_STAMP
// diagnostic build only (scripts/build_variant.sh wsstamp conv_igemm.hip -DWS_STAMP): per-workgroup cycle sums of the
// tile loop's phases (wave 0) into a buffer nothing else reads
}  // namespace
__device__ unsigned long long g_ws_stamp[1024 * 8];
extern "C" int hipseg_debug_ws_stamps(void* host, int nwg) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_ws_stamp), (size_t)nwg * 8 * sizeof(unsigned long long)) == hipSuccess ? 0 : 1;
}
namespace {
#endif

// LD (round 4, hipseg_conv3_bnrelu_in): the INPUT tensor is the pre-normalisation output of the previous convolution and
// the convolution runs over relu(in * ld_scale[c] + ld_shift[c]) -- BatchNorm + ReLU applied in the consumer's load path
// (SURVEY section 2.2), so the activated tensor and the bn_relu_apply pass that wrote it never exist.  The wave that
// DMA'd a piece rewrites it in LDS after its own counted wait and before the tile's barrier (its 8 channels are one
// octet: scale / shift are wave-uniform scalars); lanes whose pixel lies outside the image keep the zero the DMA
// filled in (the convolution pads the ACTIVATED tensor with zeros).  Measured cost on 64 -> 64 at 256 x 256: +7 us on
// an 82-us launch (profiles/r04_bn_on_load.txt) against the 40-57 us pass it removes.
template <int KCH, int NB, bool DBG, bool AFF = false, bool LD = false>
__global__ __launch_bounds__(256, 2) void conv3_wstat_kernel(ConvArgs p, int total_tiles, int tiles_y8) {
    typedef bf16 T;
    typedef WsGeo<KCH, NB> G;
    constexpr int MODE = HIPSEG_CONV3, NT = 9, HW = G::HWs, NPIXP = G::NPIXP, NG = G::NG, OPW = G::OPW;
    constexpr int A_BYTES = G::A_BYTES, NAW = G::NAW, NBUF = G::NBUF, DIST = G::DIST, MT = G::MT, NACC = G::NACC;
    constexpr int DEPTH = G::DEPTH, NST = G::NST, NQ = KCH * NT;
    typedef __attribute__((address_space(3))) void lds_void;
    typedef const __attribute__((address_space(1))) void glb_void;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int wn = wave % NB, wm = wave / NB;
    const T* in0 = reinterpret_cast<const T*>(p.in0);
    const T* in1 = reinterpret_cast<const T*>(p.in1);
    T* out0 = reinterpret_cast<T*>(p.out0);
    T* out1 = reinterpret_cast<T*>(p.out1);
    const T* zero = reinterpret_cast<const T*>(&g_zero16);

    // ---- stationary weights (packed [tap][Kp/8][Np][8]): lane (n = r, k-octet h) of every (tap, chunk) fragment
    bf16x8 wreg[NT][KCH];
    {
        const unsigned char* wp = reinterpret_cast<const unsigned char*>(p.wp);
        const int kgp = p.Kp / 8;
#pragma unroll
        for (int tap = 0; tap < NT; ++tap)
#pragma unroll
            for (int c = 0; c < KCH; ++c)
                wreg[tap][c] = *reinterpret_cast<const bf16x8*>(
                    wp + (((size_t)tap * kgp + c * 2 + h) * p.Np + wn * 32 + r) * 16);
    }
    const int n = wn * 32 + r;  // this lane's output channel (accumulator column); N == Np is a launch condition
    const float bv = p.bias ? p.bias[n] : 0.f;
    constexpr bool aff = AFF;  // fused inference epilogue (see ConvArgs); compile-time
    const float scl = aff ? p.post_scale[n] : 1.f;

    // ---- per-lane DMA constants: pixel (py, px) of the halo tile for each 64-pixel group
    int poff[NG], pyx[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int pix = g * 64 + lane, py = pix / HW, px = pix - py * HW;
        poff[g] = py * p.W + px;
        pyx[g] = pix < G::NPIX ? (py | (px << 8)) : (0x7f | (0x7f << 8));  // pad pixels: always out of range
    }
    // A-fragment pixel of each M sub-tile (rows r of the MFMA; see sub_px)
    int hbase[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int py = 2 * (wm * MT + i) + (r >> 4), px = sub_px<MODE>(r);
        hbase[i] = (h * NPIXP + py * HW + px) * 16;
    }

    // ---- tile cursors (incremental: no per-tile divisions).  c = compute cursor, n = staging cursor (DIST ahead)
    struct Cur {
        int tile, tx, ty, img;
    };
    const int per_img = p.tiles_x * tiles_y8, GS = gridDim.x;
    const int sx = GS % p.tiles_x, sy = (GS / p.tiles_x) % tiles_y8, si = GS / per_img;
    auto advance = [&](Cur& c) {
        c.tile += GS;
        c.tx += sx;
        const int cx = c.tx >= p.tiles_x;
        c.tx -= cx ? p.tiles_x : 0;
        c.ty += sy + cx;
        const int cy = c.ty >= tiles_y8;
        c.ty -= cy ? tiles_y8 : 0;
        c.img += si + cy;
    };
    Cur cc, cn;
    cc.tile = blockIdx.x;
    cc.tx = cc.tile % p.tiles_x;
    cc.ty = (cc.tile / p.tiles_x) % tiles_y8;
    cc.img = cc.tile / per_img;
    cn = cc;

    // one 1-KiB piece (octet, 64-pixel group) of the tile under the staging cursor
    auto piece = [&](int j, unsigned char* base, int img, int y0, int x0, bool live) {
        const int oct = wave * OPW + j % OPW, g = j / OPW;  // the octets of one pixel group at adjacent steps (L1 reuse)
        const int c0 = oct * 8;  // wave-uniform
        const T* src0;
        int cs;
        if (c0 < p.C0) {
            src0 = in0 + c0;
            cs = p.C0;
        } else {
            src0 = in1 + (c0 - p.C0);
            cs = p.C1;
        }
        const int gy = y0 - 1 + (pyx[g] & 0xff), gx = x0 - 1 + (pyx[g] >> 8);
        const bool ok = live && c0 < p.K && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W &&
                        !(DBG && (p.debug & 1));
        const int e = (img * p.H + y0 - 1) * p.W + x0 - 1 + poff[g];
        const T* src = ok ? src0 + (long)e * cs : zero;
        dma_piece_ptr(src, (unsigned)(uintptr_t)(lds_void*)(base + (oct * NG + g) * 1024));
    };

    // prologue: DIST tiles go out at once
#pragma unroll
    for (int d = 0; d < DIST; ++d) {
        const bool live = cn.tile < total_tiles;
#pragma unroll
        for (int j = 0; j < NAW; ++j) piece(j, smem + d * A_BYTES, cn.img, cn.ty * G::THS, cn.tx * TW, live);
        advance(cn);
    }

    unsigned char* scr = smem + NBUF * A_BYTES + wave * G::SCR;
    // per-lane store constants: the read-back lane owns 8 channels (cv) of pixel (lane >> 2) of a 16-pixel row
    const int s_px = lane >> 2, s_nn = wn * 32 + (lane & 3) * 8;
    T* const s_dst = s_nn < p.N0 ? out0 + s_nn : out1 + (s_nn - p.N0);
    const int s_stride = s_nn < p.N0 ? p.N0 : p.N1;
    int s_x[2];  // pixel column of the lane's pixel in tile rows 2s (half 0) and 2s + 1 (half 1: rotated sub-tile rows)
    s_x[0] = sub_px<MODE>(s_px);
    s_x[1] = sub_px<MODE>(16 + s_px);

    // BatchNorm partial statistics: accumulated in registers over ALL tiles of this workgroup (lane = channel),
    // one row per (workgroup, wm) written at the end -- see hipseg_conv_stats_rows().
    float ssum = 0.f, ssq = 0.f;
    int cur = 0;
    int hist = 0;                // bit k: the tile k+1 iterations back was interior (issued all its stores)
#ifdef WS_STAMP
    unsigned long long st_wait = 0, st_bar = 0, st_main = 0, st_epi = 0, st_n = 0;
    const unsigned long long st_t0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    for (; cc.tile < total_tiles; advance(cc)) {
#ifdef WS_STAMP
        const unsigned long long ta = __builtin_amdgcn_s_memtime();
#endif
        // Counted wait for this tile's pieces.  VMEM ops retire in issue order; per wave the ops younger than
        // P(tile k) are: S(k-2), P(k+1), S(k-1).  The count is exact only when both previous tiles were interior
        // (every predicated store really issued); otherwise, and for the first two tiles, drain.
        if (hist == 3)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DIST - 1) * NAW + DIST * NST) : "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef WS_STAMP
        const unsigned long long tb = __builtin_amdgcn_s_memtime();
#endif
        if constexpr (LD) {
            // scale / shift of an octet are wave-uniform: re-read per tile through scalar loads (scalar-cache hits, one
            // wait per octet) instead of living in registers across the tile loop -- the kernel has none to spare (32
            // more values spilled 26 VGPRs to scratch; one scalar load per PIECE with its own wait cost 55 us)
            unsigned char* sAw = smem + cur * A_BYTES;
            const int ly0 = cc.ty * G::THS - 1, lx0 = cc.tx * TW - 1;
#pragma unroll
            for (int o = 0; o < OPW; ++o) {
                const int oct = __builtin_amdgcn_readfirstlane(wave * OPW + o);
                // (constant address space: SCALAR loads -- as vector loads they would count in vmcnt and the compiler's wait
                // for them would drain the LDS-DMA pieces of the next tiles and the previous tile's stores: 137 us)
                typedef const __attribute__((address_space(4))) float cfloat;
                cfloat* sp = (cfloat*)(p.ld_scale + oct * 8);
                cfloat* hp = (cfloat*)(p.ld_shift + oct * 8);
                float sc[8], sh[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    sc[e] = sp[e];
                    sh[e] = hp[e];
                }
                bf16x8 v[NG];
#pragma unroll
                for (int g = 0; g < NG; ++g) v[g] = *(reinterpret_cast<bf16x8*>(sAw + (oct * NG + g) * 1024) + lane);
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    const int gy = ly0 + (pyx[g] & 0xff), gx = lx0 + (pyx[g] >> 8);
                    const bool in = (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[g][e] = (bf16)(in ? fmaxf((float)v[g][e] * sc[e] + sh[e], 0.f) : 0.f);
                    *(reinterpret_cast<bf16x8*>(sAw + (oct * NG + g) * 1024) + lane) = v[g];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
#ifdef WS_STAMP
        const unsigned long long tc = __builtin_amdgcn_s_memtime();
#endif
        const bool more = cn.tile < total_tiles;
        unsigned char* nbase = smem + ((cur + DIST) % NBUF) * A_BYTES;
        const int ny0 = cn.ty * G::THS, nx0 = cn.tx * TW, nimg = cn.img;
        const unsigned char* sA = smem + cur * A_BYTES;
        cur = (cur + 1) % NBUF;

        f32x16 acc[NACC];
#pragma unroll
        for (int i = 0; i < NACC; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

        // step q = (chunk, tap): MT reads + MT MFMAs; fragments are fetched DEPTH - 1 steps ahead
        bf16x8 af[DEPTH][MT];
        auto fetch = [&](int slot, int q) {
            const int c = q / NT, tap = q % NT;
#pragma unroll
            for (int i = 0; i < MT; ++i)
                af[slot][i] = *reinterpret_cast<const bf16x8*>(sA + hbase[i] + (c * 2 * NPIXP + tap_off<MODE>(tap)) * 16);
        };
#pragma unroll
        for (int q = 0; q < DEPTH - 1; ++q) fetch(q, q);
        constexpr int PSTEP = NQ / NAW;  // a DMA piece every PSTEP steps
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            if (q + DEPTH - 1 < NQ) fetch((q + DEPTH - 1) % DEPTH, q + DEPTH - 1);
            if (q % PSTEP == 0 && q / PSTEP < NAW) piece(q / PSTEP, nbase, nimg, ny0, nx0, more);
            if (DBG && (p.debug & 4)) continue;  // ablation build only
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int ai = MT == 1 ? (q & 1) : i;
                acc[ai] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[q % DEPTH][i], wreg[q % NT][q / NT], acc[ai], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        static_assert((NAW - 1) * PSTEP < NQ, "all pieces issued inside the step walk");
#ifdef WS_STAMP
        const unsigned long long td = __builtin_amdgcn_s_memtime();
#endif
        advance(cn);
        if (MT == 1) {
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[0][e] += acc[1][e];
        }

        // ---- wave-private epilogue: bias, statistics, bf16 transpose through a 1-KiB LDS tile (16 px x 32 ch),
        // 16-byte stores.  Interior tiles (all of the U-Net's) take the branch-free path.
        const int y0 = cc.ty * G::THS, x0 = cc.tx * TW, img = cc.img;
        const bool interior = y0 + G::THS <= p.H && x0 + TW <= p.W && !(DBG && (p.debug & 24));
        const int tbase = (img * p.H + y0) * p.W + x0;  // first pixel of the tile
        if (DBG && (p.debug & 8)) {
            hist = 0;
            continue;
        }
        // Software-pipelined over the 2 * MT half sub-tiles (16 pixels x 32 channels each) on TWO scratch tiles: the
        // read-back of half k is issued, then half k + 1 is converted and written to the other tile, and only then half
        // k's 16-byte store consumes the read -- the LDS round trip (several hundred cycles while the other workgroup's
        // fragment reads keep the LDS busy: round-4 stamps put the epilogue at 3410 cycles per tile, as long as the MFMA
        // loop, four dependent round trips) runs under the next half's conversions instead of in front of them.
        bf16x8 pend = {};
        int pend_rel = 0;
        bool pend_ok = false;
#pragma unroll
        for (int idx = 0; idx < 2 * MT; ++idx) {
            const int i = idx >> 1, half = idx & 1;
            const int yr = 2 * (wm * MT + i);  // first of the sub-tile's two tile rows
            unsigned char* const sc = scr + (idx & 1) * 1024;
            if (interior) {
#pragma unroll
                for (int e8 = 0; e8 < 8; ++e8) {
                    const int e = half * 8 + e8;
                    const int rr = (e & 3) + 8 * (e >> 2) + 4 * h;  // rr >> 4 == half
                    const float v = aff ? fmaxf(fmaf(acc[i][e], scl, bv), 0.f) : acc[i][e] + bv;
                    reinterpret_cast<bf16*>(sc)[(rr & 15) * 32 + r] = (bf16)v;
                    ssum += v;
                    ssq += v * v;
                }
            } else {
#pragma unroll
                for (int e8 = 0; e8 < 8; ++e8) {
                    const int e = half * 8 + e8;
                    const int rr = (e & 3) + 8 * (e >> 2) + 4 * h;
                    const float v = aff ? fmaxf(fmaf(acc[i][e], scl, bv), 0.f) : acc[i][e] + bv;
                    reinterpret_cast<bf16*>(sc)[(rr & 15) * 32 + r] = (bf16)v;
                    if (y0 + yr + half < p.H && x0 + sub_px<MODE>(rr) < p.W) {
                        ssum += v;
                        ssq += v * v;
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (idx > 0 && pend_ok) *reinterpret_cast<bf16x8*>(s_dst + (long)(tbase + pend_rel) * s_stride) = pend;
            pend = *reinterpret_cast<const bf16x8*>(sc + (s_px * 32 + (lane & 3) * 8) * 2);
            pend_rel = (yr + half) * p.W + s_x[half];
            pend_ok = interior || (y0 + yr + half < p.H && x0 + s_x[half] < p.W && !(DBG && (p.debug & 16)));
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        if (pend_ok) *reinterpret_cast<bf16x8*>(s_dst + (long)(tbase + pend_rel) * s_stride) = pend;
        hist = ((hist << 1) | (interior ? 1 : 0)) & 3;
#ifdef WS_STAMP
        {
            const unsigned long long te = __builtin_amdgcn_s_memtime();
            st_wait += tb - ta;
            st_bar += tc - tb;
            st_main += td - tc;
            st_epi += te - td;
            st_n += 1;
        }
#endif
    }
#ifdef WS_STAMP
    if (tid == 0 && blockIdx.x < 1024) {
        unsigned long long* o = g_ws_stamp + blockIdx.x * 8;
        o[0] = st_wait; o[1] = st_bar; o[2] = st_main; o[3] = st_epi; o[4] = st_n;
        o[5] = __builtin_amdgcn_s_memtime() - st_t0;
        o[6] = __builtin_amdgcn_s_memrealtime() - st_r0;
        o[7] = st_r0;
    }
#endif
    if (p.stats) {
        const float S = ssum + __shfl_xor(ssum, 32, 64);
        const float Q = ssq + __shfl_xor(ssq, 32, 64);
        const size_t row = (size_t)blockIdx.x * G::WM + wm;
        if (h == 0) {
            p.stats[(row * 2 + 0) * p.N + n] = S;
            p.stats[(row * 2 + 1) * p.N + n] = Q;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int KCH, int NB>
int launch_wstat(const ConvArgs& a, hipStream_t s) {
    typedef WsGeo<KCH, NB> G;
    if (int rc = hs_set_max_lds(reinterpret_cast<const void*>(&conv3_wstat_kernel<KCH, NB, false>), (size_t)G::LDS)) return rc;
    if (int rc = hs_set_max_lds(reinterpret_cast<const void*>(&conv3_wstat_kernel<KCH, NB, true>), (size_t)G::LDS)) return rc;
    const int tiles_y8 = cdiv(a.H, G::THS);
    const long total = (long)a.B * a.tiles_x * tiles_y8;
    const long grid = 2 * a.ncu;  // wstat_grid(): total >= 4 * ncu
    if (a.ld_scale) {
        if (int rc = hs_set_max_lds(reinterpret_cast<const void*>(&conv3_wstat_kernel<KCH, NB, false, false, true>), (size_t)G::LDS))
            return rc;
        hipLaunchKernelGGL((conv3_wstat_kernel<KCH, NB, false, false, true>), dim3((unsigned)grid), dim3(256), G::LDS, s, a,
                           (int)total, tiles_y8);
    } else if (a.post_scale) {
        if (int rc = hs_set_max_lds(reinterpret_cast<const void*>(&conv3_wstat_kernel<KCH, NB, false, true>), (size_t)G::LDS))
            return rc;
        hipLaunchKernelGGL((conv3_wstat_kernel<KCH, NB, false, true>), dim3((unsigned)grid), dim3(256), G::LDS, s, a,
                           (int)total, tiles_y8);
    } else if (a.debug)
        hipLaunchKernelGGL((conv3_wstat_kernel<KCH, NB, true>), dim3((unsigned)grid), dim3(256), G::LDS, s, a, (int)total,
                           tiles_y8);
    else
        hipLaunchKernelGGL((conv3_wstat_kernel<KCH, NB, false>), dim3((unsigned)grid), dim3(256), G::LDS, s, a,
                           (int)total, tiles_y8);
    HS_LAUNCH_CHECK("conv3_wstat");
    return HIPSEG_OK;
}

template <int MODE, int BN, int THT = 16>
int launch_dma(const ConvArgs& a0, hipStream_t s) {
    constexpr int HH = MODE == HIPSEG_CONV3 ? THT + 2 : (MODE == HIPSEG_CONV2S2 ? 2 * THT : THT);
    constexpr int NPIXP = (HH * Geo<MODE>::HW + 63) / 64 * 64, NT = Geo<MODE>::NT;
    constexpr size_t ring = DmaBufs<BN, THT>::value * (size_t)(2 * NPIXP * 16 + NT * 2 * BN * 16) + 1024;
    constexpr int NW = DmaWaves<BN, THT>::value;
    constexpr size_t scratch = (size_t)NW * 32 * (BN / WG<BN, NW, THT>::WN) * sizeof(float);  // epilogue transposes
    constexpr size_t lds = ring > scratch ? ring : scratch;
    static_assert(lds <= 163840, "LDS budget");
    ConvArgs a = a0;
    a.tiles_y = cdiv(a.H, THT);
    if (int rc = hs_set_max_lds(reinterpret_cast<const void*>(&conv_igemm_dma_kernel<MODE, BN, THT>), (size_t)lds)) return rc;
    const long grid = (long)a.B * a.tiles_x * a.tiles_y * a.ntn;
    static const bool no_xcd = getenv("HIPSEG_NO_XCD") != nullptr;
    a.xcd = (!no_xcd && grid % 8 == 0 && grid >= 64) ? (int)(grid / 8) : 0;
    if constexpr (MODE == HIPSEG_CONV3) {
        if (a.post_scale) {  // inference epilogue: its own instantiation
            if (int rc = hs_set_max_lds(reinterpret_cast<const void*>(&conv_igemm_dma_kernel<MODE, BN, THT, true>), (size_t)lds))
                return rc;
            hipLaunchKernelGGL((conv_igemm_dma_kernel<MODE, BN, THT, true>), dim3((unsigned)grid), dim3(64 * NW), lds, s, a);
            HS_LAUNCH_CHECK("conv_igemm_dma(affine)");
            return HIPSEG_OK;
        }
    }
    hipLaunchKernelGGL((conv_igemm_dma_kernel<MODE, BN, THT>), dim3((unsigned)grid), dim3(64 * NW), lds, s, a);
    HS_LAUNCH_CHECK("conv_igemm_dma");
    return HIPSEG_OK;
}

template <int MODE>
int launch_gemm1(const ConvArgs& a0, hipStream_t s) {
    constexpr size_t lds = 3 * (size_t)(256 * 64 * 2 + 8 * 128 * 16);  // ring (144 KiB) > epilogue scratch (64 KiB)
    ConvArgs a = a0;
    if (int rc = hs_set_max_lds(reinterpret_cast<const void*>(&gemm1_kernel<MODE>), (size_t)lds)) return rc;
    const long grid = (long)a.B * a.tiles_x * a.tiles_y * a.ntn;
    static const bool no_xcd = getenv("HIPSEG_NO_XCD") != nullptr;
    a.xcd = (!no_xcd && grid % 8 == 0 && grid >= 64) ? (int)(grid / 8) : 0;
    if constexpr (MODE == HIPSEG_CONV2S2) {
        if (a.bw_x) {  // data gradient + BatchNorm-backward sums of its output (rows = pixel tiles)
            HS_REQUIRE(a.stats && a.bw_bn && a.N0 % 8 == 0 && !a.N1, "conv_gemm1: BatchNorm-backward epilogue operands");
            if (int rc = hs_set_max_lds(reinterpret_cast<const void*>(&gemm1_kernel<MODE, true>), (size_t)lds)) return rc;
            hipLaunchKernelGGL((gemm1_kernel<MODE, true>), dim3((unsigned)grid), dim3(512), lds, s, a);
            HS_LAUNCH_CHECK("conv_gemm1(bn sums)");
            return HIPSEG_OK;
        }
    }
    hipLaunchKernelGGL((gemm1_kernel<MODE>), dim3((unsigned)grid), dim3(512), lds, s, a);
    HS_LAUNCH_CHECK("conv_gemm1");
    return HIPSEG_OK;
}

template <int THT, int SO>
int launch_ring64(const ConvArgs& a0, hipStream_t s) {
    typedef A64Geo<THT, SO> G;
    ConvArgs a = a0;
    a.tiles_y = cdiv(a.H, THT);
    if (int rc = hs_set_max_lds(reinterpret_cast<const void*>(&conv3_ring64_kernel<THT, SO>), (size_t)G::LDS)) return rc;
    const long grid = (long)a.B * a.tiles_x * a.tiles_y * a.ntn;
    static const bool no_xcd = getenv("HIPSEG_NO_XCD") != nullptr;
    a.xcd = (!no_xcd && grid % 8 == 0 && grid >= 64) ? (int)(grid / 8) : 0;
    if (a.post_scale) {
        if (int rc = hs_set_max_lds(reinterpret_cast<const void*>(&conv3_ring64_kernel<THT, SO, true>), (size_t)G::LDS)) return rc;
        hipLaunchKernelGGL((conv3_ring64_kernel<THT, SO, true>), dim3((unsigned)grid), dim3(512), G::LDS, s, a);
    } else
        hipLaunchKernelGGL((conv3_ring64_kernel<THT, SO>), dim3((unsigned)grid), dim3(512), G::LDS, s, a);
    HS_LAUNCH_CHECK("conv3_ring64");
    return HIPSEG_OK;
}

template <int MODE>
int launch_dma_bn(const ConvArgs& a, int bn, hipStream_t s) {
    if (bn == 128) return launch_dma<MODE, 128>(a, s);
    if (bn == 64) return launch_dma<MODE, 64>(a, s);
    return launch_dma<MODE, 32>(a, s);
}

int launch_dma_mode(const ConvArgs& a, int mode, int bn, hipStream_t s) {
    switch (mode) {
        case HIPSEG_CONV3: return launch_dma_bn<HIPSEG_CONV3>(a, bn, s);
        case HIPSEG_CONV1: return launch_dma_bn<HIPSEG_CONV1>(a, bn, s);
        case HIPSEG_CONV2S2: return launch_dma_bn<HIPSEG_CONV2S2>(a, bn, s);
        default: return launch_dma_bn<HIPSEG_CONVT>(a, bn, s);
    }
}

template <typename T, int MODE, int BN>
int launch(const ConvArgs& a, hipStream_t s) {
    constexpr int KC = KT<T>::KC;
    constexpr int NPIX = Geo<MODE>::HH * Geo<MODE>::HW, NT = Geo<MODE>::NT;
    size_t lds = (size_t)(KC * NPIX + NT * KC * BN) * sizeof(T);
    constexpr int WN = WG<BN, 4>::WN;
    const size_t scratch = (size_t)4 * 32 * (BN / WN) * sizeof(float);  // per-wave fp32 transpose tiles (epilogue)
    if (lds < scratch) lds = scratch;
    const long grid = (long)a.B * a.tiles_x * a.tiles_y * a.ntn;
    if constexpr (MODE == HIPSEG_CONV3) {
        if (a.post_scale) {  // inference epilogue: its own instantiation
            hipLaunchKernelGGL((conv_igemm_kernel<T, MODE, BN, true>), dim3((unsigned)grid), dim3(256), lds, s, a);
            HS_LAUNCH_CHECK("conv_igemm(affine)");
            return HIPSEG_OK;
        }
    }
    hipLaunchKernelGGL((conv_igemm_kernel<T, MODE, BN>), dim3((unsigned)grid), dim3(256), lds, s, a);
    HS_LAUNCH_CHECK("conv_igemm");
    return HIPSEG_OK;
}

template <typename T, int MODE>
int launch_bn(const ConvArgs& a, int bn, hipStream_t s) {
    if (bn == 128) return launch<T, MODE, 128>(a, s);
    if (bn == 64) return launch<T, MODE, 64>(a, s);
    return launch<T, MODE, 32>(a, s);
}

template <typename T>
int launch_mode(const ConvArgs& a, int mode, int bn, hipStream_t s) {
    switch (mode) {
        case HIPSEG_CONV3: return launch_bn<T, HIPSEG_CONV3>(a, bn, s);
        case HIPSEG_CONV1: return launch_bn<T, HIPSEG_CONV1>(a, bn, s);
        case HIPSEG_CONV2S2: return launch_bn<T, HIPSEG_CONV2S2>(a, bn, s);
        default: return launch_bn<T, HIPSEG_CONVT>(a, bn, s);
    }
}

// Shape test for the weights-stationary kernel (conv3_wstat_kernel): <= 64-channel 3x3 layers with enough 128-pixel
// tiles for two persistent workgroups per CU.  Returns the launch grid (0 = not applicable).
int wstat_grid(int dtype, int mode, int C0, int C1, int N0, int N1, int B, int H, int W) {
    static const bool off = getenv("HIPSEG_NO_WSTAT") != nullptr || getenv("HIPSEG_NO_DMA") != nullptr;
    if (off || dtype != HIPSEG_BF16 || mode != HIPSEG_CONV3) return 0;
    if (C0 % 8 || C1 % 8 || N0 % 8 || N1 % 8) return 0;
    const int K = C0 + C1, N = N0 + N1, Kp = (K + 15) / 16 * 16;
    if (!(N == 32 || N == 64) || !(Kp == 32 || Kp == 64)) return 0;
    const long total = (long)B * cdiv(W, TW) * cdiv(H, 8);
    const int ncu = device_cus();
    if (total < 4 * ncu) return 0;
    return 2 * ncu;
}

int bn_for(int N) {
    const int bn_max = 128;
    const int bn = N > 64 ? 128 : (N > 32 ? 64 : 32);
    return bn > bn_max ? bn_max : bn;
}

}  // namespace

extern "C" int hipseg_kpad(int K, int dtype) {
    const int kc = dtype == HIPSEG_BF16 ? KT<bf16>::KC : KT<float>::KC;
    return (K + kc - 1) / kc * kc;
}
extern "C" int hipseg_npad(int N) {
    const int bn = bn_for(N);
    return (N + bn - 1) / bn * bn;
}
// rows of the statistics workspace: one per 64 output pixels (4 per 16x16 tile)
extern "C" int hipseg_conv_mtiles(int B, int H, int W) { return 4 * B * cdiv(H, TH) * cdiv(W, TW); }
// rows hipseg_conv_igemm() writes for this call (<= hipseg_conv_mtiles(), which sizes the workspace)
extern "C" int hipseg_conv_stats_rows(int dtype, int mode, int C0, int C1, int N0, int N1, int B, int H, int W) {
    const int g = wstat_grid(dtype, mode, C0, C1, N0, N1, B, H, W);
    if (g) return g * (N0 + N1 == 64 ? 2 : 4);  // one row per (workgroup, pixel-row wave group)
    if (const int r16 = conv3_m16_rows(dtype, mode, C0, C1, N0, N1, B, H, W)) return conv3_m16_stats_rows(r16, N0 + N1, B, H, W);
    return hipseg_conv_mtiles(B, H, W);
}

int convt_stream_bws_rows(int C0, int N, int B, int H, int W, int ncu);

// Which kernel runs a ConvTranspose2d data gradient (mode CONV2S2: dy with C0 = Cout channels on the 2H x 2W grid -> dx
// with N0 = Cin channels on H x W) WITH the BatchNorm-backward epilogue: 0 none, 1 the streaming kernel, 2 gemm1_kernel;
// *rows = partial rows it writes.  Mirrors conv_igemm_impl's dispatch order for that mode.
static int convt_dgrad_bws_kernel(int dtype, int C0, int N0, int B, int H, int W, int* rows) {
    static const bool off = getenv("HIPSEG_NO_CONVT_DGRAD_BNSTATS") != nullptr || getenv("HIPSEG_NO_DMA") != nullptr ||
                            getenv("HIPSEG_NO_GEMM1") != nullptr;  // A/B switches
    *rows = 0;
    if (off || dtype != HIPSEG_BF16 || N0 % 8) return 0;
    if (convt_stream_applies(dtype, HIPSEG_CONV2S2, C0, 0, N0, 0, B, H, W)) {
        *rows = convt_stream_bws_rows(C0, N0, B, H, W, device_cus());
        return *rows ? 1 : 0;  // (the plain launch would take the streaming kernel too: no gemm1 fallback)
    }
    const size_t in_bytes = (size_t)B * 2 * H * 2 * W * (size_t)C0 * 2, w_bytes = (size_t)9 * hipseg_kpad(C0, dtype) * hipseg_npad(N0) * 2;
    if (in_bytes > ((size_t)1 << 30) || w_bytes > ((size_t)1 << 30)) return 0;
    if (C0 % 64 || hipseg_npad(N0) != N0 || N0 % 128 || hipseg_kpad(C0, dtype) != C0 || 4 * C0 < 256) return 0;
    *rows = B * cdiv(W, TW) * cdiv(H, TH);
    return 2;
}

static int conv_igemm_impl(int dtype, int mode, const void* in0, int C0, const void* in1, int C1, const void* wp,
                           const float* bias, const float* post_scale, void* out0, int N0, void* out1, int N1,
                           float* stats, int B, int H, int W, hipseg_stream_t stream, const void* bw_x = nullptr,
                           const float* bw_bn = nullptr, const float* ld_scale = nullptr, const float* ld_shift = nullptr) {
    HS_REQUIRE(dtype == HIPSEG_F32 || dtype == HIPSEG_BF16, "conv_igemm: bad dtype %d", dtype);
    HS_REQUIRE(mode >= HIPSEG_CONV3 && mode <= HIPSEG_CONVT, "conv_igemm: bad mode %d", mode);
    HS_REQUIRE(in0 && wp && out0 && C0 > 0 && N0 > 0, "conv_igemm: null operand or empty channel range");
    HS_REQUIRE(B > 0 && H > 0 && W > 0, "conv_igemm: empty pixel grid (%d,%d,%d)", B, H, W);
    HS_REQUIRE((C1 == 0) == (in1 == nullptr), "conv_igemm: in1/C1 mismatch");
    HS_REQUIRE((N1 == 0) == (out1 == nullptr), "conv_igemm: out1/N1 mismatch");
    HS_REQUIRE(!(mode == HIPSEG_CONVT && (N1 != 0 || stats)), "conv_igemm: CONVT takes one output, no stats");
    ConvArgs a;
    a.in0 = in0;
    a.in1 = in1;
    a.wp = wp;
    a.bias = bias;
    a.post_scale = post_scale;
    a.out0 = out0;
    a.out1 = out1;
    a.stats = stats;
    a.bw_x = bw_x;
    a.bw_bn = bw_bn;
    a.ld_scale = ld_scale;
    a.ld_shift = ld_shift;
    a.C0 = C0;
    a.C1 = C1;
    a.N0 = N0;
    a.N1 = N1;
    a.B = B;
    a.H = H;
    a.W = W;
    a.Hi = mode == HIPSEG_CONV2S2 ? 2 * H : H;
    a.Wi = mode == HIPSEG_CONV2S2 ? 2 * W : W;
    a.K = C0 + C1;
    a.Kp = hipseg_kpad(a.K, dtype);
    a.N = mode == HIPSEG_CONVT ? 4 * N0 : N0 + N1;
    a.Np = hipseg_npad(a.N);
    a.tiles_x = cdiv(W, TW);
    a.tiles_y = cdiv(H, TH);
    const int bn = bn_for(a.N);
    a.ntn = a.Np / bn;
    const int vec = dtype == HIPSEG_BF16 ? 8 : 4;
    a.vec_ok = (C0 % vec == 0) && (C1 % vec == 0);
    // ablation bits produce WRONG results by design: they exist only in ablation builds
    // (scripts/build_variant.sh <tag> <file> -DHIPSEG_ABLATE), never in the shipped library
#ifdef HIPSEG_ABLATE
    static const int dbg = getenv("HIPSEG_IGEMM_DEBUG") ? atoi(getenv("HIPSEG_IGEMM_DEBUG")) : 0;
#else
    constexpr int dbg = 0;
#endif
    a.debug = dbg;
    a.ncu = device_cus();
    a.xcd = 0;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == HIPSEG_BF16) {
        static const bool no_dma = getenv("HIPSEG_NO_DMA") != nullptr;  // debugging switch: generic kernel only
        if (!bw_x && wstat_grid(dtype, mode, C0, C1, N0, N1, B, H, W)) {
            if (a.Kp == 64) return a.Np == 64 ? launch_wstat<4, 2>(a, s) : launch_wstat<4, 1>(a, s);
            return a.Np == 64 ? launch_wstat<2, 2>(a, s) : launch_wstat<2, 1>(a, s);
        }
        HS_REQUIRE(!ld_scale, "conv3_bnrelu_in: no kernel with the load-side BatchNorm takes this shape");
        // the ring kernel addresses its operands through buffer descriptors with 2^30 / 2^31 out-of-range markers
        const size_t in_bytes = (size_t)B * a.Hi * a.Wi * (size_t)(C0 > C1 ? C0 : C1) * 2;
        const size_t w_bytes = (size_t)9 * a.Kp * a.Np * 2;
        const bool buf_ok = in_bytes <= ((size_t)1 << 30) && w_bytes <= ((size_t)1 << 30);
        if (bw_x && mode == HIPSEG_CONV2S2) {  // (hipseg_convT_dgrad_bnstats checked the shape)
            int rows;
            const int k = convt_dgrad_bws_kernel(dtype, C0, N0, B, H, W, &rows);
            HS_REQUIRE(k && !C1 && !N1 && !bias, "convT_dgrad_bnstats: no kernel with the BatchNorm-backward epilogue takes this shape");
            return k == 1 ? convt_stream_launch(a, mode, s) : launch_gemm1<HIPSEG_CONV2S2>(a, s);
        }
        if (bw_x) {  // (hipseg_conv3_dgrad_bnstats checked that the shape has a kernel with that epilogue)
            const int r16 = conv3_m16_rows(dtype, mode, C0, C1, N0, N1, B, H, W);
            HS_REQUIRE(r16 && !N1, "conv3_dgrad_bnstats: no kernel with the BatchNorm-backward epilogue takes this shape");
            return conv3_m16_launch(a, r16, s);
        }
        {
            if (const int r16 = conv3_m16_rows(dtype, mode, C0, C1, N0, N1, B, H, W)) return conv3_m16_launch(a, r16, s);
            // (the streaming kernel adds the bias in the ConvT forward only and never writes statistics rows: any other
            // request with a bias / statistics takes the GEMM kernels below -- ADVICE round 3)
            if (!dbg && !stats && (mode == HIPSEG_CONVT || !bias) &&
                convt_stream_applies(dtype, mode, C0, C1, N0, N1, B, H, W))
                return convt_stream_launch(a, mode, s);
        }
        if (a.vec_ok && !no_dma && buf_ok) {
            // ConvTranspose2d forward / data gradient as a one-tap GEMM on pixel-major 64-channel stages
            static const bool no_g1 = getenv("HIPSEG_NO_GEMM1") != nullptr;  // A/B switch
            // (K >= 256: with one or two 64-channel stages there is nothing to pipeline and the 144-KiB ring allows one
            // workgroup per CU -- measured 32 vs 28 us on dec4.up, 19 vs 18 on dec3.up)
            if (!no_g1 && !dbg && (mode == HIPSEG_CONVT || mode == HIPSEG_CONV2S2) && C1 == 0 && N1 == 0 && C0 % 64 == 0 &&
                a.N % 128 == 0 && a.Kp == C0 && N0 % 8 == 0 && (mode == HIPSEG_CONVT ? C0 : 4 * C0) >= 256)
                return mode == HIPSEG_CONVT ? launch_gemm1<HIPSEG_CONVT>(a, s) : launch_gemm1<HIPSEG_CONV2S2>(a, s);
            // 512-pixel tall tiles for 3x3 layers with enough of them to fill the chip (1 or 2 workgroups per CU)
            const bool no_tall = false;
            const long tall_wgs = (long)a.B * a.tiles_x * cdiv(H, 32) * a.ntn;
            // 128-wide tiles whose K splits into whole activation super-chunks inside one source tensor
            static const bool no_r64 = getenv("HIPSEG_NO_RING64") != nullptr;
            const bool tall = mode == HIPSEG_CONV3 && !no_tall && H >= 32 && bn >= 64 && tall_wgs >= (bn == 128 ? 256 : 512);
            // a 128-wide tiling that leaves half the CUs without a workgroup (256 channels at 32x32: 128 workgroups on 256
            // CUs): 64-wide tiles double the grid -- dec1.c0 59 -> 46 us, dec1.c1 and its data gradient 34 -> 28 us.  (With
            // exactly one 128-wide workgroup per CU the 64-wide form measured SLOWER: bott.c1 70 -> 78 us.)
            if (mode == HIPSEG_CONV3 && bn == 128 && (long)a.B * a.tiles_x * a.tiles_y * a.ntn * 2 <= a.ncu) {
                ConvArgs b = a;
                b.ntn = a.Np / 64;
                return launch_dma<HIPSEG_CONV3, 64, 16>(b, s);
            }
            if (mode == HIPSEG_CONV3 && bn == 128 && !no_r64 && !dbg && a.K == a.Kp) {
                if (a.Kp % 32 == 0 && (C1 == 0 || C0 % 32 == 0))
                    return tall ? launch_ring64<32, 4>(a, s) : launch_ring64<16, 4>(a, s);
            }
            if (mode == HIPSEG_CONV3 && !no_tall && H >= 32 && bn >= 64 && tall_wgs >= (bn == 128 ? 256 : 512)) {
                if (bn == 128) return launch_dma<HIPSEG_CONV3, 128, 32>(a, s);
                return launch_dma<HIPSEG_CONV3, 64, 32>(a, s);  // (32-wide tiles measured slower tall)
            }
            return launch_dma_mode(a, mode, bn, s);
        }
        return launch_mode<bf16>(a, mode, bn, s);
    }
    return launch_mode<float>(a, mode, bn, s);
}

extern "C" int hipseg_conv_igemm(int dtype, int mode, const void* in0, int C0, const void* in1, int C1,
                                 const void* wp, const float* bias, void* out0, int N0, void* out1, int N1,
                                 float* stats, int B, int H, int W, hipseg_stream_t stream) {
    return conv_igemm_impl(dtype, mode, in0, C0, in1, C1, wp, bias, nullptr, out0, N0, out1, N1, stats, B, H, W, stream);
}

// Data gradient of a 3x3 convolution whose INPUT was relu(bn(x)) (the second convolution of a ConvBlock,
// /root/reference/models/processing_blocks.py:43-48): out = d(loss)/d(relu(bn(x))) as hipseg_conv_igemm(CONV3, in = dY,
// wp = the data-gradient operand) computes it, and IN THE SAME KERNEL the BatchNorm-backward sums of that output,
// [sum g | sum g * xhat] with g = out where x * scale + shift > 0, xhat = (x - mean) * invstd -- what
// hipseg_bn_bwd_reduce would produce from a second pass over `out` and `x`.  `partial` receives
// hipseg_conv3_dgrad_bnstats_rows() rows of [2][N] floats for hipseg_colsum_finalize(partial, rows, 2, N, sums, ...).
// bn = [mean | invstd | scale | shift], N floats each (the vectors hipseg_bn_finalize wrote in the forward pass).
extern "C" int hipseg_conv3_dgrad_bnstats_rows(int dtype, int C, int N, int B, int H, int W) {
    static const bool off = getenv("HIPSEG_NO_DGRAD_BNSTATS") != nullptr;  // A/B switch
    if (off) return 0;
    if (wstat_grid(dtype, HIPSEG_CONV3, C, 0, N, 0, B, H, W)) return 0;  // (that kernel has no such epilogue)
    const int r16 = conv3_m16_rows(dtype, HIPSEG_CONV3, C, 0, N, 0, B, H, W);
    return r16 ? conv3_m16_stats_rows(r16, N, B, H, W) : 0;
}

extern "C" int hipseg_conv3_dgrad_bnstats(int dtype, const void* dy, int C, const void* wp, void* out, int N, const void* x,
                                          const float* bn, float* partial, int B, int H, int W, hipseg_stream_t stream) {
    HS_REQUIRE(x && bn && partial, "conv3_dgrad_bnstats: null operand");
    HS_REQUIRE(hipseg_conv3_dgrad_bnstats_rows(dtype, C, N, B, H, W) > 0,
               "conv3_dgrad_bnstats: unsupported shape (ask hipseg_conv3_dgrad_bnstats_rows first)");
    return conv_igemm_impl(dtype, HIPSEG_CONV3, dy, C, nullptr, 0, wp, nullptr, nullptr, out, N, nullptr, 0, partial, B, H, W,
                           stream, x, bn);
}

// The same for the data gradient of ConvTranspose2d(k2, s2) (/root/reference/models/processing_blocks.py:102: its input is
// the previous ConvBlock's activated output, so its data gradient is that block's dout): dx = CONV2S2(dy, wp) on the H x W
// grid and, in the same kernel, the BatchNorm-backward rows [sum g | sum g * xhat] of dx against x = that block's second
// pre-normalisation tensor (Cin channels, H x W) and its bn vectors.  rows() = 0: no kernel with the epilogue takes the shape.
extern "C" int hipseg_convT_dgrad_bnstats_rows(int dtype, int Cout, int Cin, int B, int H, int W) {
    int rows;
    convt_dgrad_bws_kernel(dtype, Cout, Cin, B, H, W, &rows);
    return rows;
}

extern "C" int hipseg_convT_dgrad_bnstats(int dtype, const void* dy, int Cout, const void* wp, void* dx, int Cin, const void* x,
                                          const float* bn, float* partial, int B, int H, int W, hipseg_stream_t stream) {
    HS_REQUIRE(x && bn && partial, "convT_dgrad_bnstats: null operand");
    HS_REQUIRE(hipseg_convT_dgrad_bnstats_rows(dtype, Cout, Cin, B, H, W) > 0,
               "convT_dgrad_bnstats: unsupported shape (ask hipseg_convT_dgrad_bnstats_rows first)");
    return conv_igemm_impl(dtype, HIPSEG_CONV2S2, dy, Cout, nullptr, 0, wp, nullptr, nullptr, dx, Cin, nullptr, 0, partial, B, H,
                           W, stream, x, bn);
}

// conv3x3 over relu(in * scale[c] + shift[c]) (zero-padded): the BatchNorm + ReLU of the PREVIOUS layer applied in this
// convolution's load path, so that layer's activated tensor is never written or read (the second
// convolution of a ConvBlock at the full-resolution levels, /root/reference/models/processing_blocks.py:44-46).  Same
// arithmetic as hipseg_bn_relu_apply followed by hipseg_conv_igemm: bit-identical results.
extern "C" int hipseg_conv3_bnrelu_in_applies(int dtype, int C, int N, int B, int H, int W) {
    static const bool off = getenv("HIPSEG_NO_BN_ON_LOAD") != nullptr;  // A/B switch
    if (off || (C != 64 && C != 32)) return 0;
    return wstat_grid(dtype, HIPSEG_CONV3, C, 0, N, 0, B, H, W) ? 1 : 0;
}

extern "C" int hipseg_conv3_bnrelu_in(int dtype, const void* in, int C, const float* scale, const float* shift, const void* wp,
                                      const float* bias, void* out, int N, float* stats, int B, int H, int W,
                                      hipseg_stream_t stream) {
    HS_REQUIRE(scale && shift, "conv3_bnrelu_in: scale and shift are required");
    HS_REQUIRE(hipseg_conv3_bnrelu_in_applies(dtype, C, N, B, H, W),
               "conv3_bnrelu_in: unsupported shape (ask hipseg_conv3_bnrelu_in_applies first)");
    return conv_igemm_impl(dtype, HIPSEG_CONV3, in, C, nullptr, 0, wp, bias, nullptr, out, N, nullptr, 0, stats, B, H, W, stream,
                           nullptr, nullptr, scale, shift);
}

// Inference form of conv3x3 -> BatchNorm(running statistics) -> ReLU in ONE kernel: the per-channel affine is applied to
// the fp32 accumulators in the epilogue, so neither the pre-normalisation tensor nor a second pass over it exists.
extern "C" int hipseg_conv_affine_relu(int dtype, const void* in0, int C0, const void* in1, int C1, const void* wp,
                                       const float* scale, const float* shift, void* out, int N, int B, int H, int W,
                                       hipseg_stream_t stream) {
    HS_REQUIRE(scale && shift, "conv_affine_relu: scale and shift are required");
    return conv_igemm_impl(dtype, HIPSEG_CONV3, in0, C0, in1, C1, wp, shift, scale, out, N, nullptr, 0, nullptr, B, H, W,
                           stream);
}
