// 3x3 convolution (forward and data gradient) for the >= 128-channel layers of the U-Nets, bf16, gfx950.
// Replaces nn.Conv2d(Cin, Cout, 3, padding=1) forward / its autograd data gradient
// (/root/reference/models/processing_blocks.py:43,46) for N % 128 == 0 output channels and K % 32 == 0 input channels.
//
// Why a second implicit-GEMM kernel next to conv_igemm.hip's ring kernel (which measured 0.35-0.44 of the bf16 MFMA
// peak, limited by the lock-step of its 8 waves at every 16-channel weight chunk and by an exposed LDS-transpose
// epilogue, with exactly one workgroup per CU):
//   * SWAPPED operands on v_mfma_f32_16x16x32_bf16: output CHANNELS are the MFMA rows (A operand = weights), output
//     PIXELS the MFMA columns (B operand = activations).  A lane then holds 4 consecutive channels of ONE pixel per
//     accumulator block; with the channel permutation below, 8 consecutive channels per pixel over the wave's two
//     blocks = one 16-byte NHWC store straight from registers.  No LDS transpose, no epilogue barrier.
//   * The weights never touch LDS: every wave owns 32 output channels and reads its two A fragments per (tap,
//     32-channel stage) with buffer_load_dwordx4 from the packed operand ([tap][K/8][Np][8], L2/L1 resident) two taps
//     ahead.  No wave of a workgroup loads what another one loads, nothing is written to LDS for them, and there is
//     no per-chunk barrier: the only barrier left is the activation stage's (one per 9 taps x THT x 2 MFMAs).
//   * LDS holds only a 3-slot ring of halo tiles ([octet][pixel][8] images filled by LDS-DMA, one octet per wave):
//     72 KiB per 4-wave workgroup, so TWO independent workgroups share a CU: one's prologue / epilogue / barrier
//     waits run under the other's MFMAs.
//   * The 16x16x32 shape holds a higher clock than 32x32x16 under load (MI355X_MICROARCH.md, DVFS give-back item 7).
// Workgroup tile: THT x 16 output pixels (THT = 16 or 8 tile rows) x 128 channels; wave tile: all pixels x 32 channels.
// Per (tap, stage) and wave: 2 weight loads, THT ds_read_b128, 2 * THT MFMAs.
// NB = 2 (round 4): a 64-channel workgroup tile for layers whose output channel count is a multiple of 64 but not of 128
// (128 -> 64 at 128 x 128 and its mirror data gradient in the U-Net; until round 4 they ran on conv_igemm.hip's ring
// kernel at 0.30 of the MFMA peak): the four waves are 2 channel blocks x 2 halves of the tile's pixel rows, so a wave
// still owns 32 channels and issues one ds_read_b128 per two MFMAs.
#include <stdlib.h>

#include "common.h"
#include "conv_args.h"

#ifdef M16_STAMP
// diagnostic build only (scripts/build_variant.sh stamp conv3_m16.hip -DM16_STAMP): per-workgroup s_memtime /
// s_memrealtime stamps into a buffer nothing else reads (MI355X_MICROARCH.md, DVFS give-back item 6)
__device__ unsigned long long g_m16_stamp[16384 * 8];
extern "C" int hipseg_debug_m16_stamps(void* host, int nwg) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_m16_stamp), (size_t)nwg * 8 * sizeof(unsigned long long)) == hipSuccess ? 0 : 1;
}
#define STAMP(i) do { if (tid == 0 && blockIdx.x < 16384) g_m16_stamp[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#define STAMP_RT(i) do { if (tid == 0 && blockIdx.x < 16384) g_m16_stamp[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define STAMP(i)
#define STAMP_RT(i)
#endif

namespace {

template <int THT>
struct M16Geo {
    static constexpr int NW = 4, BN = 128, KS = 32, SO = 4, NT = 9;
#ifndef M16_PF
#define M16_PF 6
#endif
    static constexpr int D = 2;        // weight fragments are loaded D taps ahead
    static constexpr int PF = M16_PF;  // activation fragments in flight (ds_read_b128 issued PF pixel blocks ahead)
    static constexpr int HW = 16 + 2, HH = THT + 2, NPIX = HH * HW;
    static constexpr int NGRP = (NPIX + 63) / 64, NPIXA = NGRP * 64;  // 64-pixel DMA groups of the halo tile
    static constexpr int A_BYTES = SO * NPIXA * 16, NSLOT = 3;
    static constexpr int LDS = NSLOT * A_BYTES;
    static constexpr int NPW = NGRP;  // LDS-DMA pieces per wave and stage (wave w stages octet w of every group)
    static_assert(NPW <= NT, "at most one piece per tap");
    static_assert(NT % (D + 1) == 0, "the weight register ring is indexed statically");
    static_assert(NPIXA % 16 == 0, "octet planes keep the bank phase (conflict-free ds_read_b128)");
    static_assert(2 * D + 2 * NT + NPW <= 63, "vmcnt is a 6-bit counter");
};

// EPI: 0 = conv + bias (+ batch statistics rows when p.stats), 1 = inference epilogue relu(conv * scale + shift),
//      2 = data gradient whose output is the dy of a BatchNorm+ReLU: besides storing it, reduce the BatchNorm-backward
//          sums [sum g | sum g * xhat] (g = dy where relu(bn(x)) > 0) of the tile into p.stats (see the epilogue)
template <int THT, int EPI, int NB = 4>
__global__ __launch_bounds__(256, 2) void conv3_m16_kernel(ConvArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)  // buffer-resource builtins exist in the device pass only
    typedef M16Geo<THT> G;
    typedef bf16 T;
    constexpr bool AFF = EPI == 1, BWS = EPI == 2;
    constexpr int RS = 4 / NB, THW = THT / RS;  // pixel-row splits of the tile over the waves, tile rows per wave
    static_assert(NB == 4 || NB == 2, "128- or 64-channel workgroup tiles");
    constexpr int NT = G::NT, D = G::D, PF = G::PF, HW = G::HW, NPIX = G::NPIX, NPIXA = G::NPIXA, NGRP = G::NGRP;
    constexpr int A_BYTES = G::A_BYTES, NPW = G::NPW;
    typedef __attribute__((address_space(3))) void lds_void;
    constexpr unsigned OOB_LANE = 0x80000000u;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    STAMP(0);
    STAMP_RT(1);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;  // MFMA row/column index and k-octet (operands) / row quad (results)
    const int bid = xcd_block(blockIdx.x, p.xcd);
    const int ntile = bid % p.ntn, mtile = bid / p.ntn;
    const int tx = mtile % p.tiles_x;
    const int ty = (mtile / p.tiles_x) % p.tiles_y;
    const int img = mtile / (p.tiles_x * p.tiles_y);
    const int wcol = wave % NB, rsp = wave / NB;  // the wave's 32-channel block and pixel-row part of the workgroup tile
    const int y0 = ty * THT, x0 = tx * 16, nw = ntile * (32 * NB) + wcol * 32;
    const int nst = p.Kp / G::KS;

    const unsigned bytes0 = (unsigned)((size_t)p.B * p.Hi * p.Wi * p.C0 * sizeof(T));
    const unsigned bytes1 = (unsigned)((size_t)p.B * p.Hi * p.Wi * p.C1 * sizeof(T));
    const __amdgpu_buffer_rsrc_t r_in0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in0), 0, (int)bytes0, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_in1 =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in1 ? p.in1 : p.in0), 0, (int)bytes1, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_w = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(p.wp), 0, (int)((size_t)NT * p.Kp * p.Np * sizeof(T)), 0x00020000);

    // ---- activations: a piece = one octet of halo pixels 64 j .. 64 j + 63 (1 KiB of the [octet][pixel][8] image).
    // Per-lane byte offsets against either source; border / padding pixels are out of range (the buffer bounds check
    // writes zeros to LDS).  Wave w stages octet w of every group.
    unsigned avo0[NGRP], avo1[NGRP];
    const int my_oct = wave;
#pragma unroll
    for (int j = 0; j < NGRP; ++j) {
        const int pix = j * 64 + lane, hy = pix / HW, hx = pix - hy * HW;
#ifdef HIPSEG_ABLATE
        // bit 16: every workgroup stages the halo tile of ONE of 8 tile positions (L2-resident activations)
        const int ay0 = (p.debug & 16) ? 16 : y0, ax0 = (p.debug & 16) ? 16 * (blockIdx.x & 1) : x0;
        const int aimg = (p.debug & 16) ? (blockIdx.x & 3) : img;
        const int iy = ay0 - 1 + hy, ix = ax0 - 1 + hx;
        const bool ok = pix < NPIX && iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi;
        const long apix = ((long)aimg * p.Hi + iy) * p.Wi + ix;
#else
        const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
        const bool ok = pix < NPIX && iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi;
        const long apix = ((long)img * p.Hi + iy) * p.Wi + ix;
#endif
        avo0[j] = ok ? (unsigned)(apix * p.C0 * (long)sizeof(T) + my_oct * 16) : OOB_LANE;
        avo1[j] = ok ? (unsigned)(apix * p.C1 * (long)sizeof(T) + my_oct * 16) : OOB_LANE;
    }
    // branch-free: the source tensor of a stage is a scalar select (a branch here also made the compiler's own vmcnt
    // bookkeeping for the weight loads pessimistic: it waited one tap early)
    auto pieceA = [&](int slot, int stage, int j, int oct) {
        const int c0 = stage * G::KS;
        bool live = stage < nst;  // pieces of stages past the end keep the VMEM count static: no lane fetches
#ifdef HIPSEG_ABLATE
        live = live && !(p.debug & 1);
#endif
        const bool second = c0 >= p.C0;
        lds_void* dst = (lds_void*)(smem + slot * A_BYTES + (oct * NPIXA + j * 64) * 16);
        const __amdgpu_buffer_rsrc_t r = second ? r_in1 : r_in0;
        const unsigned so = (unsigned)(second ? c0 - p.C0 : c0) * 2u;
        const unsigned vo = live ? (second ? avo1[j] : avo0[j]) : OOB_LANE;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, dst, 16, vo, so, 0, 0);
    };

    // ---- weights: A fragment of block j = MFMA rows i <-> channels nw + (i >> 2) * 8 + j * 4 + (i & 3), so that the
    // lane holding rows 4 lg .. 4 lg + 3 of both blocks holds channels nw + 8 lg .. + 7: one 16-byte store per pixel.
    const int kgp = p.Kp / 8;
    unsigned wvo[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int ch = nw + (li >> 2) * 8 + j * 4 + (li & 3);
        wvo[j] = (unsigned)(((size_t)lg * p.Np + ch) * 16);
    }
    bf16x8 W[D + 1][2];
    auto loadW = [&](int ring, int stage, int tap) {
        unsigned so = (unsigned)((tap * kgp + stage * 4) * p.Np) * 16u;
#ifdef HIPSEG_ABLATE
        if (p.debug & 2) so = 0;  // every fragment from the same 2 KiB: L1 resident
#endif
#pragma unroll
        for (int j = 0; j < 2; ++j)
            W[ring][j] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r_w, wvo[j], so, 0));
    };

    // epilogue constants, fetched before the main loop (a dependent global load at the head of the epilogue is exposed)
    const int ch0 = nw + lg * 8;
    float bv[8], sc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        bv[k] = p.bias ? p.bias[ch0 + k] : 0.f;
        sc[k] = AFF ? p.post_scale[ch0 + k] : 1.f;
    }

    f32x4 acc[2][THW];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int b = 0; b < THW; ++b) acc[j][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    // B fragment of pixel block b (tile row rsp * THW + b) at tap (ky, kx): halo pixel (row + ky) * HW + kx + li of octet
    // plane lg
    const unsigned abase = (unsigned)((lg * NPIXA + rsp * THW * HW + li) * 16);

    // ---- prologue: stages 0 and 1, the weights of the first D taps
#pragma unroll
    for (int j = 0; j < NPW; ++j) pieceA(0, 0, j, wave);
#pragma unroll
    for (int j = 0; j < NPW; ++j) pieceA(1, 1, j, wave);
#pragma unroll
    for (int t = 0; t < D; ++t) loadW(t, 0, t);

    int slot = 0;
    for (int s = 0; s < nst; ++s) {
        // VMEM operations retire in issue order, so "this wave's pieces of stage s have landed" is a counted wait: the
        // operations issued after the last of them are (steady state) 2 (NT - NPW) weight loads of stage s - 2 and
        // 2 NT + NPW operations of stage s - 1; stage 1 has 2 D + 2 NT + NPW (fewer: used for every s >= 1).
        if (s == 0)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW + 2 * D) : "memory");
        else
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * D + 2 * NT + NPW) : "memory");
        __builtin_amdgcn_s_barrier();  // everyone's pieces of stage s landed; everyone left the slot stage s + 2 will fill
        __builtin_amdgcn_sched_barrier(0);
#ifdef M16_STAMP
        if (s == 0) STAMP(2);
#endif
        const unsigned char* rd = smem + abase + slot * A_BYTES;
        const int nslot = slot == 0 ? 2 : slot - 1;  // (slot + 2) % 3
        const int wst = s + 1 < nst ? s + 1 : s;       // weights of the wrapped taps (clamped: a harmless re-read)
        auto rdA = [&](int idx) {
            const int tap = idx / THW, b = idx % THW;
            return *reinterpret_cast<const bf16x8*>(rd + ((b + tap / 3) * HW + tap % 3) * 16);
        };
        bf16x8 af[PF];
#pragma unroll
        for (int i = 0; i < PF; ++i) af[i] = rdA(i);
#pragma unroll
        for (int tap = 0; tap < NT; ++tap) {
            if (tap + D < NT)
                loadW((tap + D) % (D + 1), s, tap + D);
            else
                loadW((tap + D) % (D + 1), wst, tap + D - NT);
            if (tap < NPW) pieceA(nslot, s + 2, tap, wave);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int b = 0; b < THW; ++b) {
                const int idx = tap * THW + b;
#ifdef HIPSEG_ABLATE
                if (!(p.debug & 4))
#endif
                {
                    acc[0][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W[tap % (D + 1)][0], af[idx % PF], acc[0][b], 0, 0, 0);
                    acc[1][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W[tap % (D + 1)][1], af[idx % PF], acc[1][b], 0, 0, 0);
                }
                if (idx + PF < NT * THW) af[idx % PF] = rdA(idx + PF);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        slot = slot == 2 ? 0 : slot + 1;
    }

    STAMP(3);
    // ---- epilogue, straight from the accumulators: lane (li, lg) holds pixel column li of every tile row b and
    // channels ch0 .. ch0 + 7 (block j, register e -> ch0 + 4 j + e)
#ifdef HIPSEG_ABLATE
    if (p.debug & 8) return;
#endif
    T* dst;
    int stride;
    if (ch0 < p.N0) {
        dst = reinterpret_cast<T*>(p.out0) + ch0;
        stride = p.N0;
    } else {
        dst = reinterpret_cast<T*>(p.out1) + (ch0 - p.N0);
        stride = p.N1;
    }
    float ssum[8], ssq[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        ssum[k] = 0.f;
        ssq[k] = 0.f;
    }
    const int x = x0 + li;
    if constexpr (BWS) {
        // The tile just computed is dy of relu(bn(xr)) (xr = p.bw_x, the convolution output that BatchNorm normalised;
        // same NHWC shape as this output, single destination).  bn_bwd_reduce_kernel (bn.hip) would re-read both
        // tensors to form sum g and sum g * xhat with g = dy where xr * scale + shift > 0, xhat = (xr - mean) * invstd:
        // formed here from the bf16-rounded output (what that kernel would read back) and one 16-byte read of xr per
        // store.  Rows go to p.stats in the layout hipseg_colsum_finalize(nblk = tiles, rows = 2) sums.
        const float* bn = p.bw_bn + ch0;
        float mn[8], is[8], s2[8], sh[8];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(bn + 4 * h);
            const f32x4 c = *reinterpret_cast<const f32x4*>(bn + p.N + 4 * h);
            const f32x4 d = *reinterpret_cast<const f32x4*>(bn + 2 * (size_t)p.N + 4 * h);
            const f32x4 e = *reinterpret_cast<const f32x4*>(bn + 3 * (size_t)p.N + 4 * h);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                mn[4 * h + k] = a[k];
                is[4 * h + k] = c[k];
                s2[4 * h + k] = d[k];
                sh[4 * h + k] = e[k];
            }
        }
        const T* rx = reinterpret_cast<const T*>(p.bw_x) + ch0;
        constexpr int RB = THW < 8 ? THW : 8;  // rows of xr in flight per lane
#pragma unroll
        for (int b0 = 0; b0 < THW; b0 += RB) {
            bf16x8 xr[RB];
#pragma unroll
            for (int r = 0; r < RB; ++r) {
                const int y = y0 + rsp * THW + b0 + r;
                const bool in = y < p.H && x < p.W;
                const long pix = in ? (long)(img * p.H + y) * p.W + x : 0;
                xr[r] = *reinterpret_cast<const bf16x8*>(rx + pix * p.N);
            }
#pragma unroll
            for (int r = 0; r < RB; ++r) {
                const int b = b0 + r, y = y0 + rsp * THW + b;
                const bool in = y < p.H && x < p.W;
                bf16x8 o;
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int k = j * 4 + e;
                        o[k] = (T)(acc[j][b][e] + bv[k]);
                        const float xv = (float)xr[r][k];
                        const float g = (in && xv * s2[k] + sh[k] > 0.f) ? (float)o[k] : 0.f;
                        ssum[k] += g;
                        ssq[k] += g * ((xv - mn[k]) * is[k]);
                    }
                if (in) *reinterpret_cast<bf16x8*>(dst + ((long)(img * p.H + y) * p.W + x) * stride) = o;
            }
        }
    } else {
#pragma unroll
    for (int b = 0; b < THW; ++b) {
        const int y = y0 + rsp * THW + b;
        const bool in = y < p.H && x < p.W;
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = j * 4 + e;
                const float v = AFF ? fmaxf(fmaf(acc[j][b][e], sc[k], bv[k]), 0.f) : acc[j][b][e] + bv[k];
                o[k] = (T)v;
                if (in) {
                    ssum[k] += v;
                    ssq[k] += v * v;
                }
            }
        if (in) *reinterpret_cast<bf16x8*>(dst + ((long)(img * p.H + y) * p.W + x) * stride) = o;
    }
    }
    if (p.stats) {
        // one statistics row per (workgroup tile, pixel-row part); the 16 pixel columns of a channel sit in the 16 lanes
        // of a row quad
#pragma unroll
        for (int k = 0; k < 8; ++k) {
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) {
                ssum[k] += __shfl_xor(ssum[k], o, 64);
                ssq[k] += __shfl_xor(ssq[k], o, 64);
            }
        }
        if (li == 0) {
            float* row = p.stats + ((size_t)mtile * RS + rsp) * 2 * p.N + ch0;
            *reinterpret_cast<f32x4*>(row) = f32x4{ssum[0], ssum[1], ssum[2], ssum[3]};
            *reinterpret_cast<f32x4*>(row + 4) = f32x4{ssum[4], ssum[5], ssum[6], ssum[7]};
            *reinterpret_cast<f32x4*>(row + p.N) = f32x4{ssq[0], ssq[1], ssq[2], ssq[3]};
            *reinterpret_cast<f32x4*>(row + p.N + 4) = f32x4{ssq[4], ssq[5], ssq[6], ssq[7]};
        }
    }
#ifdef M16_STAMP
    STAMP(4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    STAMP(5);
    STAMP_RT(6);
    if (tid == 0 && blockIdx.x < 16384) g_m16_stamp[blockIdx.x * 8 + 7] = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // XCC_ID
#endif
#else
    (void)p;
#endif
}

template <int THT, int NB>
int launch_m16(const ConvArgs& a0, hipStream_t s) {
    typedef M16Geo<THT> G;
    ConvArgs a = a0;
    a.tiles_y = cdiv(a.H, THT);
    a.ntn = a.N / (32 * NB);  // (N, not the packed operand's Np: 192 channels are packed as 256 columns, the last 64 unused)
    const long grid = (long)a.B * a.tiles_x * a.tiles_y * a.ntn;
    a.xcd = (grid % 8 == 0 && grid >= 64) ? (int)(grid / 8) : 0;
    const int threads = 256;
    if (a.bw_x) {
        if (int rc = hs_set_max_lds(reinterpret_cast<const void*>(&conv3_m16_kernel<THT, 2, NB>), (size_t)G::LDS)) return rc;
        hipLaunchKernelGGL((conv3_m16_kernel<THT, 2, NB>), dim3((unsigned)grid), dim3(threads), G::LDS, s, a);
    } else if (a.post_scale) {
        if (int rc = hs_set_max_lds(reinterpret_cast<const void*>(&conv3_m16_kernel<THT, 1, NB>), (size_t)G::LDS)) return rc;
        hipLaunchKernelGGL((conv3_m16_kernel<THT, 1, NB>), dim3((unsigned)grid), dim3(threads), G::LDS, s, a);
    } else {
        if (int rc = hs_set_max_lds(reinterpret_cast<const void*>(&conv3_m16_kernel<THT, 0, NB>), (size_t)G::LDS)) return rc;
        hipLaunchKernelGGL((conv3_m16_kernel<THT, 0, NB>), dim3((unsigned)grid), dim3(threads), G::LDS, s, a);
    }
    HS_LAUNCH_CHECK("conv3_m16");
    return HIPSEG_OK;
}

}  // namespace

// Tile rows of the 16x16x32 kernel for this shape (16 or 8), 0 = the kernel does not take the shape.
int conv3_m16_rows(int dtype, int mode, int C0, int C1, int N0, int N1, int B, int H, int W) {
    static const bool off = getenv("HIPSEG_NO_M16") != nullptr || getenv("HIPSEG_NO_DMA") != nullptr;
    if (off || dtype != HIPSEG_BF16 || mode != HIPSEG_CONV3) return 0;
    const int K = C0 + C1, N = N0 + N1;
    static const bool no64 = getenv("HIPSEG_NO_M16_BN64") != nullptr;  // A/B switch: 64-channel tiles to the ring kernel
    if (N % 64 || (N % 128 && no64) || K % 32 || (C1 && C0 % 32) || N0 % 8 || N1 % 8) return 0;
    // buffer addressing: byte offsets below 2^30 (2^31 marks an out-of-range lane)
    const size_t in_bytes = (size_t)B * H * W * (size_t)(C0 > C1 ? C0 : C1) * 2, w_bytes = (size_t)9 * K * N * 2;
    if (in_bytes > ((size_t)1 << 30) || w_bytes > ((size_t)1 << 30)) return 0;
    // 16-row tiles when they give every CU its two workgroups, else 8-row tiles (twice the grid)
    const long wgs16 = (long)B * cdiv(W, 16) * cdiv(H, 16) * (N % 128 ? N / 64 : N / 128);
    static const char* force = getenv("HIPSEG_M16_ROWS");
    if (force) return atoi(force) == 8 ? 8 : 16;
    return wgs16 >= 2 * (long)device_cus() ? 16 : 8;
}

// statistics rows of a launch: one per tile of `rows` x 16 pixels, two (the tile's pixel-row halves) with 64-channel tiles
int conv3_m16_stats_rows(int rows, int N, int B, int H, int W) { return B * cdiv(W, 16) * cdiv(H, rows) * (N % 128 ? 2 : 1); }

int conv3_m16_launch(const ConvArgs& a, int rows, hipStream_t s) {
    if (a.N % 128) return rows == 16 ? launch_m16<16, 2>(a, s) : launch_m16<8, 2>(a, s);
    return rows == 16 ? launch_m16<16, 4>(a, s) : launch_m16<8, 4>(a, s);
}
