// Weight-gradient GEMM for gfx950 (MFMA):  G[tap][u][v] = sum_pixels P[tap(pixel)][u] * Q[pixel][v]
//
// The reduction dimension is the pixel index, which is the OUTER dimension of NHWC tensors, so
// both MFMA operands are needed "k-major".  The LDS images keep the global [pixel][channel] layout
// (coalesced staging, any 3x3 tap is just a row offset into the halo image) and
//   bf16: fragments come from ds_read_b64_tr_b16 (hardware transposed read, 4 pixels x 16 channels
//         per 16-lane group), feeding v_mfma_f32_32x32x16_bf16;
//   f32 : v_mfma_f32_32x32x2_f32 takes one float per lane, read straight from the image.
// A workgroup owns one (u-tile, v-tile) pair for ALL taps (9 accumulator tiles per wave for a 3x3
// conv) and walks a strided subset of the 16x16 pixel tiles; partial results go to fp32 slabs that a
// second kernel sums in fixed order into the parameter's native layout (deterministic, no atomics).
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int TW = 16;
#ifndef NPW_PPR
#define NPW_PPR 4
#endif

template <typename T>
struct WT;
template <>
struct WT<bf16> {
    static constexpr int TH = 16, UC = 64, VC = 64, WU = 2, WV = 2, SWZ = 1;
};
template <>
struct WT<float> {
    static constexpr int TH = 8, UC = 32, VC = 128, WU = 1, WV = 4, SWZ = 0;
};

// bf16 images: 64-channel (128 B) rows; the 64-byte half a transposed read touches is XOR-swapped on
// every second pixel pair so that the 4 pixel rows of one ds_read_b64_tr_b16 hit 4 different
// 64-byte bank quarters (conflict-free without padding).
template <typename T>
__device__ __forceinline__ int swz(int pix) {
    return WT<T>::SWZ ? ((pix >> 1) & 1) << 5 : 0;
}

struct WgArgs {
    const void* p0;
    const void* p1;
    const void* q;
    float* slabs;
    int CU0, CU1, CU, CV, CUp, CVp;
    int B, H, W;     // Q pixel grid
    int PH, PW;      // P spatial dims
    int ps, pa, pb;  // P pixel = (ps*y + pa, ps*x + pb)   (NT == 1 modes)
    int tiles_x, tiles_y, ntiles, S, UT, VT;
    int vec_ok_p, vec_ok_q;
    // ConvTranspose2d weight gradient as ONE GEMM: P "pixel" (y,x) gathers its 4*Cout channels from the two
    // dY rows 2y, 2y+1 (each row holds (b, co) = 2*Cout contiguous values at column 2x).  0 = plain tensor.
    int convt_cout;
    int debug;  // ablation bits (HIPSEG_WGRAD_DEBUG): 1 skip P staging, 2 skip Q staging, 4 skip MFMA, 8 skip slab store
    int xcd;    // XCD-aware workgroup order: 0 off, else grid / 8 (see xcd_block)
    // BatchNorm + ReLU applied to P on LOAD (hipseg_conv_wgrad_bnrelu_p): P holds the pre-normalisation tensor and the
    // gradient is taken against relu(P * p_scale[u] + p_shift[u]) (zero-padded).  NULL = off.
    const float* p_scale;
    const float* p_shift;
};

// Workgroups are dealt round-robin to the 8 XCDs (each with a private L2).  Give every XCD a CONTIGUOUS run of
// logical workgroup ids instead, so the (u-tile, v-tile) workgroups of one pixel split -- which read the same
// pixels' channel slices, each slice UT or VT times -- share an L2 (guide T1; needs grid % 8 == 0).
__device__ __forceinline__ int xcd_block(int bid, int cpx) { return cpx ? (bid & 7) * cpx + (bid >> 3) : bid; }

// element offset of channel c (first of an aligned 8/4-vector) of P pixel (img, gy, gx)
__device__ __forceinline__ long p_offset(const WgArgs& a, int img, int gy, int gx, int c, const void*& base) {
    if (a.convt_cout) {
        const int two = 2 * a.convt_cout;
        const int ra = c / two, rem = c - ra * two;
        base = a.p0;
        return (((long)img * a.PH + 2 * gy + ra) * a.PW + 2 * gx) * a.convt_cout + rem;
    }
    const long pixoff = ((long)img * a.PH + a.ps * gy + a.pa) * a.PW + a.ps * gx + a.pb;
    if (c < a.CU0) {
        base = a.p0;
        return pixoff * a.CU0 + c;
    }
    base = a.p1;
    return pixoff * a.CU1 + (c - a.CU0);
}

template <typename T, int NT>
__global__ __launch_bounds__(256) void wgrad_kernel(WgArgs a) {
    constexpr int TH = WT<T>::TH, UC = WT<T>::UC, VC = WT<T>::VC, WU = WT<T>::WU, WV = WT<T>::WV;
    constexpr int PP = UC, QP = VC;  // LDS row pitch (elements)
    constexpr int HALO = NT == 9 ? 1 : 0;
    constexpr int PHH = TH + 2 * HALO, PHW = TW + 2 * HALO, NPP = PHH * PHW, NPQ = TH * TW;
    constexpr int VEC = VecOf<T>::N;
    typedef typename VecOf<T>::type vec_t;
    static_assert(WU * WV == 4 && UC == 32 * WU && VC == 32 * WV, "one 32x32 block per wave");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T* sP = reinterpret_cast<T*>(smem);  // [NPP][PP]
    T* sQ = sP + NPP * PP;               // [NPQ][QP]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wu = wave / WV, wv = wave % WV;

    const int bid = xcd_block(blockIdx.x, a.xcd);
    const int vt = bid % a.VT;
    const int ut = (bid / a.VT) % a.UT;
    const int s = bid / (a.VT * a.UT);
    const int u0 = ut * UC, v0 = vt * VC;

    const T* p0 = reinterpret_cast<const T*>(a.p0);
    const T* p1 = reinterpret_cast<const T*>(a.p1);
    const T* q = reinterpret_cast<const T*>(a.q);

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

    for (int tile = s; tile < a.ntiles; tile += a.S) {
        const int tx = tile % a.tiles_x;
        const int ty = (tile / a.tiles_x) % a.tiles_y;
        const int img = tile / (a.tiles_x * a.tiles_y);
        const int y0 = ty * TH, x0 = tx * TW;
        __syncthreads();
        // ---------------- stage P (halo image for the 3x3 case), UC channels
        {
            constexpr int CVP = UC / VEC, NCELL = NPP * CVP;
#pragma unroll 4
            for (int cell = tid; cell < NCELL; cell += 256) {
                const int pix = cell / CVP, cv = cell % CVP;
                const int hy = pix / PHW, hx = pix % PHW;
                const int gy = y0 + hy - HALO, gx = x0 + hx - HALO;  // Q-grid coordinates
                const int c = u0 + cv * VEC;
                vec_t v;
#pragma unroll
                for (int e = 0; e < VEC; ++e) v[e] = (T)0.f;
                if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W && c < a.CU) {
                    if (a.vec_ok_p) {
                        const void* base;
                        const long off = p_offset(a, img, gy, gx, c, base);
                        v = *reinterpret_cast<const vec_t*>(reinterpret_cast<const T*>(base) + off);
                    } else {
#pragma unroll
                        for (int e = 0; e < VEC; ++e) {
                            const int cc = c + e;
                            if (cc < a.CU) {
                                const void* base;
                                const long off = p_offset(a, img, gy, gx, cc, base);
                                v[e] = reinterpret_cast<const T*>(base)[off];
                            }
                        }
                    }
                }
                *reinterpret_cast<vec_t*>(sP + (size_t)pix * PP + ((cv * VEC) ^ swz<T>(pix))) = v;
            }
        }
        // ---------------- stage Q, VC channels
        {
            constexpr int CVQ = VC / VEC, NCELL = NPQ * CVQ;
#pragma unroll 4
            for (int cell = tid; cell < NCELL; cell += 256) {
                const int pix = cell / CVQ, cv = cell % CVQ;
                const int gy = y0 + pix / TW, gx = x0 + pix % TW;
                const int c = v0 + cv * VEC;
                vec_t v;
#pragma unroll
                for (int e = 0; e < VEC; ++e) v[e] = (T)0.f;
                if (gy < a.H && gx < a.W && c < a.CV) {
                    const long pixoff = ((long)img * a.H + gy) * a.W + gx;
                    if (a.vec_ok_q) {
                        v = *reinterpret_cast<const vec_t*>(q + pixoff * a.CV + c);
                    } else {
#pragma unroll
                        for (int e = 0; e < VEC; ++e)
                            if (c + e < a.CV) v[e] = q[pixoff * a.CV + c + e];
                    }
                }
                *reinterpret_cast<vec_t*>(sQ + (size_t)pix * QP + ((cv * VEC) ^ swz<T>(pix))) = v;
            }
        }
        __syncthreads();
        // ---------------- MFMA: k runs over the pixels of the tile
        if constexpr (sizeof(T) == 2) {
            // transposed-read lane geometry: 16-lane group g = lane>>4 covers channels 16*(g&1)..+15,
            // k half h = g>>1; lane 4q+p of the group addresses block row q, columns 4p..4p+3.
            const int tq = (lane & 15) >> 2, tp = lane & 3;
            const int chA = wu * 32 + 16 * ((lane >> 4) & 1) + 4 * tp;
            const int chB = wv * 32 + 16 * ((lane >> 4) & 1) + 4 * tp;
            typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
#pragma unroll 2
            for (int y = 0; y < TH; ++y) {
                const int kx0 = 8 * h + tq;  // pixel column (k) of this lane's first block row
                bf16x8 bfrag;
                {
                    const bf16* bq = reinterpret_cast<const bf16*>(sQ);
                    const int q0 = y * TW + kx0, q1 = q0 + 4;
                    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                        (lds_bf16x4*)(bq + (size_t)q0 * QP + (chB ^ swz<T>(q0))));
                    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                        (lds_bf16x4*)(bq + (size_t)q1 * QP + (chB ^ swz<T>(q1))));
                    bfrag = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                }
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int ky = NT == 9 ? t / 3 : 0, kx = NT == 9 ? t % 3 : 0;
                    const bf16* ap = reinterpret_cast<const bf16*>(sP);
                    const int a0 = (y + ky) * PHW + kx0 + kx, a1 = a0 + 4;
                    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                        (lds_bf16x4*)(ap + (size_t)a0 * PP + (chA ^ swz<T>(a0))));
                    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                        (lds_bf16x4*)(ap + (size_t)a1 * PP + (chA ^ swz<T>(a1))));
                    const bf16x8 afrag = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag, bfrag, acc[t], 0, 0, 0);
                }
            }
        } else {
            const float* fP = reinterpret_cast<const float*>(sP);
            const float* fQ = reinterpret_cast<const float*>(sQ);
#pragma unroll 1
            for (int y = 0; y < TH; ++y) {
#pragma unroll
                for (int kk = 0; kk < TW / 2; ++kk) {
                    const int x = 2 * kk + h;
                    const float bv = fQ[(size_t)(y * TW + x) * QP + wv * 32 + r];
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const int ky = NT == 9 ? t / 3 : 0, kx = NT == 9 ? t % 3 : 0;
                        const float av = fP[(size_t)((y + ky) * PHW + x + kx) * PP + wu * 32 + r];
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[t], 0, 0, 0);
                    }
                }
            }
        }
    }
    // ---------------- write this split's slab: [S][NT][CUp][CVp]
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
            const int u = u0 + wu * 32 + row, v = v0 + wv * 32 + r;
            a.slabs[(((size_t)s * NT + t) * a.CUp + u) * a.CVp + v] = acc[t][e];
        }
    }
}


// ---------------------------------------------------------------------------------------------------
// bf16 fast path: LDS images filled by LDS-DMA (global_load_lds_dwordx4) into a double-buffered ring,
// the next pixel tile streams in while the MFMAs of the current one run (one barrier per tile).
// A 1-KiB DMA piece = 8 pixel rows x 128 B; the XOR swizzle of the image is applied to the per-lane
// SOURCE chunk (LDS destination stays linear), out-of-image / padded channels read a zero word.
__device__ uint4 g_wg_zero16 = {0u, 0u, 0u, 0u};

template <int NT, int UB, int VB>
struct DmaGeo {
    static constexpr int TH = 16, NW = 8;
    static constexpr int KS = NW / (UB * VB), KH = TH / KS;  // pixel-row split ("k-split") and rows per wave
    static constexpr int HALO = NT == 9 ? 1 : 0;
    static constexpr int PHW = TW + 2 * HALO, NPP = (TH + 2 * HALO) * PHW, NPQ = TH * TW;
    static constexpr int PCH = 32 * UB, QCH = 32 * VB;                 // channels per LDS pixel row
    static constexpr int PPX = 512 / PCH, QPX = 512 / QCH;             // pixels per 1-KiB DMA piece
    static constexpr int NPC_P = (NPP + PPX - 1) / PPX, NPC_Q = NPQ / QPX;
    static constexpr int NPW_P = (NPC_P + NW - 1) / NW, NPW_Q = (NPC_Q + NW - 1) / NW, NPW = NPW_P + NPW_Q;
    static constexpr int PP_BYTES = NPC_P * 1024, Q_BYTES = NPC_Q * 1024, BUF = PP_BYTES + Q_BYTES;
    static constexpr int NR = KH + 2 * HALO;                           // halo rows a wave walks per tile
    // DMA pieces issued per halo row: everything goes out in the FIRST rows of the walk, so a piece has most of a
    // tile's MFMA time (plus DIST - 1 whole tiles) to land; spread over all rows the last pieces would be issued
    // just before the barrier that waits for them (measured: the whole HBM latency exposed per tile).
    static constexpr int PPR_ = NPW_PPR;
    static constexpr int PPR = PPR_ < (NPW + NR - 1) / NR ? (NPW + NR - 1) / NR : PPR_;
    static constexpr int NBUF_FIT = (160 * 1024 - 1024) / BUF;
    static constexpr int NBUF = NBUF_FIT >= 4 ? 4 : (NBUF_FIT >= 3 ? 3 : 2), DIST = NBUF - 1;
    static constexpr size_t RED_BYTES = (size_t)4 * NT * 16 * 64 * 4;  // 4 parked accumulator sets
    static constexpr size_t RING_BYTES = (size_t)NBUF * BUF + 1024;    // + 1-KiB sink for pad pieces
    static constexpr size_t LDS = RING_BYTES > RED_BYTES ? RING_BYTES : RED_BYTES;
    static_assert(NPQ % QPX == 0 && LDS <= 160 * 1024, "geometry");
};

// UB x VB = 32-channel blocks of the workgroup tile (1 or 2 each): 64u x 64v runs 2x2 waves x 2 pixel halves,
// a 32-channel side gives its waves to the pixel split instead (no MFMA work on padding channels) and its LDS
// image shrinks to 64 B per pixel (no swizzle needed), which also deepens the ring (up to 4 tiles in flight).
template <int NT, int UB, int VB, bool DBG>
__global__ __launch_bounds__(512, 1) void wgrad_dma_kernel(WgArgs a) {
    typedef bf16 T;
    typedef DmaGeo<NT, UB, VB> G;
    constexpr int KH = G::KH, HALO = G::HALO, PHW = G::PHW, NPP = G::NPP, PCH = G::PCH, QCH = G::QCH;
    constexpr int PP_BYTES = G::PP_BYTES, BUF = G::BUF, NPW_P = G::NPW_P, NPW = G::NPW, NR = G::NR, PPR = G::PPR;
    constexpr int NW = G::NW, NPC_P = G::NPC_P, NPC_Q = G::NPC_Q, NBUF = G::NBUF, DIST = G::DIST;
    typedef __attribute__((address_space(3))) void lds_void;
    typedef const __attribute__((address_space(1))) void glb_void;
    typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5;
    const int uv = wave % (UB * VB), wu = uv % UB, wv = uv / UB, kh = wave / (UB * VB);
    const int bid = xcd_block(blockIdx.x, a.xcd);
    const int vt = bid % a.VT;
    const int ut = (bid / a.VT) % a.UT;
    const int s = bid / (a.VT * a.UT);
    const int u0 = ut * PCH, v0 = vt * QCH;
    const char* zero = reinterpret_cast<const char*>(&g_wg_zero16);
    unsigned char* sink = smem + NBUF * BUF;

    // ---- staging addresses.  Everything that does not change from tile to tile is folded into per-lane constants
    // (source base incl. channel, pixel stride, channel validity); a piece then costs ~a dozen VALU ops.
    // A lane copies 16 B = 8 channels of one pixel of its wave's piece; for the 128-B (64-channel) rows the XOR
    // swizzle of the image is applied to the SOURCE chunk and depends on bit 1 of the pixel index only, which is
    // bit 1 of the lane's pixel row inside the piece.
    constexpr int LPP_P = PCH / 8, LPP_Q = QCH / 8;  // lanes per pixel
    const int prowP = lane / LPP_P, prowQ = lane / LPP_Q;
    const int lchP = (PCH == 64 ? ((lane % LPP_P) ^ (((prowP >> 1) & 1) << 2)) : (lane % LPP_P)) * 8;
    const int lchQ = (QCH == 64 ? ((lane % LPP_Q) ^ (((prowQ >> 1) & 1) << 2)) : (lane % LPP_Q)) * 8;
    const int cP = u0 + lchP, cQ = v0 + lchQ;
    const bool okcP = cP < a.CU && !(DBG && (a.debug & 1)), okcQ = cQ < a.CV && !(DBG && (a.debug & 2));
    const char* lbaseP;
    int strideP;  // bytes between consecutive P "pixels"
    int ps = a.ps, pa = a.pa;
    if (a.convt_cout) {  // gather of (row parity, column parity, co) from the two dY rows 2y, 2y+1 (see p_offset)
        const int two = 2 * a.convt_cout, ra = cP / two, rem = cP - ra * two;
        lbaseP = reinterpret_cast<const char*>(a.p0) + ((long)ra * a.PW * a.convt_cout + rem) * 2;
        strideP = a.convt_cout * 2;
        ps = 2;
        pa = 0;
    } else if (cP < a.CU0) {
        lbaseP = reinterpret_cast<const char*>(a.p0) + (long)cP * 2;
        strideP = a.CU0 * 2;
    } else {
        lbaseP = reinterpret_cast<const char*>(a.p1) + (long)(cP - a.CU0) * 2;
        strideP = a.CU1 * 2;
    }
    const int pb = a.convt_cout ? 0 : a.pb;
    const char* lbaseQ = reinterpret_cast<const char*>(a.q) + (long)cQ * 2;
    const int strideQ = a.CV * 2;

    // tile walk tile = s, s + S, ... as incremental (tx, ty, img) updates (no per-tile divisions)
    const int per_img = a.tiles_x * a.tiles_y;
    const int sx = a.S % a.tiles_x, sy = (a.S / a.tiles_x) % a.tiles_y, si = a.S / per_img;
    int ntx = s % a.tiles_x, nty = (s / a.tiles_x) % a.tiles_y, nimg = s / per_img, ntile = s;  // staging cursor
    auto advance = [&]() {
        ntile += a.S;
        ntx += sx;
        const int cx = ntx >= a.tiles_x;
        ntx -= cx ? a.tiles_x : 0;
        nty += sy + cx;
        const int cy = nty >= a.tiles_y;
        nty -= cy ? a.tiles_y : 0;
        nimg += si + cy;
    };

    // one DMA piece of the tile at (img, y0, x0); `live` = false turns it into a zero fill (no HBM traffic).
    // Every wave issues exactly NPW pieces per tile (ragged piece rows go to the sink) so that "tile landed"
    // is a counted vmcnt wait.
    auto piece = [&](int j, unsigned char* base, int img, int y0, int x0, bool live) {
        if (j < NPW_P) {
            const int pc = j * NW + wave;
            const bool real = (j + 1) * NW <= NPC_P || pc < NPC_P;  // wave-uniform
            const int pixb = pc * G::PPX, row0 = pixb / PHW, col0 = pixb - row0 * PHW;  // wave-uniform
            int px = col0 + prowP, py = row0;
            if (px >= PHW) {
                px -= PHW;
                py += 1;
            }
            const int gy = y0 - HALO + py, gx = x0 - HALO + px;
            const bool ok = live && real && okcP && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W &&
                            (NPP % G::PPX == 0 || pixb + prowP < NPP);
            const int ub = (img * a.PH + ps * (y0 - HALO) + pa) * a.PW + ps * (x0 - HALO) + pb;  // wave-uniform
            const int e = ub + ps * (py * a.PW + px);
            const char* src = ok ? lbaseP + (long)e * strideP : zero;
            dma_piece_ptr(src, (unsigned)(uintptr_t)(lds_void*)(real ? base + pc * 1024 : sink));
        } else {
            const int pc = (j - NPW_P) * NW + wave;
            const bool real = (j - NPW_P + 1) * NW <= NPC_Q || pc < NPC_Q;  // wave-uniform
            const int pixb = pc * G::QPX, row0 = pixb / TW, col0 = pixb % TW;  // wave-uniform
            const int gy = y0 + row0, gx = x0 + col0 + prowQ;
            const bool ok = live && real && okcQ && gy < a.H && gx < a.W;
            const int e = (img * a.H + gy) * a.W + gx;
            const char* src = ok ? lbaseQ + (long)e * strideQ : zero;
            dma_piece_ptr(src, (unsigned)(uintptr_t)(lds_void*)(real ? base + PP_BYTES + pc * 1024 : sink));
        }
    };

    // a wave owns a 32u x 32v accumulator for ALL taps (144 AGPRs for 3x3) over KH of the 16 pixel rows.  Two
    // waves per SIMD: one wave's address arithmetic / fragment shuffles run under the other wave's MFMAs.
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

    // transposed-read lane geometry (see wgrad_kernel)
    const int tq = (lane & 15) >> 2, tp = lane & 3;
    const int chA = wu * 32 + 16 * ((lane >> 4) & 1) + 4 * tp;
    const int chB = wv * 32 + 16 * ((lane >> 4) & 1) + 4 * tp;
    const int kx0 = 8 * h + tq;
    static_assert((KH * TW) % 4 == 0 && (KH * PHW) % 4 == 0, "k-split offset must keep the swizzle phase");
    auto swzP = [](int pix) { return PCH == 64 ? ((pix >> 1) & 1) << 5 : 0; };
    auto swzQ = [](int pix) { return QCH == 64 ? ((pix >> 1) & 1) << 5 : 0; };

    // prologue: the first DIST tiles go out at once
#pragma unroll
    for (int d = 0; d < DIST; ++d) {
        const bool live = ntile < a.ntiles;
#pragma unroll
        for (int j = 0; j < NPW; ++j) piece(j, smem + d * BUF, nimg, nty * G::TH, ntx * TW, live);
        advance();
    }
    int cur = 0;
    for (int tile = s; tile < a.ntiles; tile += a.S) {
        // tile's pieces are the oldest outstanding ones; DIST - 1 younger tiles stay in flight across the barrier
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DIST - 1) * NPW) : "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // staging target: the slot every wave has just left; its pieces are issued between the MFMAs below
        const bool more = ntile < a.ntiles;
        unsigned char* nbase = smem + ((cur + DIST) % NBUF) * BUF;
        const int ny0 = nty * G::TH, nx0 = ntx * TW;
        const bf16* sP = reinterpret_cast<const bf16*>(smem + cur * BUF) + (size_t)kh * KH * PHW * PCH;
        const bf16* sQ = reinterpret_cast<const bf16*>(smem + cur * BUF + PP_BYTES) + (size_t)kh * KH * TW * QCH;
        cur = (cur + 1) % NBUF;

        // Row-major walk over the HALO rows of this wave's pixel rows: halo row r feeds taps ky = r - y of the
        // output rows y = r, r-1, r-2, so its 3 transposed reads (12 pixels -> the 3 kx fragments by in-register
        // element shifts) are issued ONCE and reused by up to 9 MFMAs.  Raw reads for row r+1 are issued before
        // the MFMAs of row r and only shuffled into fragments after them.
        bf16x4 raw[3];
        bf16x8 bfr[4];
        auto readA = [&](int r) {
            const int a0 = r * PHW + kx0, a1 = a0 + 4, a2 = a0 + 8;
            raw[0] = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(sP + (size_t)a0 * PCH + (chA ^ swzP(a0))));
            raw[1] = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(sP + (size_t)a1 * PCH + (chA ^ swzP(a1))));
            if constexpr (NT == 9)
                raw[2] = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(sP + (size_t)a2 * PCH + (chA ^ swzP(a2))));
        };
        auto readB = [&](int slot, int y) {
            const int q0 = y * TW + kx0, q1 = q0 + 4;
            const bf16x4 blo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(sQ + (size_t)q0 * QCH + (chB ^ swzQ(q0))));
            const bf16x4 bhi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(sQ + (size_t)q1 * QCH + (chB ^ swzQ(q1))));
            bfr[slot] = __builtin_shufflevector(blo, bhi, 0, 1, 2, 3, 4, 5, 6, 7);
        };
        readA(0);
        readB(0, 0);
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            bf16x8 af[3];
            const bf16x8 lm = __builtin_shufflevector(raw[0], raw[1], 0, 1, 2, 3, 4, 5, 6, 7);
            af[0] = lm;
            if constexpr (NT == 9) {
                const bf16x8 mh = __builtin_shufflevector(raw[1], raw[2], 0, 1, 2, 3, 4, 5, 6, 7);
                af[1] = __builtin_shufflevector(lm, mh, 1, 2, 3, 4, 5, 6, 7, 12);
                af[2] = __builtin_shufflevector(lm, mh, 2, 3, 4, 5, 6, 7, 12, 13);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (r + 1 < NR) readA(r + 1);
            if (r + 1 < KH) readB((r + 1) & 3, r + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int jj = 0; jj < PPR; ++jj)
                if (r * PPR + jj < NPW) piece(r * PPR + jj, nbase, nimg, ny0, nx0, more);
            if (DBG && (a.debug & 4)) continue;  // ablation build only (splits the scheduling region)
#pragma unroll
            for (int ky = 0; ky <= 2 * HALO; ++ky) {
                const int y = r - ky;
                if (y < 0 || y >= KH) continue;
#pragma unroll
                for (int kx = 0; kx <= 2 * HALO; ++kx)
                    acc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[kx], bfr[y & 3], acc[ky * 3 + kx], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        static_assert(NR * PPR >= NPW, "all pieces of a tile are issued inside its row walk");
        advance();
    }
    // sum the KS pixel-row partials through LDS (binary tree): the upper half of the remaining waves park their
    // accumulators in the (now idle) ring, the lower half add them; the kh == 0 waves write the slab.
    // [slot][t][e4][lane] float4 -> conflict-free 16-byte accesses.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int step = G::KS / 2; step >= 1; step >>= 1) {
        __syncthreads();
        f32x4* red = reinterpret_cast<f32x4*>(smem) + (size_t)(((kh & (step - 1)) * (UB * VB) + uv)) * (NT * 4 * 64);
        if (kh >= step && kh < 2 * step) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int e4 = 0; e4 < 4; ++e4) {
                    f32x4 v4 = {acc[t][e4 * 4 + 0], acc[t][e4 * 4 + 1], acc[t][e4 * 4 + 2], acc[t][e4 * 4 + 3]};
                    red[(t * 4 + e4) * 64 + lane] = v4;
                }
        }
        __syncthreads();
        if (kh < step) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int e4 = 0; e4 < 4; ++e4) {
                    const f32x4 o = red[(t * 4 + e4) * 64 + lane];
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[t][e4 * 4 + i] += o[i];
                }
        }
    }
    if (kh == 0 && !(DBG && (a.debug & 8))) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
                const int u = u0 + wu * 32 + row, v = v0 + wv * 32 + (lane & 31);
                a.slabs[(((size_t)s * NT + t) * a.CUp + u) * a.CVp + v] = acc[t][e];
            }
    }
}

// ---------------------------------------------------------------------------------------------------
// 3x3 weight gradient, >= 64 channels on both sides (13 of the 16 launches of a U-Net step): v_mfma_f32_16x16x32_bf16
// on operands that BOTH come out of pixel-major LDS images through ds_read_b64_tr_b16 -- at any pixel offset, so the
// three kx fragments of a halo row are three reads instead of one read + VALU element shifts (162 VALU instructions
// per tile in wgrad_dma_kernel, whose MFMA loop ran at 0.57 of the held clock with no operand traffic at all).
//   G[tap][u][v] = sum_pixels P[pixel + tap][u] * Q[pixel][v]:  A operand = P (rows = 16 u channels), B operand = Q
//   (columns = 16 v channels), K = 32 pixels = the two 16-pixel tile rows y, y + 1 (k octet g: pixels 4g .. 4g + 3 of
//   row y, then of row y + 1 -- the order is free as long as both operands use it).
// Workgroup = 64 u x 64 v x 9 taps, 8 waves = (2 x 2 blocks of 32 channels) x 2 pixel-row halves ("k-split", summed
// through LDS once at the end); a wave owns 9 x 2 x 2 accumulator blocks (144 registers) and walks 4 row pairs per
// 16 x 16 pixel tile: 18 P fragments + 2 Q fragments -> 36 MFMAs per pair.
// LDS images keep the global [pixel][64 channels] order (every 1-KiB LDS-DMA piece = 8 pixels x 128 B contiguous);
// the 32-byte channel slots of a pixel row are XOR-swizzled with (pixel >> 1) & 3, so that the 8 consecutive pixels a
// 32-lane half of a transposed read touches land on 8 different 32-byte bank groups (conflict-free at every offset).
// The swizzle is applied on the DMA's SOURCE side (LDS destinations stay linear); fragment addresses are 8 per-lane
// bases (one per pixel phase) plus compile-time offsets: no address arithmetic in the loop.
#ifdef WG_STAMP
// diagnostic build only: per-workgroup cycle sums of the tile loop's phases (wave 0), into a buffer nothing else reads
__device__ unsigned long long g_wg_stamp[4096 * 8];
extern "C" int hipseg_debug_wg_stamps(void* host, int nwg) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_wg_stamp), (size_t)nwg * 8 * sizeof(unsigned long long)) == hipSuccess ? 0 : 1;
}
#endif

struct Tr16Geo {
    static constexpr int TH = 16, NW = 8, KH = 8, NT = 9;
    static constexpr int PHW = TW + 2, NPP = (TH + 2) * PHW, NPQ = TH * TW;
    static constexpr int NPC_P = (NPP + 7) / 8, NPC_Q = NPQ / 8;  // 1-KiB pieces (8 pixels x 128 B)
    static constexpr int NPW_P = (NPC_P + NW - 1) / NW, NPW_Q = NPC_Q / NW, NPW = NPW_P + NPW_Q;
    static constexpr int PP_BYTES = NPC_P * 1024, Q_BYTES = NPC_Q * 1024, BUF = PP_BYTES + Q_BYTES;
    static constexpr size_t RED_BYTES = (size_t)4 * NT * 16 * 64 * 4;  // 4 parked accumulator sets
    static constexpr size_t RING_BYTES = (size_t)2 * BUF + 1024;       // + 1-KiB sink for pad pieces
    static constexpr size_t LDS = RING_BYTES > RED_BYTES ? RING_BYTES : RED_BYTES;
    static constexpr int PF = 3;  // P fragments in flight (each feeds 4 MFMAs = 64 cycles)
    static_assert(LDS <= 160 * 1024 && NPC_Q % NW == 0, "geometry");
};

// PAIR: ONE launch computes the weight gradients of TWO layers over the same pixel grid (the two convolutions of a
// ConvBlock): problem a has tA = UT * VT channel tiles, problem b tB; each of the S pixel splits owns tA + tB
// workgroups, so S = CUs / (tA + tB) instead of CUs / tA and CUs / tB in two launches -- the slab volume
// (workgroups x 147 KB of fp32 partial sums, written here and read back by the reduction) of BOTH layers together is
// what ONE layer's was.  The grid is padded to a multiple of 8 and dealt to the XCDs in contiguous runs of logical
// ids (s-major), so the workgroups of one pixel split share an L2.
// RAGGED (round 4): H or W not a multiple of 16 (ClipUnet-224's 56 x 56 and 28 x 28 levels).  The tile grid is rounded up
// and a lane whose pixel lies beyond the image's bottom / right edge does not fetch (zero fill: it contributes nothing):
// its halo coordinates are two more per-lane constants per piece, compared with two per-tile limits.  A separate
// instantiation, so the whole-tile kernels of the U-Nets keep their 3-instruction pieces.
// ONLOAD (round 4): see WgArgs::p_scale -- the wave that DMA'd a P piece rewrites it in LDS as relu(x * scale + shift)
// after its wait and before the tile's barrier; lanes whose halo pixel lies outside the image keep the DMA's zero.  The
// weight gradient of a convolution whose input exists only pre-BatchNorm (the second convolution of a full-resolution
// ConvBlock when the forward applied BatchNorm + ReLU on load, hipseg_conv3_bnrelu_in).  Whole tiles, single launch.
template <bool PAIR, bool RAGGED = false, bool ONLOAD = false>
__global__ __launch_bounds__(512, 1) void wgrad3_tr16_kernel(WgArgs a, WgArgs b2, int tA, int tB) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef Tr16Geo G;
    constexpr int NT = 9, PHW = G::PHW, NPP = G::NPP, NW = G::NW, NPC_P = G::NPC_P, NPW_P = G::NPW_P, NPW_Q = G::NPW_Q;
    constexpr int PP_BYTES = G::PP_BYTES, BUF = G::BUF, PF = G::PF;
    typedef __attribute__((address_space(3))) void lds_void;
    typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
    constexpr unsigned OOB = 0x80000000u;

    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kh = wave >> 2, wu = wave & 3;  // pixel-row half, 16-channel u block of the wave (all 64 v channels)
    int vt, ut, s;
    if (PAIR) {
        const int cpx = gridDim.x >> 3;
        const int logical = (blockIdx.x & 7) * cpx + (blockIdx.x >> 3);
        if (logical >= a.S * (tA + tB)) return;  // grid padding (whole workgroup, before any barrier)
        s = logical / (tA + tB);
        int rem = logical - s * (tA + tB);
        if (rem >= tA) {  // second problem (workgroup-uniform: the arguments stay scalar)
            rem -= tA;
            a = b2;
        }
        vt = rem % a.VT;
        ut = rem / a.VT;
    } else {
        const int bid = xcd_block(blockIdx.x, a.xcd);
        vt = bid % a.VT;
        ut = (bid / a.VT) % a.UT;
        s = bid / (a.VT * a.UT);
    }
    const int u0 = ut * 64, v0 = vt * 64;
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_void*)smem;  // 1-KiB aligned (declaration): bits 5-6 are slot bits
    const unsigned sink = lds0 + 2 * BUF;

    // ---- staging (LDS-DMA through buffer addressing; an out-of-range per-lane offset zero-fills).  The main loop is
    // bound by VECTOR INSTRUCTION ISSUE (a 16x16x32 MFMA leaves 8 of its 16 cycles to other vector instructions; two
    // waves x (104 transposed reads + the DMA address arithmetic) per tile filled them): a piece therefore costs 3
    // VALU instructions -- per-lane byte offsets relative to the tile are constants (whole tiles only: launch
    // condition H % 16 == 0, W % 16 == 0), the tile's origin is a scalar offset, and a lane whose halo pixel falls
    // outside the image is found by AND-ing its per-piece edge flags with the tile's edge mask.
    const bool second = u0 >= a.CU0;  // the 64-channel u tile lies inside ONE source tensor (launch condition)
    const int cstrideP = (second ? a.CU1 : a.CU0) * 2, cstrideQ = a.CV * 2;
    // (descriptor base moved back by one row + one pixel: per-lane offsets are then relative to the halo origin
    //  (y0 - 1, x0 - 1) and non-negative, the tile offset is that of (y0, x0) and non-negative)
    const unsigned biasP = (unsigned)((a.W + 1) * cstrideP);
    const v4i_t r_p = make_rsrc(reinterpret_cast<const char*>(second ? a.p1 : a.p0) - biasP,
                                (unsigned)((size_t)a.B * a.H * a.W * cstrideP) + biasP);
    const v4i_t r_q = make_rsrc(a.q, (unsigned)((size_t)a.B * a.H * a.W * cstrideQ));
    const unsigned soP = (unsigned)(second ? u0 - a.CU0 : u0) * 2u, soQ = (unsigned)v0 * 2u;
    // lane -> (pixel row of the piece, 16-byte slot); source slot = destination slot with its 32-byte index XOR-ed
    const int prow = lane >> 3, slot = lane & 7;
    const unsigned colsrc = (unsigned)((((slot >> 1) ^ ((prow >> 1) & 3)) << 1) | (slot & 1)) * 16u;
    float psc[ONLOAD ? 8 : 1], psh[ONLOAD ? 8 : 1];  // ONLOAD: scale / shift of the 8 channels this lane copies
    if constexpr (ONLOAD) {
        const int c0 = u0 + (int)(colsrc >> 1);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            psc[e] = a.p_scale[c0 + e];
            psh[e] = a.p_shift[c0 + e];
        }
    }
    // 5 flags per P piece: halo pixel in the top / bottom halo row, left / right halo column, lane beyond the halo tile
    unsigned pofs[NPW_P], qofs[NPW_Q], pflags = 0;
    unsigned pyx[RAGGED ? NPW_P : 1];  // RAGGED: halo row | halo column << 8 of the lane's pixel of piece j
#pragma unroll
    for (int j = 0; j < NPW_P; ++j) {
        const int hp = (j * NW + wave) * 8 + prow, dy = hp / PHW, dx = hp - dy * PHW;
        if (RAGGED) pyx[j] = (unsigned)(dy | (dx << 8));
        pofs[j] = (unsigned)((dy * a.W + dx) * cstrideP) + colsrc;
        const unsigned f = hp < NPP ? (unsigned)((dy == 0) | ((dy == G::TH + 1) << 1) | ((dx == 0) << 2) | ((dx == PHW - 1) << 3)) : 16u;
        pflags |= f << (5 * j);
    }
#pragma unroll
    for (int j = 0; j < NPW_Q; ++j) {
        const int pix = (j * NW + wave) * 8 + prow;
        qofs[j] = (unsigned)(((pix >> 4) * a.W + (pix & 15)) * cstrideQ) + colsrc;
    }
    const int per_img = a.tiles_x * a.tiles_y;
    const int sx = a.S % a.tiles_x, sy = (a.S / a.tiles_x) % a.tiles_y, si = a.S / per_img;
    int ntx = s % a.tiles_x, nty = (s / a.tiles_x) % a.tiles_y, nimg = s / per_img, ntile = s;  // staging cursor
    auto advance = [&]() {
        ntile += a.S;
        ntx += sx;
        const int cx = ntx >= a.tiles_x;
        ntx -= cx ? a.tiles_x : 0;
        nty += sy + cx;
        const int cy = nty >= a.tiles_y;
        nty -= cy ? a.tiles_y : 0;
        nimg += si + cy;
    };
    // per-tile scalars of the tile under the staging cursor: its byte offsets and its edge mask (bit k set = the halo
    // side k lies outside the image; bit 4 always set: lanes beyond the halo tile never fetch)
    struct TileS {
        unsigned soP, soQ, edge;
        int ylim, xlim;  // RAGGED: first halo row / column beyond the image (H - y0 + 1, W - x0 + 1)
        bool qcolok;     // RAGGED: this lane's Q tile column is inside the image
    };
    const int qcol = 8 * (wave & 1) + prow, qrow0 = wave >> 1;  // Q piece j holds tile row 4 j + qrow0, column qcol
    auto tile_scalars = [&](int img, int y0, int x0) {
        TileS t;
        const unsigned pix = (unsigned)((img * a.H + y0) * a.W + x0);
        t.soP = pix * (unsigned)cstrideP + soP;
        t.soQ = pix * (unsigned)cstrideQ + soQ;
        t.edge = (unsigned)((y0 == 0) | ((y0 + G::TH >= a.H) << 1) | ((x0 == 0) << 2) | ((x0 + TW >= a.W) << 3)) | 16u;
        t.ylim = a.H - y0 + 1;
        t.xlim = a.W - x0 + 1;
        t.qcolok = qcol < a.W - x0;
        return t;
    };
    // piece j (0 .. NPW - 1) into ring slot `base` (absolute LDS byte address)
    auto piece = [&](int j, unsigned base, const TileS& t) {
        if (j < NPW_P) {
            const int pc = j * NW + wave;
            const bool real = (j + 1) * NW <= NPC_P || pc < NPC_P;  // wave-uniform
            const unsigned hit = pflags & (t.edge << (5 * j));
            bool bad = hit != 0;
            if (RAGGED) bad = bad || (int)(pyx[j] & 0xffu) >= t.ylim || (int)(pyx[j] >> 8) >= t.xlim;
            const unsigned vo = bad ? OOB : pofs[j];
            if (real) dma_piece(r_p, base + pc * 1024, vo, t.soP);
        } else {
            const int pc = (j - NPW_P) * NW + wave;
            unsigned vo = qofs[j - NPW_P];
            if (RAGGED) vo = (t.qcolok && 4 * (j - NPW_P) + qrow0 < t.ylim - 1) ? vo : OOB;
            dma_piece(r_q, base + PP_BYTES + pc * 1024, vo, t.soQ);
        }
    };

    // ---- fragment addresses.  Transposed read: 16-lane group g = lane >> 4 is k octet g; its lane 4q + p addresses
    // pixel (4g + q) of the fragment's tile row, channels 4p .. 4p + 3 of the 16-channel block.
    // Addresses are kept as absolute 32-bit LDS addresses (the LDS base folded in ONCE: "smem + offset" inside the loop
    // is an add of a link-time symbol that the compiler cannot fold and would keep a second copy of every base alive).
    const int g4q = 4 * (lane >> 4) + ((lane & 15) >> 2), p4 = lane & 3;
    // P: halo pixel hp = h0 + C, C = (2 rp + ky + j) * PHW + kx (compile time); swizzle phase depends on C & 7 only
    const int h0 = kh * G::KH * PHW + g4q;
    unsigned pb[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) pb[m] = lds0 + (unsigned)(h0 * 128 + ((wu ^ (((h0 + m) >> 1) & 3)) * 32) + p4 * 8);
    // Q: pixel q = q0 + (2 rp + j) * 16: phase always that of q0; the wave reads all four 16-channel v blocks
    const int q0 = kh * G::KH * TW + g4q;
    unsigned qb[4];
#pragma unroll
    for (int vb = 0; vb < 4; ++vb) qb[vb] = lds0 + (unsigned)(PP_BYTES + q0 * 128 + ((vb ^ ((q0 >> 1) & 3)) * 32) + p4 * 8);

    // wave tile: 16 u x 64 v x 9 taps = 36 accumulator blocks.  A P fragment feeds 4 MFMAs (64 cycles), so PF = 3
    // fragments in flight cover the LDS latency; per row pair 18 + 8 transposed reads for 36 MFMAs.
    f32x4 acc[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[t][i] = f32x4{0.f, 0.f, 0.f, 0.f};

    // prologue: the first tile
    unsigned cur_edge = 16u;  // ONLOAD: edge mask of the tile being computed (staged one iteration earlier)
    if (ntile < a.ntiles) {
        const TileS t0 = tile_scalars(nimg, nty * G::TH, ntx * TW);
        cur_edge = t0.edge;
#pragma unroll
        for (int j = 0; j < G::NPW; ++j) piece(j, lds0, t0);
    }
    advance();
    int cur = 0;
#ifdef WG_STAMP
    unsigned long long st_dma = 0, st_bar = 0, st_cmp = 0, st_n = 0;
    const unsigned long long st_t0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    for (int tile = s; tile < a.ntiles; tile += a.S) {
#ifdef WG_STAMP
        const unsigned long long ta = __builtin_amdgcn_s_memtime();
#endif
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of the tile have landed
#ifdef WG_STAMP
        const unsigned long long tb = __builtin_amdgcn_s_memtime();
#endif
        if constexpr (ONLOAD) {
#pragma unroll
            for (int j = 0; j < NPW_P; ++j) {
                const int pc = j * NW + wave;
                if ((j + 1) * NW <= NPC_P || pc < NPC_P) {  // wave-uniform
                    bf16x8* q = reinterpret_cast<bf16x8*>(smem + cur * BUF + pc * 1024) + lane;
                    bf16x8 v = *q;
                    const bool in = !(pflags & (cur_edge << (5 * j)));
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (bf16)(in ? fmaxf((float)v[e] * psc[e] + psh[e], 0.f) : 0.f);
                    *q = v;
                }
            }
        }
        __builtin_amdgcn_s_barrier();                      // everyone's have; everyone left the other slot
        __builtin_amdgcn_sched_barrier(0);
#ifdef WG_STAMP
        const unsigned long long tc = __builtin_amdgcn_s_memtime();
#endif
        const bool more = ntile < a.ntiles;  // (the wait at the top is vmcnt(0): no piece has to be issued for the count)
        const unsigned nbase = lds0 + (cur ^ 1) * BUF;
        const TileS tn = tile_scalars(nimg, nty * G::TH, ntx * TW);
        // (pb / qb carry the ring slot's offset: updated in place per tile)
        auto readP = [&](int f) {  // fragment f = (row pair, tap)
            const int rp = f / NT, tap = f % NT;
            const int C0 = (2 * rp + tap / 3) * PHW + tap % 3, C1 = C0 + PHW;
            const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(uintptr_t)pb[C0 & 7] + C0 * 16);
            const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(uintptr_t)pb[C1 & 7] + C1 * 16);
            return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        };
        auto readQ = [&](int rp, int vb) {
            const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(uintptr_t)qb[vb] + (2 * rp) * TW * 16);
            const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(uintptr_t)qb[vb] + (2 * rp + 1) * TW * 16);
            return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        };
        bf16x8 af[PF], bq[4];
#pragma unroll
        for (int vb = 0; vb < 4; ++vb) bq[vb] = readQ(0, vb);
#pragma unroll
        for (int i = 0; i < PF; ++i) af[i] = readP(i);
#pragma unroll
        for (int rp = 0; rp < 4; ++rp) {
#pragma unroll
            for (int tap = 0; tap < NT - 2; ++tap) {
                const int f = rp * NT + tap;
                if (more && f % 3 == 0 && f / 3 < G::NPW) piece(f / 3, nbase, tn);
#ifdef HIPSEG_ABLATE
                if (!(a.debug & 4))
#endif
                {
#pragma unroll
                    for (int vb = 0; vb < 4; ++vb)
                        acc[tap][vb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[f % PF], bq[vb], acc[tap][vb], 0, 0, 0);
                }
                if (f + PF < 4 * NT) af[f % PF] = readP(f + PF);
                __builtin_amdgcn_sched_barrier(0);
            }
            // The Q fragments are single-buffered: the LAST TWO taps of a row pair run v-block-major, and each v block's
            // fragment of the NEXT row pair is read as soon as its last MFMA has issued (>= 6 MFMAs = 96 cycles before
            // its first use).
            {
                const int f = rp * NT + NT - 2;
                if (more && f % 3 == 0 && f / 3 < G::NPW) piece(f / 3, nbase, tn);
                if (more && (f + 1) % 3 == 0 && (f + 1) / 3 < G::NPW) piece((f + 1) / 3, nbase, tn);
#pragma unroll
                for (int vb = 0; vb < 4; ++vb) {
#ifdef HIPSEG_ABLATE
                    if (!(a.debug & 4))
#endif
                    {
                        acc[NT - 2][vb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[f % PF], bq[vb], acc[NT - 2][vb], 0, 0, 0);
                        acc[NT - 1][vb] =
                            __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[(f + 1) % PF], bq[vb], acc[NT - 1][vb], 0, 0, 0);
                    }
                    if (rp + 1 < 4) bq[vb] = readQ(rp + 1, vb);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (f + PF < 4 * NT) af[f % PF] = readP(f + PF);
                if (f + 1 + PF < 4 * NT) af[(f + 1) % PF] = readP(f + 1 + PF);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        static_assert(G::NPW <= 12, "all pieces of a tile are issued inside its fragment walk");
#ifdef WG_STAMP
        {
            const unsigned long long td = __builtin_amdgcn_s_memtime();
            st_dma += tb - ta;
            st_bar += tc - tb;
            st_cmp += td - tc;
            st_n += 1;
        }
#endif
        advance();
        cur_edge = tn.edge;
        {  // fragment bases follow the ring slot (in place: no second set of address registers)
            const unsigned d = cur ? (unsigned)-BUF : (unsigned)BUF;
#pragma unroll
            for (int m = 0; m < 8; ++m) pb[m] += d;
#pragma unroll
            for (int vb = 0; vb < 4; ++vb) qb[vb] += d;
        }
        cur ^= 1;
    }
    // ---- k-split halves meet in LDS (the ring is idle now), the kh == 0 waves write the slab [S][9][CUp][CVp]:
    // accumulator block (tap, vb): lane holds column v = lane & 15, rows u = 4 (lane >> 4) + e
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    f32x4* red = reinterpret_cast<f32x4*>(smem) + (size_t)(wave & 3) * (NT * 4 * 64);
    if (kh == 1) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) red[(t * 4 + i) * 64 + lane] = acc[t][i];
    }
    __syncthreads();
    if (kh == 0) {
        const int vcol = lane & 15, ur = 4 * (lane >> 4);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int vb = 0; vb < 4; ++vb) {
                const f32x4 o = red[(t * 4 + vb) * 64 + lane];
                const int u = u0 + wu * 16 + ur, v = v0 + vb * 16 + vcol;
                float* dst = a.slabs + (((size_t)s * NT + t) * a.CUp + u) * a.CVp + v;
#pragma unroll
                for (int e = 0; e < 4; ++e) dst[(size_t)e * a.CVp] = acc[t][vb][e] + o[e];
            }
    }
#ifdef WG_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tid == 0 && blockIdx.x < 4096) {
        unsigned long long* o = g_wg_stamp + blockIdx.x * 8;
        o[0] = st_dma; o[1] = st_bar; o[2] = st_cmp; o[3] = st_n;
        o[4] = __builtin_amdgcn_s_memtime() - st_t0;
        o[5] = __builtin_amdgcn_s_memrealtime() - st_r0;
        o[6] = st_r0;
    }
#endif
#else
    (void)a;
    (void)b2;
    (void)tA;
    (void)tB;
#endif
}

int launch_tr16(const WgArgs& a, hipStream_t s) {
    const dim3 grid((unsigned)(a.S * a.UT * a.VT));
    if (a.p_scale) {  // (whole tiles, single source: checked by hipseg_conv_wgrad_bnrelu_p_applies)
        if (int rc = hs_set_max_lds(reinterpret_cast<const void*>(&wgrad3_tr16_kernel<false, false, true>), (size_t)Tr16Geo::LDS))
            return rc;
        hipLaunchKernelGGL((wgrad3_tr16_kernel<false, false, true>), grid, dim3(512), Tr16Geo::LDS, s, a, a, 0, 0);
    } else if (a.H % 16 || a.W % 16) {
        if (int rc = hs_set_max_lds(reinterpret_cast<const void*>(&wgrad3_tr16_kernel<false, true>), (size_t)Tr16Geo::LDS)) return rc;
        hipLaunchKernelGGL((wgrad3_tr16_kernel<false, true>), grid, dim3(512), Tr16Geo::LDS, s, a, a, 0, 0);
    } else {
        if (int rc = hs_set_max_lds(reinterpret_cast<const void*>(&wgrad3_tr16_kernel<false>), (size_t)Tr16Geo::LDS)) return rc;
        hipLaunchKernelGGL(wgrad3_tr16_kernel<false>, grid, dim3(512), Tr16Geo::LDS, s, a, a, 0, 0);
    }
    HS_LAUNCH_CHECK("conv_wgrad_tr16");
    return HIPSEG_OK;
}

int launch_tr16_pair(const WgArgs& a, const WgArgs& b, hipStream_t s) {
    const int tA = a.UT * a.VT, tB = b.UT * b.VT;
    const int grid = (a.S * (tA + tB) + 7) / 8 * 8;
    if (a.H % 16 || a.W % 16) {
        if (int rc = hs_set_max_lds(reinterpret_cast<const void*>(&wgrad3_tr16_kernel<true, true>), (size_t)Tr16Geo::LDS)) return rc;
        hipLaunchKernelGGL((wgrad3_tr16_kernel<true, true>), dim3((unsigned)grid), dim3(512), Tr16Geo::LDS, s, a, b, tA, tB);
    } else {
        if (int rc = hs_set_max_lds(reinterpret_cast<const void*>(&wgrad3_tr16_kernel<true>), (size_t)Tr16Geo::LDS)) return rc;
        hipLaunchKernelGGL(wgrad3_tr16_kernel<true>, dim3((unsigned)grid), dim3(512), Tr16Geo::LDS, s, a, b, tA, tB);
    }
    HS_LAUNCH_CHECK("conv_wgrad_tr16(pair)");
    return HIPSEG_OK;
}

template <int NT, int UB, int VB>
int launch_dma(const WgArgs& a, hipStream_t s) {
    constexpr size_t lds = DmaGeo<NT, UB, VB>::LDS;
    if (int rc = hs_set_max_lds(reinterpret_cast<const void*>(&wgrad_dma_kernel<NT, UB, VB, false>), (size_t)lds)) return rc;
    if (int rc = hs_set_max_lds(reinterpret_cast<const void*>(&wgrad_dma_kernel<NT, UB, VB, true>), (size_t)lds)) return rc;
    const dim3 grid((unsigned)(a.S * a.UT * a.VT));
    if (a.debug)
        hipLaunchKernelGGL((wgrad_dma_kernel<NT, UB, VB, true>), grid, dim3(512), lds, s, a);
    else
        hipLaunchKernelGGL((wgrad_dma_kernel<NT, UB, VB, false>), grid, dim3(512), lds, s, a);
    HS_LAUNCH_CHECK("conv_wgrad_dma");
    return HIPSEG_OK;
}

template <int NT>
int launch_dma_blocks(const WgArgs& a, int UB, int VB, hipStream_t s) {
    if (UB == 2) return VB == 2 ? launch_dma<NT, 2, 2>(a, s) : launch_dma<NT, 2, 1>(a, s);
    return VB == 2 ? launch_dma<NT, 1, 2>(a, s) : launch_dma<NT, 1, 1>(a, s);
}

// sum the S slabs in fixed order and scatter into the parameter's native layout
__global__ void wgrad_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ dw, int S, int sstep, int NT,
                                    int CU, int CV, int CUp, int CVp, int mode, int ab) {
    const long total = (long)NT * CU * CV;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int v = (int)(i % CV);
        const int u = (int)((i / CV) % CU);
        const int t = (int)(i / ((long)CV * CU));
        const size_t off = ((size_t)t * CUp + u) * CVp + v;
        const size_t stride = (size_t)NT * CUp * CVp * sstep;
        float sum = 0.f;
        for (int s = 0; s < S; ++s) sum += slabs[off + s * stride];
        size_t o;
        if (mode == HIPSEG_CONV3)
            o = ((size_t)v * CU + u) * 9 + t;  // (Cout=v, Cin=u, ky, kx)
        else if (mode == HIPSEG_CONV1)
            o = (size_t)v * CU + u;  // (Cout=v, Cin=u)
        else {  // ConvT: u = ab*Cout + co, CU = 4*Cout -> (Cin=v, Cout=co, a, b)
            const int cout = CU / 4, tab = u / cout, co = u - tab * cout;
            o = ((size_t)v * cout + co) * 4 + tab;
        }
        dw[o] = sum;
    }
}

// Slab reduction of the 3x3 weight gradient in ONE launch for any number of slabs: a block owns a UTL(u) x 32(v) x TG(taps)
// tile and splits the S slabs over NS slices of UTL * 32 lanes (up to 1024 threads: enough loads in flight -- the
// reduction is a chain of L2/HBM round trips otherwise); slice partials meet in LDS and are summed in slice order, then
// written in the native (Cout=v, Cin=u, ky, kx) layout.  UTL = 4, TG = 9 (144-byte output runs) for the layers whose
// (CV/32) x (CU/4) grid fills the chip; the <= 64-channel layers, whose slab volume is the same 37.7 MB but whose 4 x 32
// grid is only 8..64 blocks (one CU's 64 B/clk load path each: 15-16 us against 8 us for the wide layers), take UTL = 1
// and TG = 3: (CV/32) x CU x 3 blocks.  Replaced the in-place pre-reduce + reduce3 pair (two launches).
template <int NS, int UTL, int TG>
__global__ __launch_bounds__(32 * UTL * NS) void wgrad_reduce3_wide_kernel(const float* __restrict__ slabs,
                                                                           float* __restrict__ dw, int S, int CU, int CV,
                                                                           int CUp, int CVp,
                                                                           const float* __restrict__ slabs2 = nullptr,
                                                                           float* __restrict__ dw2 = nullptr, int CU2 = 0,
                                                                           int CUp2 = 0, int yb0 = 0) {
    // a wave reads full 128-byte slab rows (32 v) per load instruction
    constexpr int UV = 32 * UTL;
    __shared__ float red[NS][UV][TG];
    const int tid = threadIdx.x, sl = tid / UV, uv = tid % UV, u = uv >> 5, v = uv & 31;
    int by = blockIdx.y;
    if (slabs2 && by >= yb0) {  // second weight tensor of a paired launch (same S, CV): grid rows [yb0, ...)
        by -= yb0;
        slabs = slabs2;
        dw = dw2;
        CU = CU2;
        CUp = CUp2;
    }
    const int u0 = by * UTL, v0 = blockIdx.x * 32, t0 = blockIdx.z * TG;
    const size_t tstride = (size_t)CUp * CVp, sstride = 9 * tstride;
    const float* src = slabs + (size_t)(u0 + u) * CVp + v0 + v + t0 * tstride;  // padded to 32: always in bounds
    float acc[TG];
#pragma unroll
    for (int t = 0; t < TG; ++t) acc[t] = 0.f;
    for (int s = sl; s < S; s += NS) {
#pragma unroll
        for (int t = 0; t < TG; ++t) acc[t] += src[s * sstride + t * tstride];
    }
#pragma unroll
    for (int t = 0; t < TG; ++t) red[sl][uv][t] = acc[t];
    __syncthreads();
    for (int o = tid; o < UV * TG; o += UV * NS) {  // index within the tile: vv * (UTL * TG) + uu * TG + t
        const int vv = o / (UTL * TG), rem = o % (UTL * TG), uu = rem / TG, t = rem % TG;
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < NS; ++k) sum += red[k][uu * 32 + vv][t];
        if (v0 + vv < CV && u0 + uu < CU) dw[((size_t)(v0 + vv) * CU + u0 + uu) * 9 + t0 + t] = sum;
    }
}

struct Plan {
    int UT, VT, S, ntiles, tiles_x, tiles_y, CUp, CVp, NT;
};

template <typename T>
Plan make_plan(int mode, int CU, int CV, int B, int H, int W) {
    Plan p;
    p.NT = mode == HIPSEG_CONV3 ? 9 : 1;
    p.UT = cdiv(CU, WT<T>::UC);
    p.VT = cdiv(CV, WT<T>::VC);
    p.CUp = p.UT * WT<T>::UC;
    p.CVp = p.VT * WT<T>::VC;
    p.tiles_x = cdiv(W, TW);
    p.tiles_y = cdiv(H, WT<T>::TH);
    p.ntiles = B * p.tiles_x * p.tiles_y;
    // workgroups per launch ~ CUs (bf16: the double-buffered DMA kernel runs one 146-KB workgroup per CU;
    // f32: two single-buffered ones).  Fewer, fatter splits also keep the slab traffic small.
    const int ncu = device_cus();
    int S = (sizeof(T) == 2 ? ncu : 2 * ncu) / (p.UT * p.VT);
    if (S < 1) S = 1;
    if (S > p.ntiles) S = p.ntiles;
    p.S = S;
    return p;
}

Plan plan_for(int dtype, int mode, int CU, int CV, int B, int H, int W) {
    return dtype == HIPSEG_BF16 ? make_plan<bf16>(mode, CU, CV, B, H, W) : make_plan<float>(mode, CU, CV, B, H, W);
}

template <typename T, int NT>
int launch(const WgArgs& a, hipStream_t s) {
    constexpr int HALO = NT == 9 ? 1 : 0;
    constexpr int NPP = (WT<T>::TH + 2 * HALO) * (TW + 2 * HALO), NPQ = WT<T>::TH * TW;
    const size_t lds = ((size_t)NPP * WT<T>::UC + (size_t)NPQ * WT<T>::VC) * sizeof(T);
    if (int rc = hs_set_max_lds(reinterpret_cast<const void*>(&wgrad_kernel<T, NT>), (size_t)lds)) return rc;
    hipLaunchKernelGGL((wgrad_kernel<T, NT>), dim3((unsigned)(a.S * a.UT * a.VT)), dim3(256), lds, s, a);
    HS_LAUNCH_CHECK("conv_wgrad");
    return HIPSEG_OK;
}

}  // namespace

extern "C" size_t hipseg_wgrad_workspace_elems(int mode, int CU, int CV, int B, int H, int W) {
    if (mode == HIPSEG_CONVT) CU *= 4;  // one GEMM over the (a, b, co) = 4*Cout gathered channels
    size_t m = 0;
    for (int dt = 0; dt < 2; ++dt) {
        const Plan p = plan_for(dt, mode, CU, CV, B, H, W);
        const size_t e = (size_t)p.S * p.NT * p.CUp * p.CVp;
        if (e > m) m = e;
    }
    return m;
}

static int conv_wgrad_impl(int dtype, int mode, const void* p0, int CU0, const void* p1, int CU1, const void* q, int CV,
                           float* dw, float* slabs, int B, int H, int W, hipseg_stream_t stream, const float* p_scale,
                           const float* p_shift);

extern "C" int hipseg_conv_wgrad(int dtype, int mode, const void* p0, int CU0, const void* p1, int CU1,
                                 const void* q, int CV, float* dw, float* slabs, int B, int H, int W,
                                 hipseg_stream_t stream) {
    return conv_wgrad_impl(dtype, mode, p0, CU0, p1, CU1, q, CV, dw, slabs, B, H, W, stream, nullptr, nullptr);
}

// 3x3 weight gradient against relu(P * scale[u] + shift[u]) with P the PRE-normalisation tensor (BatchNorm + ReLU of the
// previous layer applied in the load path, the counterpart of hipseg_conv3_bnrelu_in in backward): the second
// convolution of a ConvBlock whose activated intermediate was never written.  bf16, >= 64 channels in multiples of 64 on
// both sides, whole 16 x 16 tiles (the tr16 kernel's single-launch form).
extern "C" int hipseg_conv_wgrad_bnrelu_p_applies(int dtype, int CU, int CV, int B, int H, int W) {
    static const bool off = getenv("HIPSEG_NO_BN_ON_LOAD") != nullptr || getenv("HIPSEG_NO_WGRAD_TR16") != nullptr ||
                            getenv("HIPSEG_NO_DMA") != nullptr;
    if (off || dtype != HIPSEG_BF16 || CU % 64 || CV % 64 || H % 16 || W % 16) return 0;
    return (size_t)B * H * W * (size_t)(CU > CV ? CU : CV) * 2 <= ((size_t)1 << 30) ? 1 : 0;
}

extern "C" int hipseg_conv_wgrad_bnrelu_p(int dtype, const void* p, int CU, const float* scale, const float* shift,
                                          const void* q, int CV, float* dw, float* slabs, int B, int H, int W,
                                          hipseg_stream_t stream) {
    HS_REQUIRE(scale && shift, "conv_wgrad_bnrelu_p: scale and shift are required");
    HS_REQUIRE(hipseg_conv_wgrad_bnrelu_p_applies(dtype, CU, CV, B, H, W),
               "conv_wgrad_bnrelu_p: unsupported shape (ask hipseg_conv_wgrad_bnrelu_p_applies first)");
    return conv_wgrad_impl(dtype, HIPSEG_CONV3, p, CU, nullptr, 0, q, CV, dw, slabs, B, H, W, stream, scale, shift);
}

static int conv_wgrad_impl(int dtype, int mode, const void* p0, int CU0, const void* p1, int CU1, const void* q, int CV,
                           float* dw, float* slabs, int B, int H, int W, hipseg_stream_t stream, const float* p_scale,
                           const float* p_shift) {
    HS_REQUIRE(dtype == HIPSEG_F32 || dtype == HIPSEG_BF16, "conv_wgrad: bad dtype %d", dtype);
    HS_REQUIRE(mode == HIPSEG_CONV3 || mode == HIPSEG_CONV1 || mode == HIPSEG_CONVT, "conv_wgrad: bad mode %d", mode);
    HS_REQUIRE(p0 && q && dw && slabs && CU0 > 0 && CV > 0, "conv_wgrad: null operand or empty channel range");
    HS_REQUIRE((CU1 == 0) == (p1 == nullptr), "conv_wgrad: p1/CU1 mismatch");
    HS_REQUIRE(B > 0 && H > 0 && W > 0, "conv_wgrad: empty pixel grid");
    const int CU = mode == HIPSEG_CONVT ? 4 * CU0 : CU0 + CU1;
    HS_REQUIRE(!(mode == HIPSEG_CONVT && CU1 != 0), "conv_wgrad: CONVT takes a single P tensor");
    Plan pl = plan_for(dtype, mode, CU, CV, B, H, W);
    const int vec = dtype == HIPSEG_BF16 ? 8 : 4;
    static const bool no_dma = getenv("HIPSEG_NO_DMA") != nullptr;
    const bool dma = dtype == HIPSEG_BF16 && CU0 % vec == 0 && CU1 % vec == 0 && CV % vec == 0 && !no_dma;
    // DMA kernel: the workgroup tile is UB x VB 32-channel blocks (a <= 32-channel side is not padded to 64).
    // Never needs more slab space than the generic plan hipseg_wgrad_workspace_elems() sizes for.
    const int UB = CU > 32 ? 2 : 1, VB = CV > 32 ? 2 : 1;
    if (dma) {
        pl.UT = cdiv(CU, 32 * UB);
        pl.VT = cdiv(CV, 32 * VB);
        pl.CUp = pl.UT * 32 * UB;
        pl.CVp = pl.VT * 32 * VB;
    }
    WgArgs a;
    a.p0 = p0;
    a.p1 = p1;
    a.q = q;
    a.slabs = slabs;
    a.CU0 = CU0;
    a.CU1 = CU1;
    a.CU = CU;
    a.CV = CV;
    a.CUp = pl.CUp;
    a.CVp = pl.CVp;
    a.B = B;
    a.H = H;
    a.W = W;
    a.ps = mode == HIPSEG_CONVT ? 2 : 1;
    a.PH = a.ps * H;
    a.PW = a.ps * W;
    a.pa = 0;
    a.pb = 0;
    a.tiles_x = pl.tiles_x;
    a.tiles_y = pl.tiles_y;
    a.ntiles = pl.ntiles;
    a.S = pl.S;
    a.UT = pl.UT;
    a.VT = pl.VT;
    a.vec_ok_p = (CU0 % vec == 0) && (CU1 % vec == 0);
    a.vec_ok_q = (CV % vec == 0);
    a.convt_cout = mode == HIPSEG_CONVT ? CU0 : 0;
    a.p_scale = p_scale;
    a.p_shift = p_shift;
#ifdef HIPSEG_ABLATE  // ablation bits give wrong results by design: ablation builds only (scripts/build_variant.sh)
    static const int dbg = getenv("HIPSEG_WGRAD_DEBUG") ? atoi(getenv("HIPSEG_WGRAD_DEBUG")) : 0;
#else
    constexpr int dbg = 0;
#endif
    a.debug = dbg;
    static const bool no_xcd = getenv("HIPSEG_NO_XCD") != nullptr;
    const int nwg = pl.S * pl.UT * pl.VT;
    a.xcd = (!no_xcd && pl.UT * pl.VT > 1 && nwg % 8 == 0) ? nwg / 8 : 0;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const long total = (long)pl.NT * CU * CV;
    const int rgrid = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    for (int ab = 0; ab < 1; ++ab) {
        int rc;
        static const bool no_tr16 = getenv("HIPSEG_NO_WGRAD_TR16") != nullptr;  // A/B switch
        static const bool no_ragged = getenv("HIPSEG_NO_WGRAD_RAGGED") != nullptr;  // A/B switch: whole tiles only
        const bool tr16 = dma && !no_tr16 && !dbg && mode == HIPSEG_CONV3 && CU % 64 == 0 && CV % 64 == 0 &&
                          ((H % 16 == 0 && W % 16 == 0) || !no_ragged) &&
                          (CU1 == 0 || CU0 % 64 == 0) &&
                          (size_t)B * H * W * (size_t)(CU0 > CV ? (CU0 > CU1 ? CU0 : CU1) : (CV > CU1 ? CV : CU1)) * 2 <= ((size_t)1 << 30);
        HS_REQUIRE(tr16 || !p_scale, "conv_wgrad_bnrelu_p: the shape does not take the kernel with the load-side BatchNorm");
        if (tr16)
            rc = launch_tr16(a, s);
        else if (dma)
            rc = pl.NT == 9 ? launch_dma_blocks<9>(a, UB, VB, s) : launch_dma_blocks<1>(a, UB, VB, s);
        else if (dtype == HIPSEG_BF16)
            rc = pl.NT == 9 ? launch<bf16, 9>(a, s) : launch<bf16, 1>(a, s);
        else
            rc = pl.NT == 9 ? launch<float, 9>(a, s) : launch<float, 1>(a, s);
        if (rc) return rc;
        int S = pl.S, sstep = 1;
        if (mode == HIPSEG_CONV3) {  // one launch for any number of slabs
            const dim3 rg(cdiv(CV, 32), cdiv(CU, 4));  // (CVp, CUp are multiples of 32)
            if (pl.S >= 32 && rg.x * rg.y < 128)  // many slabs of a small weight tensor: narrow tiles, 3 tap groups
                hipLaunchKernelGGL((wgrad_reduce3_wide_kernel<32, 1, 3>), dim3(rg.x, CU, 3), dim3(1024), 0, s, slabs, dw, pl.S,
                                   CU, CV, pl.CUp, pl.CVp);
            else if (pl.S >= 8)
                hipLaunchKernelGGL((wgrad_reduce3_wide_kernel<8, 4, 9>), rg, dim3(1024), 0, s, slabs, dw, pl.S, CU, CV, pl.CUp,
                                   pl.CVp);
            else if (pl.S >= 4)
                hipLaunchKernelGGL((wgrad_reduce3_wide_kernel<4, 4, 9>), rg, dim3(512), 0, s, slabs, dw, pl.S, CU, CV, pl.CUp,
                                   pl.CVp);
            else
                hipLaunchKernelGGL((wgrad_reduce3_wide_kernel<2, 4, 9>), rg, dim3(256), 0, s, slabs, dw, pl.S, CU, CV, pl.CUp,
                                   pl.CVp);
            HS_LAUNCH_CHECK("wgrad_reduce(wide)");
            continue;
        }
        // 1x1 / ConvT: many splits of a small weight tensor are pre-reduced in chunks of 16 slabs in place (wide grid),
        // then the (transposing) final pass walks the chunk heads only
        if (pl.S >= 32) {
            sstep = 16;
            S = cdiv(pl.S, sstep);
            const long E = (long)pl.NT * pl.CUp * pl.CVp;
            hipLaunchKernelGGL(colreduce_inplace_kernel, dim3((unsigned)cdiv(E, 64), S), dim3(256), 0, s, slabs, pl.S, E,
                               sstep);
            HS_LAUNCH_CHECK("wgrad_prereduce");
        }
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(rgrid), dim3(256), 0, s, slabs, dw, S, sstep, pl.NT, CU, CV, pl.CUp,
                           pl.CVp, mode, ab);
        HS_LAUNCH_CHECK("wgrad_reduce");
    }
    return HIPSEG_OK;
}

// ---------------------------------------------------------------------------------------------------
// Paired 3x3 weight gradients (the two convolutions of a ConvBlock, same pixel grid, same Cout = CV) in ONE launch
// of wgrad3_tr16_kernel<true> + ONE reduction launch; see the kernel's PAIR comment.
namespace {

bool tr16_shape_ok(int CU0, int CU1, int CV, int B, int H, int W) {
    const int CU = CU0 + CU1;
    const int cmax = CU0 > CV ? (CU0 > CU1 ? CU0 : CU1) : (CV > CU1 ? CV : CU1);
    static const bool no_ragged = getenv("HIPSEG_NO_WGRAD_RAGGED") != nullptr;  // A/B switch: whole tiles only
    return CU % 64 == 0 && CV % 64 == 0 && ((H % 16 == 0 && W % 16 == 0) || !no_ragged) && (CU1 == 0 || CU0 % 64 == 0) &&
           (size_t)B * H * W * (size_t)cmax * 2 <= ((size_t)1 << 30);
}

// pixel splits of the paired launch, 0 = the pair is not taken
int pair_splits(int dtype, int CUa0, int CUa1, int CUb, int CV, int B, int H, int W) {
    static const bool off = getenv("HIPSEG_NO_WGRAD_PAIR") != nullptr || getenv("HIPSEG_NO_WGRAD_TR16") != nullptr ||
                            getenv("HIPSEG_NO_DMA") != nullptr;
    if (off || dtype != HIPSEG_BF16) return 0;
    if (!tr16_shape_ok(CUa0, CUa1, CV, B, H, W) || !tr16_shape_ok(CUb, 0, CV, B, H, W)) return 0;
    const int tA = ((CUa0 + CUa1) / 64) * (CV / 64), tB = (CUb / 64) * (CV / 64);
    const int ncu = device_cus();
    const int ntiles = B * cdiv(H, 16) * cdiv(W, 16);
    int S = ncu / (tA + tB);
    // worth it when the paired grid still fills the chip (>= 90 % of the CUs; alone each layer fills it) and every
    // layer would have been split at least twice on its own (otherwise there is little slab traffic to save); a pixel
    // grid with fewer tiles than that caps the split count of either form, the pair then only adds workgroups
    if (S < 1 || ncu / tA < 2 || ncu / tB < 2) return 0;
    if (S >= ntiles) S = ntiles;
    else if ((long)S * (tA + tB) * 10 < (long)ncu * 9) return 0;
    // Both layers' slabs live in the ONE workspace the caller sized with hipseg_wgrad_workspace_elems() of either layer
    // (include/hipseg.h).  With enough tiles the pair's S * (tA + tB) tile slabs are at most the CU count, like a single
    // layer's; when the TILE count caps the splits (small images) the pair needs S slabs of BOTH layers, more than
    // either layer's own S -- such shapes are not paired (round 4: found as a GPU fault on a 2 x 16 x 24 level once
    // ragged grids reached this path; whole-tile grids of few tiles had been writing past the workspace unnoticed).
    const size_t need = (size_t)S * 9 * (size_t)(CUa0 + CUa1 + CUb) * CV;
    const size_t wa = hipseg_wgrad_workspace_elems(HIPSEG_CONV3, CUa0 + CUa1, CV, B, H, W);
    const size_t wb = hipseg_wgrad_workspace_elems(HIPSEG_CONV3, CUb, CV, B, H, W);
    if (need > (wa > wb ? wa : wb)) return 0;
    return S;
}

WgArgs tr16_args(const void* p0, int CU0, const void* p1, int CU1, const void* q, int CV, float* slabs, int S, int B, int H,
                 int W) {
    WgArgs a;
    a.p0 = p0;
    a.p1 = p1;
    a.q = q;
    a.slabs = slabs;
    a.CU0 = CU0;
    a.CU1 = CU1;
    a.CU = CU0 + CU1;
    a.CV = CV;
    a.UT = a.CU / 64;
    a.VT = CV / 64;
    a.CUp = a.CU;
    a.CVp = CV;
    a.B = B;
    a.H = H;
    a.W = W;
    a.ps = 1;
    a.PH = H;
    a.PW = W;
    a.pa = 0;
    a.pb = 0;
    a.tiles_x = cdiv(W, 16);  // (ragged edges: wgrad3_tr16_kernel<PAIR, RAGGED = true>)
    a.tiles_y = cdiv(H, 16);
    a.ntiles = B * a.tiles_x * a.tiles_y;
    a.S = S;
    a.vec_ok_p = 1;
    a.vec_ok_q = 1;
    a.convt_cout = 0;
    a.p_scale = nullptr;
    a.p_shift = nullptr;
    a.debug = 0;
    a.xcd = 0;
    return a;
}

}  // namespace

extern "C" int hipseg_conv_wgrad_pair_applies(int dtype, int CUa0, int CUa1, int CUb, int CV, int B, int H, int W) {
    return pair_splits(dtype, CUa0, CUa1, CUb, CV, B, H, W) > 0 ? 1 : 0;
}

extern "C" int hipseg_conv_wgrad_pair(int dtype, const void* pa0, int CUa0, const void* pa1, int CUa1, const void* qa,
                                      float* dwa, const void* pb, int CUb, const void* qb, float* dwb, int CV, float* slabs,
                                      int B, int H, int W, hipseg_stream_t stream) {
    HS_REQUIRE(pa0 && qa && dwa && pb && qb && dwb && slabs, "conv_wgrad_pair: null operand");
    HS_REQUIRE((CUa1 == 0) == (pa1 == nullptr), "conv_wgrad_pair: pa1/CUa1 mismatch");
    const int S = pair_splits(dtype, CUa0, CUa1, CUb, CV, B, H, W);
    HS_REQUIRE(S > 0, "conv_wgrad_pair: the pair is not taken for this shape (ask hipseg_conv_wgrad_pair_applies first)");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int CUa = CUa0 + CUa1;
    float* slabs_b = slabs + (size_t)S * 9 * CUa * CV;  // (within hipseg_wgrad_workspace_elems of either layer)
    const WgArgs a = tr16_args(pa0, CUa0, pa1, CUa1, qa, CV, slabs, S, B, H, W);
    const WgArgs b = tr16_args(pb, CUb, nullptr, 0, qb, CV, slabs_b, S, B, H, W);
    if (int rc = launch_tr16_pair(a, b, s)) return rc;
    // one reduction launch over both weight tensors (grid rows of the second one behind the first one's)
    if (S >= 32 && cdiv(CV, 32) * (cdiv(CUa, 4) + cdiv(CUb, 4)) < 128)
        hipLaunchKernelGGL((wgrad_reduce3_wide_kernel<32, 1, 3>), dim3(cdiv(CV, 32), CUa + CUb, 3), dim3(1024), 0, s, slabs, dwa,
                           S, CUa, CV, CUa, CV, slabs_b, dwb, CUb, CUb, CUa);
    else if (S >= 8)
        hipLaunchKernelGGL((wgrad_reduce3_wide_kernel<8, 4, 9>), dim3(cdiv(CV, 32), CUa / 4 + CUb / 4), dim3(1024), 0, s, slabs,
                           dwa, S, CUa, CV, CUa, CV, slabs_b, dwb, CUb, CUb, CUa / 4);
    else if (S >= 4)
        hipLaunchKernelGGL((wgrad_reduce3_wide_kernel<4, 4, 9>), dim3(cdiv(CV, 32), CUa / 4 + CUb / 4), dim3(512), 0, s, slabs,
                           dwa, S, CUa, CV, CUa, CV, slabs_b, dwb, CUb, CUb, CUa / 4);
    else
        hipLaunchKernelGGL((wgrad_reduce3_wide_kernel<2, 4, 9>), dim3(cdiv(CV, 32), CUa / 4 + CUb / 4), dim3(256), 0, s, slabs,
                           dwa, S, CUa, CV, CUa, CV, slabs_b, dwb, CUb, CUb, CUa / 4);
    HS_LAUNCH_CHECK("wgrad_reduce(pair)");
    return HIPSEG_OK;
}
