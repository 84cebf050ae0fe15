// ConvTranspose2d(kernel 2, stride 2) backward-filter AND bias gradient in one kernel + one reduction launch, bf16, gfx950.
// Replaces the autograd backward of nn.ConvTranspose2d(Cin, Cout, 2, stride=2) w.r.t. weight and bias
// (/root/reference/models/processing_blocks.py:102,106,128,132):
//     dW[ci][co][a][b] = sum_{n,i,j} X[n,i,j,ci] * dY[n,2i+a,2j+b,co]          db[co] = sum_{n,y,x} dY[n,y,x,co]
//
// One GEMM: rows u = ci (P = X, NHWC), columns v = (a, b, co) = 4 Cout "channels" of a strided VIEW of dY -- the X pixel
// (i, j) sees the two contiguous runs dY[2i + a][2j .. 2j + 1][:] (2 Cout values each) -- K = the X pixels.  db is the
// column sum of that same view (every dY element appears in it exactly once), so it falls out of the Q fragments the
// MFMAs already hold in registers: no second pass over dY (round 3: a column-sum launch + its finalize per layer, and
// the weight gradient itself went through the generic one-tap kernel with 64 x 64 tiles: 4 transposed LDS reads per
// MFMA and 32 FLOP per staged byte = 0.08 of the MFMA peak on the 512 -> 256 stage).
//
// Workgroup tile 128 u x 128 v over 8 x 16 = 128 pixel tiles of X; 8 waves = 2 pixel-row halves ("k-split", summed
// through LDS once at the end) x 2 x 2 wave tiles of 64 u x 64 v = 4 x 4 accumulator blocks of v_mfma_f32_16x16x32_bf16
// (64 registers): a K step of 32 pixels (two tile rows) is 8 + 8 transposed reads (ds_read_b64_tr_b16) for 16 MFMAs.
// LDS keeps the global [pixel][128 channels] order (a 1-KiB LDS-DMA piece = 4 pixels x 256 B); the 32-byte channel
// slots of a pixel row are XOR-swizzled with (pixel & 7) on the DMA's SOURCE side, so the 4 pixels a 16-lane group of a
// transposed read touches -- and the 8 consecutive pixels of two groups -- sit in different bank groups.  Two 64-KiB
// stages: the next tile streams in under the current tile's MFMAs.  Partial results go to fp32 slabs [S][CUp][CVp] (+
// [S][CVp] column sums) that the reduction launch sums in fixed order into the parameter layouts (deterministic).
#include <stdlib.h>

#include "common.h"

namespace {

struct TwArgs {
    const void* x;
    const void* dy;
    float* slabs;
    float* bslab;
    int Cin, Cout, CV, B, H, W;
    int tiles_x, tiles_y, ntiles, S, UT, VT, CUp, CVp, xcd;
};

struct TwGeo {
    static constexpr int TH = 8, TW = 16, NPIX = TH * TW, UC = 128, VC = 128, NW = 8;
    static constexpr int ROWB = 256;                       // bytes per pixel row of an LDS image (128 bf16)
    static constexpr int IMG = NPIX * ROWB, BUF = 2 * IMG;  // P image, Q image
    static constexpr int NPC = IMG / 1024, NPWI = NPC / NW; // 1-KiB pieces per image; per wave and image
    static constexpr size_t RED = (size_t)4 * 16 * 64 * 16 + 2 * 4 * 64 * 4;  // 4 parked accumulator sets + bias partials
    static constexpr size_t LDS = 2 * BUF;
    static_assert(RED <= LDS && LDS <= 160 * 1024 && NPC % NW == 0, "geometry");
};

__device__ __forceinline__ int xcd_block(int bid, int cpx) { return cpx ? (bid & 7) * cpx + (bid >> 3) : bid; }

__global__ __launch_bounds__(512, 1) void convt_wgrad_kernel(TwArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef TwGeo G;
    typedef __attribute__((address_space(3))) void lds_void;
    typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
    constexpr unsigned OOB = 0x80000000u;
    constexpr int NPWI = G::NPWI, IMG = G::IMG, BUF = G::BUF;

    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kh = wave >> 2, wu = wave & 1, wv = (wave >> 1) & 1;
    const int bid = xcd_block(blockIdx.x, a.xcd);
    const int vt = bid % a.VT, ut = (bid / a.VT) % a.UT, s = bid / (a.VT * a.UT);
    const int u0 = ut * G::UC, v0 = vt * G::VC;
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_void*)smem;

    // ---- staging: per-lane byte offsets relative to the tile origin (constants), the tile origin is a scalar offset
    const int cstrideP = a.Cin * 2;                 // bytes between X pixels
    const int cstrideQ = 2 * a.Cout * 2;            // bytes between the dY runs of consecutive X pixels (two dY pixels)
    const int rowQ = 2 * a.W * a.Cout * 2;          // bytes of one dY row
    const v4i_t r_p = make_rsrc(a.x, (unsigned)((size_t)a.B * a.H * a.W * cstrideP));
    const v4i_t r_q = make_rsrc(a.dy, (unsigned)((size_t)a.B * 2 * a.H * rowQ));
    const int prow = lane >> 4, d16 = lane & 15;   // pixel of the piece, 16-byte destination slot of its 256-B row
    const int ph = ((wave & 1) << 2) + prow;       // (pixel & 7) of every piece of this wave: pieces are j * 8 + wave
    const int csrc = (((d16 >> 1) ^ ph) << 4) + ((d16 & 1) << 3);  // source channel of the slot (swizzle on the source)
    unsigned pofs[NPWI], qofs[NPWI];
    {
        const int cu = u0 + csrc, cv = v0 + csrc;
        const int av = cv / (2 * a.Cout), wi = cv - av * 2 * a.Cout;  // dY row parity and offset inside the run
#pragma unroll
        for (int j = 0; j < NPWI; ++j) {
            const int pp = (j * G::NW + wave) * 4 + prow, r = pp >> 4, c = pp & 15;
            pofs[j] = cu < a.Cin ? (unsigned)((r * a.W + c) * cstrideP + csrc * 2) : OOB;
            qofs[j] = cv < a.CV ? (unsigned)((2 * r + av) * rowQ + c * cstrideQ + wi * 2) : OOB;
        }
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
    // Ragged image edges (H % 8 or W % 16 != 0: ClipUnet's 28 x 28 and 56 x 56 stages): a pixel of the tile that lies
    // outside the image must contribute nothing to dW and to the dY column sums, so its lanes do not fetch (zero fill).
    // The pixel of a lane is (row 2 j + (wave >> 2), column (4 wave + prow) & 15) of the tile: the column is a per-lane
    // constant (one compare per tile), the row is wave-uniform (a scalar compare per piece).
    const int lcol = (4 * wave + prow) & 15, lrow0 = wave >> 2;
    struct TileS {
        unsigned soP, soQ;
        int rows;     // tile rows inside the image
        bool colok;   // this lane's tile column is inside the image
    };
    auto tile_scalars = [&]() {
        TileS t;
        const int y0 = nty * G::TH, x0 = ntx * G::TW;
        t.soP = (unsigned)(((nimg * a.H + y0) * a.W + x0) * cstrideP + u0 * 2);
        t.soQ = (unsigned)((nimg * 2 * a.H + 2 * y0) * rowQ + x0 * cstrideQ);
        t.rows = a.H - y0;
        t.colok = lcol < a.W - x0;
        return t;
    };
    auto pieceP = [&](int j, unsigned base, const TileS& t) {
        const bool ok = t.colok && 2 * j + lrow0 < t.rows;
        dma_piece(r_p, base + (j * G::NW + wave) * 1024, ok ? pofs[j] : OOB, t.soP);
    };
    auto pieceQ = [&](int j, unsigned base, const TileS& t) {
        const bool ok = t.colok && 2 * j + lrow0 < t.rows;
        dma_piece(r_q, base + IMG + (j * G::NW + wave) * 1024, ok ? qofs[j] : OOB, t.soQ);
    };

    // ---- fragment addresses (absolute LDS addresses).  Transposed read: 16-lane group g = lane >> 4 is k octet g; its
    // lane 4q + p addresses pixel 4g + q of a tile row, channels 4p .. 4p + 3 of a 16-channel block.  The swizzle
    // phase (pixel & 7) = (4g + q) & 7 is a per-lane constant: one base per 16-channel block, rows are immediates.
    const int g4q = 4 * (lane >> 4) + ((lane & 15) >> 2), p4 = lane & 3;
    unsigned pb[4], qb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        pb[i] = lds0 + (unsigned)(g4q * G::ROWB + (((wu * 4 + i) ^ (g4q & 7)) << 5) + p4 * 8);
        qb[i] = lds0 + (unsigned)(IMG + g4q * G::ROWB + (((wv * 4 + i) ^ (g4q & 7)) << 5) + p4 * 8);
    }
    auto rd = [&](unsigned base, int row) {
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(uintptr_t)base + row * (G::TW * G::ROWB / 8));
        const bf16x4 hi =
            __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(uintptr_t)base + (row + 1) * (G::TW * G::ROWB / 8));
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[i][k] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum[4] = {0.f, 0.f, 0.f, 0.f};
    const bool do_bias = ut == 0 && wu == 0;  // (wave-uniform) one wave per (v half, pixel-row half) sums the dY columns

    // prologue: the first tile
    if (ntile < a.ntiles) {
        const TileS t0 = tile_scalars();
#pragma unroll
        for (int j = 0; j < NPWI; ++j) {
            pieceP(j, lds0, t0);
            pieceQ(j, lds0, t0);
        }
    }
    advance();
    int cur = 0;
    for (int tile = s; tile < a.ntiles; tile += a.S) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of the tile have landed
        __builtin_amdgcn_s_barrier();                      // everyone's have; everyone left the other stage
        __builtin_amdgcn_sched_barrier(0);
        const bool more = ntile < a.ntiles;
        const unsigned nbase = lds0 + (cur ^ 1) * BUF;
        const TileS tn = tile_scalars();
        bf16x8 pf[2][4], qf[2][4];
        const int row0 = 4 * kh;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            pf[0][i] = rd(pb[i], row0);
            qf[0][i] = rd(qb[i], row0);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            if (ks == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    pf[1][i] = rd(pb[i], row0 + 2);
                    qf[1][i] = rd(qb[i], row0 + 2);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (more) {  // (workgroup-uniform) two pieces of the next tile per four MFMAs
                    if (ks == 0)
                        pieceP(i, nbase, tn);
                    else
                        pieceQ(i, nbase, tn);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    acc[i][k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf[ks][i], qf[ks][k], acc[i][k], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (do_bias) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int e = 0; e < 8; ++e) bsum[k] += (float)qf[ks][k][e];
            }
        }
        advance();
        {
            const unsigned dd = cur ? (unsigned)-BUF : (unsigned)BUF;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                pb[i] += dd;
                qb[i] += dd;
            }
        }
        cur ^= 1;
    }
    // ---- the pixel-row halves meet in LDS (the stages are idle now); the kh == 0 waves write the slab [S][CUp][CVp].
    // Accumulator block (i, k): lane holds column v = lane & 15, rows u = 4 (lane >> 4) + e.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    f32x4* red = reinterpret_cast<f32x4*>(smem) + (size_t)(wave & 3) * (16 * 64);
    float* bred = reinterpret_cast<float*>(smem + (size_t)4 * 16 * 64 * 16);  // [wv][4 blocks][64 lanes]
    if (kh == 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int k = 0; k < 4; ++k) red[(i * 4 + k) * 64 + lane] = acc[i][k];
        if (do_bias) {
#pragma unroll
            for (int k = 0; k < 4; ++k) bred[(wv * 4 + k) * 64 + lane] = bsum[k];
        }
    }
    __syncthreads();
    if (kh == 0) {
        const int vcol = lane & 15, ur = 4 * (lane >> 4);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const f32x4 o = red[(i * 4 + k) * 64 + lane];
                const int u = u0 + wu * 64 + i * 16 + ur, v = v0 + wv * 64 + k * 16 + vcol;
                float* dst = a.slabs + ((size_t)s * a.CUp + u) * a.CVp + v;
#pragma unroll
                for (int e = 0; e < 4; ++e) dst[(size_t)e * a.CVp] = acc[i][k][e] + o[e];
            }
        if (do_bias) {
            // column v = lane & 15 of block k: the four 16-lane groups hold the four k octets' partial sums
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float t = bsum[k] + bred[(wv * 4 + k) * 64 + lane];
                t += __shfl_xor(t, 16, 64);
                t += __shfl_xor(t, 32, 64);
                if (lane < 16) a.bslab[(size_t)s * a.CVp + v0 + wv * 64 + k * 16 + lane] = t;
            }
        }
    }
#else
    (void)a;
#endif
}

// Sum the S slabs in slab order and scatter into the parameter layouts: dw (Cin, Cout, 2, 2), db (Cout).
// grid (CVp / 64, Cin + 1): row y < Cin: u = y, 64 columns v = (a, b, co); row Cin: the bias, 64 output channels per
// block (x < ceil(Cout / 64)).  A block = 64 columns x NS slab slices; slice partials meet in LDS, summed in slice order.
template <int NS>
__global__ __launch_bounds__(64 * NS) void convt_wgrad_reduce_kernel(const float* __restrict__ slabs,
                                                                      const float* __restrict__ bslab,
                                                                      float* __restrict__ dw, float* __restrict__ db, int S,
                                                                      int Cin, int Cout, int CUp, int CVp) {
    __shared__ float red[NS][64];
    const int tid = threadIdx.x, c = tid & 63, sl = tid >> 6;
    float acc = 0.f;
    if ((int)blockIdx.y < Cin) {
        const int u = blockIdx.y, v = blockIdx.x * 64 + c;
        const float* src = slabs + (size_t)u * CVp + v;
        const size_t ss = (size_t)CUp * CVp;
        for (int s = sl; s < S; s += NS) acc += src[s * ss];
        red[sl][c] = acc;
        __syncthreads();
        if (sl == 0 && v < 4 * Cout) {
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < NS; ++k) t += red[k][c];
            const int ab = v / Cout, co = v - ab * Cout;
            dw[((size_t)u * Cout + co) * 4 + ab] = t;
        }
    } else {
        const int co = blockIdx.x * 64 + c;
        if (blockIdx.x * 64 >= (unsigned)Cout) return;  // (whole block)
        if (co < Cout) {
            for (int s = sl; s < S; s += NS) {
                const float* row = bslab + (size_t)s * CVp + co;
                acc += (row[0] + row[Cout]) + (row[2 * Cout] + row[3 * Cout]);
            }
        }
        red[sl][c] = acc;
        __syncthreads();
        if (sl == 0 && co < Cout) {
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < NS; ++k) t += red[k][c];
            db[co] = t;
        }
    }
}

struct TwPlan {
    int UT, VT, S, ntiles, tiles_x, tiles_y, CUp, CVp;
};

TwPlan tw_plan(int Cin, int Cout, int B, int H, int W) {
    TwPlan p;
    p.UT = cdiv(Cin, TwGeo::UC);
    p.VT = cdiv(4 * Cout, TwGeo::VC);
    p.CUp = p.UT * TwGeo::UC;
    p.CVp = p.VT * TwGeo::VC;
    p.tiles_x = cdiv(W, TwGeo::TW);
    p.tiles_y = cdiv(H, TwGeo::TH);
    p.ntiles = B * p.tiles_x * p.tiles_y;
    int S = device_cus() / (p.UT * p.VT);
    if (S < 1) S = 1;
    if (S > p.ntiles) S = p.ntiles;
    p.S = S;
    return p;
}

bool tw_applies(int dtype, int Cin, int Cout, int B, int H, int W) {
    static const bool off = getenv("HIPSEG_NO_CONVT_WGRAD") != nullptr;  // A/B switch: generic one-tap kernel + column sum
    if (off || dtype != HIPSEG_BF16 || Cin % 8 || Cout % 8) return false;
    const size_t xb = (size_t)B * H * W * Cin * 2, yb = (size_t)B * 4 * H * W * Cout * 2;
    return xb <= ((size_t)1 << 30) && yb <= ((size_t)1 << 30);
}

}  // namespace

extern "C" size_t hipseg_convT_wgrad_workspace_elems(int Cin, int Cout, int B, int H, int W) {
    // the fused kernel's slabs + column sums, or (shapes it does not take) the generic weight gradient's slabs followed by
    // the column-sum partials
    const TwPlan p = tw_plan(Cin, Cout, B, H, W);
    const size_t fused = (size_t)p.S * p.CUp * p.CVp + (size_t)p.S * p.CVp;
    size_t generic = hipseg_wgrad_workspace_elems(HIPSEG_CONVT, Cout, Cin, B, H, W);
    for (int dt = 0; dt < 2; ++dt) {
        const size_t c = generic + (size_t)hipseg_colsum_blocks((long)B * 4 * H * W, Cout, dt) * Cout;
        if (c > generic) generic = c;
    }
    return fused > generic ? fused : generic;
}

extern "C" int hipseg_convT_wgrad_bias(int dtype, const void* dy, const void* x, float* dw, float* db, float* work, int B,
                                       int H, int W, int Cin, int Cout, hipseg_stream_t stream) {
    HS_REQUIRE(dtype == HIPSEG_F32 || dtype == HIPSEG_BF16, "convT_wgrad_bias: bad dtype %d", dtype);
    HS_REQUIRE(dy && x && dw && db && work && B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "convT_wgrad_bias: bad arguments");
    if (!tw_applies(dtype, Cin, Cout, B, H, W)) {
        if (int rc = hipseg_conv_wgrad(dtype, HIPSEG_CONVT, dy, Cout, nullptr, 0, x, Cin, dw, work, B, H, W, stream)) return rc;
        float* part = work + hipseg_wgrad_workspace_elems(HIPSEG_CONVT, Cout, Cin, B, H, W);
        return hipseg_colsum(dtype, dy, (long)B * 4 * H * W, Cout, part, db, stream);
    }
    const TwPlan p = tw_plan(Cin, Cout, B, H, W);
    TwArgs a;
    a.x = x;
    a.dy = dy;
    a.slabs = work;
    a.bslab = work + (size_t)p.S * p.CUp * p.CVp;
    a.Cin = Cin;
    a.Cout = Cout;
    a.CV = 4 * Cout;
    a.B = B;
    a.H = H;
    a.W = W;
    a.tiles_x = p.tiles_x;
    a.tiles_y = p.tiles_y;
    a.ntiles = p.ntiles;
    a.S = p.S;
    a.UT = p.UT;
    a.VT = p.VT;
    a.CUp = p.CUp;
    a.CVp = p.CVp;
    const int nwg = p.S * p.UT * p.VT;
    a.xcd = (p.UT * p.VT > 1 && nwg % 8 == 0) ? nwg / 8 : 0;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (int rc = hs_set_max_lds(reinterpret_cast<const void*>(&convt_wgrad_kernel), TwGeo::LDS)) return rc;
    hipLaunchKernelGGL(convt_wgrad_kernel, dim3((unsigned)nwg), dim3(512), TwGeo::LDS, s, a);
    HS_LAUNCH_CHECK("convT_wgrad");
    const dim3 rg((unsigned)(p.CVp / 64), (unsigned)(Cin + 1));
    if (p.S >= 64)
        hipLaunchKernelGGL(convt_wgrad_reduce_kernel<16>, rg, dim3(1024), 0, s, a.slabs, a.bslab, dw, db, p.S, Cin, Cout, p.CUp,
                           p.CVp);
    else
        hipLaunchKernelGGL(convt_wgrad_reduce_kernel<4>, rg, dim3(256), 0, s, a.slabs, a.bslab, dw, db, p.S, Cin, Cout, p.CUp,
                           p.CVp);
    HS_LAUNCH_CHECK("convT_wgrad_reduce");
    return HIPSEG_OK;
}
