// Launch arguments shared by the implicit-GEMM convolution kernels (conv_igemm.hip, conv3_m16.hip).
#pragma once
#include "common.h"

struct ConvArgs {
    const void* in0;
    const void* in1;
    const void* wp;
    const float* bias;
    const float* post_scale;  // inference epilogue (hipseg_conv_affine_relu): y = relu(conv * post_scale[n] + bias[n]),
                              // `bias` then holds the folded shift; NULL = plain conv + bias
    void* out0;
    void* out1;
    float* stats;
    // data gradient feeding a BatchNorm backward (hipseg_conv3_dgrad_bnstats): bw_x = the tensor that BatchNorm
    // normalised (same NHWC shape as out0), bw_bn = its [mean | invstd | scale | shift] x N vectors; the kernel then
    // writes [sum g | sum g * xhat] rows of the output into `stats`.  NULL = off.
    const void* bw_x;
    const float* bw_bn;
    // BatchNorm + ReLU applied on LOAD (hipseg_conv3_bnrelu_in, weights-stationary kernel only): the convolution runs
    // over relu(in0 * ld_scale[c] + ld_shift[c]), zero-padded.  NULL = off.
    const float* ld_scale;
    const float* ld_shift;
    int C0, C1, N0, N1;
    int B, H, W;    // GEMM-M pixel grid
    int Hi, Wi;     // input spatial dims
    int K, Kp, N, Np;
    int tiles_x, tiles_y, ntn;
    int vec_ok;
    int ncu;      // compute units of the current device (grid of the persistent weights-stationary kernel)
    int debug;  // ablation bits (HIPSEG_IGEMM_DEBUG): 1 skip A staging, 2 skip B staging, 4 skip MFMA, 8 skip epilogue
    int xcd;    // XCD-aware workgroup order: 0 off, else grid / 8 (see xcd_block)
};

// Workgroups are dealt round-robin to the 8 XCDs (each with a private L2).  Give every XCD a CONTIGUOUS run of
// logical workgroup ids instead, so the N-tile workgroups of one pixel tile (same halo tile) and neighbouring
// pixel tiles (shared halo rows) hit the same L2 (guide T1; needs grid % 8 == 0).
__device__ __forceinline__ int xcd_block(int bid, int cpx) { return cpx ? (bid & 7) * cpx + (bid >> 3) : bid; }
