/*
 * hipseg.h -- C ABI of the MI355X-native (gfx950) U-Net / ClipUnet training hot path.
 *
 * The reference (MattiDeBeer/image-segmentation) has no FFI: its hot path is the
 * implicit ATen op set behind models/processing_blocks.py, models/UNet.py,
 * models/CLIP_models.py and models/losses.py.  Each entry point
 * below replaces one of those implicit ops (cited as file:line under /root/reference)
 * and is what a Python/ctypes, cgo or JNI binding for this path would bind.
 *
 * Conventions
 *   - plain pointers + sizes, no torch types.  All pointers are DEVICE pointers
 *     (HBM) unless named *_host.  `stream` is a hipStream_t passed as void*.
 *   - activations are dense NHWC ("channels-last"): elem(n,y,x,c) = base[((n*H+y)*W+x)*C+c].
 *   - dtype: HIPSEG_F32 (float) or HIPSEG_BF16 (bfloat16 storage, fp32 accumulate).
 *     Parameters, BN statistics, gradients of parameters and all reductions are fp32.
 *   - every function only enqueues work on `stream` (no allocation, no sync: safe
 *     under hipGraph capture) and returns 0 on success or a negative HIPSEG_E* code;
 *     hipseg_last_error() gives the message for the calling thread.
 *   - NOT exported, on purpose: the data-parallel gradient all-reduce (SURVEY 8b lists a possible
 *     `bucket_allreduce(ptr, count, dtype, comm, stream)`).  The host of this path is Python on
 *     PyTorch-ROCm, which owns the RCCL communicator (torch.distributed, backend "nccl"); wrapping
 *     ncclAllReduce behind a second C entry point would only duplicate that plumbing.  The bucket
 *     logic lives in image-segmentation_amd/hipseg/ddp.py (HipDDP, replacing
 *     scripts/train_distributed.py:35); the kernels of this library write parameter gradients
 *     straight into its flat fp32 buckets.
 */
#ifndef HIPSEG_H
#define HIPSEG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HIPSEG_F32 0
#define HIPSEG_BF16 1

#define HIPSEG_OK 0
#define HIPSEG_EINVAL (-1) /* bad argument (shape/dtype/null) */
#define HIPSEG_EHIP (-2)   /* HIP runtime error at launch */

/* implicit-GEMM "conv" modes (hipseg_conv_igemm) */
#define HIPSEG_CONV3 0   /* 3x3, pad 1, stride 1                                  */
#define HIPSEG_CONV1 1   /* 1x1                                                   */
#define HIPSEG_CONV2S2 2 /* 2x2, stride 2 (the data-gradient of ConvTranspose2d)   */
#define HIPSEG_CONVT 3   /* ConvTranspose2d k2 s2 forward: 1x1 GEMM + pixel shuffle */

typedef void* hipseg_stream_t;

const char* hipseg_last_error(void);
int hipseg_abi_version(void);

/* ---- packed-weight geometry (pure host functions) -------------------------------- */
/* K (reduction channels) is padded to the kernel's K-chunk, N (GEMM columns) to the
 * kernel's column tile.  Packed layout: [tap][Kp/G][Np][G], G = 8 (bf16) or 1 (f32).  */
int hipseg_kpad(int K, int dtype);
int hipseg_npad(int N);
/* rows of the BatchNorm statistics workspace for a (B,H,W) pixel grid (one per 64 output pixels) */
/* UPPER BOUND of the rows of the BatchNorm statistics workspace for a (B,H,W) pixel grid (one per 64 output
 * pixels): sizes the allocation */
int hipseg_conv_mtiles(int B, int H, int W);
/* rows hipseg_conv_igemm() actually WRITES for a call with these arguments (the persistent kernels write one
 * row per workgroup wave-group instead of one per 64 pixels); pass it to hipseg_bn_finalize() */
int hipseg_conv_stats_rows(int dtype, int mode, int C0, int C1, int N0, int N1, int B, int H, int W);

/* Conv2d weight (Cout,Cin,kh,kw) fp32 -> packed [tap][Kp/G][Np][G] in `dtype`.
 * transpose=0: forward operand      (K = Cin, N = Cout, tap = ky*kw+kx)
 * transpose=1: data-gradient operand (K = Cout, N = Cin, tap flipped: 3x3 conv dgrad is a
 *              3x3 conv of dY with W[co][ci][2-ky][2-kx]).
 * replaces: the implicit weight layout transforms inside cuDNN for nn.Conv2d
 * (models/processing_blocks.py:43,46). */
int hipseg_pack_conv_weight(const float* w, void* wp, int dtype, int Cout, int Cin, int ksize,
                            int transpose, hipseg_stream_t stream);
/* both operands of one Conv2d weight (transpose=0 into wp, transpose=1 into wpt) in one launch */
int hipseg_pack_conv_weight_both(const float* w, void* wp, void* wpt, int dtype, int Cout, int Cin,
                                 int ksize, hipseg_stream_t stream);
/* ConvTranspose2d weight (Cin,Cout,2,2) fp32.
 * transpose=0: forward operand  (HIPSEG_CONVT: 1 tap, K = Cin, N = 4*Cout, n = (a*2+b)*Cout+co)
 * transpose=1: data-gradient operand (HIPSEG_CONV2S2: 4 taps (a,b), K = Cout, N = Cin).
 * (models/processing_blocks.py:102,128) */
int hipseg_pack_convT_weight(const float* w, void* wp, int dtype, int Cin, int Cout, int transpose,
                             hipseg_stream_t stream);

/* All conv / ConvT weights of a model in ONE launch (the per-layer calls above are latency-bound helpers).
 * A descriptor table is filled on the host (hipseg_pack_desc_fill into a buffer of n * hipseg_pack_desc_size()
 * bytes), copied to the device by the caller, and replayed every step:
 *   kind 0: Conv2d weight (Cout,Cin,k,k) -> wp = forward operand, wpt = data-gradient operand
 *           (= hipseg_pack_conv_weight_both);  kind 1: ConvTranspose2d weight (Cin,Cout,2,2) -> wp = forward
 *           operand, wpt = data-gradient operand (= hipseg_pack_convT_weight transpose 0 / 1).
 * max_total = the largest packed operand in elements (sizes the grid). */
size_t hipseg_pack_desc_size(void);
int hipseg_pack_desc_fill(void* host_descs, int index, const float* w, void* wp, void* wpt, int kind, int dtype,
                          int Cout, int Cin, int ksize);
int hipseg_pack_batch(const void* dev_descs, int n, int dtype, long max_total, hipseg_stream_t stream);

/* ---- implicit-GEMM convolution (MFMA) ---------------------------------------------
 * out[n,y,x,:] = bias + sum_taps sum_c in[n, tap(y,x), c] * W[tap][c][:]
 *   in0/in1 : up to two NHWC sources concatenated along channels (C0 + C1 = K); in1 may be
 *             NULL with C1 = 0.  This is torch.cat([x, skip], 1) eliminated
 *             (models/processing_blocks.py:108).
 *   out0/out1: output channel range split over two NHWC tensors (N0 + N1 = N); out1 may be
 *             NULL.  Used by the data-gradient of a dual-source conv.
 *   (H, W)  : the GEMM-M pixel grid = output grid for CONV3/CONV1/CONV2S2 (input grid is
 *             2H x 2W for CONV2S2), INPUT grid for CONVT (output is 2H x 2W, N0 = Cout).
 *   stats   : NULL or float[hipseg_conv_mtiles()][2][N]: partial column sums and sums of squares of the fp32
 *             results (the BatchNorm batch-statistics partials, fused epilogue); rows [0, hipseg_conv_stats_rows())
 *             are written, each output pixel is counted in exactly one of them.
 * replaces: aten::conv2d 3x3/1x1 (processing_blocks.py:43,46), its dgrad, and
 *           aten::conv_transpose2d fwd/dgrad (processing_blocks.py:102,106). */
int hipseg_conv_igemm(int dtype, int mode, const void* in0, int C0, const void* in1, int C1,
                      const void* wp, const float* bias, void* out0, int N0, void* out1, int N1,
                      float* stats, int B, int H, int W, hipseg_stream_t stream);

/* ---- inference: conv3x3 -> BatchNorm2d (running statistics) -> ReLU in one kernel ------------
 * out = relu(conv3x3(cat(in0, in1)) * scale[n] + shift[n]), the affine applied to the fp32 accumulators in the
 * epilogue.  The caller folds the conv bias and the BatchNorm into the two vectors:
 *   scale = gamma / sqrt(running_var + eps),  shift = beta + (conv_bias - running_mean) * scale.
 * Same operands, packing and dispatch as hipseg_conv_igemm(HIPSEG_CONV3).
 * replaces: Conv2d -> BatchNorm2d.eval() -> ReLU of ConvBlock (models/processing_blocks.py:42-48) under
 *           model.eval() + torch.no_grad() (the validation loops, models/model_wrappers.py:193-215). */
int hipseg_conv_affine_relu(int dtype, const void* in0, int C0, const void* in1, int C1, const void* wp,
                            const float* scale, const float* shift, void* out, int N, int B, int H, int W,
                            hipseg_stream_t stream);

/* Data gradient of ConvTranspose2d(k2, s2) (processing_blocks.py:102,106: `self.up`) + the BatchNorm-backward sums of its
 * output, in one kernel.  The ConvTranspose2d's input is the previous ConvBlock's activated output (models/UNet.py:66-71:
 * bottleneck -> dec1.up, dec_k.conv -> dec_k+1.up), so dx IS that block's dout: dx = CONV2S2(dy (Cout, 2H x 2W), wp = the
 * data-gradient operand of hipseg_pack_convT) on the H x W grid, and partial receives hipseg_convT_dgrad_bnstats_rows()
 * rows of [2][Cin] floats = [sum g | sum g * xhat], g = dx where x * scale + shift > 0 (x = that block's second
 * pre-normalisation tensor, bn = its [mean | invstd | scale | shift] vectors) -- the rows hipseg_bn_bwd_reduce would
 * produce from a second pass over dx and x (hipseg_convblock_t::dout_rows).  rows() == 0: no kernel with that epilogue
 * takes the shape (use hipseg_conv_igemm in mode HIPSEG_CONV2S2 + hipseg_bn_bwd_reduce). */
int hipseg_convT_dgrad_bnstats_rows(int dtype, int Cout, int Cin, int B, int H, int W);
int hipseg_convT_dgrad_bnstats(int dtype, const void* dy, int Cout, const void* wp, void* dx, int Cin, const void* x,
                               const float* bn, float* partial, int B, int H, int W, hipseg_stream_t stream);

/* ---- BatchNorm + ReLU applied in the CONSUMER's load path (round 4) ------------------------------
 * The second convolution of a ConvBlock reads relu(bn(raw1)); at the full-resolution levels (<= 64 channels) writing that
 * activated tensor and reading it back (hipseg_bn_relu_apply: 2 x 134 MB at 64 channels x 16 x 256 x 256) costs more
 * than transforming the pre-normalisation tensor on its way into the kernels that consume it:
 *   hipseg_conv3_bnrelu_in     : out = conv3x3(relu(in * scale[c] + shift[c]), zero-padded) (+ bias, + statistics rows as
 *                                hipseg_conv_igemm); the wave that LDS-DMA'd a piece of the halo tile rewrites it in LDS.
 *   hipseg_conv_wgrad_bnrelu_p : the 3x3 weight gradient against relu(p * scale[u] + shift[u]) (P operand rewritten in
 *                                LDS the same way), dw (CV, CU, 3, 3).
 * Same arithmetic as apply-then-consume: results are bit-identical to hipseg_bn_relu_apply followed by
 * hipseg_conv_igemm / hipseg_conv_wgrad.  *_applies: 1 when the shape has a kernel with the load-side transform (bf16;
 * conv: 32 or 64 input channels, 32 / 64 output channels, >= 4 x CUs tiles of 8 x 16 pixels; weight gradient: channel
 * counts multiples of 64, H and W multiples of 16).  hipseg_convblock_forward / _backward take this path on their own
 * when both apply to the block's second convolution and the paired weight gradient does not (then `a1` is not written).
 * replaces: aten::native_batch_norm + aten::relu_ materialising the intermediate of
 *           models/processing_blocks.py:44-46 (and its re-read by cuDNN's backward-filter). */
int hipseg_conv3_bnrelu_in_applies(int dtype, int C, int N, int B, int H, int W);
int hipseg_conv3_bnrelu_in(int dtype, const void* in, int C, const float* scale, const float* shift, const void* wp,
                           const float* bias, void* out, int N, float* stats, int B, int H, int W,
                           hipseg_stream_t stream);
int hipseg_conv_wgrad_bnrelu_p_applies(int dtype, int CU, int CV, int B, int H, int W);
int hipseg_conv_wgrad_bnrelu_p(int dtype, const void* p, int CU, const float* scale, const float* shift, const void* q,
                               int CV, float* dw, float* slabs, int B, int H, int W, hipseg_stream_t stream);

/* ---- data gradient + BatchNorm-backward sums in one kernel ----------------------------------------
 * hipseg_conv3_dgrad_bnstats: out = the data gradient hipseg_conv_igemm(HIPSEG_CONV3, in0 = dy, wp = data-gradient
 * operand) computes, for a convolution whose INPUT was relu(bn(x)); in the same kernel the BatchNorm-backward sums
 * of `out`, [sum g | sum g * xhat] with g = out where x * scale + shift > 0 and xhat = (x - mean) * invstd, are reduced
 * per workgroup tile into `partial` (rows x [2][N] floats) -- the rows hipseg_colsum_finalize(partial, rows, 2, N, ...)
 * sums, and what hipseg_bn_bwd_reduce(dy = out, x, ...) would compute from a second pass over both tensors.
 * bn = [mean | invstd | scale | shift], N floats each.  hipseg_conv3_dgrad_bnstats_rows: rows written for the shape,
 * 0 = no kernel with that epilogue takes it (use hipseg_conv_igemm + hipseg_bn_bwd_reduce).
 * replaces: aten::convolution_backward (input gradient) of the second Conv2d of a ConvBlock followed by the
 *           reduction half of aten::native_batch_norm_backward + threshold_backward of the first
 *           BatchNorm2d/ReLU (models/processing_blocks.py:43-45 under autograd). */
int hipseg_conv3_dgrad_bnstats_rows(int dtype, int C, int N, int B, int H, int W);
int hipseg_conv3_dgrad_bnstats(int dtype, const void* dy, int C, const void* wp, void* out, int N, const void* x,
                               const float* bn, float* partial, int B, int H, int W, hipseg_stream_t stream);

/* ---- weight gradient (MFMA, split over pixel chunks) --------------------------------
 * G[tap][u][v] = sum_pixels P[n, tap(y,x), u] * Q[n, y, x, v]
 *   mode HIPSEG_CONV3 : P = layer input (p0|p1 dual source, CU = Cin), Q = dY (CV = Cout),
 *                       result written as Conv2d weight grad (Cout,Cin,3,3).
 *   mode HIPSEG_CONVT : P = dY of the transposed conv (2H x 2W, CU = Cout), Q = its input
 *                       (H x W, CV = Cin); result written as (Cin,Cout,2,2).
 *   mode HIPSEG_CONV1 : 1x1 conv weight grad (Cout,Cin,1,1), P = input, Q = dY.
 *   (H, W) is Q's pixel grid.  `slabs` is a float workspace of
 *   hipseg_wgrad_workspace_elems(...) elements; dw (fp32, the parameter's native layout) is
 *   OVERWRITTEN with the reduced result.
 * replaces: cuDNN conv backward-filter (autograd of processing_blocks.py:43,46,102). */
size_t hipseg_wgrad_workspace_elems(int mode, int CU, int CV, int B, int H, int W);
int hipseg_conv_wgrad(int dtype, int mode, const void* p0, int CU0, const void* p1, int CU1,
                      const void* q, int CV, float* dw, float* slabs, int B, int H, int W,
                      hipseg_stream_t stream);

/* ConvTranspose2d(kernel 2, stride 2) backward w.r.t. weight AND bias in one MFMA kernel + one reduction launch:
 *   dw[ci][co][a][b] = sum_{n,i,j} x[n,i,j,ci] * dy[n,2i+a,2j+b,co]   (fp32, (Cin, Cout, 2, 2), OVERWRITTEN)
 *   db[co]           = sum_{n,y,x} dy[n,y,x,co]                       (fp32, (Cout), OVERWRITTEN)
 * x: NHWC (B,H,W,Cin), dy: NHWC (B,2H,2W,Cout).  The bias gradient is the column sum of the dY fragments the weight
 * gradient's MFMAs already hold -- dY is read once.  `work`: hipseg_convT_wgrad_workspace_elems(...) floats.  Shapes the
 * fused kernel does not take (fp32; Cin % 8 or Cout % 8 != 0) run hipseg_conv_wgrad(HIPSEG_CONVT) followed by
 * hipseg_colsum inside the same call.
 * replaces: aten::convolution_backward (weight and bias gradients) of nn.ConvTranspose2d under autograd
 *           (models/processing_blocks.py:102,106,128,132). */
size_t hipseg_convT_wgrad_workspace_elems(int Cin, int Cout, int B, int H, int W);
int hipseg_convT_wgrad_bias(int dtype, const void* dy, const void* x, float* dw, float* db, float* work, int B, int H,
                            int W, int Cin, int Cout, hipseg_stream_t stream);

/* The 3x3 weight gradients of the TWO convolutions of a ConvBlock (same pixel grid, same Cout = CV) in one launch +
 * one reduction launch: problem a = (pa0|pa1 dual source with CUa0 + CUa1 channels, qa = dY of the first conv) -> dwa
 * (CV, CUa0 + CUa1, 3, 3); problem b = (pb with CUb channels, qb = dY of the second conv) -> dwb (CV, CUb, 3, 3).
 * Every pixel split of the launch then owns the channel tiles of BOTH layers, so there are half as many splits as in
 * two hipseg_conv_wgrad launches and half the fp32 partial-sum traffic (written by the MFMA kernel, read back by the
 * reduction).  Same arithmetic and summation order per element as hipseg_conv_wgrad with that split count.
 * hipseg_conv_wgrad_pair_applies: 1 when the pair is taken for the shape (bf16; >= 64 channels, multiples of 64, on all
 * sides; H, W multiples of 16; the paired grid fills >= 90 % of the CUs), else 0 -> use hipseg_conv_wgrad twice.
 * `slabs`: max(hipseg_wgrad_workspace_elems(a), hipseg_wgrad_workspace_elems(b)) floats suffice.
 * replaces: the weight-gradient halves of aten::convolution_backward of both Conv2d of a ConvBlock
 *           (models/processing_blocks.py:43,46 under autograd). */
int hipseg_conv_wgrad_pair_applies(int dtype, int CUa0, int CUa1, int CUb, int CV, int B, int H, int W);
int hipseg_conv_wgrad_pair(int dtype, const void* pa0, int CUa0, const void* pa1, int CUa1, const void* qa, float* dwa,
                           const void* pb, int CUb, const void* qb, float* dwb, int CV, float* slabs, int B, int H,
                           int W, hipseg_stream_t stream);

/* ---- BatchNorm2d (train: batch statistics; eval: running statistics) ---------------
 * bn_finalize: reduce the conv epilogue partials -> mean, invstd, scale = gamma*invstd,
 *   shift = beta - mean*scale; running_mean/var updated with momentum (unbiased var) and
 *   num_batches_tracked += 1 when the pointers are non-NULL.  count = B*H*W.
 * bn_eval_params: scale/shift/mean/invstd from running statistics.
 * (nn.BatchNorm2d, models/processing_blocks.py:44,47; eps 1e-5, momentum 0.1) */
int hipseg_bn_finalize(float* stats, int mtiles, int C, double count, const float* gamma,
                       const float* beta, float eps, float momentum, float* running_mean,
                       float* running_var, int64_t* num_batches_tracked, float* mean,
                       float* invstd, float* scale, float* shift, hipseg_stream_t stream);
int hipseg_bn_eval_params(const float* gamma, const float* beta, const float* running_mean,
                          const float* running_var, float eps, int C, float* mean, float* invstd,
                          float* scale, float* shift, hipseg_stream_t stream);
/* scale = gamma / sqrt(running_var + eps), shift = beta - running_mean * scale + conv_bias * scale (conv_bias may be
 * NULL): the two vectors hipseg_conv_affine_relu takes, one launch per layer. */
int hipseg_bn_fold(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                   const float* conv_bias, float eps, int C, float* scale, float* shift, hipseg_stream_t stream);
/* y = relu(x*scale + shift); pool != 0 additionally applies MaxPool2d(2,2) and writes the
 * pooled (H/2 x W/2) tensor only.  (processing_blocks.py:44-45,47-48,73) */
int hipseg_bn_relu_apply(int dtype, const void* x, const float* scale, const float* shift, void* y,
                         int B, int H, int W, int C, int pool, hipseg_stream_t stream);
/* Backward of y = [maxpool](relu(bn(x))).  dy is at the pooled resolution when pool != 0.
 * Step 1 (reduce): partial[blk][2][C] <- sum g, sum g*xhat   (g = dy routed through pool/relu)
 * Step 2 (finalize, hipseg_colsum_finalize with rows = 2): sums[2][C] = dbeta, dgamma
 * Step 3 (apply): dx = scale*(g - dbeta/count - xhat*dgamma/count)   (train)
 *                 dx = scale*g                                        (eval != 0)
 *         and, when dbias != NULL, dbias[c] += sum_pixels dx (atomic; conv bias gradient). */
int hipseg_bn_bwd_blocks(int B, int H, int W, int C, int dtype, int pool);
int hipseg_bn_bwd_reduce(int dtype, const void* dy, const void* x, const float* mean,
                         const float* invstd, const float* scale, const float* shift,
                         float* partial, int B, int H, int W, int C, int pool,
                         hipseg_stream_t stream);
int hipseg_bn_bwd_apply(int dtype, const void* dy, const void* x, const float* mean,
                        const float* invstd, const float* scale, const float* shift,
                        const float* sums, double count, int eval, void* dx, float* dbias, int B,
                        int H, int W, int C, int pool, hipseg_stream_t stream);
/* The same two passes when the gradient arrives as TWO tensors of dy's shape (the normalised tensor has two consumers:
 * an encoder block's pooled output feeds the next block and a decoder's skip input, models/UNet.py:64-72): every element
 * is read as round(dy + dy2) in the activation dtype -- the value autograd's own accumulation pass would have written
 * -- and that pass (read 2, write 1 tensor) does not run.  dy2 == NULL: identical to the one-tensor entry points. */
int hipseg_bn_bwd_reduce2(int dtype, const void* dy, const void* dy2, const void* x, const float* mean,
                          const float* invstd, const float* scale, const float* shift, float* partial,
                          int B, int H, int W, int C, int pool, hipseg_stream_t stream);
int hipseg_bn_bwd_apply2(int dtype, const void* dy, const void* dy2, const void* x, const float* mean,
                         const float* invstd, const float* scale, const float* shift, const float* sums,
                         double count, int eval, void* dx, float* dbias, int B, int H, int W, int C, int pool,
                         hipseg_stream_t stream);

/* out[r][c] = sum_blk partial[blk][r][c]  (rows = 1 or 2), fixed order (deterministic).  zero_out != NULL: the same
 * launch also writes zero_out[0..C) = 0 (the conv-bias gradient in front of a train-mode BatchNorm is exactly 0).
 * bn_finalize clobbers its `stats` workspace. */
int hipseg_colsum_finalize(float* partial, int nblk, int rows, int C, float* out, float* zero_out,
                           hipseg_stream_t stream);
/* per-channel sum over pixels of an NHWC tensor: out[c] = sum_p x[p][c] (bias gradients). */
int hipseg_colsum_blocks(long npix, int C, int dtype);
int hipseg_colsum(int dtype, const void* x, long npix, int C, float* partial, float* out,
                  hipseg_stream_t stream);

/* ---- block-level entry points (csrc/block.hip) -----------------------------------------
 * One C call = the whole launch sequence of ConvBlock.forward / its backward
 * (models/processing_blocks.py:40-52; :69-77 with pool = 1; :108-109 with x1 = the skip tensor), for callers that
 * launch eagerly (the reference's TrainingWrapper.train, models/model_wrappers.py:162-180).  The struct holds raw device
 * pointers only; every buffer is allocated by the caller:
 *   stats   : float[hipseg_conv_mtiles(B,H,W) * 2 * Cout]            (forward, train mode)
 *   bn1/bn2 : float[4 * Cout] each = mean | invstd | scale | shift   (written by forward, read by backward)
 *   partial : float[hipseg_bn_bwd_blocks(B,H,W,Cout,dtype,pool=0|1) * 2 * Cout] (max of both)
 *   slabs   : float[max hipseg_wgrad_workspace_elems(...)] of the two weight gradients
 *   colpart : float[hipseg_colsum_blocks(B*H*W, Cout, dtype) * Cout] (eval-mode backward only)
 *   sums1/2 : float[2 * Cout] = dbeta | dgamma of the two BatchNorm layers
 * forward : x0 (|x1) -> raw1 -> a1 -> raw2 -> out (H/2 x W/2 when pool).  wp1/wp2: packed forward operands.
 * backward: dout -> draw2 -> dw2, da1 -> draw1 -> dw1 [, dx0 | dx1 when need_dx].  wp1t/wp2t: packed data-gradient
 *           operands (hipseg_pack_conv_weight(..., transpose = 1)).  draw1 may be the SAME buffer as draw2 (d(raw2) is
 *           dead once da1 and dw2 exist); with two distinct buffers the backward runs both weight gradients as one
 *           paired launch where hipseg_conv_wgrad_pair_applies() says so.  Where hipseg_conv3_dgrad_bnstats_rows()
 *           is non-zero (and fits `partial`) the first layer's BatchNorm-backward sums come out of the second layer's
 *           data-gradient kernel. */
typedef struct hipseg_convblock {
    int32_t dtype, B, H, W, C0, C1, Cout, train, pool, need_dx;
    float eps, momentum;
    const void *x0, *x1, *wp1, *wp2, *wp1t, *wp2t;
    const float *b1, *g1, *be1, *b2, *g2, *be2;
    float *rm1, *rv1, *rm2, *rv2;
    int64_t *nbt1, *nbt2;
    void *raw1, *a1, *raw2, *out;
    float *bn1, *bn2, *stats;
    const void *dout, *dout2; /* dout2: optional second gradient of `out` (two consumers), see hipseg_bn_bwd_apply2 */
    void *draw2, *da1, *draw1, *dx0, *dx1;
    float *dw1, *dw2, *db1, *db2, *sums1, *sums2, *partial, *slabs, *colpart;
    /* > 0: `partial` already holds that many [2][Cout] rows of the second layer's BatchNorm-backward reduction, left
     * there by the kernel that produced dout (hipseg_head_bwd_bnrelu); backward then skips that reduce launch.  Goes
     * with forward's out == NULL (train mode, no pool): the second layer's BatchNorm + ReLU is applied by the consumer
     * when it loads raw2 (hipseg_head_fwd_bnrelu), `out` is never written. */
    int32_t dout_rows;
} hipseg_convblock_t;
size_t hipseg_convblock_size(void);
int hipseg_convblock_forward(const hipseg_convblock_t* args, hipseg_stream_t stream);
int hipseg_convblock_backward(const hipseg_convblock_t* args, hipseg_stream_t stream);

/* ---- 1x1 stem / head ----------------------------------------------------------------
 * stem: Conv2d(Cin, Cout, 1) on an NCHW fp32 image -> NHWC activations (models/UNet.py:39,62).
 * stem_bwd: dW (Cout,Cin), db (Cout) from dY (NHWC) and the NCHW image (no data gradient:
 *           the image does not require grad).  dw/db are overwritten. */
int hipseg_stem_fwd(int dtype, const float* x_nchw, const float* w, const float* b, void* y, int B,
                    int Cin, int H, int W, int Cout, hipseg_stream_t stream);
int hipseg_stem_bwd_blocks(int B, int H, int W);
int hipseg_stem_bwd(int dtype, const float* x_nchw, const void* dy, float* partial, float* dw,
                    float* db, int B, int Cin, int H, int W, int Cout, hipseg_stream_t stream);
/* the same with dY = dy + dy2 (dy2 may be NULL): the stem output feeds both the first encoder block and, as the skip
 * tensor, the last decoder block (models/UNet.py:62,72); autograd would first sum the two gradients in a pass of its
 * own (3 x the tensor in traffic), here the second one is read beside the first. */
int hipseg_stem_bwd2(int dtype, const float* x_nchw, const void* dy, const void* dy2, float* partial, float* dw,
                     float* db, int B, int Cin, int H, int W, int Cout, hipseg_stream_t stream);
/* head: Conv2d(Cin, Cout<=8, 1) NHWC activations -> NCHW fp32 logits (models/UNet.py:55,73).
 * head_bwd: dX (NHWC, dtype), dW (Cout,Cin), db (Cout) from NCHW fp32 dlogits. */
int hipseg_head_fwd(int dtype, const void* x, const float* w, const float* b, float* logits_nchw,
                    int B, int H, int W, int Cin, int Cout, hipseg_stream_t stream);
int hipseg_head_bwd_blocks(int B, int H, int W);
int hipseg_head_bwd(int dtype, const void* x, const float* dlogits_nchw, const float* w, void* dx,
                    float* partial, float* dw, float* db, int B, int H, int W, int Cin, int Cout,
                    hipseg_stream_t stream);

/* head over the LAST ConvBlock's pre-normalisation output (models/UNet.py:72-73: dec4 -> out): that block's final
 * BatchNorm + ReLU (processing_blocks.py:33-34) is applied in the head's load path -- x := round_dtype(relu(raw * scale +
 * shift)), the value bn_relu_apply would have stored, so logits / dW / db / dX are bit-identical to
 * hipseg_bn_relu_apply + hipseg_head_fwd / hipseg_head_bwd -- and the activated tensor never exists.
 * head_bwd_bnrelu also leaves the BatchNorm-backward partial sums of that layer, bn_partial[hipseg_head_bwd_blocks()][2][Cin]
 * = [sum g | sum g * xhat] with g = dX where raw * scale + shift > 0 (hipseg_bn_bwd_reduce's rows; hand them to
 * hipseg_colsum_finalize / hipseg_convblock_t::dout_rows): the reduce pass over dX and raw is not needed. */
int hipseg_head_fwd_bnrelu(int dtype, const void* raw, const float* scale, const float* shift, const float* w,
                           const float* b, float* logits_nchw, int B, int H, int W, int Cin, int Cout,
                           hipseg_stream_t stream);
int hipseg_head_bwd_bnrelu(int dtype, const void* raw, const float* mean, const float* invstd, const float* scale,
                           const float* shift, const float* dlogits_nchw, const float* w, void* dx, float* partial,
                           float* dw, float* db, float* bn_partial, int B, int H, int W, int Cin, int Cout,
                           hipseg_stream_t stream);

/* ---- bilinear resize, align_corners=True (processing_blocks.py:107), NHWC ---------- */
int hipseg_bilinear_fwd(int dtype, const void* x, void* y, int B, int Hi, int Wi, int Ho, int Wo,
                        int C, hipseg_stream_t stream);
int hipseg_bilinear_bwd(int dtype, const void* dy, void* dx, int B, int Hi, int Wi, int Ho, int Wo,
                        int C, hipseg_stream_t stream);

/* ---- losses (models/losses.py) ----------------------------------------------------------
 * ce: nn.CrossEntropyLoss() on NCHW fp32 logits (B,C,H,W), int64 targets (B,H,W)
 *     (HybridLoss.forward, losses.py:13-15).  loss[0] = mean NLL over the non-ignored
 *     (target != -100) pixels, loss[1] = their count.
 * ce_bwd: dlogits = (softmax - onehot) * gscale[0] / loss[1]; gscale is a DEVICE scalar
 *     (the upstream gradient, e.g. GradScaler's scale) so no host sync is needed; `loss`
 *     is the 2-float buffer ce_fwd wrote. */
int hipseg_loss_blocks(long n);
int hipseg_ce_fwd(const float* logits, const int64_t* target, float* partial, float* loss, int B,
                  int C, long HW, hipseg_stream_t stream);
int hipseg_ce_bwd(const float* logits, const int64_t* target, const float* gscale, const float* loss,
                  float* dlogits, int B, int C, long HW, hipseg_stream_t stream);
/* bce_dice: HybridLossBinary.forward (losses.py:24-36): BCEWithLogits(mean) + smp DiceLoss
 *     (mode binary, from_logits=True applied to sigmoid(pred), smooth 0, eps 1e-7).
 *     sums[4] = {sum bce, sum p*t, sum p, sum t} with p = sigmoid(sigmoid(x)); loss[0] = total. */
int hipseg_bce_dice_fwd(const float* logits, const float* target, float* partial, float* sums,
                        float* loss, long n, hipseg_stream_t stream);
int hipseg_bce_dice_bwd(const float* logits, const float* target, const float* sums,
                        const float* gscale, float* dlogits, long n, hipseg_stream_t stream);
/* segmentation metrics (IoU / PixelAccuracy, losses.py:38-63,129-154): CxC confusion
 * matrix of argmax(logits) vs target, conf[t*C + p] (int64 counts; conf is overwritten). */
int hipseg_confusion(const float* logits, const int64_t* target, long long* conf, int B, int C,
                     long HW, hipseg_stream_t stream);

/* ---- layout helpers ---------------------------------------------------------------------- */
/* NCHW fp32 <-> NHWC dtype (for standalone block use with contiguous NCHW callers). */
int hipseg_nchw_to_nhwc(int dtype, const float* x, void* y, int B, int C, int H, int W,
                        hipseg_stream_t stream);
int hipseg_nhwc_to_nchw(int dtype, const void* x, float* y, int B, int C, int H, int W,
                        hipseg_stream_t stream);

/* ---- dataset-record decode (the data format in front of the path) -----------------------------
 * images: n records of H x W x 3 uint8 (HWC), masks: n records of H x W uint8 (38 = cat, 75 = dog, 255 = border).
 * out_images: float32 (n,3,H,W) = byte / 255.0; out_masks: int64 (n,H,W) =
 *   record has a cat pixel ? (m == 38) + (m == 255) : 2*(m == 75) + 2*(m == 255).
 * cat_flags: int[n] workspace (per-record "any cat pixel").  H*W must be a multiple of 4.
 * replaces: CustomImageDataset._deserialize_datapoint / _deserialize_numpy, customDatasets/datasets.py:92-135
 * (the reference decodes one record at a time with numpy on the host; its records are 256 x 256). */
int hipseg_decode_records(const uint8_t* images, const uint8_t* masks, float* out_images, int64_t* out_masks,
                          int* cat_flags, int n, int H, int W, hipseg_stream_t stream);

/* found[0] = 1.0f when any gradient element of the descriptors' tensors is inf or NaN (the caller zeroes found[0] first;
 * untouched otherwise).  The inf check GradScaler.step needs, without torch's unscale pass over the gradients (which
 * rewrites every element x 1.0 when the optimizer takes the scale itself, as hipseg_adam_step does).
 * replaces: aten::_amp_foreach_non_finite_check_and_unscale_ as called by GradScaler._check_inf_per_device
 *           (models/model_wrappers.py:176 `scaler.step(optimizer)`). */
int hipseg_grads_nonfinite(const void* host_descs, int ntensors, float* found, hipseg_stream_t stream);

/* ---- optimiser step ----------------------------------------------------------------------------------
 * replaces: torch.optim.Adam.step as driven by GradScaler.step (models/model_wrappers.py:124,176,979): the whole
 * parameter group in ceil(ntensors / 88) launches + a one-thread launch that advances the step counter.  host_descs: HOST table of hipseg_adam_desc_size()-byte entries
 * {param, grad, exp_avg, exp_avg_sq (device fp32 pointers), numel} filled by hipseg_adam_desc_fill; it is read during
 * the call only (descriptors travel by value in the kernel arguments).
 * state: device int[2] {step count, reserved}.  found_inf / grad_scale: device floats or NULL
 * (GradScaler contract: found_inf != 0 skips the step entirely; gradients are divided by grad_scale). */
size_t hipseg_adam_desc_size(void);
int hipseg_adam_desc_fill(void* host_descs, int index, float* p, const float* g, float* m, float* v, long n);
int hipseg_adam_step(const void* host_descs, int ntensors, int* state, const float* found_inf,
                     const float* grad_scale, float lr, float beta1, float beta2, float eps,
                     float weight_decay, hipseg_stream_t stream);

/* ---- on-device training augmentation (the per-step caller in front of the path) -------------------
 * replaces: DataAugmentor.forward / DataAugmentorPrompt.forward, models/processing_blocks.py:344-384,386-451 (kornia
 * RandomHorizontalFlip + RandomRotation(90, nearest) on image||mask[||prompt], ColorJitter + RandomGaussianBlur(5x5) on
 * the image; called every step at models/model_wrappers.py:165,968).  kornia 0.8.0's arithmetic: PARITY UNPINNED.
 * images (B,3,H,W) fp32 NCHW in [0,1]; masks (B,H,W) int64 or NULL; extra (B,n_extra,H,W) fp32 geometric-only
 * channels or NULL.  params: B rows of HIPSEG_AUG_NPARAM floats:
 *   [0] keep (!=0: sample copied through untouched)  [1] flip (!=0)  [2] cos, [3] sin of the rotation angle
 *   [4] brightness factor  [5] contrast factor  [6] saturation factor  [7] hue shift (radians)  [8] blur sigma
 * order: int[4], a permutation of {0 brightness, 1 contrast, 2 saturation, 3 hue} (one per call).
 * partial: fp32 workspace of hipseg_augment_workspace_elems(B).  Outputs must not alias inputs. */
#define HIPSEG_AUG_NPARAM 16
size_t hipseg_augment_workspace_elems(int B);
/* params table from `uniforms` (B x 8 floats in [0,1)): [0] flip draw, [1] rotate draw, [2] angle, [3..6] brightness /
 * contrast / saturation / hue, [7] sigma -- kornia's documented defaults are the caller's arguments
 * (RandomHorizontalFlip p, RandomRotation p and degrees, ColorJitter ranges, RandomGaussianBlur sigma range);
 * sample b is kept untouched when b % keep_stride == 0 (keep_stride = augmentations_per_datapoint + 1). */
int hipseg_augment_params(const float* uniforms, float* params, int B, int keep_stride, float flip_p,
                          float rotate_p, float degrees, float brightness, float contrast,
                          float saturation, float hue, float sigma_lo, float sigma_hi,
                          hipseg_stream_t stream);
int hipseg_augment(const float* images, const int64_t* masks, const float* extra, int n_extra,
                   const float* params, const int* order, float* partial, float* out_images,
                   int64_t* out_masks, float* out_extra, int B, int H, int W, hipseg_stream_t stream);

/* ---- stream / event plumbing of the data-parallel step --------------------------------------
 * External event-record nodes inside a captured hipGraph: eager work on another stream (the RCCL bucket all-reduces of
 * hipseg/ddp.py) starts when the replayed graph passes the point where the event was recorded during capture.
 * hipseg_event_record_external: hipEventRecordWithFlags(hipEventRecordExternal) -- a graph node while `stream` is
 *   capturing, a plain record otherwise.  hipseg_stream_wait_event: hipStreamWaitEvent.
 * replaces: the autograd-hook -> NCCL stream hand-over inside torch's DistributedDataParallel reducer
 *           (scripts/train_distributed.py:35), for a step that is replayed as a hipGraph. */
int hipseg_event_create(void** event);
int hipseg_event_destroy(void* event);
int hipseg_event_record_external(void* event, hipseg_stream_t stream);
int hipseg_stream_wait_event(hipseg_stream_t stream, void* event);

/* Average one flat fp32 gradient bucket over the ranks of an RCCL communicator, in place, on `stream`
 * (ncclAllReduce(ncclFloat32, ncclAvg)): SURVEY section 8b's `bucket_allreduce(ptr, count, dtype, comm, stream)` for hosts
 * that own the communicator (`comm` = ncclComm_t).  RCCL is bound at the first call (the librccl already in the process,
 * else the system's); the Python host of this repo reduces through torch.distributed instead (hipseg/ddp.py: its
 * process group owns the communicator and does not hand it out).
 * replaces: the per-bucket NCCL all-reduce of torch DDP (scripts/train_distributed.py:35, models/model_wrappers.py:978). */
int hipseg_bucket_allreduce(void* bucket, size_t count, int dtype, void* comm, hipseg_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* HIPSEG_H */
