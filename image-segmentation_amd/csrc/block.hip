// Block-level entry points: the whole launch sequence of one ConvBlock forward / backward behind ONE C call.
// Replaces ConvBlock.forward / its autograd backward (/root/reference/models/processing_blocks.py:40-52 and :69-77,
// :108-109 for the pooled / dual-source forms) for callers that launch eagerly: the reference's unchanged
// TrainingWrapper.train (models/model_wrappers.py:162-180) issues every step from Python, and 13 ctypes calls + their
// argument marshalling per block cost more host time than the GPU needs for the kernels of the small layers.
// Nothing is computed here: the functions chain the per-op entry points of this library on the given stream, with the
// workspaces the caller allocated.
#include "common.h"

static inline const float* bnv(const float* bn, int C, int i) { return bn + (size_t)i * C; }

extern "C" size_t hipseg_convblock_size(void) { return sizeof(hipseg_convblock_t); }

// conv3x3 (+bias, + batch statistics) -> BatchNorm parameters of this forward -> BN-apply + ReLU (+ 2x2 max-pool)
static int conv_bn_relu(const hipseg_convblock_t* a, const void* in0, int c0, const void* in1, int c1, const void* wp,
                        const float* b, const float* g, const float* be, float* rm, float* rv, int64_t* nbt, void* raw,
                        void* act, float* bn, int pool, hipseg_stream_t s) {
    const int C = a->Cout, B = a->B, H = a->H, W = a->W, dt = a->dtype;
    float *mean = bn, *invstd = bn + C, *scale = bn + 2 * (size_t)C, *shift = bn + 3 * (size_t)C;
    if (a->train) {
        if (int rc = hipseg_conv_igemm(dt, HIPSEG_CONV3, in0, c0, in1, c1, wp, b, raw, C, nullptr, 0, a->stats, B, H, W, s))
            return rc;
        const int rows = hipseg_conv_stats_rows(dt, HIPSEG_CONV3, c0, c1, C, 0, B, H, W);
        if (int rc = hipseg_bn_finalize(a->stats, rows, C, (double)B * H * W, g, be, a->eps, a->momentum, rm, rv, nbt, mean,
                                        invstd, scale, shift, s))
            return rc;
    } else {
        if (int rc = hipseg_conv_igemm(dt, HIPSEG_CONV3, in0, c0, in1, c1, wp, b, raw, C, nullptr, 0, nullptr, B, H, W, s))
            return rc;
        if (int rc = hipseg_bn_eval_params(g, be, rm, rv, a->eps, C, mean, invstd, scale, shift, s)) return rc;
    }
    if (!act) return HIPSEG_OK;  // (applied by the consumer on load, see hipseg_convblock_forward)
    return hipseg_bn_relu_apply(dt, raw, scale, shift, act, B, H, W, C, pool, s);
}

// BatchNorm + ReLU of the first layer applied in the second convolution's load path (and in its weight gradient's):
// train mode, both kernels with the load-side transform take the shape, and the paired weight gradient (which reads the
// activated tensor for its second problem) does not.  Decided from shapes only: forward and backward agree.
static bool bn_on_load(const hipseg_convblock_t* a) {
    const int C = a->Cout;
    return a->train && hipseg_conv3_bnrelu_in_applies(a->dtype, C, C, a->B, a->H, a->W) &&
           hipseg_conv_wgrad_bnrelu_p_applies(a->dtype, C, C, a->B, a->H, a->W) &&
           !hipseg_conv_wgrad_pair_applies(a->dtype, a->C0, a->C1, C, C, a->B, a->H, a->W);
}

extern "C" int hipseg_convblock_forward(const hipseg_convblock_t* a, hipseg_stream_t s) {
    HS_REQUIRE(a && a->x0 && a->wp1 && a->wp2 && a->raw1 && a->a1 && a->raw2 && a->bn1 && a->bn2,
               "convblock_forward: null operand");
    HS_REQUIRE(!a->train || a->stats, "convblock_forward: train mode needs the statistics workspace");
    // out == NULL: the consumer applies the second layer's BatchNorm + ReLU itself when it loads raw2 (bn2 holds scale /
    // shift): hipseg_head_fwd_bnrelu.  Train mode without a pool only (what that consumer implements).
    HS_REQUIRE(a->out || (a->train && !a->pool), "convblock_forward: out may be NULL only in train mode without a pool");
    if (bn_on_load(a)) {
        // conv -> statistics -> (scale, shift); the activated intermediate a1 is NOT written: the second convolution
        // transforms raw1 on load
        const int C = a->Cout, B = a->B, H = a->H, W = a->W, dt = a->dtype;
        float *bn1 = a->bn1, *bn2 = a->bn2;
        if (int rc = hipseg_conv_igemm(dt, HIPSEG_CONV3, a->x0, a->C0, a->x1, a->C1, a->wp1, a->b1, a->raw1, C, nullptr, 0,
                                       a->stats, B, H, W, s))
            return rc;
        if (int rc = hipseg_bn_finalize(a->stats, hipseg_conv_stats_rows(dt, HIPSEG_CONV3, a->C0, a->C1, C, 0, B, H, W), C,
                                        (double)B * H * W, a->g1, a->be1, a->eps, a->momentum, a->rm1, a->rv1, a->nbt1, bn1,
                                        bn1 + C, bn1 + 2 * (size_t)C, bn1 + 3 * (size_t)C, s))
            return rc;
        if (int rc = hipseg_conv3_bnrelu_in(dt, a->raw1, C, bn1 + 2 * (size_t)C, bn1 + 3 * (size_t)C, a->wp2, a->b2, a->raw2, C,
                                            a->stats, B, H, W, s))
            return rc;
        if (int rc = hipseg_bn_finalize(a->stats, hipseg_conv_stats_rows(dt, HIPSEG_CONV3, C, 0, C, 0, B, H, W), C,
                                        (double)B * H * W, a->g2, a->be2, a->eps, a->momentum, a->rm2, a->rv2, a->nbt2, bn2,
                                        bn2 + C, bn2 + 2 * (size_t)C, bn2 + 3 * (size_t)C, s))
            return rc;
        if (!a->out) return HIPSEG_OK;
        return hipseg_bn_relu_apply(dt, a->raw2, bn2 + 2 * (size_t)C, bn2 + 3 * (size_t)C, a->out, B, H, W, C, a->pool, s);
    }
    if (int rc = conv_bn_relu(a, a->x0, a->C0, a->x1, a->C1, a->wp1, a->b1, a->g1, a->be1, a->rm1, a->rv1, a->nbt1, a->raw1,
                              a->a1, a->bn1, 0, s))
        return rc;
    return conv_bn_relu(a, a->a1, a->Cout, nullptr, 0, a->wp2, a->b2, a->g2, a->be2, a->rm2, a->rv2, a->nbt2, a->raw2, a->out,
                        a->bn2, a->pool, s);
}

// backward through [pool](relu(bn(raw))): [dbeta | dgamma] -> sums, conv-bias gradient -> dbias, d(raw) -> draw.
// reduced_rows > 0: a->partial already holds that many [2][C] rows of the reduction (written by the data-gradient
// kernel that produced dy), the reduce launch is skipped.
static int bn_relu_bwd(const hipseg_convblock_t* a, const void* dy, const void* dy2, const void* raw, const float* bn, int pool,
                       float* sums, float* dbias, void* draw, hipseg_stream_t s, int reduced_rows = 0) {
    const int C = a->Cout, B = a->B, H = a->H, W = a->W, dt = a->dtype;
    const int nblk = reduced_rows ? reduced_rows : hipseg_bn_bwd_blocks(B, H, W, C, dt, pool);
    if (!reduced_rows)
        if (int rc = hipseg_bn_bwd_reduce2(dt, dy, dy2, raw, bnv(bn, C, 0), bnv(bn, C, 1), bnv(bn, C, 2), bnv(bn, C, 3),
                                           a->partial, B, H, W, C, pool, s))
            return rc;
    if (int rc = hipseg_colsum_finalize(a->partial, nblk, 2, C, sums, a->train ? dbias : nullptr, s)) return rc;
    if (int rc = hipseg_bn_bwd_apply2(dt, dy, dy2, raw, bnv(bn, C, 0), bnv(bn, C, 1), bnv(bn, C, 2), bnv(bn, C, 3), sums,
                                      (double)B * H * W, a->train ? 0 : 1, draw, nullptr, B, H, W, C, pool, s))
        return rc;
    if (!a->train) return hipseg_colsum(dt, draw, (long)B * H * W, C, a->colpart, dbias, s);
    return HIPSEG_OK;
}

extern "C" int hipseg_convblock_backward(const hipseg_convblock_t* a, hipseg_stream_t s) {
    HS_REQUIRE(a && a->dout && a->x0 && a->raw1 && a->a1 && a->raw2 && a->bn1 && a->bn2 && a->draw2 && a->da1 && a->draw1 &&
                   a->dw1 && a->dw2 && a->db1 && a->db2 && a->sums1 && a->sums2 && a->partial && a->slabs && a->wp2t,
               "convblock_backward: null operand");
    HS_REQUIRE(a->train || a->colpart, "convblock_backward: eval mode needs the column-sum workspace");
    HS_REQUIRE(!a->need_dx || (a->wp1t && a->dx0 && ((a->C1 == 0) == (a->dx1 == nullptr))), "convblock_backward: dx operands");
    const int C = a->Cout, B = a->B, H = a->H, W = a->W, dt = a->dtype;
    // both weight gradients in one launch where the pair is taken (half the partial-sum traffic): needs d(raw2) alive
    // until d(raw1) exists, i.e. two distinct buffers
    const bool pair = a->draw1 != a->draw2 && hipseg_conv_wgrad_pair_applies(dt, a->C0, a->C1, C, C, B, H, W);
    // second conv layer
    // (dout_rows > 0: the kernel that produced dout -- hipseg_head_bwd_bnrelu -- already left that many rows of this
    // layer's reduction in a->partial)
    HS_REQUIRE(a->dout_rows >= 0 && (!a->dout_rows || (!a->pool && !a->dout2)), "convblock_backward: dout_rows");
    if (int rc = bn_relu_bwd(a, a->dout, a->dout2, a->raw2, a->bn2, a->pool, a->sums2, a->db2, a->draw2, s, a->dout_rows))
        return rc;
    if (bn_on_load(a)) {  // (a1 was never written: the weight gradient transforms raw1 on load)
        if (int rc = hipseg_conv_wgrad_bnrelu_p(dt, a->raw1, C, a->bn1 + 2 * (size_t)C, a->bn1 + 3 * (size_t)C, a->draw2, C,
                                                a->dw2, a->slabs, B, H, W, s))
            return rc;
    } else if (!pair)
        if (int rc = hipseg_conv_wgrad(dt, HIPSEG_CONV3, a->a1, C, nullptr, 0, a->draw2, C, a->dw2, a->slabs, B, H, W, s))
            return rc;
    // data gradient of the second conv; where a kernel with that epilogue takes the shape it also reduces the
    // BatchNorm-backward sums of the first layer (its output IS that layer's dy), saving a pass over da1 and raw1
    // (only when its rows fit the `partial` workspace as include/hipseg.h sizes it: the reduce kernels' block counts)
    int fused_rows = hipseg_conv3_dgrad_bnstats_rows(dt, C, C, B, H, W);
    {
        const int n0 = hipseg_bn_bwd_blocks(B, H, W, C, dt, 0), n1 = a->pool ? hipseg_bn_bwd_blocks(B, H, W, C, dt, 1) : 0;
        if (fused_rows > (n0 > n1 ? n0 : n1)) fused_rows = 0;
    }
    if (fused_rows) {
        if (int rc = hipseg_conv3_dgrad_bnstats(dt, a->draw2, C, a->wp2t, a->da1, C, a->raw1, a->bn1, a->partial, B, H, W, s))
            return rc;
    } else if (int rc = hipseg_conv_igemm(dt, HIPSEG_CONV3, a->draw2, C, nullptr, 0, a->wp2t, nullptr, a->da1, C, nullptr, 0,
                                          nullptr, B, H, W, s))
        return rc;
    // first conv layer
    if (int rc = bn_relu_bwd(a, a->da1, nullptr, a->raw1, a->bn1, 0, a->sums1, a->db1, a->draw1, s, fused_rows)) return rc;
    if (pair) {
        if (int rc = hipseg_conv_wgrad_pair(dt, a->x0, a->C0, a->x1, a->C1, a->draw1, a->dw1, a->a1, C, a->draw2, a->dw2, C,
                                            a->slabs, B, H, W, s))
            return rc;
    } else if (int rc = hipseg_conv_wgrad(dt, HIPSEG_CONV3, a->x0, a->C0, a->x1, a->C1, a->draw1, C, a->dw1, a->slabs, B, H, W,
                                          s))
        return rc;
    if (a->need_dx)
        return hipseg_conv_igemm(dt, HIPSEG_CONV3, a->draw1, C, nullptr, 0, a->wp1t, nullptr, a->dx0, a->C0, a->dx1, a->C1,
                                 nullptr, B, H, W, s);
    return HIPSEG_OK;
}
