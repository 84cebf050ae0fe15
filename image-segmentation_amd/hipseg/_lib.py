"""ctypes binding of libhipseg.so -- the C ABI declared in include/hipseg.h.

The product path has NO CPU fallback: if the library is missing or fails to load,
importing this module raises, and every op raises RuntimeError on a non-zero return.
"""
import ctypes
import functools
import os
from ctypes import c_char_p, c_double, c_float, c_int, c_long, c_size_t, c_void_p

# PyTorch-ROCm bundles its own libamdhip64; it must be the HIP runtime instance this library binds to (same
# process, same streams, same device context).  Importing torch first makes the loader resolve libhipseg's
# libamdhip64 dependency to torch's already-loaded copy; loading ours first would create a second runtime
# that does not see torch's device context ("no ROCm-capable device is detected" at the first launch).
import torch  # noqa: F401  (must precede ctypes.CDLL below)

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HIPSEG_LIB") or os.path.join(HERE, "lib", "libhipseg.so")  # HIPSEG_LIB: A/B builds

F32, BF16 = 0, 1
CONV3, CONV1, CONV2S2, CONVT = 0, 1, 2, 3
AUG_NPARAM = 16  # HIPSEG_AUG_NPARAM

P = c_void_p
I = c_int
L = c_long

# name -> (restype, argtypes); mirrors include/hipseg.h line by line
PROTOTYPES = {
    "hipseg_last_error": (c_char_p, []),
    "hipseg_abi_version": (I, []),
    "hipseg_kpad": (I, [I, I]),
    "hipseg_npad": (I, [I]),
    "hipseg_conv_mtiles": (I, [I, I, I]),
    "hipseg_conv_stats_rows": (I, [I, I, I, I, I, I, I, I, I]),
    "hipseg_pack_conv_weight": (I, [P, P, I, I, I, I, I, P]),
    "hipseg_pack_conv_weight_both": (I, [P, P, P, I, I, I, I, P]),
    "hipseg_pack_convT_weight": (I, [P, P, I, I, I, I, P]),
    "hipseg_pack_desc_size": (c_size_t, []),
    "hipseg_pack_desc_fill": (I, [P, I, P, P, P, I, I, I, I, I]),
    "hipseg_pack_batch": (I, [P, I, I, L, P]),
    "hipseg_conv_igemm": (I, [I, I, P, I, P, I, P, P, P, I, P, I, P, I, I, I, P]),
    "hipseg_event_create": (I, [ctypes.POINTER(c_void_p)]),
    "hipseg_event_destroy": (I, [P]),
    "hipseg_event_record_external": (I, [P, P]),
    "hipseg_stream_wait_event": (I, [P, P]),
    "hipseg_bucket_allreduce": (I, [P, c_size_t, I, P, P]),
    "hipseg_conv_affine_relu": (I, [I, P, I, P, I, P, P, P, P, I, I, I, I, P]),
    "hipseg_convT_dgrad_bnstats_rows": (I, [I, I, I, I, I, I]),
    "hipseg_convT_dgrad_bnstats": (I, [I, P, I, P, P, I, P, P, P, I, I, I, P]),
    "hipseg_conv3_bnrelu_in_applies": (I, [I, I, I, I, I, I]),
    "hipseg_conv3_bnrelu_in": (I, [I, P, I, P, P, P, P, P, I, P, I, I, I, P]),
    "hipseg_conv_wgrad_bnrelu_p_applies": (I, [I, I, I, I, I, I]),
    "hipseg_conv_wgrad_bnrelu_p": (I, [I, P, I, P, P, P, I, P, P, I, I, I, P]),
    "hipseg_conv3_dgrad_bnstats_rows": (I, [I, I, I, I, I, I]),
    "hipseg_conv3_dgrad_bnstats": (I, [I, P, I, P, P, I, P, P, P, I, I, I, P]),
    "hipseg_wgrad_workspace_elems": (c_size_t, [I, I, I, I, I, I]),
    "hipseg_conv_wgrad": (I, [I, I, P, I, P, I, P, I, P, P, I, I, I, P]),
    "hipseg_convT_wgrad_workspace_elems": (c_size_t, [I, I, I, I, I]),
    "hipseg_convT_wgrad_bias": (I, [I, P, P, P, P, P, I, I, I, I, I, P]),
    "hipseg_conv_wgrad_pair_applies": (I, [I, I, I, I, I, I, I, I]),
    "hipseg_conv_wgrad_pair": (I, [I, P, I, P, I, P, P, P, I, P, P, I, P, I, I, I, P]),
    "hipseg_bn_finalize": (I, [P, I, I, c_double, P, P, c_float, c_float, P, P, P, P, P, P, P, P]),
    "hipseg_bn_eval_params": (I, [P, P, P, P, c_float, I, P, P, P, P, P]),
    "hipseg_bn_fold": (I, [P, P, P, P, P, c_float, I, P, P, P]),
    "hipseg_bn_relu_apply": (I, [I, P, P, P, P, I, I, I, I, I, P]),
    "hipseg_bn_bwd_blocks": (I, [I, I, I, I, I, I]),
    "hipseg_bn_bwd_reduce": (I, [I, P, P, P, P, P, P, P, I, I, I, I, I, P]),
    "hipseg_bn_bwd_apply": (I, [I, P, P, P, P, P, P, P, c_double, I, P, P, I, I, I, I, I, P]),
    "hipseg_bn_bwd_reduce2": (I, [I, P, P, P, P, P, P, P, P, I, I, I, I, I, P]),
    "hipseg_bn_bwd_apply2": (I, [I, P, P, P, P, P, P, P, P, c_double, I, P, P, I, I, I, I, I, P]),
    "hipseg_colsum_finalize": (I, [P, I, I, I, P, P, P]),
    "hipseg_colsum_blocks": (I, [L, I, I]),
    "hipseg_colsum": (I, [I, P, L, I, P, P, P]),
    "hipseg_stem_fwd": (I, [I, P, P, P, P, I, I, I, I, I, P]),
    "hipseg_stem_bwd_blocks": (I, [I, I, I]),
    "hipseg_stem_bwd": (I, [I, P, P, P, P, P, I, I, I, I, I, P]),
    "hipseg_stem_bwd2": (I, [I, P, P, P, P, P, P, I, I, I, I, I, P]),
    "hipseg_head_fwd": (I, [I, P, P, P, P, I, I, I, I, I, P]),
    "hipseg_head_bwd_blocks": (I, [I, I, I]),
    "hipseg_head_fwd_bnrelu": (I, [I, P, P, P, P, P, P, I, I, I, I, I, P]),
    "hipseg_head_bwd_bnrelu": (I, [I, P, P, P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, P]),
    "hipseg_head_bwd": (I, [I, P, P, P, P, P, P, P, I, I, I, I, I, P]),
    "hipseg_bilinear_fwd": (I, [I, P, P, I, I, I, I, I, I, P]),
    "hipseg_bilinear_bwd": (I, [I, P, P, I, I, I, I, I, I, P]),
    "hipseg_loss_blocks": (I, [L]),
    "hipseg_ce_fwd": (I, [P, P, P, P, I, I, L, P]),
    "hipseg_ce_bwd": (I, [P, P, P, P, P, I, I, L, P]),
    "hipseg_bce_dice_fwd": (I, [P, P, P, P, P, L, P]),
    "hipseg_bce_dice_bwd": (I, [P, P, P, P, P, L, P]),
    "hipseg_confusion": (I, [P, P, P, I, I, L, P]),
    "hipseg_nchw_to_nhwc": (I, [I, P, P, I, I, I, I, P]),
    "hipseg_nhwc_to_nchw": (I, [I, P, P, I, I, I, I, P]),
    "hipseg_decode_records": (I, [P, P, P, P, P, I, I, I, P]),
    "hipseg_adam_desc_size": (c_size_t, []),
    "hipseg_adam_desc_fill": (I, [P, I, P, P, P, P, L]),
    "hipseg_grads_nonfinite": (I, [P, I, P, P]),
    "hipseg_adam_step": (I, [P, I, P, P, P, c_float, c_float, c_float, c_float, c_float, P]),
    "hipseg_augment_workspace_elems": (c_size_t, [I]),
    "hipseg_augment_params": (I, [P, P, I, I] + [c_float] * 9 + [P]),
    "hipseg_augment": (I, [P, P, P, I, P, P, P, P, P, P, I, I, I, P]),
    "hipseg_convblock_size": (c_size_t, []),
    "hipseg_convblock_forward": (I, [P, P]),
    "hipseg_convblock_backward": (I, [P, P]),
}


class ConvBlockArgs(ctypes.Structure):
    """hipseg_convblock_t of include/hipseg.h (block-level entry points): raw device pointers + geometry."""
    _fields_ = ([(n, ctypes.c_int32) for n in ("dtype", "B", "H", "W", "C0", "C1", "Cout", "train", "pool", "need_dx")]
                + [("eps", c_float), ("momentum", c_float)]
                + [(n, c_void_p) for n in ("x0", "x1", "wp1", "wp2", "wp1t", "wp2t", "b1", "g1", "be1", "b2", "g2", "be2",
                                            "rm1", "rv1", "rm2", "rv2", "nbt1", "nbt2", "raw1", "a1", "raw2", "out", "bn1",
                                            "bn2", "stats", "dout", "dout2", "draw2", "da1", "draw1", "dx0", "dx1", "dw1", "dw2",
                                            "db1", "db2", "sums1", "sums2", "partial", "slabs", "colpart")]
                + [("dout_rows", ctypes.c_int32)])

# functions whose int return value is a geometry answer, not a status code
_PURE = {"hipseg_abi_version", "hipseg_kpad", "hipseg_npad", "hipseg_conv_mtiles", "hipseg_conv_stats_rows", "hipseg_conv3_dgrad_bnstats_rows", "hipseg_convT_dgrad_bnstats_rows", "hipseg_conv_wgrad_pair_applies", "hipseg_conv3_bnrelu_in_applies",
         "hipseg_conv_wgrad_bnrelu_p_applies", "hipseg_bn_bwd_blocks",
         "hipseg_colsum_blocks", "hipseg_stem_bwd_blocks", "hipseg_head_bwd_blocks", "hipseg_loss_blocks",
         "hipseg_wgrad_workspace_elems", "hipseg_convT_wgrad_workspace_elems", "hipseg_last_error", "hipseg_pack_desc_size", "hipseg_augment_workspace_elems", "hipseg_adam_desc_size",
         "hipseg_convblock_size"}


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python __graft_entry__.py` (hipcc --offload-arch=gfx950). "
            "The HIP path has no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()
if ctypes.sizeof(ConvBlockArgs) != lib.hipseg_convblock_size():
    raise ImportError(f"hipseg_convblock_t layout mismatch: ctypes {ctypes.sizeof(ConvBlockArgs)} B, library "
                      f"{lib.hipseg_convblock_size()} B (rebuild libhipseg.so)")


class HipsegError(RuntimeError):
    pass


def _wrap(name):
    fn = getattr(lib, name)
    if name in _PURE:
        # pure functions of shapes (and of the current device's CU count: one device per process): memoised, a ctypes
        # round trip costs 2-3 us of host time per kernel launch otherwise
        return fn if name == "hipseg_last_error" else functools.lru_cache(maxsize=8192)(fn)

    def call(*a):
        rc = fn(*a)
        if rc != 0:
            raise HipsegError(f"{name} failed (rc={rc}): {lib.hipseg_last_error().decode()}")

    call.__name__ = name
    return call


for _n in PROTOTYPES:
    globals()[_n[len("hipseg_"):]] = _wrap(_n)
