"""torch.autograd.Function wrappers over the C ABI (include/hipseg.h).

PyTorch is plumbing here: device memory (caching allocator), the current HIP stream and the
autograd tape.  Every numerical step runs in libhipseg.so.  Activations are NHWC in HBM and are
handed around as logical-NCHW tensors with channels_last strides, so module signatures keep the
reference's (B, C, H, W) shapes.
"""
import contextlib
import os
import weakref

import torch

from . import _lib as L

BN_EPS = 1e-5
BN_MOMENTUM = 0.1

_FORCED = [os.environ.get("HIPSEG_PRECISION", "auto").lower()]


def precision():
    """'bf16' or 'fp32'.  auto: bf16 under torch.autocast (the reference's wrappers train under
    autocast, models/model_wrappers.py:170), fp32 otherwise (TestWrapper, model_wrappers.py:383)."""
    f = _FORCED[0]
    if f in ("bf16", "fp32"):
        return f
    return "bf16" if torch.is_autocast_enabled() else "fp32"


@contextlib.contextmanager
def precision_mode(p):
    assert p in ("auto", "bf16", "fp32")
    old, _FORCED[0] = _FORCED[0], p
    try:
        yield
    finally:
        _FORCED[0] = old


def _dt(t):
    if t.dtype == torch.bfloat16:
        return L.BF16
    if t.dtype == torch.float32:
        return L.F32
    raise TypeError(f"hipseg: unsupported activation dtype {t.dtype}")


def _tdtype(prec):
    return torch.bfloat16 if prec == "bf16" else torch.float32


def _stream():
    """raw handle of the current stream of the current device.  (torch.cuda.current_stream().cuda_stream builds a Stream
    object and resolves the device index in Python: 9.5 us per call, 0.8 ms of host time per train step at one call
    per kernel launch -- scripts/prof_python.py)"""
    return _raw_stream(_cur_device())


_raw_stream = torch._C._cuda_getCurrentRawStream
_cur_device = torch._C._cuda_getDevice


def _require_gpu(t):
    if not t.is_cuda:
        raise RuntimeError("hipseg: the HIP path needs CUDA/HIP tensors (no CPU fallback); got a CPU tensor")


def ptr(t):
    return 0 if t is None else t.data_ptr()


def nhwc_empty(B, C, H, W, dtype, device):
    """logical (B,C,H,W) tensor whose storage is dense NHWC."""
    return torch.empty((B, H, W, C), dtype=dtype, device=device).permute(0, 3, 1, 2)


def is_nhwc(t):
    return t.dim() == 4 and t.permute(0, 2, 3, 1).is_contiguous()


def as_nhwc(t, dtype):
    """dense-NHWC view/copy of a logical NCHW tensor in `dtype` (layout plumbing at block boundaries)."""
    _require_gpu(t)
    if t.dtype == dtype and is_nhwc(t):
        return t
    return t.to(dtype).permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)


def _f32(n, device):
    return torch.empty(n, dtype=torch.float32, device=device)


# ---- gradient destinations.  HipDDP attaches to each parameter the slot of its flat fp32 reduction bucket
# (`p._hipseg_slot = (flat bucket, element offset, ids of the parameters whose slot this backward has handed out)`); the backward kernels then write parameter gradients STRAIGHT into
# the bucket (a fresh view per backward, so autograd's AccumulateGrad adopts it without a copy) and the reducer's hooks
# have nothing to move.  Parameters without a slot, or whose .grad is being accumulated into, get an ordinary new tensor.
def grad_out(p):
    s = getattr(p, "_hipseg_slot", None)
    if s is None or p.grad is not None:
        return torch.empty(p.shape, dtype=torch.float32, device=p.device)
    flat, o, taken = s
    if id(p) in taken:  # second use of the parameter in this backward: its slot already holds the first partial gradient
        return torch.empty(p.shape, dtype=torch.float32, device=p.device)
    taken.add(id(p))
    return flat[o:o + p.numel()].view(p.shape)


def grad_out_pair(first, second):
    """one contiguous fp32 tensor [first.numel() + second.numel()] that is BOTH parameters' bucket slots when they are
    adjacent in that order (BatchNorm bias, weight: the layout bn_bwd's [sum g, sum g*xhat] vector has), else new."""
    n1, n2 = first.numel(), second.numel()
    a, b = getattr(first, "_hipseg_slot", None), getattr(second, "_hipseg_slot", None)
    if (a is not None and b is not None and a[0] is b[0] and a[1] + n1 == b[1] and first.grad is None
            and second.grad is None and id(first) not in a[2] and id(second) not in a[2]):
        a[2].add(id(first))
        a[2].add(id(second))
        return a[0][a[1]:a[1] + n1 + n2]
    return torch.empty(n1 + n2, dtype=torch.float32, device=first.device)


# ---------------------------------------------------------------------------------------------
# optional per-launch timing (bench.py's roofline leg): when PROFILE is a list, every kernel launch of the
# convolution / BatchNorm / stem / head / loss / resize groups is bracketed by HIP events on the launch stream and
# recorded as (kernel key, algorithmic FLOPs, algorithmic HBM bytes, event0, event1) -- FLOPs for the MFMA-bound
# groups, bytes (every operand read once, every result written once) for the HBM-bound ones,
# both for the convolutions that sit between the two rooflines.  None (default) = zero overhead.
PROFILE = None
_TAPS = {L.CONV3: 9, L.CONV1: 1, L.CONV2S2: 4, L.CONVT: 1}
_MODE_NAME = {L.CONV3: "CONV3", L.CONV1: "CONV1", L.CONV2S2: "CONV2S2", L.CONVT: "CONVT"}


_EVENT_POOL = []  # timing events are created once: building two per launch costs the eager profiling loop ~10 us each


def _timed(key, flops, fn, *args, nbytes=0.0):
    if PROFILE is None:
        return fn(*args)
    i = 2 * len(PROFILE)
    while len(_EVENT_POOL) < i + 2:
        _EVENT_POOL.append(torch.cuda.Event(enable_timing=True))
    e0, e1 = _EVENT_POOL[i], _EVENT_POOL[i + 1]
    e0.record()
    fn(*args)
    e1.record()
    PROFILE.append((key, flops, nbytes, e0, e1))


def _hbm(key, nbytes, fn, *args):
    """an HBM-bound launch: timed (when profiling) against its algorithmic bytes"""
    if PROFILE is None:
        return fn(*args)
    _timed("hbm:" + key, 0.0, fn, *args, nbytes=float(nbytes))


def _esz(dt):
    return 2 if dt == L.BF16 else 4


def igemm(dt, mode, in0, c0, in1, c1, wp, bias, out0, n0, out1, n1, stats, B, H, W):
    """hipseg_conv_igemm on the current stream (tensors or None in, raw pointers out)."""
    if PROFILE is None:  # (the common case: no key string, no flop count)
        L.conv_igemm(dt, mode, ptr(in0), c0, ptr(in1), c1, ptr(wp), ptr(bias), ptr(out0), n0, ptr(out1), n1, ptr(stats), B,
                     H, W, _stream())
        return
    N = 4 * n0 if mode == L.CONVT else n0 + n1
    bn = 128 if N > 64 else (64 if N > 32 else 32)
    key = f"conv_igemm<{'bf16' if dt == L.BF16 else 'f32'},{_MODE_NAME[mode]},BN{bn}>"
    flops = 2.0 * B * H * W * N * (c0 + c1) * _TAPS[mode]
    nbytes = float(B * H * W) * (c0 + c1 + N) * _esz(dt)  # input read once + output written once (CONVT: 4 Cout per pixel)
    _timed(key, flops, L.conv_igemm, dt, mode, ptr(in0), c0, ptr(in1), c1, ptr(wp), ptr(bias), ptr(out0), n0,
           ptr(out1), n1, ptr(stats), B, H, W, _stream(), nbytes=nbytes)


# ---- packed MFMA operands of the weights.  A model packs ALL its conv / ConvT weights with one launch at the start of
# a forward (prepack); the per-layer helpers below first look the weight up in that cache (same storage, same
# version counter) and only pack individually when it is not there (stand-alone block calls, tests).
# keyed by the weight tensor OBJECT (weakly): an entry dies with its parameter.  (Keyed by address, a freed model's entry
# was served to a new tensor that the allocator placed at the same address with the same shape and version counter.)
_PLANS = weakref.WeakKeyDictionary()  # module -> {dt: _PackPlan}; dies with the module (no id() reuse aliasing)


class _PackPlan:
    __slots__ = ("ptrs", "owners", "params", "bufs", "desc", "max_total", "n")


def _pack_sizes(w, kind, dt):
    if kind == 0:
        cout, cin, k, _ = w.shape
        return k * k * L.kpad(cin, dt) * L.npad(cout), k * k * L.kpad(cout, dt) * L.npad(cin)
    cin, cout = w.shape[0], w.shape[1]
    return L.kpad(cin, dt) * L.npad(4 * cout), 4 * L.kpad(cout, dt) * L.npad(cin)


def _scan_pack_owners(module):
    owners = []
    for m in module.modules():
        if isinstance(m, torch.nn.ConvTranspose2d) and m.kernel_size == (2, 2):
            owners.append((m, 1))
        elif isinstance(m, torch.nn.Conv2d) and m.kernel_size == (3, 3):
            owners.append((m, 0))
    return owners


def prepack(module, prec):
    """Pack every 3x3 Conv2d / 2x2 ConvTranspose2d weight of `module` into its MFMA operand layouts (forward and
    data-gradient) with ONE kernel launch.  The descriptor table and the operand buffers persist per (module, dt).
    This runs at the head of every forward, in front of the first kernel of an eagerly launched step (the reference's
    loop synchronises on loss.item() every step, models/model_wrappers.py:180, so the GPU idles until it returns): the
    layer list is scanned once per module and revalidated by identity (20 `is` tests, not a walk over 86 submodules);
    a layer added to the module later is simply packed on its own by its first convolution call."""
    import ctypes
    if os.environ.get("HIPSEG_NO_PREPACK"):  # A/B switch: per-layer pack launches
        return
    dt = L.BF16 if prec == "bf16" else L.F32
    plans = _PLANS.get(module)
    if plans is None:
        plans = _PLANS[module] = {}
    plan = plans.get(dt)
    if plan is not None:
        for (m, _), w, p0 in zip(plan.owners, plan.params, plan.ptrs):
            if m.weight is not w or w.data_ptr() != p0:
                plan = None  # a parameter was replaced or moved: rebuild
                break
    if plan is None:
        owners = _scan_pack_owners(module)
        if not owners:
            return
        params = [m.weight for m, _ in owners]
        dev = params[0].device
        td = _tdtype(prec)
        plan = _PackPlan()
        plan.owners, plan.params, plan.n = owners, params, len(params)
        plan.ptrs = tuple(w.data_ptr() for w in params)
        plan.bufs, plan.max_total = [], 0
        host = ctypes.create_string_buffer(plan.n * L.pack_desc_size())
        for i, ((_, kind), w) in enumerate(zip(owners, params)):
            n0, n1 = _pack_sizes(w, kind, dt)
            wp, wpt = torch.empty(n0, dtype=td, device=dev), torch.empty(n1, dtype=td, device=dev)
            plan.bufs.append((wp, wpt))
            plan.max_total = max(plan.max_total, n0, n1)
            if kind == 0:
                cout, cin, k = w.shape[0], w.shape[1], w.shape[2]
            else:
                cin, cout, k = w.shape[0], w.shape[1], 2
            L.pack_desc_fill(ctypes.addressof(host), i, ptr(w), ptr(wp), ptr(wpt), kind, dt, cout, cin, k)
        plan.desc = torch.frombuffer(bytearray(host.raw), dtype=torch.uint8).to(dev)
        plans[dt] = plan
    L.pack_batch(ptr(plan.desc), plan.n, dt, plan.max_total, _stream())
    for w, p0, (wp, wpt) in zip(plan.params, plan.ptrs, plan.bufs):
        _set_cached_pack(w, dt, (w._version, p0, wp, wpt))


# packed operands of a weight tensor: an attribute of the tensor object itself, {dt: (version, data_ptr, wp, wpt)} (dies
# with the tensor; an attribute read instead of a weak-dictionary lookup on every convolution call)
def _set_cached_pack(w, dt, entry):
    c = getattr(w, "_hipseg_pack", None)
    if c is None:
        c = {}
        w._hipseg_pack = c
    c[dt] = entry


def _cached_pack(w, dt):
    c = getattr(w, "_hipseg_pack", None)
    if c is None:
        return None
    e = c.get(dt)
    if e is not None and e[0] == w._version and e[1] == w.data_ptr() and e[2].device == w.device:
        return e[2], e[3]
    return None


def _pack_conv(w, dt, transpose):
    c = _cached_pack(w, dt)
    if c is not None:
        return c[1] if transpose else c[0]
    cout, cin, k, _ = w.shape
    K, N = (cout, cin) if transpose else (cin, cout)
    n = k * k * L.kpad(K, dt) * L.npad(N)
    wp = torch.empty(n, dtype=torch.bfloat16 if dt == L.BF16 else torch.float32, device=w.device)
    L.pack_conv_weight(ptr(w), ptr(wp), dt, cout, cin, k, int(transpose), _stream())
    return wp


def _pack_conv_both(w, dt):
    """forward and data-gradient operands of one conv weight, one launch."""
    c = _cached_pack(w, dt)
    if c is not None:
        return c
    cout, cin, k, _ = w.shape
    td = torch.bfloat16 if dt == L.BF16 else torch.float32
    wp = torch.empty(k * k * L.kpad(cin, dt) * L.npad(cout), dtype=td, device=w.device)
    wpt = torch.empty(k * k * L.kpad(cout, dt) * L.npad(cin), dtype=td, device=w.device)
    L.pack_conv_weight_both(ptr(w), ptr(wp), ptr(wpt), dt, cout, cin, k, _stream())
    return wp, wpt


def _pack_convT(w, dt, transpose):
    c = _cached_pack(w, dt)
    if c is not None:
        return c[1] if transpose else c[0]
    cin, cout = w.shape[0], w.shape[1]
    if transpose:
        n = 4 * L.kpad(cout, dt) * L.npad(cin)
    else:
        n = L.kpad(cin, dt) * L.npad(4 * cout)
    wp = torch.empty(n, dtype=torch.bfloat16 if dt == L.BF16 else torch.float32, device=w.device)
    L.pack_convT_weight(ptr(w), ptr(wp), dt, cin, cout, int(transpose), _stream())
    return wp


def _wgrad(dt, mode, p0, p1, q, dw, B, H, W):
    cu0 = p0.shape[1]
    cu1 = p1.shape[1] if p1 is not None else 0
    cv = q.shape[1]
    slabs = _f32(L.wgrad_workspace_elems(mode, cu0 + cu1, cv, B, H, W), dw.device)
    nt = 9 if mode == L.CONV3 else (4 if mode == L.CONVT else 1)
    key = f"conv_wgrad<{'bf16' if dt == L.BF16 else 'f32'},{_MODE_NAME[mode]}>(+reduce)"
    # bytes: both operands read once (CONVT: p = dy has 4 output pixels of cu0 channels per input pixel)
    nbytes = float(B * H * W) * ((4 if mode == L.CONVT else 1) * (cu0 + cu1) + cv) * _esz(dt)
    _timed(key, 2.0 * B * H * W * (cu0 + cu1) * cv * nt, L.conv_wgrad, dt, mode, ptr(p0), cu0, ptr(p1), cu1, ptr(q),
           cv, ptr(dw), ptr(slabs), B, H, W, _stream(), nbytes=nbytes)


def _wgrad_pair(dt, pa0, pa1, qa, dwa, pb, qb, dwb, B, H, W):
    """both 3x3 weight gradients of a ConvBlock in one launch (hipseg_conv_wgrad_pair; the caller asked
    L.conv_wgrad_pair_applies).  Timed as ONE entry of the weight-gradient group with the FLOPs of both layers."""
    ca0 = pa0.shape[1]
    ca1 = pa1.shape[1] if pa1 is not None else 0
    cb, cv = pb.shape[1], qa.shape[1]
    slabs = _f32(max(L.wgrad_workspace_elems(L.CONV3, ca0 + ca1, cv, B, H, W), L.wgrad_workspace_elems(L.CONV3, cb, cv, B, H, W)),
                 dwa.device)
    key = f"conv_wgrad<{'bf16' if dt == L.BF16 else 'f32'},CONV3>(+reduce)"
    _timed(key, 2.0 * B * H * W * (ca0 + ca1 + cb) * cv * 9, L.conv_wgrad_pair, dt, ptr(pa0), ca0, ptr(pa1), ca1, ptr(qa),
           ptr(dwa), ptr(pb), cb, ptr(qb), ptr(dwb), cv, ptr(slabs), B, H, W, _stream(),
           nbytes=float(B * H * W) * (ca0 + ca1 + cb + 2 * cv) * _esz(dt))


class _BN:
    """per-layer BatchNorm state of one forward (device vectors of length C)."""
    __slots__ = ("mean", "invstd", "scale", "shift")

    def __init__(self, C, device):
        v = _f32(4 * C, device)
        self.mean, self.invstd, self.scale, self.shift = v[:C], v[C:2 * C], v[2 * C:3 * C], v[3 * C:]


def _conv_bn_relu(dt, x0, x1, w, b, gamma, beta, rm, rv, nbt, train, pool, need_t=False, inference=False, lazy=False,
                  in_bn=None):
    """conv3x3 (+bias, +BN batch statistics in the epilogue) -> BN finalize -> BN-apply+ReLU(+pool).
    Returns (raw conv output, activated output, bn state, data-gradient weight operand or None).
    `inference` (eval mode and no gradient wanted: the validation loops, model_wrappers.py:193-215) without a pool runs
    conv -> running-statistics BatchNorm -> ReLU as ONE kernel (hipseg_conv_affine_relu): no pre-normalisation tensor,
    no second pass; returns (None, activated output, None, None)."""
    B, _, H, W = x0.shape
    cout = w.shape[0]
    dev = x0.device
    c0 = x0.shape[1]
    c1 = x1.shape[1] if x1 is not None else 0
    if inference and not train and not pool:
        fold = _f32(2 * cout, dev)  # [scale | shift]: BatchNorm (running statistics) and the conv bias folded
        s = _stream()
        L.bn_fold(ptr(gamma), ptr(beta), ptr(rm), ptr(rv), ptr(b), BN_EPS, cout, ptr(fold), ptr(fold[cout:]), s)
        act = nhwc_empty(B, cout, H, W, x0.dtype, dev)
        key = f"conv_igemm<{'bf16' if dt == L.BF16 else 'f32'},CONV3,BN{128 if cout > 64 else (64 if cout > 32 else 32)}>"
        _timed(key, 2.0 * B * H * W * cout * (c0 + c1) * 9, L.conv_affine_relu, dt, ptr(x0), c0, ptr(x1), c1,
               ptr(_pack_conv(w, dt, False)), ptr(fold), ptr(fold[cout:]), ptr(act), cout, B, H, W, s)
        return None, act, None, None
    if need_t:
        wp, wpt = _pack_conv_both(w, dt)
    else:
        wp, wpt = _pack_conv(w, dt, False), None
    raw = nhwc_empty(B, cout, H, W, x0.dtype, dev)
    bn = _BN(cout, dev)
    s = _stream()
    if train:
        stats = _f32(L.conv_mtiles(B, H, W) * 2 * cout, dev)
        if in_bn is not None:  # x0 is a PRE-normalisation tensor: relu(x0 * scale + shift) applied in the load path
            key = f"conv_igemm<bf16,CONV3,BN{128 if cout > 64 else (64 if cout > 32 else 32)}>"
            _timed(key, 2.0 * B * H * W * cout * c0 * 9, L.conv3_bnrelu_in, dt, ptr(x0), c0, ptr(in_bn.scale), ptr(in_bn.shift),
                   ptr(wp), ptr(b), ptr(raw), cout, ptr(stats), B, H, W, s, nbytes=B * H * W * (c0 + cout) * _esz(dt))
        else:
            igemm(dt, L.CONV3, x0, c0, x1, c1, wp, b, raw, cout, None, 0, stats, B, H, W)
        rows = L.conv_stats_rows(dt, L.CONV3, c0, c1, cout, 0, B, H, W)
        L.bn_finalize(ptr(stats), rows, cout, float(B * H * W), ptr(gamma), ptr(beta), BN_EPS, BN_MOMENTUM, ptr(rm), ptr(rv),
                      ptr(nbt), ptr(bn.mean), ptr(bn.invstd), ptr(bn.scale), ptr(bn.shift), s)
    else:
        igemm(dt, L.CONV3, x0, c0, x1, c1, wp, b, raw, cout, None, 0, None, B, H, W)
        L.bn_eval_params(ptr(gamma), ptr(beta), ptr(rm), ptr(rv), BN_EPS, cout, ptr(bn.mean), ptr(bn.invstd),
                         ptr(bn.scale), ptr(bn.shift), s)
    if lazy:  # the consumer applies BatchNorm + ReLU when it loads `raw` (ConvBlockFn with a head)
        return raw, None, bn, wpt
    Ho, Wo = (H // 2, W // 2) if pool else (H, W)
    act = nhwc_empty(B, cout, Ho, Wo, x0.dtype, dev)
    _hbm("bn_relu_apply" + ("+pool" if pool else ""), B * H * W * cout * _esz(dt) * (1.25 if pool else 2.0),
         L.bn_relu_apply, dt, ptr(raw), ptr(bn.scale), ptr(bn.shift), ptr(act), B, H, W, cout, int(pool), s)
    return raw, act, bn, wpt


def _bn_relu_bwd(dt, dy, raw, bn, train, pool, bias, gamma, beta, reduced=None, dy2=None):
    """backward through [pool](relu(bn(raw))): returns (d_raw, dgamma, dbeta, dbias_conv); `bias`, `gamma`, `beta` are
    the conv-bias / BN parameters (for their gradient destinations).  `reduced` = (partial, rows): the reduction's
    partial rows already exist (written by the data-gradient kernel that produced dy, hipseg_conv3_dgrad_bnstats)."""
    B, C, H, W = raw.shape
    dev = raw.device
    s = _stream()
    sums = grad_out_pair(beta, gamma)  # [sum g | sum g*xhat] = [dbeta | dgamma]
    if reduced is not None:
        partial, nblk = reduced
    else:
        nblk = L.bn_bwd_blocks(B, H, W, C, dt, int(pool))
        partial = _f32(nblk * 2 * C, dev)
        n2 = (0.25 if pool else 1.0) if dy2 is not None else 0.0  # (a second gradient tensor of dy's size)
        _hbm("bn_bwd_reduce" + ("+pool" if pool else ""), B * H * W * C * _esz(dt) * ((1.25 if pool else 2.0) + n2),
             L.bn_bwd_reduce2, dt, ptr(dy), ptr(dy2), ptr(raw), ptr(bn.mean), ptr(bn.invstd), ptr(bn.scale), ptr(bn.shift), ptr(partial),
             B, H, W, C, int(pool), s)
    # conv bias in front of train-mode BN: d(bias) = sum_pixels d_raw == 0 exactly (sum(g - mean g) = 0 and
    # sum(xhat) = 0); the reference's autograd returns only rounding noise here (~1e-8).  The finalize launch writes it.
    dbias = grad_out(bias)
    L.colsum_finalize(ptr(partial), nblk, 2, C, ptr(sums), ptr(dbias) if train else 0, s)
    draw = nhwc_empty(B, C, H, W, raw.dtype, dev)
    n2 = (0.25 if pool else 1.0) if dy2 is not None else 0.0
    _hbm("bn_bwd_apply" + ("+pool" if pool else ""), B * H * W * C * _esz(dt) * ((2.25 if pool else 3.0) + n2),
         L.bn_bwd_apply2, dt, ptr(dy), ptr(dy2), ptr(raw), ptr(bn.mean), ptr(bn.invstd), ptr(bn.scale), ptr(bn.shift), ptr(sums),
         float(B * H * W), 0 if train else 1, ptr(draw), 0, B, H, W, C, int(pool), s)
    if not train:
        npix = B * H * W
        part = _f32(L.colsum_blocks(npix, C, dt) * C, dev)
        L.colsum(dt, ptr(draw), npix, C, ptr(part), ptr(dbias), s)
    return draw, sums[C:], sums[:C], dbias


_NO_FUSED_INFERENCE = bool(os.environ.get("HIPSEG_NO_FUSED_INFERENCE"))  # A/B switch (scripts/bench_infer.py)
_NO_BLOCK_CALLS = bool(os.environ.get("HIPSEG_NO_BLOCK_CALLS"))  # A/B switch: per-op ctypes calls instead of one per block
_PEROP_ON_LOAD = True  # the per-op path takes BatchNorm-on-load where the block call does (tests switch it off: materialised reference)


def _block_forward(ctx, dt, x0, x1, w1, b1, g1, be1, w2, b2, g2, be2, rm1, rv1, nbt1, rm2, rv2, nbt2, train, pool, grad,
                   lazy=False):
    """ConvBlock forward through ONE C call (hipseg_convblock_forward: conv -> BN statistics -> BN-apply+ReLU twice);
    Python only allocates.  Same kernels, same order as the per-op path (_conv_bn_relu).  Returns (out, tensors to save
    for backward); `lazy`: out is None, the last BN-apply + ReLU is left to the consumer of raw2 (saved[6]) / bn2."""
    import ctypes

    B, _, H, W = x0.shape
    cout, dev, td = w1.shape[0], x0.device, x0.dtype
    if grad:
        wp1, wp1t = _pack_conv_both(w1, dt)
        wp2, wp2t = _pack_conv_both(w2, dt)
    else:
        wp1, wp2, wp1t, wp2t = _pack_conv(w1, dt, False), _pack_conv(w2, dt, False), None, None
    raw1, a1, raw2 = (nhwc_empty(B, cout, H, W, td, dev) for _ in range(3))
    out = None if lazy else (nhwc_empty(B, cout, H // 2, W // 2, td, dev) if pool else nhwc_empty(B, cout, H, W, td, dev))
    bnv = _f32(8 * cout, dev)
    stats = _f32(L.conv_mtiles(B, H, W) * 2 * cout, dev) if train else None
    A = L.ConvBlockArgs()
    A.dtype, A.B, A.H, A.W, A.C0, A.C1, A.Cout = dt, B, H, W, x0.shape[1], (x1.shape[1] if x1 is not None else 0), cout
    A.train, A.pool, A.eps, A.momentum = int(train), int(pool), BN_EPS, BN_MOMENTUM
    A.x0, A.x1, A.wp1, A.wp2 = ptr(x0), ptr(x1), ptr(wp1), ptr(wp2)
    A.b1, A.g1, A.be1, A.b2, A.g2, A.be2 = ptr(b1), ptr(g1), ptr(be1), ptr(b2), ptr(g2), ptr(be2)
    A.rm1, A.rv1, A.rm2, A.rv2, A.nbt1, A.nbt2 = ptr(rm1), ptr(rv1), ptr(rm2), ptr(rv2), ptr(nbt1), ptr(nbt2)
    A.raw1, A.a1, A.raw2, A.out = ptr(raw1), ptr(a1), ptr(raw2), ptr(out)
    A.bn1, A.bn2, A.stats = bnv.data_ptr(), bnv.data_ptr() + 16 * cout, ptr(stats)
    L.convblock_forward(ctypes.addressof(A), _stream())
    if grad:
        ctx.train, ctx.pool, ctx.dt, ctx.blk = train, pool, dt, A
        ctx.small = (b1, g1, be1, b2, g2, be2)
        return out, (x0, x1, w1, w2, raw1, a1, raw2, wp1t, wp2t, bnv)
    return out, ()


def _block_partial_rows(dt, B, C, H, W, pool, up_cout=0):
    """rows of BatchNorm-backward partial sums a block's backward may see: the reduce kernels' blocks, the tiles of the
    data-gradient kernel that reduces the first layer's sums in its epilogue (hipseg_conv3_dgrad_bnstats), the blocks
    of the head kernel that reduces the second layer's (hipseg_head_bwd_bnrelu), or the rows of the data gradient of a
    ConvTranspose2d with `up_cout` output channels that consumes the block's output (hipseg_convT_dgrad_bnstats)."""
    return max(L.bn_bwd_blocks(B, H, W, C, dt, int(pool)), L.bn_bwd_blocks(B, H, W, C, dt, 0),
               L.conv3_dgrad_bnstats_rows(dt, C, C, B, H, W), L.head_bwd_blocks(B, H, W),
               L.convT_dgrad_bnstats_rows(dt, up_cout, C, B, H, W) if up_cout else 0)


def _block_backward(ctx, dout, dout2=None, reduced=None):
    """the matching backward through hipseg_convblock_backward (BN backward x2, weight gradients x2, data gradients).
    `reduced` = (partial, rows): the second layer's BatchNorm-backward rows already sit in that workspace."""
    import ctypes

    x0, x1, w1, w2, raw1, a1, raw2, wp1t, wp2t, bnv = ctx.saved_tensors[:10]
    A, dt, train, pool = ctx.blk, ctx.dt, ctx.train, ctx.pool
    B, C, H, W = raw2.shape
    dev, td = raw2.device, raw2.dtype
    dout = as_nhwc(dout, td)
    dout2 = as_nhwc(dout2, td) if dout2 is not None else None
    b1, g1, be1, b2, g2, be2 = ctx.small
    c0 = x0.shape[1]
    c1 = x1.shape[1] if x1 is not None else 0
    need0 = ctx.needs_input_grad[0]
    need1 = x1 is not None and ctx.needs_input_grad[1]
    draw, da1 = nhwc_empty(B, C, H, W, td, dev), nhwc_empty(B, C, H, W, td, dev)
    # d raw1 re-uses d raw2's buffer, unless both weight gradients run as ONE paired launch after d raw1 exists
    draw1 = nhwc_empty(B, C, H, W, td, dev) if L.conv_wgrad_pair_applies(dt, c0, c1, C, C, B, H, W) else draw
    dx0 = dx1 = None
    if need0 or need1:
        dx0 = nhwc_empty(B, c0, H, W, td, dev)
        dx1 = nhwc_empty(B, c1, H, W, td, dev) if c1 else None
    dw1, dw2, db1, db2 = grad_out(w1), grad_out(w2), grad_out(b1), grad_out(b2)
    sums1, sums2 = grad_out_pair(be1, g1), grad_out_pair(be2, g2)
    partial = reduced[0] if reduced is not None else _f32(_block_partial_rows(dt, B, C, H, W, pool) * 2 * C, dev)
    A.dout_rows = reduced[1] if reduced is not None else 0
    slabs = _f32(max(L.wgrad_workspace_elems(L.CONV3, c0 + c1, C, B, H, W), L.wgrad_workspace_elems(L.CONV3, C, C, B, H, W)), dev)
    colpart = None if train else _f32(L.colsum_blocks(B * H * W, C, dt) * C, dev)
    A.dout, A.draw2, A.da1, A.draw1, A.dx0, A.dx1 = ptr(dout), ptr(draw), ptr(da1), ptr(draw1), ptr(dx0), ptr(dx1)
    A.dout2 = ptr(dout2)
    A.dw1, A.dw2, A.db1, A.db2, A.sums1, A.sums2 = ptr(dw1), ptr(dw2), ptr(db1), ptr(db2), ptr(sums1), ptr(sums2)
    A.partial, A.slabs, A.colpart = ptr(partial), ptr(slabs), ptr(colpart)
    A.wp1t, A.wp2t, A.need_dx = ptr(wp1t), ptr(wp2t), int(dx0 is not None)
    L.convblock_backward(ctypes.addressof(A), _stream())
    return (dx0, dx1, dw1, db1, sums1[C:], sums1[:C], dw2, db2, sums2[C:], sums2[:C])


# A/B switch (read by models/processing_blocks.py): last ConvBlock and 1x1 head as two autograd nodes, with the BN-apply,
# head and BN-backward reduce launches each on their own
_NO_HEAD_FUSE = bool(os.environ.get("HIPSEG_NO_HEAD_FUSE"))
# likewise: a ConvBlock and the ConvTranspose2d that consumes its output as two nodes (plain data gradient + reduce launch)
_NO_UP_FUSE = bool(os.environ.get("HIPSEG_NO_UP_FUSE"))


def _head_fwd(dt, x, w, b, bn=None):
    """1x1 head on NHWC activations -> NCHW fp32 logits; `bn`: x is a pre-normalisation tensor, relu(x * scale + shift)
    is applied on load (hipseg_head_fwd_bnrelu)."""
    B, cin, H, W = x.shape
    cout = w.shape[0]
    logits = torch.empty((B, cout, H, W), dtype=torch.float32, device=x.device)
    nbytes = B * H * W * (cin * _esz(dt) + cout * 4)
    if bn is None:
        _hbm("head_fwd", nbytes, L.head_fwd, dt, ptr(x), ptr(w), ptr(b), ptr(logits), B, H, W, cin, cout, _stream())
    else:
        _hbm("head_fwd+bn_relu", nbytes, L.head_fwd_bnrelu, dt, ptr(x), ptr(bn[2]), ptr(bn[3]), ptr(w), ptr(b), ptr(logits),
             B, H, W, cin, cout, _stream())
    return logits


def _head_bwd(dt, x, dl, w, bias, bn=None, bn_partial=None):
    """(dx, dw, db) of the head; `bn` = (mean, invstd, scale, shift) and `bn_partial`: the on-load form, which also leaves
    hipseg_head_bwd_blocks() rows of the layer's BatchNorm-backward sums in bn_partial."""
    B, cin, H, W = x.shape
    cout = w.shape[0]
    dl = dl.float().contiguous()
    nblk = L.head_bwd_blocks(B, H, W)
    part = _f32(nblk * cout * (cin + 1), x.device)
    dx = nhwc_empty(B, cin, H, W, x.dtype, x.device)
    dw, db = grad_out(w), grad_out(bias)
    nbytes = B * H * W * (2 * cin * _esz(dt) + cout * 4)
    if bn is None:
        _hbm("head_bwd", nbytes, L.head_bwd, dt, ptr(x), ptr(dl), ptr(w), ptr(dx), ptr(part), ptr(dw), ptr(db), B, H, W, cin,
             cout, _stream())
    else:
        _hbm("head_bwd+bn_bwd_sums", nbytes, L.head_bwd_bnrelu, dt, ptr(x), ptr(bn[0]), ptr(bn[1]), ptr(bn[2]), ptr(bn[3]),
             ptr(dl), ptr(w), ptr(dx), ptr(part), ptr(dw), ptr(db), ptr(bn_partial), B, H, W, cin, cout, _stream())
    return dx, dw, db


def _convT_fwd(dt, x, w, b):
    """nn.ConvTranspose2d(Cin, Cout, 2, stride=2) on NHWC activations"""
    B, cin, H, W = x.shape
    cout = w.shape[1]
    y = nhwc_empty(B, cout, 2 * H, 2 * W, x.dtype, x.device)
    igemm(dt, L.CONVT, x, cin, None, 0, _pack_convT(w, dt, False), b, y, cout, None, 0, None, B, H, W)
    return y


def _convT_bwd(dt, x, w, bias, dy, need_dx, bn_x=None, bn=None, partial=None):
    """(dx, dw, db, rows) of the ConvTranspose2d.  `bn_x`, `bn` = (mean, invstd, scale, shift), `partial`: x is
    relu(bn(bn_x)) of a train- or eval-mode BatchNorm whose backward comes next -- where a kernel with that epilogue takes
    the shape the data gradient also leaves `rows` rows of its sums in `partial` (hipseg_convT_dgrad_bnstats), else 0."""
    B, cin, H, W = x.shape
    cout = w.shape[1]
    dev = x.device
    s = _stream()
    dy = as_nhwc(dy, x.dtype)
    dw, db = grad_out(w), grad_out(bias)
    # weight AND bias gradient from one kernel (the bias gradient is the column sum of the dY fragments its MFMAs
    # hold): dY is read once, one reduction launch writes both parameters' layouts
    work = _f32(L.convT_wgrad_workspace_elems(cin, cout, B, H, W), dev)
    key = f"conv_wgrad<{'bf16' if dt == L.BF16 else 'f32'},CONVT>(+bias,+reduce)"
    _timed(key, 2.0 * B * H * W * 4 * cout * cin, L.convT_wgrad_bias, dt, ptr(dy), ptr(x), ptr(dw), ptr(db), ptr(work), B,
           H, W, cin, cout, s, nbytes=float(B * H * W) * (4 * cout + cin) * _esz(dt))
    dx, rows = None, 0
    if need_dx:
        wpt = _pack_convT(w, dt, True)
        dx = nhwc_empty(B, cin, H, W, x.dtype, dev)
        rows = L.convT_dgrad_bnstats_rows(dt, cout, cin, B, H, W) if bn is not None else 0
        if rows:
            _timed(f"conv_igemm<bf16,CONV2S2,BN{128 if cin > 64 else 64}>+bn_bwd_sums", 2.0 * B * H * W * 4 * cout * cin,
                   L.convT_dgrad_bnstats, dt, ptr(dy), cout, ptr(wpt), ptr(dx), cin, ptr(bn_x), ptr(bn[0]), ptr(partial), B, H, W,
                   s, nbytes=float(B * H * W) * (4 * cout + 2 * cin) * _esz(dt))
        else:
            igemm(dt, L.CONV2S2, dy, cout, None, 0, wpt, None, dx, cin, None, 0, None, B, H, W)
    return dx, dw, db, rows


class ConvBlockFn(torch.autograd.Function):
    """[cat(x0,x1)] -> conv3x3 -> BN -> ReLU -> conv3x3 -> BN -> ReLU [-> MaxPool2d(2,2)]
    = ConvBlock / ConvBlockDownsample / the conv half of ConvBlockUpsampleSkip
    (models/processing_blocks.py:40-52, 69-77, 108-109).
    `two`: return the output TWICE (the second an alias of the first), one per consumer -- an encoder block's pooled
    output feeds the next block and a decoder block's skip input (models/UNet.py:64-72).  Given one tensor object,
    autograd sums the two gradients in an elementwise pass of its own before this backward runs; with one alias per
    consumer each gradient arrives on its own and the BatchNorm-backward kernels read both (hipseg_bn_bwd_*2).
    `hw`, `hb`: weight and bias of the 1x1 head that consumes the block's output (models/UNet.py:72-73, dec4 -> out);
    the function then returns the head's NCHW fp32 logits.  In train mode the block's last BatchNorm + ReLU is applied
    in the head's load path and the head's backward leaves that layer's BatchNorm-backward sums behind
    (hipseg_head_fwd_bnrelu / hipseg_head_bwd_bnrelu): two full-resolution passes fewer; otherwise block and head simply
    run one after the other.
    `uw`, `ub`: weight and bias of the ConvTranspose2d(k2, s2) that consumes the block's output (models/UNet.py:66-71,
    processing_blocks.py:102: bottleneck -> dec1.up, dec_k.conv -> dec_k+1.up); the function then returns the up-sampled
    tensor, and the ConvTranspose2d's data gradient -- which IS this block's dout -- also reduces the second layer's
    BatchNorm-backward sums where a kernel with that epilogue takes the shape (hipseg_convT_dgrad_bnstats)."""

    @staticmethod
    def forward(ctx, x0, x1, w1, b1, g1, be1, w2, b2, g2, be2, rm1, rv1, nbt1, rm2, rv2, nbt2, train, pool,
                no_grad=False, two=False, hw=None, hb=None, uw=None, ub=None):
        dt = _dt(x0)
        # (grad mode is off inside Function.forward, and needs_input_grad stays True for parameters under
        # torch.no_grad(): the caller passes whether a graph is being recorded at all)
        grad = any(ctx.needs_input_grad) and not no_grad
        head, up = hw is not None, uw is not None
        if (head or up) and (two or pool or (head and up)):
            raise ValueError("ConvBlockFn: a head / ConvTranspose2d tail consumes the un-pooled output of a block with one consumer")
        lazy = head and grad and train
        out, saved = ConvBlockFn._forward(ctx, dt, x0, x1, w1, b1, g1, be1, w2, b2, g2, be2, rm1, rv1, nbt1, rm2, rv2, nbt2,
                                          train, pool, grad, lazy)
        ctx.head = ctx.lazy = ctx.up = False
        if up:
            y = _convT_fwd(dt, out, uw, ub)
            if grad:
                ctx.up, ctx.ubias, ctx.nsaved = True, ub, len(saved)
                saved = saved + (uw, out)
        if head:
            if lazy:  # saved[6] = raw2; scale / shift of its BatchNorm from this forward's statistics
                logits = _head_fwd(dt, saved[6], hw, hb, ConvBlockFn._bn2(ctx, saved))
            else:
                logits = _head_fwd(dt, out, hw, hb)
            if grad:
                ctx.head, ctx.lazy, ctx.hbias, ctx.nsaved = True, lazy, hb, len(saved)
                saved = saved + (hw,) + (() if lazy else (out,))
        if grad:
            ctx.save_for_backward(*saved)
        if up:
            return y
        if head:
            return logits
        if two:
            ctx.set_materialize_grads(False)  # an unused alias hands None to backward, not a tensor of zeros
            return out, out.detach()
        return out

    @staticmethod
    def _bn2(ctx, saved):
        """(mean, invstd, scale, shift) of the second layer's BatchNorm in this forward"""
        if ctx.blk is not None:
            bnv = saved[9]
            C = bnv.numel() // 8
            return tuple(bnv[(4 + i) * C:(5 + i) * C] for i in range(4))
        bn = ctx.bn2
        return bn.mean, bn.invstd, bn.scale, bn.shift

    @staticmethod
    def _forward(ctx, dt, x0, x1, w1, b1, g1, be1, w2, b2, g2, be2, rm1, rv1, nbt1, rm2, rv2, nbt2, train, pool, grad, lazy):
        inf = not grad and not train and not _NO_FUSED_INFERENCE  # nothing saved, no backward: fused inference kernels
        ctx.blk = None
        if PROFILE is None and not inf and not _NO_BLOCK_CALLS:  # one C call for the whole block (host cost)
            return _block_forward(ctx, dt, x0, x1, w1, b1, g1, be1, w2, b2, g2, be2, rm1, rv1, nbt1, rm2, rv2, nbt2, train,
                                  pool, grad, lazy)
        # the block call's rule (csrc/block.hip, bn_on_load): first BatchNorm + ReLU in the second convolution's load path
        B, _, H, W = x0.shape
        C, c0, c1 = w1.shape[0], x0.shape[1], (x1.shape[1] if x1 is not None else 0)
        on_load = bool(_PEROP_ON_LOAD and train and not inf and L.conv3_bnrelu_in_applies(dt, C, C, B, H, W)
                       and L.conv_wgrad_bnrelu_p_applies(dt, C, C, B, H, W)
                       and not L.conv_wgrad_pair_applies(dt, c0, c1, C, C, B, H, W))
        raw1, a1, bn1, wp1t = _conv_bn_relu(dt, x0, x1, w1, b1, g1, be1, rm1, rv1, nbt1, train, False, grad, inf, lazy=on_load)
        if on_load:  # (a1 is None: never written, never read)
            raw2, out, bn2, wp2t = _conv_bn_relu(dt, raw1, None, w2, b2, g2, be2, rm2, rv2, nbt2, train, pool, grad, inf, lazy,
                                                 in_bn=bn1)
        else:
            raw2, out, bn2, wp2t = _conv_bn_relu(dt, a1, None, w2, b2, g2, be2, rm2, rv2, nbt2, train, pool, grad, inf, lazy)
        if inf or not grad:
            return out, ()
        ctx.bn1, ctx.bn2, ctx.train, ctx.pool, ctx.dt, ctx.on_load = bn1, bn2, train, pool, dt, on_load
        ctx.small = (b1, g1, be1, b2, g2, be2)  # leaf parameters: only their gradient destinations are needed
        return out, (x0, x1, w1, w2, raw1, a1, raw2, wp1t, wp2t)

    @staticmethod
    def backward(ctx, dout, dout2=None):
        if dout is None:
            dout, dout2 = dout2, None
        if dout is None:
            return (None,) * 24
        tail = (None,) * 10
        reduced = None
        up_grads = (None, None)
        if ctx.up:  # dout = d(up-sampled tensor): through the ConvTranspose2d first
            saved = ctx.saved_tensors
            uw, out = saved[ctx.nsaved], saved[ctx.nsaved + 1]
            raw2 = saved[6]
            B, C, H, W = raw2.shape
            partial = _f32(_block_partial_rows(ctx.dt, B, C, H, W, False, uw.shape[1]) * 2 * C, raw2.device)
            dout, duw, dub, rows = _convT_bwd(ctx.dt, out, uw, ctx.ubias, dout, True, raw2, ConvBlockFn._bn2(ctx, saved), partial)
            if rows:
                reduced = (partial, rows)
            up_grads = (duw, dub)
        if ctx.head:  # dout = d(logits): through the head first
            saved = ctx.saved_tensors
            hw = saved[ctx.nsaved]
            raw2 = saved[6]
            B, C, H, W = raw2.shape
            if ctx.lazy:
                partial = _f32(_block_partial_rows(ctx.dt, B, C, H, W, False) * 2 * C, raw2.device)
                dout, dhw, dhb = _head_bwd(ctx.dt, raw2, dout, hw, ctx.hbias, ConvBlockFn._bn2(ctx, saved), partial)
                reduced = (partial, L.head_bwd_blocks(B, H, W))
            else:
                dout, dhw, dhb = _head_bwd(ctx.dt, saved[ctx.nsaved + 1], dout, hw, ctx.hbias)
            tail = tail + (dhw, dhb)
        else:
            tail = tail + (None, None)
        tail = tail + up_grads
        if ctx.blk is not None:
            return _block_backward(ctx, dout, dout2, reduced) + tail
        x0, x1, w1, w2, raw1, a1, raw2, wp1t, wp2t = ctx.saved_tensors[:9]
        dt, train, pool = ctx.dt, ctx.train, ctx.pool
        B, C, H, W = raw2.shape
        dev = raw2.device
        s = _stream()
        dout = as_nhwc(dout, raw2.dtype)
        dout2 = as_nhwc(dout2, raw2.dtype) if dout2 is not None else None
        b1, g1, be1, b2, g2, be2 = ctx.small
        c0 = x0.shape[1]
        c1 = x1.shape[1] if x1 is not None else 0
        # same launch sequence as hipseg_convblock_backward (csrc/block.hip): both weight gradients as one paired launch
        # and the first layer's BatchNorm-backward sums out of the data-gradient epilogue, where the shapes allow
        pair = bool(L.conv_wgrad_pair_applies(dt, c0, c1, C, C, B, H, W))
        # ---- second conv layer
        draw2, dg2, dbe2, db2 = _bn_relu_bwd(dt, dout, raw2, ctx.bn2, train, pool, b2, g2, be2, reduced, dy2=dout2)
        dw2 = grad_out(w2)
        if ctx.on_load:  # (a1 was never written: the weight gradient transforms raw1 on load)
            slabs = _f32(L.wgrad_workspace_elems(L.CONV3, C, C, B, H, W), dev)
            _timed("conv_wgrad<bf16,CONV3>(+reduce)", 2.0 * B * H * W * C * C * 9, L.conv_wgrad_bnrelu_p, dt, ptr(raw1), C,
                   ptr(ctx.bn1.scale), ptr(ctx.bn1.shift), ptr(draw2), C, ptr(dw2), ptr(slabs), B, H, W, s,
                   nbytes=B * H * W * 2 * C * _esz(dt) + 9 * C * C * 4)
        elif not pair:
            _wgrad(dt, L.CONV3, a1, None, draw2, dw2, B, H, W)
        if wp2t is None:
            wp2t = _pack_conv(w2, dt, True)
        da1 = nhwc_empty(B, C, H, W, raw2.dtype, dev)
        rows = L.conv3_dgrad_bnstats_rows(dt, C, C, B, H, W)
        # (the same rule as hipseg_convblock_backward: the fused epilogue only when its rows fit the `partial` workspace
        # as include/hipseg.h sizes it -- the reduce kernels' block counts)
        if rows > max(L.bn_bwd_blocks(B, H, W, C, dt, 0), L.bn_bwd_blocks(B, H, W, C, dt, 1) if pool else 0):
            rows = 0
        reduced = None
        if rows:
            partial = _f32(rows * 2 * C, dev)
            bn1 = ctx.bn1
            # (its own timing group: the epilogue adds a read of raw1 per output tile -- the time of the BatchNorm
            # reduce launch it replaces -- at the same algorithmic FLOPs as the plain data gradient)
            _timed("conv_igemm<bf16,CONV3,BN128>+bn_bwd_sums", 2.0 * B * H * W * C * C * 9,
                   L.conv3_dgrad_bnstats, dt, ptr(draw2), C, ptr(wp2t), ptr(da1), C, ptr(raw1), ptr(bn1.mean), ptr(partial),
                   B, H, W, s)
            reduced = (partial, rows)
        else:
            igemm(dt, L.CONV3, draw2, C, None, 0, wp2t, None, da1, C, None, 0, None, B, H, W)
        # ---- first conv layer
        draw1, dg1, dbe1, db1 = _bn_relu_bwd(dt, da1, raw1, ctx.bn1, train, False, b1, g1, be1, reduced)
        dw1 = grad_out(w1)
        if pair:
            _wgrad_pair(dt, x0, x1, draw1, dw1, a1, draw2, dw2, B, H, W)
        else:
            _wgrad(dt, L.CONV3, x0, x1, draw1, dw1, B, H, W)
        dx0 = dx1 = None
        need0 = ctx.needs_input_grad[0]
        need1 = x1 is not None and ctx.needs_input_grad[1]
        if need0 or need1:
            if wp1t is None:
                wp1t = _pack_conv(w1, dt, True)
            dx0 = nhwc_empty(B, c0, H, W, raw2.dtype, dev)
            dx1 = nhwc_empty(B, c1, H, W, raw2.dtype, dev) if c1 else None
            igemm(dt, L.CONV3, draw1, C, None, 0, wp1t, None, dx0, c0, dx1, c1, None, B, H, W)
        return (dx0, dx1, dw1, db1, dg1, dbe1, dw2, db2, dg2, dbe2) + tail


class ConvT2x2Fn(torch.autograd.Function):
    """nn.ConvTranspose2d(Cin, Cout, kernel_size=2, stride=2) (models/processing_blocks.py:102,128)."""

    @staticmethod
    def forward(ctx, x, w, b):
        dt = _dt(x)
        y = _convT_fwd(dt, x, w, b)
        ctx.save_for_backward(x, w)
        ctx.dt, ctx.bias = dt, b
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dx, dw, db, _ = _convT_bwd(ctx.dt, x, w, ctx.bias, dy, ctx.needs_input_grad[0])
        return dx, dw, db


class Conv1x1Fn(torch.autograd.Function):
    """nn.Conv2d(C0 + C1, Cout, kernel_size=1) on NHWC activations, optionally on cat([x0, x1], dim=1) without building
    the concatenation (models/prompt_segmentation.py:55,88-89: `prompt_fusion` on cat([attention_output,
    prompt_embedding])).  Forward / data gradient = the implicit-GEMM entry point in mode CONV1 (dual source / dual
    destination), weight gradient = hipseg_conv_wgrad in mode CONV1, bias gradient = the column sum."""

    @staticmethod
    def forward(ctx, x0, x1, w, b):
        dt = _dt(x0)
        B, c0, H, W = x0.shape
        c1 = x1.shape[1] if x1 is not None else 0
        cout = w.shape[0]
        if w.shape[1] != c0 + c1 or tuple(w.shape[2:]) != (1, 1):
            raise ValueError(f"Conv1x1Fn: weight {tuple(w.shape)} does not match {c0}+{c1} input channels")
        if x1 is not None and (x1.shape[0] != B or x1.shape[2:] != x0.shape[2:] or x1.dtype != x0.dtype):
            raise ValueError(f"Conv1x1Fn: second source {tuple(x1.shape)} does not match {tuple(x0.shape)}")
        wp = _pack_conv(w, dt, False)
        y = nhwc_empty(B, cout, H, W, x0.dtype, x0.device)
        igemm(dt, L.CONV1, x0, c0, x1, c1, wp, b, y, cout, None, 0, None, B, H, W)
        ctx.save_for_backward(x0, x1, w)
        ctx.dt, ctx.bias = dt, b
        return y

    @staticmethod
    def backward(ctx, dy):
        x0, x1, w = ctx.saved_tensors
        dt = ctx.dt
        B, c0, H, W = x0.shape
        c1 = x1.shape[1] if x1 is not None else 0
        cout = w.shape[0]
        dev = x0.device
        dy = as_nhwc(dy, x0.dtype)
        dw = grad_out(w)
        _wgrad(dt, L.CONV1, x0, x1, dy, dw, B, H, W)
        db = None
        if ctx.bias is not None:
            npix = B * H * W
            part = _f32(L.colsum_blocks(npix, cout, dt) * cout, dev)
            db = grad_out(ctx.bias)
            L.colsum(dt, ptr(dy), npix, cout, ptr(part), ptr(db), _stream())
        dx0 = dx1 = None
        if ctx.needs_input_grad[0] or (x1 is not None and ctx.needs_input_grad[1]):
            wpt = _pack_conv(w, dt, True)
            dx0 = nhwc_empty(B, c0, H, W, x0.dtype, dev)
            dx1 = nhwc_empty(B, c1, H, W, x0.dtype, dev) if c1 else None
            igemm(dt, L.CONV1, dy, cout, None, 0, wpt, None, dx0, c0, dx1, c1, None, B, H, W)
        return dx0, dx1, dw, db


class BilinearFn(torch.autograd.Function):
    """F.interpolate(x, size, mode='bilinear', align_corners=True) (models/processing_blocks.py:107)."""

    @staticmethod
    def forward(ctx, x, Ho, Wo):
        dt = _dt(x)
        B, C, Hi, Wi = x.shape
        y = nhwc_empty(B, C, Ho, Wo, x.dtype, x.device)
        _hbm("bilinear_fwd", B * C * (Hi * Wi + Ho * Wo) * _esz(dt), L.bilinear_fwd, dt, ptr(x), ptr(y), B, Hi, Wi, Ho, Wo, C,
             _stream())
        ctx.geo = (dt, B, C, Hi, Wi, Ho, Wo)
        return y

    @staticmethod
    def backward(ctx, dy):
        dt, B, C, Hi, Wi, Ho, Wo = ctx.geo
        dy = as_nhwc(dy, _tdtype("bf16" if dt == L.BF16 else "fp32"))
        dx = nhwc_empty(B, C, Hi, Wi, dy.dtype, dy.device)
        _hbm("bilinear_bwd", B * C * (Hi * Wi + Ho * Wo) * _esz(dt), L.bilinear_bwd, dt, ptr(dy), ptr(dx), B, Hi, Wi, Ho, Wo, C,
             _stream())
        return dx, None, None


class StemFn(torch.autograd.Function):
    """1x1 stem conv on the NCHW fp32 image -> NHWC activations (models/UNet.py:39,62).
    `two`: return the activations TWICE (the second an alias of the first).  The U-Nets use the stem output both as the
    first encoder block's input and as the last decoder block's skip tensor; given one tensor object, autograd sums
    the two gradients in a pass of its own (read 2, write 1 tensor of the largest activation size) before this
    backward runs.  With one alias per consumer each gradient arrives on its own and hipseg_stem_bwd2 reads both."""

    @staticmethod
    def forward(ctx, x, w, b, prec, two=False):
        B, cin, H, W = x.shape
        cout = w.shape[0]
        td = _tdtype(prec)
        dt = L.BF16 if prec == "bf16" else L.F32
        y = nhwc_empty(B, cout, H, W, td, x.device)
        _hbm("stem_fwd", B * H * W * (cin * 4 + cout * _esz(dt)), L.stem_fwd, dt, ptr(x), ptr(w), ptr(b), ptr(y), B, cin, H, W,
             cout, _stream())
        ctx.save_for_backward(x)
        ctx.dt, ctx.wshape, ctx.params = dt, w.shape, (w, b)
        ctx.set_materialize_grads(False)  # an unused alias hands None to backward, not a tensor of zeros
        return (y, y.detach()) if two else y

    @staticmethod
    def backward(ctx, dy, dy2=None):
        if ctx.needs_input_grad[0]:
            raise NotImplementedError("hipseg: gradient w.r.t. the input image is not part of the training hot path")
        (x,) = ctx.saved_tensors
        B, cin, H, W = x.shape
        cout = ctx.wshape[0]
        td = _tdtype("bf16" if ctx.dt == L.BF16 else "fp32")
        if dy is None:
            dy, dy2 = dy2, None
        if dy is None:
            return None, None, None, None, None
        dy = as_nhwc(dy, td)
        dy2 = as_nhwc(dy2, td) if dy2 is not None else None
        nblk = L.stem_bwd_blocks(B, H, W)
        part = _f32(nblk * (cin + 1) * cout, x.device)
        dw, db = grad_out(ctx.params[0]), grad_out(ctx.params[1])
        _hbm("stem_bwd", B * H * W * (cin * 4 + (2 if dy2 is not None else 1) * cout * _esz(ctx.dt)), L.stem_bwd2, ctx.dt,
             ptr(x), ptr(dy), ptr(dy2), ptr(part), ptr(dw), ptr(db), B, cin, H, W, cout, _stream())
        return None, dw, db, None, None


class HeadFn(torch.autograd.Function):
    """1x1 head conv NHWC -> NCHW fp32 logits (models/UNet.py:55,73)."""

    @staticmethod
    def forward(ctx, x, w, b):
        dt = _dt(x)
        B, cin, H, W = x.shape
        cout = w.shape[0]
        logits = torch.empty((B, cout, H, W), dtype=torch.float32, device=x.device)
        _hbm("head_fwd", B * H * W * (cin * _esz(dt) + cout * 4), L.head_fwd, dt, ptr(x), ptr(w), ptr(b), ptr(logits), B, H, W,
             cin, cout, _stream())
        ctx.save_for_backward(x, w)
        ctx.dt, ctx.bias = dt, b
        return logits

    @staticmethod
    def backward(ctx, dl):
        x, w = ctx.saved_tensors
        B, cin, H, W = x.shape
        cout = w.shape[0]
        dl = dl.float().contiguous()
        nblk = L.head_bwd_blocks(B, H, W)
        part = _f32(nblk * cout * (cin + 1), x.device)
        dx = nhwc_empty(B, cin, H, W, x.dtype, x.device)
        dw, db = grad_out(w), grad_out(ctx.bias)
        _hbm("head_bwd", B * H * W * (2 * cin * _esz(ctx.dt) + cout * 4), L.head_bwd, ctx.dt, ptr(x), ptr(dl), ptr(w), ptr(dx),
             ptr(part), ptr(dw), ptr(db), B, H, W, cin, cout, _stream())
        return dx, dw, db


class CrossEntropyFn(torch.autograd.Function):
    """nn.CrossEntropyLoss()(pred, target) = HybridLoss.forward (models/losses.py:13-15)."""

    @staticmethod
    def forward(ctx, logits, target):
        B, C = logits.shape[0], logits.shape[1]
        HW = logits[0, 0].numel()
        part = _f32(L.loss_blocks(B * HW) * 2, logits.device)
        loss = _f32(2, logits.device)
        _hbm("ce_fwd", B * HW * (C * 4 + 8), L.ce_fwd, ptr(logits), ptr(target), ptr(part), ptr(loss), B, C, HW, _stream())
        ctx.save_for_backward(logits, target, loss)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        logits, target, loss = ctx.saved_tensors
        B, C = logits.shape[0], logits.shape[1]
        HW = logits[0, 0].numel()
        gs = g.reshape(1).float().contiguous()
        dl = torch.empty_like(logits)
        _hbm("ce_bwd", B * HW * (2 * C * 4 + 8), L.ce_bwd, ptr(logits), ptr(target), ptr(gs), ptr(loss), ptr(dl), B, C, HW,
             _stream())
        return dl, None


class BceDiceFn(torch.autograd.Function):
    """BCEWithLogitsLoss + smp DiceLoss(binary) on sigmoid(pred) = HybridLossBinary.forward
    (models/losses.py:24-36)."""

    @staticmethod
    def forward(ctx, logits, target):
        n = logits.numel()
        part = _f32(L.loss_blocks(n) * 4, logits.device)
        sums = _f32(4, logits.device)
        loss = _f32(1, logits.device)
        L.bce_dice_fwd(ptr(logits), ptr(target), ptr(part), ptr(sums), ptr(loss), n, _stream())
        ctx.save_for_backward(logits, target, sums)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        logits, target, sums = ctx.saved_tensors
        gs = g.reshape(1).float().contiguous()
        dl = torch.empty_like(logits)
        L.bce_dice_bwd(ptr(logits), ptr(target), ptr(sums), ptr(gs), ptr(dl), logits.numel(), _stream())
        return dl, None


def confusion_matrix(logits, target):
    """CxC int64 counts conf[t, p] of argmax(logits) vs target (metrics, models/losses.py:38-63,129-154)."""
    _require_gpu(logits)
    logits = logits.float().contiguous()
    target = target.long().contiguous()
    B, C = logits.shape[0], logits.shape[1]
    conf = torch.empty((C, C), dtype=torch.int64, device=logits.device)
    L.confusion(ptr(logits), ptr(target), ptr(conf), B, C, logits[0, 0].numel(), _stream())
    return conf


def augment(images, masks, extra, params, order):
    """hipseg_augment on the current stream: flip + nearest rotation on image||mask[||extra], colour jitter + 5x5
    Gaussian blur on the image (models/processing_blocks.py:344-384).  images (B,3,H,W) fp32, masks (B,H,W) int64 or
    None, extra (B,E,H,W) fp32 or None, params (B, AUG_NPARAM) fp32, order int32[4] -- all on the GPU."""
    _require_gpu(images)
    images = images.float().contiguous()
    B, C, H, W = images.shape
    if C != 3:
        raise ValueError(f"augment: expected 3 image channels, got {C}")
    if params.shape != (B, L.AUG_NPARAM) or params.dtype != torch.float32 or order.dtype != torch.int32:
        raise ValueError("augment: params must be (B, AUG_NPARAM) fp32 and order int32[4]")
    dev = images.device
    out = torch.empty_like(images)
    om = oe = None
    ne = 0
    if masks is not None:
        masks = masks.long().contiguous()
        if masks.shape != (B, H, W):
            raise ValueError(f"augment: masks must be (B,H,W), got {tuple(masks.shape)}")
        om = torch.empty_like(masks)
    if extra is not None:
        extra = extra.float().contiguous()
        ne = extra.shape[1]
        oe = torch.empty_like(extra)
    part = _f32(L.augment_workspace_elems(B), dev)
    L.augment(ptr(images), ptr(masks), ptr(extra), ne, ptr(params.contiguous()), ptr(order.contiguous()), ptr(part),
              ptr(out), ptr(om), ptr(oe), B, H, W, _stream())
    return out, om, oe
