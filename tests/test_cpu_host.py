"""CPU-only checks of the host side: the C-ABI library loads and exports every symbol that
include/hipseg.h declares, pure-host geometry functions answer sanely, argument validation fails
loudly, the drop-in modules keep the reference's state_dict layout, the product path refuses CPU
tensors (no CPU fallback), and the N > 1 gradient reducer is correct under gloo with world_size 2."""
import os
import time
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "hipseg.h")


@pytest.fixture(scope="module")
def L():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge

    ge.build()  # no-op when the .so is fresh; hipcc cross-compiles gfx950 without a GPU
    from hipseg import _lib

    return _lib


def header_symbols():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(hipseg_[a-z0-9_A-Z]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(L):
    import ctypes

    lib = ctypes.CDLL(L.LIB_PATH)
    declared = header_symbols()
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/hipseg.h but not exported"
    assert sorted(L.PROTOTYPES) == declared, "ctypes prototypes and header are out of sync"
    out = subprocess.run(["nm", "-D", "--defined-only", L.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (hipseg_\w+)", out))
    assert set(declared) <= exported


def test_host_geometry_functions(L):
    assert L.abi_version() == 1
    assert L.kpad(3, L.BF16) == 16 and L.kpad(32, L.BF16) == 32 and L.kpad(33, L.F32) == 40
    assert L.npad(32) == 32 and L.npad(64) == 64 and L.npad(65) == 128 and L.npad(512) == 512 and L.npad(3) == 32
    assert L.conv_mtiles(16, 256, 256) == 4 * 16 * 16 * 16
    assert L.conv_mtiles(1, 28, 28) == 16
    # rows actually written: the per-64-pixel grid, except for the persistent weights-stationary kernel
    # (bf16 3x3, <= 64 channels, >= 1024 tiles of 8x16 pixels): one row per (workgroup, wave row group)
    assert L.conv_stats_rows(L.F32, L.CONV3, 64, 0, 64, 0, 16, 256, 256) == L.conv_mtiles(16, 256, 256)
    # the 16x16x32-MFMA kernel (N % 128 == 0, K % 32 == 0): one row per workgroup tile -- 16 x 16 pixels when that
    # gives every CU its two workgroups, else 8 x 16
    assert L.conv_stats_rows(L.BF16, L.CONV3, 64, 0, 128, 0, 16, 256, 256) == 16 * 16 * 16
    assert L.conv_stats_rows(L.BF16, L.CONV3, 512, 0, 512, 0, 16, 32, 32) == 16 * 2 * 4
    assert L.conv_stats_rows(L.BF16, L.CONV3, 48, 0, 128, 0, 16, 256, 256) == L.conv_mtiles(16, 256, 256)  # K % 32 != 0
    assert L.conv_stats_rows(L.BF16, L.CONV3, 64, 0, 64, 0, 1, 32, 32) == L.conv_mtiles(1, 32, 32)
    assert L.conv_stats_rows(L.BF16, L.CONV3, 64, 0, 64, 0, 16, 256, 256) == 512 * 2
    assert L.conv_stats_rows(L.BF16, L.CONV3, 32, 32, 32, 0, 16, 256, 256) == 512 * 4
    assert L.wgrad_workspace_elems(L.CONV3, 64, 64, 16, 256, 256) > 0
    assert L.loss_blocks(10) == 1 and L.loss_blocks(10 ** 9) == 1024
    assert L.bn_bwd_blocks(16, 256, 256, 64, L.BF16, 0) >= 1


def test_argument_validation_fails_loudly(L):
    # no GPU work is enqueued: validation happens before any launch
    with pytest.raises(L.HipsegError, match="bad dtype"):
        L.conv_igemm(7, L.CONV3, 1, 8, 0, 0, 1, 0, 1, 8, 0, 0, 0, 1, 16, 16, 0)
    with pytest.raises(L.HipsegError, match="null operand"):
        L.conv_igemm(L.BF16, L.CONV3, 0, 8, 0, 0, 1, 0, 1, 8, 0, 0, 0, 1, 16, 16, 0)
    with pytest.raises(L.HipsegError, match="in1/C1 mismatch"):
        L.conv_igemm(L.BF16, L.CONV3, 1, 8, 0, 8, 1, 0, 1, 8, 0, 0, 0, 1, 16, 16, 0)
    with pytest.raises(L.HipsegError, match="even H, W"):
        L.bn_relu_apply(L.F32, 1, 1, 1, 1, 1, 15, 16, 8, 1, 0)
    with pytest.raises(L.HipsegError, match="unsupported"):
        L.ce_fwd(1, 1, 1, 1, 1, 99, 16, 0)
    with pytest.raises(L.HipsegError, match="out_channels"):
        L.head_fwd(L.F32, 1, 1, 1, 1, 1, 8, 8, 32, 9, 0)
    # round 4: the head over a pre-normalisation tensor, and the ConvTranspose2d data gradient with BatchNorm sums
    with pytest.raises(L.HipsegError, match="bad arguments"):
        L.head_fwd_bnrelu(L.BF16, 1, 0, 1, 1, 1, 1, 1, 8, 8, 32, 3, 0)  # no scale vector
    with pytest.raises(L.HipsegError, match="out_channels"):
        L.head_fwd_bnrelu(L.BF16, 1, 1, 1, 1, 1, 1, 1, 8, 8, 32, 9, 0)
    with pytest.raises(L.HipsegError, match="bad arguments"):
        L.head_bwd_bnrelu(L.BF16, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 1, 8, 8, 32, 3, 0)  # no row buffer for the sums
    with pytest.raises(L.HipsegError, match="null operand"):
        L.convT_dgrad_bnstats(L.BF16, 1, 64, 1, 1, 128, 0, 1, 1, 2, 16, 16, 0)  # no pre-normalisation tensor
    A = L.ConvBlockArgs()
    for f in ("x0", "wp1", "wp2", "raw1", "a1", "raw2", "bn1", "bn2", "stats"):
        setattr(A, f, 1)
    A.dtype, A.B, A.H, A.W, A.C0, A.Cout, A.train, A.pool = L.BF16, 1, 16, 16, 32, 32, 0, 0
    import ctypes

    with pytest.raises(L.HipsegError, match="out may be NULL only in train mode"):
        L.convblock_forward(ctypes.addressof(A), 0)  # eval mode without an output tensor


def test_dropin_modules_keep_reference_state_dict_layout(L, golden):
    from models.CLIP_models import ClipUnet
    from models.UNet import LargeUNet, UNet
    from oracle import torch_ref as R

    for cls, arch in ((UNet, "UNet"), (LargeUNet, "LargeUNet")):
        m = cls()
        sd, ref = m.state_dict(), R.make_state(arch)
        assert list(sd) == list(ref)
        for k in sd:
            assert tuple(sd[k].shape) == tuple(ref[k].shape) and sd[k].dtype == ref[k].dtype, k
    m = ClipUnet(clip_feature_extractor=torch.nn.Identity())
    assert list(m.state_dict()) == list(R.make_state("ClipUnet"))
    # ClipUnetPrompt (models/prompt_segmentation.py:32-95): the key order the reference's own instance reported when the
    # fixture was generated (tests/golden/make_golden.py gen_round3_prompt)
    from models.prompt_segmentation import ClipUnetPrompt, PromptEncoder

    mp = ClipUnetPrompt(clip_feature_extractor=torch.nn.Identity())
    assert list(mp.state_dict()) == [str(k) for k in golden("prompt_r3")["prompt/state_keys"]]
    assert mp.prompt_fusion.weight.shape == (512, 1024, 1, 1) and mp.out.out_channels == 1
    assert [k for k, _ in PromptEncoder().named_children()] == ["enc1", "enc2", "enc3", "conv"]
    # the attributes models/helperFunctions.py:45-78 introspects stay JSON-serialisable
    import json

    for _, mod in UNet().named_modules():
        for a in ("in_channels", "out_channels", "kernel_size", "padding", "num_features"):
            if hasattr(mod, a):
                json.dumps(getattr(mod, a))
    # checkpoints written by the oracle/reference layout load strictly
    from oracle import fill

    m = UNet()
    m.load_state_dict(fill.fill_state_dict(R.make_state("UNet")), strict=True)


def test_product_path_has_no_cpu_fallback(L):
    from models.losses import HybridLoss, IoU
    from models.processing_blocks import ConvBlock
    from models.UNet import UNet

    with pytest.raises(RuntimeError, match="no CPU fallback"):
        UNet()(torch.rand(1, 3, 16, 16))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ConvBlock(4, 8)(torch.rand(1, 4, 8, 8))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        HybridLoss()(torch.rand(1, 3, 8, 8), torch.zeros(1, 8, 8, dtype=torch.long))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        IoU()(torch.rand(1, 3, 8, 8), torch.zeros(1, 8, 8, dtype=torch.long))
    with pytest.raises(ValueError, match="divisible"):
        UNet()(torch.rand(1, 3, 12, 16))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "image-segmentation_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), os.path.join(dp, f)


WORKER = r"""
import os, sys
sys.path[:0] = [{root!r}, os.path.join({root!r}, "image-segmentation_amd")]
import torch, torch.distributed as dist, torch.nn as nn
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
from hipseg.ddp import HipDDP
# ---- defer_comm (the order bench.py --gpus N uses): reducer built, local steps run -- and on a GPU the hipGraphs
# captured -- BEFORE any process group exists; attach() then ties the replicas together in place
assert not dist.is_initialized()
torch.manual_seed(300 + rank)
netd = nn.Sequential(nn.Conv2d(3, 8, 3, padding=1), nn.BatchNorm2d(8), nn.ReLU(), nn.Conv2d(8, 4, 1))
ddpd = HipDDP(netd, overlap="events", defer_comm=True, world_size=world, bucket_cap_mb=0.0001, first_bucket_mb=0.00001)
ptrs = [p.data_ptr() for p in netd.parameters()] + [b.data_ptr() for b in netd.buffers()]
mom = torch.full((5,), float(rank + 1)); cnt = torch.tensor([rank + 3], dtype=torch.int32)   # "optimizer state"
xd = torch.rand(2, 3, 8, 8)
for it in range(2):                                   # local warm-up steps: hooks fire, nothing is reduced
    ddpd.zero_grad(set_to_none=True)
    ddpd(xd).square().mean().backward()
    g_local = [p.grad.detach().clone() for p in netd.parameters()]
    ddpd.broadcast_buffers_now(); ddpd.allreduce_on_events()
    assert ddpd.stats["buckets_reduced"] == 0
    assert all(torch.equal(p.grad, g) for p, g in zip(netd.parameters(), g_local))
    with torch.no_grad():
        for p in netd.parameters():
            p.add_(p.grad, alpha=-0.1 * (rank + 1))   # replicas drift apart
assert sorted(ddpd.ready_order()) == list(range(len(ddpd.buckets)))
try:
    HipDDP(nn.Linear(2, 2), defer_comm=True, world_size=world)        # overlap=True cannot be deferred
    raise SystemExit("defer_comm with overlap=True was accepted")
except ValueError:
    pass
dist.init_process_group("gloo", init_method="env://", rank=rank, world_size=world)
ddpd.attach(extra_state=[mom, cnt])
assert ptrs == [p.data_ptr() for p in netd.parameters()] + [b.data_ptr() for b in netd.buffers()]   # in place
assert torch.equal(mom, torch.full((5,), 1.0)) and int(cnt) == 3      # rank 0's state everywhere
chk = torch.cat([p.detach().reshape(-1) for p in netd.parameters()] + [netd[1].running_mean, netd[1].running_var])
lo, hi = chk.clone(), chk.clone()
dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
assert torch.equal(lo, hi)
try:
    ddpd.attach()
    raise SystemExit("second attach() was accepted")
except RuntimeError:
    pass
torch.manual_seed(7)
xsd = torch.rand(world, 2, 3, 8, 8)
ddpd.zero_grad(set_to_none=True)
ddpd(xsd[rank]).square().mean().backward()
g_local = [p.grad.detach().clone() for p in netd.parameters()]
ddpd.allreduce_on_events()
assert ddpd.stats["buckets_reduced"] == len(ddpd.buckets)
for p, g in zip(netd.parameters(), g_local):
    ref = g.clone(); dist.all_reduce(ref); ref /= world
    assert torch.allclose(p.grad, ref, rtol=1e-6, atol=1e-8)
ddpd.remove_hooks()
# ---- bench.py's replica check: bit-identical replicas pass, ONE diverged rank (or a NaN on one rank) fails on EVERY rank
import bench
ok, gap = bench.replicas_in_sync(list(netd.parameters()), dist)
assert ok and gap == 0.0
with torch.no_grad():
    if rank == world - 1:
        netd[0].weight[0, 0, 0, 0] += 1e-6
ok, gap = bench.replicas_in_sync(list(netd.parameters()), dist)
assert not ok and gap > 0.0, (ok, gap)
with torch.no_grad():
    if rank == world - 1:
        netd[0].weight[0, 0, 0, 0] = float("nan")
ok, gap = bench.replicas_in_sync(list(netd.parameters()), dist)
assert not ok and gap != gap
torch.manual_seed(100 + rank)                      # different init per rank: rank 0 must win
net = nn.Sequential(nn.Conv2d(3, 8, 3, padding=1), nn.BatchNorm2d(8), nn.ReLU(), nn.Conv2d(8, 4, 1))
ddp = HipDDP(net, bucket_cap_mb=0.0001, first_bucket_mb=0.00001)   # tiny caps -> several buckets
assert len(ddp.buckets) >= 3, len(ddp.buckets)
assert list(ddp.state_dict())[0].startswith("module.")
# parameters broadcast from rank 0
ref = [p.detach().clone() for p in net.parameters()]
for p in ref:
    dist.broadcast(p, 0)
assert all(torch.equal(a, b) for a, b in zip(ref, net.parameters()))
torch.manual_seed(7)
xs = torch.rand(world, 2, 3, 8, 8)                 # every rank knows every shard
for it in range(3):
    ddp.zero_grad(set_to_none=(it % 2 == 0))
    ddp(xs[rank]).square().mean().backward()
    mine = [p.grad.detach().clone() for p in net.parameters()]
    # reference: average of the per-rank local gradients (BN statistics stay per replica)
    net2 = nn.Sequential(nn.Conv2d(3, 8, 3, padding=1), nn.BatchNorm2d(8), nn.ReLU(), nn.Conv2d(8, 4, 1))
    acc = None
    for r in range(world):
        net2.load_state_dict(net.state_dict()); net2.train(); net2.zero_grad()
        # running stats were already updated by this iteration's forward on `net`; they do not
        # influence train-mode outputs, so the local gradients are reproducible
        net2(xs[r]).square().mean().backward()
        g = [p.grad.clone() for p in net2.parameters()]
        acc = g if acc is None else [a + b for a, b in zip(acc, g)]
    for a, b in zip(mine, acc):
        assert torch.allclose(a, b / world, rtol=1e-5, atol=1e-7), (it, (a - b / world).abs().max())
    with torch.no_grad():
        for p in net.parameters():
            p.add_(p.grad, alpha=-0.1)
    # replicas stay in lock-step
    chk = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    lo, hi = chk.clone(), chk.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    assert torch.equal(lo, hi)
# BN buffers follow rank 0 at the next training forward
ddp(xs[rank])
rm = net[1].running_mean.clone(); rm0 = rm.clone(); dist.broadcast(rm0, 0)
# (rank 0 broadcast happened BEFORE this forward's update, so after the update ranks differ again;
#  check instead that the flat buffer is shared storage and was synchronised at forward entry)
assert net[1].running_mean.data_ptr() >= ddp._flat_buffers.data_ptr()
with ddp.no_sync():
    ddp.zero_grad(); ddp(xs[rank]).sum().backward()
# explicit (non-overlapped) reduction, the form the hipGraph-captured bench step uses
net3 = nn.Sequential(nn.Conv2d(3, 8, 3, padding=1), nn.BatchNorm2d(8), nn.ReLU(), nn.Conv2d(8, 4, 1))
ddp3 = HipDDP(net3, overlap=False, bucket_cap_mb=0.0001, first_bucket_mb=0.00001)
ddp3.zero_grad(set_to_none=True)
ddp3(xs[rank]).square().mean().backward()
local = [p.grad.detach().clone() for p in net3.parameters()]
ddp3.reduce_gradients()
for p, g in zip(net3.parameters(), local):
    ref = g.clone(); dist.all_reduce(ref); ref /= world
    assert torch.allclose(p.grad, ref, rtol=1e-6, atol=1e-8)
    assert any(p.grad.data_ptr() == v.data_ptr() for b in ddp3.buckets for v in b.views)
# a parameter that takes no part in backward (ClipUnet's dead bottleneck): zero-filled slot, .grad stays None
class Unused(nn.Module):
    def __init__(self):
        super().__init__()
        self.a = nn.Linear(4, 4); self.dead = nn.Linear(4, 4); self.b = nn.Linear(4, 2)
    def forward(self, x):
        return self.b(self.a(x))
for overlap in (True, False):
    torch.manual_seed(5)
    net4 = Unused()
    ddp4 = HipDDP(net4, overlap=overlap, bucket_cap_mb=1.0, first_bucket_mb=1.0)   # one bucket holds all three
    assert len(ddp4.buckets) == 1
    ddp4.buckets[0].flat.fill_(123.0)                # stale garbage must not be reduced
    xr = torch.full((3, 4), float(rank + 1))
    ddp4(xr).sum().backward()
    if not overlap:
        ddp4.reduce_gradients()
    assert net4.dead.weight.grad is None and net4.dead.bias.grad is None
    bi, pi = ddp4._where[net4.dead.weight]
    assert float(ddp4.buckets[bi].views[pi].abs().max()) == 0.0
    exp = sum(float(r + 1) for r in range(world)) / world * 3
    assert torch.allclose(net4.b.bias.grad, torch.full((2,), 3.0)), net4.b.bias.grad
    assert torch.allclose(net4.a.bias.grad, (net4.b.weight.sum(0) * 3).detach())
    ddp4.remove_hooks()
# ---- overlap="events" (what `bench.py --gpus N` runs by default): the control flow on CPU through the host event shim
def mknet():
    return nn.Sequential(nn.Conv2d(3, 8, 3, padding=1), nn.BatchNorm2d(8), nn.ReLU(), nn.Conv2d(8, 4, 1))
torch.manual_seed(11)
net5 = mknet()
ddp5 = HipDDP(net5, overlap="events", bucket_cap_mb=0.0001, first_bucket_mb=0.00001)
assert len(ddp5.buckets) >= 3
for it in range(3):
    ddp5.zero_grad(set_to_none=(it != 1))
    ddp5(xs[rank]).square().mean().backward()
    local = [p.grad.detach().clone() for p in net5.parameters()]       # hooks ran, nothing reduced yet
    order = ddp5.ready_order()
    assert sorted(order) == list(range(len(ddp5.buckets))), order
    seen = [None] * world
    dist.all_gather_object(seen, order)
    assert all(o == order for o in seen), seen                         # identical collective order on every rank
    before = ddp5.stats["buckets_reduced"]
    ddp5.allreduce_on_events()
    assert ddp5.stats["buckets_reduced"] - before == len(ddp5.buckets)
    for p, g in zip(net5.parameters(), local):
        ref = g.clone(); dist.all_reduce(ref); ref /= world
        assert torch.allclose(p.grad, ref, rtol=1e-6, atol=1e-8), it
        assert any(p.grad.data_ptr() == v.data_ptr() for b in ddp5.buckets for v in b.views)
# no_sync(): gradients stay local, nothing is reduced
with ddp5.no_sync():
    ddp5.zero_grad(set_to_none=True)
    ddp5(xs[rank]).square().mean().backward()
    before = ddp5.stats["buckets_reduced"]
    ddp5.allreduce_on_events()
    assert ddp5.stats["buckets_reduced"] == before
# unused parameters in events mode: zero-filled slot, the bucket is still reduced, .grad stays None
torch.manual_seed(5)
net6 = Unused()
ddp6 = HipDDP(net6, overlap="events", bucket_cap_mb=1.0, first_bucket_mb=1.0)
ddp6.buckets[0].flat.fill_(123.0)
ddp6(torch.full((3, 4), float(rank + 1))).sum().backward()
ddp6.allreduce_on_events()
assert net6.dead.weight.grad is None
bi, pi = ddp6._where[net6.dead.weight]
assert float(ddp6.buckets[bi].views[pi].abs().max()) == 0.0
assert torch.allclose(net6.b.bias.grad, torch.full((2,), 3.0))
ddp6.remove_hooks(); ddp5.remove_hooks()
# ---- a backward that RAISES leaves no stale state behind (HipDDP.reset via the new-backward check in the hooks)
for mode in (True, "events"):
    torch.manual_seed(21)
    net7 = mknet()
    ddp7 = HipDDP(net7, overlap=mode, bucket_cap_mb=0.0001, first_bucket_mb=0.00001)
    fired = []
    def boom(p):
        if not fired:
            fired.append(1)
            raise RuntimeError("injected hook failure")
    h = net7[0].weight.register_post_accumulate_grad_hook(boom)   # the LAST gradient of backward: buckets are in flight
    try:
        ddp7(xs[rank]).square().mean().backward()
        raise SystemExit("the injected failure did not propagate")
    except RuntimeError as e:
        assert "injected" in str(e)
    assert ddp7._cb_queued                                         # the end-of-backward callback never ran
    ddp7.zero_grad(set_to_none=True)
    ddp7(xs[rank]).square().mean().backward()
    local = None
    if mode == "events":
        assert sorted(ddp7.ready_order()) == list(range(len(ddp7.buckets)))
        ddp7.allreduce_on_events()
    assert ddp7.stats.get("resets", 0) == 1 and not ddp7._cb_queued
    net8 = mknet(); acc = None
    for r in range(world):
        net8.load_state_dict(net7.state_dict()); net8.train(); net8.zero_grad()
        net8(xs[r]).square().mean().backward()
        g = [p.grad.clone() for p in net8.parameters()]
        acc = g if acc is None else [a + b for a, b in zip(acc, g)]
    for p, b in zip(net7.parameters(), acc):
        assert torch.allclose(p.grad, b / world, rtol=1e-5, atol=1e-7), mode
    h.remove(); ddp7.remove_hooks()
# ---- a module applied twice in one backward: the sum of both partial gradients is what gets averaged
class Twice(nn.Module):
    def __init__(self):
        super().__init__()
        self.f = nn.Linear(4, 4); self.g = nn.Linear(4, 2)
    def forward(self, x):
        return self.g(self.f(torch.relu(self.f(x))))
torch.manual_seed(9)
net9 = Twice(); ddp9 = HipDDP(net9, bucket_cap_mb=0.0001, first_bucket_mb=0.00001)
xr = torch.arange(12.0).reshape(3, 4) / 10 + rank
ddp9(xr).square().sum().backward()
net10 = Twice(); net10.load_state_dict(net9.state_dict()); acc = None
for r in range(world):
    net10.zero_grad(); net10(torch.arange(12.0).reshape(3, 4) / 10 + r).square().sum().backward()
    g = [p.grad.clone() for p in net10.parameters()]
    acc = g if acc is None else [a + b for a, b in zip(acc, g)]
for p, b in zip(net9.parameters(), acc):
    assert torch.allclose(p.grad, b / world, rtol=1e-5, atol=1e-6)
ddp9.remove_hooks()
# ---- ONE autograd node that owns the parameters of several modules (what ops.ConvBlockFn is with a head / a
# ConvTranspose2d tail: block + consumer): all its gradients arrive in one burst, in input order, across several buckets
class FusedFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2):
        h = torch.relu(x @ w1.t() + b1)
        ctx.save_for_backward(x, w1, w2, h)
        return h @ w2.t() + b2
    @staticmethod
    def backward(ctx, dy):
        x, w1, w2, h = ctx.saved_tensors
        dh = (dy @ w2) * (h > 0)
        return dh @ w1, dh.t() @ x, dh.sum(0), dy.t() @ h, dy.sum(0)
class Fused(nn.Module):
    def __init__(self, fused):
        super().__init__()
        self.pre = nn.Linear(4, 4); self.a = nn.Linear(4, 8); self.b = nn.Linear(8, 3); self.fused = fused
    def forward(self, x):
        x = self.pre(x)
        if self.fused:
            return FusedFn.apply(x, self.a.weight, self.a.bias, self.b.weight, self.b.bias)
        return self.b(torch.relu(self.a(x)))
for mode in (True, "events", False):
    torch.manual_seed(13)
    net11 = Fused(True)
    ddp11 = HipDDP(net11, overlap=mode, bucket_cap_mb=0.0001, first_bucket_mb=0.00001)
    assert len(ddp11.buckets) >= 4
    xr = torch.arange(20.0).reshape(5, 4) / 7 - rank
    ddp11.zero_grad(set_to_none=True)
    ddp11(xr).square().mean().backward()
    if mode == "events":
        order = ddp11.ready_order()
        assert sorted(order) == list(range(len(ddp11.buckets))), order
        seen = [None] * world
        dist.all_gather_object(seen, order)
        assert all(o == order for o in seen), seen
        ddp11.allreduce_on_events()
    elif mode is False:
        ddp11.reduce_gradients()
    net12 = Fused(False); net12.load_state_dict(net11.state_dict()); acc = None
    for r in range(world):
        net12.zero_grad(); net12(torch.arange(20.0).reshape(5, 4) / 7 - r).square().mean().backward()
        g = [p.grad.clone() for p in net12.parameters()]
        acc = g if acc is None else [a + b for a, b in zip(acc, g)]
    for (k, p), b in zip(net11.named_parameters(), acc):
        assert torch.allclose(p.grad, b / world, rtol=1e-5, atol=1e-6), (mode, k)
    ddp11.remove_hooks()
dist.barrier(); dist.destroy_process_group()
print("RANK_OK", rank)
"""


@pytest.mark.parametrize("world", [2, 4, 8])
def test_hipddp_gloo_world(tmp_path, world):
    """bucketed gradient averaging + rank-0 broadcast semantics on CPU (gloo) at the world sizes the metric names
    (2, 4, 8 processes; models/model_wrappers.py:964-983): deferred communicator (defer_comm / attach), identical
    ready_order() on every rank, averaged gradients = mean of the per-rank ones, unused-parameter zero fill, no_sync(),
    a raising backward, a module applied twice, and bench.py's replica check with one deliberately diverged rank."""
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29541 + world), WORLD_SIZE=str(world),
               OMP_NUM_THREADS="1" if world > 2 else "2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"RANK_OK {r}" in o, o[-3000:]


def test_off_path_losses_and_perturbations(L, golden):
    """The names the reference's callers import that are OFF the kernel path (plain torch ops, CPU-runnable):
    CombinedConfusionLoss against the reference's own outputs (tests/golden/losses_r2.npz), Dice / DiceBinary against
    hand-derived known answers (smp 0.4.0 absent: parity unpinned), the robustness perturbations against their
    definitions (models/processing_blocks.py:454-592)."""
    import random

    from models import losses as ls, processing_blocks as pb
    from oracle import fill

    g = golden("losses_r2")
    logits = torch.from_numpy(fill.uniform("loss.logits", (2, 3, 32, 32), -3.0, 3.0))
    tgt = torch.from_numpy(fill.randint("loss.t", (2, 32, 32), 3))
    for tag, kw in (("default", {}), ("pairs", {"incorrect_penalty": 1.5, "confusion_pairs": [(0, 1), (1, 2)],
                                                "confusion_penalty": 3.0})):
        lg = logits.clone().requires_grad_(True)
        v = ls.CombinedConfusionLoss(**kw)(lg, tgt)
        v.backward()
        assert abs(float(v) - float(g[f"ccl_{tag}"])) <= 1e-6
        np.testing.assert_allclose(lg.grad.numpy(), g[f"ccl_{tag}_grad"], rtol=1e-5, atol=1e-9)
    # Dice: uniform logits -> softmax(softmax) = 1/3; target all class 0 -> dice_0 = (2/3)/(4/3), absent classes
    # contribute 0 loss -> score = 1 - mean([1/2, 0, 0]) = 5/6
    z = torch.zeros(2, 3, 4, 4)
    assert abs(float(ls.Dice()(z, torch.zeros(2, 4, 4, dtype=torch.long))) - 5.0 / 6.0) < 1e-6
    # DiceBinary: zero logits, all-ones target: p = sigmoid(sigmoid(0)) = sigmoid(.5); score = 2p/(p+1)
    s = 1 / (1 + np.exp(-0.5))
    assert abs(float(ls.DiceBinary()(torch.zeros(2, 1, 4, 4), torch.ones(2, 4, 4))) - 2 * s / (s + 1)) < 1e-6
    assert float(ls.DiceBinary()(torch.zeros(2, 1, 4, 4), torch.zeros(2, 4, 4))) == 1.0  # empty target: loss masked
    # perturbations
    img = torch.rand(2, 3, 12, 12)
    assert torch.equal(pb.ContrastChange(1.5)(img), (img * 1.5).clamp(0, 1))
    assert torch.equal(pb.BrightnessChange(30)(img), (img + 30 / 255.0).clamp(0, 1))
    n = pb.GaussianPixelNoise(10)(img)
    assert n.shape == img.shape and float(n.min()) >= 0 and float(n.max()) <= 1 and not torch.equal(n, img)
    blur = pb.RepeatedBlur(1)(img)
    assert abs(float(blur[0, 0, 5, 5]) - float(img[0, 0, 4:7, 4:7].mean())) < 1e-6
    corner = (img[0, 0, 0, 0] + 2 * img[0, 0, 0, 1] + 2 * img[0, 0, 1, 0] + 4 * img[0, 0, 1, 1]) / 9  # reflect border
    assert abs(float(blur[0, 0, 0, 0]) - float(corner)) < 1e-6
    assert pb.RepeatedBlur(3)(img).shape == img.shape
    random.seed(4)
    occ = pb.Occlusion(5)(img.clone())
    assert all(int((occ[i] == 0).all(0).sum()) >= 25 for i in range(2))
    sp = pb.SaltAndPepper(0.5)(img)
    frac = float(((sp == 1).all(1) | (sp == 0).all(1)).float().mean())
    assert 0.3 < frac < 0.7
    with pytest.raises(ImportError, match="torchvision"):
        pb.ResNet34FeatureExtractor()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        pb.DataAugmentor(4)(img, torch.zeros(2, 12, 12, dtype=torch.long))


def test_augment_oracle_exact_parts():
    """oracle/augment.py (the checker of the HIP augmentation kernels; kornia arithmetic, PARITY UNPINNED): the parts
    that have exact answers -- identity parameters, pure flip, kept samples, a 90-degree turn, hue full-turn,
    normalised blur taps."""
    import math

    from oracle import augment as A

    torch.manual_seed(0)
    B, H, W = 4, 12, 12
    img = torch.rand(B, 3, H, W)
    msk = torch.randint(0, 3, (B, H, W))
    p = torch.zeros(B, 16)
    p[:, 2] = 1.0                      # cos = 1: no rotation
    p[:, 4:7] = 1.0                    # unit colour factors
    p[:, 8] = 1e-3                     # sigma -> 0: blur is the identity
    p[0, 0] = 1.0                      # sample 0 kept
    p[1, 1] = 1.0                      # sample 1 flipped
    p[2, 2], p[2, 3] = 0.0, 1.0        # sample 2 turned by 90 degrees
    p[3, 7] = 2 * math.pi              # sample 3: hue shifted by a full turn
    out, om, _ = A.augment(img, msk, None, p, [0, 1, 2, 3])
    assert torch.equal(out[0], img[0]) and torch.equal(om[0], msk[0])
    np.testing.assert_allclose(out[1].numpy(), img[1].flip(-1).numpy(), atol=1e-6)
    assert torch.equal(om[1], msk[1].flip(-1))
    turned = {tuple(torch.rot90(msk[2], k).reshape(-1).tolist()) for k in (1, 3)}
    assert tuple(om[2].reshape(-1).tolist()) in turned  # a quarter turn (direction = kornia's sign convention)
    np.testing.assert_allclose(out[3].numpy(), img[3].numpy(), atol=2e-6)
    # a real blur preserves the mean of a constant image and the label set is preserved by the geometry
    p[1, 8] = 1.5
    out2, om2, _ = A.augment(torch.full((B, 3, H, W), 0.25), msk, None, p, [3, 2, 1, 0])
    np.testing.assert_allclose(out2[1].numpy(), 0.25, atol=1e-6)
    assert set(om2.unique().tolist()) <= {0, 1, 2}


# ----------------------------------------------------------------------------- bench.py supervisor (loop ladder)
def _bench():
    import importlib

    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    return importlib.import_module("bench")


def test_bench_ladder_order():
    b = _bench()
    assert b.ladder_for("auto", 8) == ["evgraph", "eager", "splitgraph"]
    assert b.ladder_for("auto", 1) == ["graph", "eager"]
    assert b.ladder_for("auto", 1, force_ddp=True) == ["evgraph", "eager", "splitgraph"]
    assert b.ladder_for("eager", 2) == ["eager", "splitgraph"]
    assert b.ladder_for("splitgraph", 1) == ["graph", "eager"]


def test_bench_supervisor_walks_the_ladder_with_fake_workers(monkeypatch, capsys):
    """a failing / hanging first loop makes the supervisor start fresh workers on the next loop with a NEW rendezvous
    port; rank 0 relays the JSON line with `fallback_from` (scripts/train_distributed.py:13-23 is what a worker does)"""
    import json

    b = _bench()
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("MASTER_PORT", "29500")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "3", "--loop", "auto"])
    args = b.parse()
    calls = []

    def fake(cmd, env, limit, errp, abort=None):
        calls.append((cmd, env["MASTER_PORT"], env["HIPSEG_BENCH_WORKER"], limit))
        open(errp, "w").write("boom on rank 1\n")
        loop = cmd[cmd.index("--loop") + 1]
        if loop == "evgraph":
            return -9, "", True           # hung: killed at the limit
        if loop == "eager":
            return 4, "", False           # replicas out of sync
        return 0, "RCCL banner\n" + json.dumps({"metric": "m", "value": 1.0}) + "\n", False

    class Solo:  # (this test drives ONE supervisor: the agreement between supervisors has its own test below)
        def port(self, attempt):
            return None

        def report(self, attempt, rc):
            pass

        def peer_failed(self, attempt):
            return False

        def outcome(self, attempt, timeout):
            return None

    assert b.supervise(args, 2, start=fake, agreement=Solo()) == 0
    doc = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    assert [f["loop"] for f in doc["fallback_from"]] == ["evgraph", "eager"]
    assert "timed out" in doc["fallback_from"][0]["why"] and "exit code 4" in doc["fallback_from"][1]["why"]
    assert "boom on rank 1" in doc["fallback_from"][0]["stderr_tail"]
    assert doc["ladder"] == ["evgraph", "eager", "splitgraph"] and doc["value"] == 1.0
    assert [c[0][c[0].index("--loop") + 1] for c in calls] == ["evgraph", "eager", "splitgraph"]
    assert all(c[0].count("--loop") == 1 and "auto" not in c[0] for c in calls)     # the requested loop was replaced
    assert len({c[1] for c in calls}) == 3 and all(c[2] == "1" for c in calls)      # a fresh port per attempt
    # all loops failing -> non-zero, nothing on stdout; a configuration error (exit 2) ends the ladder at once
    assert b.supervise(args, 2, start=lambda *a: (open(a[3], "w").close(), (1, "", False))[1], agreement=Solo()) == 1
    assert capsys.readouterr().out.strip() == ""
    n = []
    assert b.supervise(args, 2, start=lambda *a: (n.append(1), open(a[3], "w").close(), (2, "", False))[2],
                       agreement=Solo()) == 2
    assert len(n) == 1


def test_bench_supervisors_stay_on_the_same_attempt(monkeypatch, tmp_path):
    """the per-rank supervisors share nothing but the file system.  Rank 1's worker dies EARLY in the
    first loop while rank 0's would sit in a collective: rank 0's supervisor must stop its worker at once (not after the
    process-group timeout), and BOTH must start the next loop together, on the SAME freshly probed port -- never one
    attempt apart.  Two supervisors as threads, fake workers, the real Agreement directory."""
    import json
    import threading

    b = _bench()
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("MASTER_PORT", "29500")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "3", "--loop", "auto"])
    args = b.parse()
    log, lock = [], threading.Lock()

    def make_fake(rank):
        def fake(cmd, env, limit, errp, abort=None):
            loop = cmd[cmd.index("--loop") + 1]
            open(errp, "w").write(f"rank {rank} loop {loop}\n")
            with lock:
                log.append((rank, loop, int(env["MASTER_PORT"]), int(env["HIPSEG_BENCH_ATTEMPT"]), time.time()))
            if loop == "evgraph":
                if rank == 1:
                    time.sleep(0.3)
                    return 3, "", False                       # capture failed on this rank only
                t0 = time.time()                              # rank 0: would hang in its next collective
                while time.time() - t0 < 60:
                    if abort is not None and abort():
                        return -15, "", False
                    time.sleep(0.05)
                return -9, "", True
            return 0, (json.dumps({"metric": "m", "value": 2.0}) + "\n") if rank == 0 else "", False
        return fake

    rcs, outs = {}, {}

    def sup(rank):
        ag = b.Agreement(2, rank, directory=str(tmp_path / "rdzv"))
        import contextlib
        import io

        buf = io.StringIO()
        with contextlib.redirect_stdout(buf) if rank == 0 else contextlib.nullcontext():
            rcs[rank] = b.supervise(args, 2, start=make_fake(rank), agreement=ag, rank=rank)
        outs[rank] = buf.getvalue()

    t0 = time.time()
    ths = [threading.Thread(target=sup, args=(r,)) for r in (0, 1)]
    [t.start() for t in ths]
    [t.join(120) for t in ths]
    assert rcs == {0: 0, 1: 0} and time.time() - t0 < 30      # rank 0 did not wait out its 60-s "collective"
    by = {(r, a): (loop, port) for r, loop, port, a, _ in log}
    assert by[(0, 0)][0] == by[(1, 0)][0] == "evgraph" and by[(0, 1)][0] == by[(1, 1)][0] == "eager"
    assert by[(0, 0)][1] == by[(1, 0)][1] and by[(0, 1)][1] == by[(1, 1)][1] and by[(0, 0)][1] != by[(0, 1)][1]
    assert len(log) == 4                                        # two attempts per rank, nobody ran a third
    doc = json.loads(outs[0].strip().splitlines()[-1])
    assert doc["value"] == 2.0 and [f["loop"] for f in doc["fallback_from"]] == ["evgraph"]
    assert "peer" in doc["fallback_from"][0]["why"]


def test_bench_run_child_kills_the_whole_process_group(tmp_path):
    """a hung worker (and the helpers it spawned) is gone after the limit"""
    b = _bench()
    pidfile = tmp_path / "pids"
    code = ("import os,subprocess,sys,time\n"
            "c = subprocess.Popen([sys.executable, '-c', 'import time; time.sleep(300)'])\n"
            f"open({str(pidfile)!r}, 'w').write(f'{{os.getpid()}} {{c.pid}}')\n"
            "print('started', flush=True)\ntime.sleep(300)\n")
    t0 = time.time()
    rc, out, timed_out = b.run_child([sys.executable, "-c", code], dict(os.environ), 3.0, str(tmp_path / "err"))
    assert timed_out and rc != 0 and time.time() - t0 < 30
    for pid in map(int, pidfile.read_text().split()):
        for _ in range(50):
            try:
                os.kill(pid, 0)
            except ProcessLookupError:
                break
            time.sleep(0.1)
        else:
            # (a zombie still answers kill(pid, 0): it must at least not be running)
            state = open(f"/proc/{pid}/stat").read().split()[2]
            assert state == "Z", f"process {pid} survived the limit (state {state})"
    rc, out, timed_out = b.run_child([sys.executable, "-c", "print('x'); raise SystemExit(5)"], dict(os.environ), 30.0,
                                     str(tmp_path / "err2"))
    assert (rc, out.strip(), timed_out) == (5, "x", False)


def test_bench_ladder_end_to_end_subprocess(tmp_path):
    """real supervisor + real worker processes on this GPU-less box: the first loop fails by the env switch, the second
    because there is no GPU -- the supervisor reports both and exits non-zero without a JSON line"""
    env = dict(os.environ, HIPSEG_BENCH_FAIL_LOOP="graph", HIPSEG_BENCH_ATTEMPT_TIMEOUT="200")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1"], env=env,
                       capture_output=True, text=True, timeout=400)
    if r.returncode == 0:
        pytest.skip("a GPU is present: the second loop succeeded (covered by the -m gpu rehearsal)")
    assert r.stdout.strip() == ""
    assert "injected failure of loop 'graph'" in r.stderr and "loop 'graph' failed (exit code 3)" in r.stderr
    assert "starting fresh workers with 'eager'" in r.stderr and "ladder exhausted" in r.stderr


def test_bench_two_ranks_under_torchrun_walk_the_ladder_in_lockstep(tmp_path):
    """the driver's N > 1 command line (`python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2`) on this
    GPU-less box: both rank supervisors find each other through the Agreement directory (keyed by the torchrun agent's pid),
    every loop of the ladder fails on both ranks (no GPU), and BOTH supervisors start each next loop together and give up
    together -- three attempts each, no rank left waiting out a rendezvous timeout."""
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the loops would succeed (covered by the -m gpu rehearsal)")
    env = dict(os.environ, HIPSEG_BENCH_ATTEMPT_TIMEOUT="120")
    env.pop("WORLD_SIZE", None)
    t0 = time.time()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", "29733", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
                        "--warmup", "1"], env=env, capture_output=True, text=True, timeout=500)
    assert r.returncode != 0 and time.time() - t0 < 120
    lines = r.stderr.splitlines()
    for rank in (0, 1):
        for loop, nxt in (("evgraph", "starting fresh workers with 'eager'"), ("eager", "starting fresh workers with 'splitgraph'")):
            assert any(f"[bench supervisor rank {rank}] loop '{loop}' failed" in l and nxt in l for l in lines), \
                (rank, loop, r.stderr[-3000:])
    # (torchrun terminates the other rank as soon as the first one exits non-zero: the last line is certain for one rank only)
    assert any("loop 'splitgraph' failed" in l and "ladder exhausted" in l for l in lines), r.stderr[-3000:]
    assert "{\"metric\"" not in r.stdout
