"""WORKER of tests/test_gpu_ddp.py::test_hipddp_two_ranks_sharing_the_gpu: TWO real ranks on GPU 0 (gloo transport -- RCCL
refuses two ranks on one device), each with its own batch, the real U-Net and the real kernels.  For every reducer mode
(autograd hooks with overlap, external events, packed) the gradients HipDDP leaves must be EXACTLY (g_rank0 + g_rank1) / 2 of
the plain local steps (two addends: the sum has no order), replicas and BatchNorm buffers must start from rank 0's values,
and ready_order() must agree.  Replaces `DDP(model, device_ids=[rank])` (scripts/train_distributed.py:35)."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "image-segmentation_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)


def main():
    from hipseg.ddp import HipDDP
    from models.losses import HybridLoss
    from models.UNet import UNet

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    crit = HybridLoss()
    g = torch.Generator().manual_seed(100 + rank)  # a different batch per rank
    x = torch.rand(2, 3, 64, 64, generator=g).cuda()
    t = torch.randint(0, 3, (2, 64, 64), generator=g).cuda()

    def step(net):
        with torch.autocast("cuda"):
            loss = crit(net(x), t)
        loss.backward()
        return loss

    for mode in (True, "events", False):
        torch.manual_seed(7 + rank)  # replicas start DIFFERENT: the constructor must tie them to rank 0
        model = UNet().cuda().train()
        ddp = HipDDP(model, device_ids=[0], overlap=mode, first_bucket_mb=0.05, bucket_cap_mb=4.0)
        chk = torch.cat([p.detach().reshape(-1) for p in model.parameters()] + [b.detach().float().reshape(-1) for b in model.buffers()])
        lo, hi = chk.clone(), chk.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        assert torch.equal(lo, hi), f"{mode}: replicas differ after construction"
        # expected: the plain local step's gradients, summed over the two ranks, halved
        model.zero_grad(set_to_none=True)
        with ddp.no_sync():
            step(ddp)
        if mode == "events":
            ddp.allreduce_on_events()  # (inside no_sync: nothing is reduced)
        torch.cuda.synchronize()
        local = [p.grad.detach().clone() for p in model.parameters()]
        want = torch.cat([gl.reshape(-1) for gl in local])
        dist.all_reduce(want)
        want /= world
        # the data-parallel step (BatchNorm running statistics moved once more: they do not enter train-mode gradients)
        ddp.zero_grad(set_to_none=True)
        step(ddp)
        if mode == "events":
            order = ddp.ready_order()
            seen = [None] * world
            dist.all_gather_object(seen, order)
            assert all(o == order for o in seen), seen
            ddp.allreduce_on_events()
        elif mode is False:
            ddp.reduce_gradients()
        torch.cuda.synchronize()
        got = torch.cat([p.grad.detach().reshape(-1) for p in model.parameters()])
        assert torch.equal(got, want), f"{mode}: max |diff| {float((got - want).abs().max())}"
        assert float(want.abs().max()) > 0 and not torch.equal(want, torch.cat([gl.reshape(-1) for gl in local]))
        assert ddp.stats["zero_filled_slots"] == 0
        ddp.remove_hooks()
        del ddp, model
    dist.barrier()
    dist.destroy_process_group()
    print("RANK_OK", rank, flush=True)


if __name__ == "__main__":
    main()
