"""Adam with the reference's hyper-parameter surface, as ONE HIP launch per parameter group.

Drop-in for `torch.optim.Adam` where the reference's wrappers take `optimizer_class` / `optimizer_args`
(models/model_wrappers.py:40-41,124: `optim.Adam`, `{'lr': 0.001, 'weight_decay': 1e-4}`), stepped through
`GradScaler.step` (model_wrappers.py:176,979).  GradScaler hands `found_inf` / `grad_scale` to optimisers that declare
`_step_supports_amp_scaling`; the kernel divides by the scale and skips the whole step on overflow, so no host sync
and the step is hipGraph-capturable (the step counter lives on the device).
`step` also accepts GradScaler's (deprecated but still offered) `grad_scaler=` hand-over: the optimiser then runs the inf
check itself -- one read-only pass (hipseg_grads_nonfinite) instead of torch's unscale pass, which rewrites every gradient
x 1.0 for optimisers that take the scale themselves (31 us + two fills + two copies per U-Net step).  When a torch
release stops offering it, `found_inf` / `grad_scale` arrive as attributes as before and nothing else changes."""
import ctypes
import os
import warnings

import torch

from . import _lib as L
from . import ops as _ops
from .ops import _require_gpu, _stream, ptr


# GradScaler announces on every first call that it will stop passing itself to `step(grad_scaler=)`; this optimiser works
# with and without that hand-over (see the module docstring), so the notice carries no information for its users
warnings.filterwarnings("ignore", message="GradScaler is going to stop passing itself", category=FutureWarning)


class Adam(torch.optim.Optimizer):
    _step_supports_amp_scaling = True  # GradScaler.step passes .grad_scale / .found_inf instead of unscaling

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False, *, maximize=False,
                 foreach=None, capturable=True, differentiable=False, fused=None):
        if amsgrad or maximize or differentiable:
            raise NotImplementedError("hipseg.optim.Adam implements the reference's configuration "
                                      "(amsgrad / maximize / differentiable off)")
        if isinstance(lr, torch.Tensor):
            lr = float(lr)
        if not 0.0 <= lr or not 0.0 <= eps or not 0.0 <= weight_decay or not (0.0 <= betas[0] < 1.0 and 0.0 <= betas[1] < 1.0):
            raise ValueError(f"invalid Adam hyper-parameters: lr={lr} betas={betas} eps={eps} weight_decay={weight_decay}")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        self._hip = {}  # group index -> per-group device state

    # ------------------------------------------------------------------ per-group state
    def _group_state(self, gi, group):
        st = self._hip.get(gi)
        params = [p for p in group["params"]]
        if st is not None and st["params"] is not None and len(st["params"]) == len(params) and \
                all(a is b for a, b in zip(st["params"], params)):
            return st
        for p in params:
            _require_gpu(p)
            if p.dtype != torch.float32 or not p.is_contiguous():
                raise TypeError("hipseg.optim.Adam updates contiguous fp32 parameters")
        dev = params[0].device
        n = sum((p.numel() + 3) // 4 * 4 for p in params)
        # moments live in two flat allocations (16-byte aligned slices); exposed per parameter through self.state
        flat_m, flat_v = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
        o = 0
        for p in params:
            s = self.state[p]
            m, v = flat_m[o:o + p.numel()].view_as(p), flat_v[o:o + p.numel()].view_as(p)
            if "exp_avg" in s:  # (state loaded from a checkpoint)
                m.copy_(s["exp_avg"])
                v.copy_(s["exp_avg_sq"])
            s["exp_avg"], s["exp_avg_sq"] = m, v
            o += (p.numel() + 3) // 4 * 4
        counter = torch.zeros(2, dtype=torch.int32, device=dev)
        counter[0] = getattr(self, "_loaded_steps", {}).pop(gi, 0)  # (restored by load_state_dict)
        st = {"params": params, "flat": (flat_m, flat_v), "counter": counter, "tables": {}}
        self._hip[gi] = st
        return st

    def _table(self, st, active):
        """host descriptor table for the parameters that have a gradient; cached per pointer set (the caching allocator
        hands back the same gradient addresses step after step).  Read by hipseg_adam_step during the call only."""
        key = tuple((p.data_ptr(), p.grad.data_ptr(), self.state[p]["exp_avg"].data_ptr()) for p in active)
        t = st["tables"].get(key)
        if t is not None:
            return t
        if len(st["tables"]) > 16:
            st["tables"].clear()
        host = ctypes.create_string_buffer(len(active) * L.adam_desc_size())
        for i, p in enumerate(active):
            g = p.grad
            if g.dtype != torch.float32 or not g.is_contiguous() or g.device != p.device:
                raise TypeError("hipseg.optim.Adam needs contiguous fp32 gradients on the parameter's device")
            s = self.state[p]
            L.adam_desc_fill(ctypes.addressof(host), i, ptr(p), ptr(g), ptr(s["exp_avg"]), ptr(s["exp_avg_sq"]), p.numel())
        st["tables"][key] = host
        return host

    # ------------------------------------------------------------------ step
    def _amp_from_scaler(self, scaler):
        """GradScaler handed itself over (`grad_scaler=`): do what its `step` would have done before calling us --
        `_check_inf_per_device` when the gradients are still scaled (here: one read-only pass over them), else the
        `found_inf` that `unscale_` recorded -- and return (found_inf, grad_scale) for the kernel."""
        from torch.amp.grad_scaler import OptState

        ost = scaler._per_optimizer_states[id(self)]
        if ost["stage"] is OptState.READY:
            found = None
            for gi, group in enumerate(self.param_groups):
                if not group["params"]:
                    continue
                st = self._group_state(gi, group)
                active = [p for p in st["params"] if p.grad is not None]
                if not active:
                    continue
                if found is None:
                    found = st.get("found")
                    if found is None:
                        found = st["found"] = torch.zeros(1, device=active[0].device)
                    found.zero_()
                L.grads_nonfinite(ctypes.addressof(self._table(st, active)), len(active), ptr(found), _stream())
            if found is None:
                raise AssertionError("No inf checks were recorded for this optimizer.")
            ost["found_inf_per_device"] = {found.device: found}  # (what scaler.update() reads)
            return found, scaler._get_scale_async()
        found = sum(t.to(next(iter(ost["found_inf_per_device"])), non_blocking=True) for t in ost["found_inf_per_device"].values())
        return found, None  # (unscale_() was called: the gradients are already unscaled)

    @torch.no_grad()
    def step(self, closure=None, grad_scaler=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if grad_scaler is not None:
            found_inf, grad_scale = self._amp_from_scaler(grad_scaler)
        else:
            found_inf = getattr(self, "found_inf", None)
            grad_scale = getattr(self, "grad_scale", None)
        for gi, group in enumerate(self.param_groups):
            if not group["params"]:
                continue
            st = self._group_state(gi, group)
            active = [p for p in st["params"] if p.grad is not None]
            if not active:
                continue
            host = self._table(st, active)
            b1, b2 = group["betas"]
            # (4 reads + 3 writes of 4 bytes per parameter: weight, gradient, two moments)
            _ops._hbm("adam_step", 28 * sum(p.numel() for p in active), L.adam_step, ctypes.addressof(host), len(active),
                      ptr(st["counter"]), ptr(found_inf.float() if found_inf is not None else None),
                      ptr(grad_scale.float() if grad_scale is not None else None),
                      float(group["lr"]), float(b1), float(b2), float(group["eps"]), float(group["weight_decay"]),
                      _stream())
            # the kernel wrote the parameters through raw pointers: tell autograd (saved-tensor checks, the packed-weight
            # cache and anything else keyed on ._version must see an in-place update, as after torch.optim.Adam.step)
            torch.autograd.graph.increment_version(active)
        return loss

    def zero_grad(self, set_to_none=True):
        """torch.optim.Optimizer.zero_grad.  The set_to_none form (the default, and what the reference's loops get from
        `optimizer.zero_grad()`, models/model_wrappers.py:166,969) is the first thing an eagerly launched step does
        after the previous step's loss.item() synchronisation: the GPU idles until the first forward kernel is queued,
        so the generic implementation's per-parameter bookkeeping (0.13 ms for the U-Net's 76 parameters) is replaced
        by the plain loop it amounts to."""
        if not set_to_none:
            return super().zero_grad(set_to_none=False)
        for group in self.param_groups:
            for p in group["params"]:
                p.grad = None

    # ------------------------------------------------------------------ checkpointing (torch.optim.Adam's layout)
    def state_dict(self):
        """torch.optim.Adam-compatible: per-parameter `step` (the group's device counter), `exp_avg`, `exp_avg_sq`."""
        for gi, group in enumerate(self.param_groups):
            if gi not in self._hip and gi not in getattr(self, "_loaded_steps", {}):
                continue  # no device counter and nothing loaded: leave whatever `step` entries exist untouched
            n = self.step_count(gi)
            for p in group["params"]:
                if "exp_avg" in self.state.get(p, {}):
                    self.state[p]["step"] = torch.tensor(float(n))
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._hip = {}  # moments are re-homed into fresh flat buffers (and the counter restored) at the next step
        self._loaded_steps = {}
        for gi, group in enumerate(self.param_groups):
            steps = [int(self.state[p]["step"]) for p in group["params"] if "step" in self.state.get(p, {})]
            if steps:
                self._loaded_steps[gi] = max(steps)

    def device_state(self):
        """every device tensor of the optimizer state (flat moments, step counters): what a data-parallel wrapper
        broadcasts from rank 0 when local steps ran before the replicas were tied together (HipDDP.attach)."""
        out = []
        for gi, group in enumerate(self.param_groups):
            if group["params"]:
                st = self._group_state(gi, group)
                out += [st["flat"][0], st["flat"][1], st["counter"]]
        return out

    if os.environ.get("HIPSEG_NO_FUSED_INF_CHECK"):  # A/B switch: GradScaler's own unscale / inf-check pass
        _step_with_scaler = step

        @torch.no_grad()
        def step(self, closure=None):  # noqa: F811  (no `grad_scaler` parameter: GradScaler keeps its own inf check)
            return type(self)._step_with_scaler(self, closure)

    def step_count(self, group=0):
        """number of applied (non-skipped) steps of a group (host sync).  The count is PER GROUP (one device counter),
        not per parameter as in torch.optim.Adam: a parameter whose gradient is None on some steps shares the bias
        correction of its group.  Right after load_state_dict() (device state not rebuilt yet) it is the loaded count."""
        st = self._hip.get(group)
        if st is None:
            return int(getattr(self, "_loaded_steps", {}).get(group, 0))
        return int(st["counter"][0])
