"""Data-parallel gradient reducer: bucketed all-reduce over RCCL/xGMI on a side HIP stream,
overlapped with backward.  Replaces torch DDP as used by the reference
(scripts/train_distributed.py:35 `DDP(model, device_ids=[rank])`, models/model_wrappers.py:974-978).

Semantics kept from DDP: parameters (and buffers) broadcast from rank 0 at construction; gradients
averaged over ranks every backward; BatchNorm statistics stay per-replica (no SyncBN) with the
buffers re-broadcast from rank 0 at every training forward; `.module` exposes the wrapped model so
`model.module.__class__.__name__` (model_wrappers.py:868) and `module.`-prefixed checkpoints work.

Design for xGMI (point-to-point links, no switch): few, large messages.  Gradients are packed into
flat fp32 buckets in REVERSE registration order (out/dec4 first, stem last -- the order backward
produces them); each bucket is all-reduced as soon as its last gradient has been accumulated, on a
dedicated stream that waits on an event recorded on the compute stream, so only the last (small:
stem/enc1) bucket is exposed after backward.  One process per GPU; no data-path collective other
than this all-reduce.
"""
import os

import torch
import torch.distributed as dist
import torch.nn as nn


_ALIGN = 4  # slot starts are multiples of 4 floats = 16 bytes: vectorised optimiser / reduction kernels stay aligned


def _aligned(n):
    return (n + _ALIGN - 1) // _ALIGN * _ALIGN


class _Bucket:
    __slots__ = ("flat", "params", "views", "offsets", "pending", "work", "arrived", "ext_ev")

    def __init__(self, params, device):
        self.offsets, o = [], 0
        for p in params:
            self.offsets.append(o)
            o += _aligned(p.numel())
        self.flat = torch.zeros(o, dtype=torch.float32, device=device)  # (padding stays zero: reduced, never read)
        self.params = params
        self.views = [self.flat[o:o + p.numel()].view_as(p) for o, p in zip(self.offsets, params)]
        self.pending = len(params)
        self.arrived = [False] * len(params)
        self.work = None
        self.ext_ev = None  # overlap="events": HIP event recorded (as a graph node under capture) when the bucket is complete


class _HipEvents:
    """overlap="events" on a GPU: HIP events; under capture the record is an EXTERNAL event-record graph node."""

    def __init__(self):
        from . import _lib as L

        self._L = L

    def create(self):
        import ctypes

        h = ctypes.c_void_p()
        self._L.event_create(ctypes.byref(h))
        return h.value

    def record(self, ev, device):
        self._L.event_record_external(ev, torch.cuda.current_stream(device).cuda_stream)

    def wait(self, stream, ev):
        self._L.stream_wait_event(stream.cuda_stream, ev)

    def destroy(self, ev):
        self._L.event_destroy(ev)


class _HostEvents:
    """overlap="events" without a GPU (gloo tests of the control flow): there is no asynchrony to order, so an event is
    a token and record / wait do nothing -- allreduce_on_events() then runs the collectives synchronously, in the
    order the buckets became ready."""

    def create(self):
        return object()

    def record(self, ev, device):
        pass

    def wait(self, stream, ev):
        pass

    def destroy(self, ev):
        pass


class HipDDP(nn.Module):
    last_quiesce = None  # how the most recent quiesce_before_capture() fenced ("retired" / "sleep")
    last_capture_attempts = None  # attempts the most recent capture_graphs() needed

    def __init__(self, module, device_ids=None, process_group=None, bucket_cap_mb=25.0, first_bucket_mb=1.0,
                 broadcast_buffers=True, overlap=True, force_collectives=False, grad_in_bucket=True,
                 defer_comm=False, world_size=None):
        """device_ids : accepted for call compatibility with `DDP(model, device_ids=[rank])`
                        (scripts/train_distributed.py:35); the module's own device is used.
        overlap=True : reduce each bucket from autograd hooks during backward on a side HIP stream.  Works in eager
                       loops and inside a hipGraph capture (the side-stream collectives become forked branches of
                       the graph, joined at the end of backward).
        overlap=False: no hooks; the caller runs pack_gradients() / allreduce_packed() after backward
                       (split-graph form: fwd+bwd+pack in one graph, the collective eager, the optimiser in a second).
        overlap="events": for a forward+backward that is REPLAYED as a hipGraph while the collectives stay eager RCCL
                       calls: the hooks only record one HIP event per bucket at the point of backward where the
                       bucket is complete -- under capture an EXTERNAL event-record node of the graph
                       (hipseg_event_record_external) -- and the caller runs allreduce_on_events() right after launching
                       the graph: the communication stream waits for each bucket's event and reduces it while the
                       rest of the replayed backward is still executing.  Overlapped like overlap=True, host cost of a
                       graph launch, and no RCCL call inside any capture.
        force_collectives : issue every collective even when the group has ONE rank (RCCL runs them as device-side
                       no-op/copies), so the whole hook -> bucket -> event -> side-stream all-reduce -> join path can
                       be rehearsed and tested on a single-GPU box.
        grad_in_bucket : let the HIP backward kernels write parameter gradients directly into the bucket slots.
        defer_comm     : build everything that needs NO communicator (buckets, slots, hooks, bucket events, the flat
                       buffer tensor) now and leave the process group to attach(): the step can then be warmed up and
                       its hipGraphs captured (overlap="events" / overlap=False hold no collective) in a process that
                       has no c10d / RCCL thread yet; attach() afterwards does the rank-0 broadcasts of the constructor
                       IN PLACE (captured addresses stay valid).  `world_size` (default: $WORLD_SIZE) tells the hooks
                       whether there will be anything to reduce."""
        super().__init__()
        if defer_comm:
            if overlap is True:
                raise ValueError("defer_comm needs overlap='events' or overlap=False: overlap=True issues its collectives "
                                 "from the backward hooks and cannot run without a process group")
            self.world = int(world_size if world_size is not None else os.environ.get("WORLD_SIZE", "1"))
        else:
            if not dist.is_initialized():
                raise RuntimeError("HipDDP needs an initialised torch.distributed process group (backend 'nccl' = RCCL)")
            self.world = dist.get_world_size(process_group)
        self.module = module
        self.pg = process_group
        self._comm_ready = False
        self.active = self.world > 1 or bool(force_collectives)
        self.stats = {"buckets_reduced": 0, "comm_stream_collectives": 0, "hook_calls": 0, "hook_copies": 0,
                      "zero_filled_slots": 0}
        self.broadcast_buffers = broadcast_buffers
        params = [p for p in module.parameters() if p.requires_grad]
        if not params:
            raise ValueError("module has no trainable parameters")
        self.device = params[0].device
        self.on_gpu = self.device.type == "cuda"
        self._avg = False  # set by attach()
        # ---- buckets, reverse registration order; first bucket small so the reduction starts early
        self.buckets, self._where = [], {}
        cur, cur_bytes, cap = [], 0, first_bucket_mb * 2 ** 20
        for p in reversed(params):
            if p.dtype != torch.float32:
                raise TypeError("HipDDP reduces fp32 master gradients")
            cur.append(p)
            cur_bytes += p.numel() * 4
            if cur_bytes >= cap:
                self.buckets.append(_Bucket(cur, self.device))
                cur, cur_bytes, cap = [], 0, bucket_cap_mb * 2 ** 20
        if cur:
            self.buckets.append(_Bucket(cur, self.device))
        for bi, b in enumerate(self.buckets):
            for pi, p in enumerate(b.params):
                self._where[p] = (bi, pi)
        self.comm_stream = torch.cuda.Stream(device=self.device) if self.on_gpu else None
        self._cb_queued = False
        self._require_sync = True
        with torch.no_grad():
            # floating-point buffers (BN running statistics) are re-pointed into ONE flat tensor so the
            # per-forward DDP buffer broadcast is a single collective with no gather/scatter copies
            fbufs = [b for b in module.buffers() if b.is_floating_point()]
            self._flat_buffers = None
            if fbufs:
                flat = torch.cat([b.reshape(-1).float() for b in fbufs])
                o = 0
                for b in fbufs:
                    b.data = flat[o:o + b.numel()].view_as(b)
                    o += b.numel()
                self._flat_buffers = flat
        self.overlap = overlap
        self.events_mode = overlap == "events"
        self._order_building, self._ready_order = [], []
        self._events = None
        if self.events_mode:
            self._events = _HipEvents() if self.on_gpu else _HostEvents()
            for b in self.buckets:
                b.ext_ev = self._events.create()
        self._params = params
        self._hook_handles = []
        if overlap:
            for p in params:
                self._hook_handles.append(p.register_post_accumulate_grad_hook(self._hook))
        # gradient-as-bucket-view from the source: the HIP backward kernels write each parameter gradient straight into
        # its bucket slot (hipseg.ops.grad_out), so neither the hooks nor pack_gradients() move any data for them
        # A slot is handed out at most ONCE per backward (`_taken`, cleared when a backward ends): a parameter that is
        # used twice in one backward (a shared block, tied weights) gets a fresh tensor for its second partial gradient,
        # so autograd sums two different buffers instead of two aliases of the same memory.
        self._slotted = []
        self._taken = set()
        if self.on_gpu and grad_in_bucket:
            for b in self.buckets:
                for p, o in zip(b.params, b.offsets):
                    p._hipseg_slot = (b.flat, o, self._taken)
                    self._slotted.append(p)
        self._graph_task = None
        if not defer_comm:
            self.attach(process_group)

    def attach(self, process_group=None, extra_state=()):
        """Bind the reducer to the (now initialised) process group and do what DDP's constructor does: parameters and
        buffers take rank 0's values.  Everything is written IN PLACE, so hipGraphs captured before attach() keep
        pointing at live memory.  `extra_state`: further tensors that must start identical on every rank when local
        steps ran before attach() (optimizer moments / step counters, GradScaler scale) -- broadcast the same way."""
        if self._comm_ready:
            raise RuntimeError("HipDDP.attach() called twice")
        if not dist.is_initialized():
            raise RuntimeError("HipDDP needs an initialised torch.distributed process group (backend 'nccl' = RCCL)")
        self.pg = process_group
        # RCCL averages inside the collective; gloo has no AVG (CPU tests; GPU tensors through gloo in the shared-GPU
        # rehearsal of bench.py, where RCCL refuses two ranks on one device): SUM, then divide
        self._avg = self.on_gpu and dist.get_backend(process_group) == "nccl"
        world = dist.get_world_size(process_group)
        if world != self.world:
            raise RuntimeError(f"HipDDP was prepared for world size {self.world}, the process group has {world}")
        with torch.no_grad():
            self._bcast([p.data for p in self.module.parameters()])
            if self._flat_buffers is not None:
                dist.broadcast(self._flat_buffers, 0, group=self.pg)
            for b in self.module.buffers():
                if not b.is_floating_point():
                    dist.broadcast(b, 0, group=self.pg)
            self._bcast([t for t in extra_state if t.is_floating_point()])
            for t in extra_state:
                if not t.is_floating_point():
                    dist.broadcast(t, 0, group=self.pg)
        self._comm_ready = True
        return self

    @staticmethod
    def watchdog_idle(timeout=5.0):
        """True once torch's ProcessGroupNCCL watchdog has RETIRED every collective it was handed (its work list is
        empty), False if that did not happen within `timeout`, None when it cannot be observed.  Read from the
        process group's flight recorder (`_dump_nccl_trace(onlyActive=True)` lists exactly the entries the watchdog
        has not retired yet); needs TORCH_NCCL_TRACE_BUFFER_SIZE > 0 before the group is created
        (`HipDDP.enable_watchdog_trace()`)."""
        import pickle
        import time

        try:
            from torch._C._distributed_c10d import _dump_nccl_trace
        except ImportError:
            return None
        if int(os.environ.get("TORCH_NCCL_TRACE_BUFFER_SIZE", "0") or 0) <= 0:
            return None
        deadline = time.monotonic() + timeout
        while True:
            try:
                doc = pickle.loads(_dump_nccl_trace(includeCollectives=True, includeStackTraces=False, onlyActive=True))
            except Exception:  # noqa: BLE001  (signature / format drift: not observable)
                return None
            if not doc.get("entries"):
                return True
            if time.monotonic() > deadline:
                return False
            time.sleep(0.005)

    @staticmethod
    def enable_watchdog_trace():
        """call BEFORE dist.init_process_group: turns on the flight recorder that watchdog_idle() reads."""
        os.environ.setdefault("TORCH_NCCL_TRACE_BUFFER_SIZE", "512")

    @staticmethod
    def quiesce_before_capture(seconds=0.3):
        """Call before ANY hipGraph capture in a process that has issued eager collectives (with or without collectives
        inside the capture).  torch's ProcessGroupNCCL watchdog thread polls the end events of the EAGER collectives
        issued so far (warm-up steps) until it has seen them complete; a capture pulls the process group's internal RCCL
        stream into capture mode, and HIP then refuses `hipEventQuery` on such an event ("HIP error: operation not
        permitted on an event last recorded in a capturing stream", gpurun_out/a5/pytest.log of round 2) -- the watchdog
        aborts the process; in the default global capture mode its query also invalidates the capture.
        The fence is a CONDITION, not a delay: drain the device (every eager collective has completed), then wait until
        the watchdog has retired every entry of its work list (watchdog_idle(): nothing left for it to query; collectives
        issued DURING capture are never handed to it).  Only when the flight recorder is unavailable does it fall back
        to sleeping `seconds` (three watchdog poll periods).  Returns how it fenced: "retired" / "sleep"."""
        import time

        if torch.cuda.is_available():
            torch.cuda.synchronize()
        idle = HipDDP.watchdog_idle()
        HipDDP.last_quiesce = "retired" if idle else "sleep"
        if not idle:
            time.sleep(seconds)
        return HipDDP.last_quiesce

    @staticmethod
    def graph_capture(graph, stream=None, pool=None):
        """THE capture recipe (bench.py and the tests use this one helper).
        No process group yet (HipDDP(defer_comm=True): the recommended order -- capture first, communicator after):
        the plain `torch.cuda.graph` capture of the single-GPU path; the process has no c10d / RCCL thread.
        With a live process group (graphs that HOLD collectives, overlap=True): quiesce_before_capture(), then capture
        in "thread_local" error mode -- other threads of the process (torch's RCCL watchdog, RCCL's own helpers) may
        call the HIP runtime while we capture; kernels launched by the autograd thread on the capturing stream are
        captured in either mode."""
        kw = {}
        if dist.is_available() and dist.is_initialized():
            HipDDP.quiesce_before_capture()
            kw["capture_error_mode"] = "thread_local"
        else:
            HipDDP.last_quiesce = "no process group yet"
        if stream is not None:
            kw["stream"] = stream
        if pool is not None:
            kw["pool"] = pool
        return torch.cuda.graph(graph, **kw)

    @staticmethod
    def capture_graphs(fns, stream=None, reducer=None, attempts=2):
        """Capture each callable of `fns` into its own hipGraph (later ones allocate from the first one's pool) with
        graph_capture().  Returns (graphs, results); `last_capture_attempts` says how many attempts it took.
        History: with a process group ALIVE, a capture was occasionally invalidated before its FIRST kernel launch
        (hipErrorStreamCaptureInvalidated reported by that launch, ~1 run in 5 of the event-graph test in rounds 2-3,
        with the watchdog's work list observably empty and thread_local capture mode): the invalidation came from a
        thread other than the capturing one, and the only other threads of those processes were c10d's (watchdog,
        heartbeat monitor, store) and RCCL's.  Since round 4 the event-graph / split-graph paths capture BEFORE the
        process group exists (defer_comm), so those threads are not there.  The retry stays as a guard and is LOUD:
        stderr line, `last_capture_attempts` > 1 in the bench JSON, and tests/test_gpu_ddp.py fails on it.
        A retry is only sound for graphs WITHOUT collectives (a rank that re-captured RCCL calls its peers captured
        once would desynchronise the communicator): pass attempts=1 for those."""
        last = None
        for attempt in range(attempts):
            graphs, outs, pool = [], [], None
            try:
                for fn in fns:
                    g = torch.cuda.CUDAGraph()
                    with HipDDP.graph_capture(g, stream=stream, pool=pool):
                        outs.append(fn())
                    graphs.append(g)
                    pool = g.pool()
                HipDDP.last_capture_attempts = attempt + 1
                return graphs, outs
            except Exception as e:  # noqa: BLE001
                last = e
                import sys

                print(f"[HipDDP] hipGraph capture attempt {attempt + 1}/{attempts} FAILED: {e!r}"[:400], file=sys.stderr,
                      flush=True)
                graphs = outs = None
                try:
                    torch.cuda.synchronize()
                except Exception:  # noqa: BLE001
                    pass
                if reducer is not None:
                    reducer.reset()
        HipDDP.last_capture_attempts = attempts
        raise last

    def remove_hooks(self):
        """detach this reducer from the module's parameters (before wrapping the same module again)."""
        for h in self._hook_handles:
            h.remove()
        self._hook_handles = []
        for p in self._slotted:
            slot = getattr(p, "_hipseg_slot", None)
            if slot is not None and any(slot[0] is b.flat for b in self.buckets):  # (a later reducer may own it now)
                del p._hipseg_slot
        self._slotted = []

    # ------------------------------------------------------------------ helpers
    def _bcast(self, tensors):
        if not tensors:
            return
        flat = torch.cat([t.reshape(-1).float() for t in tensors])
        dist.broadcast(flat, 0, group=self.pg)
        o = 0
        for t in tensors:
            t.copy_(flat[o:o + t.numel()].view_as(t))
            o += t.numel()

    def no_sync(self):
        """context manager: accumulate local gradients without reducing (gradient accumulation)."""
        ddp = self

        class _Ctx:
            def __enter__(self_):
                ddp._require_sync = False

            def __exit__(self_, *a):
                ddp._require_sync = True

        return _Ctx()

    def reset(self):
        """Re-arm the per-backward state.  Autograd does not run the end-of-backward callback when backward RAISES (a
        hook failing during a capture, an OOM): `_finalize` is then never called, the buckets stay half-counted and
        `_cb_queued` stays set, so a later backward would never be reduced.  Call this after a failed backward / capture
        before using the reducer again (the hooks also call it when they see a new backward with stale state)."""
        if self.on_gpu:
            torch.cuda.current_stream(self.device).wait_stream(self.comm_stream)
        for b in self.buckets:
            if b.work is not None and not self.on_gpu:
                try:
                    b.work.wait()
                except Exception:  # noqa: BLE001
                    pass
            b.work = None
            b.pending = len(b.params)
            b.arrived = [False] * len(b.params)
        self._order_building = []
        self._cb_queued = False
        self._taken.clear()
        self.stats["resets"] = self.stats.get("resets", 0) + 1

    # ------------------------------------------------------------------ backward side
    def _hook(self, p):
        if not self._require_sync or not self.active:
            return
        task = torch._C._current_graph_task_id()
        if task != self._graph_task:  # first hook of a new backward
            if self._cb_queued:  # ... but the previous one never reached _finalize (it raised): stale counters
                self.reset()
            self._graph_task = task
        self.stats["hook_calls"] += 1
        bi, pi = self._where[p]
        b = self.buckets[bi]
        view = b.views[pi]
        if p.grad.data_ptr() != view.data_ptr():
            view.copy_(p.grad)  # (gradients not produced by the HIP kernels, or accumulated ones)
            self.stats["hook_copies"] += 1
            p.grad = view  # gradient-as-bucket-view: the optimiser reads the reduced values in place
        if b.arrived[pi]:  # a parameter used twice in one backward fires once per accumulation pass; count it once
            return
        b.arrived[pi] = True
        b.pending -= 1
        if not self._cb_queued:
            torch.autograd.Variable._execution_engine.queue_callback(self._finalize)
            self._cb_queued = True
        if b.pending == 0:
            self._launch(b)

    def _launch(self, b):
        if not self.events_mode and not self._comm_ready:
            raise RuntimeError("HipDDP(defer_comm=True): attach() the process group before a backward that reduces")
        if self.events_mode:  # mark the point; allreduce_on_events() issues the collective behind it
            self._events.record(b.ext_ev, self.device)
            self._order_building.append(b)
            self.stats["event_records"] = self.stats.get("event_records", 0) + 1
            return
        self.stats["buckets_reduced"] += 1
        if self.on_gpu:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            self.comm_stream.wait_event(ev)
            with torch.cuda.stream(self.comm_stream):
                if self._avg:
                    b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.AVG, group=self.pg, async_op=True)
                else:
                    dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.pg)
                    b.flat.div_(self.world)
                    b.work = None
            self.stats["comm_stream_collectives"] += 1
        else:  # gloo (CPU tests): no AVG op, no streams
            b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)

    def _finalize(self):
        """end of backward: join the communication stream, re-arm the buckets."""
        for b in self.buckets:
            if b.pending != 0 and b.pending != len(b.params):
                # parameters that took no part in this backward (e.g. ClipUnet's dead bottleneck): their slots are
                # zero-filled so no stale data is reduced, and their .grad stays None, as under torch DDP with
                # find_unused_parameters
                for v, p, a in zip(b.views, b.params, b.arrived):
                    if not a:
                        v.zero_()
                        self.stats["zero_filled_slots"] += 1
                self._launch(b)
            if b.work is not None:
                if self.on_gpu:
                    b.work.wait()  # makes the CURRENT stream wait for the collective (no host block)
                else:
                    b.work.wait()
                    b.flat.div_(self.world)
                b.work = None
            b.pending = len(b.params)
            b.arrived = [False] * len(b.params)
        if self.on_gpu and not self._avg and self._comm_ready and not self.events_mode:
            # (gloo on GPU tensors: the collectives ran synchronously under the communication stream, no work handles)
            torch.cuda.current_stream(self.device).wait_stream(self.comm_stream)
        if self.events_mode:  # the order in which this backward completed its buckets (kept across graph replays)
            self._ready_order, self._order_building = self._order_building, []
        self._cb_queued = False
        self._taken.clear()

    # ------------------------------------------------------------------ explicit (non-overlapped) reduction
    def pack_gradients(self):
        """copy every .grad into its flat bucket slot (multi-tensor copy: a few launches, capturable)."""
        views, grads = [], []
        self._taken.clear()
        for b in self.buckets:
            for v, p in zip(b.views, b.params):
                if p.grad is None:
                    v.zero_()  # unused parameter: contributes zero, keeps .grad None (see use_bucket_grads)
                elif p.grad.data_ptr() != v.data_ptr():
                    views.append(v)
                    grads.append(p.grad)
        self._had_grad = [[p.grad is not None for p in b.params] for b in self.buckets]
        if views:
            torch._foreach_copy_(views, grads)

    def allreduce_packed(self):
        """average the flat buckets over ranks on the current stream and point .grad at the reduced views.
        (Before attach(): a local step -- nothing to average with.)"""
        if self.active and self._comm_ready:
            for b in self.buckets:
                self.stats["buckets_reduced"] += 1
                if self._avg:
                    dist.all_reduce(b.flat, op=dist.ReduceOp.AVG, group=self.pg)
                else:
                    dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.pg)
                    b.flat.div_(self.world)
        self.use_bucket_grads()

    def allreduce_on_events(self):
        """overlap="events": reduce every bucket of the backward just issued (eagerly, or by launching the captured
        graph) on the communication stream, each behind its bucket-complete event, then make the current stream wait
        for the communication stream.  Call it right after backward / graph.replay()."""
        if not self.events_mode:
            raise RuntimeError('allreduce_on_events() belongs to HipDDP(overlap="events")')
        if not self.active or not self._ready_order or not self._require_sync:  # (inside no_sync(): nothing to reduce)
            return
        if not self._comm_ready:  # defer_comm: local warm-up / capture steps before attach()
            return
        comm = self.comm_stream
        for b in self._ready_order:
            self._events.wait(comm, b.ext_ev)
            if self.on_gpu:
                with torch.cuda.stream(comm):
                    if self._avg:
                        dist.all_reduce(b.flat, op=dist.ReduceOp.AVG, group=self.pg)
                    else:
                        dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.pg)
                        b.flat.div_(self.world)
            else:  # gloo: no AVG op, no streams
                dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.pg)
                b.flat.div_(self.world)
            self.stats["buckets_reduced"] += 1
            self.stats["comm_stream_collectives"] += 1
        if self.on_gpu:
            torch.cuda.current_stream(self.device).wait_stream(comm)

    def ready_order(self):
        """bucket indices in the order the last backward completed them (overlap="events"); identical on every rank for
        the same model -- the order in which allreduce_on_events() issues the collectives, which must agree across ranks
        (as torch DDP's bucket order does, scripts/train_distributed.py:35)."""
        return [next(i for i, x in enumerate(self.buckets) if x is b) for b in self._ready_order]

    def __del__(self):
        ev = getattr(self, "_events", None)
        if ev is not None:
            for b in getattr(self, "buckets", []):
                if b.ext_ev:
                    try:
                        ev.destroy(b.ext_ev)
                    except Exception:  # noqa: BLE001  (interpreter shutdown)
                        pass
                    b.ext_ev = None

    def use_bucket_grads(self):
        had = getattr(self, "_had_grad", None)
        for bi, b in enumerate(self.buckets):
            for pi, (p, v) in enumerate(zip(b.params, b.views)):
                if had is None or had[bi][pi]:
                    p.grad = v

    def reduce_gradients(self):
        self.pack_gradients()
        self.allreduce_packed()

    def broadcast_buffers_now(self):
        if self.broadcast_buffers and self.active and self._comm_ready and self._flat_buffers is not None:
            dist.broadcast(self._flat_buffers, 0, group=self.pg)

    # ------------------------------------------------------------------ forward side
    def forward(self, *args, **kwargs):
        self._taken.clear()  # (a new step; inside no_sync() no end-of-backward callback clears it)
        if self.module.training and torch.is_grad_enabled():
            self.broadcast_buffers_now()
        return self.module(*args, **kwargs)
