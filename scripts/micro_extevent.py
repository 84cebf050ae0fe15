#!/usr/bin/env python3
"""do EXTERNAL events recorded inside a hipGraph order work on another stream?  (graph: A -> record(ev) -> B;
side stream after the launch: wait(ev) -> read A's result while B still runs).  torch refuses Event(external=True)
on ROCm, so the record / wait go through libhipseg (hipEventRecordWithFlags(hipEventRecordExternal))."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "image-segmentation_amd")]
import torch
from hipseg import _lib as L

n = 1 << 26  # 64 Mi floats = 256 MB
a = torch.zeros(n, device="cuda")
b = torch.zeros(n, device="cuda")
out = torch.zeros(1, device="cuda")
main, side = torch.cuda.Stream(), torch.cuda.Stream()
evp = ctypes.c_void_p()
L.event_create(ctypes.byref(evp))
ev = evp.value


def body():
    for _ in range(10):                           # long prefix: a wait bound to the PREVIOUS replay's record would let the
        b.add_(1.0)                               # side stream read `a` ~0.8 ms before A of this replay has run
    a.add_(1.0)                                   # A
    L.event_record_external(ev, main.cuda_stream)  # external record node (when capturing)
    for _ in range(20):                           # B: long tail
        b.add_(1.0)


torch.cuda.set_stream(main)
body()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=main):
    body()
torch.cuda.synchronize()
a.zero_()
ok, t_side = True, []
for it in range(1, 7):
    t0 = time.perf_counter()
    g.replay()
    L.stream_wait_event(side.cuda_stream, ev)
    with torch.cuda.stream(side):
        out.copy_(a[:1])  # must see A of THIS replay: value == it
        done = torch.cuda.Event()
        done.record()
    done.synchronize()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    v = float(out)
    t_side.append((t1 - t0, t2 - t0))
    ok &= v == float(it)
    print(f"replay {it}: side saw {v} (want {it}); side done after {1e3 * (t1 - t0):.3f} ms, graph after {1e3 * (t2 - t0):.3f} ms", flush=True)
print("ORDER_OK" if ok else "ORDER_BROKEN", "OVERLAP_OK" if all(0.2 * t < s < 0.6 * t for s, t in t_side[1:]) else "NO_OVERLAP")
L.event_destroy(ev)
