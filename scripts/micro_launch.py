#!/usr/bin/env python3
"""cost of a dependent kernel node in a hipGraph replay vs eager launches (tiny kernels; one stream)"""
import time
import torch

x = torch.zeros(64, device="cuda")
big = torch.zeros(16 * 256 * 256 * 32, device="cuda", dtype=torch.bfloat16)  # 67 MB: a BN-apply sized tensor
s = torch.cuda.Stream()
torch.cuda.set_stream(s)


def body(n, tiny=True):
    for _ in range(n):
        if tiny:
            x.add_(1.0)
        else:
            big.add_(1.0)


for tiny in (True, False):
    for n in (50, 200):
        body(n, tiny)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            body(n, tiny)
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            g.replay()
        torch.cuda.synchronize()
        tg = (time.perf_counter() - t0) / 20
        t0 = time.perf_counter()
        for _ in range(20):
            body(n, tiny)
        torch.cuda.synchronize()
        te = (time.perf_counter() - t0) / 20
        print(f"{'tiny' if tiny else '67MB'} n={n}: graph {tg / n * 1e6:.2f} us/node, eager {te / n * 1e6:.2f} us/launch", flush=True)
