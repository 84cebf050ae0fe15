"""Build libhipseg.so (HIP kernels + C ABI) in-tree with hipcc for gfx950."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libhipseg.so")
SOURCES = ["pack.hip", "bn.hip", "pointwise.hip", "loss.hip", "records.hip", "augment.hip", "optim.hip", "sync.hip", "conv_igemm.hip", "conv3_m16.hip", "convt_stream.hip", "conv_wgrad.hip", "convt_wgrad.hip", "block.hip"]
# -Wno-inline-asm: the LDS-DMA helpers (csrc/common.h, dma_piece*) name m0 in their clobber lists because they write it;
# clang answers every instantiation with "inline asm clobber list contains reserved registers: m0" (~250 lines per build)
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-value", "-Wno-inline-asm"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    os.makedirs(LIBDIR, exist_ok=True)
    headers = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "conv_args.h"), os.path.join(os.path.dirname(os.path.dirname(HERE)), "include", "hipseg.h")]
    objs, jobs = [], []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(LIBDIR, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [src] + headers):
            jobs.append([_hipcc()] + FLAGS + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + r.stdout + r.stderr)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(LIB, objs):
        run([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
