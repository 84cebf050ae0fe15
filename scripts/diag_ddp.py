"""stand-alone replica of tests/test_gpu_ddp.py::test_overlapped_allreduce_captured_in_one_hipgraph that prints the
exception at once (under pytest a failure inside a capture aborts the process from the NCCL watchdog thread first)."""
import os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "image-segmentation_amd"), os.path.join(ROOT, "tests")]
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29549")
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method="env://", rank=0, world_size=1, device_id=torch.device("cuda", 0))
import test_gpu_ddp as T
try:
    T.test_hook_path_matches_plain_backward_bitwise(dist)
    print("test1 ok", flush=True)
    T.test_overlapped_allreduce_captured_in_one_hipgraph(dist)
    print("test2 ok", flush=True)
    T.test_unused_parameters_are_zero_filled_not_stale(dist)
    print("test3 ok", flush=True)
except BaseException:
    traceback.print_exc()
    sys.stderr.flush()
    os._exit(1)
os._exit(0)
