"""Time of faoctasr_grad_allreduce on the two gradient arenas at the current world size (world 1: what RCCL does with one rank)."""
import os, sys
import torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", 0)))
dist.init_process_group("nccl", rank=rank, world_size=world)
import faoctasr
from faoctasr.train import GradComm
comm = GradComm(None)
for n in (22_450_000, 44_620_000):
    t = torch.randn(n, device="cuda")
    comm.all_reduce(t); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): comm.all_reduce(t)
    b.record(); torch.cuda.synchronize()
    if rank == 0: print("all-reduce of %.1f MB at world %d: %.3f ms" % (n * 4 / 1e6, world, a.elapsed_time(b) / 10), flush=True)
# does the call block the host while earlier work on the stream is pending?
import time
x = torch.randn(8192, 8192, device="cuda")
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    y = x @ x                      # ~100 ms of queued GPU work
t1 = time.perf_counter()
comm.all_reduce(t)
t2 = time.perf_counter()
torch.cuda.synchronize()
t3 = time.perf_counter()
if rank == 0:
    print("host: enqueue matmuls %.2f ms, all_reduce call %.2f ms, drain %.2f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3), flush=True)
comm.close(); dist.destroy_process_group()
