"""How long does the host need to ENQUEUE one train step (no device sync inside)?  Compared with the device time per step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import faoctasr
ts = faoctasr.TrainStep(device="cuda")
g = torch.Generator().manual_seed(0)
a = (torch.rand(8, 1, 256, 256, generator=g) * 2 - 1).cuda()
b = (torch.rand(8, 1, 256, 256, generator=g) * 2 - 1).cuda()
for _ in range(2):
    ts.step(a, b)
torch.cuda.synchronize()
for B in (8, 1):
    aa, bb = a[:B].contiguous(), b[:B].contiguous()
    ts.step(aa, bb); torch.cuda.synchronize()
    t0 = time.perf_counter(); ts.step(aa, bb); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("B=%d: host enqueue %.1f ms, until device idle %.1f ms" % (B, 1e3 * (t1 - t0), 1e3 * (t2 - t0)))
