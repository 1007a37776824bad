import sys, time, torch
sys.path.insert(0, "/root/repo")
import faoctasr, bench
from faoctasr import ops
dev = torch.device("cuda", 0)
a, b = bench.make_batch(8, 256, dev, 0)
def run(prec, fuse, layout=None):
    ops.fuse_residual_grad = fuse
    if layout:
        faoctasr.TrainStep.stream_layout_f32 = layout
    torch.manual_seed(0)
    ts = faoctasr.TrainStep(device=dev, distributed=False, precision=prec)
    for _ in range(3): ts.step(a, b)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10): ts.step(a, b)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
    del ts
    return dt * 1e3
for rep in range(2):
    for fuse in (True, False):
        for layout in ("001212", "012201"):
            print("f32 fuse=%d layout=%s: %.2f ms" % (fuse, layout, run("f32", fuse, layout)), flush=True)
