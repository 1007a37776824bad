"""Where do the small torch kernels of a train step come from?  Runs a few steps under torch.profiler (with Python stacks) and prints, for
aten::copy_ / aten::add / aten::add_ / aten::fill_ / aten::zero_ / aten::cat / aten::clone, the package frames that called them."""
import sys, collections
import torch
from torch.profiler import profile, ProfilerActivity
sys.path.insert(0, "/root/repo")
import faoctasr
from faoctasr import _lib
import bench

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
_lib.load()
torch.manual_seed(0)
ts = faoctasr.TrainStep(device=torch.device("cuda", 0), distributed=False, precision=(sys.argv[2] if len(sys.argv) > 2 else "f16x2"))
batch = bench.make_batch(B, 256, torch.device("cuda", 0), 0)
for _ in range(3):
    ts.step(*batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    ts.step(*batch)
torch.cuda.synchronize()
names = ("aten::copy_", "aten::add", "aten::add_", "aten::fill_", "aten::zero_", "aten::cat", "aten::clone", "aten::contiguous", "aten::mul", "aten::sum")
cnt = collections.Counter()
for ev in prof.events():
    if ev.name in names:
        frames = [f for f in (ev.stack or []) if "octa-super-resolution_amd" in f or "faoctasr" in f or "bench.py" in f]
        cnt[(ev.name, frames[0].strip() if frames else "(no package frame)")] += 1
for (name, frame), n in sorted(cnt.items(), key=lambda kv: -kv[1])[:40]:
    print("%4d  %-18s %s" % (n, name, frame[-150:]))
