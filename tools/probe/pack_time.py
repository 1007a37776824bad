"""Time of the batched weight packing (faoctasr_conv_pack_run) for the generators' and the discriminators' images separately.
usage (GPU box): python tools/probe/pack_time.py [precision]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import faoctasr  # noqa: E402
from faoctasr import ops  # noqa: E402
import bench  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "f16x2"
dev = torch.device("cuda", 0)
ts = faoctasr.TrainStep(device=dev, distributed=False, precision=prec)
a, b = bench.make_batch(8, 256, dev, 0)
ts.step(a, b)                                   # packs inline, marks the images
ops.conv_precision = ops.PRECISIONS[prec]
for e in ops._wpack_cache.values():
    e.touched = True
plans = {"G": ops.PackPlan(ts.opt_G.params, ops.conv_precision)}
for e in ops._wpack_cache.values():
    e.touched = True
plans["D"] = ops.PackPlan(ts.opt_D.params, ops.conv_precision)
for k, p in plans.items():
    mb = sum(e.buf.numel() * 4 for e, _ in p.entries) / 1e6
    for _ in range(2):
        p.run()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5):
        p.run()
    e.record()
    torch.cuda.synchronize()
    print("%s: %d jobs, %d blocks, %.0f MB of images, %.1f us per run" % (k, p.njobs, p.nblocks, mb, s.elapsed_time(e) / 5 * 1e3))
