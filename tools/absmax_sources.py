"""Which tensors of an f16x2 train step still take a stand-alone faoctasr_absmax_bits pass (no producer-fused slot), by shape and reason.
usage (GPU box): python tools/absmax_sources.py [batch]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import faoctasr  # noqa: E402
from faoctasr import ops  # noqa: E402
import bench  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ts = faoctasr.TrainStep(device=torch.device("cuda", 0), distributed=False, precision="f16x2")
batch = bench.make_batch(B, 256, torch.device("cuda", 0), 0)
for _ in range(2):
    ts.step(*batch)
ops.absmax_log = {}
ts.step(*batch)
torch.cuda.synchronize()
tot = 0
for (shape, why), n in sorted(ops.absmax_log.items(), key=lambda kv: -kv[1] * (kv[0][0][0] * kv[0][0][1] * kv[0][0][2] * kv[0][0][3])):
    mb = 4 * shape[0] * shape[1] * shape[2] * shape[3] / 1e6
    tot += n * mb
    print("%3d x %-22s %7.1f MB each  (%s)" % (n, shape, mb, why))
print("total %.0f MB read per step by stand-alone passes" % tot)
