"""Where does the host spend its time ENQUEUEING a train step?  cProfile over a few batch-1 steps (the step is host-bound there)."""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import faoctasr

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ts = faoctasr.TrainStep(device="cuda", precision=(sys.argv[2] if len(sys.argv) > 2 else "f16x2"))
g = torch.Generator().manual_seed(0)
a = (torch.rand(B, 1, 256, 256, generator=g) * 2 - 1).cuda()
b = (torch.rand(B, 1, 256, 256, generator=g) * 2 - 1).cuda()
for _ in range(3):
    ts.step(a, b)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    ts.step(a, b)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
