"""Does the stride-2 weight gradient read LDS it has not written?  Run it between kernels that leave large values in LDS and compare
with an fp64 reference every time."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import faoctasr
from faoctasr import ops

torch.manual_seed(0)
dev = "cuda"
cases = [("convT c7", "convT", 128, 64, 4, 2, 1, 0, 8, 128), ("conv d2a", "conv", 64, 128, 4, 2, 1, 0, 8, 128), ("conv d2b", "conv", 128, 256, 4, 2, 1, 0, 8, 64),
         ("conv d2c", "conv", 64, 128, 4, 2, 1, 0, 8, 64)]
big = torch.randn(8, 64, 256, 256, device=dev) * 1e4
wbig = torch.randn(64, 64, 3, 3, device=dev)
for name, kind, ci, co, k, s, p, op, B, H in cases:
    x = torch.randn(B, ci, H, H, device=dev)
    if kind == "conv":
        w = torch.randn(co, ci, k, k, device=dev, requires_grad=True)
        f = lambda w: ops.conv2d(x, w, None, s, p, False)
        fr = lambda w: torch.nn.functional.conv2d(x.double(), w, None, s, p)
    else:
        w = torch.randn(ci, co, k, k, device=dev, requires_grad=True)
        f = lambda w: ops.conv_transpose2d(x, w, None, s, p, op)
        fr = lambda w: torch.nn.functional.conv_transpose2d(x.double(), w, None, s, p, op)
    y = f(w)
    dy = torch.randn_like(y)
    wd = w.detach().double().requires_grad_(True)
    (gref,) = torch.autograd.grad(fr(wd), wd, dy.double())
    errs = []
    for i in range(8):
        if i % 2:
            ops.conv2d(big, wbig, None, 1, 1, False)          # dirty the LDS of every CU with 1e4-scale values
            xx = big.clone().requires_grad_(True)
            yy = ops.conv2d(xx, wbig.clone().requires_grad_(True), None, 1, 1, False)
        (g,) = torch.autograd.grad(f(w), w, dy)
        errs.append(float((g.double() - gref).norm() / gref.norm()))
    print(name, ["%.2e" % e for e in errs], flush=True)
