"""Per-layer comparison of the bf16x3 conv kernels against the exact f32 MFMA kernels on the real layer shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import faoctasr
from faoctasr import ops
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from conv_bench import SHAPES
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
H = int(sys.argv[2]) if len(sys.argv) > 2 else 192
def rel(a, b): return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))
torch.manual_seed(0)
for name, kind, ci, co, k, s, p, extra, div in SHAPES:
    hin = H // div
    x = torch.randn(B, ci, hin, hin, device="cuda")
    w = torch.randn((co, ci, k, k) if kind == "conv" else (ci, co, k, k), device="cuda") * 0.02
    res = {}
    for prec in (0, 2):
        ops.conv_precision = prec
        xg = x.clone().requires_grad_(True)
        y = ops.conv2d(xg, w, None, s, p, bool(extra)) if kind == "conv" else ops.conv_transpose2d(xg, w, None, s, p, extra)
        if prec == 0:
            dy = torch.randn_like(y)
        (gx,) = torch.autograd.grad(y, xg, dy)
        res[prec] = (y.detach(), gx)
    ops.conv_precision = 0
    print("%-24s in %3d  fwd rel %.2e   dgrad rel %.2e" % (name, hin, rel(res[2][0], res[0][0]), rel(res[2][1], res[0][1])))
