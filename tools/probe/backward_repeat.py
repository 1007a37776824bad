"""Backward of the SAME forward graph, repeated: separates run-to-run differences that come from the forward pass (split-K atomics
moving an activation across a LeakyReLU / ReLU kink) from differences born in the backward kernels themselves."""
import os, sys, random
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import faoctasr
from faoctasr import ops
from faoctasr.utils import set_requires_grad

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
H = 256
torch.manual_seed(0); random.seed(1234)
ts = faoctasr.TrainStep(device="cuda", overlap_wgrad=False)
g = torch.Generator().manual_seed(7)
a = (torch.rand(B, 1, H, H, generator=g) * 2 - 1).cuda()
b = (torch.rand(B, 1, H, H, generator=g) * 2 - 1).cuda()
o = ts.forward_generators(a, b)
set_requires_grad([ts.netD_A, ts.netD_B], False)
L = ts.generator_loss(o, a, b)
grads = []
for i in range(5):
    ts.opt_G.zero_grad()
    L["loss_G"].backward(retain_graph=True)
    torch.cuda.synchronize()
    grads.append(ts.opt_G.grad.clone())
ref = grads[0].double()
for i in range(1, 5):
    x = grads[i].double()
    print("G backward repeat %d: arena rel L2 %.3e" % (i, float((x - ref).norm() / ref.norm())))
    worst = []
    for n, off, p in zip(ts.opt_G.names, ts.opt_G.offsets, ts.opt_G.params):
        k = p.numel()
        worst.append((float((x[off:off + k] - ref[off:off + k]).norm()), n, float(ref[off:off + k].norm())))
    worst.sort(reverse=True)
    print("   ", ["%s %.1e (|g| %.1e)" % (n, e, m) for e, n, m in worst[:5]])
