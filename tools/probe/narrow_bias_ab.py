import sys, torch
sys.path.insert(0, "/root/repo")
import faoctasr
from faoctasr import ops
faoctasr._lib.load()
ops.conv_precision = ops.PRECISIONS["f16x2"]
def t(fn, n=30):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for (C, M, H, st, pad) in ((128, 256, 32, 2, 1), (256, 512, 16, 1, 1), (256, 512, 17, 1, 1)):
    x = torch.randn(8, C, H, H, device="cuda"); w = torch.randn(M, C, 4, 4, device="cuda") * 0.02; b = torch.randn(M, device="cuda")
    with torch.no_grad():
        print(C, M, H, st, "bias %.1f us   no bias %.1f us   route %s" % (t(lambda: ops.conv2d(x, w, b, st, pad)), t(lambda: ops.conv2d(x, w, None, st, pad)), faoctasr._lib.load().faoctasr_last_route()))
