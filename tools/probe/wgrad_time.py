"""Kernel time of weight gradients through the C ABI (accumulate mode, no autograd)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import faoctasr
from faoctasr._lib import call, ptr, stream_ptr, load
load()
dev = "cuda"
for (N, C, H, M, k, s, p) in ((8, 256, 32, 256, 3, 1, 1), (8, 64, 256, 64, 3, 1, 1), (8, 128, 128, 64, 3, 1, 1), (8, 64, 128, 64, 7, 1, 3)):
    OH = (H + 2 * p - k) // s + 1
    x = torch.randn(N, C, H, H, device=dev); dy = torch.randn(N, M, OH, OH, device=dev); dw = torch.zeros(M, C, k, k, device=dev)
    st = stream_ptr()
    fn = lambda: call("conv2d_wgrad", ptr(x), ptr(dy), ptr(dw), N, C, H, H, M, k, k, s, p, 0, 1, 0, st)
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(30): fn()
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) / 30 * 1e3
    fl = 2.0 * N * M * OH * OH * C * k * k
    print("wgrad N%d C%d H%d M%d k%d: %.1f us  %.1f TF" % (N, C, H, M, k, us, fl / us / 1e6), flush=True)
