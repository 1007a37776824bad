"""Kernel time of the stem input / weight gradient through the C ABI (no autograd in the loop)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import faoctasr
from faoctasr._lib import call, ptr, stream_ptr, load
load()
dev = "cuda"
for (N, C, H, M) in ((8, 1, 256, 64), (8, 1, 256, 128), (8, 3, 128, 64), (8, 1, 128, 64)):
    OH = H // 2
    x = torch.randn(N, C, H, H, device=dev); dy = torch.randn(N, M, OH, OH, device=dev); w = torch.randn(M, C, 4, 4, device=dev)
    dx = torch.empty_like(x); dw = torch.zeros_like(w)
    st = stream_ptr()
    def t(fn, n=50):
        fn(); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n): fn()
        e.record(); torch.cuda.synchronize()
        return s.elapsed_time(e) / n * 1e3
    td = t(lambda: call("conv2d_dgrad", ptr(dy), ptr(w), ptr(dx), N, C, H, H, M, 4, 4, 2, 1, None, 0, 0, st))
    tw = t(lambda: call("conv2d_wgrad", ptr(x), ptr(dy), ptr(dw), N, C, H, H, M, 4, 4, 2, 1, 0, 1, 0, st))
    print("N%d C%d H%d M%d: dgrad %.1f us, wgrad %.1f us (dy %.1f MB)" % (N, C, H, M, td, tw, dy.numel() * 4 / 1e6), flush=True)
