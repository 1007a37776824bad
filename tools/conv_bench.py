"""Per-layer-shape timing of the convolution kernels at the benchmark workload (256x256, batch 8).
Prints TFLOP/s (f32 MFMA peak 157.3) for forward, input-gradient and weight-gradient of every distinct conv shape."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import faoctasr  # noqa: E402
from faoctasr import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
H = int(sys.argv[2]) if len(sys.argv) > 2 else 256
ONLY = sys.argv[3].split(",") if len(sys.argv) > 3 else None      # e.g. "c8,c5"
FWD_ONLY = len(sys.argv) > 4 and sys.argv[4] == "fwd"
if os.environ.get("FAOCTASR_PRECISION"):
    ops.conv_precision = ops.PRECISIONS[os.environ["FAOCTASR_PRECISION"]]
SHAPES = [
    # name, kind, Cin, Cout, k, stride, pad, outpad/reflect, input size divisor
    ("c1 stem 1->64 4x4s2", "conv", 1, 64, 4, 2, 1, 0, 1),
    ("c2 64->128 3x3", "conv", 64, 128, 3, 1, 1, 0, 2),
    ("c2 128->64 3x3", "conv", 128, 64, 3, 1, 1, 0, 2),
    ("c3 64->64 7x7 refl", "conv", 64, 64, 7, 1, 3, 1, 2),
    ("c3 128->64 7x7 refl", "conv", 128, 64, 7, 1, 3, 1, 2),
    ("c4 64->128 3x3s2", "conv", 64, 128, 3, 2, 1, 0, 2),
    ("c4 128->256 3x3s2", "conv", 128, 256, 3, 2, 1, 0, 4),
    ("c5 256->256 3x3", "conv", 256, 256, 3, 1, 1, 0, 8),
    ("c6 256->128 T3x3s2", "convT", 256, 128, 3, 2, 1, 1, 8),
    ("c6 128->64 T3x3s2", "convT", 128, 64, 3, 2, 1, 1, 4),
    ("c7 128->64 T4x4s2", "convT", 128, 64, 4, 2, 1, 0, 2),
    ("c8 64->64 3x3", "conv", 64, 64, 3, 1, 1, 0, 1),
    ("c9 64->1 3x3", "conv", 64, 1, 3, 1, 1, 0, 1),
    ("d1 1->64 4x4s2", "conv", 1, 64, 4, 2, 1, 0, 1),
    ("d2 64->128 4x4s2", "conv", 64, 128, 4, 2, 1, 0, 2),
    ("d2 128->256 4x4s2", "conv", 128, 256, 4, 2, 1, 0, 4),
    ("d2 256->512 4x4s2", "conv", 256, 512, 4, 2, 1, 0, 8),
    ("d2 512->512 4x4s2", "conv", 512, 512, 4, 2, 1, 0, 16),
    ("d3 512->512 4x4s1", "conv", 512, 512, 4, 1, 1, 0, 32),
]


def timeit(fn, n=10):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


def main():
    faoctasr._lib.load()
    tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
    print("%-24s %9s | %8s %7s | %8s %7s | %8s %7s" % ("layer (B=%d, %d^2)" % (B, H), "GFLOP", "fwd us", "TF/s", "dgrad us", "TF/s", "wgrad us", "TF/s"))
    for name, kind, ci, co, k, s, p, extra, div in SHAPES:
        if ONLY and name.split()[0] not in ONLY:
            continue
        hin = H // div
        x = torch.randn(B, ci, hin, hin, device="cuda")
        if kind == "conv":
            w = torch.randn(co, ci, k, k, device="cuda") * 0.02
            f = lambda: ops.conv2d(x, w, None, s, p, bool(extra))     # noqa: E731
        else:
            w = torch.randn(ci, co, k, k, device="cuda") * 0.02
            f = lambda: ops.conv_transpose2d(x, w, None, s, p, extra)  # noqa: E731
        y = f()
        if kind == "conv":
            flop = 2.0 * y.numel() * ci * k * k
        else:
            flop = 2.0 * x.numel() * co * k * k
        t_f = timeit(f)
        if FWD_ONLY:
            print("%-24s %9.2f | %8.1f %7.1f" % (name, flop / 1e9, t_f * 1e3, flop / t_f / 1e9))
            continue
        xg = x.clone().requires_grad_(True)
        wg = w.clone().requires_grad_(True)
        dy = torch.randn_like(y)
        # dgrad only / wgrad only through autograd with the other input frozen
        yd = ops.conv2d(xg, w, None, s, p, bool(extra)) if kind == "conv" else ops.conv_transpose2d(xg, w, None, s, p, extra)
        t_d = timeit(lambda: torch.autograd.grad(yd, xg, dy, retain_graph=True))
        yw = ops.conv2d(x, wg, None, s, p, bool(extra)) if kind == "conv" else ops.conv_transpose2d(x, wg, None, s, p, extra)
        t_w = timeit(lambda: torch.autograd.grad(yw, wg, dy, retain_graph=True))
        print("%-24s %9.2f | %8.1f %7.1f | %8.1f %7.1f | %8.1f %7.1f" % (name, flop / 1e9, t_f * 1e3, flop / t_f / 1e9, t_d * 1e3, flop / t_d / 1e9,
                                                                   t_w * 1e3, flop / t_w / 1e9))


if __name__ == "__main__":
    main()
