"""Per-shape time of every convolution-family C-ABI call inside one benchmark train step (HIP events around each launch).
usage: python tools/step_breakdown.py [batch] [size] [precision]"""
import os
import sys
from collections import defaultdict

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import faoctasr  # noqa: E402
from faoctasr import _lib  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
H = int(sys.argv[2]) if len(sys.argv) > 2 else 256
PREC = sys.argv[3] if len(sys.argv) > 3 else "f32"


class Timer:
    names = set(bench.GATHER + bench.WGRAD)

    def __init__(self):
        self.rec = []

    def add(self, name, args, s, e, route):
        sh = args[4:13] if name.endswith("_fwd") else args[3:12]
        self.rec.append((name + ":" + bench.ROUTES.get(route, "route%d" % route), tuple(int(v) for v in sh), bench.conv_flops(name, args), s, e))


def main():
    _lib.load()
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    ts = faoctasr.TrainStep(device=dev, precision=PREC, overlap_wgrad=False)      # one stream: a call's events bracket its own kernels
    a, b = bench.make_batch(B, H, dev, 0)
    for _ in range(2):
        ts.step(a, b)
    torch.cuda.synchronize()
    t = Timer()
    _lib.launch_timer = t
    n = 2
    for _ in range(n):
        ts.step(a, b)
    torch.cuda.synchronize()
    _lib.launch_timer = None
    agg = defaultdict(lambda: [0, 0.0, 0.0])
    for name, sh, fl, s, e in t.rec:
        k = (name, sh)
        agg[k][0] += 1
        agg[k][1] += s.elapsed_time(e)
        agg[k][2] += fl
    rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
    tot = sum(v[1] for v in agg.values()) / n
    print("total conv-family ms/step %.2f" % tot)
    print("%-40s %-36s %5s %9s %8s %7s %6s" % ("call", "N,C,IH,IW,M,KH,KW,stride,pad", "n/stp", "ms/step", "us/call", "TF/s", "%"))
    for (name, sh), (c, ms, fl) in rows:
        print("%-40s %-36s %5d %9.3f %8.1f %7.1f %6.1f" % (name, ",".join(map(str, sh)), c // n, ms / n, 1e3 * ms / c, fl / ms / 1e9, 100 * ms / n / tot))


if __name__ == "__main__":
    main()
