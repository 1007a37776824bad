"""Run-to-run spread of one train step from identical state and inputs (the step is not bit-reproducible: fp32 atomics), with the
weight gradients / discriminator branches on side streams and without.  A race would show as a spread far above the single-stream one.
usage: python tools/run_to_run.py [batch] [size] [pairs] [only] [--reproducible-forward]"""
import os
import random
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import faoctasr  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
H = int(sys.argv[2]) if len(sys.argv) > 2 else 256
PAIRS = int(sys.argv[3]) if len(sys.argv) > 3 else 3


REPRO = "--reproducible-forward" in sys.argv
PREC = [a.split("=")[1] for a in sys.argv if a.startswith("--precision=")]
PREC = PREC[0] if PREC else "f16x2"


def one(overlap, a, b):
    torch.manual_seed(0)
    random.seed(1234)
    ts = faoctasr.TrainStep(device="cuda", overlap_wgrad=overlap, reproducible_forward=REPRO, precision=PREC)
    L = ts.step(a, b, sync=True)
    out = (L, ts.opt_G.grad.clone(), ts.opt_D.grad.clone(), [(a.names, a.offsets, [p.numel() for p in a.params]) for a in (ts.opt_G, ts.opt_D)])
    del ts
    return out


def main():
    g = torch.Generator().manual_seed(7)
    a = (torch.rand(B, 1, H, H, generator=g) * 2 - 1).cuda()
    b = (torch.rand(B, 1, H, H, generator=g) * 2 - 1).cuda()
    d = lambda x, y: float((x.double() - y.double()).norm() / y.double().norm())
    ref = one(False, a, b)
    for overlap in ((False, True, False, True) if len(sys.argv) <= 4 else (False,)):
        for _ in range(PAIRS):
            r = one(overlap, a, b)
            print("overlap=%d  G arena %.3e  D arena %.3e  loss_G %.3e  loss_D_A %.3e" % (
                overlap, d(r[1], ref[1]), d(r[2], ref[2]), abs(r[0]["loss_G"] - ref[0]["loss_G"]) / abs(ref[0]["loss_G"]),
                abs(r[0]["loss_D_A"] - ref[0]["loss_D_A"]) / abs(ref[0]["loss_D_A"])), flush=True)
            for ai, tag in ((0, "G"), (1, "D")):
                names, offs, nums = r[3][ai]
                worst = []
                for n, o, k in zip(names, offs, nums):
                    x, y = r[1 + ai][o:o + k].double(), ref[1 + ai][o:o + k].double()
                    worst.append((float((x - y).norm()), n, float(y.norm())))
                worst.sort(reverse=True)
                print("   ", tag, ["%s %.1e (|g| %.1e)" % (n, e, m) for e, n, m in worst[:4]], flush=True)


if __name__ == "__main__":
    main()
