"""Several hundred train steps from one seed at two contraction precisions, beside a SECOND run of the exact-f32 step: the step is not
bit-reproducible (fp32 atomics), GAN training amplifies any difference, so two f32 runs drift apart too -- that drift is the yardstick
for the f16x2 run's.  Prints the losses every EVERY steps and the relative distance of the parameter arenas to the first f32 run.
usage: python tools/long_run.py [steps] [batch] [size] [every]"""
import os
import random
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import faoctasr  # noqa: E402

STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 300
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
H = int(sys.argv[3]) if len(sys.argv) > 3 else 256
EVERY = int(sys.argv[4]) if len(sys.argv) > 4 else 25
NBATCH = 16
KEYS = ("loss_G", "loss_D_A", "loss_D_B")


def batches():
    g = torch.Generator().manual_seed(11)
    out = []
    for _ in range(NBATCH):
        # smooth-ish images rather than white noise: a low-resolution field upsampled, plus a little noise
        lo = torch.rand(B, 1, H // 8, H // 8, generator=g)
        a = torch.nn.functional.interpolate(lo, size=(H, H), mode="bilinear", align_corners=False) * 2 - 1
        hi = torch.nn.functional.interpolate(torch.rand(B, 1, H // 4, H // 4, generator=g), size=(H, H), mode="bilinear", align_corners=False) * 2 - 1
        out.append(((a + 0.05 * torch.randn(B, 1, H, H, generator=g)).clamp(-1, 1).cuda(), hi.cuda()))
    return out


def run(precision, data):
    torch.manual_seed(0)
    random.seed(1234)
    ts = faoctasr.TrainStep(device="cuda", precision=precision)
    log, snaps = [], []
    for i in range(STEPS):
        a, b = data[i % NBATCH]
        last = (i + 1) % EVERY == 0 or i == 0
        L = ts.step(a, b, sync=last)
        if last:
            log.append((i + 1, [float(L[k]) for k in KEYS]))
            snaps.append((ts.opt_G.flat.clone(), ts.opt_D.flat.clone()))
    return log, snaps


def main():
    data = batches()
    runs = [("f32", run("f32", data)), ("f32 again", run("f32", data)), ("f16x2", run("f16x2", data)), ("bf16x3", run("bf16x3", data))]
    ref = runs[0][1]
    d = lambda x, y: float((x.double() - y.double()).norm() / y.double().norm())
    print("%d steps, batch %d, %dx%d, %d distinct synthetic batches cycled; columns per run: loss_G loss_D_A loss_D_B | arena distance to the first f32 run (G, D)" % (STEPS, B, H, H, NBATCH))
    for j, (step, _) in enumerate(ref[0]):
        line = "step %4d" % step
        for name, (log, snaps) in runs:
            line += " | %-9s %8.4f %7.4f %7.4f" % (name, *log[j][1])
            if name != "f32":
                line += "  dG %.1e dD %.1e" % (d(snaps[j][0], ref[1][j][0]), d(snaps[j][1], ref[1][j][1]))
        print(line, flush=True)
    bad = [name for name, (log, _) in runs for _, v in log if not all(x == x and abs(x) < 1e6 for x in v)]
    print("non-finite or exploding losses:", bad or "none")


if __name__ == "__main__":
    main()
