"""Sweep TrainStep.stream_layout over set partitions of the six roles (<= 4 streams) with bench.py; prints ms/step per layout.
usage: python tools/layout_search.py [n_random] [seed]     (run on the GPU box; each layout is one short bench.py run;
LAYOUT_EXTRA="--force-launch --gpus 1" in the environment sweeps the data-parallel path: RCCL communicator at world size 1)"""
import json
import os
import random
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def partitions(n, kmax):
    def rec(i, labels, k):
        if i == n:
            yield "".join(map(str, labels))
            return
        for c in range(min(k + 1, kmax)):
            yield from rec(i + 1, labels + [c], max(k, c + 1))
    yield from rec(1, [0], 1)


def run(layout, extra=()):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-alt", "--no-roofline", "--no-graph", "--steps", "8", "--warmup", "3",
           "--layout", layout] + list(extra) + os.environ.get("LAYOUT_EXTRA", "").split()
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout.decode()
    lines = [l for l in out.splitlines() if l.startswith("{")]
    return json.loads(lines[-1])["ms_per_step"] if lines else float("nan")


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    random.seed(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    allp = [p for p in partitions(6, 4) if len(set(p)) >= 2]
    pick = ["012201", "001212", "001232", "012101", "012221", "012102", "012312", "012301"] + random.sample(allp, n)
    res = []
    for p in pick:
        ms = run(p)
        res.append((ms, p))
        print("%s %.3f" % (p, ms), flush=True)
    res.sort()
    print("best:", res[:8], flush=True)


if __name__ == "__main__":
    main()
