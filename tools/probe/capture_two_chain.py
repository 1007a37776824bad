"""Does capturing the train step as a hipGraph survive (a) the two-chain generator schedule and (b) a stream waiting on its own event?
Round 2 recorded a process crash inside hipStreamEndCapture for both and worked around it without keeping a log.
Each configuration runs in a child process with faulthandler on; this driver prints exit status and the tail of each child's output.

    python tools/probe/capture_two_chain.py            # all configurations
    python tools/probe/capture_two_chain.py --child two_chains=1,selfwait=0,layout=001212
"""
import faulthandler
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def child(spec):
    faulthandler.enable(all_threads=True)
    cfg = dict(kv.split("=") for kv in spec.split(","))
    import random

    import torch

    import faoctasr
    from faoctasr import train as T
    from oracle import octa_oracle as O
    faoctasr.TrainStep.overlap_min_pixels = 0
    faoctasr.TrainStep.capture_two_chains = cfg.get("two_chains", "1") == "1"
    if "layout" in cfg:
        faoctasr.TrainStep.stream_layout = cfg["layout"]
    if cfg.get("selfwait", "0") == "1":                 # re-issue the waits a stream would make on its own events
        T._wait = lambda waiter, on: waiter.wait_stream(on)
        T._after = lambda waiter, mark: mark is not None and waiter.wait_event(mark.event)
    H, B = int(cfg.get("H", 192)), int(cfg.get("B", 2))

    def fresh():
        random.seed(1234)
        nets = {"A2B": faoctasr.NetworkA2B(), "B2A": faoctasr.NetworkB2A(), "D_A": faoctasr.FS_DiscriminatorA(1), "D_B": faoctasr.FS_DiscriminatorB(1)}
        specs = {"A2B": O.spec_network_a2b(), "B2A": O.spec_network_b2a(), "D_A": O.spec_fs_discriminator("sum"), "D_B": O.spec_fs_discriminator("cat")}
        for k, n in nets.items():
            n.load_state_dict(O.make_state(specs[k], k, 0), strict=True)
            n.cuda().train()
        return faoctasr.TrainStep(nets["A2B"], nets["B2A"], nets["D_A"], nets["D_B"])
    batches = [tuple(t.cuda() for t in O.synthetic_batch(B, H, seed=99 + s)) for s in range(2)]
    ts = fresh()
    eager = [ts.step(a, b, sync=True) for a, b in batches]
    del ts
    tg = fresh()
    print("stream handles: caller-side roles", {k: hex(getattr(tg, k).cuda_stream) for k in ("_side", "_side_D", "_idt", "_aba")},
          "branches", [hex(b.cuda_stream) for b in tg._branch], flush=True)
    print("capturing ...", flush=True)
    gs = faoctasr.GraphedTrainStep(tg, *batches[0])
    print("captured", flush=True)
    graph = [gs.step(a, b, sync=True) for a, b in batches]
    torch.cuda.synchronize()
    worst = max(abs(graph[0][k] - eager[0][k]) / max(abs(eager[0][k]), 1e-9) for k in eager[0])
    print("OK step-0 losses graph vs eager: worst relative difference %.2e; loss_G %.6f / %.6f; step 1 loss_G %.6f / %.6f"
          % (worst, graph[0]["loss_G"], eager[0]["loss_G"], graph[1]["loss_G"], eager[1]["loss_G"]), flush=True)


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(sys.argv[2])
        return
    configs = ["two_chains=0,selfwait=0,layout=001212", "two_chains=1,selfwait=0,layout=001212", "two_chains=1,selfwait=0,layout=001232",
               "two_chains=0,selfwait=1,layout=001212", "two_chains=1,selfwait=1,layout=001212", "two_chains=1,selfwait=1,layout=012345"]
    if len(sys.argv) > 1:
        configs = sys.argv[1:]
    env0 = dict(os.environ, PYTHONFAULTHANDLER="1", AMD_LOG_LEVEL=os.environ.get("AMD_LOG_LEVEL", "1"))
    shim = os.path.join(ROOT, "tools", "probe", "bin", "libhipcaptrace.so")
    outdir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(outdir, exist_ok=True)
    for spec in configs:
        print("=== %s" % spec, flush=True)
        env = dict(env0)
        trace = None
        if os.environ.get("CAPTRACE") and os.path.exists(shim):        # record the capture's event / wait / launch sequence (hip_capture_trace.cpp)
            trace = os.path.join(outdir, "captrace_%s.txt" % spec.replace(",", "_").replace("=", ""))
            env.update(LD_PRELOAD=shim, HIPCAPTRACE_OUT=trace)
            if "defer=1" in spec:
                env["HIPCAPTRACE_DEFER_DESTROY"] = "1"
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", spec], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                               timeout=420)
            rc, out = r.returncode, r.stdout.decode(errors="replace")
        except subprocess.TimeoutExpired as e:
            rc, out = "timeout", (e.stdout or b"").decode(errors="replace")
        lines = out.strip().splitlines()
        print("\n".join("    " + l for l in lines[-40:]))
        print("    ==> exit %s%s" % (rc, " (signal %d)" % -rc if isinstance(rc, int) and rc < 0 else ""), flush=True)
        if trace and os.path.exists(trace):
            r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "probe", "capture_trace_report.py"), trace], stdout=subprocess.PIPE,
                               stderr=subprocess.STDOUT)
            print("\n".join("    | " + l for l in r.stdout.decode(errors="replace").splitlines()), flush=True)


if __name__ == "__main__":
    main()
