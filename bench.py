"""Benchmark of the north-star path: one full G+D train step (train.py:166-269 semantics) on synthetic
256x256 single-channel OCTA batches.

  python bench.py --gpus N --steps K --warmup W
N=1: BASELINE.json configs[1] (256x256x1, batch 8, fp32, 1 MI355X).  N>1 (launched by the driver through
torch.distributed.run, one rank per GPU): the same per-GPU batch on every rank (weak scaling), gradients of the two
flat arenas all-reduced over RCCL per optimizer phase.  Rank 0 prints ONE JSON line.

Plain `python bench.py --gpus N` (no launcher) spawns the N ranks itself before touching the GPU.

Beside the headline value the line carries
  roofline:           the dominant convolution kernel family BY TIME (HIP events on the launch stream around every
                      convolution-type C-ABI call over extra steps of the same workload, tagged with the kernel family the
                      dispatcher chose): algorithmic FLOP / measured time against the f32 MFMA peak (157.3 TFLOP/s), plus
                      the executed-FLOP fraction and, when profiles/r02_*.json exist, rocprofv3's MFMA-busy share and HBM bytes;
  roofline_families:  the same for every family (winograd, patch, gather_flat, wgrad_s1, wgrad_patch, wgrad_flat, ...);
  roofline_hbm:       Haar DWT / SSIM kernels, forward and backward, against the 8 TB/s HBM peak;
  cpu_baseline:       the CPU oracle (oracle/octa_oracle.py, kind "port") on this box's host cores, bounded sample, with
                      cached masks (value) and with the reference's per-call Python mask loops (reference_style_masks).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch                      # noqa: E402
import torch.distributed as dist  # noqa: E402

F32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md, "Peak FP32 (matrix)"
F16_MFMA_PEAK_TFLOPS = 2500.0     # MI355X_MICROARCH.md, "Peak BF16/FP16 MFMA": ~2.5 PF dense
GATHER = ("conv2d_fwd", "conv2d_dgrad", "conv_transpose2d_fwd", "conv_transpose2d_dgrad")
WGRAD = ("conv2d_wgrad", "conv_transpose2d_wgrad")


def host_threads():
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 16))


def conv_flops(name, a):
    """Algorithmic FLOPs (2*MACs) of one convolution-family launch from its C-ABI arguments."""
    if name in ("conv2d_fwd", "conv_transpose2d_fwd"):
        N, C, IH, IW, M, KH, KW, stride, pad = a[4:13]
    else:
        N, C, IH, IW, M, KH, KW, stride, pad = a[3:12]
    if name.startswith("conv2d"):
        OH, OW = (IH + 2 * pad - KH) // stride + 1, (IW + 2 * pad - KW) // stride + 1
        return 2.0 * N * M * OH * OW * C * KH * KW
    return 2.0 * N * C * IH * IW * M * KH * KW          # transposed: every input pixel meets every tap


ROUTES = {1: "gather_flat", 2: "patch", 3: "winograd", 4: "bf16x3", 5: "m1_head", 6: "narrow", 7: "stem_dgrad", 16: "stem_wgrad", 11: "wgrad_flat", 12: "wgrad_patch", 13: "wgrad_s1", 14: "m1_wgrad",
          15: "wgrad_x3"}
KERNEL_OF = {"narrow": "igemm_nm_kernel (GEMM-shaped implicit GEMM on the packed weights, maps narrower than 24)",
             "gather_flat": "igemm_gather_kernel<64> (flat implicit GEMM, maps narrower than 24)",
             "patch": "igemm_patch_kernel (LDS-patch implicit GEMM: 7x7, stride-2, transposed phases)",
             "winograd": "igemm_wino_kernel (Winograd F(2x2,3x3), stride-1 3x3 forward + input gradient)",
             "bf16x3": "igemm_bf16x3_kernel (split-operand LDS-patch implicit GEMM, 3 16-bit MFMAs per product: conv fwd / dgrad, convT fwd / dgrad)",
             "m1_head": "conv_m1_fwd_kernel (64 -> 1 head, VALU)",
             "stem_dgrad": "stem_dgrad_kernel (4x4 stride-2 stems with 1..4 input channels, input gradient, VALU)",
             "stem_wgrad": "stem_wgrad_kernel (4x4 stride-2 stems with 1..4 input channels, weight gradient, VALU)",
             "wgrad_flat": "igemm_wgrad_kernel (flat weight gradient, narrow maps)",
             "wgrad_patch": "wgrad_patch_kernel (LDS-patch weight gradient: stride 2, ragged shapes)",
             "wgrad_s1": "wgrad_s1_kernel (stride-1 / 4x4 stride-2 weight gradient, staging pipelined inside the MFMA loop)",
             "m1_wgrad": "conv_m1_wgrad_kernel (64 -> 1 head, VALU)",
             "wgrad_x3": "wgrad_x3_kernel / wgrad_x3_row_kernel (split-operand weight gradient: transposed-read X image)"}
#: share of the algorithmic (direct-convolution) FLOP a family really executes on the matrix pipe (split operands: three 16-bit
#: MFMAs per fp32 product)
EXECUTED = {"winograd": 16.0 / 36.0, "bf16x3": 3.0, "wgrad_x3": 3.0}
#: fp32-equivalent peak of the pipe a family runs on: the exact-f32 MFMA unless listed; a split-operand product costs three 16-bit
#: MFMAs, so the most fp32-class FLOP/s that arithmetic can deliver is the 16-bit dense peak / 3
PEAK = {"bf16x3": F16_MFMA_PEAK_TFLOPS / 3.0, "wgrad_x3": F16_MFMA_PEAK_TFLOPS / 3.0}
SPLIT_NAME = {"f16x2": "f16x2 (fp16 hi/lo pairs of x * 2^k, v_mfma_f32_32x32x16_f16)", "bf16x3": "bf16x3 (bf16 hi/lo pairs, v_mfma_f32_32x32x16_bf16)"}


class LaunchTimer:
    """HIP events (torch.cuda.Event on the launch stream = torch's current stream) around every convolution-type C-ABI call,
    tagged with the kernel family the dispatcher chose (faoctasr_last_route)."""

    def __init__(self, names):
        self.names = set(names)
        self.rec = []

    def add(self, name, args, s, e, route):
        self.rec.append((ROUTES.get(route, "route%d" % route), conv_flops(name, args), s, e))

    def families(self, nsteps):
        out = {}
        for fam, fl, s, e in self.rec:
            d = out.setdefault(fam, {"ms": 0.0, "flop": 0.0, "launches": 0})
            d["ms"] += s.elapsed_time(e)
            d["flop"] += fl
            d["launches"] += 1
        res = {}
        for fam, d in out.items():
            tf = d["flop"] / (d["ms"] * 1e-3) / 1e12
            ex = EXECUTED.get(fam, 1.0)
            peak = PEAK.get(fam, F32_MFMA_PEAK_TFLOPS)
            hw_peak = F16_MFMA_PEAK_TFLOPS if fam in PEAK else F32_MFMA_PEAK_TFLOPS      # the pipe's own dense peak, for the executed FLOP
            res[fam] = {"kernel": KERNEL_OF.get(fam, fam), "launches_per_step": d["launches"] // nsteps, "ms_per_step": round(d["ms"] / nsteps, 3),
                        "avg_launch_us": round(1e3 * d["ms"] / d["launches"], 2), "gflop_per_launch": round(d["flop"] / d["launches"] / 1e9, 3),
                        "peak_tflops": round(peak, 1), "tflops_algorithmic": round(tf, 2), "frac_algorithmic": round(tf / peak, 4),
                        "tflops_executed": round(tf * ex, 2), "executed_pipe_peak_tflops": hw_peak, "frac_executed": round(tf * ex / hw_peak, 4)}
        return res


DTYPE = {"f32": "f32 (exact-f32 MFMA: v_mfma_f32_32x32x2_f32 / 16x16x4_f32)",
         "f16x2": "f32 (tensors and accumulators fp32; the convolutions' operands enter the matrix cores as f16x2 pairs: hi = f16(x 2^k), lo = f16(x 2^k - hi), "
                  "k from the tensor's largest magnitude; 3 v_mfma_f32_32x32x16_f16 per product) on every map >= 24 wide with >= 16 channels; "
                  "narrow maps, 1-channel stems / head and everything that is not a convolution: exact f32; error class of the exact-f32 MFMA kernels: precision_evidence)",
         "bf16x3": "f32 tensors and fp32 accumulators; convolution operands as bf16 hi/lo pairs (3 v_mfma_f32_32x32x16_bf16 per product: 16 significant bits), f32 elsewhere"}
#: why the f16x2 step is reported as the fp32 headline ("error at or below the exact-f32 kernels' and the step beats
#: the f32-MFMA step"): measured, committed, and re-checked by the GPU tests named here
PRECISION_EVIDENCE = {
    "claim": "the f16x2 contraction is in the exact-f32 MFMA kernels' error class; it is not a reduced-precision run",
    "per_layer_error_vs_fp64": "94 (layer shape, y / dx / dw) measurements: f16x2 error = 0.65x .. 1.69x the exact-f32 kernels' (mean 0.97x, 68 of 94 at or "
                               "below), both 1.5e-7 .. 1e-6 relative L2; bf16x3 ~15x higher -- profiles/r04_f16x2_layer_errors.txt, "
                               "tests/test_gpu_ops.py::test_conv2d_f16x2 / test_conv_transpose2d_f16x2",
    "step_error_vs_fp64_oracle": "192^2 batch 2, step 0, full gradient of each network vs an fp64 run of the oracle: f16x2 4.3e-3 / 3.1e-3 / 3.2e-6 / 3.0e-4 "
                                 "against exact-f32 4.6e-3 / 3.6e-3 / 3.4e-6 / 5.1e-3 (A2B / B2A / D_A / D_B) -- profiles/r04_step_error_vs_fp64.txt, "
                                 "tests/test_gpu_step.py::test_step_gradients_vs_fp64_oracle_f16x2_beside_f32",
    "ten_step_trajectory": "ten consecutive steps from one initial state, fp32 CPU oracle vs exact-f32 HIP vs f16x2 HIP: both HIP runs drift from the oracle at "
                           "the same rate (1e-7 at step 0, 1e-4 at step 2, ~1.3e-2 at steps 7-9; worst deviation f16x2 1.7e-2, exact-f32 1.3e-2) -- "
                           "profiles/r04_ten_step_trajectory.txt, tests/test_gpu_step.py::test_ten_step_trajectory_f16x2_tracks_the_oracle_like_exact_f32",
    "three_hundred_steps": "300 steps at the headline workload from one seed: parameter-arena distance to an exact-f32 run (G / D) at step 300 -- a second "
                           "exact-f32 run 5.9e-2 / 5.3e-2 (the step is not bit-reproducible and GAN training amplifies it), f16x2 6.0e-2 / 5.2e-2, bf16x3 "
                           "6.2e-2 / 5.6e-2; no non-finite loss -- profiles/r04_long_run_300_steps.txt, tools/long_run.py",
    "reference_fixtures": "same bars as the exact-f32 step against the reference's fixtures (losses 1e-3, gradient norms 2e-3, 3 configs, all steps): "
                          "tests/test_gpu_step.py::test_train_step_f16x2_precision",
    "bare_mfma_probe": "32x32 tile, K = 576 .. 65536: f16x2 2.7e-7 .. 3.6e-6 vs the fp32 FMA chain / v_mfma_f32_32x32x2_f32 3.2e-7 .. 3.4e-6 -- "
                       "tools/probe/split_precision_error.hip, profiles/r04_split_precision_error.log",
    "exact_f32_step": "the same step on the exact-f32 MFMA kernels is measured by this run too: field exact_f32_mfma"}


def make_batch(B, H, device, rank):
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    a = (torch.rand(B, 1, H, H, generator=g) * 2 - 1).to(device)
    b = (torch.rand(B, 1, H, H, generator=g) * 2 - 1).to(device)
    return a, b


def cpu_baseline(H, budget_s=15.0):
    """The CPU oracle on the host cores, bounded: batch-2 steps (per-image work identical to the workload) after one untimed
    warm-up step, repeated until ~budget_s seconds of timed CPU work have accumulated.  Two legs: masks of
    the frequency split cached / vectorised (the headline `value`), and built per call by a Python double loop the way the
    reference does (utils.py:71-91) -- `reference_style_masks`."""
    from oracle import octa_oracle as O
    nt = host_threads()
    torch.set_num_threads(nt)
    S = O.StepOracle(seed=0)
    B = 2
    a, b = O.synthetic_batch(B, H)
    S.train_step(a, b)

    def leg(budget, max_steps):
        steps, total = 0, 0.0
        while total < budget and steps < max_steps:
            t0 = time.time()
            S.train_step(a, b)
            total += time.time() - t0
            steps += 1
        return steps, total

    steps, total = leg(budget_s, 8)
    out = {"value": round(B * steps / total, 4), "unit": "images/s", "cores": nt, "kind": "port",
           "sample": "CPU oracle (oracle/octa_oracle.py): %d timed %dx%d batch-%d fp32 train steps after 1 warm-up, %d torch threads, %.1f s of CPU work"
                     % (steps, H, H, B, nt, total)}
    O.MASK_STYLE = "loop"
    try:
        steps2, total2 = leg(budget_s * 0.6, 2)
    finally:
        O.MASK_STYLE = "vectorised"
    out["reference_style_masks"] = {"value": round(B * steps2 / total2, 4), "unit": "images/s",
                                    "sample": "same step with the Gaussian masks rebuilt per call by the reference's Python double loop "
                                              "(utils.py:71-91): %d timed steps, %.1f s" % (steps2, total2)}
    return out


HBM_PEAK_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E ~8 TB/s


def hbm_kernels(device, images=512, H=256):
    """HBM roofline lines of the frequency-path kernels north_star names (Haar DWT level, SSIM window), forward and backward:
    algorithmic bytes / HIP-event time on `images` 256x256 planes (134 MB per tensor, beyond the 256 MB MALL for a pair)."""
    from faoctasr import ops
    g = torch.Generator(device="cpu").manual_seed(7)
    a = torch.rand(images, 1, H, H, generator=g).to(device)
    b = torch.rand(images, 1, H, H, generator=g).to(device)
    plane = images * H * H * 4

    def timed(fn, n=10):
        fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n):
            fn()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / n * 1e-3

    ll, hi = ops.haar_afb2d(a)
    gsum = torch.ones(images, device=device)
    da, db = torch.empty_like(a), torch.empty_like(b)
    dxh = torch.empty_like(a)
    from faoctasr._lib import call, ptr, stream_ptr
    # vector FMAs per pixel of the separable 11-tap window filters: forward 5 moment maps x (11 row + 11 column taps) = 110;
    # backward the same 110 plus 5 derivative maps x 22 taps = 220 (the point-wise SSIM algebra, ~40 operations, is not counted)
    px = images * H * H
    cases = (("haar_dwt2d_fwd (DWTForward level, AFB2D)", lambda: ops.haar_afb2d(a), 2 * plane, 0),
             ("haar_dwt2d_bwd (AFB2D backward = SFB2D synthesis)",
              lambda: call("haar_dwt2d_bwd", ptr(ll), ptr(hi), ptr(dxh), images, H, H, stream_ptr()), 2 * plane, 0),
             ("ssim_fwd (11x11 Gaussian window SSIM, per-image sums)", lambda: ops.ssim(a, b), 2 * plane, 110),
             ("ssim_bwd (gradient of the SSIM mean w.r.t. both images)",
              lambda: call("ssim_bwd", ptr(a), ptr(b), ptr(gsum), images, 1.0, ptr(da), ptr(db), images, 1, H, H, stream_ptr()), 4 * plane, 220))
    out = []
    with torch.no_grad():
        for name, fn, nbytes, fma in cases:
            t = timed(fn)
            gbs = nbytes / t / 1e9
            row = {"kernel": name, "bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                   "frac": round(gbs / HBM_PEAK_GBS, 4), "algorithmic_bytes": nbytes, "us": round(t * 1e6, 1),
                   "sample": "%d planes of %dx%d fp32" % (images, H, H)}
            if fma:
                # these two are bound by the vector ALU, not by HBM: quote that fraction beside the HBM one.  Peak = the fp32 vector
                # rate of the spec sheet (157.3 TFLOP/s, which counts v_pk_fma_f32 = 2 FMA per lane and issue slot; a plain
                # v_fma_f32 stream -- what these filters are, see build.FILE_FLAGS -- tops out at half of it)
                tf = 2.0 * fma * px / t / 1e12
                row.update({"valu_tflops": round(tf, 2), "valu_peak_tflops": F32_MFMA_PEAK_TFLOPS, "valu_frac": round(tf / F32_MFMA_PEAK_TFLOPS, 4),
                            "valu_frac_of_unpacked_fma_rate": round(tf / (F32_MFMA_PEAK_TFLOPS / 2), 4), "fma_per_pixel": fma})
            out.append(row)
    return out


def self_launch(n):
    """Run this script as n ranks of one node under torch.distributed.run and return its exit code.  Nothing in this (parent)
    process has initialised the GPU, it only waits; a failing rank makes torchrun -- and so the parent -- exit non-zero."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, host_threads() // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + [a for a in sys.argv[1:] if a != "--force-launch"]
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8, help="per-GPU batch (BASELINE config 2: 8)")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--precision", default="f16x2", choices=["f16x2", "f32", "bf16x3"],
                    help="conv contraction: f16x2 = fp32 operands as scaled fp16 hi/lo pairs on the f16 MFMA, fp32 accumulate, error class of "
                         "the exact-f32 kernels (default, the headline); f32 = exact fp32 MFMA (the headline of rounds 1-3, reported beside it); "
                         "bf16x3 = bf16 hi/lo pairs (16 significant bits)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-alt", action="store_true", help="skip the secondary measurements (the same step at the other precisions)")
    ap.add_argument("--no-graph", action="store_true", help="skip the hipGraph-captured step measurement")
    ap.add_argument("--graph-only", action="store_true", help="(internal) measure only the hipGraph-captured step and print it")
    ap.add_argument("--no-overlap-exchange", action="store_true", help="data-parallel runs: keep the gradient all-reduces on the main stream")
    ap.add_argument("--prio", default=None, help="(experiments) TrainStep.stream_priorities, comma separated")
    ap.add_argument("--min-pixels", type=int, default=None, help="(experiments) TrainStep.overlap_min_pixels")
    ap.add_argument("--layout", default=None, help="(experiments) TrainStep.stream_layout, e.g. 012301")
    ap.add_argument("--eager-chain-a-on-caller", action="store_true", help="(experiments) eager steps in the capturable chain arrangement")
    ap.add_argument("--graph-variant", default=None, help="(experiments) captured step: 'side-wgrad' (weight gradients on their side stream inside the capture too), "
                                                          "'single-chain', 'one-stream'")
    ap.add_argument("--no-overlap", action="store_true", help="weight gradients on the main stream (the form the per-kernel profiles are taken in)")
    ap.add_argument("--force-launch", action="store_true", help="take the self-launch path (torch.distributed.run children) even for one GPU")
    ap.add_argument("--ddp-graph", action="store_true", help="data-parallel runs: also measure the hipGraph-captured step (RCCL inside the capture)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or args.force_launch):
        # plain `python bench.py --gpus N`: become a launcher BEFORE anything touches the GPU in this process -- the N ranks
        # are children (one per GPU, torch.distributed.run), rank 0 prints the JSON line on the inherited stdout
        sys.exit(self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != max(args.gpus, 1) and world > 1:
        print("bench.py: --gpus %d but WORLD_SIZE=%d; using WORLD_SIZE" % (args.gpus, world), file=sys.stderr)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # under torch.distributed.run the RCCL path is taken at any world size (world 1 included: --force-launch rehearses it)
    distributed = world > 1 or "TORCHELASTIC_RUN_ID" in os.environ or os.environ.get("FAOCTASR_FORCE_DIST") == "1"
    if distributed:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)

    import faoctasr
    from faoctasr import _lib
    _lib.load()                                  # fails loudly when the HIP library is missing
    torch.manual_seed(0)
    import random
    random.seed(1234 + rank)
    # distributed: TrainStep creates the RCCL communicator behind the C ABI and broadcasts rank 0's arenas / BN buffers
    if args.graph_only:
        args.no_roofline = args.no_alt = args.no_cpu_baseline = True
    if args.layout:
        faoctasr.TrainStep.stream_layout = faoctasr.TrainStep.stream_layout_comm = faoctasr.TrainStep.stream_layout_f32 = args.layout
    if args.eager_chain_a_on_caller:
        faoctasr.TrainStep.eager_chain_A_forked = False
    if args.graph_variant == "side-wgrad":
        faoctasr.TrainStep.capture_side_wgrad = True
    elif args.graph_variant == "deferred-wgrad":
        faoctasr.TrainStep.capture_side_wgrad = "deferred"
    elif args.graph_variant == "single-chain":
        faoctasr.TrainStep.capture_two_chains = False
    if args.prio:
        faoctasr.TrainStep.stream_priorities = [int(v) for v in args.prio.split(",")]
    if args.min_pixels is not None:
        faoctasr.TrainStep.overlap_min_pixels = args.min_pixels
    ts = faoctasr.TrainStep(device=device, distributed=distributed, precision=args.precision, overlap_wgrad=not args.no_overlap)
    ts.overlap_exchange = not args.no_overlap_exchange
    B, H = args.batch, args.size
    real_A, real_B = make_batch(B, H, device, rank)

    def barrier():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        ts.step(real_A, real_B)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        L = ts.step(real_A, real_B)
    barrier()
    elapsed = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    loss_G = float(L["loss_G"])
    # host time to ENQUEUE a step (no device synchronisation inside the loop): how far ahead of the GPU the host runs
    host_ms = None
    if args.steps >= 5:                       # (short runs are the rocprofv3 --pmc passes: every extra dispatch is recorded there)
        th = time.perf_counter()
        for _ in range(5):
            ts.step(real_A, real_B)
        host_ms = 1e3 * (time.perf_counter() - th) / 5
        barrier()
    ms_per_step = 1e3 * elapsed / args.steps
    value = world * B * args.steps / elapsed

    roof = None
    extra = {"host_enqueue_ms_per_step": None if host_ms is None else round(host_ms, 2)}
    if not args.no_roofline:
        # every rank runs these extra steps (they contain the gradient all-reduces); only rank 0 brackets its launches
        timer = LaunchTimer(GATHER + WGRAD) if rank == 0 else None
        _lib.launch_timer = timer
        nroof = max(1, min(3, args.steps))
        overlap, ts.overlap_wgrad = ts.overlap_wgrad, False          # one stream: each launch's events bracket that kernel alone
        for _ in range(nroof):
            ts.step(real_A, real_B)
        torch.cuda.synchronize()
        ts.overlap_wgrad = overlap
        _lib.launch_timer = None
    if rank == 0 and not args.no_roofline:
        fams = timer.families(nroof)
        # MFMA-busy share and HBM bytes per launch are NOT measured by this run: they are copied from the newest committed summaries of
        # the builder's own rocprofv3 --pmc passes over this same command (tools/profile_round.sh), and every copy says so
        import glob
        prof, src = {}, {}
        for key, pat in (("mfma", "r[0-9][0-9]_mfma_utilisation.json"), ("traffic", "r[0-9][0-9]_traffic.json")):
            found = sorted(glob.glob(os.path.join(ROOT, "profiles", pat)))
            prof[key] = {}
            if found:
                try:
                    prof[key] = json.load(open(found[-1])).get("families", {})
                    src[key] = "profiles/" + os.path.basename(found[-1])
                except Exception:
                    pass
        static_note = "static: copied from %s / %s (builder-run rocprofv3 --pmc passes of this command, tools/profile_round.sh), not measured by this run" % (
            src.get("mfma"), src.get("traffic"))
        for fam, d in fams.items():
            busy = prof["mfma"].get(fam, {}).get("mfma_busy_frac")
            if busy is not None:
                d["mfma_busy_frac_rocprof"] = busy
            d["hbm_bytes_per_launch_rocprof"] = prof["traffic"].get(fam, {}).get("hbm_bytes_per_launch")
            d["rocprof_fields_source"] = static_note
        dom = max(fams, key=lambda k: fams[k]["ms_per_step"])                 # the dominant kernel family BY TIME
        d = fams[dom]
        roof = {"bound": "mfma", "kernel": d["kernel"], "achieved": d["tflops_algorithmic"], "peak": d["peak_tflops"], "unit": "TFLOP/s",
                "frac": d["frac_algorithmic"], "traffic": d["hbm_bytes_per_launch_rocprof"], "family": dom,
                "traffic_source": static_note,
                "definition": "achieved = algorithmic (direct-convolution, fp32-equivalent) FLOP of the family's launches / their HIP-event time; "
                              "peak = the most fp32-class FLOP/s the family's arithmetic can deliver: split operands cost 3 16-bit MFMAs per product, "
                              "so 2500 / 3 = 833.3 TFLOP/s (dense f16/bf16 MFMA peak of MI355X_MICROARCH.md / 3); exact-f32 families 157.3 TFLOP/s "
                              "(at 2.4 GHz; under this load the chip holds ~2.0 GHz, DESIGN.md 5).  frac_executed = executed MFMA FLOP (3x, or "
                              "16/36 for Winograd) / the pipe's own dense peak -- the same number for split families",
                "frac_executed": d["frac_executed"], "mfma_busy_frac_rocprof": d.get("mfma_busy_frac_rocprof"),
                "launches_per_step": d["launches_per_step"], "avg_launch_us": d["avg_launch_us"], "gflop_per_launch": d["gflop_per_launch"],
                "ms_per_step": d["ms_per_step"]}
        extra["roofline_families"] = fams
        tot_ms = sum(v["ms_per_step"] for v in fams.values())
        tot_fl = sum(v["gflop_per_launch"] * v["launches_per_step"] for v in fams.values()) / 1e3
        extra["conv_ms_per_step"] = round(tot_ms, 2)
        extra["conv_tflop_per_step"] = round(tot_fl, 3)
        extra["conv_tflop_note"] = ("BASELINE.md quotes 1.566 TFLOP per image-step (12.53 at batch 8); its hook count includes every discriminator "
                                    "convolution twice (41.8 GMAC per image: D forward 31.4 -> 15.7, D input gradient 10.5 -> 5.2, D-step backward "
                                    "41.9 -> 20.9; per-layer table in DESIGN.md 5.2), so the step's algorithmic work is 1.482 TFLOP per image")
    # secondary measurements: the same step at the other precisions (the exact-f32 MFMA path was the headline of rounds 1-3)
    if rank == 0 and not distributed and not args.no_alt:
        notes = {"f32": ("exact_f32_mfma", "every convolution on v_mfma_f32_*_f32 (Winograd F(2x2,3x3) on the dense stride-1 3x3 layers): the headline "
                                           "arithmetic of rounds 1-3"),
                 "bf16x3": ("alt_precision_bf16x3", "bf16 hi/lo pairs, 3 MFMAs per product: 16 significant bits, per-layer error ~15x the exact-f32 kernels'"),
                 "f16x2": ("alt_precision_f16x2", "scaled fp16 hi/lo pairs, 3 MFMAs per product")}
        for prec in ("f32", "bf16x3", "f16x2"):
            if prec == args.precision:
                continue
            # a step of its own: the stream layout is chosen per precision when the step is built (TrainStep.stream_layout_f32)
            alt = faoctasr.TrainStep(device=device, distributed=False, precision=prec, overlap_wgrad=not args.no_overlap)
            for _ in range(3):
                alt.step(real_A, real_B)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                alt.step(real_A, real_B)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t1
            del alt
            key, note = notes[prec]
            extra[key] = {"value": round(B * args.steps / dt, 3), "unit": "images/s", "ms_per_step": round(1e3 * dt / args.steps, 3), "dtype": note}
    # the same step as ONE captured hipGraph (SURVEY 8f-1 / BASELINE config 5): reported beside the headline, never as it
    # (data-parallel: every rank captures and replays in lockstep, RCCL's kernels are graph nodes -- opt-in with --ddp-graph)
    graph_note = "whole G+D step replayed as one hipGraph: device-side replay buffer, AdamW scalars in device memory"
    if args.graph_only or (not args.no_graph and (not distributed or args.ddp_graph)):
        # in this process (round 2 ran it in a child because hipStreamEndCapture crashed on some schedules; the cause -- forked
        # streams waiting on each other, DESIGN.md 4.4 -- is removed from the schedule, so there is nothing left to isolate)
        if args.graph_variant == "one-stream":
            ts.overlap_wgrad = False
        try:                                     # a schedule the capture cannot hold raises (train._CaptureGuard): keep the headline
            gs = faoctasr.GraphedTrainStep(ts, real_A, real_B)
            for _ in range(2):
                gs.step(real_A, real_B)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                gs.step(real_A, real_B)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t1
            extra["hipgraph_step"] = {"value": round(world * B * args.steps / dt, 3), "unit": "images/s", "ms_per_step": round(1e3 * dt / args.steps, 3),
                                      "note": graph_note}
            del gs
        except faoctasr.KernelError as err:
            extra["hipgraph_step"] = {"error": str(err)}
        if args.graph_only:
            print(json.dumps({"hipgraph_step": extra["hipgraph_step"]}), flush=True)
            return
    if rank == 0 and world == 1 and not args.no_roofline:
        extra["roofline_hbm"] = hbm_kernels(device)
    if distributed:
        dist.barrier()

    if rank == 0:
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(H)
        line = {"metric": "train-step images/sec (G+D fwd+bwd) on 256x256 OCTA", "value": round(value, 3), "unit": "images/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": DTYPE[args.precision], "data": "synthetic",
                "config": {"workload": "%dx%dx1 OCTA pairs, batch %d per GPU, fp32 tensors, full G+D train step "
                                       "(4 frequency splits, 6 G fwd, 6 D fwd, 3 backward, 2 AdamW)" % (H, H, B),
                           "global_batch": world * B, "parallelism": "dp%d" % world, "loss_G": round(loss_G, 5),
                           "rccl_ranks": ts.comm.ranks if ts.comm is not None else 0},
                "roofline": roof, "cpu_baseline": cpu}
        if args.precision == "f16x2":
            line["precision_evidence"] = PRECISION_EVIDENCE
        line.update(extra)
        print(json.dumps(line), flush=True)
    if distributed:
        ts.comm.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
