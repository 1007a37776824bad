"""GPU parity at the BASELINE.json configurations that round 1 never exercised:

  config 3  256x256, batch 64, precision "bf16x3", 3-level Haar wavelet-HF loss        -> vs the CPU oracle
  config 4  one rank's shard of the 8-GPU run: batch 32 through the RCCL path (world 1)   -> vs the non-distributed step
  config 5  one rank's shard: 512x512, batch 2, SSIM + 3-level DWT terms, hipGraph-captured, RCCL exchange inside the capture

plus the run-to-run spread of the step (fp32 atomics in split-K / weight-gradient accumulation make it non-bit-reproducible;
the spread is bounded here).  The oracle is test infrastructure (oracle/octa_oracle.py); the product path is the HIP library
behind the C ABI.  Sizes: the batch-64 oracle step needs ~155 GB of host memory and ~2 min of the box's 16 host threads."""
import gc
import os
import random
import socket

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu


def host_threads():
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 16))


@pytest.fixture(scope="module")
def fa():
    import faoctasr
    faoctasr._lib.load()
    faoctasr.TrainStep.overlap_min_pixels = 0      # the multi-stream schedule at every size (product default: from 2 x 256^2 pixels on)
    return faoctasr


@pytest.fixture(scope="module")
def O():
    from oracle import octa_oracle
    return octa_oracle


@pytest.fixture(scope="module")
def rccl_world1():
    """A world-size-1 process group on the RCCL backend: the all-reduce is the identity, everything else of the data-parallel
    path (communicator behind the C ABI, replica broadcast, grad_scale = 1/world, RCCL nodes inside a captured graph) is real."""
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, world_size=1, rank=0, device_id=torch.device("cuda", 0))
    yield
    dist.destroy_process_group()


def build_nets(fa, O, seed=0):
    nets = {"A2B": fa.NetworkA2B(), "B2A": fa.NetworkB2A(), "D_A": fa.FS_DiscriminatorA(1), "D_B": fa.FS_DiscriminatorB(1)}
    specs = {"A2B": O.spec_network_a2b(), "B2A": O.spec_network_b2a(), "D_A": O.spec_fs_discriminator("sum"), "D_B": O.spec_fs_discriminator("cat")}
    for k, n in nets.items():
        n.load_state_dict(O.make_state(specs[k], k, seed), strict=True)
        n.cuda().train()
    return nets


def fresh_step(fa, O, **kw):
    random.seed(1234)
    n = build_nets(fa, O)
    return fa.TrainStep(n["A2B"], n["B2A"], n["D_A"], n["D_B"], **kw)


def rel(a, b):
    return abs(a - b) / max(abs(b), 1e-12)


_cfg3_oracle = {}


@pytest.mark.parametrize("precision", ["bf16x3", "f16x2"])
def test_config3_b64_bf16x3_dwt3_vs_oracle(fa, O, precision):
    """BASELINE configs[2]: 256^2, batch 64, 16-bit-MFMA convolutions in their split forms (bf16x3: DESIGN 4.1b; f16x2, the headline
    arithmetic: 4.1g), 3-level Haar high-band L1 term.  Step-0 losses 1e-3 relative, per-network gradient norms 2e-3 against the fp32
    CPU oracle."""
    B, H = 64, 256
    a, b = O.synthetic_batch(B, H, seed=777)
    if not _cfg3_oracle:                 # (one CPU run of the batch-64 oracle step serves both precisions)
        torch.set_num_threads(host_threads())
        S = O.StepOracle(seed=0, whf_weight=0.5, dwt_levels=3)
        random.seed(4242)      # batch 64 overflows the 50-image replay buffer: the last 14 images draw from `random` (utils.py:41-50)
        _cfg3_oracle["L"] = S.train_step(a, b)
        _cfg3_oracle["g"] = S.grad_norms()
        del S
        gc.collect()
    Lo, go = _cfg3_oracle["L"], _cfg3_oracle["g"]
    ts = fresh_step(fa, O, precision=precision, whf_weight=0.5, dwt_levels=3, distributed=False)
    random.seed(4242)          # ... so both sides start the step from the same generator state
    L = ts.step(a.cuda(), b.cuda(), sync=True)
    for k in Lo:
        assert L[k] == pytest.approx(Lo[k], rel=1e-3, abs=2e-5), (k, L[k], Lo[k])
    assert L["loss_whf"] > 0
    gn = ts.grad_norms()
    for k in gn:
        assert gn[k] == pytest.approx(go[k], rel=2e-3), (k, gn[k], go[k])
    del ts
    gc.collect()
    torch.cuda.empty_cache()


def test_config4_rccl_shard_b32_and_run_to_run_spread(fa, O, rccl_world1):
    """BASELINE configs[3], one rank's share (32 images of the global 256): `TrainStep(distributed=True)` -- RCCL communicator
    created through faoctasr_comm_create, replica broadcast, faoctasr_grad_allreduce after each backward phase, AdamW with
    grad_scale 1/world -- against the plain step on the same inputs, and the plain step against itself.
    The step is not bit-reproducible: the split-K gather kernels of the narrow discriminator layers add their partial sums with
    fp32 atomics, so a forward activation moves by ~1e-7 from run to run, and the few that sit within that distance of a
    LeakyReLU / ReLU kink change side, which moves the gradients by a DISCRETE amount (tools/run_to_run.py at this batch: the
    generator arena is either 2e-6 or 3.7e-4 away from another run, the discriminator arena takes the values 6e-7 .. 1.8e-4; with
    FAOCTASR_CONV_NO_SPLIT_K on every gather call every run agrees to 5e-7, and a repeated backward of ONE forward graph
    to 2e-6: tools/probe/backward_repeat.py).  What is asserted is a bound on that spread -- step-0 losses 2e-5 relative, gradient
    arenas 1e-3 relative L2 -- for the run-to-run pair, and the SAME bound for the RCCL step, i.e. the exchange adds nothing."""
    B, H = 32, 256
    a, b = O.synthetic_batch(B, H, seed=99)
    a, b = a.cuda(), b.cuda()
    runs = []
    for mode in ("plain", "plain", "rccl"):
        ts = fresh_step(fa, O, distributed=(mode == "rccl"))
        if mode == "rccl":
            assert ts.comm is not None and ts.comm.ranks == 1 and ts.world == 1
        else:
            assert ts.comm is None
        L = ts.step(a, b, sync=True)
        runs.append((L, ts.opt_G.grad.clone(), ts.opt_D.grad.clone(), ts.opt_G.flat.clone(), ts.grad_norms()))
        if mode == "rccl":
            ts.comm.close()
        del ts
        torch.cuda.empty_cache()
    ref = runs[0]
    for name, other in (("run-to-run", runs[1]), ("rccl", runs[2])):
        for k in ref[0]:
            assert rel(other[0][k], ref[0][k]) < 2e-5, (name, k, other[0][k], ref[0][k])
        for i in (1, 2):
            d = float((other[i].double() - ref[i].double()).norm() / ref[i].double().norm())
            assert d < 1e-3, (name, "grad arena", i, d)
        for k in ref[4]:
            assert rel(other[4][k], ref[4][k]) < 5e-4, (name, k)
        # AdamW's first update is lr*sign(g): a weight may move by 2*lr where a ~0 gradient changes sign, never by more
        assert float((other[3] - ref[3]).abs().max()) <= 2.0 * 1.3e-4 * 1.01 + 1e-7, name


_cfg5_oracle = {}


def _config5_oracle(O):
    """The fp32 CPU oracle's step 0 at 512^2, batch 2, SSIM + 3-level wavelet-HF terms: computed once for both precisions."""
    if not _cfg5_oracle:
        kw = dict(ssim_weight=1.0, whf_weight=0.5, dwt_levels=3)
        batches = [O.synthetic_batch(2, 512, seed=31 + 5 * s) for s in range(2)]
        torch.set_num_threads(host_threads())
        S = O.StepOracle(seed=0, **kw)
        _cfg5_oracle["L"] = S.train_step(*batches[0])
        _cfg5_oracle["g"] = S.grad_norms()
        _cfg5_oracle["batches"] = batches
        del S
        gc.collect()
    return _cfg5_oracle


@pytest.mark.parametrize("precision", ["f32", "bf16x3", "f16x2"])
def test_config5_512_b2_ssim_dwt_graph_vs_oracle(fa, O, rccl_world1, precision):
    """BASELINE configs[4], one rank's share: 512x512, 2 images, SSIM + 3-level wavelet-HF terms, at the exact-fp32 precision and
    at "bf16x3" (the configuration's "bf16" convolutions in their fp32-parity form, DESIGN 4.1b: the first time igemm_bf16x3 /
    wgrad_x3 run on 512-wide maps and inside a captured graph).  (1) eager step vs the CPU oracle: step-0 losses 1e-3, gradient
    norms 2e-3; (2) the same step as a captured hipGraph with the RCCL exchange inside the capture (world 1): losses follow the
    eager run over two steps (step 0: 2e-4; step 1: cycle/identity 3e-3)."""
    kw = dict(ssim_weight=1.0, whf_weight=0.5, dwt_levels=3)
    ref = _config5_oracle(O)
    Lo, go = ref["L"], ref["g"]
    dev = [(x.cuda(), y.cuda()) for x, y in ref["batches"]]
    ts = fresh_step(fa, O, distributed=False, precision=precision, **kw)
    Le = []
    for s, (x, y) in enumerate(dev):
        Le.append(ts.step(x, y, sync=True))
        if s == 0:
            gn = ts.grad_norms()
    for k in Lo:
        assert Le[0][k] == pytest.approx(Lo[k], rel=1e-3, abs=2e-5), (k, Le[0][k], Lo[k])
    for k in gn:
        assert gn[k] == pytest.approx(go[k], rel=2e-3), (k, gn[k], go[k])
    del ts
    torch.cuda.empty_cache()
    tg = fresh_step(fa, O, distributed=True, precision=precision, **kw)
    gs = fa.GraphedTrainStep(tg, dev[0][0], dev[0][1])
    Lg = [gs.step(x, y, sync=True) for x, y in dev]
    tight = ("loss_G", "loss_cycle_ABA", "loss_cycle_BAB", "loss_idt", "loss_ssim", "loss_whf")
    for k in Le[0]:
        assert Lg[0][k] == pytest.approx(Le[0][k], rel=2e-4, abs=1e-6), (0, k, Lg[0][k], Le[0][k])
    for k in tight:
        assert Lg[1][k] == pytest.approx(Le[1][k], rel=3e-3), (1, k, Lg[1][k], Le[1][k])
    assert tg.opt_G.step_count == 2 and tg.opt_D.step_count == 2
    del gs
    tg.comm.close()
