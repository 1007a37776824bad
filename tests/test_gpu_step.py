"""GPU parity of the networks and of the full train step (HIP path through the C ABI) against the
golden fixtures captured from the reference and against the CPU oracle on the same seeded inputs."""
import json
import os
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")

TIGHT = ("loss_G", "loss_cycle_ABA", "loss_cycle_BAB", "loss_idt")
LOOSE = ("loss_GAN_A2B", "loss_GAN_B2A", "loss_D_A", "loss_D_B")


def host_threads():
    """CPU threads this process may really use (the GPU box grants ~16 of the host's cores)."""
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


def build_nets(fa, O, seed=0):
    nets = {"A2B": fa.NetworkA2B(), "B2A": fa.NetworkB2A(), "D_A": fa.FS_DiscriminatorA(1), "D_B": fa.FS_DiscriminatorB(1)}
    specs = {"A2B": O.spec_network_a2b(), "B2A": O.spec_network_b2a(), "D_A": O.spec_fs_discriminator("sum"), "D_B": O.spec_fs_discriminator("cat")}
    for k, n in nets.items():
        n.load_state_dict(O.make_state(specs[k], k, seed), strict=True)
        n.cuda().train()
    return nets


def close(a, b, rtol=1e-3, atol=1e-4):
    a = a.detach().cpu().double().numpy() if torch.is_tensor(a) else np.asarray(a)
    np.testing.assert_allclose(a, np.asarray(b), rtol=rtol, atol=atol)


def test_networks_vs_golden(fa, O):
    g = np.load(os.path.join(GOLD, "golden_nets_192_b2.npz"))
    nets = build_nets(fa, O)
    real_A, real_B = O.synthetic_batch(2, 192)
    real_A, real_B = real_A.cuda(), real_B.cuda()
    with torch.no_grad():
        hf, lf = fa.frequency_split(real_A, 10, 8)
        for name, t in zip(("lf_feature", "hf_feature", "out"), nets["A2B"](lf, hf)):
            assert t.is_contiguous()
            close(t[0, 0, :8, :8], g["a2b_%s_c0" % name])
            close(t[-1, -1, -8:, -8:], g["a2b_%s_c1" % name])
            st = t.double()
            assert float(st.mean()) == pytest.approx(g["a2b_%s_stats" % name][0], rel=1e-3, abs=1e-4)
            assert float(st.std()) == pytest.approx(g["a2b_%s_stats" % name][1], rel=1e-3)
        for name, t in zip(("hf_feature", "lf_feature", "out"), nets["B2A"](hf, lf)):
            close(t[0, 0, :8, :8], g["b2a_%s_c0" % name])
            close(t[-1, -1, -8:, -8:], g["b2a_%s_c1" % name])
            assert float(t.double().std()) == pytest.approx(g["b2a_%s_stats" % name][1], rel=1e-3)
        close(nets["D_A"](real_A), g["d_a"], rtol=1e-3, atol=2e-4)
        close(nets["D_B"](real_B), g["d_b"], rtol=1e-3, atol=2e-4)
    close(nets["A2B"].state_dict()["resnet.model.2.running_mean"], g["a2b_bn_rm"], rtol=1e-3, atol=1e-5)
    close(nets["A2B"].state_dict()["shallow_up.model.2.running_var"], g["a2b_bn_rv"], rtol=1e-3, atol=1e-5)
    close(nets["D_A"].state_dict()["net.model.3.running_mean"], g["d_a_bn_rm"], rtol=1e-3, atol=1e-5)
    assert int(nets["A2B"].state_dict()["resnet.model.2.num_batches_tracked"]) == 1


def test_discriminator_minimum_size(fa, O):
    # the reference D raises below 192 px; so does this one, with torch's message
    d = fa.FS_DiscriminatorA(1).cuda()
    with pytest.raises(RuntimeError, match="Kernel size can't be greater"):
        d(torch.zeros(2, 1, 128, 128).cuda())


def _check_step(L, ref, step):
    """Tolerance policy (DESIGN.md "parity tolerance"): step 0 is well conditioned and every loss must agree to 1e-3
    relative.  Afterwards the trajectory is chaotic in fp32 (AdamW's first update is lr*sign(g)): the reference's OWN
    fp32 run deviates from an fp64 run of the same step by 1.3e-4 (step 1) and 5e-4 (step 2) on the cycle/identity
    terms and by 1e-2..3e-1 on the adversarial terms, so those get 1e-3 / 5e-3 relative and an absolute bound."""
    tight_rel = 1e-3 if step < 2 else 5e-3
    for k in TIGHT + (LOOSE if step == 0 else ()):
        assert L[k] == pytest.approx(ref[k], rel=tight_rel, abs=2e-5), (step, k, L[k], ref[k])
    if step > 0:
        for k in LOOSE:
            assert L[k] == pytest.approx(ref[k], abs=0.03 if step == 1 else 0.06), (step, k, L[k], ref[k])


@pytest.mark.parametrize("cfg", [0, 2])
def test_train_step_vs_golden(fa, O, cfg):
    """192^2 B=1 (3 steps) and 256^2 B=1 (2 steps): losses of the reference-object step."""
    with open(os.path.join(GOLD, "golden_step.json")) as f:
        gold = json.load(f)["configs"][cfg]
    H, B = gold["H"], gold["B"]
    random.seed(1234)
    n = build_nets(fa, O)
    ts = fa.TrainStep(n["A2B"], n["B2A"], n["D_A"], n["D_B"])
    assert {"A2B": sum(p.numel() for _, p in fa.live_parameters(n["A2B"])), "B2A": sum(p.numel() for _, p in fa.live_parameters(n["B2A"])),
            "D_A": sum(p.numel() for p in n["D_A"].parameters()), "D_B": sum(p.numel() for p in n["D_B"].parameters())} == gold["live_params"]
    for step in range(len(gold["steps"])):
        a, b = O.synthetic_batch(B, H, seed=1234 + 17 * step)
        L = ts.step(a.cuda(), b.cuda(), sync=True, keep=True)
        ref = gold["steps"][step]
        _check_step(L, ref, step)
        if step == 0:
            gn = ts.grad_norms()
            for k in gn:
                assert gn[k] == pytest.approx(ref["grad_norm"][k], rel=2e-3), (k, gn[k], ref["grad_norm"][k])
            assert float(fa.psnr(L["tensors"]["recovered_A"], a.cuda())) == pytest.approx(ref["psnr_recovered_A"], rel=1e-3)
            fb = L["tensors"]["fake_B"].double()
            assert float(fb.std()) == pytest.approx(ref["fake_B_stats"][1], rel=1e-3)


def test_train_step_batched_vs_oracle(fa, O):
    """B=2 at 192^2 (per-sample split semantics) against the CPU oracle AND the golden reference-object run."""
    with open(os.path.join(GOLD, "golden_step.json")) as f:
        gold = json.load(f)["configs"][1]
    assert gold["B"] == 2
    random.seed(1234)
    n = build_nets(fa, O)
    ts = fa.TrainStep(n["A2B"], n["B2A"], n["D_A"], n["D_B"])
    S = O.StepOracle(seed=0)
    torch.set_num_threads(host_threads())
    for step in range(2):
        a, b = O.synthetic_batch(2, 192, seed=1234 + 17 * step)
        L = ts.step(a.cuda(), b.cuda(), sync=True)
        Lo = S.train_step(a, b)
        _check_step(L, gold["steps"][step], step)
        _check_step(L, Lo, step)
        if step == 0:
            gn, go = ts.grad_norms(), S.grad_norms()
            for k in gn:
                assert gn[k] == pytest.approx(go[k], rel=2e-3), k


def test_extension_terms_vs_oracle(fa, O):
    """opt-in SSIM + 3-level wavelet-HF terms (BASELINE configs 3/5): step-0 losses vs the oracle."""
    random.seed(1234)
    n = build_nets(fa, O)
    ts = fa.TrainStep(n["A2B"], n["B2A"], n["D_A"], n["D_B"], ssim_weight=1.0, whf_weight=0.5, dwt_levels=3)
    S = O.StepOracle(seed=0, ssim_weight=1.0, whf_weight=0.5, dwt_levels=3)
    a, b = O.synthetic_batch(1, 192)
    L = ts.step(a.cuda(), b.cuda(), sync=True)
    Lo = S.train_step(a, b)
    for k in ("loss_G", "loss_ssim", "loss_whf", "loss_cycle_ABA", "loss_idt"):
        assert L[k] == pytest.approx(Lo[k], rel=1e-3), (k, L[k], Lo[k])
    gn, go = ts.grad_norms(), S.grad_norms()
    for k in ("A2B", "B2A"):
        assert gn[k] == pytest.approx(go[k], rel=2e-3), k
    # second step: the multi-stream schedule (the opt-in terms are split over the two generator chains)
    a, b = O.synthetic_batch(1, 192, seed=4321)
    L = ts.step(a.cuda(), b.cuda(), sync=True)
    Lo = S.train_step(a, b)
    _check_step(L, Lo, 1)
    for k in ("loss_ssim", "loss_whf"):
        assert L[k] == pytest.approx(Lo[k], rel=1e-3), (k, L[k], Lo[k])


def test_inference_path_eval_mode(fa, O):
    """generator forward with BatchNorm on running statistics (`model.eval()`, utils.py:186) vs the oracle."""
    nets = build_nets(fa, O)
    a, _ = O.synthetic_batch(2, 192, seed=5)
    # make the running statistics non-trivial: one training forward on both sides
    st = O.make_state(O.spec_network_a2b(), "A2B", 0)
    hf_o, lf_o = O.freq_split(a, 10, 8)
    with torch.no_grad():
        O.network_a2b(O.Net(st, train=True), lf_o, hf_o)
        hf, lf = fa.frequency_split(a.cuda(), 10, 8)
        nets["A2B"](lf, hf)
        ref = O.network_a2b(O.Net(st, train=False), lf_o, hf_o)[2]
    out = fa.super_resolve(nets["A2B"], a.cuda())
    assert not nets["A2B"].training
    close(out, ref, rtol=1e-3, atol=2e-4)
    # the same forward with gradients enabled takes the un-folded path (BatchNorm eval kernels): folding changes nothing beyond rounding
    nets["A2B"].eval()
    hf, lf = fa.frequency_split(a.cuda(), 10, 8)
    unfolded = nets["A2B"](lf, hf)[2]
    close(out, unfolded.detach().cpu().numpy(), rtol=1e-4, atol=2e-5)
    m = fa.evaluate_pairs(nets["A2B"], [(a[:1].cuda(), a[1:].cuda())])
    assert set(m) == {"psnr", "ssim", "mse", "nmi"} and m["mse"] > 0
    # the forward is not bit-reproducible (split-K atomics), and NMI's histogram reacts to single pixels: score ONE output on both sides
    yd = fa.super_resolve(nets["A2B"], a[:1].cuda())
    md = fa.image_metrics(yd, a[1:].cuda()).cpu().numpy()[0]
    y1, g1 = yd.cpu().numpy()[0, 0], a[1].numpy()[0]
    assert md[0] == pytest.approx(O.skimage_psnr(y1, g1), rel=1e-6)
    assert md[2] == pytest.approx(O.skimage_mse(y1, g1), rel=1e-6)
    assert md[1] == pytest.approx(O.skimage_ssim(y1, g1), rel=1e-5, abs=1e-6)
    assert md[3] == pytest.approx(O.skimage_nmi(y1, g1), rel=1e-9)
    assert m["psnr"] == pytest.approx(md[0], rel=1e-4) and m["ssim"] == pytest.approx(md[1], rel=1e-4) and m["nmi"] == pytest.approx(md[3], rel=1e-4)
    # LR schedule hook (train.py:105-110): linear decay to 0 after decay_epoch
    ts = fa.TrainStep(nets["A2B"], nets["B2A"], nets["D_A"], nets["D_B"])
    ts.lr_step(1.3e-4, fa.LambdaLR(50, 0, 10).step, 30)
    assert ts.opt_G.lr == pytest.approx(0.65e-4) and ts.opt_D.lr == pytest.approx(0.65e-4)


def test_super_resolve_vs_reference_eval_fixture(fa, O):
    """SURVEY 8f-2, generator half, pinned by the REFERENCE: `super_resolve` (frequency split + eval-mode generator with BatchNorm
    folded into the convolutions) against outputs of the reference's NetworkA2B / NetworkB2A under `model.eval()`
    (utils.py:186,202-205; tests/golden/golden_eval_192_b2.npz from oracle/gen_golden.gen_eval).  Also the un-folded eval path
    (BatchNorm eval kernels, taken when gradients are enabled) and that eval mode leaves the running statistics alone."""
    g = np.load(os.path.join(GOLD, "golden_eval_192_b2.npz"))
    lr_img, _ = O.synthetic_batch(2, 192, seed=4711)
    x = lr_img.cuda()
    for key, cls, spec, radii in (("A2B", fa.NetworkA2B, O.spec_network_a2b(), (10, 8)), ("B2A", fa.NetworkB2A, O.spec_network_b2a(), (5, 14))):
        state = O.make_eval_state(spec, key, 0)
        net = cls()
        net.load_state_dict(state, strict=True)
        net.cuda()
        k = key.lower()
        if key == "A2B":
            out = fa.super_resolve(net, x)
        else:
            net.eval()
            with torch.no_grad():
                hf, lf = fa.frequency_split(x, *radii)
                out = net(hf, lf)[2]
        assert not net.training
        o = out.cpu()
        close(o[0, 0, :8, :8], g["%s_eval_out_c0" % k], rtol=1e-3, atol=2e-4)
        close(o[-1, -1, -8:, -8:], g["%s_eval_out_c1" % k], rtol=1e-3, atol=2e-4)
        close(o[:, 0, 96, :], g["%s_eval_out_rows" % k], rtol=1e-3, atol=2e-4)
        d = o.double()
        close(np.array([float(d.mean()), float(d.std()), float(d.abs().max()), float(d.abs().mean())]), g["%s_eval_out_stats" % k], rtol=1e-3, atol=1e-5)
        hf, lf = fa.frequency_split(x, *radii)                  # gradients enabled: BatchNorm eval kernels instead of the fold
        unfolded = net(lf, hf)[2] if key == "A2B" else net(hf, lf)[2]
        close(unfolded.detach().cpu(), o.numpy(), rtol=1e-4, atol=2e-5)
        for name, v in net.state_dict().items():
            if name.endswith("running_mean") or name.endswith("running_var"):
                assert torch.equal(v.cpu(), state[name]), name


@pytest.mark.parametrize("cfg", [2, 0])
def test_train_step_bf16x3_precision(fa, O, cfg):
    """Opt-in precision "bf16x3" (conv forward / input gradient on hi/lo-split bf16 operands, 3 MFMAs per product): the
    step-0 losses and gradient norms still meet the fp32 parity bar against the REFERENCE fixtures at 256^2 (measured:
    losses <= 1e-4, gradient norms <= 2.3e-4 from the exact-f32 path at 192^2 B=2, 256^2 B=1, 256^2 B=4).
    The 192^2 batch-1 fixture is degenerate -- the deepest wavelet-branch BatchNorm of D_B normalises over 4 values and
    amplifies any 4e-6 conv perturbation ~1000x into A2B's adversarial gradient -- so there only the losses keep 1e-3 and
    the gradient norms get 1e-2."""
    with open(os.path.join(GOLD, "golden_step.json")) as f:
        gold = json.load(f)["configs"][cfg]
    random.seed(1234)
    n = build_nets(fa, O)
    ts = fa.TrainStep(n["A2B"], n["B2A"], n["D_A"], n["D_B"], precision="bf16x3")
    a, b = O.synthetic_batch(gold["B"], gold["H"], seed=1234)
    L = ts.step(a.cuda(), b.cuda(), sync=True)
    _check_step(L, gold["steps"][0], 0)
    gn = ts.grad_norms()
    for k in gn:
        assert gn[k] == pytest.approx(gold["steps"][0]["grad_norm"][k], rel=2e-3 if cfg == 2 else 1e-2), (k, gn[k])
    assert fa.ops.conv_precision == 0          # the mode is scoped to the step


@pytest.mark.parametrize("cfg", [2, 0, 1])
def test_train_step_f16x2_precision(fa, O, cfg):
    """precision "f16x2" (every wide-map convolution's three GEMMs on fp16 hi/lo-split, power-of-two-scaled operands; per-layer
    error at or below the exact-f32 kernels': test_conv2d_f16x2): the step meets the SAME bars against the REFERENCE fixtures as
    the exact-f32 step (test_train_step_vs_golden) -- step-0 losses 1e-3, gradient norms 2e-3 -- on all three fixture configs,
    over all their steps for the losses."""
    with open(os.path.join(GOLD, "golden_step.json")) as f:
        gold = json.load(f)["configs"][cfg]
    random.seed(1234)
    n = build_nets(fa, O)
    ts = fa.TrainStep(n["A2B"], n["B2A"], n["D_A"], n["D_B"], precision="f16x2")
    for s, gs in enumerate(gold["steps"]):
        a, b = O.synthetic_batch(gold["B"], gold["H"], seed=1234 + 17 * s)
        L = ts.step(a.cuda(), b.cuda(), sync=True)
        _check_step(L, gs, s)
        if s == 0:
            gn = ts.grad_norms()
            for k in gn:
                assert gn[k] == pytest.approx(gs["grad_norm"][k], rel=2e-3), (k, gn[k])
    assert fa.ops.conv_precision == 0          # the mode is scoped to the step


def test_step_gradients_vs_fp64_oracle_f16x2_beside_f32(fa, O):
    """What the f16x2 arithmetic does to the WHOLE step, measured against an fp64 run of the oracle (192^2, batch 2, step 0: the
    reference's algorithm in double precision, same weights and data): the relative L2 error of each network's full gradient (every
    live parameter, matched by state_dict name) and the losses, for the exact-f32 step and for the f16x2 step.  The split
    contraction must not cost accuracy: the four gradient errors together are held to 2.5x the exact-f32 step's (+ 1e-3), each to 2e-2,
    and the losses of both to 1e-4 (the parity bar is 1e-3).  One run: f16x2 at or below exact-f32 on all four networks.  The numbers of one
    run are kept in profiles/r04_step_error_vs_fp64.txt."""
    random.seed(1234)
    a, b = O.synthetic_batch(2, 192, seed=1234)
    torch.set_num_threads(host_threads())
    S = O.StepOracle(seed=0, dtype=torch.float64)
    L64 = S.train_step(a, b)
    g64 = {k: {name: S.state[k][name].grad for name in S.state[k] if getattr(S.state[k][name], "grad", None) is not None} for k in S.state}
    err = {}
    for prec in ("f32", "f16x2"):
        random.seed(1234)
        n = build_nets(fa, O)
        ts = fa.TrainStep(n["A2B"], n["B2A"], n["D_A"], n["D_B"], precision=prec)
        L = ts.step(a.cuda(), b.cuda(), sync=True)
        torch.cuda.synchronize()
        e = {}
        for k, net in n.items():
            num = den = 0.0
            for name, p in fa.live_parameters(net):
                ref = g64[k][name]
                d = p.grad.detach().cpu().double() - ref
                num += float((d * d).sum())
                den += float((ref * ref).sum())
            e["grad_" + k] = (num / den) ** 0.5
        for key in L64:
            e[key] = abs(L[key] - L64[key]) / max(abs(L64[key]), 1e-12)
        err[prec] = e
    for key in sorted(err["f32"]):
        print("step-vs-fp64 %-18s f32 %.3e   f16x2 %.3e" % (key, err["f32"][key], err["f16x2"][key]))
    # Each network's gradient error is a few ReLU / LeakyReLU masks that fall on the other side than in the fp64 run -- discrete events,
    # different ones in every run (D_B, whose deepest BatchNorm normalises over 8 values here, has been seen at 3e-4 and at 5e-3 under BOTH
    # arithmetics) -- so the comparison is made on the four networks together, and each one is held to an absolute bound
    tot = {p: sum(v for k, v in err[p].items() if k.startswith("grad_")) for p in err}
    assert tot["f16x2"] <= 2.5 * tot["f32"] + 1e-3, tot
    for key, e32 in err["f32"].items():
        e16 = err["f16x2"][key]
        if key.startswith("grad_"):
            assert e16 < 2e-2 and e32 < 2e-2, (key, e32, e16)
        else:
            assert e16 < 1e-4 and e32 < 1e-4, (key, e32, e16)


def test_ten_step_trajectory_f16x2_tracks_the_oracle_like_exact_f32(fa, O):
    """Ten consecutive train steps (192^2, batch 2, fresh data every step) from the same initial state: the fp32 CPU oracle, the
    exact-f32 HIP step and the f16x2 HIP step.  The trajectory is chaotic (AdamW's first updates are lr * sign(g): DESIGN.md section 2),
    so all three drift apart (measured: both HIP runs are 1e-4 from the oracle at step 2 and ~1.3e-2 at steps 7-9); the claim is that
    f16x2 drifts from the oracle no faster than the exact-f32 kernels do.  Per step, the well-conditioned totals (loss_G and its cycle /
    identity parts) of both HIP runs stay within 5e-2 relative of the oracle, and the f16x2 run's deviation averaged over steps 2-9 is held
    to 2.5x the exact-f32 run's (+ 1e-3; measured 0.9-1.3x).  One run's table: profiles/r04_ten_step_trajectory.txt."""
    steps = 10
    torch.set_num_threads(host_threads())
    random.seed(1234)
    S = O.StepOracle(seed=0)
    batches = [O.synthetic_batch(2, 192, seed=4000 + 13 * i) for i in range(steps)]
    ref = [S.train_step(a, b) for a, b in batches]
    keys = ("loss_G", "loss_cycle_ABA", "loss_cycle_BAB", "loss_idt")
    dev = {}
    for prec in ("f32", "f16x2"):
        random.seed(1234)
        n = build_nets(fa, O)
        ts = fa.TrainStep(n["A2B"], n["B2A"], n["D_A"], n["D_B"], precision=prec)
        out = [ts.step(a.cuda(), b.cuda(), sync=True) for a, b in batches]
        dev[prec] = [[abs(out[i][k] - ref[i][k]) / abs(ref[i][k]) for k in keys] for i in range(steps)]
        del ts
    for i in range(steps):
        print("trajectory step %d  " % i + "  ".join("%s f32 %.1e f16x2 %.1e" % (k, dev["f32"][i][j], dev["f16x2"][i][j]) for j, k in enumerate(keys)))
    worst = {p: max(max(r) for r in dev[p]) for p in dev}
    mean = {p: sum(max(r) for r in dev[p][2:]) / (steps - 2) for p in dev}       # steps 2..9: the drift, averaged (a single step's value is spiky)
    assert worst["f32"] < 5e-2 and worst["f16x2"] < 5e-2, worst
    assert mean["f16x2"] <= 2.5 * mean["f32"] + 1e-3, mean


def test_graph_captured_step_matches_eager_and_golden(fa, O):
    """SURVEY 8f-1 / BASELINE config 5: the whole step as one captured hipGraph (device-side replay buffer, AdamW scalars in device
    memory).  Same seeds -> the graph's losses follow the eager step and the reference fixture; capturing does not advance the
    training state (weights, moments, BN statistics, replay history, RNG)."""
    with open(os.path.join(GOLD, "golden_step.json")) as f:
        gold = json.load(f)["configs"][1]                  # 192^2 B=2, 2 steps
    H, B = gold["H"], gold["B"]
    batches = [tuple(t.cuda() for t in O.synthetic_batch(B, H, seed=1234 + 17 * s)) for s in range(3)]

    random.seed(1234)
    n = build_nets(fa, O)
    eager = fa.TrainStep(n["A2B"], n["B2A"], n["D_A"], n["D_B"])
    Le = [eager.step(a, b, sync=True) for a, b in batches]

    random.seed(1234)
    n2 = build_nets(fa, O)
    ts = fa.TrainStep(n2["A2B"], n2["B2A"], n2["D_A"], n2["D_B"])
    w0 = ts.opt_G.flat.clone()
    st = random.getstate()
    gs = fa.GraphedTrainStep(ts, batches[0][0], batches[0][1])
    assert torch.equal(ts.opt_G.flat, w0) and ts.opt_G.step_count == 0 and random.getstate() == st
    assert all(int(v) == 0 for k, v in n2["A2B"].state_dict().items() if k.endswith("num_batches_tracked"))
    Lg = [gs.step(a, b, sync=True) for a, b in batches]
    for s in range(2):
        _check_step(Lg[s], gold["steps"][s], s)
    for s in range(3):
        for k in Le[s]:
            tol = 2e-4 if s == 0 else (3e-3 if k in TIGHT else None)
            if tol is not None:
                assert Lg[s][k] == pytest.approx(Le[s][k], rel=tol, abs=1e-6), (s, k, Lg[s][k], Le[s][k])
            else:
                assert Lg[s][k] == pytest.approx(Le[s][k], abs=0.03 if s == 1 else 0.06), (s, k)
    assert ts.opt_G.step_count == 3 and ts.opt_D.step_count == 3
    # BatchNorm call counters advanced as in the eager run
    sd_e, sd_g = n["A2B"].state_dict(), n2["A2B"].state_dict()
    for k in sd_e:
        if k.endswith("num_batches_tracked"):
            assert int(sd_e[k]) == int(sd_g[k]), k
    # the discriminators' BatchNorm running statistics: their frozen pass (generator phase) and their update phase run on different
    # streams and both do a plain read-modify-write of running_mean / running_var -- the schedule orders them; a lost
    # update would remove one of the nine momentum-weighted contributions of a running mean (>= 7 % of its value); three steps of trajectory
    # drift between the two runs move the means by ~2e-3 relative.  (The running VARIANCES of the deepest wavelet-branch layers -- statistics
    # over 8 values at this size -- drift by several per cent themselves, as much as a lost update would move them: they get a loose bound.)
    for key in ("D_A", "D_B"):
        sd_e, sd_g = n[key].state_dict(), n2[key].state_dict()
        for k in sd_e:
            if k.endswith("running_mean"):
                assert float((sd_e[k] - sd_g[k]).abs().max()) <= 3e-2 * float(sd_e[k].abs().max()) + 1e-6, (key, k)
            elif k.endswith("running_var"):
                assert float((sd_e[k] - sd_g[k]).abs().max()) <= 2e-1 * float(sd_e[k].abs().max()), (key, k)
    # replay histories hold the same images (up to step-to-step rounding drift)
    assert len(ts.fake_A_buffer.data) == len(eager.fake_A_buffer.data)
    close(ts.fake_A_buffer.data[0], eager.fake_A_buffer.data[0].cpu().numpy(), rtol=1e-3, atol=1e-4)


def test_train_step_bench_workload_vs_oracle(fa, O):
    """BASELINE configs[1] at FULL size -- 256x256, batch 8, the exact shapes bench.py times (Winograd on the 3x3 layers, split-K
    on the narrow maps) -- against the CPU oracle on the same seeded inputs: step-0 losses to 1e-3, gradient norms to 2e-3,
    PSNR of the cycle reconstruction to 1e-3; and the same step with precision="f32_direct" (no Winograd) agrees as well."""
    random.seed(1234)
    torch.set_num_threads(host_threads())
    a, b = O.synthetic_batch(8, 256, seed=4321)
    S = O.StepOracle(seed=0)
    Lo = S.train_step(a, b, keep=True)
    go = S.grad_norms()
    psnr_o = float(O.psnr(Lo["tensors"]["recovered_A"], a))
    for prec in ("f32", "f32_direct"):
        random.seed(1234)
        n = build_nets(fa, O)
        ts = fa.TrainStep(n["A2B"], n["B2A"], n["D_A"], n["D_B"], precision=prec)
        L = ts.step(a.cuda(), b.cuda(), sync=True, keep=True)
        _check_step(L, Lo, 0)
        gn = ts.grad_norms()
        for k in gn:
            assert gn[k] == pytest.approx(go[k], rel=2e-3), (prec, k, gn[k], go[k])
        assert float(fa.psnr(L["tensors"]["recovered_A"], a.cuda())) == pytest.approx(psnr_o, rel=1e-3)


@pytest.mark.gpu
def test_train_step_non_square_vs_oracle(fa, O):
    """A ragged input: 192 x 256 images, batch 3 (odd), two steps against the CPU oracle -- the frequency split with different row /
    column circulants, Haar levels, every convolution route and the BatchNorm kernels on non-square maps, and from step 1 on the
    full stream schedule (batched packs, discriminator work on branch streams)."""
    random.seed(1234)
    n = build_nets(fa, O)
    ts = fa.TrainStep(n["A2B"], n["B2A"], n["D_A"], n["D_B"])
    S = O.StepOracle(seed=0)
    torch.set_num_threads(host_threads())
    g = torch.Generator().manual_seed(77)
    for step in range(2):
        a = torch.rand(3, 1, 192, 256, generator=g) * 2 - 1
        b = torch.rand(3, 1, 192, 256, generator=g) * 2 - 1
        L = ts.step(a.cuda(), b.cuda(), sync=True)
        Lo = S.train_step(a, b)
        _check_step(L, Lo, step)
        if step == 0:
            gn, go = ts.grad_norms(), S.grad_norms()
            for k in gn:
                assert gn[k] == pytest.approx(go[k], rel=2e-3), k


@pytest.mark.gpu
def test_stream_schedule_equals_single_stream(fa, O):
    """The multi-stream schedule changes when kernels run, not what they compute: three steps of it against three
    steps on one stream, same state and inputs.  Step-0 losses agree to 2e-5 (run-to-run noise); afterwards both follow the same
    trajectory within the step>=1 policy of ``_check_step``; the small-batch default (one stream below 2 x 256^2 pixels) is the
    single-stream path itself."""
    runs = {}
    for overlap in (True, False):
        random.seed(1234)
        n = build_nets(fa, O)
        ts = fa.TrainStep(n["A2B"], n["B2A"], n["D_A"], n["D_B"], overlap_wgrad=overlap)
        out = []
        for step in range(3):
            a, b = O.synthetic_batch(2, 192, seed=99 + step)
            out.append(ts.step(a.cuda(), b.cuda(), sync=True))
        runs[overlap] = out
    for k, v in runs[False][0].items():
        assert runs[True][0][k] == pytest.approx(v, rel=2e-5, abs=1e-7), k
    for step in (1, 2):
        _check_step(runs[True][step], runs[False][step], step)
    default = fa.TrainStep.overlap_min_pixels
    try:
        fa.TrainStep.overlap_min_pixels = 2 * 256 * 256
        random.seed(1234)
        n = build_nets(fa, O)
        ts = fa.TrainStep(n["A2B"], n["B2A"], n["D_A"], n["D_B"])
        a, b = O.synthetic_batch(2, 192, seed=99)
        for _ in range(2):
            ts.step(a.cuda(), b.cuda())
        assert len(ts._pack_plans) == 1           # batched packs are independent of the stream schedule
    finally:
        fa.TrainStep.overlap_min_pixels = default


def test_trailing_partial_batch_under_stream_schedule(fa, O):
    """The reference's DataLoader has no drop_last (train.py:142-146): an epoch ends in a smaller batch.  Packed-weight images
    are per (layer, N, H, W), so that step's images are not in the full batches' pack plan and get packed inside the convolution
    calls, on whichever stream reaches a layer first; readers on other streams are ordered behind that launch by events
    (ops._wpack / ops._packed; ADVICE r2: chain A's backward on one stream and an identity pass's backward on another read the
    same input-gradient image, which used to be an un-ordered torch.empty buffer for one of them).  Steps at batch 4, 4, 3, 4 with
    the multi-stream schedule against the single-stream one.  lr = 0 keeps the weights (not their packed images: AdamW still bumps
    the version, so every step repacks) identical on both sides at every step, and ``reproducible_forward`` makes the forward
    bit-reproducible, so each step's losses agree to 2e-5 and the gradient arenas to 1e-4 relative L2."""
    part = tuple(t[:3].contiguous() for t in O.synthetic_batch(4, 192, seed=60))
    batches = [O.synthetic_batch(4, 192, seed=50), O.synthetic_batch(4, 192, seed=51), part, O.synthetic_batch(4, 192, seed=61)]
    runs = {}
    for overlap in (False, True):
        random.seed(1234)
        n = build_nets(fa, O)
        ts = fa.TrainStep(n["A2B"], n["B2A"], n["D_A"], n["D_B"], lr=0.0, overlap_wgrad=overlap, reproducible_forward=True)
        out = []
        for a, b in batches:
            L = ts.step(a.cuda(), b.cuda(), sync=True)
            out.append((L, ts.opt_G.grad.clone(), ts.opt_D.grad.clone()))
        assert len(ts._pack_plans) == 2                      # one per batch shape; the partial batch did not evict the full batches' plan
        runs[overlap] = out
        del ts
    for i, ((L1, g1, d1), (L0, g0, d0)) in enumerate(zip(runs[True], runs[False])):
        for k, v in L0.items():
            assert L1[k] == pytest.approx(v, rel=2e-5, abs=1e-7), (i, k, L1[k], v)
        for x, y in ((g1, g0), (d1, d0)):
            assert float((x.double() - y.double()).norm() / y.double().norm()) < 1e-4, i


@pytest.mark.parametrize("precision", ["f32", "f16x2"])
def test_capturable_chain_arrangement_eager_vs_single_stream(fa, O, precision):
    """the arrangement of the two-chain schedule that a hipGraph capture uses (chain A on the caller's stream,
    ``eager_chain_A_forked = False``) was only ever exercised through the captured graph.  Here it runs EAGERLY, multi-stream, against
    the single-stream step: lr = 0 and ``reproducible_forward`` as in the trailing-batch test, two steps; losses to 2e-5, gradient
    arenas to 1e-4, and every BatchNorm buffer of all four networks (running statistics are read-modify-written by each pass: a missing
    event between two passes of one network shows as a lost update)."""
    batches = [O.synthetic_batch(4, 192, seed=70), O.synthetic_batch(4, 192, seed=71)]
    runs = {}
    keep = fa.TrainStep.eager_chain_A_forked
    try:
        for overlap in (False, True):
            fa.TrainStep.eager_chain_A_forked = False
            random.seed(1234)
            n = build_nets(fa, O)
            ts = fa.TrainStep(n["A2B"], n["B2A"], n["D_A"], n["D_B"], lr=0.0, overlap_wgrad=overlap, reproducible_forward=True, precision=precision)
            ts.overlap_min_pixels = 0
            out = []
            for a, b in batches:
                L = ts.step(a.cuda(), b.cuda(), sync=True)
                out.append((L, ts.opt_G.grad.clone(), ts.opt_D.grad.clone()))
            bufs = {k + "." + name: v.clone() for k, net in n.items() for name, v in net.state_dict().items() if "running_" in name}
            runs[overlap] = (out, bufs)
            del ts
    finally:
        fa.TrainStep.eager_chain_A_forked = keep
    for i, ((L1, g1, d1), (L0, g0, d0)) in enumerate(zip(runs[True][0], runs[False][0])):
        for k, v in L0.items():
            assert L1[k] == pytest.approx(v, rel=2e-5, abs=1e-7), (i, k, L1[k], v)
        for x, y in ((g1, g0), (d1, d0)):
            assert float((x.double() - y.double()).norm() / y.double().norm()) < 1e-4, i
    for k, v in runs[False][1].items():
        assert float((runs[True][1][k] - v).abs().max()) <= 1e-5 * float(v.abs().max()) + 1e-7, k


@pytest.mark.gpu
def test_reproducible_forward_mode(fa, O):
    """``TrainStep(reproducible_forward=True)``: forward convolutions carry FAOCTASR_CONV_NO_SPLIT_K, so two runs from the same
    state produce bit-identical forward tensors and losses, and gradient arenas that differ only by atomic summation order
    (<= 1e-5 relative L2; the default mode shows up to 5e-3 at batch 8 through LeakyReLU / ReLU kink flips).
    The mode changes no value beyond that: its step-0 losses equal the default mode's to 2e-5."""
    a, b = O.synthetic_batch(4, 256, seed=31)
    a, b = a.cuda(), b.cuda()
    runs = []
    for mode in (True, True, False):
        random.seed(1234)
        n = build_nets(fa, O)
        ts = fa.TrainStep(n["A2B"], n["B2A"], n["D_A"], n["D_B"], reproducible_forward=mode)
        L = ts.step(a, b, sync=True, keep=True)
        runs.append((L, ts.opt_G.grad.clone(), ts.opt_D.grad.clone()))
    (L0, g0, d0), (L1, g1, d1), (L2, _, _) = runs
    for k in ("fake_A", "fake_B", "recovered_A", "recovered_B", "idt_A", "idt_B"):
        assert torch.equal(L0["tensors"][k], L1["tensors"][k]), k
    for k, v in L0.items():
        if k != "tensors":
            assert v == L1[k], k                                          # bit-identical losses, discriminator losses included
            assert v == pytest.approx(L2[k], rel=2e-5, abs=1e-7), k
    for x, y in ((g0, g1), (d0, d1)):
        assert float((x.double() - y.double()).norm() / y.double().norm()) < 1e-5
