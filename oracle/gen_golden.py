"""Generate the committed golden fixtures under tests/golden/ FROM THE REFERENCE.

Run in the build container only:  python -m oracle.gen_golden
It imports the reference's own modules (oracle/ref_shim.py), loads the
name-keyed deterministic weights (oracle/octa_oracle.make_state) into them with
``load_state_dict(strict=True)`` -- which pins every state_dict key and shape --
and records operator outputs, network outputs and the losses of a train step
written with the reference's objects following train.py:73-126,166-269.
Fixtures are data only (inputs are regenerated from seeds by the tests).
"""
import itertools
import json
import math
import os
import random
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import octa_oracle as O      # noqa: E402
from oracle import ref_shim              # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def stats(t):
    t = t.detach().double()
    return [float(t.mean()), float(t.std()), float(t.abs().max()), float(t.abs().mean())]


def crops(t):
    t = t.detach()
    return t[0, 0, :8, :8].numpy().copy(), t[-1, -1, -8:, -8:].numpy().copy()


def build_ref_nets(R, seed=0):
    nets = {"A2B": R.model.NetworkA2B(), "B2A": R.model.NetworkB2A(),
            "D_A": R.model.FS_DiscriminatorA(1), "D_B": R.model.FS_DiscriminatorB(1)}     # train.py:73-76
    specs = {"A2B": O.spec_network_a2b(), "B2A": O.spec_network_b2a(),
             "D_A": O.spec_fs_discriminator("sum"), "D_B": O.spec_fs_discriminator("cat")}
    for k, net in nets.items():
        net.load_state_dict(O.make_state(specs[k], k, seed), strict=True)
        net.train()
    return nets, specs


def ref_split(R, x, r_hp, r_lp):
    """train.py:173-175 per sample (identical to the reference at batch 1)."""
    hfs, lfs = [], []
    for b in range(x.shape[0]):
        hfs.append(R.utils.high_pass(x[b], i=r_hp))
        lfs.append(R.utils.low_pass(x[b], i=r_lp))
    hf = torch.stack(hfs).unsqueeze(1)
    lf = torch.stack(lfs).unsqueeze(1)
    return (hf + x) / 2.0, lf


def gen_ops(R):
    g = {}
    torch.manual_seed(7)
    # --- Haar DWT J=1 on arange (known answer in SURVEY Appendix A) and J=3 on random
    x8 = torch.arange(64, dtype=torch.float32).reshape(1, 1, 8, 8)
    f1 = R.DWTForward(J=1, wave="haar", mode="reflect")
    yl, yh = f1(x8)
    g["dwt8_ll"], g["dwt8_hi"] = yl.numpy(), yh[0].numpy()
    gen = torch.Generator().manual_seed(11)
    x16 = torch.randn(2, 3, 16, 16, generator=gen, requires_grad=True)
    f3 = R.DWTForward(J=3, wave="haar", mode="reflect")
    yl, yh = f3(x16)
    g["dwt16_ll"] = yl.detach().numpy()
    for j in range(3):
        g["dwt16_hi%d" % j] = yh[j].detach().numpy()
    cot = [torch.randn(t.shape, generator=gen) for t in [yl] + list(yh)]
    (sum((c * t).sum() for c, t in zip(cot, [yl] + list(yh)))).backward()
    g["dwt16_grad"] = x16.grad.numpy().copy()
    inv = R.DWTInverse(wave="haar", mode="reflect")
    cl = [c.clone().requires_grad_(True) for c in cot]
    rec = inv((cl[0], cl[1:]))
    g["idwt16_out"] = rec.detach().numpy()
    cot2 = torch.randn(rec.shape, generator=gen)
    (rec * cot2).sum().backward()
    g["idwt16_grad_ll"] = cl[0].grad.numpy().copy()
    g["idwt16_grad_hi0"] = cl[1].grad.numpy().copy()
    rec_none = inv((cot[0], [cot[1], None, cot[3]]))
    g["idwt16_none_out"] = rec_none.numpy()
    # --- FFT Gaussian split
    x64 = torch.rand(2, 1, 64, 64, generator=gen) * 2 - 1
    for r in (5, 8, 10, 14):
        g["hp64_r%d" % r] = torch.stack([R.utils.high_pass(x64[b], i=r) for b in range(2)]).numpy()
        g["lp64_r%d" % r] = torch.stack([R.utils.low_pass(x64[b], i=r) for b in range(2)]).numpy()
    xg = x64[0].clone().requires_grad_(True)
    w = torch.randn(64, 64, generator=gen)
    (R.utils.high_pass(xg, i=10) * w).sum().backward()
    g["hp64_r10_grad"] = xg.grad.numpy().copy()
    xg = x64[0].clone().requires_grad_(True)
    (R.utils.low_pass(xg, i=8) * w).sum().backward()
    g["lp64_r8_grad"] = xg.grad.numpy().copy()
    x192 = torch.rand(1, 192, 192, generator=gen) * 2 - 1
    for r in (5, 14):
        hp, lp = R.utils.high_pass(x192, i=r), R.utils.low_pass(x192, i=r)
        g["hp192_r%d_crop" % r], g["lp192_r%d_crop" % r] = hp[:16, :16].numpy(), lp[-16:, -16:].numpy()
        g["hp192_r%d_stats" % r], g["lp192_r%d_stats" % r] = np.array(stats(hp)), np.array(stats(lp))
    x33 = torch.rand(1, 30, 34, generator=gen) * 2 - 1      # non-square
    g["hp30x34_r4"], g["lp30x34_r10"] = R.utils.high_pass(x33).numpy(), R.utils.low_pass(x33).numpy()
    # --- SSIM
    a = (torch.rand(2, 1, 32, 32, generator=gen) * 2 - 1).requires_grad_(True)
    b = (a.detach() + 0.3 * torch.randn(2, 1, 32, 32, generator=gen)).clamp(-1, 1).requires_grad_(True)
    mod = R.ssim.SSIM()
    v = mod(a, b)
    v.backward()
    g["ssim32_mean"] = np.array(float(v))
    g["ssim32_grad1"], g["ssim32_grad2"] = a.grad.numpy().copy(), b.grad.numpy().copy()
    g["ssim32_per_sample"] = R.ssim.ssim(a.detach(), b.detach(), size_average=False).numpy()
    a3 = torch.rand(1, 3, 24, 40, generator=gen)
    b3 = torch.rand(1, 3, 24, 40, generator=gen)
    g["ssim_c3"] = np.array(float(R.ssim.ssim(a3, b3)))
    # --- BCE-with-logits: gradient through the target (train.py:99,230-231)
    xi = torch.randn(2, 4, 6, 6, generator=gen)
    tt = torch.randn(2, 4, 6, 6, generator=gen, requires_grad=True)
    l = torch.nn.BCEWithLogitsLoss()(xi, tt)
    l.backward()
    g["bce_val"], g["bce_tgrad"] = np.array(float(l)), tt.grad.numpy().copy()
    np.savez_compressed(os.path.join(OUT, "golden_ops.npz"), **g)
    print("ops:", len(g), "arrays")


def gen_nets(R, H=192, B=2):
    nets, specs = build_ref_nets(R)
    real_A, real_B = O.synthetic_batch(B, H)
    g = {}
    hf, lf = ref_split(R, real_A, 10, 8)
    g["hf_stats"], g["lf_stats"] = np.array(stats(hf)), np.array(stats(lf))
    with torch.no_grad():
        outs = nets["A2B"](lf, hf)
        for name, t in zip(("lf_feature", "hf_feature", "out"), outs):
            g["a2b_%s_stats" % name] = np.array(stats(t))
            g["a2b_%s_c0" % name], g["a2b_%s_c1" % name] = crops(t)
        outs = nets["B2A"](hf, lf)
        for name, t in zip(("hf_feature", "lf_feature", "out"), outs):
            g["b2a_%s_stats" % name] = np.array(stats(t))
            g["b2a_%s_c0" % name], g["b2a_%s_c1" % name] = crops(t)
        g["d_a"] = nets["D_A"](real_A).numpy()
        g["d_b"] = nets["D_B"](real_B).numpy()
    g["a2b_bn_rm"] = nets["A2B"].state_dict()["resnet.model.2.running_mean"].numpy().copy()
    g["a2b_bn_rv"] = nets["A2B"].state_dict()["shallow_up.model.2.running_var"].numpy().copy()
    g["d_a_bn_rm"] = nets["D_A"].state_dict()["net.model.3.running_mean"].numpy().copy()
    np.savez_compressed(os.path.join(OUT, "golden_nets_%d_b%d.npz" % (H, B)), **g)
    # state_dict key/shape listing (data; pins checkpoint compatibility)
    listing = {k: {kk: list(v.shape) for kk, v in n.state_dict().items()} for k, n in nets.items()}
    with open(os.path.join(OUT, "state_dict_spec.json"), "w") as f:
        json.dump(listing, f, indent=0, sort_keys=True)
    print("nets: done")


def gen_eval(R, H=192, B=2):
    """Inference recipe of utils.py:186,202-205 with the reference's own objects: ``model.eval()`` (BatchNorm2d on running
    statistics), high_pass / low_pass of the low-resolution image, third output of the generator.  The state is
    ``make_eval_state`` (non-trivial BatchNorm biases and running statistics), so the eval-mode arithmetic -- and the
    product's BatchNorm fold -- is exercised.  utils.eval only ever receives netG_A2B (radii 10, 8); B2A (radii 5, 14 of
    train.py:190) pins the same code path of the other generator."""
    g = {}
    lr_img, _ = O.synthetic_batch(B, H, seed=4711)
    for key, ctor, spec, radii in (("A2B", R.model.NetworkA2B, O.spec_network_a2b(), (10, 8)),
                                   ("B2A", R.model.NetworkB2A, O.spec_network_b2a(), (5, 14))):
        model = ctor()
        state = O.make_eval_state(spec, key, 0)
        model.load_state_dict(state, strict=True)
        model.eval()                                                        # utils.py:186
        outs = []
        with torch.no_grad():
            for b in range(B):                                              # utils.py:202-205, one image at a time as there
                img = lr_img[b:b + 1]
                hf = R.utils.high_pass(img[0], i=radii[0]).unsqueeze(0).unsqueeze(0)
                hf = (hf + img) / 2.0
                lf = R.utils.low_pass(img[0], i=radii[1]).unsqueeze(0).unsqueeze(0)
                outs.append(model(lf, hf)[2] if key == "A2B" else model(hf, lf)[2])
        sr = torch.cat(outs)
        k = key.lower()
        g["%s_eval_out_stats" % k] = np.array(stats(sr))
        g["%s_eval_out_c0" % k], g["%s_eval_out_c1" % k] = crops(sr)
        g["%s_eval_out_rows" % k] = sr[:, 0, H // 2, :].numpy().copy()      # one full row per image
        for name, v in model.state_dict().items():                          # eval mode leaves the running statistics untouched
            if name.endswith("running_mean") or name.endswith("running_var"):
                assert torch.equal(v, state[name]), name
    np.savez_compressed(os.path.join(OUT, "golden_eval_%d_b%d.npz" % (H, B)), **g)
    print("eval:", len(g), "arrays")


def ref_train_steps(R, H, B, n_steps, seed=0):
    """train.py:73-126 construction + train.py:166-269 loop body, using the
    reference's classes; per-sample split is the only extension (B>1)."""
    random.seed(1234)
    nets, _ = build_ref_nets(R, seed)
    G_A2B, G_B2A, D_A, D_B = nets["A2B"], nets["B2A"], nets["D_A"], nets["D_B"]
    crit_GAN, crit_cycle, crit_idt = torch.nn.MSELoss(), torch.nn.L1Loss(), torch.nn.L1Loss()
    crit_feat = torch.nn.BCEWithLogitsLoss()
    opt_G = torch.optim.AdamW(itertools.chain(G_A2B.parameters(), G_B2A.parameters()), lr=1.3e-4, betas=(0.9, 0.999))
    opt_D = torch.optim.AdamW(itertools.chain(D_A.parameters(), D_B.parameters()), lr=1.3e-4, betas=(0.9, 0.999))
    target_real, target_fake = torch.ones(B), torch.zeros(B)
    bufA, bufB = R.utils.ReplayBuffer(), R.utils.ReplayBuffer()
    b1, b2, b3, b4, b5 = 0.25, 10.0, 2.0, 0.5, 0.5
    rec = []
    for step in range(n_steps):
        real_A, real_B = O.synthetic_batch(B, H, seed=1234 + 17 * step)
        t0 = time.time()
        hf, lf = ref_split(R, real_A, 10, 8)
        lf_feature_A, hf_feature_A, fake_B = G_A2B(lf, hf)
        _, _, idt_A = G_B2A(hf, lf)
        hf_feature_A = hf_feature_A.detach()
        hf, lf = ref_split(R, fake_B, 5, 14)
        hf_feature_recovered_A, _, recovered_A = G_B2A(hf, lf)
        hf, lf = ref_split(R, real_B, 5, 14)
        hf_feature_B, lf_feature_B, fake_A = G_B2A(hf, lf)
        _, _, idt_B = G_A2B(lf, hf)
        hf_feature_B = hf_feature_B.detach()
        hf, lf = ref_split(R, fake_A, 10, 8)
        _, hf_feature_recovered_B, recovered_B = G_A2B(lf, hf)
        R.utils.set_requires_grad([D_A, D_B], False)
        opt_G.zero_grad()
        L = {}
        L["loss_GAN_A2B"] = crit_GAN(D_B(fake_B), target_real) * b4
        L["loss_GAN_B2A"] = crit_GAN(D_A(fake_A), target_real) * b5
        L["loss_cycle_ABA"] = crit_cycle(recovered_A, real_A) * b3 + crit_feat(hf_feature_A, hf_feature_recovered_A)
        L["loss_cycle_BAB"] = crit_cycle(recovered_B, real_B) * b3 + b1 * crit_feat(hf_feature_B, hf_feature_recovered_B)
        L["loss_idt"] = crit_idt(real_A, idt_A) * b2 + crit_idt(real_B, idt_B) * b2
        L["loss_G"] = L["loss_GAN_A2B"] + L["loss_GAN_B2A"] + L["loss_cycle_ABA"] + L["loss_cycle_BAB"] + L["loss_idt"]
        L["loss_G"].backward()
        gn = {}
        for k in ("A2B", "B2A"):
            gn[k] = math.sqrt(sum(float((p.grad.double() ** 2).sum()) for p in nets[k].parameters() if p.grad is not None))
        opt_G.step()
        R.utils.set_requires_grad([D_A, D_B], True)
        opt_D.zero_grad()
        fa = bufA.push_and_pop(fake_A)
        L["loss_D_A"] = (crit_GAN(D_A(real_A), target_real) + crit_GAN(D_A(fa.detach()), target_fake)) * 0.5
        L["loss_D_A"].backward()
        fb = bufB.push_and_pop(fake_B)
        L["loss_D_B"] = (crit_GAN(D_B(real_B), target_real) + crit_GAN(D_B(fb.detach()), target_fake)) * 0.5
        L["loss_D_B"].backward()
        for k in ("D_A", "D_B"):
            gn[k] = math.sqrt(sum(float((p.grad.double() ** 2).sum()) for p in nets[k].parameters() if p.grad is not None))
        opt_D.step()
        r = {k: float(v) for k, v in L.items()}
        r["grad_norm"] = gn
        r["psnr_recovered_A"] = O.psnr(recovered_A.detach(), real_A)
        r["fake_B_stats"] = stats(fake_B)
        r["recovered_A_stats"] = stats(recovered_A)
        r["seconds"] = time.time() - t0
        rec.append(r)
        print("  step", step, {k: round(v, 6) for k, v in r.items() if k.startswith("loss")}, "%.1fs" % r["seconds"])
    psum = {k: float(sum(p.detach().double().abs().sum() for p in n.parameters())) for k, n in nets.items()}
    live = {}
    for k, n in nets.items():
        live[k] = int(sum(p.numel() for p in n.parameters() if p.grad is not None))
    return {"H": H, "B": B, "steps": rec, "param_abs_sum": psum, "live_params": live}


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    R = ref_shim.load()
    if "--eval-only" not in sys.argv:
        gen_ops(R)
        gen_nets(R, 192, 2)
    gen_eval(R, 192, 2)
    if "--eval-only" in sys.argv:
        return
    res = []
    for (H, B, n) in ((192, 1, 3), (192, 2, 2), (256, 1, 2)):
        print("step fixtures H=%d B=%d" % (H, B))
        res.append(ref_train_steps(R, H, B, n))
    with open(os.path.join(OUT, "golden_step.json"), "w") as f:
        json.dump({"torch": torch.__version__, "configs": res}, f, indent=1)


if __name__ == "__main__":
    main()
