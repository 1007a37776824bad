"""Train step in f32 vs bf16x3 precision from identical weights/inputs: relative differences of losses and gradient norms."""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import faoctasr
from oracle import octa_oracle as O
def build(seed=0):
    nets = {"A2B": faoctasr.NetworkA2B(), "B2A": faoctasr.NetworkB2A(), "D_A": faoctasr.FS_DiscriminatorA(1), "D_B": faoctasr.FS_DiscriminatorB(1)}
    specs = {"A2B": O.spec_network_a2b(), "B2A": O.spec_network_b2a(), "D_A": O.spec_fs_discriminator("sum"), "D_B": O.spec_fs_discriminator("cat")}
    for k, n in nets.items():
        n.load_state_dict(O.make_state(specs[k], k, seed), strict=True); n.cuda().train()
    return nets
for (H, B) in ((192, 1), (192, 2), (256, 1), (256, 4)):
    res = {}
    for prec in ("f32", "bf16x3"):
        random.seed(1234)
        n = build()
        ts = faoctasr.TrainStep(n["A2B"], n["B2A"], n["D_A"], n["D_B"], precision=prec)
        a, b = O.synthetic_batch(B, H, seed=1234)
        L = ts.step(a.cuda(), b.cuda(), sync=True)
        res[prec] = (L, ts.grad_norms())
    L0, g0 = res["f32"]; L1, g1 = res["bf16x3"]
    print("H=%d B=%d  loss rel: %s" % (H, B, {k: "%.1e" % (abs(L1[k]-L0[k])/abs(L0[k])) for k in L0}))
    print("          gradnorm rel: %s" % {k: "%.1e" % (abs(g1[k]-g0[k])/g0[k]) for k in g0})
