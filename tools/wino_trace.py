import sys, os, torch
sys.path.insert(0, "/root/repo")
import faoctasr
from faoctasr import ops
faoctasr._lib.load()
x = torch.randn(8, 64, 256, 256, device="cuda"); w = torch.randn(64, 64, 3, 3, device="cuda") * 0.05
bias = torch.zeros(4096, device="cuda")          # rows 0..63 = the (zero) bias; the WINO_TRACE build stamps s_memtime behind it
with torch.no_grad():
    for _ in range(2):
        y = ops.conv2d(x, w, bias, 1, 1, False, None, 0.2)
torch.cuda.synchronize()
t = bias.view(torch.int32).cpu().numpy().astype("int64") & 0xffffffff
p = t[1024:1152].reshape(32, 4); c = t[2048:2176].reshape(32, 4)
print("producer: store_v(wait+xform)  issue(DMA+loads)  barrier-wait   total")
for r in p[:24]:
    print("   %6d %6d %6d   %6d" % ((r[1]-r[0]) & 0xffffffff, (r[2]-r[1]) & 0xffffffff, (r[3]-r[2]) & 0xffffffff, (r[3]-r[0]) & 0xffffffff))
print("consumer: mfma-loop  barrier-wait")
for r in c[:24]:
    print("   %6d %6d" % ((r[1]-r[0]) & 0xffffffff, (r[2]-r[1]) & 0xffffffff))
