import sys, os, torch
sys.path.insert(0, "/root/repo")
import faoctasr
from faoctasr import ops
faoctasr._lib.load()
x = torch.randn(8, 64, 256, 256, device="cuda"); w = torch.randn(64, 64, 3, 3, device="cuda") * 0.05
bias = None                                      # the WINO_TRACE build stamps s_memtime into its own device buffer
with torch.no_grad():
    for _ in range(2):
        y = ops.conv2d(x, w, bias, 1, 1, False, None, 0.2)
torch.cuda.synchronize()
import ctypes, numpy as np
raw = (ctypes.c_uint * 4096)()
assert faoctasr._lib.load().faoctasr_wino_trace_read(raw, 4096) == 0, "not a WINO_TRACE=1 build (tools/variants.py)"
t = np.frombuffer(raw, dtype=np.uint32).astype("int64")
p = t[1024:1152].reshape(32, 4); c = t[2048:2176].reshape(32, 4); u = t[3072:3200].reshape(32, 4)
d = lambda a, b: int((a - b) & 0xffffffff)
print("slab | transform wave: store_v  loads  barrier | weights wave: store  loads  barrier | consumer: mfma-loop  barrier  (to next slab start)")
for i in range(16):
    nxt = d(c[i + 1][0], c[i][2]) if i + 1 < 32 else 0
    print("%4d | %6d %6d %6d | %6d %6d %6d | %6d %6d %6d" % (16 + i, d(p[i][1], p[i][0]), d(p[i][2], p[i][1]), d(p[i][3], p[i][2]),
          d(u[i][1], u[i][0]), d(u[i][2], u[i][1]), d(u[i][3], u[i][2]), d(c[i][1], c[i][0]), d(c[i][2], c[i][1]), nxt))
