"""s_memtime phase trace of igemm_bf16x3_kernel, block 0 (an SP_TRACE=1 build: tools/variants.py igemm_bf16x3.hip SP_TRACE 1;
run with FAOCTASR_LIB=tools/variants/libfaoctasr_SP_TRACE_1.so).  usage: python tools/sp_trace.py [C M H]"""
import sys, ctypes
import numpy as np
import torch
sys.path.insert(0, "/root/repo")
import faoctasr
from faoctasr import ops
lib = faoctasr._lib.load()
ops.conv_precision = ops.PRECISIONS[__import__("os").environ.get("FAOCTASR_PRECISION", "f16x2")]
C, M, H = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (64, 64, 256)))
x = torch.randn(8, C, H, H, device="cuda"); w = torch.randn(M, C, 3, 3, device="cuda") * 0.05
with torch.no_grad():
    for _ in range(2):
        y = ops.conv2d(x, w, None, 1, 1, False, None, 0.2)
torch.cuda.synchronize()
raw = (ctypes.c_uint * 4096)()
assert lib.faoctasr_sp_trace_read(raw, 4096) == 0, "not an SP_TRACE=1 build"
t = np.frombuffer(raw, dtype=np.uint32).astype("int64")
p = t[1024:1152].reshape(32, 4); c = t[2048:2176].reshape(32, 4); pl = t[3072:3200].reshape(32, 4)
d = lambda a, b: int((a - b) & 0xffffffff)
Q0 = int(__import__("os").environ.get("SP_TRACE_Q0", "16"))
print("slab | producer wave 4: split+stores+DMA  rest of DMA  patch load issue  barrier | consumer wave 0: fragment+MFMA loop  drain  barrier  (to next slab start)")
for i in range(24):
    nxt = d(c[i + 1][0], c[i][3])
    print("%4d | %6d %6d %6d %6d | %6d %6d %6d %6d" % (Q0 + i, d(p[i][0], pl[i][0]), d(p[i][1], p[i][0]), d(p[i][2], p[i][1]), d(p[i][3], p[i][2]),
          d(c[i][1], c[i][0]), d(c[i][2], c[i][1]), d(c[i][3], c[i][2]), nxt))
Q0 = int(__import__("os").environ.get("SP_TRACE_Q0", "16"))             # the build's -DSP_TRACE_Q0 (labels only)
print("block 0: kernel entry -> first patch stored %d, -> first barrier passed %d, -> consumer's last epilogue issued %d cycles" % (d(t[9], t[8]), d(t[10], t[8]), d(t[11], t[8])))
print("   producer wave 4: set-up done %d, first patch's loads issued %d, first weight slab's DMAs issued %d cycles after kernel entry" % (d(t[12], t[8]), d(t[13], t[8]), d(t[14], t[8])))
print("first recorded consumer slab starts %d cycles after kernel entry" % d(c[0][0], t[8]))
e = t[3584:3712].reshape(32, 4)
print("epilogue of consumer wave 0 (stamped at the slab index that follows it): scale + bias + activation | transpose + stores issued | re-zero")
for i in range(0, 32):
    if e[i][0]:
        print("%4d | %6d %6d %6d" % (16 + i, d(e[i][1], e[i][0]), d(e[i][2], e[i][1]), d(e[i][3], e[i][2])))
