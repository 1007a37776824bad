"""Phase trace of one wgrad_patch block (WG_TRACE=1 variant from tools/variants.py): per pixel tile, cycles spent issuing the next
tile's loads, in the MFMA loop, and in LDS stores + barrier."""
import sys
import torch
sys.path.insert(0, "/root/repo")
import faoctasr
from faoctasr import _lib
from faoctasr._lib import call, ptr, stream_ptr
_lib.load()
N, C, H, W, M = [int(v) for v in sys.argv[1:6]] if len(sys.argv) > 5 else (8, 64, 256, 256, 64)
x = torch.randn(N, C, H, W, device="cuda"); dy = torch.randn(N, M, H, W, device="cuda")
dw = torch.zeros(M, C, 3, 3, device="cuda")
for _ in range(2):
    dw.zero_()
    call("conv2d_wgrad", ptr(x), ptr(dy), ptr(dw), N, C, H, W, M, 3, 3, 1, 1, 0, 1, 0, stream_ptr())
torch.cuda.synchronize()
import ctypes, numpy as np
raw = (ctypes.c_uint * 128)()
assert _lib.load().faoctasr_wgrad_trace_read(raw, 128) == 0, "not a WG_TRACE=1 build (tools/variants.py)"
t = np.frombuffer(raw, dtype=np.uint32).astype("int64")
t = t.reshape(32, 4)
d = lambda a, b: int((a - b) & 0xffffffff)
print("tile | load-issue  mfma-loop  store+sync | total")
for r in t[:20]:
    print("     | %8d %9d %10d | %6d" % (d(r[1], r[0]), d(r[2], r[1]), d(r[3], r[2]), d(r[3], r[0])))
