"""BatchNorm2d (training) forward / backward per layer shape through the C ABI (no autograd: ~6 us of host time per call):
microseconds per call and bytes per second against the tensor passes the kernels make (forward: statistics read + apply read + write = 3;
backward: reduce reads dy, x; dx reads dy, x, writes dx = 5).     usage: python tools/norm_bench.py [batch]"""
import sys
import torch
sys.path.insert(0, "/root/repo")
import faoctasr
from faoctasr import _lib
from faoctasr._lib import call, ptr, stream_ptr

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
SHAPES = [(64, 256), (128, 128), (256, 64), (128, 64), (256, 32), (512, 31), (512, 16)]
_lib.load()


def timed(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


print("shape (B=%d)            MB/tensor |  fwd us   TB/s (3 passes) |  bwd us   TB/s (5 passes)" % B)
for C, H in SHAPES:
    x = torch.randn(B, C, H, H, device="cuda")
    dy = torch.randn(B, C, H, H, device="cuda")
    y, dx = torch.empty_like(x), torch.empty_like(x)
    gamma, beta = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    dg, db = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    stats = torch.empty(2, C, device="cuda")
    ws = _lib.workspace(x.device, C * 128)
    sp = stats.data_ptr()
    mb = x.numel() * 4 / 1e6
    fwd = lambda: call("batchnorm_train_fwd", ptr(x), ptr(gamma), ptr(beta), None, ptr(y), sp, sp + 4 * C, ptr(rm), ptr(rv), B, C, H * H, 1e-5, 0.1,
                       1, 0.2, ptr(ws), stream_ptr())
    bwd = lambda: call("batchnorm_train_bwd", ptr(x), ptr(dy), None, ptr(gamma), ptr(beta), sp, sp + 4 * C, ptr(dx), ptr(dg), ptr(db), None,
                       B, C, H * H, 1, 0.2, 1, ptr(ws), stream_ptr())
    tf, tb = timed(fwd), timed(bwd)
    print("%4d x %3d^2            %8.1f | %7.1f   %5.2f            | %7.1f   %5.2f" % (C, H, mb, tf, 3 * mb / tf, tb, 5 * mb / tb))
