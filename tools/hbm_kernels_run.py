"""The Haar DWT / SSIM kernels at bench size (512 planes of 256x256), forward and backward: the workload of bench.py's
`roofline_hbm` lines, as a stand-alone command for rocprofv3 (--kernel-trace --stats, --pmc FETCH_SIZE / WRITE_SIZE)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

for r in bench.hbm_kernels(torch.device("cuda", 0)):
    print("%-60s %8.1f us %8.1f GB/s  frac %.3f" % (r["kernel"], r["us"], r["achieved"], r["frac"]))
