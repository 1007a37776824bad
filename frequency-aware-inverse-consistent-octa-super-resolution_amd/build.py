"""Build libfaoctasr.so (hipcc, gfx950 only) in-tree, next to this file."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SOURCES = ["igemm.hip", "igemm_patch.hip", "igemm_bf16x3.hip", "igemm_wino.hip", "wgrad_patch.hip", "conv_m1.hip", "pointwise.hip", "sgemm.hip", "ssim.hip"]
LIB = os.path.join(HERE, "libfaoctasr.so")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(HERE, "csrc", s) for s in SOURCES] + [os.path.join(HERE, "csrc", "common.h"), os.path.join(HERE, "csrc", "igemm_geom.h"), os.path.join(ROOT, "include", "faoctasr.h")]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    srcs = [os.path.join(HERE, "csrc", s) for s in SOURCES if os.path.exists(os.path.join(HERE, "csrc", s))]
    extra = os.environ.get("FAOCTASR_HIPCC_FLAGS", "").split()
    cmd = [hipcc, "-O3", "--offload-arch=gfx950", "-fPIC", "-shared", "-std=c++17"] + extra + ["-I" + os.path.join(ROOT, "include"),
           "-I" + os.path.join(HERE, "csrc")] + srcs + ["-o", LIB + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
