"""Build libfaoctasr.so (hipcc, gfx950 only) in-tree, next to this file.

One object per source under ``build/obj`` (git-ignored), compiled in parallel and reused while its source and the shared
headers are older than it; the link step runs whenever any object is newer than the library.  ``build()`` reports what it
compiled, so a driver log shows whether the check really exercised the compiler (``force=True`` recompiles everything)."""
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SOURCES = ["igemm.hip", "igemm_patch.hip", "igemm_bf16x3.hip", "igemm_wino.hip", "igemm_nm.hip", "wgrad_patch.hip", "wgrad_s1.hip", "wgrad_x3.hip", "conv_m1.hip", "pointwise.hip", "norm.hip", "sgemm.hip",
           "ssim.hip", "eval.hip", "comm.hip", "conv_pack.hip", "conv_stem.hip"]
HEADERS = [os.path.join(HERE, "csrc", "common.h"), os.path.join(HERE, "csrc", "igemm_geom.h"), os.path.join(HERE, "csrc", "pack_bodies.h"), os.path.join(ROOT, "include", "faoctasr.h")]
#: per-source extra flags.  ssim.hip: the SLP vectoriser re-packs the scalar 11-tap filters into v_pk_fma_f32, which has the FLOP
#: rate of two v_fma_f32 here and costs ~85 v_mov per filtered row to keep operands in aligned register pairs
FILE_FLAGS = {"ssim.hip": ["-fno-slp-vectorize"]}
LIB = os.path.join(HERE, "libfaoctasr.so")
OBJ_DIR = os.path.join(ROOT, "build", "obj")


def _flags():
    extra = os.environ.get("FAOCTASR_HIPCC_FLAGS", "").split()
    return ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17"] + extra + ["-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(HERE, "csrc")]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def needs_build():
    srcs = [os.path.join(HERE, "csrc", s) for s in SOURCES]
    return _stale(LIB, srcs + HEADERS)


def build(force=False, verbose=False):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ_DIR, exist_ok=True)
    tag = "_".join(os.environ.get("FAOCTASR_HIPCC_FLAGS", "").split()).replace("/", "_").replace("=", "-")
    jobs, objs = [], []
    for s in SOURCES:
        src = os.path.join(HERE, "csrc", s)
        obj = os.path.join(OBJ_DIR, s + (("." + tag) if tag else "") + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + HEADERS):
            jobs.append((s, [hipcc] + _flags() + FILE_FLAGS.get(s, []) + ["-c", src, "-o", obj]))

    def run(job):
        name, cmd = job
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s\n%s" % (name, " ".join(cmd), r.stdout))
        return name

    if jobs:
        with ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            done = list(ex.map(run, jobs))
        if verbose:
            print("compiled for gfx950:", " ".join(done))
    if jobs or force or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-fPIC", "-shared"] + objs + ["-ldl", "-o", LIB + ".tmp"]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (" ".join(cmd), r.stdout))
        os.replace(LIB + ".tmp", LIB)
        if verbose:
            print("linked", LIB)
    elif verbose:
        print("up to date:", LIB)
    return LIB


if __name__ == "__main__":
    import sys
    print(build(force="--force" in sys.argv, verbose=True))
