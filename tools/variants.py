"""Build kernel variants of libfaoctasr.so for ablation experiments: one source recompiled with extra -D flags, the other
objects reused.  usage: python tools/variants.py igemm_wino.hip WINO_ABLATE 0 1 2 ...  ->  tools/variants/libfaoctasr_<macro>_<v>.so
Run a variant with FAOCTASR_LIB=tools/variants/<file> (see _lib.py)."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "frequency-aware-inverse-consistent-octa-super-resolution_amd")
sys.path.insert(0, PKG)
import build as B  # noqa: E402

OUT = os.path.join(ROOT, "tools", "variants")
FLAGS = ["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(PKG, "csrc")]


def obj(src, out, extra=()):
    subprocess.run(FLAGS + B.FILE_FLAGS.get(src, []) + list(extra) + ["-c", os.path.join(PKG, "csrc", src), "-o", out], check=True)
    return out


def main():
    src, macro, values = sys.argv[1], sys.argv[2], sys.argv[3:]
    os.makedirs(OUT, exist_ok=True)
    if "," in src:                       # the same macro in several sources: one variant library per value
        srcs = src.split(",")
        others = [s for s in B.SOURCES if s not in srcs]
        with ThreadPoolExecutor(6) as ex:
            base = list(ex.map(lambda s: obj(s, os.path.join(OUT, s + ".o")), others))
            for v in values:
                var = list(ex.map(lambda s: obj(s, os.path.join(OUT, "%s_%s_%s.o" % (s, macro, v)), ["-D%s=%s" % (macro, v)]), srcs))
                lib = os.path.join(OUT, "libfaoctasr_%s_%s.so" % (macro, v))
                subprocess.run(FLAGS + ["-shared"] + base + var + ["-ldl", "-o", lib], check=True)
                print(lib)
        return
    others = [s for s in B.SOURCES if s != src]
    with ThreadPoolExecutor(6) as ex:
        base = list(ex.map(lambda s: obj(s, os.path.join(OUT, s + ".o")), others))
        # a value may carry further macros: "1+WINO_ABLATE=16" -> -DMACRO=1 -DWINO_ABLATE=16
        flags = lambda v: ["-D%s=%s" % (macro, v.split("+")[0])] + ["-D" + e for e in v.split("+")[1:]]
        var = list(ex.map(lambda v: obj(src, os.path.join(OUT, "%s_%s_%s.o" % (src, macro, v)), flags(v)), values))
    for v, o in zip(values, var):
        lib = os.path.join(OUT, "libfaoctasr_%s_%s.so" % (macro, v))
        subprocess.run(FLAGS + ["-shared"] + base + [o, "-o", lib], check=True)
        print(lib)


if __name__ == "__main__":
    main()
