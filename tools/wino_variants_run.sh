#!/bin/bash
# One GPU pass over Winograd kernel variants built by tools/variants.py (arguments: library files under tools/variants/).
# For each: the s_memtime phase trace of block 0 when it is a WINO_TRACE build (tools/wino_trace.py) and the layer timings
# (tools/conv_bench.py c2,c5,c8 forward).
for lib in "$@"; do
    echo "== $lib"
    case "$lib" in *WINO_TRACE*) FAOCTASR_LIB=tools/variants/$lib python tools/wino_trace.py || exit 1;; esac
    FAOCTASR_LIB=tools/variants/$lib python tools/conv_bench.py 8 256 c2,c5,c8 fwd 2>/dev/null | grep -v "^layer" || exit 1
done
