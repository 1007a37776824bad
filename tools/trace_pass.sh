#!/bin/bash
# One rocprofv3 kernel-trace pass of bench.py WITH the multi-stream schedule, summarised by tools/trace_overlap.py (GPU busy time,
# kernels in flight, largest idle gaps).  usage, on the GPU box from the repo root:  bash tools/trace_pass.sh <tag> <bench.py flags...>
set -e
R=$(pwd)
TAG=$1; shift
O=$R/gpurun_out/prof_$TAG
mkdir -p "$O"
cd /tmp
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$O/trace" -o trace -- python3 "$R/bench.py" --steps 4 --warmup 2 --no-cpu-baseline --no-alt --no-graph --no-roofline "$@" > "$O/trace.log" 2>&1
cd "$R"
T=$(find "$O/trace" -name "*kernel_trace.csv" | head -1)
python3 tools/trace_overlap.py "$T" > "$R/gpurun_out/trace_overlap_$TAG.txt" 2>&1
rm -rf "$O/trace"
head -40 "$R/gpurun_out/trace_overlap_$TAG.txt"
