#!/bin/bash
# rocprofv3 kernel trace of the CAPTURED step's replays (bench.py --graph-only), summarised by tools/trace_overlap.py beside the eager
# trace of tools/trace_pass.sh: kernels in flight (sum of kernel time / GPU-busy time) and queues used by the graph executor.
# usage, on the GPU box from the repo root:  bash tools/trace_graph_pass.sh <tag>
set -e
R=$(pwd)
TAG=$1; shift
O=$R/gpurun_out/prof_$TAG
mkdir -p "$O"
cd /tmp
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$O/trace" -o trace -- python3 "$R/bench.py" --steps 4 --warmup 2 --no-cpu-baseline --no-alt --no-roofline --graph-only "$@" > "$O/trace.log" 2>&1
cd "$R"
T=$(find "$O/trace" -name "*kernel_trace.csv" | head -1)
python3 tools/trace_overlap.py "$T" > "$R/gpurun_out/trace_overlap_$TAG.txt" 2>&1
rm -rf "$O/trace"
tail -5 "$O/trace.log" | cut -c1-300
head -30 "$R/gpurun_out/trace_overlap_$TAG.txt"
