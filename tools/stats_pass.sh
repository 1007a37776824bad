#!/bin/bash
# One rocprofv3 kernel-statistics pass of bench.py on one stream (a kernel's duration is then its own).
# usage, on the GPU box from the repo root:  bash tools/stats_pass.sh <tag> <bench.py flags...>     -> gpurun_out/stats_<tag>.csv
set -e
R=$(pwd)
TAG=$1; shift
O=$R/gpurun_out/prof_$TAG
mkdir -p "$O"
cd /tmp
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -o stats -- python3 "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-alt --no-graph --no-overlap --no-roofline "$@" > "$O/stats.log" 2>&1
cd "$R"
S=$(find "$O/stats" -name "*kernel_stats.csv" | head -1)
{ echo "# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt --no-graph --no-overlap --no-roofline $*  (4 train steps on one stream + 5 host-enqueue steps = 9 steps; MI355X, 256x256 batch 8)"; cat "$S"; } > "$R/gpurun_out/stats_$TAG.csv"
rm -rf "$O/stats"
grep '^{' "$O/stats.log" | tail -1 | cut -c1-400
