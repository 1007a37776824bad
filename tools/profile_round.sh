#!/bin/bash
# The rocprofv3 passes behind profiles/<tag>_*: kernel statistics, MFMA / LDS counters, FETCH_SIZE and WRITE_SIZE (one counter set
# per pass, never together with a --sys/--hip trace), each of bench.py at the headline workload with the weight gradients on the
# main stream (--no-overlap: a kernel's duration is then its own, as in bench.py's roofline leg).
# usage, on the GPU box from the repo root:  bash tools/profile_round.sh r03      (raw output: gpurun_out/prof_<tag>/)
set -e
R=$(pwd)
TAG=${1:-r04}
O=$R/gpurun_out/prof_$TAG
mkdir -p "$O"
cd /tmp
export TMPDIR=/tmp
STATS="--steps 3 --warmup 1 --no-cpu-baseline --no-alt --no-graph --no-overlap"
PMC="--steps 2 --warmup 1 --no-cpu-baseline --no-alt --no-graph --no-roofline --no-overlap"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -o stats -- python3 "$R/bench.py" $STATS > "$O/stats.log" 2>&1
echo "stats pass done"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY \
    --output-format csv -d "$O/mfma" -o mfma -- python3 "$R/bench.py" $PMC > "$O/mfma.log" 2>&1
echo "mfma pass done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$O/fetch" -o fetch -- python3 "$R/bench.py" $PMC > "$O/fetch.log" 2>&1
echo "fetch pass done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$O/write" -o write -- python3 "$R/bench.py" $PMC > "$O/write.log" 2>&1
echo "write pass done"
cd "$R"
S=$(find "$O/stats" -name "*kernel_stats.csv" | head -1)
{ echo "# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py $STATS  (1 warm-up + 3 timed + 3 roofline steps = 7 train steps at the default precision f16x2, then the Haar/SSIM roofline_hbm kernels on 512 planes; MI355X, 256x256 batch 8)"; cat "$S"; } > "$O/${TAG}_kernel_stats_bench_b8_256.csv"
grep '^{' "$O/stats.log" | tail -1 | python3 -m json.tool > "$O/${TAG}_bench_line_under_rocprof.json"
python3 tools/profile_summary.py mfma "$(find "$O/mfma" -name "*counter_collection.csv" | head -1)" "$O/${TAG}_mfma_utilisation.json" "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -- python3 bench.py $PMC (MI355X, 256x256 batch 8, precision f16x2)"
python3 tools/profile_summary.py traffic "$(find "$O/fetch" -name "*counter_collection.csv" | head -1)" "$(find "$O/write" -name "*counter_collection.csv" | head -1)" "$O/${TAG}_traffic.json" "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py $PMC (MI355X, 256x256 batch 8, precision f16x2)"
# the same kernel statistics for the exact-f32 MFMA step (exact_f32_mfma: the headline arithmetic of rounds 1-3)
X3="--steps 3 --warmup 1 --no-cpu-baseline --no-alt --no-graph --no-overlap --no-roofline --precision f32"
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats_x3" -o stats -- python3 "$R/bench.py" $X3 > "$O/stats_x3.log" 2>&1 )
SX=$(find "$O/stats_x3" -name "*kernel_stats.csv" | head -1)
{ echo "# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py $X3  (4 train steps on one stream; MI355X, 256x256 batch 8, exact-f32 MFMA convolutions)"; cat "$SX"; } > "$O/${TAG}_kernel_stats_bench_b8_256_exact_f32.csv"
echo "exact-f32 stats pass done"
# keep what travels back small: the raw counter CSVs are tens of MiB
find "$O" -name "*counter_collection.csv" -delete
find "$O" -name "*kernel_trace.csv" -delete
ls -la "$O"
