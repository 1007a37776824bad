#!/bin/bash
# bench.py at the other per-GPU batch sizes (same binaries): one line per batch =
#   eager ms/step, eager img/s (default precision f16x2), captured-graph ms/step, host enqueue ms/step, exact-f32 img/s, bf16x3 img/s
#   (profiles/r0N_bench_batches.log); SIZE=512 in the environment benches 512x512 images
# usage, on the GPU box from the repo root:  bash tools/bench_batches.sh [batches...]
for b in ${@:-1 2 4 16 32 64}; do
    echo "== batch $b"
    steps=10; [ "$b" -ge 16 ] && steps=5; [ "$b" -ge 64 ] && steps=3
    extra=""; [ "$b" -ge 16 ] && extra="--no-graph"
    timeout -k 10 600 python bench.py --size ${SIZE:-256} --batch $b --steps $steps --warmup 2 --no-cpu-baseline --no-roofline $extra 2>/dev/null | grep '^{' | tail -1 | python3 -c "
import json, sys
d = json.loads(sys.stdin.read())
g = d.get('hipgraph_step') or {}
a = d.get('alt_precision_bf16x3') or {}
f = d.get('exact_f32_mfma') or {}
print(d['ms_per_step'], d['value'], g.get('ms_per_step'), d.get('host_enqueue_ms_per_step'), f.get('value'), a.get('value'))" || exit 1
done
