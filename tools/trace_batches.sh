#!/bin/bash
# On the GPU box: rocprofv3 kernel-trace summaries of the DiT-B/4 step at several per-GPU batches (strong-scaling sizes).
#   bash tools/trace_batches.sh <tag> <batch>...   -> gpurun_out/profiles/<tag>_dit_b4_bs<batch>_bf16_kernel_stats.csv (+ bench line)
set -o pipefail
TAG=$1; shift
OUT=gpurun_out/profiles; mkdir -p $OUT
export TMPDIR=/tmp
for b in "$@"; do
  d=gpurun_out/prof_b$b; rm -rf $d
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o p -- python3 bench.py --batch $b --steps 20 --warmup 5 --no-cpu-baseline --no-graph $EXTRA > $d.log 2>&1 || { tail -5 $d.log; exit 1; }
  cp $d/p_kernel_stats.csv $OUT/${TAG}_dit_b4_bs${b}_bf16_kernel_stats.csv
  grep '^{"metric"' $d.log | tail -1 > $OUT/${TAG}_dit_b4_bs${b}_bf16_bench_under_rocprof.json
  rm -rf $d
  echo "== batch $b: $(python3 -c "import json;d=json.load(open('$OUT/${TAG}_dit_b4_bs${b}_bf16_bench_under_rocprof.json'));print(d['ms_per_step'],'ms/step')")"
done
