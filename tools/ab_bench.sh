#!/bin/bash
# A/B on ONE box (boxes differ by +-5 %): the round-2 tree (git worktree _base, built in place) against this tree, interleaved.
#   bash tools/ab_bench.sh "<bench flags>" [rounds]
FLAGS=${1:-"--steps 30 --warmup 5"}; R=${2:-2}
for i in $(seq $R); do
  for side in _base .; do
    (cd $side && python3 bench.py $FLAGS --no-cpu-baseline --no-trace 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$side', d['config']['per_gpu_batch'], d['ms_per_step'], d['median_ms_per_step'])")
  done
done
