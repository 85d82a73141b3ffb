#!/bin/bash
# On the GPU box: interleaved A/B of one environment switch on one bench workload.  bash tools/ab_env.sh "<bench flags>" <rounds> VAR=a VAR=b ...
FLAGS=$1; ROUNDS=$2; shift; shift
for r in $(seq 1 $ROUNDS); do
  for kv in "$@"; do
    env $kv python3 bench.py $FLAGS --no-cpu-baseline --no-trace 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$kv', d['ms_per_step'], d['median_ms_per_step'])"
  done
done
