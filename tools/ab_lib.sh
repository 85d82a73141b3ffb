#!/bin/bash
# On the GPU box: interleaved A/B of measurement builds of the library on one bench workload.
#   bash tools/ab_lib.sh "<bench flags>" <rounds> <tag:libpath> [<tag:libpath> ...]      (libpath "-" = the product library)
FLAGS=$1; ROUNDS=$2; shift; shift
for r in $(seq 1 $ROUNDS); do
  for v in "$@"; do
    tag=${v%%:*}; lib=${v#*:}
    if [ "$lib" = "-" ]; then unset VAW_HIP_LIB; else export VAW_HIP_LIB=$PWD/$lib; fi
    out=gpurun_out/ab_${tag}_r$r
    timeout -k 10 400 python3 bench.py $FLAGS --no-cpu-baseline --shape-table $out.shapes > $out.log 2>&1 || { tail -3 $out.log; exit 1; }
    python3 -c "import json;d=json.loads([l for l in open('$out.log') if l.startswith('{\"metric')][-1]);print('$tag r$r', d['ms_per_step'], d['median_ms_per_step'])"
  done
done
