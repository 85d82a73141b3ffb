#!/bin/bash
# On the GPU box: rocprofv3 kernel-trace summary of one bench workload.  bash tools/trace_wl.sh <tag> <workload> [bench flags]
set -o pipefail
TAG=$1; WL=$2; shift; shift
OUT=gpurun_out/profiles; mkdir -p $OUT
export TMPDIR=/tmp
d=gpurun_out/prof_$WL; rm -rf $d
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o p -- python3 bench.py --workload $WL --no-cpu-baseline --no-graph "$@" > $d.log 2>&1 || { tail -5 $d.log; exit 1; }
cp $d/p_kernel_stats.csv $OUT/${TAG}_${WL}_kernel_stats.csv
grep '^{"metric"' $d.log | tail -1 > $OUT/${TAG}_${WL}_bench_under_rocprof.json
rm -rf $d
python3 -c "import json;d=json.load(open('$OUT/${TAG}_${WL}_bench_under_rocprof.json'));print('$WL', d['ms_per_step'],'ms/step')"
