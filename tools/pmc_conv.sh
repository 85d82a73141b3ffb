#!/bin/bash
# PMC passes over one conv shape (run on the GPU box): bash tools/pmc_conv.sh "<B H Ci Co>" "<modes>" "<counter set>" ...
export TMPDIR=/tmp
SHAPE=${1:-"128 32 384 384"}; MODES=${2:-"0 2"}; shift 2
for mode in $MODES; do
  i=0
  for set in "$@"; do
    i=$((i+1))
    d=gpurun_out/pc_${mode}_$i; rm -rf $d
    timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $d -o c -- python3 tools/one_conv.py $SHAPE $mode 3 > $d.log 2>&1 || tail -3 $d.log
    python3 tools/pmc_table.py $d/c_counter_collection.csv gemm_p8
  done
done
