#!/bin/bash
# per-kernel GroupNorm timings: base tree vs this tree (V=8 and V=4)
export TMPDIR=/tmp
run() { # name dir env
  rm -rf gpurun_out/gnp_$1
  ( cd $2 && env $3 true; export $3; timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/gnp_$1 -o p -- python3 tools/gn_bench.py ) > gpurun_out/gnp_$1.log 2>&1
  python3 tools/kstats.py gpurun_out/gnp_$1/p_kernel_stats.csv 1 | grep -i "gn_" > gpurun_out/gnp_$1.txt
}
run base _base X=1
run v8 . X=1
run v4 . VAW_GN_V4=1
