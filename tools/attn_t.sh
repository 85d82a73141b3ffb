#!/bin/bash
export TMPDIR=/tmp
for shp in "128 16 256 72" "32 16 1024 72" "128 16 256 64" "32 16 1024 64"; do
  rm -rf gpurun_out/ap_t
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ap_t -o p -- python3 tools/attn_one.py $shp > gpurun_out/ap_t.log 2>&1
  echo "== $shp"; python3 tools/kstats.py gpurun_out/ap_t/p_kernel_stats.csv 5 | grep -i "attn_bwd" | sed 's/  */ /g' | awk '{print substr($1,1,34), $(NF-5), $(NF-4), $(NF-3), $(NF-2), $(NF-1), $NF}'
done
