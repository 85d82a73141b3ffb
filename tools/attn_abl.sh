#!/bin/bash
# On the GPU box: where the time of the attention backward goes -- ablation builds (make exp XN=ablN XF=-DBIG_ABL=N XSRC=attention_bwd_big;
# 1 no slice barrier, 2 no exp, 3 no LDS waits: results are wrong, only the timing means something) against the product library.
for v in base abl4 abl5 abl6 base; do
  if [ $v = base ]; then unset VAW_HIP_LIB; else export VAW_HIP_LIB=$PWD/variance-aware-weight_amd/libvaw_hip_$v.so; fi
  echo "== $v"; timeout -k 10 120 python3 tools/attn_bench.py 2>/dev/null | grep "bwd" | grep -v "DiT-B/4"
done
