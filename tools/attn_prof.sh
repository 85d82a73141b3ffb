#!/bin/bash
# On the GPU box: kernel times + a few PMC counters of one attention shape.  bash tools/attn_prof.sh B H T hd
export TMPDIR=/tmp
rm -rf gpurun_out/ap_*
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ap_t -o p -- python3 tools/attn_one.py "$@" > gpurun_out/ap_t.log 2>&1
python3 tools/kstats.py gpurun_out/ap_t/p_kernel_stats.csv 5 | grep -i attn
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS --kernel-trace --output-format csv -d gpurun_out/ap_c -o c -- python3 tools/attn_one.py "$@" > gpurun_out/ap_c.log 2>&1
python3 tools/pmc_fold.py gpurun_out/ap_c/c_counter_collection.csv attn_bwd
timeout -k 10 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC --kernel-trace --output-format csv -d gpurun_out/ap_d -o c -- python3 tools/attn_one.py "$@" > gpurun_out/ap_d.log 2>&1
python3 tools/pmc_fold.py gpurun_out/ap_d/c_counter_collection.csv attn_bwd
