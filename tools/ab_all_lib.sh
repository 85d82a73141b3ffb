#!/bin/bash
# The round's A/B table in ONE call (one box): a saved library of the previous round (VAW_HIP_LIB) against this tree's, every
# workload, interleaved.    bash tools/ab_all_lib.sh variance-aware-weight_amd/libvaw_hip_r3.so
BASE=${1:-variance-aware-weight_amd/libvaw_hip_r3.so}
ab() { echo "== $1"; bash tools/ab_lib.sh "$1" ${2:-2} base:$BASE new:- || exit 1; }
ab "--steps 40 --warmup 8" 3
ab "--batch 128 --steps 40 --warmup 8"
ab "--batch 64 --steps 40 --warmup 8"
ab "--batch 32 --steps 40 --warmup 8"
ab "--workload dit_b2 --steps 10 --warmup 3"
ab "--workload dit_xl2_fp8 --steps 5 --warmup 2"
ab "--workload dit_xl2 --steps 5 --warmup 2"
ab "--workload unet64 --steps 6 --warmup 2"
ab "--workload adm64 --steps 3 --warmup 1"
ab "--workload unet32 --steps 40 --warmup 8"
