#!/bin/bash
# Round profiles, run ON the GPU box from the repo root:  bash tools/collect_profiles.sh r02 [workloads...]
# Writes gpurun_out/profiles/<tag>_*: rocprofv3 kernel stats + the bench line of the same run for every workload, an unprofiled
# bench line, and (dit_b4, dit_xl2_fp8) the PMC passes folded by tools/pmc_traffic.py / tools/pmc_mfma.py.  Copy what should be
# judged into profiles/.
set -o pipefail
TAG=${1:-r02}; shift
WLS=${@:-"dit_b4 unet64 adm64 dit_xl2 dit_xl2_fp8"}
OUT=gpurun_out/profiles; mkdir -p $OUT
export TMPDIR=/tmp
declare -A FLAGS=( [dit_b4]="--steps 20 --warmup 5" [unet64]="--steps 5 --warmup 2" [adm64]="--steps 3 --warmup 1" [dit_xl2]="--steps 4 --warmup 2" [dit_xl2_fp8]="--steps 4 --warmup 2" )
declare -A BS=( [dit_b4]=256 [unet64]=128 [adm64]=256 [dit_xl2]=128 [dit_xl2_fp8]=128 )
for wl in $WLS; do
  dt=bf16; [ $wl = dit_xl2_fp8 ] && dt=fp8
  base=${TAG}_${wl}_bs${BS[$wl]}_${dt}
  d=gpurun_out/prof_$wl; rm -rf $d
  echo "== $wl: kernel stats"
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o p -- python3 bench.py --workload $wl ${FLAGS[$wl]} --no-cpu-baseline --no-graph > $d.log 2>&1 || { tail -5 $d.log; exit 1; }
  cp $d/p_kernel_stats.csv $OUT/${base}_kernel_stats.csv
  grep '^{"metric"' $d.log | tail -1 > $OUT/${base}_bench_under_rocprof.json
  if [ $wl = dit_b4 ] || [ $wl = dit_xl2_fp8 ] || [ $wl = unet64 ] || [ "$PMC_ALL" = 1 ]; then
    echo "== $wl: PMC passes"
    for c in FETCH_SIZE WRITE_SIZE; do
      rm -rf gpurun_out/pmc_$c
      timeout -k 10 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_$c -o c -- python3 bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline --no-trace --no-graph > gpurun_out/pmc_$c.log 2>&1 || { tail -5 gpurun_out/pmc_$c.log; exit 1; }
    done
    python3 tools/pmc_traffic.py gpurun_out/pmc_FETCH_SIZE/c_counter_collection.csv gpurun_out/pmc_WRITE_SIZE/c_counter_collection.csv $OUT/${base}_hbm_traffic.json gemm_ || exit 1
    rm -rf gpurun_out/pmc_mfma
    timeout -k 10 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_mfma -o c -- python3 bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline --no-trace --no-graph > gpurun_out/pmc_mfma.log 2>&1 || { tail -5 gpurun_out/pmc_mfma.log; exit 1; }
    python3 tools/pmc_mfma.py gpurun_out/pmc_mfma/c_counter_collection.csv $OUT/${base}_mfma_pmc.json || exit 1
    rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/pmc_mfma
  fi
  rm -rf $d
done
echo "== dit_b4 unprofiled (defaults, with the CPU baseline)"
timeout -k 10 900 python3 bench.py > gpurun_out/bench_default.log 2>&1 || { tail -5 gpurun_out/bench_default.log; exit 1; }
grep '^{"metric"' gpurun_out/bench_default.log | tail -1 > $OUT/${TAG}_dit_b4_bs256_bf16_bench.json
ls -la $OUT
