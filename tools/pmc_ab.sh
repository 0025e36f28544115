#!/bin/bash
# SQ instruction counters of two library variants on the same 2 GB bench (A/B of instruction counts, not of time):
#   bash tools/pmc_ab.sh <tagA|base> <tagB>     -> gpurun_out/pmc_ab_<tag>.csv (per kernel: VALU / SALU / LDS instructions, wave cycles, busy cycles)
export TMPDIR=/tmp PYTHONPATH=$PWD HMSE_BENCH_NO_VERIFY=1 HMSE_BENCH_NO_MANIFEST=1
for V in "$@"; do
  if [ "$V" = base ]; then unset HMSE_LIB_VARIANT; else export HMSE_LIB_VARIANT=$V; fi
  OUT=gpurun_out/pmc_ab_$V; rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT -- python3 bench.py --no-cpu-baseline --no-other-configs --bytes 2e9 --steps 1 --warmup 0 > $OUT/bench.json 2> $OUT/err.txt || { echo "$V failed"; tail -3 $OUT/err.txt; continue; }
  python3 tools/pmc_summary.py $(find $OUT -name '*counter_collection.csv' | head -1) > gpurun_out/pmc_ab_$V.csv
  echo "== $V"; grep "l1_deflate_kernel\|^kernel" gpurun_out/pmc_ab_$V.csv | cut -c1-260
done
