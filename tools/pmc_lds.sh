#!/bin/bash
# LDS pipe counters per kernel (2 GB bench): unaligned stalls, bank and address conflicts, active cycles.  bash tools/pmc_lds.sh [variant tag ...]
export TMPDIR=/tmp PYTHONPATH=$PWD HMSE_BENCH_NO_VERIFY=1 HMSE_BENCH_NO_MANIFEST=1
for V in "${@:-base}"; do
  if [ "$V" = base ]; then unset HMSE_LIB_VARIANT; else export HMSE_LIB_VARIANT=$V; fi
  OUT=gpurun_out/pmc_lds_$V; rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --kernel-trace --pmc SQ_LDS_UNALIGNED_STALL SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_BUSY_CYCLES --output-format csv -d $OUT -- python3 bench.py --no-cpu-baseline --no-other-configs --bytes 2e9 --steps 1 --warmup 0 > $OUT/bench.json 2> $OUT/err.txt || { echo "$V failed"; tail -5 $OUT/err.txt; continue; }
  python3 tools/pmc_summary.py $(find $OUT -name '*counter_collection.csv' | head -1) > gpurun_out/pmc_lds_$V.csv
  echo "== $V"; cut -c1-240 gpurun_out/pmc_lds_$V.csv
done
