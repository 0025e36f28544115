#!/bin/bash
# Profiles of the default bench for one round (run on the GPU box through gpurun, from the repo root):
#   tools/profile_round.sh r2
# 1. rocprofv3 --kernel-trace --stats of `python3 bench.py --no-cpu-baseline --no-other-configs --steps 3 --warmup 1`  -> <tag>_kernel_stats_bench_10GB.csv
# 2. two PMC passes (FETCH_SIZE, WRITE_SIZE; counters in their own runs, with --kernel-trace only) of bench --steps 1 --warmup 0
#    -> tools/make_traffic.py -> profiles/traffic.json + trimmed CSVs
# 3. two SQ passes at 2 GB -> tools/pmc_summary.py -> <tag>_pmc_sq_counters_2GB.csv
# Everything lands under gpurun_out/prof_<tag>/; copy what is to be judged into profiles/<tag>/.
set -o pipefail
TAG=${1:-r2}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp PYTHONPATH=$PWD HMSE_BENCH_NO_VERIFY=1 HMSE_BENCH_NO_MANIFEST=1
echo "[profile] kernel trace" && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 bench.py --no-cpu-baseline --no-other-configs --steps 3 --warmup 1 > $OUT/bench_under_rocprof.json 2> $OUT/kt.err || exit 1
cp $(find $OUT/kt -name '*kernel_stats.csv' | head -1) $OUT/${TAG}_kernel_stats_bench_10GB.csv
for C in FETCH_SIZE WRITE_SIZE; do
  echo "[profile] pmc $C" && rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$C -- python3 bench.py --no-cpu-baseline --no-other-configs --steps 1 --warmup 0 > $OUT/pmc_$C.json 2> $OUT/pmc_$C.err || exit 1
done
F=$(find $OUT/pmc_FETCH_SIZE -name '*counter_collection.csv' | head -1); W=$(find $OUT/pmc_WRITE_SIZE -name '*counter_collection.csv' | head -1)
TOTAL=$(python3 -c "import json;print(json.load(open('$OUT/pmc_FETCH_SIZE.json'))['config']['total_bytes'])")
python3 tools/make_traffic.py $F $W $TOTAL $OUT/${TAG} > $OUT/traffic_summary.txt || exit 1
cp profiles/traffic.json $OUT/traffic.json
echo "[profile] sq counters (2 GB)"
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $OUT/sq1 -- python3 bench.py --no-cpu-baseline --no-other-configs --bytes 2e9 --steps 1 --warmup 0 > $OUT/sq1.json 2> $OUT/sq1.err || exit 1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY --output-format csv -d $OUT/sq2 -- python3 bench.py --no-cpu-baseline --no-other-configs --bytes 2e9 --steps 1 --warmup 0 > $OUT/sq2.json 2> $OUT/sq2.err || exit 1
python3 tools/pmc_summary.py $(find $OUT/sq1 -name '*counter_collection.csv' | head -1) $(find $OUT/sq2 -name '*counter_collection.csv' | head -1) > $OUT/${TAG}_pmc_sq_counters_2GB.csv || exit 1
python3 tools/make_sq.py $OUT/${TAG}_pmc_sq_counters_2GB.csv > $OUT/sq_summary.txt || exit 1
cp profiles/sq_valu.json $OUT/sq_valu.json
echo "[profile] done"
