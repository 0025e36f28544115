#!/bin/bash
# SQ/TA counters of the two inflate decoders on a 1 GiB shard (tools/inflate_ab.py); run on the GPU box: tools/pmc_inflate.sh
set -o pipefail
export TMPDIR=/tmp PYTHONPATH=$PWD
O=gpurun_out/pmc_ab; mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES --output-format csv -d $O/p1 -- python3 tools/inflate_ab.py ${1:-1024} > $O/p1.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_BUSY_CYCLES TA_BUSY_avr --output-format csv -d $O/p2 -- python3 tools/inflate_ab.py ${1:-1024} > $O/p2.log 2>&1 || exit 1
python3 - <<'PY'
import csv, glob, collections
for d in ("p1", "p2"):
    f = glob.glob(f"gpurun_out/pmc_ab/{d}/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"]
        if "inflate" not in kn: continue
        key = "lanes" if "lanes" in kn else "wave"
        acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[(key, r["Counter_Name"])] += 1
    for key in acc:
        print(d, key, {c: f"{v / cnt[(key, c)]:.4g}" for c, v in acc[key].items()})
PY
