#!/bin/bash
# quick GPU loop for kernel work: DEFLATE parity tests, then the 10 GB bench's per-kernel table.  bash tools/quick_check.sh <tag>
set -o pipefail
TAG=${1:-q}
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ingest.py -x -q > gpurun_out/${TAG}_tests.log 2>&1 || { tail -15 gpurun_out/${TAG}_tests.log; exit 1; }
tail -1 gpurun_out/${TAG}_tests.log
HMSE_BENCH_NO_VERIFY=${NOVERIFY:-1} HMSE_BENCH_NO_MANIFEST=1 timeout -k 5 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err || { tail -5 gpurun_out/${TAG}_bench.err; exit 1; }
python - <<PY
import json
d=json.load(open("gpurun_out/${TAG}_bench.json"))
print({k:d[k] for k in ("value","ms_per_step","cf","cf_payload")})
sr=d["stage_roofline"]
g=lambda pre,suf: sum(v["avg_ms"] for k,v in sr.items() if k.startswith(pre) and k.endswith(suf))
print("plain %.1f  dict %.1f  encode %.1f  minhash %.1f" % (g("l1_deflate_kernel","false>"), g("l1_deflate_kernel","true>"), sum(v["avg_ms"] for k,v in sr.items() if k.startswith("l1_encode")), g("l4_minhash","")))
for k,v in sorted(sr.items(), key=lambda kv:-kv[1]["avg_ms"])[:9]: print("  %7.2f %s" % (v["avg_ms"], k))
PY
