#!/bin/bash
# A/B of experiment builds of the library (hmse_amd/csrc/Makefile: libhmse_hip_<tag>.so): bash tools/ab_variants.sh <bytes> <tag> [<tag> ...]
# ("base" = the product library).  Prints the per-stage sums of bench.py for every variant.
BYTES=$1; shift
for V in "$@"; do
  if [ "$V" = base ]; then unset HMSE_LIB_VARIANT; else export HMSE_LIB_VARIANT=$V; fi
  HMSE_BENCH_NO_VERIFY=1 HMSE_BENCH_NO_MANIFEST=1 timeout -k 5 300 python bench.py --bytes $BYTES --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs > gpurun_out/ab_$V.json 2> gpurun_out/ab_$V.err || { echo "$V FAILED"; tail -3 gpurun_out/ab_$V.err; continue; }
  python - <<PY
import json
d=json.load(open("gpurun_out/ab_$V.json")); sr=d["stage_roofline"]
g=lambda pre,suf: sum(v["avg_ms"] for k,v in sr.items() if k.startswith(pre) and k.endswith(suf))
print("%-8s step %.1f ms  cf %.4f | plain %.1f dict %.1f encode %.1f minhash %.1f | " % ("$V", d["ms_per_step"], d["cf"], g("l1_deflate_kernel","false>"), g("l1_deflate_kernel","true>"), sum(v["avg_ms"] for k,v in sr.items() if k.startswith("l1_encode")), g("l4_minhash","")) + " ".join("%.1f" % v["avg_ms"] for k,v in sorted(sr.items(), key=lambda kv:-kv[1]["avg_ms"])[:6]))
PY
done
