"""Per-kernel sums of rocprofv3 PMC counters from one or more counter_collection.csv files (this library's kernels only)."""
import csv, re, sys
from collections import defaultdict
OURS = re.compile(r"(l2_hash_kernel|l3_sha256_kernel|l4_minhash_kernel|l1_deflate_kernel|l1_encode_kernel|l1_inflate_kernel)")
acc = defaultdict(lambda: defaultdict(float))
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path, newline="")):
        if OURS.search(r["Kernel_Name"]):
            k = re.sub(r"\(.*$", "", re.sub(r"^void ", "", r["Kernel_Name"])).replace("dfl::", "").replace("ifl::", "").replace(", ", ",")
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
names = sorted({c for v in acc.values() for c in v})
print("kernel," + ",".join(names))
for k in sorted(acc):
    print(k + "," + ",".join(f"{acc[k].get(c, 0):.4g}" for c in names))
