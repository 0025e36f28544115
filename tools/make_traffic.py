"""profiles/traffic.json from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) of `bench.py --steps 1 --warmup 0`:
python tools/make_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <total_bytes> [trimmed-csv-prefix]

Per kernel of this library: counter sums over its dispatches / number of dispatches (= per launch), and
hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE counts half of a wide coalesced read —
/opt/skills/guides/MI355X_MICROARCH.md, HBM / rocprofv3 section)."""
import csv, json, os, re, sys

OURS = re.compile(r"(l2_hash_kernel|l2_resolve_kernel|l3_sha256_kernel|l4_minhash_kernel|l1_deflate_kernel|l1_encode_kernel|l1_inflate_kernel|assemble_kernel|lsh_|fo_insert|dedup_lookup)")


def short(name: str) -> str:
    n = re.sub(r"^void ", "", name)
    n = re.sub(r"\(.*$", "", n)                       # drop the argument list
    n = re.sub(r"^(dfl|ifl)::", "", n)
    n = n.replace(", ", ",")
    if n.startswith("l4_minhash_kernel"):
        n = "l4_minhash_kernel"                       # one name whatever the launch variant (bench.py STAGE_NAMES)
    return n


def load(path, trimmed=None):
    acc = {}
    rows = []
    with open(path, newline="") as f:
        rd = csv.DictReader(f)
        for r in rd:
            if not OURS.search(r["Kernel_Name"]):
                continue
            k = short(r["Kernel_Name"])
            e = acc.setdefault(k, [0.0, set()])
            e[0] += float(r["Counter_Value"])
            e[1].add(r["Dispatch_Id"])
            rows.append((r["Dispatch_Id"], k, r["Counter_Name"], r["Counter_Value"], r["Start_Timestamp"], r["End_Timestamp"]))
    if trimmed:
        with open(trimmed, "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Dispatch_Id", "Kernel", "Counter_Name", "Counter_Value", "Start_Timestamp", "End_Timestamp"])
            w.writerows(rows)
    return {k: (v[0] / max(1, len(v[1])), len(v[1])) for k, v in acc.items()}


def main():
    fetch_csv, write_csv, total = sys.argv[1], sys.argv[2], int(sys.argv[3])
    pre = sys.argv[4] if len(sys.argv) > 4 else None
    fe = load(fetch_csv, pre + "_pmc_FETCH_SIZE.csv" if pre else None)
    wr = load(write_csv, pre + "_pmc_WRITE_SIZE.csv" if pre else None)
    out = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes) -- python3 bench.py --no-cpu-baseline "
                     "--steps 1 --warmup 0; per launch: hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE counts half of a wide "
                     "coalesced read, /opt/skills/guides/MI355X_MICROARCH.md HBM section; exact for l2_hash_kernel's 16-B/lane stream, "
                     "uncalibrated for per-lane 64-B blocks such as l3_sha256_kernel's, where the raw value already equals the algorithmic bytes)",
           "total_bytes": total, "round": 3, "kernels": {}}
    for k in sorted(set(fe) | set(wr)):
        f, nf = fe.get(k, (0.0, 0))
        w, nw = wr.get(k, (0.0, 0))
        out["kernels"][k] = {"fetch_size_kb": f, "write_size_kb": w, "launches": max(nf, nw), "hbm_bytes": int((2 * f + w) * 1024)}
    json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic.json"), "w"), indent=1)
    for k, v in out["kernels"].items():
        print(f"{k:70s} {v['hbm_bytes'] / 1e9:8.2f} GB  ({v['launches']} launches)")


if __name__ == "__main__":
    main()
