"""profiles/sq_valu.json from the per-kernel SQ counter sums (tools/pmc_summary.py output): per kernel family the share of
SIMD-busy cycles in which a VALU instruction was executing and the instruction mix.  bench.py puts it into roofline.valu_issue.
python tools/make_sq.py <pmc_sq_counters.csv>"""
import csv, json, os, sys

lines = [ln.rstrip("\n") for ln in open(sys.argv[1]) if ln.strip()]
cols = lines[0].split(",")[1:]
rows = []
for ln in lines[1:]:                                   # kernel names contain commas (template arguments): numbers are the last fields
    parts = ln.split(",")
    rows.append({"kernel": ",".join(parts[: len(parts) - len(cols)]), **dict(zip(cols, parts[len(parts) - len(cols):]))})
fam = {}
for r in rows:
    k = r["kernel"]
    name = "l1_deflate_kernel (match)" if k.startswith("l1_deflate_kernel") else k.split("<")[0]
    f = fam.setdefault(name, {})
    for c, v in r.items():
        if c != "kernel":
            f[c] = f.get(c, 0.0) + float(v)
WAVES_PER_SIMD = {"l1_deflate_kernel (match)": 8, "l4_minhash_kernel": 8, "l1_encode_kernel": 8, "l1_inflate_kernel": 7, "l2_hash_kernel": 7}  # resident (launch bounds / LDS)
out = {"source": "rocprofv3 --pmc, two passes at 2 GB (tools/profile_round.sh); valu_active_per_wave_cycle = SQ_ACTIVE_INST_VALU / "
                 "SQ_WAVE_CYCLES (share of a resident wave's cycles with one of ITS vector instructions executing); x resident waves per "
                 "SIMD = valu_busy_per_simd, the SIMD's VALU utilisation (1.0 = the vector ALU never idles); *_per_valu = instruction mix",
       "round": 3, "kernels": {}}
for name, f in sorted(fam.items()):
    wc = f.get("SQ_WAVE_CYCLES", 0.0)
    e = {}
    if wc:
        e["valu_active_per_wave_cycle"] = round(f.get("SQ_ACTIVE_INST_VALU", 0.0) / wc, 4)
        if name in WAVES_PER_SIMD:
            e["waves_per_simd"] = WAVES_PER_SIMD[name]
            e["valu_busy_per_simd"] = round(WAVES_PER_SIMD[name] * f.get("SQ_ACTIVE_INST_VALU", 0.0) / wc, 3)
        e["lds_active_per_wave_cycle"] = round(f.get("SQ_ACTIVE_INST_LDS", 0.0) / wc, 4)
    if f.get("SQ_INSTS_VALU"):
        e["salu_per_valu"] = round(f.get("SQ_INSTS_SALU", 0.0) / f["SQ_INSTS_VALU"], 3)
        e["lds_per_valu"] = round(f.get("SQ_INSTS_LDS", 0.0) / f["SQ_INSTS_VALU"], 3)
        if f.get("SQ_INSTS_LDS"):
            e["lds_bank_conflict_cycles_per_lds_inst"] = round(f.get("SQ_LDS_BANK_CONFLICT", 0.0) / f["SQ_INSTS_LDS"], 3)
    out["kernels"][name] = e
    print(name, e)
json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "sq_valu.json"), "w"), indent=1)
