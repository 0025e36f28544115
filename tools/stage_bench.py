"""Per-stage timing of the HIP hot path on one GPU (developer tool; bench.py is the contract)."""
import argparse
import json
import time

import numpy as np
import torch

from hmse_amd import IngestConfig, corpus, ops


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e-3)
    return r, min(ts)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mib", type=int, default=1024)
    ap.add_argument("--stages", default="l2,l3,dedup,l4,lsh,l1")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--depth", type=int, default=0)
    a = ap.parse_args()
    cfg = IngestConfig(chain_depth=a.depth)
    n = a.mib << 20
    t0 = time.time(); host = corpus.wiki_synth(n); tg = time.time() - t0
    dev = torch.device("cuda:0")
    d = torch.from_numpy(host).to(dev)
    res = {"bytes": n, "gen_s": round(tg, 2)}
    st = a.stages.split(",")
    cuts, t = timed(lambda: ops.l2_cdc(d, cfg), a.reps)
    nch = cuts.numel() - 1
    res["l2"] = {"s": t, "GBps": n / t / 1e9, "chunks": nch, "avg": n / max(nch, 1)}
    if "l3" in st:
        dg, t = timed(lambda: ops.l3_sha256(d, cuts), a.reps)
        res["l3"] = {"s": t, "GBps": n / t / 1e9}
        (fo, rc), t = timed(lambda: ops.l3_dedup(dg), a.reps)
        uniq = (fo == torch.arange(nch, device=dev)).nonzero().flatten()
        lens = cuts[1:] - cuts[:-1]
        ub = int(lens[uniq].sum().item())
        res["dedup"] = {"s": t, "unique_chunks": uniq.numel(), "unique_bytes_frac": ub / n}
    else:
        uniq = torch.arange(nch, device=dev); ub = n
    if "l4" in st:
        sig, t = timed(lambda: ops.l4_minhash(d, cuts, cfg, uniq), a.reps)
        res["l4"] = {"s": t, "GBps_unique": ub / t / 1e9}
        (keys, base), t = timed(lambda: ops.l4_lsh(sig, cfg), a.reps)
        res["lsh"] = {"s": t, "hit_rate": float((base >= 0).float().mean().item())}
    else:
        base = None
    if "l1" in st:
        try:
            (out, off, kind), t = timed(lambda: ops.l1_deflate(d, cuts, cfg, uniq, base), a.reps)
            res["l1"] = {"s": t, "GBps_unique": ub / t / 1e9, "out_bytes": out.numel(), "cf_payload": ub / max(out.numel(), 1),
                         "delta_frac": float((kind == 2).float().mean().item())}
        except Exception as e:  # noqa: BLE001
            res["l1"] = {"error": str(e)}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
