"""Validation harness: the ablation matrix of VALIDATION_METHODS.md:454-471 on the GPU path.

For every corpus profile and layer configuration: n trials (different generator seeds), CF = input / (stored + record overhead)
as mean with 95 % CI (Student t), ingest GiB/s, and the paired layer contributions (Δ CF between nested configurations with CI and
paired t-test p-value), as the reference's analysis plan asks.  python tools/ablation.py [MiB per trial] [trials]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from scipy import stats
from hmse_amd import ABLATIONS, IngestConfig, corpus, ingest

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 512
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda:0")
ORDER = ["l1_only", "l1_cdc", "l1_cdc_dedupe", "full", "l4_only"]
LABEL = {"l1_only": "Baseline (L1)", "l1_cdc": "+CDC (L1+L2)", "l1_cdc_dedupe": "+Dedupe (L1-L3)", "full": "Full (L1-L4)", "l4_only": "L4 only (L2+L4)"}


def ci95(x):
    x = np.asarray(x, float)
    if x.size < 2:
        return float(x.mean()), float("nan")
    return float(x.mean()), float(stats.t.ppf(0.975, x.size - 1) * x.std(ddof=1) / np.sqrt(x.size))


print(f"# ablation matrix: {mib} MiB per trial, n = {trials} trials (generator seeds 42..), 1 x MI355X\n")
for prof in ("wikipedia", "arxiv", "news", "code"):
    cf = {k: [] for k in ORDER}
    rate = {k: [] for k in ORDER}
    for tr in range(trials):
        host, _ = corpus.load(prof, mib << 20, seed=42 + tr)
        data = torch.from_numpy(host).to(dev)
        for k in ORDER:
            cfg = IngestConfig(layers=ABLATIONS[k])
            ingest.ingest_shard(data, cfg, want_stats=False)            # warm the allocator
            torch.cuda.synchronize(); t0 = time.perf_counter()
            res = ingest.ingest_shard(data, cfg)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            st = ingest.merge_stats([res.stats])
            cf[k].append(st["cf"]); rate[k].append(data.numel() / dt / 2**30)
            del res
    print(f"## {prof}\n\n| configuration | CF mean ± 95 % CI | GiB/s |\n|---|---|---|")
    for k in ORDER:
        m, h = ci95(cf[k])
        print(f"| {LABEL[k]} | {m:.3f} ± {h:.3f} | {np.mean(rate[k]):.1f} |")
    print("\n| layer contribution (paired over trials) | Δ CF ± 95 % CI | p (paired t) |\n|---|---|---|")
    for a, b, nm in (("l1_only", "l1_cdc", "L2 (CDC)"), ("l1_cdc", "l1_cdc_dedupe", "L3 (dedupe)"), ("l1_cdc_dedupe", "full", "L4 (similarity + delta)")):
        d = np.asarray(cf[b]) - np.asarray(cf[a])
        m, h = ci95(d)
        p = stats.ttest_rel(cf[b], cf[a]).pvalue if trials > 1 and d.std() > 0 else float("nan")
        print(f"| {nm} | {m:+.3f} ± {h:.3f} | {p:.2g} |")
    print(flush=True)
