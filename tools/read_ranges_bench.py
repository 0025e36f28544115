"""Request-driven read (README.md:1444-1448, 1621-1675, gate 1329): latency per request and aggregate rate of read.StoreReader.
    python tools/read_ranges_bench.py [store MiB] [requests] [request KiB]
Ingests wiki-synth, packs the manifest, opens the store once, then (a) N single requests one by one (latency: chunk map lookup,
closure, ONE inflate call over the closure, assembly, SHA-256 of the touched chunks, host sync), (b) the same N requests as one batch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from hmse_amd import IngestConfig, corpus, ingest, manifest, read

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
nreq = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
rk = int(sys.argv[3]) if len(sys.argv) > 3 else 32
dev = torch.device("cuda:0")
cfg = IngestConfig()
host = corpus.wiki_synth(mib << 20, seed=42)
res = ingest.ingest_shard(torch.from_numpy(host).to(dev), cfg)
m = manifest.build_manifest(res)
del res
t0 = time.perf_counter(); rd = read.StoreReader(m, dev); torch.cuda.synchronize(); t_open = time.perf_counter() - t0
rng = np.random.default_rng(1329)
off = rng.integers(0, host.size - (rk << 10), nreq)
reqs = [(int(o), rk << 10) for o in off]
rd.read_ranges(reqs[:8]); torch.cuda.synchronize()
lat, dec, pulled = [], 0, 0
for rq in reqs:
    t0 = time.perf_counter(); (g,) = rd.read_ranges([rq]); torch.cuda.synchronize(); lat.append(time.perf_counter() - t0)
    dec += rd.last["bytes_decoded"]; pulled += rd.last["dictionaries_pulled_in"]
ok = all(np.array_equal(rd.read_ranges([rq])[0].cpu().numpy(), host[rq[0]: rq[0] + rq[1]]) for rq in reqs[:50])
lat = np.array(lat) * 1e3
t0 = time.perf_counter(); out = rd.read_ranges(reqs); torch.cuda.synchronize(); tb = time.perf_counter() - t0
print(f"store {mib} MiB ({len(m.index)} records, {len(m.chunk_map)} chunks), opened in {t_open * 1e3:.0f} ms; {nreq} requests of {rk} KiB, SHA-256 of every touched chunk verified")
print(f"one by one: median {np.median(lat):.2f} ms, p95 {np.percentile(lat, 95):.2f} ms, mean {lat.mean():.2f} ms per request; decoded {dec / nreq / 1024:.1f} KiB per request "
      f"({pulled / nreq:.2f} dictionaries pulled in per request); first 50 equal the input: {ok}")
print(f"one batch of {nreq}: {tb * 1e3:.1f} ms = {nreq * (rk << 10) / tb / 2**30:.2f} GiB/s of requested bytes; decoded {rd.last['bytes_decoded'] / 2**20:.1f} MiB for {rd.last['bytes_requested'] / 2**20:.1f} MiB requested")
