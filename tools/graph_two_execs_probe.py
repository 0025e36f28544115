"""Probe: does a SECOND instantiated hipGraph disturb the replays of the first?  (Round 3: the two-phase chain of a multi-rank
stream keeps two graphs per piece size alive — phase A and phase B.)   python tools/graph_two_execs_probe.py <variant>
Variants: base (capture A, replay, capture B, replay A ...), both_first (capture A and B before any replay), shared_pool.
Prints, after every step, whether phase A's outputs (exchange row + cut count) still equal the eagerly computed ones."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from hmse_amd import IngestConfig, corpus, stream_dist

variant = sys.argv[1] if len(sys.argv) > 1 else "base"
dev = torch.device("cuda:0")
cfg = IngestConfig(seg_size=1 << 20)
P = 2 << 20
data = torch.from_numpy(corpus.wiki_synth(4 * P, seed=11))
s = stream_dist.DistStreamIngest(cfg, data.numel(), P, dev, 1, 0, graph=False)
s.data.copy_(data)
seg_off = s._entry(P)[0]


def A():
    s._call_hash(P, seg_off)


def B():
    s._call_encode(P, s._row)


def check(tag, want_row, want_state):
    torch.cuda.synchronize()
    n = int(s._state[2].item())
    ok = torch.equal(s._row[: 32 + 32 * n], want_row[: 32 + 32 * n]) and n == int(want_state[2].item())
    print(f"[{variant}] {tag}: phase A outputs {'OK' if ok else 'WRONG'} (n_new {n}, status {int(s._state[7].item())})", flush=True)
    if not ok:
        os._exit(3)     # stop before a chain with garbage arguments is launched again
    return ok


A(); B()                                   # piece 0 eagerly (kernel attributes set outside any capture)
torch.cuda.synchronize()
st1 = s._state.clone()                     # state in front of piece 1
A(); torch.cuda.synchronize()
row1, stA = s._row.clone(), s._state.clone()
pool = torch.cuda.graph_pool_handle() if variant == "shared_pool" else None


def capture(fn):
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, pool=pool):
        fn()
    return g


def reset():
    s._state.copy_(st1); s._row.zero_(); torch.cuda.synchronize()


reset(); gA = capture(A)
if variant == "both_first":
    s._state.copy_(stA); gB = capture(B); reset()
gA.replay(); check("A replay 1", row1, stA)
reset(); gA.replay(); check("A replay 2 (nothing in between)", row1, stA)
if variant != "both_first":
    s._state.copy_(stA); gB = capture(B)
    reset(); gA.replay(); check("A replay after B was CAPTURED", row1, stA)
reset(); gA.replay(); torch.cuda.synchronize(); gB.replay(); torch.cuda.synchronize()
print(f"[{variant}] B replayed: state {s._state.tolist()[:10]}", flush=True)
reset(); gA.replay(); check("A replay after B was REPLAYED", row1, stA)
reset(); gA.replay(); check("A replay once more", row1, stA)
