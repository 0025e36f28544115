"""Document-aligned segments and shards (BASELINE.json north_star: "the corpus shards naturally by document across the 8 GPUs";
README.md:1431-1497 treats the article as the unit of a read).

The pipeline's unit of independence is the SEGMENT: cut-point resolution restarts at every segment start and a segment's end is a
forced cut (include/hmse.h hmse_l2_cdc, `seg_off`).  The default segmentation is fixed 4 MiB runs; this module builds `seg_off`
from the documents instead: a segment is a run of WHOLE documents of at most `seg_size` bytes (a document longer than that is cut
every `seg_size` bytes), so no chunk ever spans two documents' segments, a document never straddles two ranks, and the chunking of
a document does not depend on what was ingested in front of it beyond its own segment.  Shards are runs of whole segments, dealt so
that every rank gets about the same number of bytes.  The CPU oracle takes the same `seg_off` (orc.cdc(data, cfg, seg_off)), so
parity holds for any segmentation.  Host-side plumbing over torch; the kernels are the same.
"""
from __future__ import annotations

import numpy as np
import torch


def document_starts(data: torch.Tensor, sep: bytes = b"\n= ", skip: int = 1, slab: int = 256 << 20) -> np.ndarray:
    """Byte offsets at which documents start: 0 and every occurrence of `sep` + `skip` (wiki-synth and MediaWiki dumps start an
    article with a "= Title =" line: the pattern is the newline in front of it, the document starts one byte later).  Scans the
    (device or host) tensor in slabs; returns ascending int64 offsets on the host."""
    n = data.numel()
    pat = list(sep)
    if not pat or skip < 0 or skip > len(pat):
        raise ValueError("document_starts: need a non-empty separator and 0 <= skip <= len(sep)")
    out = [np.zeros(1, np.int64)]
    for a in range(0, n, slab):
        b = min(n, a + slab + len(pat) - 1)
        x = data[a:b]
        m = x[: x.numel() - len(pat) + 1] == pat[0]
        for j in range(1, len(pat)):
            m &= x[j: x.numel() - len(pat) + 1 + j] == pat[j]
        idx = m.nonzero().flatten().cpu().numpy().astype(np.int64) + a + skip
        out.append(idx[(idx > 0) & (idx < n)])
    return np.unique(np.concatenate(out))


def document_seg_off(doc_starts: np.ndarray, n: int, seg_size: int) -> np.ndarray:
    """Segments of whole documents: each segment starts at a document start (or, inside a document longer than seg_size, seg_size
    bytes after the previous boundary) and runs to the LAST document start within seg_size bytes.  Returns int64 seg_off[n_seg + 1]."""
    d = np.asarray(doc_starts, np.int64)
    offs = [0]
    s = 0
    while s < n:
        lim = s + seg_size
        if lim >= n:
            offs.append(n)
            break
        k = np.searchsorted(d, lim, side="right") - 1          # last document start <= lim
        nxt = int(d[k]) if k >= 0 and d[k] > s else lim         # none inside: a long document, capped at seg_size
        offs.append(nxt)
        s = nxt
    return np.asarray(offs, np.int64)


def deal_segments(seg_off: np.ndarray, world: int) -> list:
    """Shards = runs of whole segments with about n / world bytes each: rank r gets segments [i_r, i_{r+1}), i_r = the first
    segment starting at or after r n / world.  Returns [(seg_lo, seg_hi)] per rank (possibly empty for a tiny corpus)."""
    so = np.asarray(seg_off, np.int64)
    n, n_seg = int(so[-1]), len(so) - 1
    cut = [int(np.searchsorted(so[:-1], -(-r * n // world), side="left")) for r in range(world)] + [n_seg]
    cut = np.maximum.accumulate(np.minimum(cut, n_seg))
    return [(int(cut[r]), int(cut[r + 1])) for r in range(world)]


def shard_seg_off(seg_off: np.ndarray, seg_lo: int, seg_hi: int, device=None) -> tuple:
    """(byte_lo, byte_hi, the shard's own seg_off (relative to byte_lo) as a device tensor for ingest_shard / ops.l2_cdc)."""
    so = np.asarray(seg_off, np.int64)
    lo, hi = int(so[seg_lo]), int(so[seg_hi])
    rel = so[seg_lo: seg_hi + 1] - lo if seg_hi > seg_lo else np.zeros(2, np.int64)
    t = torch.from_numpy(np.ascontiguousarray(rel))
    return lo, hi, (t.to(device) if device is not None else t)
