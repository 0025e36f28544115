"""Chunk manifest: the reference's packed on-disk records, written by the host from a ShardResult.

Layouts (little-endian, packed) exactly as in the reference:
  ChunkIndex  40 B {sha256[32], lba u32, length u16, refcount u16}        README.md:1263-1270, 2646-2651
  DeltaChunk   8 B {base_lba u32, base_length u16, delta_length u16} + delta_data[]   README.md:2182-2189
  pointer      8 B {target_lba u32, target_length u16, flags u16}          README.md:1312 ("LBA + offset")
`lba` is a byte offset into the blob divided by `lba_unit` (the reference's LBA is a 512-byte SD
sector; the unit is stored in the header and every record in the blob is aligned to it).
The per-chunk map (README.md:1448 "Chunk Map [#301, #302d, ...]") is a u32 index-table slot plus a
type tag per chunk.  `reconstruct()` is the three-branch read path of README.md:1621-1675, used
as the end-to-end verifier (VALIDATION_METHODS.md:257 "100 % lossless reconstruction required").
"""
from __future__ import annotations

import hashlib
import struct
import zlib
from dataclasses import dataclass

import numpy as np

from .config import KIND_DELTA, KIND_FULL, KIND_POINTER

MAGIC = b"HMSEMI35"
CHUNK_INDEX_DTYPE = np.dtype([("sha256", "u1", 32), ("lba", "<u4"), ("length", "<u2"), ("refcount", "<u2")])
DELTA_HDR_DTYPE = np.dtype([("base_lba", "<u4"), ("base_length", "<u2"), ("delta_length", "<u2")])
POINTER_DTYPE = np.dtype([("target_lba", "<u4"), ("target_length", "<u2"), ("flags", "<u2")])
MAP_DTYPE = np.dtype([("slot", "<u4"), ("raw_length", "<u2"), ("kind", "u1"), ("pad", "u1")])
assert CHUNK_INDEX_DTYPE.itemsize == 40 and DELTA_HDR_DTYPE.itemsize == 8 and POINTER_DTYPE.itemsize == 8


@dataclass
class Manifest:
    lba_unit: int
    index: np.ndarray      # CHUNK_INDEX_DTYPE [n_unique]
    chunk_map: np.ndarray  # MAP_DTYPE [n_chunks]
    pointers: np.ndarray   # POINTER_DTYPE [n_pointer]
    blob: np.ndarray       # uint8: FULL streams and DeltaChunk records, lba_unit-aligned

    def to_bytes(self) -> bytes:
        hdr = MAGIC + struct.pack("<IIQQQQ", 1, self.lba_unit, len(self.index), len(self.chunk_map), len(self.pointers), self.blob.size)
        return hdr + self.index.tobytes() + self.chunk_map.tobytes() + self.pointers.tobytes() + self.blob.tobytes()

    @staticmethod
    def from_bytes(b: bytes) -> "Manifest":
        assert b[:8] == MAGIC
        ver, unit, nu, nc, npt, nb = struct.unpack_from("<IIQQQQ", b, 8)
        assert ver == 1
        o = 8 + struct.calcsize("<IIQQQQ")
        idx = np.frombuffer(b, CHUNK_INDEX_DTYPE, nu, o); o += nu * 40
        cmap = np.frombuffer(b, MAP_DTYPE, nc, o); o += nc * 8
        ptr = np.frombuffer(b, POINTER_DTYPE, npt, o); o += npt * 8
        blob = np.frombuffer(b, np.uint8, nb, o)
        return Manifest(unit, idx, cmap, ptr, blob)


def build_manifest(res, first_occ_local: np.ndarray | None = None) -> Manifest:
    """Host-side writer for a single-shard ShardResult (all chunks and their first occurrences local)."""
    cuts = res.cuts.cpu().numpy().astype(np.int64)
    n = len(cuts) - 1
    lens = np.diff(cuts)
    uniq = res.uniq_ids.cpu().numpy()
    u = len(uniq)
    off = res.stream_off.cpu().numpy().astype(np.int64)
    kind_u = res.kind.cpu().numpy()
    base = res.base.cpu().numpy() if res.base is not None else np.full(u, -1, np.int64)
    streams = res.streams.cpu().numpy()
    slen = np.diff(off)
    rec_len = slen + np.where(kind_u == KIND_DELTA, 8, 0)
    total = int(rec_len.sum())
    unit = 1
    while (total + unit * u) // unit >= 2**32:
        unit *= 2
    rec_off = np.zeros(u + 1, np.int64)
    np.cumsum((rec_len + unit - 1) // unit * unit, out=rec_off[1:])
    blob = np.zeros(int(rec_off[-1]), np.uint8)
    index = np.zeros(u, CHUNK_INDEX_DTYPE)
    dg = res.digests.cpu().numpy() if res.digests is not None else None
    rc = res.refcount.cpu().numpy() if res.refcount is not None else None
    index["lba"] = rec_off[:-1] // unit
    index["length"] = rec_len
    if dg is not None:
        index["sha256"] = dg[uniq]
        index["refcount"] = np.minimum(rc[uniq], 65535)
    else:
        index["refcount"] = 1
    for k in range(u):  # host packing loop (not the hot path)
        o = int(rec_off[k])
        s = streams[off[k]:off[k + 1]]
        if kind_u[k] == KIND_DELTA:
            b = int(base[k])
            hdr = np.zeros(1, DELTA_HDR_DTYPE)
            hdr["base_lba"] = index["lba"][b]; hdr["base_length"] = index["length"][b]; hdr["delta_length"] = len(s)
            blob[o:o + 8] = np.frombuffer(hdr.tobytes(), np.uint8)
            o += 8
        blob[o:o + len(s)] = s
    cmap = np.zeros(n, MAP_DTYPE)
    cmap["raw_length"] = np.minimum(lens, 65535)
    slot_of = np.full(n, -1, np.int64)
    slot_of[uniq] = np.arange(u)
    if res.first_occ is not None:
        fo = res.first_occ.cpu().numpy() - res.chunk_base
        assert (fo >= 0).all() and (fo < n).all(), "build_manifest needs every first occurrence in this shard"
    else:
        fo = np.arange(n)
    cmap["slot"] = slot_of[fo]
    is_ptr = fo != np.arange(n)
    cmap["kind"] = np.where(is_ptr, KIND_POINTER, kind_u[slot_of[fo]])
    ptr = np.zeros(int(is_ptr.sum()), POINTER_DTYPE)
    tgt = slot_of[fo[is_ptr]]
    ptr["target_lba"] = index["lba"][tgt]; ptr["target_length"] = index["length"][tgt]; ptr["flags"] = KIND_POINTER
    return Manifest(unit, index, cmap, ptr, blob)


def reconstruct(m: Manifest) -> bytes:
    """Read path (README.md:1621-1675): FULL -> inflate; POINTER -> target; DELTA -> inflate with zdict=base."""
    by_lba = {int(e["lba"]): i for i, e in enumerate(m.index)}
    slot_kind = np.zeros(len(m.index), np.uint8)
    own = m.chunk_map["kind"] != KIND_POINTER
    slot_kind[m.chunk_map["slot"][own]] = m.chunk_map["kind"][own]
    cache: dict[int, bytes] = {}

    def raw_of(slot: int) -> bytes:
        if slot in cache:
            return cache[slot]
        e = m.index[slot]
        o = int(e["lba"]) * m.lba_unit
        rec = m.blob[o:o + int(e["length"])].tobytes()
        if slot_kind[slot] == KIND_DELTA:  # DeltaChunk record: 8-byte header, then a stream with the base as dictionary
            base_lba, base_len, dlen = struct.unpack_from("<IHH", rec, 0)
            d = zlib.decompressobj(-15, zdict=raw_of(by_lba[base_lba]))
            out = d.decompress(rec[8:8 + dlen]) + d.flush()
        else:
            d = zlib.decompressobj(-15)
            out = d.decompress(rec) + d.flush()
        assert d.eof and not d.unused_data
        if e["sha256"].any():
            assert hashlib.sha256(out).digest() == e["sha256"].tobytes(), "SHA-256 mismatch on reconstruct"
        cache[slot] = out
        return out

    return b"".join(raw_of(int(s)) for s in m.chunk_map["slot"])
