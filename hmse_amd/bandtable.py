"""L4 band tables on disk: the reference's bucket-list layout — per bucket a 4-byte header
{band_hash u16, count u16} and a list of 3-byte chunk ids (README.md:1937-1945, 1974-1978), one table per band.

Host-side format code (numpy), written from the band keys the L4b kernel returns.  File layout (little-endian):
  magic "HMSEBAND" | u32 version | u32 bands | u32 band_bits | u32 id_bytes (3) | u64 n_ids
  per band: u64 n_headers | headers {band_hash u16, count u16}[n_headers] | ids u8[3 * n_ids]
Buckets appear in ascending band_hash order, ids ascending inside a bucket (= ingest order, so the first id of a bucket is the
earliest chunk in it: the candidate the base rule prefers).  A bucket with more than 65535 ids continues in further headers
with the same band_hash.  bucket = band key & (2^band_bits - 1); a candidate from a bucket is confirmed by comparing the
signatures' band rows (the 32-bit key kept in HBM already decides it on the device)."""
from __future__ import annotations

import struct

import numpy as np

MAGIC = b"HMSEBAND"
HDR_DTYPE = np.dtype([("band_hash", "<u2"), ("count", "<u2")])


def write_band_tables(band_keys: np.ndarray, band_bits: int = 16, signatures: np.ndarray | None = None) -> bytes:
    """Band tables in the reference's bucket-list layout; with `signatures` (u32[n][n_hashes]) a trailing section
    "HMSESIGS" | u32 n_hashes | band keys u32[n][bands] | signatures u32[n][n_hashes] follows, from which a resumed ingest
    rebuilds its content-addressed tables without recomputing a single MinHash (hmse_amd/stream.py resume)."""
    keys = np.ascontiguousarray(band_keys).view(np.uint32).reshape(band_keys.shape)
    n, bands = keys.shape
    if n >= 1 << 24:
        raise ValueError("3-byte chunk ids hold at most 16 777 215 stored chunks (README.md:1943)")
    if not 1 <= band_bits <= 16:
        raise ValueError("band_hash is a u16")
    out = [MAGIC, struct.pack("<IIIIQ", 1, bands, band_bits, 3, n)]
    ids_all = np.arange(n, dtype=np.uint32)
    for b in range(bands):
        bucket = keys[:, b] & np.uint32((1 << band_bits) - 1)
        order = np.argsort(bucket, kind="stable")                 # by bucket, ids ascending inside
        sb = bucket[order]
        starts = np.flatnonzero(np.r_[True, sb[1:] != sb[:-1]]) if n else np.zeros(0, np.int64)
        counts = np.diff(np.r_[starts, n])
        pieces = (counts + 65534) // 65535                         # continuation headers for crowded buckets
        hdr = np.zeros(int(pieces.sum()), HDR_DTYPE)
        hdr["band_hash"] = np.repeat(sb[starts], pieces)
        first = np.cumsum(pieces) - pieces
        full = np.full(hdr.shape[0], 65535, np.int64)
        last_idx = first + pieces - 1
        full[last_idx] = counts - (pieces - 1) * 65535
        hdr["count"] = full
        ids = ids_all[order]
        id3 = np.zeros((n, 3), np.uint8)
        id3[:, 0] = ids & 0xFF; id3[:, 1] = (ids >> 8) & 0xFF; id3[:, 2] = (ids >> 16) & 0xFF
        out += [struct.pack("<Q", hdr.shape[0]), hdr.tobytes(), id3.tobytes()]
    if signatures is not None:
        sg = np.ascontiguousarray(signatures).view(np.uint32).reshape(signatures.shape)
        assert sg.shape[0] == n
        out += [b"HMSESIGS", struct.pack("<I", sg.shape[1]), keys.astype("<u4").tobytes(), sg.astype("<u4").tobytes()]
    return b"".join(out)


def read_signatures(buf: bytes):
    """-> (band keys u32[n][bands], signatures u32[n][n_hashes] or None) from the trailing section of write_band_tables()."""
    assert buf[:8] == MAGIC
    ver, bands, band_bits, id_bytes, n = struct.unpack_from("<IIIIQ", buf, 8)
    o = 8 + struct.calcsize("<IIIIQ")
    for _ in range(bands):
        (nh,) = struct.unpack_from("<Q", buf, o)
        o += 8 + 4 * nh + 3 * n
    if buf[o:o + 8] != b"HMSESIGS":
        return None, None
    (nhash,) = struct.unpack_from("<I", buf, o + 8)
    o += 12
    keys = np.frombuffer(buf, "<u4", n * bands, o).reshape(n, bands); o += 4 * n * bands
    sig = np.frombuffer(buf, "<u4", n * nhash, o).reshape(n, nhash)
    return keys, sig


def read_band_tables(buf: bytes):
    """-> (band_bits, [per band: (band_hash u16[], start i64[], count i64[], ids u32[])]) with continuation headers merged."""
    assert buf[:8] == MAGIC
    ver, bands, band_bits, id_bytes, n = struct.unpack_from("<IIIIQ", buf, 8)
    assert ver == 1 and id_bytes == 3
    o = 8 + struct.calcsize("<IIIIQ")
    tables = []
    for _ in range(bands):
        (nh,) = struct.unpack_from("<Q", buf, o); o += 8
        hdr = np.frombuffer(buf, HDR_DTYPE, nh, o); o += 4 * nh
        id3 = np.frombuffer(buf, np.uint8, 3 * n, o).reshape(n, 3).astype(np.uint32); o += 3 * n
        ids = id3[:, 0] | (id3[:, 1] << 8) | (id3[:, 2] << 16)
        cnt = hdr["count"].astype(np.int64)
        start = np.cumsum(cnt) - cnt
        keep = np.r_[True, hdr["band_hash"][1:] != hdr["band_hash"][:-1]] if nh else np.zeros(0, bool)
        grp = np.cumsum(keep) - 1
        tot = np.bincount(grp, weights=cnt).astype(np.int64) if nh else np.zeros(0, np.int64)
        tables.append((hdr["band_hash"][keep], start[keep], tot, ids))
    return band_bits, tables


def candidates(tables, band: int, band_hash: int) -> np.ndarray:
    """Ids in one bucket, ascending (empty if the bucket does not exist)."""
    h, start, cnt, ids = tables[band]
    i = np.searchsorted(h, band_hash)
    if i >= len(h) or h[i] != band_hash:
        return ids[:0]
    return ids[start[i]: start[i] + cnt[i]]
