"""ctypes binding of the CPU oracle (oracle/_build/liborc.so).  TEST INFRASTRUCTURE ONLY.

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg — never from
hmse_amd/ (the product path).  See oracle/hmse_oracle.h for the parity-pinning status.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liborc.so")


class Cfg(C.Structure):
    """Mirror of `hmse_cfg` (include/hmse.h)."""

    _fields_ = [(n, C.c_uint32) for n in (
        "struct_size", "min_size", "avg_size", "max_size", "norm_level", "seg_size",
        "n_hashes", "shingle", "seed_base", "bands", "rows", "band_bits",
        "level", "chain_depth", "layers", "delta_max_ratio_pct")]


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, f) for f in ("hmse_oracle.c", "hmse_oracle_deflate.c", "hmse_oracle_inflate.c", "hmse_oracle.h")]
    stale = (not os.path.exists(_LIB_PATH)) or any(
        os.path.exists(s) and os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        u8p, u16p, u32p, u64p, i64p = (C.POINTER(t) for t in (C.c_uint8, C.c_uint16, C.c_uint32, C.c_uint64, C.c_int64))
        cfgp = C.POINTER(Cfg)
        L.orc_cfg_default.argtypes = [cfgp]
        L.orc_gear_table.argtypes = [u64p]
        L.orc_cdc_masks.argtypes = [cfgp, u64p, u64p]
        L.orc_cdc.restype = C.c_uint64
        L.orc_cdc.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, cfgp, C.c_void_p, C.c_uint64]
        L.orc_cdc_reference_skeleton.restype = C.c_uint64
        L.orc_cdc_reference_skeleton.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64]
        L.orc_sha256.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]
        L.orc_sha256_chunks.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
        L.orc_dedup.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
        L.orc_murmur3_x86_32.restype = C.c_uint32
        L.orc_murmur3_x86_32.argtypes = [C.c_void_p, C.c_int, C.c_uint32]
        L.orc_minhash.argtypes = [C.c_void_p, C.c_uint64, cfgp, C.c_void_p]
        L.orc_minhash_chunks.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, cfgp, C.c_void_p]
        L.orc_lsh.argtypes = [C.c_void_p, C.c_uint64, cfgp, C.c_void_p, C.c_void_p]
        L.orc_deflate.restype = C.c_int64
        L.orc_deflate.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, cfgp, C.c_void_p, C.c_uint64]
        L.orc_deflate_bound.restype = C.c_uint32
        L.orc_deflate_bound.argtypes = [C.c_uint32]
        L.orc_deflate_chunks.restype = C.c_int
        L.orc_deflate_chunks.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, cfgp,
                                         C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
        L.orc_deflate_matches.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, cfgp, C.c_void_p, C.c_void_p]
        L.orc_inflate.restype = C.c_int
        L.orc_inflate.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]
        L.orc_inflate_chunks.restype = C.c_uint64
        L.orc_inflate_chunks.argtypes = [C.c_void_p] * 5 + [C.c_uint64] + [C.c_void_p] * 3
        _lib = L
    return _lib


def default_cfg(**kw) -> Cfg:
    c = Cfg()
    lib().orc_cfg_default(C.byref(c))
    for k, v in kw.items():
        if not hasattr(c, k):
            raise AttributeError(k)
        setattr(c, k, v)
    return c


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _u8(buf) -> np.ndarray:
    a = np.frombuffer(buf, dtype=np.uint8) if not isinstance(buf, np.ndarray) else buf
    return np.ascontiguousarray(a, dtype=np.uint8)


def gear_table() -> np.ndarray:
    t = np.zeros(256, dtype=np.uint64)
    lib().orc_gear_table(t.ctypes.data_as(C.POINTER(C.c_uint64)))
    return t


def cdc_masks(cfg: Cfg):
    a, b = C.c_uint64(), C.c_uint64()
    lib().orc_cdc_masks(C.byref(cfg), C.byref(a), C.byref(b))
    return a.value, b.value


def seg_offsets(n: int, seg: int) -> np.ndarray:
    k = max(1, -(-n // seg)) if n else 1
    off = np.minimum(np.arange(k + 1, dtype=np.uint64) * np.uint64(seg), np.uint64(n))
    return off


def cdc(data, cfg: Cfg, seg_off: np.ndarray | None = None) -> np.ndarray:
    d = _u8(data)
    n = d.size
    if seg_off is None:
        seg_off = seg_offsets(n, cfg.seg_size)
    seg_off = np.ascontiguousarray(seg_off, dtype=np.uint64)
    cap = n // max(1, cfg.min_size) + len(seg_off) + 2
    cuts = np.zeros(cap, dtype=np.uint64)
    nc = lib().orc_cdc(_p(d), n, _p(seg_off), len(seg_off) - 1, C.byref(cfg), _p(cuts), cap)
    assert nc < cap
    return cuts[: nc + 1].copy()


def cdc_reference_skeleton(data) -> np.ndarray:
    d = _u8(data)
    cap = d.size // 1024 + 4
    cuts = np.zeros(cap, dtype=np.uint64)
    nc = lib().orc_cdc_reference_skeleton(_p(d), d.size, _p(cuts), cap)
    return cuts[: nc + 1].copy()


def sha256(data) -> bytes:
    d = _u8(data)
    out = np.zeros(32, dtype=np.uint8)
    lib().orc_sha256(_p(d), d.size, _p(out))
    return out.tobytes()


def sha256_chunks(data, cuts: np.ndarray) -> np.ndarray:
    d = _u8(data)
    cuts = np.ascontiguousarray(cuts, dtype=np.uint64)
    n = len(cuts) - 1
    out = np.zeros((n, 32), dtype=np.uint8)
    lib().orc_sha256_chunks(_p(d), _p(cuts), n, _p(out))
    return out


def dedup(digests: np.ndarray):
    dg = np.ascontiguousarray(digests, dtype=np.uint8).reshape(-1, 32)
    n = dg.shape[0]
    fo = np.zeros(n, dtype=np.uint64)
    rc = np.zeros(n, dtype=np.uint32)
    lib().orc_dedup(_p(dg), n, _p(fo), _p(rc))
    return fo, rc


def murmur3(key: bytes, seed: int) -> int:
    b = np.frombuffer(bytes(key) + b"\0", dtype=np.uint8).copy()
    return int(lib().orc_murmur3_x86_32(_p(b), len(key), seed & 0xFFFFFFFF))


def minhash(data, cfg: Cfg) -> np.ndarray:
    d = _u8(data)
    sig = np.zeros(cfg.n_hashes, dtype=np.uint32)
    lib().orc_minhash(_p(d), d.size, C.byref(cfg), _p(sig))
    return sig


def minhash_chunks(data, cuts, cfg: Cfg, chunk_ids=None) -> np.ndarray:
    d = _u8(data)
    cuts = np.ascontiguousarray(cuts, dtype=np.uint64)
    ids = None if chunk_ids is None else np.ascontiguousarray(chunk_ids, dtype=np.uint64)
    n = (len(cuts) - 1) if ids is None else len(ids)
    sig = np.zeros((n, cfg.n_hashes), dtype=np.uint32)
    lib().orc_minhash_chunks(_p(d), _p(cuts), _p(ids), n, C.byref(cfg), _p(sig))
    return sig


def lsh(sig: np.ndarray, cfg: Cfg):
    s = np.ascontiguousarray(sig, dtype=np.uint32).reshape(-1, cfg.n_hashes)
    n = s.shape[0]
    keys = np.zeros((n, cfg.bands), dtype=np.uint32)
    base = np.zeros(n, dtype=np.int64)
    lib().orc_lsh(_p(s), n, C.byref(cfg), _p(keys), _p(base))
    return keys, base


def deflate(chunk, cfg: Cfg, zdict=None) -> bytes:
    c = _u8(chunk)
    d = _u8(zdict) if zdict is not None and len(zdict) else None
    cap = c.size + 64
    out = np.zeros(cap, dtype=np.uint8)
    n = lib().orc_deflate(_p(c), c.size, _p(d), 0 if d is None else d.size, C.byref(cfg), _p(out), cap)
    if n < 0:
        raise RuntimeError(f"orc_deflate -> {n}")
    return out[:n].tobytes()


def deflate_matches(chunk, cfg: Cfg, zdict=None):
    c = _u8(chunk)
    d = _u8(zdict) if zdict is not None and len(zdict) else None
    ml = np.zeros(c.size, dtype=np.uint16)
    md = np.zeros(c.size, dtype=np.uint16)
    lib().orc_deflate_matches(_p(c), c.size, _p(d), 0 if d is None else d.size, C.byref(cfg), _p(ml), _p(md))
    return ml, md


def deflate_chunks(data, cuts, cfg: Cfg, chunk_ids=None, base=None):
    d = _u8(data)
    cuts = np.ascontiguousarray(cuts, dtype=np.uint64)
    ids = None if chunk_ids is None else np.ascontiguousarray(chunk_ids, dtype=np.uint64)
    n = (len(cuts) - 1) if ids is None else len(ids)
    b = None if base is None else np.ascontiguousarray(base, dtype=np.int64)
    sel = np.arange(n, dtype=np.uint64) if ids is None else ids
    cap = int((cuts[sel + 1] - cuts[sel]).sum()) + 5 * n + 64 if n else 64
    out = np.zeros(cap, dtype=np.uint8)
    off = np.zeros(n + 1, dtype=np.uint64)
    kind = np.zeros(n, dtype=np.uint8)
    rc = lib().orc_deflate_chunks(_p(d), _p(cuts), _p(ids), _p(b), n, C.byref(cfg), _p(out), cap, _p(off), _p(kind))
    if rc != 0:
        raise RuntimeError(f"orc_deflate_chunks -> {rc}")
    return out[: int(off[n])].copy(), off, kind


def inflate(stream, raw_len: int, zdict=None):
    """Raw-DEFLATE decode; returns (bytes, 0) or (None, negative reason). README.md:2397-2400."""
    st = _u8(stream)
    d = _u8(zdict) if zdict is not None and len(zdict) else None
    out = np.zeros(max(raw_len, 1), dtype=np.uint8)
    rc = lib().orc_inflate(_p(st), st.size, _p(d), 0 if d is None else d.size, _p(out), raw_len)
    return (out[:raw_len].tobytes(), 0) if rc == 0 else (None, rc)


def inflate_chunks(streams, stream_off, kind, base, raw_len, stream_len=None):
    """Batch form of hmse_l1_inflate -> (raw bytes, raw_off, ok flags)."""
    st = _u8(streams)
    off = np.ascontiguousarray(stream_off, dtype=np.uint64)
    kd = np.ascontiguousarray(kind, dtype=np.uint8)
    n = kd.size
    b = None if base is None else np.ascontiguousarray(base, dtype=np.int64)
    sl = None if stream_len is None else np.ascontiguousarray(stream_len, dtype=np.uint32)
    raw_off = np.zeros(n + 1, dtype=np.uint64)
    np.cumsum(np.asarray(raw_len, dtype=np.uint64), out=raw_off[1:])
    raw = np.zeros(max(int(raw_off[-1]), 1), dtype=np.uint8)
    ok = np.zeros(max(n, 1), dtype=np.uint8)
    lib().orc_inflate_chunks(_p(st), _p(off), _p(sl), _p(kd), _p(b), n, _p(raw_off), _p(raw), _p(ok))
    return raw[: int(raw_off[-1])], raw_off, ok[:n].astype(bool)
