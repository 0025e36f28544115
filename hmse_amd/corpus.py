"""Corpus inputs: real corpora from $HMSE_CORPUS_DIR when present, otherwise wiki-synth(seed=42).

The generator is host C (csrc/corpus_synth.c): block b depends only on (seed, profile, b), so each
rank generates exactly its own shard.  Incompressible control: PRNG bytes, seed 0xDEADBEEF
(VALIDATION_METHODS.md:213).  SURVEY.md §8d "Inputs".
"""
from __future__ import annotations

import os

import numpy as np

from . import _lib

BLOCK = 1 << 20
PROFILES = {"wikipedia": 0, "arxiv": 1, "news": 2, "code": 3}
REAL_NAMES = {"enwik8": "enwik8", "enwik9": "enwik9", "wikipedia": "wikipedia.txt", "arxiv": "arxiv.txt",
              "news": "news.txt", "code": "code.txt"}


def wiki_synth(n_bytes: int, seed: int = 42, profile: str = "wikipedia", first_block: int = 0,
               out: np.ndarray | None = None, threads: int | None = None) -> np.ndarray:
    """n_bytes of the synthetic corpus starting at block `first_block` (1 MiB blocks)."""
    n_blocks = -(-n_bytes // BLOCK)
    if out is None:
        out = np.empty(n_blocks * BLOCK, dtype=np.uint8)
    assert out.dtype == np.uint8 and out.size >= n_blocks * BLOCK and out.flags.c_contiguous
    threads = threads or min(32, os.cpu_count() or 1)
    rc = _lib.corpus_lib().hmse_corpus_generate(out.ctypes.data, first_block, n_blocks, BLOCK, seed, PROFILES[profile], threads)
    if rc != 0:
        raise RuntimeError(f"hmse_corpus_generate -> {rc}")
    return out[:n_bytes]


def random_bytes(n_bytes: int, seed: int = 0xDEADBEEF) -> np.ndarray:
    return np.random.Generator(np.random.PCG64(seed)).integers(0, 256, n_bytes, dtype=np.uint8)


def load(name: str, n_bytes: int, first_byte: int = 0, seed: int = 42):
    """(array, source) — a real corpus slice if $HMSE_CORPUS_DIR holds it, else the synthetic one."""
    d = os.environ.get("HMSE_CORPUS_DIR")
    fn = REAL_NAMES.get(name, name)
    if d and os.path.exists(os.path.join(d, fn)):
        mm = np.memmap(os.path.join(d, fn), dtype=np.uint8, mode="r")
        return np.ascontiguousarray(mm[first_byte:first_byte + n_bytes]), f"file:{fn}"
    prof = name if name in PROFILES else "wikipedia"
    assert first_byte % BLOCK == 0
    return wiki_synth(n_bytes, seed, prof, first_byte // BLOCK), f"wiki-synth(seed={seed},profile={prof})"
