import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (oracle/ is test infrastructure: only tests may import it)."""
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def corpus_small():
    """4 MiB + change of wiki-synth(seed=42) — covers a segment boundary at the default 4 MiB."""
    from hmse_amd import corpus
    return corpus.wiki_synth(5 * (1 << 20) + 12345, seed=42)


def words_text(n, seed=42, vocab=300):
    rng = np.random.default_rng(seed)
    words = [bytes(rng.integers(97, 123, rng.integers(2, 9), dtype=np.uint8)) for _ in range(vocab)]
    out = b" ".join(words[i] for i in rng.integers(0, vocab, n // 5 + 8))
    return np.frombuffer(out[:n], dtype=np.uint8).copy()
