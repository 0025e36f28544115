"""hmse_amd — MI355X-native HMSE ingest hot path (L2 FastCDC -> L3 SHA-256 -> L4 MinHash/LSH -> L1 DEFLATE)."""
from .config import ABLATIONS, IngestConfig, KIND_DELTA, KIND_FULL, KIND_POINTER  # noqa: F401

__all__ = ["IngestConfig", "ABLATIONS", "KIND_FULL", "KIND_POINTER", "KIND_DELTA"]
