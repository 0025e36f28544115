"""IngestConfig — the reference's compile-time constants as one frozen dataclass.

Mirrors `hmse_cfg` (include/hmse.h) and the `#define`s of the reference skeletons
(README.md:2354-2355, 2444-2447, 2575-2576, 2644; SURVEY.md §5 "Config / flags").
"""
from __future__ import annotations

from dataclasses import asdict, dataclass, replace

from ._lib import HmseCfg

LAYER_L1, LAYER_L2, LAYER_L3, LAYER_L4 = 1, 2, 4, 8
KIND_FULL, KIND_POINTER, KIND_DELTA = 0, 1, 2

# ablation matrix (VALIDATION_METHODS.md:458-464) == degradation modes (README.md:745-770)
ABLATIONS = {
    "l1_only": LAYER_L1,
    "l1_cdc": LAYER_L1 | LAYER_L2,
    "l1_cdc_dedupe": LAYER_L1 | LAYER_L2 | LAYER_L3,
    "full": LAYER_L1 | LAYER_L2 | LAYER_L3 | LAYER_L4,
    "l4_only": LAYER_L2 | LAYER_L4,
    # not in the reference's matrix: BASELINE.json configs[1] (FastCDC + SHA-256 dedup only, no codec)
    "cdc_dedupe": LAYER_L2 | LAYER_L3,
}


@dataclass(frozen=True)
class IngestConfig:
    min_size: int = 2048          # 2/8/32 KiB keeps the reference's min=avg/4, max=4*avg
    avg_size: int = 8192
    max_size: int = 32768
    norm_level: int = 2
    seg_size: int = 4 << 20
    n_hashes: int = 128           # README.md:2575
    shingle: int = 4              # README.md:2584-2586
    seed_base: int = 0            # README.md:2589
    bands: int = 4                # README.md:1987-1996
    rows: int = 32
    band_bits: int = 16
    level: int = 9                # README.md:2374
    chain_depth: int = 0
    layers: int = LAYER_L1 | LAYER_L2 | LAYER_L3 | LAYER_L4
    delta_max_ratio_pct: int = 0  # README.md:2175's optional 20 % gate

    @staticmethod
    def reference_preset() -> "IngestConfig":
        """The literal 1/4/16 KiB constants of README.md:2444-2446."""
        return IngestConfig(min_size=1024, avg_size=4096, max_size=16384)

    def with_(self, **kw) -> "IngestConfig":
        return replace(self, **kw)

    def to_c(self) -> HmseCfg:
        import ctypes
        c = HmseCfg()
        c.struct_size = ctypes.sizeof(HmseCfg)
        for k, v in asdict(self).items():
            setattr(c, k, int(v))
        return c
