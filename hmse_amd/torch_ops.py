"""`torch.ops.hmse.*`: the stage operators as PyTorch custom ops (SURVEY.md §8b "Callers": the Python driver via
torch.ops.hmse.* — tensors in, tensors out, current HIP stream).  Thin registrations over hmse_amd/ops.py, which is a thin
layer over the C-ABI (include/hmse.h): the configuration travels as plain integers, u64 arrays as int64, u32 as int32.
There is no CPU implementation behind them: a host tensor raises ops.HmseError, exactly like the functions in ops.py."""
from __future__ import annotations

import torch

from . import ops
from .config import IngestConfig

_lib = torch.library.Library("hmse", "DEF")
_lib.define("l2_cdc(Tensor data, int min_size=2048, int avg_size=8192, int max_size=32768, int norm_level=2, int seg_size=4194304) -> Tensor")
_lib.define("l3_sha256(Tensor data, Tensor cuts) -> Tensor")
_lib.define("l3_dedup(Tensor digests) -> (Tensor, Tensor)")
_lib.define("l4_minhash(Tensor data, Tensor cuts, Tensor? chunk_ids=None, int seed_base=0) -> Tensor")
_lib.define("l4_lsh(Tensor sig, int bands=4, int rows=32, int band_bits=16) -> (Tensor, Tensor)")
_lib.define("l1_deflate(Tensor data, Tensor cuts, Tensor? chunk_ids=None, Tensor? base=None, int level=9, int chain_depth=0, "
            "int delta_max_ratio_pct=0, bool base_is_chunk_id=False) -> (Tensor, Tensor, Tensor)")
_lib.define("l1_inflate(Tensor streams, Tensor stream_off, Tensor kind, Tensor? base, Tensor raw_len) -> (Tensor, Tensor, Tensor)")


def _l2_cdc(data, min_size=2048, avg_size=8192, max_size=32768, norm_level=2, seg_size=4 << 20):
    return ops.l2_cdc(data, IngestConfig(min_size=min_size, avg_size=avg_size, max_size=max_size, norm_level=norm_level, seg_size=seg_size))


def _l4_minhash(data, cuts, chunk_ids=None, seed_base=0):
    return ops.l4_minhash(data, cuts, IngestConfig(seed_base=seed_base), chunk_ids)


def _l4_lsh(sig, bands=4, rows=32, band_bits=16):
    return ops.l4_lsh(sig, IngestConfig(bands=bands, rows=rows, band_bits=band_bits))


def _l1_deflate(data, cuts, chunk_ids=None, base=None, level=9, chain_depth=0, delta_max_ratio_pct=0, base_is_chunk_id=False):
    return ops.l1_deflate(data, cuts, IngestConfig(level=level, chain_depth=chain_depth, delta_max_ratio_pct=delta_max_ratio_pct), chunk_ids, base,
                          base_is_chunk_id)


def _l1_inflate(streams, stream_off, kind, base, raw_len):
    return ops.l1_inflate(streams, stream_off, kind, base, raw_len)


# one implementation for every dispatch key: the function itself refuses tensors that do not live in HBM
for _name, _fn in (("l2_cdc", _l2_cdc), ("l3_sha256", ops.l3_sha256), ("l3_dedup", ops.l3_dedup), ("l4_minhash", _l4_minhash),
                   ("l4_lsh", _l4_lsh), ("l1_deflate", _l1_deflate), ("l1_inflate", _l1_inflate)):
    _lib.impl(_name, _fn, "CompositeExplicitAutograd")
