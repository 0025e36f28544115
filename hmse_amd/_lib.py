"""ctypes loader for the C-ABI libraries built in hmse_amd/csrc (include/hmse.h).

The HIP library is the product: if it is missing the import FAILS LOUDLY — there is no CPU
fallback anywhere in this package (the CPU oracle lives in oracle/ and is test infrastructure).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
# HMSE_LIB_VARIANT=<tag> (diagnostics / A-B experiments only): load csrc/libhmse_hip_<tag>.so instead — e.g. the stamps or diag build
HIP_LIB_PATH = os.path.join(_CSRC, "libhmse_hip" + ("_" + os.environ["HMSE_LIB_VARIANT"] if os.environ.get("HMSE_LIB_VARIANT") else "") + ".so")
CORPUS_LIB_PATH = os.path.join(_CSRC, "libhmse_corpus.so")


class HmseCfg(C.Structure):
    """Mirror of `hmse_cfg` (include/hmse.h)."""

    _fields_ = [(n, C.c_uint32) for n in (
        "struct_size", "min_size", "avg_size", "max_size", "norm_level", "seg_size",
        "n_hashes", "shingle", "seed_base", "bands", "rows", "band_bits",
        "level", "chain_depth", "layers", "delta_max_ratio_pct")]


class HmseGl4(C.Structure):
    """Mirror of `hmse_gl4` (include/hmse.h): the global-L4 arrays of a multi-rank stream (device pointers and capacities)."""

    _fields_ = [("struct_size", C.c_uint32), ("world", C.c_uint32), ("rank", C.c_uint32), ("reserved", C.c_uint32),
                ("sig_cap", C.c_uint64), ("max_stored_g", C.c_uint64), ("gstate", C.c_void_p), ("sig_g", C.c_void_p), ("band_keys_g", C.c_void_p),
                ("base_g", C.c_void_p), ("lsh_tables_g", C.c_void_p), ("lsh_slots_g", C.c_uint64), ("g_owner", C.c_void_p), ("g_local", C.c_void_p),
                ("ug", C.c_void_p), ("base_global", C.c_void_p), ("req_counts", C.c_void_p), ("req_slots", C.c_void_p), ("ghost_chunk0", C.c_uint64)]


def build(force: bool = False) -> None:
    """Compile every HIP extension for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    args = ["make", "-C", _CSRC, "-s", "-j8"]
    if force:
        subprocess.check_call(["make", "-C", _CSRC, "-s", "clean"])
    subprocess.check_call(args)


_hip = None
_corpus = None

_VP, _U64, _U32, _SZ = C.c_void_p, C.c_uint64, C.c_uint32, C.c_size_t


def hip_lib():
    global _hip
    if _hip is None:
        if not os.path.exists(HIP_LIB_PATH):
            raise ImportError(
                f"{HIP_LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hmse_amd has no CPU fallback)")
        # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64 (same SONAME as
        # /opt/rocm's).  Import torch FIRST so that the C-ABI library binds to the runtime that owns the
        # tensors and streams it will be handed; loaded the other way round, the process ends up with
        # two runtimes and every launch fails with a HIP error.
        import torch  # noqa: F401
        L = C.CDLL(HIP_LIB_PATH)
        cfgp = C.POINTER(HmseCfg)
        L.hmse_cfg_default.argtypes = [cfgp]
        L.hmse_cfg_validate.argtypes = [cfgp]
        L.hmse_cfg_validate.restype = C.c_int
        L.hmse_abi_version.restype = C.c_int
        L.hmse_strerror.restype = C.c_char_p
        L.hmse_strerror.argtypes = [C.c_int]
        L.hmse_gear_table.argtypes = [_VP]
        L.hmse_workspace_bytes.restype = _SZ
        L.hmse_workspace_bytes.argtypes = [C.c_int, _U64, cfgp]
        L.hmse_l2_cdc.restype = C.c_int
        L.hmse_l2_cdc.argtypes = [_VP, _U64, _VP, _U32, cfgp, _VP, _U64, _VP, _VP, _VP, _SZ, _VP]
        L.hmse_l3_sha256.restype = C.c_int
        L.hmse_l3_sha256.argtypes = [_VP, _U64, _VP, _U64, _VP, _VP, _SZ, _VP]
        L.hmse_l3_dedup.restype = C.c_int
        L.hmse_l3_dedup.argtypes = [_VP, _U64, _VP, _VP, _VP, _SZ, _VP]
        L.hmse_l3_index_slots.restype = _U64
        L.hmse_l3_index_slots.argtypes = [_U64]
        L.hmse_l3_index_update.restype = C.c_int
        L.hmse_l3_index_update.argtypes = [_VP, _U64, _U64, _VP, _VP, _VP, _U64, _VP]
        L.hmse_l4_lsh_slots.restype = _U64
        L.hmse_l4_lsh_slots.argtypes = [_U64]
        L.hmse_l4_lsh_update.restype = C.c_int
        L.hmse_l4_lsh_update.argtypes = [_VP, _U64, _U64, cfgp, _VP, _VP, _VP, _U64, _U32, _VP]
        L.hmse_l4_minhash.restype = C.c_int
        L.hmse_l4_minhash.argtypes = [_VP, _U64, _VP, _VP, _U64, cfgp, _VP, _VP, _SZ, _VP]
        L.hmse_l4_lsh.restype = C.c_int
        L.hmse_l4_lsh.argtypes = [_VP, _U64, cfgp, _VP, _VP, _VP, _SZ, _VP]
        L.hmse_l1_deflate.restype = C.c_int
        L.hmse_l1_deflate.argtypes = [_VP, _U64, _VP, _VP, _VP, _U64, cfgp, _VP, _U64, _VP, _VP, _VP, _VP, _SZ, _VP]
        L.hmse_l1_deflate_ex.restype = C.c_int
        L.hmse_l1_deflate_ex.argtypes = [_VP, _U64, _VP, _VP, _VP, _U64, cfgp, _U32, _VP, _U64, _VP, _VP, _VP, _VP, _SZ, _VP]
        L.hmse_l1_deflate_record_bytes.restype = _U64
        L.hmse_l1_deflate_record_bytes.argtypes = [_U32]
        L.hmse_l1_deflate_record_bytes_dict.restype = _U64
        L.hmse_l1_deflate_record_bytes_dict.argtypes = [_U32]
        L.hmse_l1_inflate_mode.restype = C.c_int
        L.hmse_l1_inflate_mode.argtypes = [C.c_int]
        L.hmse_l1_inflate.restype = C.c_int
        L.hmse_l1_inflate.argtypes = [_VP, _U64, _VP, _VP, _VP, _VP, _U64, _VP, _VP, _U64, _VP, _VP, _VP, _SZ, _VP]
        L.hmse_read_assemble.restype = C.c_int
        L.hmse_read_assemble.argtypes = [_VP, _U64, _VP, _U64, _VP, _VP, _VP, _U64, _VP, _VP]
        L.hmse_stream_batch_workspace_bytes.restype = _U64
        L.hmse_stream_batch_workspace_bytes.argtypes = [_U64, cfgp]
        L.hmse_stream_workspace_init.restype = C.c_int
        L.hmse_stream_workspace_init.argtypes = [_VP, _SZ, _U64, cfgp, _VP]
        L.hmse_stream_batch.restype = C.c_int
        L.hmse_stream_batch.argtypes = [_VP, _U64, _U64, _VP, _U32, cfgp, _VP, _VP, _U64, _VP, _VP, _VP, _VP, _U64, _VP, _U64, _VP, _VP, _VP, _VP,
                                        _U64, _VP, _VP, _VP, _U64, _VP, _SZ, _VP]
        L.hmse_stream_row_bytes.restype = _U64
        L.hmse_stream_row_bytes.argtypes = [_U64, cfgp]
        L.hmse_stream_piece_hash.restype = C.c_int
        L.hmse_stream_piece_hash.argtypes = [_VP, _U64, _U64, _U64, _VP, _U32, cfgp, _VP, _VP, _U64, _VP, _VP, _SZ, _VP]
        L.hmse_stream_piece_encode.restype = C.c_int
        L.hmse_stream_piece_encode.argtypes = [_VP, _U64, _U64, _U64, cfgp, _VP, _VP, _U32, _U32, _VP, _VP, _VP, _U64, _VP, _VP, _VP, _U64, _VP, _U64,
                                               _VP, _VP, _VP, _VP, _U64, _VP, _VP, _VP, _U64, _VP, _SZ, _VP]
        L.hmse_stream_sig_cap.restype = _U64
        L.hmse_stream_sig_cap.argtypes = [_U64, cfgp]
        L.hmse_stream_sig_row_bytes.restype = _U64
        L.hmse_stream_sig_row_bytes.argtypes = [_U64, cfgp]
        L.hmse_stream_piece_sign.restype = C.c_int
        L.hmse_stream_piece_sign.argtypes = [_VP, _U64, _U64, _U64, cfgp, _VP, _VP, _U32, _U32, _VP, _VP, _VP, _U64, _VP, _VP, _VP, _U64, _VP, _U64,
                                             _VP, _VP, _VP, _SZ, _VP]
        L.hmse_stream_piece_bases.restype = C.c_int
        L.hmse_stream_piece_bases.argtypes = [_U64, cfgp, _VP, _VP, C.POINTER(HmseGl4), _VP, _VP, _VP, _VP, _SZ, _VP]
        L.hmse_stream_piece_encode_g.restype = C.c_int
        L.hmse_stream_piece_encode_g.argtypes = [_VP, _U64, _U64, _U64, cfgp, _VP, _VP, _VP, _VP, _VP, _VP, _U64, _VP, _SZ, _VP]
        L.hmse_manifest_pack.restype = C.c_int
        L.hmse_manifest_pack.argtypes = [_VP, _VP, _VP, _VP, _VP, _U64, _VP, _VP, _VP, _U64, _VP, _U64, _U32, _VP, _U32, _VP, _U32, _VP,
                                         _VP, _U64, _VP, _VP, _VP, _U64, _VP, _VP, _SZ, _VP]
        L.hmse_manifest_pack_ex.restype = C.c_int
        L.hmse_manifest_pack_ex.argtypes = [_VP, _VP, _VP, _VP, _VP, _U64, _VP, _VP, _VP, _U64, _VP, _U64, _U32, _VP, _U32, _U32, _VP, _U32, _VP,
                                            _VP, _U64, _VP, _VP, _VP, _U64, _VP, _VP, _SZ, _VP]
        L.hmse_profile_enable.argtypes = [C.c_int]
        L.hmse_profile_read.restype = C.c_int
        L.hmse_profile_read.argtypes = [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.c_int]
        L.hmse_profile_counter.restype = C.c_int
        L.hmse_profile_counter.argtypes = [C.c_int, C.POINTER(C.c_uint64), C.c_int]
        _hip = L
    return _hip


def corpus_lib():
    global _corpus
    if _corpus is None:
        if not os.path.exists(CORPUS_LIB_PATH):
            raise ImportError(f"{CORPUS_LIB_PATH} is missing: run __graft_entry__.build()")
        L = C.CDLL(CORPUS_LIB_PATH)
        L.hmse_corpus_generate.restype = C.c_int
        L.hmse_corpus_generate.argtypes = [_VP, _U64, _U64, _U32, _U64, C.c_int, C.c_int]
        _corpus = L
    return _corpus


EXPORTED_SYMBOLS = (
    "hmse_cfg_default", "hmse_cfg_validate", "hmse_abi_version", "hmse_strerror", "hmse_gear_table",
    "hmse_workspace_bytes", "hmse_l2_cdc", "hmse_l3_sha256", "hmse_l3_dedup", "hmse_l3_index_slots", "hmse_l3_index_update",
    "hmse_l4_lsh_slots", "hmse_l4_lsh_update", "hmse_l4_minhash",
    "hmse_l4_lsh", "hmse_l1_deflate", "hmse_l1_deflate_ex", "hmse_l1_deflate_record_bytes", "hmse_l1_deflate_record_bytes_dict", "hmse_l1_inflate", "hmse_l1_inflate_mode", "hmse_read_assemble", "hmse_manifest_pack", "hmse_manifest_pack_ex", "hmse_stream_batch", "hmse_stream_batch_workspace_bytes", "hmse_stream_workspace_init", "hmse_stream_row_bytes", "hmse_stream_piece_hash", "hmse_stream_piece_encode", "hmse_stream_sig_cap", "hmse_stream_sig_row_bytes", "hmse_stream_piece_sign", "hmse_stream_piece_bases", "hmse_stream_piece_encode_g", "hmse_profile_enable", "hmse_profile_read", "hmse_profile_counter")
