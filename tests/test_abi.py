"""CPU: the C-ABI library loads and exports every symbol include/hmse.h declares; host-side entry points
(config, sizing, Gear table, error strings) behave; the device ops refuse to run without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "hmse.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(hmse_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from hmse_amd import _lib
    lib = _lib.hip_lib()
    names = declared_functions()
    assert {"hmse_l2_cdc", "hmse_l3_sha256", "hmse_l3_dedup", "hmse_l4_minhash", "hmse_l4_lsh", "hmse_l1_deflate",
            "hmse_workspace_bytes", "hmse_cfg_default", "hmse_cfg_validate"} <= set(names)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/hmse.h but not exported"
    assert set(_lib.EXPORTED_SYMBOLS) == set(names)
    assert lib.hmse_abi_version() == 2


def test_library_exports_nothing_but_the_declared_abi():
    """exported == declared: no diagnostic hook or probe entry point ships in the product library."""
    import subprocess
    from hmse_amd import _lib
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.HIP_LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if ln.split()[1:2] and ln.split()[1] in "TW"}
    c_abi = sorted(s for s in exported if not s.startswith("_"))
    undeclared = [s for s in c_abi if s not in set(declared_functions()) and not s.startswith("__hip")]
    assert undeclared == [], undeclared
    assert [s for s in c_abi if s.startswith("hmsedbg")] == []
    src = "".join(open(os.path.join(ROOT, "hmse_amd", "csrc", f)).read() for f in ("l4_minhash.hip", "l1_inflate.hip"))
    assert src.count("#ifdef HMSE_DIAG") >= 3 and "getenv" not in re.sub(r"#ifdef HMSE_DIAG.*?#e(lse|ndif)", "", src, flags=re.S)


def test_deflate_record_bytes_formula_matches_the_abi():
    """ops.record_bytes (evaluated on the device for a whole shard) == hmse_l1_deflate_record_bytes[_dict] for every chunk
    length; the FULL stream (copied out in whole 16-byte stores) fits the token area it overwrites, the DELTA slot behind
    the tokens is 16-byte aligned and leaves 16 bytes behind the largest stream."""
    import torch
    from hmse_amd import _lib, ops
    lib = _lib.hip_lib()
    lens = torch.arange(0, 32769, dtype=torch.int64)
    got = ops.record_bytes(lens).numpy()
    got_d = ops.record_bytes(lens, torch.ones_like(lens, dtype=torch.bool)).numpy()
    mixed = ops.record_bytes(lens, lens % 2 == 1).numpy()
    want = np.array([lib.hmse_l1_deflate_record_bytes(int(v)) for v in range(32769)], dtype=np.int64)
    want_d = np.array([lib.hmse_l1_deflate_record_bytes_dict(int(v)) for v in range(32769)], dtype=np.int64)
    assert np.array_equal(got, want) and np.array_equal(got_d, want_d)
    assert np.array_equal(mixed, np.where(np.arange(32769) % 2 == 1, want_d, want))
    L = lens.numpy()
    body = 1296 + 4 * ((L + 3) & ~3)
    assert (body % 16 == 0).all() and (want >= body).all() and (want % 256 == 0).all() and (want_d % 256 == 0).all()
    assert (((L[1:] + 5 + 15) & ~15) <= body[1:] - 1296).all()          # FULL stream over the tokens, 16-byte stores
    assert (want_d - body - (L + 5) >= 16).all()                        # DELTA slot
    assert lib.hmse_l1_deflate_record_bytes(32769) == 0 and lib.hmse_l1_deflate_record_bytes_dict(32769) == 0


def test_cfg_default_matches_oracle_and_dataclass(orc):
    from hmse_amd import IngestConfig, _lib
    lib = _lib.hip_lib()
    c = _lib.HmseCfg()
    lib.hmse_cfg_default(C.byref(c))
    o = orc.default_cfg()
    d = IngestConfig().to_c()
    assert C.sizeof(c) == 64 == c.struct_size
    for name, _ in _lib.HmseCfg._fields_:
        assert getattr(c, name) == getattr(o, name) == getattr(d, name), name
    assert lib.hmse_cfg_validate(C.byref(c)) == 0


@pytest.mark.parametrize("kw", [dict(min_size=32), dict(avg_size=6000), dict(max_size=65536), dict(n_hashes=64), dict(shingle=5),
                                dict(bands=5), dict(level=10), dict(seg_size=1024), dict(norm_level=9)])
def test_cfg_validate_rejects(kw):
    from hmse_amd import IngestConfig, _lib
    c = IngestConfig(**kw).to_c()
    assert _lib.hip_lib().hmse_cfg_validate(C.byref(c)) == -1  # HMSE_EINVAL, like miniz/mbedtls negative status codes


def test_reference_preset_and_ablations():
    from hmse_amd import ABLATIONS, IngestConfig, _lib
    p = IngestConfig.reference_preset()                      # README.md:2444-2446
    assert (p.min_size, p.avg_size, p.max_size) == (1024, 4096, 16384)
    assert _lib.hip_lib().hmse_cfg_validate(C.byref(p.to_c())) == 0
    assert set(ABLATIONS) == {"l1_only", "l1_cdc", "l1_cdc_dedupe", "full", "l4_only", "cdc_dedupe"}  # VALIDATION_METHODS.md:458-464 + BASELINE configs[1]
    assert ABLATIONS["full"] == 15


def test_gear_table_and_sizing_host_side(orc):
    from hmse_amd import IngestConfig, _lib, ops
    t = np.zeros(256, np.uint64)
    _lib.hip_lib().hmse_gear_table(t.ctypes.data)
    assert np.array_equal(t, orc.gear_table())
    cfg = IngestConfig()
    for stage in (2, 3, 4, 5, 6, 7):
        assert ops.workspace_bytes(stage, 1 << 20, cfg) > 0
    assert ops.workspace_bytes(2, 1 << 30, cfg) > ops.workspace_bytes(2, 1 << 20, cfg)
    assert ops.workspace_bytes(99, 1, cfg) == 0
    assert _lib.hip_lib().hmse_strerror(-2) == b"buffer or workspace too small"


def test_product_path_fails_loudly_without_gpu():
    """No CPU fallback anywhere in hmse_amd: host tensors are refused."""
    import torch
    from hmse_amd import IngestConfig, ops
    x = torch.zeros(1000, dtype=torch.uint8)
    with pytest.raises(ops.HmseError):
        ops.l2_cdc(x, IngestConfig())
    with pytest.raises(ops.HmseError):
        ops.l3_sha256(x, torch.tensor([0, 1000]))
    with pytest.raises(ops.HmseError):
        ops.l4_minhash(x, torch.tensor([0, 1000]), IngestConfig())


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "hmse_amd")
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".h", ".c", ".cpp")):
                src = open(os.path.join(dp, fn), errors="ignore").read()
                assert "import oracle" not in src and "from oracle" not in src and "liborc" not in src, fn


def test_torch_ops_are_registered_and_have_no_cpu_path():
    """the stage operators are reachable as torch.ops.hmse.* (tensors in, tensors out); like ops.py they
    refuse host tensors — there is no CPU implementation behind the registration."""
    import torch
    from hmse_amd import ops, torch_ops  # noqa: F401  (importing registers the library)
    for name in ("l2_cdc", "l3_sha256", "l3_dedup", "l4_minhash", "l4_lsh", "l1_deflate", "l1_inflate"):
        assert hasattr(torch.ops.hmse, name), name
    assert "min_size" in str(torch.ops.hmse.l2_cdc.default._schema)
    x = torch.zeros(5000, dtype=torch.uint8)
    with pytest.raises(ops.HmseError):
        torch.ops.hmse.l2_cdc(x)
    with pytest.raises(ops.HmseError):
        torch.ops.hmse.l3_sha256(x, torch.tensor([0, 5000]))


def test_inflate_mode_knob_and_minhash_workspace():
    """hmse_l1_inflate_mode accepts 0/1/2 only; the MinHash workspace carries the 2 MiB memo table (include/hmse.h)."""
    from hmse_amd import IngestConfig, _lib, ops
    lib = _lib.hip_lib()
    assert lib.hmse_l1_inflate_mode(3) == -1 and lib.hmse_l1_inflate_mode(-1) == -1
    for m in (1, 2, 0):
        assert lib.hmse_l1_inflate_mode(m) == 0
    assert ops.workspace_bytes(ops.STAGE_MINHASH, 1000, IngestConfig()) == 256 + (8 << 18)
