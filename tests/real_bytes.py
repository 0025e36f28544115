"""TEST INFRASTRUCTURE: bytes that are NOT this build's generator, gathered at run time from the box the test runs on (the
reference's corpora are absent and cannot be fetched: VALIDATION_METHODS.md:227-230): Python standard-library sources,
ROCm / system C headers, shared objects (ELF: tables, long zero runs, machine code), docs with UTF-8 multibyte text.
Deterministic for a given image (sorted walks).  Nothing is read from /root/reference."""
import os

import numpy as np


def _walk(roots, exts, limit, min_size=512):
    out, tot = [], 0
    for root in roots:
        for dp, dn, fn in os.walk(root):
            dn.sort()
            for f in sorted(fn):
                if not f.endswith(exts):
                    continue
                p = os.path.join(dp, f)
                try:
                    if os.path.islink(p) or not os.path.isfile(p):
                        continue
                    b = open(p, "rb").read()
                except OSError:
                    continue
                if len(b) < min_size:
                    continue
                out.append(b); tot += len(b)
                if tot >= limit:
                    return out
    return out


def gather(total: int = 64 << 20) -> np.ndarray:
    """>= `total` bytes when the image holds them (it does: python3.10 + /opt/rocm); the mix is ~40 % Python sources, ~35 % C/C++
    headers, ~15 % ELF shared objects, ~10 % docs."""
    parts = []
    parts += _walk(["/usr/lib/python3.10", "/usr/lib/python3/dist-packages"], (".py",), total * 2 // 5)
    parts += _walk(["/opt/rocm/include", "/usr/include"], (".h", ".hpp"), total * 7 // 20)
    elf = _walk(["/usr/lib/x86_64-linux-gnu", "/opt/rocm/lib"], (".so", ".so.1", ".so.6"), total * 3 // 20, min_size=200_000)
    parts += [b[: 6 << 20] for b in elf]                      # a slice of each: headers, tables, code
    parts += _walk(["/usr/share/doc", "/usr/share/common-licenses", "/usr/share/i18n", "/usr/share/perl5"], ("", ), total // 10)
    blob = b"".join(parts)
    if len(blob) < total:                                      # a smaller image: whatever else Python ships
        blob += b"".join(_walk(["/usr/local/lib/python3.10/dist-packages"], (".py", ".txt", ".md", ".json"), total - len(blob)))
    return np.frombuffer(blob[:total], dtype=np.uint8).copy()
