"""Static audit of the shipped code objects for the hazard that has hung the GPU three times.

    python tools/isa_audit.py [path/to/libhmse_hip.so] [--json]

CPU only.  (1) Takes the gfx950 code objects out of the library's `.hip_fatbin` section (clang offload bundles), disassembles them with
/opt/rocm/llvm/bin/llvm-objdump, builds every kernel's control-flow graph and finds its DIVERGENT loops — a natural loop whose latch branch
is conditional on EXEC (`s_cbranch_execnz` / `s_cbranch_execz`), or which an EXEC-conditional branch leaves (the structuriser's form of a
loop that lanes leave at different times: EXEC is narrowed from iteration to iteration) — and lists every CROSS-LANE instruction inside one:
`ds_bpermute_b32` / `ds_permute_b32` / `ds_swizzle_b32`, `v_readlane_b32`, `v_readfirstlane_b32`, `v_writelane_b32`, `v_permlane*`,
and DPP row / wave operations.  In such a loop a cross-lane read may find its source lane switched off: a `ds_bpermute` from an
inactive lane returns 0, `v_readfirstlane` reads another lane than the source meant — the inflate kernel's endless loop of round 1
and the dictionary queue's of round 3.  A finding is not a bug by itself (the compiler's own waterfall loops are of this shape); the
pinned list in tests/golden/isa_audit.json says, per kernel and instruction, how many there are and why each is safe, and
tests/test_host.py fails when the audit finds anything that is not pinned.
(2) Lists the HIP runtime entry points the library imports: a captured chain must consist of kernel nodes only (memset /
memcpy nodes replayed with garbage arguments from the graph's second launch on), so no *Async memset / memcpy may be imported at all."""
import json
import os
import re
import struct
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/llvm/bin"
CROSS = ("ds_bpermute_b32", "ds_permute_b32", "ds_swizzle_b32", "v_readlane_b32", "v_readfirstlane_b32", "v_writelane_b32")
DPP_RE = re.compile(r"\b(row_shr|row_shl|row_ror|row_bcast|row_mirror|row_half_mirror|row_newbcast|row_share|row_xmask|wave_shr|wave_shl|wave_ror|wave_rol|quad_perm)\b")
LINE_RE = re.compile(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):")
TGT_RE = re.compile(r"<([^>+]+)\+0x([0-9a-fA-F]+)>\s*$")
FUNC_RE = re.compile(r"^([0-9a-f]+) <(.+)>:$")


def code_objects(lib_path):
    """The gfx950 ELF images inside the library's .hip_fatbin section."""
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat.bin")
        subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", lib_path, os.path.join(td, "copy.so")], check=True)
        d = open(fat, "rb").read()
    out, pos = [], 0
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    while True:
        i = d.find(magic, pos)
        if i < 0:
            break
        n = struct.unpack_from("<Q", d, i + 24)[0]
        p = i + 32
        for _ in range(n):
            off, size, idl = struct.unpack_from("<QQQ", d, p)
            p += 24
            ident = d[p:p + idl].decode()
            p += idl
            if size and "gfx950" in ident:
                out.append(d[i + off:i + off + size])
        pos = i + len(magic)
    return out


def demangle(names):
    r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout.splitlines()
    short = []
    for s in r:
        s = re.sub(r"^void ", "", s)
        s = re.sub(r"\((?:[^()]|\([^()]*\))*\)$", "", s)      # drop the argument list
        short.append(s)
    return dict(zip(names, short))


def disassemble(elf_bytes):
    with tempfile.NamedTemporaryFile(suffix=".elf") as f:
        f.write(elf_bytes); f.flush()
        txt = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--mcpu=gfx950", f.name], capture_output=True, text=True, check=True).stdout
    funcs, cur, base = {}, None, 0
    for ln in txt.splitlines():
        m = FUNC_RE.match(ln)
        if m:
            base = int(m.group(1), 16)
            cur = funcs.setdefault(m.group(2), [])
            continue
        m = LINE_RE.match(ln)
        if not m or cur is None:
            continue
        mn, ops, addr = m.group(1), m.group(2), int(m.group(3), 16)
        t = TGT_RE.search(ln)
        tgt = int(t.group(2), 16) if (t and mn.startswith(("s_cbranch", "s_branch"))) else None
        cur.append((addr - base, mn, ops, tgt))
    return funcs


def audit_function(ins):
    """-> (findings, loops, divergent loops); a finding = (offset, mnemonic, operands, (loop header offset, latch offset)) for a cross-lane
    instruction inside a divergent NATURAL loop of the kernel's control-flow graph (back edges by dominators: the code layout places loop
    blocks in front of their header, so address ranges would not do)."""
    n = len(ins)
    if n == 0:
        return [], 0, 0
    idx = {off: i for i, (off, _, _, _) in enumerate(ins)}
    leaders = {0}
    for i, (off, mn, ops, tgt) in enumerate(ins):
        if mn.startswith(("s_cbranch", "s_branch")):
            if tgt in idx:
                leaders.add(idx[tgt])
            if i + 1 < n:
                leaders.add(i + 1)
        elif mn in ("s_endpgm", "s_setpc_b64") and i + 1 < n:
            leaders.add(i + 1)
    starts = sorted(leaders)
    blk_of = {}
    blocks = []
    for b, st in enumerate(starts):
        en = starts[b + 1] if b + 1 < len(starts) else n
        blocks.append((st, en))
        for i in range(st, en):
            blk_of[i] = b
    nb = len(blocks)
    succ = [[] for _ in range(nb)]
    for b, (st, en) in enumerate(blocks):
        off, mn, ops, tgt = ins[en - 1]
        if mn.startswith("s_cbranch"):
            if tgt in idx:
                succ[b].append(blk_of[idx[tgt]])
            if en < n:
                succ[b].append(blk_of[en])
        elif mn == "s_branch":
            if tgt in idx:
                succ[b].append(blk_of[idx[tgt]])
        elif mn in ("s_endpgm", "s_setpc_b64"):
            pass
        elif en < n:
            succ[b].append(blk_of[en])
    pred = [[] for _ in range(nb)]
    for b in range(nb):
        for t in succ[b]:
            pred[t].append(b)
    # dominators (iterative, bit sets as Python ints)
    full = (1 << nb) - 1
    dom = [full] * nb
    dom[0] = 1
    changed = True
    order = list(range(nb))
    while changed:
        changed = False
        for b in order[1:]:
            d = full
            for q in pred[b]:
                d &= dom[q]
            d |= 1 << b
            if d != dom[b]:
                dom[b] = d
                changed = True
    loops = []   # (header block, latch block, body set)
    for u in range(nb):
        for h in succ[u]:
            if (dom[u] >> h) & 1:          # h dominates u: back edge
                body = {h, u}
                stack = [u]
                while stack:
                    x = stack.pop()
                    if x == h:
                        continue
                    for q in pred[x]:
                        if q not in body:
                            body.add(q); stack.append(q)
                loops.append((h, u, body))
    EXECB = ("s_cbranch_execnz", "s_cbranch_execz")
    div = []
    for h, u, body in loops:
        latch_mn = ins[blocks[u][1] - 1][1]
        divergent = latch_mn in EXECB
        if not divergent:
            for b in body:
                if ins[blocks[b][1] - 1][1] in EXECB and any(t not in body for t in succ[b]):
                    divergent = True
                    break
        if divergent:
            div.append((h, u, body))
    # VGPRs the compiler uses as SGPR spill space (v_writelane destinations): their lane reads and writes ignore EXEC and carry scalars
    spill = {ops.split(",")[0].strip() for _, mn, ops, _ in ins if mn == "v_writelane_b32"}
    found = []
    for i, (off, mn, ops, _) in enumerate(ins):
        cross = mn in CROSS or mn.startswith("v_permlane") or mn.endswith("_dpp") or bool(DPP_RE.search(ops))
        if not cross:
            continue
        if mn == "v_writelane_b32" or (mn == "v_readlane_b32" and ops.split(",")[1].strip() in spill):
            continue
        b = blk_of[i]
        inside = [l for l in div if b in l[2]]
        if inside:
            l = min(inside, key=lambda l: len(l[2]))
            found.append((off, mn, ops, (ins[blocks[l[0]][0]][0], ins[blocks[l[1]][1] - 1][0])))
    return found, len(loops), len(div)


def audit(lib_path):
    report, names = {}, []
    for co in code_objects(lib_path):
        for fn, ins in disassemble(co).items():
            found, n_loops, n_div = audit_function(ins)
            report[fn] = {"instructions": len(ins), "loops": n_loops, "divergent_loops": n_div,
                          "findings": [{"offset": f"{o:#x}", "op": mn, "operands": ops, "loop": [f"{l[0]:#x}", f"{l[1]:#x}"]} for o, mn, ops, l in found]}
            names.append(fn)
    dm = demangle(names)
    return {dm[k]: v for k, v in report.items()}


def summary(report):
    """{kernel: {op: count}} over the kernels with findings — what tests/golden/isa_audit.json pins."""
    out = {}
    for k, v in report.items():
        c = {}
        for f in v["findings"]:
            c[f["op"]] = c.get(f["op"], 0) + 1
        if c:
            out[k] = c
    return out


def imported_hip_calls(lib_path):
    txt = subprocess.run([f"{LLVM}/llvm-readelf", "--dyn-syms", "-W", lib_path], capture_output=True, text=True, check=True).stdout
    return sorted({ln.split()[-1].split("@")[0] for ln in txt.splitlines() if " UND " in ln and re.search(r"\bhip[A-Z]", ln)})


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    lib = args[0] if args else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "hmse_amd", "csrc", "libhmse_hip.so")
    rep = audit(lib)
    if "--json" in sys.argv:
        print(json.dumps({"summary": summary(rep), "hip_imports": imported_hip_calls(lib)}, indent=1))
        sys.exit(0)
    tot = sum(len(v["findings"]) for v in rep.values())
    print(f"{lib}: {len(rep)} kernels, {sum(v['loops'] for v in rep.values())} loops, {sum(v['divergent_loops'] for v in rep.values())} divergent, "
          f"{tot} cross-lane instructions inside divergent loops")
    for k, v in sorted(rep.items()):
        if v["findings"]:
            print(f"\n{k}  ({v['instructions']} instructions, {v['loops']} loops, {v['divergent_loops']} divergent)")
            for f in v["findings"]:
                print(f"   {f['offset']:>8}  {f['op']:22s} {f['operands'][:60]:60s} loop {f['loop'][0]}..{f['loop'][1]}")
    print("\nHIP runtime entry points imported:", " ".join(imported_hip_calls(lib)))
